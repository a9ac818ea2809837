#!/bin/bash
# PMC passes over the two fused attention kernels at the bench shape (1200 sequences; run on the GPU box): where the waves'
# cycles go (issue / wait / LDS / VMEM / MFMA), LDS activity and bank conflicts, HBM bytes.  usage: scripts/pmc_mha.sh <outdir>
out=$1; root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for b in bench_mha_core bench_mha_core_bwd; do
    timeout 300 rocprofv3 --pmc $set -d "$root/$out/pass${i}_$b" --output-format csv -- python3 "$root/scripts/$b.py" > "$root/$out/pass${i}_$b.log" 2>&1
  done
done
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"(mha_core_(?:fwd|bwd)_kernel<[^>]*>)", k)
        if not m:
            continue
        a = agg[m.group(1)][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open(sys.argv[1] + "/mha_pmc_summary.txt", "w") as o:
    for k in sorted(agg):
        o.write(k + "\n")
        c = agg[k]
        for n in sorted(c):
            o.write("   %-34s launches %3d  avg per launch %.6g\n" % (n, c[n][1], c[n][0] / c[n][1]))
        g = lambda n: c[n][0] / c[n][1] if n in c and c[n][1] else None
        wc = g("SQ_WAVE_CYCLES")
        if wc:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU"):
                if g(n) is not None:
                    o.write("   => %-22s %.3f of wave cycles\n" % (n, g(n) / wc))
        if g("GRBM_GUI_ACTIVE"):
            cu_cycles = g("GRBM_GUI_ACTIVE") / 8 * 256
            if g("SQ_LDS_IDX_ACTIVE") is not None:
                o.write("   => LDS array active %.3f of kernel cycles x CUs; bank-conflict cycles %.3g (%.3f of active)\n" % (
                    g("SQ_LDS_IDX_ACTIVE") / cu_cycles, g("SQ_LDS_BANK_CONFLICT") or 0, (g("SQ_LDS_BANK_CONFLICT") or 0) / max(1.0, g("SQ_LDS_IDX_ACTIVE"))))
            if g("SQ_WAVES"):
                o.write("   => waves launched %.0f\n" % g("SQ_WAVES"))
        if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("SQ_BUSY_CYCLES"):
            o.write("   => MFMA busy cycles %.4g\n" % g("SQ_VALU_MFMA_BUSY_CYCLES"))
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
            o.write("   => L2 hit rate %.3f\n" % (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
        if g("FETCH_SIZE") is not None:
            o.write("   => FETCH_SIZE %.1f MB raw (x2 for wide streaming reads on gfx950), WRITE_SIZE %s MB\n" % (g("FETCH_SIZE") / 1024, ("%.1f" % (g("WRITE_SIZE") / 1024)) if g("WRITE_SIZE") is not None else "?"))
print(open(sys.argv[1] + "/mha_pmc_summary.txt").read())
PY
