#!/bin/bash
# PMC passes over the RoIAlign kernels on the bench shape (run on the GPU box): L1 (TCP) / L2 (TCC) hit rates, LDS activity,
# wave occupancy of the two channels-last kernels.  usage: scripts/pmc_roi.sh <outdir>
out=$1; root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" \
           "TCC_REQ_sum TCC_READ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_VALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d "$root/$out/pass$i" --output-format csv -- python3 "$root/scripts/bench_roi.py" > "$root/$out/pass$i.log" 2>&1
done
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "roi_align" not in k:
            continue
        import re
        m = re.search(r"(roi_\w+)", k)
        name = m.group(1) if m else k[:60]
        a = agg[name][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open(sys.argv[1] + "/roi_pmc_summary.txt", "w") as o:
    for k in sorted(agg):
        o.write(k + "\n")
        c = agg[k]
        for n in sorted(c):
            o.write("   %-34s launches %3d  avg per launch %.6g\n" % (n, c[n][1], c[n][0] / c[n][1]))
        g = lambda n: c[n][0] / c[n][1] if n in c and c[n][1] else None
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
            o.write("   => L2 hit rate %.3f\n" % (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
        if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
            o.write("   => L1 (TCP): %.4g accesses, %.4g read requests to L2 (%.3f of accesses)\n" % (
                g("TCP_TOTAL_CACHE_ACCESSES_sum"), g("TCP_TCC_READ_REQ_sum"), g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")))
        if g("SQ_LDS_IDX_ACTIVE") is not None and g("GRBM_GUI_ACTIVE"):
            o.write("   => LDS array active %.3f of kernel cycles x CUs (SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 * 256)); bank-conflict cycles %.3g\n" % (
                g("SQ_LDS_IDX_ACTIVE") / (g("GRBM_GUI_ACTIVE") / 8 * 256), g("SQ_LDS_BANK_CONFLICT") or 0))
        if g("FETCH_SIZE") is not None:
            o.write("   => FETCH_SIZE %.1f MB raw (x2 for wide streaming reads on gfx950), WRITE_SIZE %s MB\n" % (g("FETCH_SIZE") / 1024, ("%.1f" % (g("WRITE_SIZE") / 1024)) if g("WRITE_SIZE") is not None else "?"))
print(open(sys.argv[1] + "/roi_pmc_summary.txt").read())
PY
