#!/bin/bash
# PMC passes over one lab GEMM variant (run on the GPU box): scripts/pmc_lab.sh <shape> <variant> <outdir>
shape=$1; variant=$2; out=$3
root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d "$root/$out/pass$i" --output-format csv -- "$root/scripts/_gemm_lab" pmc "$shape" "$variant" 6 > "$root/$out/pass$i.log" 2>&1
done
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f32" not in r["Kernel_Name"]:
            continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open(sys.argv[1] + "/summary.txt", "w") as o:
    for k in sorted(agg):
        o.write("%-34s launches %3d  avg %.4g\n" % (k, agg[k][1], agg[k][0] / agg[k][1]))
print(open(sys.argv[1] + "/summary.txt").read())
PY
