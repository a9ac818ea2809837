#!/bin/bash
# LAB: same-box A/B of library variants on bench.py's step: usage scripts/ab_bench.sh <outdir> <variant> [<variant> ...]
# ("" = the product library); alternates the variants twice, 20 steps each, writes the GEMM-by-shape tables.
out=$1; shift
mkdir -p "$out"
for rep in 1 2; do
  for v in "$@"; do
    tag=${v:-product}
    AIT_LAB_LIB=$v python scripts/bench_lab.py --steps 20 --warmup 5 --no-cpu-baseline --no-ab --gemm-table "$out/gemm_${tag}_$rep.txt" > "$out/bench_${tag}_$rep.json" 2> "$out/bench_${tag}_$rep.err"
    python - "$out/bench_${tag}_$rep.json" "$tag" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-10s %.2f pairs/s  %.2f ms/step  GEMM %.2f ms/step %.1f TFLOP/s" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["gemm_ms_per_step"], d["roofline"]["achieved"]))
PY
  done
done
