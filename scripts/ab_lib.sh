#!/bin/bash
# usage: scripts/ab_lib.sh <variant> <out-prefix> [bench args] -- bench.py alternating the product library and scripts/_lab/libait_hip_<variant>.so
# (the variant is copied over ait_amd/libait_hip.so for its runs and the product restored behind them); same box, same run
v=$1; out=$2; shift 2
cp ait_amd/libait_hip.so /tmp/_prod.so
for i in 1 2; do
  cp /tmp/_prod.so ait_amd/libait_hip.so
  python bench.py --no-cpu-baseline --no-ab --no-probe-pass "$@" > ${out}_prod$i.json 2>/dev/null
  cp scripts/_lab/libait_hip_$v.so ait_amd/libait_hip.so
  python bench.py --no-cpu-baseline --no-ab --no-probe-pass "$@" > ${out}_$v$i.json 2>/dev/null
done
cp /tmp/_prod.so ait_amd/libait_hip.so
python - <<PY
import json
for k in ("prod1","${v}1","prod2","${v}2"):
    l=json.load(open("${out}_%s.json"%k)); print(k, "%.2f pairs/s  median %.3f ms  mean %.3f ms  gemm %.2f ms %.1f TF/s" % (l["value"], l["ms_per_step"], l["ms_per_step_mean"], l["roofline"]["gemm_ms_per_step"], l["roofline"]["achieved"]))
PY
