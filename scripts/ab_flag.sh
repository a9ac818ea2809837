#!/bin/bash
# usage: scripts/ab_flag.sh <module.FLAG=value> <out-prefix> [bench args]  -- bench.py twice alternating: product, then with the test hook set
# (same box, same run).  e.g. scripts/ab_flag.sh ait_amd.faster_rcnn._PAIR_GRADS=False gpurun_out/ab_pair --steps 20 --warmup 5
flag=$1; out=$2; shift 2
mod=${flag%.*}; rest=${flag##*.}; name=${rest%%=*}; val=${rest#*=}
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-ab "$@" > ${out}_prod$i.json 2>/dev/null
  python -c "import sys, runpy, importlib; sys.argv=['bench.py','--no-cpu-baseline','--no-ab']+sys.argv[1:]; m=importlib.import_module('$mod'); setattr(m,'$name',$val); runpy.run_path('bench.py', run_name='__main__')" "$@" > ${out}_flag$i.json 2>/dev/null
done
python - <<PY
import json,glob
for k in ("prod1","flag1","prod2","flag2"):
    l=json.load(open("${out}_%s.json"%k)); print(k, "%.2f pairs/s  median %.3f ms  mean %.3f ms  gemm %.2f ms" % (l["value"], l["ms_per_step"], l["ms_per_step_mean"], l["roofline"]["gemm_ms_per_step"]))
PY
