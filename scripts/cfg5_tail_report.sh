#!/bin/bash
# usage: scripts/cfg5_tail_report.sh <out.txt> -- the bf16 configuration's proposal tail: library node (bf16 storage) against the
# module composition on MIOpen's bf16 convolutions, same box, alternating; the node's products by shape; the bf16 kernel's thin
# last rounds.  One GPU call, a few minutes.
out=$1
{
echo "## (1) bench.py --config cfg5, alternating: tail on MIOpen (A/B hook) | on the library (product)   pairs/s  median ms/step  product TFLOP/s"
for v in miopen library miopen library; do
  timeout 600 python scripts/ab_cfg5_tail.py $v --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-ab 2>/dev/null < /dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-8s %7.2f  %7.3f  %6.1f' % ('$v', d['value'], d['ms_per_step'], d['roofline']['achieved']))"
done
echo
echo "## (2) the tail node alone (scripts/bench_tail16.py: ait_tail_fwd + ait_tail_bwd, 4096 + 8 maps, bf16 product form), products by shape"
timeout 300 python scripts/bench_tail16.py 2>/dev/null < /dev/null
echo
echo "## (3) bf16 product kernel, whole rounds (65536 rows = 256 row tiles) against a thin third round (65792, 66048); scripts/bench_bf16s_cut.py"
timeout 300 python scripts/bench_bf16s_cut.py 2>/dev/null < /dev/null
} > "$out" 2>&1
cat "$out"
