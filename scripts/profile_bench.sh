#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box: kernel trace + stats of the timed region, FETCH_SIZE / WRITE_SIZE
# PMC passes (separate runs).  usage: scripts/profile_bench.sh <outdir> <tag>
out=$1; tag=$2; root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$root/$out/trace" --output-format csv -- python3 "$root/bench.py" --steps 10 --warmup 5 --no-cpu-baseline --no-ab --no-probe-pass > "$root/$out/bench_under_prof.json" 2> "$root/$out/trace.err"
cd "$root"
python3 scripts/trace_stats.py "$out/trace" 5 10 "$out/${tag}_timed_region_kernel_stats.csv" > /dev/null 2>&1
python3 scripts/trace_categories.py "$out/${tag}_timed_region_kernel_stats.csv" 10 > "$out/${tag}_categories.txt" 2>&1
python3 scripts/trace_gaps.py "$out/trace" 5 10 20 >> "$out/${tag}_categories.txt" 2>&1
python3 scripts/trace_timeline.py "$out/trace" 5 10 > "$out/${tag}_step_timeline.txt" 2>&1
cp "$(find "$out/trace" -name "*kernel_stats.csv" -printf "%s %p\n" | sort -rn | head -1 | cut -d" " -f2-)" "$out/${tag}_full_run_kernel_stats.csv" 2>/dev/null   # the largest: bench.py's own process, not the mfma_peak child
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c -d "$root/$out/pmc_$c" --output-format csv -- python3 "$root/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-ab --no-probe-pass > /dev/null 2> "$root/$out/pmc_$c.err"
done
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d "$root/$out/pmc_clock" --output-format csv -- python3 "$root/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-ab --no-probe-pass > /dev/null 2> "$root/$out/pmc_clock.err"
cd "$root"
python3 scripts/pmc_clock.py "$out/pmc_clock" gemm_f32_stream > "$out/${tag}_gemm_mfma_util_and_clock.txt" 2>&1
python3 scripts/pmc_clock.py "$out/pmc_clock" gemm_f32_stream list | awk 'f && $3 > 100; /effective clock/ {f=1}' | tail -330 > "$out/${tag}_effective_clock_per_dispatch.txt" 2>&1
rm -rf "$out/pmc_clock"
python3 scripts/pmc_summary.py "$out/pmc_FETCH_SIZE" "$out/pmc_fetch_${tag}.csv" > /dev/null 2>&1
python3 scripts/pmc_summary.py "$out/pmc_WRITE_SIZE" "$out/pmc_write_${tag}.csv" > /dev/null 2>&1
rm -rf "$out/trace" "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE"
cat "$out/${tag}_categories.txt"
