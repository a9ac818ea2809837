#!/bin/bash
# The round's committed measurement set in one GPU call: usage scripts/final_profiles.sh <outdir> <tag>  (tag = r05 ...)
#   bench line (the driver's command), rocprofv3 passes over it (scripts/profile_bench.sh), GEMM-by-shape tables,
#   the cfg5 / cfg3 lines, cfg5's kernel categories.  Copy what is to be judged from <outdir> into profiles/.
out=$1; tag=$2; root=$(pwd)
mkdir -p "$out"
python bench.py --steps 20 --warmup 5 > "$out/${tag}_bench_line.json" 2> "$out/bench.err"
bash scripts/profile_bench.sh "$out" "$tag" > /dev/null 2>&1
mv "$out/bench_under_prof.json" "$out/${tag}_bench_under_rocprof.json"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ab --gemm-table "$out/${tag}_gemm_by_shape.txt" > /dev/null 2>> "$out/bench.err"
python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-ab --gemm-table "$out/${tag}_cfg5_gemm_by_shape.txt" > "$out/${tag}_cfg5_bench_line.json" 2>> "$out/bench.err"
python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-ab > "$out/${tag}_cfg3_bench_line.json" 2>> "$out/bench.err"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$root/$out/trace5" --output-format csv -- python3 "$root/bench.py" --config cfg5 --steps 6 --warmup 3 --no-cpu-baseline --no-ab --no-probe-pass > /dev/null 2> "$root/$out/trace5.err"
cd "$root"
python3 scripts/trace_stats.py "$out/trace5" 3 6 "$out/${tag}_cfg5_timed_region_kernel_stats.csv" > /dev/null 2>&1
python3 scripts/trace_categories.py "$out/${tag}_cfg5_timed_region_kernel_stats.csv" 6 > "$out/${tag}_cfg5_categories.txt" 2>&1
rm -rf "$out/trace5"
head -c 600 "$out/${tag}_bench_line.json"; echo; head -c 300 "$out/${tag}_cfg5_bench_line.json"; echo; head -c 300 "$out/${tag}_cfg3_bench_line.json"; echo
ls "$out"
