out=gpurun_out/p5; root=$(pwd); mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$root/$out/trace5" --output-format csv -- python3 "$root/bench.py" --config cfg5 --steps 6 --warmup 3 --no-cpu-baseline --no-ab --no-probe-pass > /dev/null 2> "$root/$out/trace5.err"
cd "$root"
python3 scripts/trace_stats.py "$out/trace5" 3 6 "$out/cfg5_stats.csv" > /dev/null 2>&1
python3 scripts/trace_categories.py "$out/cfg5_stats.csv" 6 > "$out/cfg5_categories.txt" 2>&1
rm -rf "$out/trace5"
cat $out/cfg5_categories.txt
