#!/bin/bash
# which bf16-storage kernel instantiations a script launches (rocprofv3 kernel names)
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tt
timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/tt --output-format csv -- python3 "$root/scripts/bench_tail16.py" 4096 8 > /tmp/tt.out 2> /tmp/tt.err < /dev/null
f=$(find /tmp/tt -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -z "$f" ]; then echo "no stats file"; tail -5 /tmp/tt.err; exit 0; fi
grep "gemm_bf16s_kernel" "$f" | cut -c1-230 | head -30
