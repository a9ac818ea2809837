#!/bin/bash
# Samples the GPU's shader clock / power while (a) the register-only MFMA loop and (b) the AIT GEMM
# microbenchmark run: is the fp32 GEMM clock/power-limited?
cd "$(dirname "$0")/.."
sample() { for i in 1 2 3 4 5 6; do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power" | tr "\n" " "; echo; sleep 0.5; done; }
echo "== idle"; sample | head -2
[ -x scripts/_mfma_peak ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/mfma_peak.hip -o scripts/_mfma_peak
echo "== register-only MFMA loop"; (for i in 1 2 3 4 5 6 7 8; do ./scripts/_mfma_peak > /dev/null; done) & sleep 1; sample; wait
echo "== ait_gemm_f32 microbenchmark"; (python scripts/bench_gemm.py 1200 > /tmp/bg.txt 2>&1; python scripts/bench_gemm.py 1200 >> /tmp/bg.txt 2>&1) & sleep 6; sample; wait; tail -3 /tmp/bg.txt
