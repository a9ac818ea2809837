#!/bin/bash
# MfmaUtil and effective clock of the bf16-storage product kernels in the bf16 configuration's step: usage scripts/pmc_cfg5.sh <out.txt>
# (rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace over bench.py --config cfg5; scripts/pmc_clock.py)
out=$1; root=$(pwd); d=/tmp/pmc5
rm -rf $d
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $d --output-format csv -- python3 "$root/bench.py" --config cfg5 --steps 3 --warmup 2 --no-cpu-baseline --no-ab --no-probe-pass > /dev/null 2> /tmp/pmc5.err < /dev/null
cd "$root"
{
echo "# bench.py --config cfg5 under rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace (scripts/pmc_cfg5.sh): the bf16-storage kernels"
timeout 120 python3 scripts/pmc_clock.py $d gemm_bf16s < /dev/null
} > "$out" 2>&1
rm -rf $d
cat "$out" | cut -c1-230
