#!/bin/bash
# Builds scripts/_gemm_lab (gfx950 cross-compile).  The round-1 kernel for the A/B comes from the history.
set -e
cd "$(dirname "$0")/.."
if git cat-file -e 81edb57:ait_amd/csrc/gemm_f32_impl.h 2>/dev/null; then
  git show 81edb57:ait_amd/csrc/gemm_f32_impl.h | sed -e 's/namespace ait_gemm/namespace ait_gemm_old/' \
      -e "s|#include \"common.h\"|#include \"$PWD/ait_amd/csrc/common.h\"|" -e 's/#pragma once//' \
      -e 's/lds_void/lds_void_old/g' -e 's/glb_void/glb_void_old/g' > /tmp/ait_old_gemm_f32_impl.h
  OLD=""
else
  OLD="-DNO_OLD"
fi
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize $LAB_CXXFLAGS --offload-arch=gfx950 -std=c++17 -I include $OLD scripts/gemm_lab.hip -o scripts/_gemm_lab -Wno-unused-result 2>&1 | grep -E "error:" -A5 || true
ls -la scripts/_gemm_lab
