"""Tiny driver for PMC passes: runs each GEMM shape a few times (run under rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ait_amd import ops
M = 76800
for (m, n, k, ta, tb, sk) in [(M, 1536, 512, False, True, 1), (M, 512, 2048, False, True, 1),
                              (M, 512, 2048, False, False, 1), (512, 2048, M, True, False, 16)]:
    a = torch.randn((k, m) if ta else (m, k), device="cuda")
    b = torch.randn((n, k) if tb else (k, n), device="cuda")
    out = torch.zeros(m, n, device="cuda")
    for _ in range(3):
        ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, split_k=sk)
torch.cuda.synchronize()
