"""Driver for PMC passes over the hand-written kernels on the bench shapes (run under
rocprofv3 --pmc <counters>): the four GEMM layouts, attention fwd/bwd, LayerNorm, RoIAlign."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ait_amd import ops
from ait_amd.system import _wgrad
M = 76800
for (m, n, k, ta, tb, sk) in [(M, 1536, 512, False, True, 1), (M, 512, 2048, False, True, 1),
                              (M, 512, 2048, False, False, 1), (512, 2048, M, True, False, 16)]:
    a = torch.randn((k, m) if ta else (m, k), device="cuda")
    b = torch.randn((n, k) if tb else (k, n), device="cuda")
    out = torch.zeros(m, n, device="cuda")
    for _ in range(3):
        ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, split_k=sk)
qkv = torch.randn(M, 1536, device="cuda")
dqkv = torch.empty_like(qkv)
for _ in range(3):
    O, P = ops.attn_fwd(qkv, 0, qkv, 512, qkv, 1024, 1200, 8, 64, 64, 2, 64, 0.125, 0.1, 7)
    ops.attn_bwd(qkv, 0, qkv, 512, qkv, 1024, P, torch.ones_like(O), 1200, 8, 64, 64, 0.125, 0.1, 7,
                 dqkv, 0, dqkv, 512, dqkv, 1024)
torch.cuda.synchronize()
