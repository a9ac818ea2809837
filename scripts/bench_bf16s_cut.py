"""microbenchmark: bf16-storage products whose tile count leaves a thin last round (device time from the library's event pair)"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ait_amd import ops, _lib

def run(M, N, K, conv=False, reps=20):
    dev = "cuda"
    if conv:
        n = M // 16
        x = torch.randn(M, 512, device=dev).to(torch.bfloat16)
        w = ops.conv_weight_to_bf16(torch.randn(N, 3, 3, 512, device=dev) * 0.02)
        geom = ops.conv_geom(n, (4, 4), (4, 4), (3, 3), 1, 1)
        f = lambda: ops.conv_fwd_bf16s(x, w, geom, 512, N)
    else:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = torch.randn(N, K, device=dev).to(torch.bfloat16)
        f = lambda: ops.gemm_bf16s(a, b, want32=False, want16=True)
    for _ in range(3):
        f()
    pr = _lib.Probe(4 * reps)
    with pr:
        for _ in range(reps):
            f()
    torch.cuda.synchronize()
    ms = sorted(e[2] for e in pr.entries())
    med = ms[len(ms) // 2]
    print("M=%6d N=%5d K=%5d %s: %7.1f us  %7.1f TF/s" % (M, N, K, "conv" if conv else "    ", med * 1e3, 2.0 * M * N * K / med / 1e9))

for M in (65536, 65792, 66048):
    run(M, 512, 4608, conv=True)
    run(M, 512, 4608)
    run(M, 512, 2048)
    run(M, 2048, 512)
run(200704, 512, 2048)
run(200704, 2048, 512)
