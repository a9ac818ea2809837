"""Micro-benchmark of ait_gemm_f32 on the AIT shapes (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ait_amd import ops  # noqa: E402


def timeit(fn, n=20, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


bp = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1200
if "bf16" in sys.argv:
    ops.set_matmul_dtype("bf16")
if "bf16x3" in sys.argv:
    ops.set_matmul_dtype("bf16x3")
M = bp * 64
shapes = [("qkv   NT", M, 1536, 512, False, True), ("proj  NT", M, 512, 512, False, True),
          ("ffn1  NT", M, 2048, 512, False, True), ("ffn2  NT", M, 512, 2048, False, True),
          ("fc    NT", M, 512, 64, False, True), ("emb   NT", bp * 49, 512, 1024, False, True),
          ("dgrad NN", M, 512, 2048, False, False), ("dgrad NN", M, 2048, 512, False, False),
          ("wgrad TN", 2048, 512, M, True, False), ("wgrad TN", 512, 512, M, True, False)]
for name, m, n, k, ta, tb in shapes:
    a = torch.randn((k, m) if ta else (m, k), device="cuda")
    b = torch.randn((n, k) if tb else (k, n), device="cuda")
    sk = 1
    if ta:
        sk = max(1, min(64, (256 * 4) // (((m + 127) // 128) * ((n + 127) // 128))))
    out = torch.zeros(m, n, device="cuda")
    ms = timeit(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, split_k=sk))
    A = a.t() if ta else a
    B = b.t() if tb else b
    ms_t = timeit(lambda: torch.matmul(A, B))
    fl = 2.0 * m * n * k
    print("%s M=%6d N=%5d K=%6d splitk=%2d : %7.3f ms %6.1f TF/s (%.0f%% of 157.3) | torch %7.3f ms %6.1f TF/s"
          % (name, m, n, k, sk, ms, fl / ms / 1e9, 100 * fl / ms / 1e9 / 157.3, ms_t, fl / ms_t / 1e9))
