"""ait_mha_core_fwd against the four launches it replaces, 1200 sequences (bench shapes), ms per block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _lab_lib  # noqa: F401  (AIT_LAB_LIB=<variant> selects a lab build of the library)
import torch
from ait_amd import ops
dev = "cuda"
n = 1200
def timeit(fn, it=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
qkv = torch.randn(n * 64, 1536, device=dev)
sk_w, sk_b = torch.randn(512, 64, device=dev) * 0.3, torch.randn(512, device=dev) * 0.1
fc_w = torch.randn(512, 64, device=dev) * 0.125
res = torch.randn(n * 64, 512, device=dev)
g, b = torch.rand(512, device=dev) + 0.5, torch.randn(512, device=dev) * 0.1
for p, save in ((0.1, True), (0.0, False)):
    def four():
        O, P = ops.attn_fwd(qkv, 0, qkv, 512, qkv, 1024, n, 8, 64, 64, 2, 0, 0.125, p, 5, save_p=save)
        u, gate, s = ops.sh_fwd(O, sk_w, sk_b)
        f = ops.gemm(u.reshape(n * 64, 64), fc_w)
        return ops.ln_fwd(f, None, res, g, b, n * 64, 64, 64, 1, 1e-6, p, 6, save_stats=save)
    def one():
        return ops.mha_core_fwd(qkv, 0, qkv, 512, qkv, 1024, n, 2, 0, p, 5, sk_w, sk_b, fc_w, res, g, b, 1e-6, p, 6, save=save)
    print("%-10s four launches %.3f ms   fused %.3f ms" % ("training" if save else "inference", timeit(four), timeit(one)))
