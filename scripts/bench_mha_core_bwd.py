"""ait_mha_core_bwd against the three launches it replaces (fc's input gradient, ait_sh_bwd, ait_attn_bwd), ms per block,
at the sequence counts of the bench configurations (cfg2: 1200, cfg5: 4096)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _lab_lib  # noqa: F401  (AIT_LAB_LIB=<variant> selects a lab build of the library)
import torch
from ait_amd import ops
dev = "cuda"
def timeit(fn, it=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
for n in (1200, 4096):
    qkv = torch.randn(n * 64, 1536, device=dev)
    sk_w, sk_b = torch.randn(512, 64, device=dev) * 0.3, torch.randn(512, device=dev) * 0.1
    fc_w = torch.randn(512, 64, device=dev) * 0.125
    p = 0.1
    O, P = ops.attn_fwd(qkv, 0, qkv, 512, qkv, 1024, n, 8, 64, 64, 2, 0, 0.125, p, 5)
    _, gate, _ = ops.sh_fwd(O, sk_w, sk_b)
    df = torch.randn(n * 64, 512, device=dev)
    dqkv = torch.empty_like(qkv)
    du = torch.empty(n * 64, 64, device=dev)
    def three(parts=None):
        ops.gemm(df, fc_w, trans_b=False, out=du)
        dO, dg = ops.sh_bwd(du.reshape(n, 64, 64), O, gate, sk_w)
        ops.attn_bwd(qkv, 0, qkv, 512, qkv, 1024, P, dO, n, 8, 64, 64, 0.125, p, 5, dqkv, 0, dqkv, 512, dqkv, 1024)
    dO, _ = ops.sh_bwd(du.reshape(n, 64, 64), O, gate, sk_w)
    t_g = timeit(lambda: ops.gemm(df, fc_w, trans_b=False, out=du))
    t_s = timeit(lambda: ops.sh_bwd(du.reshape(n, 64, 64), O, gate, sk_w))
    t_a = timeit(lambda: ops.attn_bwd(qkv, 0, qkv, 512, qkv, 1024, P, dO, n, 8, 64, 64, 0.125, p, 5, dqkv, 0, dqkv, 512, dqkv, 1024))
    def one():
        return ops.mha_core_bwd(df, fc_w, O, gate, sk_w, qkv, 0, qkv, 512, qkv, 1024, P, n, p, 5, dqkv, 0, dqkv, 512, dqkv, 1024)
    print("n = %4d: three launches %.3f ms (fc dgrad %.3f, sh_bwd %.3f, attn_bwd %.3f)   fused %.3f ms" % (n, timeit(three), t_g, t_s, t_a, timeit(one)))
