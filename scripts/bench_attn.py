import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ait_amd import ops
n = 1200
qkv = torch.randn(n * 64, 1536, device="cuda"); q = torch.randn(n * 64, 512, device="cuda"); kv49 = torch.randn(n * 49, 1024, device="cuda")
def timeit(fn, k=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / k
O, P = ops.attn_fwd(qkv, 0, qkv, 512, qkv, 1024, n, 8, 64, 64, 2, 0, 0.125, 0.1, 1)
dO = torch.randn_like(O); dqkv = torch.empty_like(qkv); dq = torch.empty_like(q); dkv = torch.empty_like(kv49)
print("self fwd %.3f ms" % timeit(lambda: ops.attn_fwd(qkv, 0, qkv, 512, qkv, 1024, n, 8, 64, 64, 2, 0, 0.125, 0.1, 1)))
print("self bwd %.3f ms" % timeit(lambda: ops.attn_bwd(qkv, 0, qkv, 512, qkv, 1024, P, dO, n, 8, 64, 64, 0.125, 0.1, 1, dqkv, 0, dqkv, 512, dqkv, 1024)))
O2, P2 = ops.attn_fwd(q, 0, kv49, 0, kv49, 512, n, 8, 64, 64, 0, 0, 0.125, 0.1, 1, kv_rows=49)
print("cross49 fwd %.3f ms" % timeit(lambda: ops.attn_fwd(q, 0, kv49, 0, kv49, 512, n, 8, 64, 64, 0, 0, 0.125, 0.1, 1, kv_rows=49)))
print("cross49 bwd %.3f ms" % timeit(lambda: ops.attn_bwd(q, 0, kv49, 0, kv49, 512, P2, dO, n, 8, 64, 64, 0.125, 0.1, 1, dq, 0, dkv, 0, dkv, 512, kv_rows=49)))
