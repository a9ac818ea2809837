"""The proposal tail (SK block at stride 2 -> layer4 -> mean) in isolation on the bench shapes, forward and
forward + backward, with the per-shape table of the library's GEMM launches (live HIP events): what SURVEY 8(f1)
is measured by.   python scripts/time_top.py [bp]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ait_amd import _lib, tuning

tuning.use_tuned_miopen_db(0)
import bench

bp = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
dev = torch.device("cuda:0")
m = bench.build_model(300, dev).train()


def timeit(fn, n=10, w=3):
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


x = torch.randn(bp, 1024, 8, 8, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
q = torch.randn(4, 1024, 8, 8, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)


import ait_amd.faster_rcnn as fr


def tail(train=True):
    def f():
        if fr._TAIL_FUSED:
            y, z = m._tail(x, q)
        else:
            a, b = m.sk(x_props=x, x_query=q, stride=2)
            y = m._head_to_tail(a, subsampled=True)
            z = m._head_to_tail(b, subsampled=True)
        if train:
            (y.sum() + z.sum()).backward()
    return f


def sk_only():
    a, b = m.sk(x_props=x, x_query=q, stride=2)
    (a.sum() + b.sum()).backward()


x4 = torch.randn(bp, 1024, 4, 4, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)


def l4_only():
    m._head_to_tail(x4, subsampled=True).sum().backward()


print("bp = %d" % bp)
for fused in (True, False):
    fr._TAIL_FUSED = fused
    print("%s: tail forward %.3f ms, forward + backward %.3f ms"
          % ("ait_tail_* (one node)" if fused else "nn.Module composition (PyTorch-ROCm convolutions)", timeit(tail(False)), timeit(tail(True))))
fr._TAIL_FUSED = True
print("  module composition, SK only   fwd + bwd     %.3f ms" % timeit(sk_only))
print("  module composition, layer4    fwd + bwd     %.3f ms" % timeit(l4_only))
pr = _lib.Probe(4096)
with pr:
    tail(True)()
torch.cuda.synchronize()
tab = collections.OrderedDict()
for kind, work, ms, dims in pr.entries():
    if kind == _lib.PROBE_GEMM:
        t = tab.setdefault(dims, [0.0, 0, work])
        t[0] += ms
        t[1] += 1
tot = 0.0
for k, (ms, n, work) in sorted(tab.items(), key=lambda kv: -kv[1][0]):
    print("gemm M=%6d N=%5d K=%6d ta=%d tb=%d splits=%2d : x%d %8.1f us  %6.1f TF/s" % (k + (n, 1e3 * ms / n, work / (ms / n) / 1e9)))
    tot += ms
print("library GEMM launches: %.3f ms in all" % tot)
