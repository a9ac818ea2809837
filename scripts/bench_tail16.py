"""microbenchmark: the proposal tail node (ait_tail_fwd + ait_tail_bwd) at cfg5's size under the bf16 product form; device time of
its products by shape from the library's event pairs.  usage: python scripts/bench_tail16.py [bp bs]"""
import sys
import collections
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab_lib  # noqa: F401  (AIT_LAB_LIB=<name>: a lab build of the library)
import ait_amd.faster_rcnn as fr
from ait_amd import ops, _lib

bp, bs = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 8)
torch.manual_seed(1)
m = fr.resnet(('__background__', 'fg'), 101, pretrained=False, class_agnostic=True, num_K=3)
m.create_architecture()
m = m.cuda().train()
x0 = torch.randn(bp, 1024, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last)
q0 = torch.randn(bs, 1024, 8, 8, device="cuda")
cot = torch.randn(bp + bs, 2048, device="cuda")
ops.set_matmul_dtype("bf16")
def step():
    x, q = x0.clone().requires_grad_(True), q0.clone().requires_grad_(True)
    yp, yq = m._tail(x, q)
    (torch.cat([yp, yq]) * cot).sum().backward()
for _ in range(3):
    step()
pr = _lib.Probe(4096)
n = 5
with pr:
    for _ in range(n):
        step()
torch.cuda.synchronize()
tab = collections.OrderedDict()
for kind, work, ms, dims in pr.entries():
    t = tab.setdefault(dims, [0.0, 0, work])
    t[0] += ms; t[1] += 1
tot = 0.0
for dims, (ms, cnt, work) in sorted(tab.items(), key=lambda kv: -kv[1][0]):
    print("M=%6d N=%5d K=%6d ta=%d tb=%d splits=%3d : %2d/step %8.1f us %7.1f TF/s %6.2f ms/step" % (dims + (cnt // n, 1e3 * ms / cnt, work / (ms / cnt) / 1e9, ms / n)))
    tot += ms / n
print("total %.2f ms/step in products" % tot)
