"""Time the torch/MIOpen sub-networks of the detector in isolation (fwd+bwd) on the bench shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ait_amd import tuning
tuning.use_tuned_miopen_db(0)
import bench
dev = torch.device("cuda:0")
m = bench.build_model(300, dev)
def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
def fb(mod, x, fwd_only=False):
    def f():
        y = mod(x)
        y = y[0] if isinstance(y, tuple) else y
        if not fwd_only: y.sum().backward()
    return f
x = torch.randn(1200, 1024, 8, 8, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
x4 = torch.randn(1200, 1024, 4, 4, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
m.train()
print("layer4 (subsampled input, as the detector runs it) fwd      %.2f ms" % timeit(fb(lambda t: m._head_to_tail(t, subsampled=True), x4, True)))
print("layer4 (subsampled input, as the detector runs it) fwd+bwd  %.2f ms" % timeit(fb(lambda t: m._head_to_tail(t, subsampled=True), x4)))
print("sk_props stride 2 fwd+bwd                                   %.2f ms" % timeit(fb(lambda t: m.sk.sk_props(t, 2), x)))
im = torch.randn(4, 3, 600, 1000, device=dev)
print("backbone(image) fwd       %.2f ms" % timeit(fb(m.RCNN_base, im, True)))
print("backbone(image) fwd+bwd   %.2f ms" % timeit(fb(m.RCNN_base, im)))
q = torch.randn(4, 3, 128, 128, device=dev)
print("backbone(query) fwd+bwd   %.2f ms" % timeit(fb(m.RCNN_base, q)))
f = torch.randn(4, 1024, 38, 63, device=dev, requires_grad=True)
conv = lambda t: m.RCNN_rpn.RPN_bbox_pred(torch.relu(m.RCNN_rpn.RPN_Conv(t)))
print("rpn convs fwd+bwd         %.2f ms" % timeit(fb(conv, f)))
p = torch.randn(1200, 1024, 7, 7, device=dev, requires_grad=True); qq = torch.randn(4, 1024, 8, 8, device=dev, requires_grad=True)
m.train()
print("AIT transformer fwd       %.2f ms" % timeit(lambda: m.transformer(x_props=p, x_query=qq)))
print("AIT transformer fwd+bwd   %.2f ms" % timeit(lambda: m.transformer(x_props=p, x_query=qq).sum().backward()))
fi = torch.randn(4, 1024, 38, 63, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
fq = torch.randn(4, 1024, 8, 8, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
def coatt():
    a, b = m.coattention(x_img=fi, x_qry=fq)
    (a.sum() + b.sum()).backward()
print("image-level co-attention fwd+bwd  %.2f ms" % timeit(coatt))
