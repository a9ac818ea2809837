"""Host-side cost per call of the trunk's building blocks on tiny inputs (where the GPU work is a few microseconds and
the forward is bound by how fast the host can issue): F.conv2d, the fused frozen-BN pass, a whole Bottleneck."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ait_amd.faster_rcnn as fr
from ait_amd import tuning
tuning.use_tuned_miopen_db(0)
dev = torch.device("cuda:0")
x = torch.randn(4, 256, 16, 16, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
conv = torch.nn.Conv2d(256, 256, 3, 1, 1, bias=False).to(dev).to(memory_format=torch.channels_last)
bn = torch.nn.BatchNorm2d(256).to(dev).eval()
for p in bn.parameters():
    p.requires_grad_(False)
blk = fr.Bottleneck(1024, 256).to(dev).to(memory_format=torch.channels_last)
for m in blk.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.eval()
        for p in m.parameters():
            p.requires_grad_(False)
xb = torch.randn(4, 1024, 8, 8, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)


def t(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return host * 1e6


y = conv(x)
print("F.conv2d (grad mode)          %6.1f us/call" % t(lambda: conv(x)))
print("bn_act  (grad mode)           %6.1f us/call" % t(lambda: fr.bn_act(y, bn)))
with torch.no_grad():
    print("F.conv2d (no_grad)            %6.1f us/call" % t(lambda: conv(x)))
    print("bn_act  (no_grad)             %6.1f us/call" % t(lambda: fr.bn_act(y, bn)))
print("Bottleneck forward (grad)     %6.1f us/call  (3 conv + 3 bn_act)" % t(lambda: blk(xb)))
out = blk(xb)
g = torch.randn_like(out)
print("Bottleneck fwd+bwd            %6.1f us/call" % t(lambda: blk(xb).backward(g), 100))
