"""Per direction (fwd / dgrad / wgrad): MIOpen's pick for the layer4 / trunk 1x1 convolutions on
channels-last tensors vs ait_gemm_f32 on the same token-major matrices (TFLOP/s)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from ait_amd import ops, tuning
from ait_amd.system import _wgrad
tuning.use_tuned_miopen_db(0)
def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
cases = [("layer4", 1200, 4, 4, 1024, 512), ("layer4", 1200, 4, 4, 1024, 2048), ("layer4", 1200, 4, 4, 512, 2048),
         ("layer4", 1200, 4, 4, 2048, 512), ("layer3", 4, 38, 63, 1024, 256), ("layer3", 4, 38, 63, 256, 1024),
         ("layer2", 4, 75, 125, 512, 128), ("layer2", 4, 75, 125, 128, 512), ("layer1", 4, 150, 250, 256, 64), ("layer1", 4, 150, 250, 64, 256)]
for name, n, h, w_, cin, cout in cases:
    M = n * h * w_
    x = torch.randn(n, cin, h, w_, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, cout, h, w_, device="cuda").contiguous(memory_format=torch.channels_last)
    xm = x.permute(0, 2, 3, 1).reshape(M, cin); wm = w.view(cout, cin); dym = dy.permute(0, 2, 3, 1).reshape(M, cout)
    fl = 2.0 * M * cin * cout
    cb = lambda mask: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, mask)
    t = [timeit(lambda: F.conv2d(x, w)), timeit(lambda: ops.gemm(xm, wm, exact=True)),
         timeit(lambda: cb([True, False, False])), timeit(lambda: ops.gemm(dym, wm, trans_b=False, exact=True)),
         timeit(lambda: cb([False, True, False])), timeit(lambda: _wgrad(dym, xm))]
    print("%-7s M=%6d %4d->%4d | fwd miopen %5.0f mine %5.0f | dgrad miopen %5.0f mine %5.0f | wgrad miopen %5.0f mine %5.0f  TF/s  (ms: %s)"
          % (name, M, cin, cout, *[fl / x_ / 1e9 for x_ in t], " ".join("%.3f" % x_ for x_ in t)))
