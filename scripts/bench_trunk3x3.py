"""The C4 trunk's 3x3 convolutions (resnet_sys_transformer_sk_dilat.py:85-96: conv2 of every bottleneck) on MIOpen (f32,
channels-last, what the step runs) against the library's implicit GEMM (ait_conv_*_f32: f32 products on the bf16 matrix pipe),
forward / data gradient / weight gradient, at the bench shapes (4 targets of 600 x 1000, 4 queries of 128 x 128).  us per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from ait_amd import ops, tuning
tuning.use_tuned_miopen_db(0)
def timeit(fn, n=15, w=4):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# (stage, channels, map, stride of the stage's first block's conv2 input -> the block's conv2 runs on the stage's output size)
CASES = [("layer1 x3 (frozen: forward only)", 64, (150, 250)), ("layer2 x4", 128, (75, 125)), ("layer3 x6", 256, (38, 63)),
         ("query layer2", 128, (16, 16)), ("query layer3", 256, (8, 8))]
for name, c, (h, w) in CASES:
    x = torch.randn(4, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = torch.randn(c, c, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    dy = torch.randn(4, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    fl = 2.0 * 4 * h * w * c * c * 9
    t_f = timeit(lambda: F.conv2d(x, wt, None, 1, 1))
    t_d = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
    t_w = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
    line = "%-34s C=%3d %3dx%3d  MIOpen fwd %7.1f dgrad %7.1f wgrad %7.1f us (%.0f / %.0f / %.0f TFLOP/s)" % (
        name, c, h, w, t_f, t_d, t_w, fl / t_f / 1e6, fl / t_d / 1e6, fl / t_w / 1e6)
    if ops.conv_supported((h, w), (h, w), 1, c, c):
        xm = x.detach().permute(0, 2, 3, 1).reshape(4 * h * w, c)
        wm = wt.detach().permute(0, 2, 3, 1).contiguous()
        dym = dy.permute(0, 2, 3, 1).reshape(4 * h * w, c)
        geom = ops.conv_geom(4, (h, w), (h, w), (3, 3), 1, 1)
        a_f = timeit(lambda: ops.conv_fwd(xm, wm, geom))
        a_d = timeit(lambda: ops.conv_bwd_data(dym, wm, geom))
        a_w = timeit(lambda: ops.conv_bwd_weight(dym, xm, geom, 3, 3))
        line += " | library fwd %7.1f dgrad %7.1f wgrad %7.1f us (%.0f / %.0f / %.0f)" % (a_f, a_d, a_w, fl / a_f / 1e6, fl / a_d / 1e6, fl / a_w / 1e6)
    else:
        line += " | library: shape not taken"
    print(line, flush=True)
