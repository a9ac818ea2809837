"""RoIAlign fwd/bwd micro-benchmark on the bench shapes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import _lab_lib  # noqa: F401  (AIT_LAB_LIB=<name>: a lab build of the library)
from ait_amd.roi_layers import roi_align
bs, P, C = 4, 300, 1024
feat = torch.randn(bs, C, 38, 63, device="cuda", requires_grad=True)
rs = np.random.RandomState(5)
side_w, side_h = rs.uniform(32, 480, bs * P), rs.uniform(32, 480, bs * P)
x1, y1 = rs.uniform(0, 1000 - side_w), rs.uniform(0, 600 - side_h)
rois = torch.from_numpy(np.stack([rs.randint(0, bs, bs * P).astype(np.float64), x1, y1, x1 + side_w, y1 + side_h], 1).astype(np.float32)).cuda()
def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
y = roi_align(feat, rois, (7, 7), 1 / 16., 0)
g = torch.randn_like(y)
fwd = timeit(lambda: roi_align(feat.detach(), rois, (7, 7), 1 / 16., 0))
def fb():
    feat.grad = None
    roi_align(feat, rois, (7, 7), 1 / 16., 0).backward(g)
both = timeit(fb)
fb_bytes = bs * C * 38 * 63 * 4 + bs * P * C * 49 * 4
print("roi_align fwd %.3f ms (%.2f TB/s algorithmic)  bwd %.3f ms (%.2f TB/s)" % (fwd, fb_bytes / fwd / 1e9, both - fwd, fb_bytes / (both - fwd) / 1e9))

# channels-last / token-major kernels (ait_roi_align_nhwc_*)
from ait_amd.roi_layers import ROIAlign
op = ROIAlign((7, 7), 1 / 16., 0, channels_last=True)
fcl = feat.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
y2 = op(fcl, rois)
g2 = torch.randn_like(y2)          # channels-last grad, like the one the AIT hands back
fwd2 = timeit(lambda: op(fcl.detach(), rois))
def fb2():
    fcl.grad = None
    op(fcl, rois).backward(g2)
both2 = timeit(fb2)
print("channels-last fwd %.3f ms (%.2f TB/s algorithmic)  bwd %.3f ms (%.2f TB/s)" % (fwd2, fb_bytes / fwd2 / 1e9, both2 - fwd2, fb_bytes / (both2 - fwd2) / 1e9))
print("max |nhwc - nchw| fwd: %.3e" % float((y2 - y).abs().max()))
