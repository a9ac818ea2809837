"""Batched on-device NMS on the proposal layer's shapes (4 images x 12000 sorted boxes, keep 2000)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ait_amd.roi_layers import nms_sorted_batched
rng = np.random.RandomState(0)
B, n = 4, 12000
x1 = rng.uniform(0, 900, (B, n)); y1 = rng.uniform(0, 500, (B, n))
w = rng.uniform(16, 400, (B, n)); h = rng.uniform(16, 300, (B, n))
boxes = torch.from_numpy(np.stack([x1, y1, np.minimum(x1 + w, 999), np.minimum(y1 + h, 599)], -1).astype(np.float32)).cuda()
def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
keep, cnt = nms_sorted_batched(boxes, 0.7, 2000)
print("kept per image:", cnt.tolist(), " batched nms %.3f ms" % timeit(lambda: nms_sorted_batched(boxes, 0.7, 2000)))
