import sys, torch
sys.path.insert(0, '.')
from ait_amd import ops
R, bs, F = 1200, 4, 2048
props, query = torch.randn(R, F, device='cuda'), torch.randn(bs, F, device='cuda')
wb, bb = torch.randn(4, F, device='cuda'), torch.randn(4, device='cuda')
w1, b1 = torch.randn(8, 2 * F, device='cuda'), torch.randn(8, device='cuda')
w2, b2 = torch.randn(2, 8, device='cuda'), torch.randn(2, device='cuda')
for _ in range(5): ops.heads_fwd(props, query, wb, bb, w1, b1, w2, b2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.heads_fwd(props, query, wb, bb, w1, b1, w2, b2)
e1.record(); torch.cuda.synchronize()
print("heads_fwd %.1f us per call (with host overhead)" % (e0.elapsed_time(e1) / 50 * 1e3))
