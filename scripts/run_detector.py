"""Smoke/timing of the full detector on the GPU box: train fwd+bwd at (bs, P)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ait_amd.config import cfg, cfg_from_list
from ait_amd.faster_rcnn import resnet

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
P = int(sys.argv[2]) if len(sys.argv) > 2 else 300
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg_from_list(['TRAIN.BATCH_SIZE', P])
torch.manual_seed(0); np.random.seed(3)
m = resnet(('bg', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3); m.create_architecture()
m = m.cuda().train()
im = torch.randn(bs, 3, 600, 1000, device='cuda'); q = torch.randn(bs, 3, 128, 128, device='cuda')
info = torch.tensor([[600, 1000, 1.0]] * bs, device='cuda')
gt = torch.zeros(bs, 20, 5, device='cuda')
rs = np.random.RandomState(1)
for b in range(bs):
    for g in range(3):
        w, h = rs.uniform(64, 400, 2); x1 = rs.uniform(0, 1000 - w); y1 = rs.uniform(0, 600 - h)
        gt[b, g] = torch.tensor([x1, y1, x1 + w, y1 + h, 1.0])
nb = torch.full((bs,), 3, device='cuda')
def step():
    m.zero_grad(set_to_none=True)
    out = m(im, q, info, gt, nb)
    loss = out[3].mean() + out[4].mean() + out[5].mean() + out[6].mean() + out[7].mean()
    loss.backward()
    return out, loss
out, loss = step()
torch.cuda.synchronize()
print("rois", tuple(out[0].shape), "cls_prob", tuple(out[1].shape), "losses", [float(x) for x in out[3:8]], "label sum", int(out[8].sum()))
ng = sum(1 for p in m.parameters() if p.requires_grad and p.grad is None)
print("trainable params without grad:", ng, [n for n, p in m.named_parameters() if p.requires_grad and p.grad is None][:6])
for _ in range(2): step()
torch.cuda.synchronize(); t = time.time()
for _ in range(steps): step()
torch.cuda.synchronize(); dt = (time.time() - t) / steps
print("step %.1f ms  -> %.2f pairs/s (bs=%d P=%d)" % (dt * 1e3, bs / dt, bs, P))
print("mem GB", torch.cuda.max_memory_allocated() / 2**30)
if len(sys.argv) > 4:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        step(); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=35, max_name_column_width=60))
