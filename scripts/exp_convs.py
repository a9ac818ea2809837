"""Experiment: MIOpen settings for the torch-side convolutions (backbone / layer4 / SK)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "base"
if "bench" in mode:
    torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
model = bench.build_model(300, dev)
if "cl" in mode:
    model.RCNN_base.to(memory_format=torch.channels_last)
    model.RCNN_top.to(memory_format=torch.channels_last)
opt = bench.make_optimizer(model)
np.random.seed(3)
batch = bench.synth_batch(4, 1000, dev)
if "cl" in mode:
    batch[0] = batch[0].contiguous(memory_format=torch.channels_last)
    batch[1] = batch[1].contiguous(memory_format=torch.channels_last)
def step():
    opt.zero_grad(set_to_none=True)
    out = model(*batch)
    bench.total_cost(out).backward()
    opt.step()
for _ in range(4): step()
torch.cuda.synchronize(); t = time.time()
for _ in range(8): step()
torch.cuda.synchronize(); dt = (time.time() - t) / 8
print("mode %s: %.1f ms/step, %.2f pairs/s" % (mode, dt * 1e3, 4 / dt))
