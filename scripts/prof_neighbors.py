"""For the large torch-side copy_/add_/fill_ ops of one training step: what ran just before / after
them (CPU-side op order), to attribute autograd-inserted work to a call site."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
import bench
from ait_amd import tuning
tuning.use_tuned_miopen_db(0)
dev = torch.device("cuda:0")
model = bench.build_model(300, dev)
opt = bench.make_optimizer(model)
np.random.seed(3)
batch = bench.synth_batch(4, 1000, dev)
def step():
    opt.zero_grad(set_to_none=True)
    out = model(*batch)
    bench.total_cost(out).backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.cpu_parent is None or
       (e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::"))]
evs = sorted(evs, key=lambda e: e.time_range.start)
names = [e.name for e in evs]
tot = {}
for i, e in enumerate(evs):
    if e.name in ("aten::copy_", "aten::add_", "aten::fill_", "aten::add", "aten::mul", "aten::sum", "aten::zero_") and e.device_time_total > 40:
        ctx = " | ".join(names[max(0, i - 4):i]) + "  >>>  " + " | ".join(names[i + 1:i + 3])
        key = (e.name, str(e.input_shapes)[:60], ctx[:230])
        t = tot.setdefault(key, [0.0, 0]); t[0] += e.device_time_total; t[1] += 1
for (n, sh, ctx), (us, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:28]:
    print("%7.0f us x%d %-10s %s\n          %s" % (us, c, n, sh, ctx))
