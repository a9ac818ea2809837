"""Which call sites own the torch-side elementwise kernels of one training step?
torch.profiler over 2 steps, grouped by (op, input shapes) and by python source line."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = sorted(ka, key=lambda e: -e.self_device_time_total)
print("== by op + shapes (self device time per step, ms)")
for e in rows[:70]:
    print("%8.3f  x%-4d %-38s %s" % (e.self_device_time_total / 2e3, e.count // 2, e.key[:38], str(e.input_shapes)[:110]))
ks = prof.key_averages(group_by_stack_n=6)
print("== by stack, elementwise only")
want = ("aten::add", "aten::copy_", "aten::mul", "aten::sum", "aten::add_", "aten::mul_", "aten::div", "aten::sub", "aten::clone", "aten::contiguous", "aten::cat", "aten::fill_", "aten::zero_")
rows = sorted((e for e in ks if e.key in want), key=lambda e: -e.self_device_time_total)
for e in rows[:45]:
    st = [s for s in e.stack if "ait_amd" in s or "bench" in s or "autograd" in s][:3]
    print("%8.3f  x%-4d %-14s %s" % (e.self_device_time_total / 2e3, e.count // 2, e.key, " <- ".join(s.split("/")[-1][:60] for s in st)))
