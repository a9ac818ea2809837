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
want = ("aten::add", "aten::copy_", "aten::mul", "aten::sum", "aten::add_", "aten::mul_", "aten::div", "aten::sub",
        "aten::fill_", "aten::zero_", "aten::mean", "aten::cat", "aten::max", "aten::index", "aten::where",
        "aten::masked_fill_", "aten::_softmax", "aten::clamp", "aten::exp", "aten::sort", "aten::gather",
        "aten::native_layer_norm", "aten::native_dropout", "aten::bmm", "aten::mm", "aten::addmm",
        "aten::max_pool2d_with_indices", "aten::max_pool2d_with_indices_backward", "aten::_foreach_add_",
        "aten::_foreach_mul_", "aten::threshold_backward", "aten::relu", "aten::sigmoid")
rows = sorted((e for e in ka if e.key.startswith("aten::") and e.self_device_time_total > 0
               and "convolution" not in e.key), key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows) / 2e3
print("== torch ops excluding convolutions: %.2f ms/step" % tot)
for e in rows[:60]:
    print("%7.3f ms  x%-3d %-34s %s" % (e.self_device_time_total / 2e3, e.count // 2, e.key[:34], str(e.input_shapes)[:120]))
