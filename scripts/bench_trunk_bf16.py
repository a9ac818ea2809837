"""MIOpen convolutions of the C4 trunk in f32 against bf16 tensors (channels-last, benchmark mode): forward, data gradient,
weight gradient, ms per call -- what a bf16 trunk could buy at cfg5 (8 images of 600 x 1000)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
torch.backends.cudnn.benchmark = True


def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


cases = [("l1 3x3", 8, 64, 64, 150, 250, 3), ("l1 1x1", 8, 64, 256, 150, 250, 1), ("l2 3x3", 8, 128, 128, 75, 125, 3),
         ("l2 1x1a", 8, 512, 128, 75, 125, 1), ("l2 1x1b", 8, 128, 512, 75, 125, 1), ("l3 3x3", 8, 256, 256, 38, 63, 3),
         ("l3 1x1a", 8, 1024, 256, 38, 63, 1), ("l3 1x1b", 8, 256, 1024, 38, 63, 1)]
tot = {torch.float32: [0, 0, 0], torch.bfloat16: [0, 0, 0]}
for name, n, cin, cout, h, w, k in cases:
    row = []
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(n, cin, h, w, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(cout, cin, k, k, device="cuda", dtype=dt) * 0.05).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(n, cout, h, w, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
        pad = k // 2
        cb = lambda mask: torch.ops.aten.convolution_backward(dy, x, wt, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1, mask)
        t = [timeit(lambda: F.conv2d(x, wt, None, 1, pad)), timeit(lambda: cb([True, False, False])), timeit(lambda: cb([False, True, False]))]
        row.append(t)
        for i in range(3): tot[dt][i] += t[i]
    fl = 2.0 * n * h * w * cin * cout * k * k
    print("%-8s | f32 fwd %.3f dgrad %.3f wgrad %.3f ms (%.0f TF/s fwd) | bf16 fwd %.3f dgrad %.3f wgrad %.3f ms (%.0f TF/s fwd)"
          % (name, *row[0], fl / row[0][0] / 1e9, *row[1], fl / row[1][0] / 1e9), flush=True)
print("sum f32 ", [round(v, 3) for v in tot[torch.float32]], " bf16 ", [round(v, 3) for v in tot[torch.bfloat16]])
