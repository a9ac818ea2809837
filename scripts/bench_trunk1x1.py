"""The C4 trunk's 1x1 convolutions (channels-last = token-major GEMMs): what a bottleneck's 1x1 step costs today
(MIOpen convolution + the frozen-BN/ReLU pass, backward: the BN/ReLU pass + MIOpen dgrad + wgrad + the residual add)
against the library's GEMM with the BatchNorm folded into the weights and shift / residual / ReLU / mask in the
epilogue.  ms per call, image trunk (4 x 600x1000) and query trunk (4 x 128x128) shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from ait_amd import ops, tuning
from ait_amd.system import _wgrad
tuning.use_tuned_miopen_db(0)


def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


cases = [("img l1 c1", 4, 150, 250, 64, 64, 1), ("img l1 c3", 4, 150, 250, 64, 256, 3), ("img l1 c1b", 4, 150, 250, 256, 64, 2),
         ("img l2 c1", 4, 75, 125, 512, 128, 3), ("img l2 c3", 4, 75, 125, 128, 512, 4), ("img l2 c1a", 4, 75, 125, 256, 128, 1),
         ("img l2 ds", 4, 75, 125, 256, 512, 1),
         ("img l3 c1", 4, 38, 63, 1024, 256, 5), ("img l3 c3", 4, 38, 63, 256, 1024, 6), ("img l3 c1a", 4, 38, 63, 512, 256, 1),
         ("img l3 ds", 4, 38, 63, 512, 1024, 1),
         ("qry l2 c1", 4, 16, 16, 512, 128, 3), ("qry l3 c3", 4, 8, 8, 256, 1024, 6)]
tot = {"fwd_now": 0.0, "fwd_new": 0.0, "bwd_now": 0.0, "bwd_new": 0.0}
for name, n, h, w_, cin, cout, count in cases:
    M = n * h * w_
    x = torch.randn(n, cin, h, w_, device="cuda").contiguous(memory_format=torch.channels_last).relu_()
    w = (torch.randn(cout, cin, 1, 1, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, cout, h, w_, device="cuda").contiguous(memory_format=torch.channels_last)
    res = torch.randn_like(dy)
    scale, shift = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
    xm = x.permute(0, 2, 3, 1).reshape(M, cin); wm = w.view(cout, cin); dym = dy.permute(0, 2, 3, 1).reshape(M, cout)
    resm = res.permute(0, 2, 3, 1).reshape(M, cout)
    wf = (wm * scale[:, None]).contiguous()
    y = ops.bn_act_fwd(F.conv2d(x, w), scale, shift, res, True)
    ym = y.permute(0, 2, 3, 1).reshape(M, cout)
    out = torch.empty(M, cout, device="cuda"); dx = torch.empty(M, cin, device="cuda")
    cb = lambda mask: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, mask)
    t_conv = timeit(lambda: F.conv2d(x, w))
    t_bn = timeit(lambda: ops.bn_act_fwd(dy, scale, shift, res, True))
    t_g = timeit(lambda: ops.gemm(xm, wf, bias=shift, residual=resm, relu=True, out=out))
    t_bnb = timeit(lambda: ops.bn_act_bwd(dy, y, scale, True, True))
    t_dg = timeit(lambda: cb([True, False, False]))
    t_wg = timeit(lambda: cb([False, True, False]))
    t_add = timeit(lambda: torch.add(x, x))
    t_gdg = timeit(lambda: ops.gemm_relu_bwd(dym, wf, xm, out=dx))          # dgrad + mask by the input's ReLU
    t_gwg = timeit(lambda: _wgrad(dym, xm))
    t_mask = timeit(lambda: ops.bn_act_bwd(dy, y, scale, True, False))       # (the one elementwise pass a fused chain may still need)
    # correctness of the fused forward against the two-step one
    ref = ops.bn_act_fwd(F.conv2d(x, w), scale, shift, res, True).permute(0, 2, 3, 1).reshape(M, cout)
    ops.gemm(xm, wf, bias=shift, residual=resm, relu=True, out=out)
    err = float((out - ref).abs().max() / ref.abs().max())
    print("%-10s M=%6d %4d->%4d x%d | fwd: conv %.3f + bn %.3f  vs gemm %.3f | bwd: bn %.3f + dgrad %.3f + wgrad %.3f + add %.3f  vs"
          " gemm-dgrad(mask) %.3f + my-wgrad %.3f (mask pass %.3f) | err %.1e"
          % (name, M, cin, cout, count, t_conv, t_bn, t_g, t_bnb, t_dg, t_wg, t_add, t_gdg, t_gwg, t_mask, err), flush=True)
    tot["fwd_now"] += count * (t_conv + t_bn); tot["fwd_new"] += count * t_g
    tot["bwd_now"] += count * (t_bnb + t_dg + t_wg + t_add); tot["bwd_new"] += count * (t_gdg + min(t_wg, t_gwg))
print("per step (counts as in ResNet50 C4, frozen layer1 forward only is included in fwd):", {k: round(v, 3) for k, v in tot.items()})
