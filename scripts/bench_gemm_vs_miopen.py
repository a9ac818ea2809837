"""How fast does MIOpen's NHWC implicit-GEMM run the AIT GEMM shapes (as 1x1 convolutions on
channels-last tensors), next to ait_gemm_f32 and rocBLAS?  Sets the practical ceiling for the
fp32 matrix pipe on this chip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from ait_amd import ops
torch.backends.cudnn.benchmark = True
def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (bp, cin, cout) in ((1200, 512, 2048), (1200, 2048, 512), (1200, 512, 1536), (1200, 512, 512), (1200, 1024, 512)):
    M = bp * 64
    x = torch.randn(bp, cin, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last)
    xm = x.permute(0, 2, 3, 1).reshape(M, cin); wm = w.view(cout, cin)
    fl = 2.0 * M * cin * cout
    t_conv = timeit(lambda: F.conv2d(x, w))
    t_mine = timeit(lambda: ops.gemm(xm, wm))
    t_blas = timeit(lambda: xm @ wm.t())
    # dgrad / wgrad through autograd-free calls
    dy = torch.randn(bp, cout, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last)
    dym = dy.permute(0, 2, 3, 1).reshape(M, cout)
    t_cd = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]))
    t_md = timeit(lambda: ops.gemm(dym, wm, trans_b=False))
    t_cw = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
    from ait_amd.system import _wgrad
    t_mw = timeit(lambda: _wgrad(dym, xm))
    print("M=%d K=%d N=%d | fwd: miopen %.0f  mine %.0f  rocblas %.0f | dgrad: miopen %.0f mine %.0f | wgrad: miopen %.0f mine %.0f  TF/s"
          % (M, cin, cout, fl / t_conv / 1e9, fl / t_mine / 1e9, fl / t_blas / 1e9, fl / t_cd / 1e9, fl / t_md / 1e9, fl / t_cw / 1e9, fl / t_mw / 1e9))
