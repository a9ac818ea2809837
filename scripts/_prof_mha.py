import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ait_amd import ops, _lib
exec(open(os.path.join(os.path.dirname(__file__), "bench_mha_core.py")).read().split("for p, save in")[0])
L = _lib.lib()
prof = torch.zeros(16, dtype=torch.int64, device=dev)
L.ait_mha_core_set_prof.argtypes = [ctypes.c_void_p]; L.ait_mha_core_set_prof.restype = None
names = ["load q,k", "QK^T", "softmax+P store+dropout", "PV", "O store + fc_w load issue", "B: gate", "C: head sum", "D: fc product", "f store", "E: dropout+residual", "row stats", "y store"]
for p, save in ((0.1, True), (0.0, False)):
    one = lambda: ops.mha_core_fwd(qkv, 0, qkv, 512, qkv, 1024, n, 2, 0, p, 5, sk_w, sk_b, fc_w, res, g, b, 1e-6, p, 6, save=save)
    one(); torch.cuda.synchronize()
    L.ait_mha_core_set_prof(ctypes.c_void_p(prof.data_ptr())); prof.zero_()
    one(); torch.cuda.synchronize()
    L.ait_mha_core_set_prof(None)
    v = prof.cpu().tolist()
    print("training" if save else "inference", "cycles per sequence (wave 0), total %d" % (sum(v[:12]) // n))
    for i, nm in enumerate(names): print("  %-28s %7d" % (nm, v[i] // n))
    print("  timed: %.3f ms" % timeit(one))
