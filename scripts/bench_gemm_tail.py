"""Micro-benchmark of the step's SHAPE TAIL (profiles/r05_gemm_by_shape.txt: the launches under ~150 TFLOP/s) through
ait_amd.ops.gemm / bgemm, one line per shape: us per launch and TFLOP/s.  AIT_LAB_LIB=<name> times a lab variant
(scripts/build_variant.py).  Run on the GPU box:  python scripts/bench_gemm_tail.py [filter]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import _lab_lib  # noqa: E402,F401
from ait_amd import ops  # noqa: E402

# (M, N, K, trans_a, trans_b, split_k, batch)   -- as bench.py --gemm-table prints them; batch > 1: ops.bgemm over [4, 8]
SHAPES = [
    (19328, 2048, 512, 0, 1, 1, 1), (19328, 2048, 512, 0, 0, 1, 1), (76800, 512, 512, 0, 1, 1, 1), (76800, 1024, 512, 0, 1, 1, 1),
    (76800, 1024, 320, 0, 0, 1, 1), (58800, 1024, 512, 0, 1, 1, 1),
    (512, 64, 76800, 1, 0, 127, 1), (512, 64, 76800, 1, 0, 64, 1), (512, 64, 76800, 1, 0, 253, 1), (512, 512, 76800, 1, 0, 64, 1),
    (512, 1024, 19328, 1, 0, 8, 1), (19328, 1024, 512, 0, 1, 1, 1), (19328, 512, 1024, 0, 1, 1, 1),
    (9576, 512, 1024, 0, 0, 1, 1), (9576, 512, 1024, 0, 1, 1, 1), (9576, 1024, 512, 0, 0, 1, 1), (9576, 1024, 512, 0, 1, 1, 1),
    (9576, 512, 512, 0, 0, 1, 1), (9576, 512, 512, 0, 1, 1, 1), (9576, 512, 9216, 0, 1, 1, 1),
    (512, 1024, 9576, 1, 0, 32, 1), (512, 1024, 9576, 1, 0, 8, 1), (512, 1024, 9576, 1, 0, 16, 1), (512, 1024, 9576, 1, 0, 64, 1),
    (1024, 512, 9576, 1, 0, 32, 1), (512, 512, 9576, 1, 0, 32, 1), (512, 512, 9576, 1, 0, 16, 1), (512, 512, 9576, 1, 0, 64, 1), (64, 512, 9576, 1, 0, 8, 1),
    (512, 64, 9576, 1, 0, 32, 1), (9576, 64, 512, 0, 0, 1, 1), (9576, 64, 512, 0, 1, 1, 1), (9576, 512, 64, 0, 0, 1, 1), (9576, 512, 64, 0, 1, 1, 1),
    (256, 1024, 320, 0, 0, 1, 1), (256, 512, 1024, 0, 0, 1, 1), (256, 512, 1024, 0, 1, 1, 1), (256, 1024, 512, 0, 0, 1, 1),
    (256, 1024, 512, 0, 1, 1, 1), (256, 512, 512, 0, 0, 1, 1), (256, 512, 512, 0, 1, 1, 1), (256, 64, 512, 0, 0, 1, 1), (256, 512, 64, 0, 1, 1, 1),
    (64, 1024, 1152, 0, 1, 1, 1), (64, 1024, 128, 0, 1, 1, 1), (512, 64, 1200, 1, 0, 8, 1),
    (1024, 512, 256, 1, 0, 8, 1), (512, 1024, 256, 1, 0, 1, 1), (1024, 1152, 64, 1, 0, 1, 1), (1024, 128, 64, 1, 0, 1, 1),
    (512, 512, 256, 1, 0, 8, 1), (512, 64, 256, 1, 0, 8, 1), (256, 512, 64, 0, 1, 1, 1),
    (64, 64, 2394, 1, 0, 1, 32), (64, 64, 2394, 0, 0, 1, 32), (2394, 64, 64, 1, 0, 1, 32), (2394, 64, 64, 0, 0, 1, 32),
    (2394, 64, 64, 0, 1, 1, 32), (64, 2394, 64, 0, 1, 1, 32),
]


def timeit(fn, n=30, w=5):
    """median DEVICE time of the launch: the library's own event pair around it (ait_amd._lib.Probe -- what bench.py's
    roofline sums), not host wall time: a Python call costs ~15 us, more than the small launches run"""
    from ait_amd import _lib
    for _ in range(w):
        fn()
    pr = _lib.Probe(4 * n)
    with pr:
        for _ in range(n):
            fn()
    torch.cuda.synchronize()
    ms = sorted(e[2] for e in pr.entries() if e[0] == _lib.PROBE_GEMM)
    return ms[len(ms) // 2]


# the AIT's weight gradients: K-range counts around the product's choice (AIT_TAIL_WGRAD=1 selects this list)
WGRAD = [(1536, 512, 76800, 1, 0, sp, 1) for sp in (64, 16, 20, 21, 24, 32, 40, 42)] + \
        [(512, 512, 76800, 1, 0, sp, 1) for sp in (64, 32, 56, 63)] + [(1024, 512, 76800, 1, 0, sp, 1) for sp in (32, 16, 24, 31)] + \
        [(2048, 512, 76800, 1, 0, sp, 1) for sp in (16, 8, 15)] + [(512, 2048, 58800, 1, 0, sp, 1) for sp in (16, 15)] + \
        [(512, 4608, 19328, 1, 0, sp, 1) for sp in (8, 7, 14)]
# the RPN heads' weight gradient (64 / 128 rows): K-range counts (AIT_TAIL_WGRAD=2)
RPNW = [(64, 512, 9576, 1, 0, sp, 1) for sp in (8, 16, 32, 40, 64)] + [(128, 512, 19152, 1, 0, sp, 1) for sp in (8, 16, 32, 64, 72)]
if os.environ.get("AIT_TAIL_WGRAD") == "1":
    SHAPES = WGRAD
if os.environ.get("AIT_TAIL_WGRAD") == "2":
    SHAPES = RPNW
flt = sys.argv[1] if len(sys.argv) > 1 else ""
tot = 0.0
for m, n, k, ta, tb, sk, batch in SHAPES:
    name = "%6d %5d %6d %d %d %3d b%-2d" % (m, n, k, ta, tb, sk, batch)
    if flt and flt not in name:
        continue
    if batch == 1:
        a = torch.randn((k, m) if ta else (m, k), device="cuda")
        b = torch.randn((n, k) if tb else (k, n), device="cuda")
        out = torch.zeros(m, n, device="cuda")
        fn = lambda: ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=out, split_k=sk, accumulate=False)
    else:
        a = torch.randn((4, 8, k, m) if ta else (4, 8, m, k), device="cuda")
        b = torch.randn((4, 8, n, k) if tb else (4, 8, k, n), device="cuda")
        out = torch.zeros(4, 8, m, n, device="cuda")
        fn = lambda: ops.bgemm(a, b, bool(ta), bool(tb), out=out)
    ms = timeit(fn)
    fl = 2.0 * m * n * k * batch
    tot += ms
    print("%s : %8.1f us %7.1f TF/s" % (name, ms * 1e3, fl / ms / 1e9), flush=True)
print("sum %.3f ms" % tot)
