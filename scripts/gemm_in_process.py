"""The AIT GEMM shapes timed from Python through the C ABI (one launch per event pair), first in a fresh process,
then again after the detector's training step has run in the same process: separates what the kernel does from
what its surroundings in the model's step do to it.  usage: python scripts/gemm_in_process.py [--with-model]"""
import sys
import torch
sys.path.insert(0, ".")
from ait_amd import ops

SHAPES = [("qkv NT", 76800, 1536, 512, False, True), ("ffn1 NT", 76800, 2048, 512, False, True),
          ("ffn2 NT", 76800, 512, 2048, False, True), ("dgrad NN", 76800, 512, 2048, False, False)]


def run(tag):
    for name, M, N, K, ta, tb in SHAPES:
        a = torch.randn(M, K, device="cuda")
        b = torch.randn((N, K) if tb else (K, N), device="cuda")
        bias = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        for _ in range(4):
            ops.gemm(a, b, trans_a=ta, trans_b=tb, bias=bias, out=out)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(24)]
        for e0, e1 in ev:
            e0.record()
            ops.gemm(a, b, trans_a=ta, trans_b=tb, bias=bias, out=out)
            e1.record()
        torch.cuda.synchronize()
        t = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
        print(f"{tag:12s} {name:9s} {2.0 * M * N * K / t[len(t) // 2] / 1e9:7.1f} TFLOP/s (median of 24)", flush=True)


run("fresh")
if "--with-model" in sys.argv:
    import numpy as np
    import bench
    from ait_amd import tuning
    tuning.use_tuned_miopen_db(0)
    dev = torch.device("cuda:0")
    model = bench.build_model(300, dev)
    opt = bench.make_optimizer(model)
    np.random.seed(3)
    batch = bench.synth_batch(4, 1000, dev)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        bench.total_cost(model(*batch)).backward()
        opt.step()
    torch.cuda.synchronize()
    run("after steps")
