"""bench.py -- (target, query) pairs/sec, fwd+bwd+SGD step, ResNet50, 300 proposals, on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; each rank trains on its own shard of the global batch (4 pairs per GPU, weak
scaling); gradients are all-reduced over RCCL/xGMI by DDP.  Rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over one batch of synthetic input: _fasterRCNN.forward in
.train() (backbone x2, co-attention, RPN + NMS, RoI sampling, RoIAlign, AIT transformer, SKNet,
layer4, heads, losses) + backward + optimizer.step(), inputs already resident in HBM.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "(target,query) pairs/sec fwd+bwd, ResNet50 300 proposals, at 1/2/4/8 MI355X"
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz


def synth_batch(bs, seed, device, im_hw=(600, 1000), q=128, max_gt=20, n_gt=3):
    """SURVEY.md 8d synthetic inputs: randn target/query, 3 random GT boxes of side 64..400."""
    g = torch.Generator().manual_seed(seed)
    rs = np.random.RandomState(seed)
    im = torch.randn(bs, 3, im_hw[0], im_hw[1], generator=g)
    qr = torch.randn(bs, 3, q, q, generator=g)
    info = torch.tensor([[im_hw[0], im_hw[1], 1.0]] * bs)
    gt = torch.zeros(bs, max_gt, 5)
    for b in range(bs):
        for k in range(n_gt):
            w, h = rs.uniform(64, 400, 2)
            x1, y1 = rs.uniform(0, im_hw[1] - w), rs.uniform(0, im_hw[0] - h)
            gt[b, k] = torch.tensor([x1, y1, x1 + w, y1 + h, 1.0])
    nb = torch.full((bs,), n_gt, dtype=torch.long)
    return [t.to(device) for t in (im, qr, info, gt, nb)]


def build_model(P, device):
    from ait_amd.config import cfg_from_list
    from ait_amd.faster_rcnn import resnet
    cfg_from_list(['TRAIN.BATCH_SIZE', P])      # P RoIs per image reach RoIAlign / AIT
    torch.manual_seed(1234)                     # identical initial weights on every rank
    m = resnet(('__background__', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3)
    m.create_architecture()
    return m.to(device).train()


def make_optimizer(model, lr=0.001, momentum=0.9, weight_decay=0.0001):
    """trainval_net_voc.py:289-305 with cfgs/res50.yml (DOUBLE_BIAS False, BIAS_DECAY False)."""
    decay, no_decay = [], []
    for k, v in model.named_parameters():
        if v.requires_grad:
            (no_decay if 'bias' in k else decay).append(v)
    groups = [{'params': decay, 'weight_decay': weight_decay}, {'params': no_decay, 'weight_decay': 0.0}]
    # same update rule as the reference's torch.optim.SGD; on the GPU as one multi-tensor kernel
    fused = all(p.is_cuda for p in decay + no_decay)
    return torch.optim.SGD(groups, lr=lr, momentum=momentum, fused=True) if fused else \
        torch.optim.SGD(groups, lr=lr, momentum=momentum)


def total_cost(out):
    return out[3].mean() + out[4].mean() + out[5].mean() + out[7].mean() + out[6].mean()


def cpu_baseline(P, seconds_budget=25.0):
    """The CPU oracle (oracle/detector_ref.py, a port of the reference path; the reference itself
    cannot run backward on CPU) timed on this box's host cores on ONE pair at a time."""
    try:
        from oracle import detector_ref
    except Exception as e:          # the checker is optional for the timing run
        return {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port",
                "sample": "oracle unavailable: %r" % (e,)}
    cores = min(64, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    return detector_ref.time_train_step(P=P, cores=cores, seconds_budget=seconds_budget)


def pmc_traffic_per_launch(kernel_prefix="gemm_f32_kernel"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this
    same command (profiles/pmc_fetch_r*.csv, pmc_write_r*.csv; separate --pmc runs, summarised by
    scripts/pmc_summary.py).  Correction per MI355X_MICROARCH.md "HBM": FETCH_SIZE reports half
    the bytes of wide coalesced reads on gfx950, so it is doubled; WRITE_SIZE is exact; both are
    in KiB.  PMC counters cannot be read from inside this process, hence the committed files."""
    import csv
    import glob
    out = {}
    for kind in ("fetch", "write"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_%s_r*.csv" % kind)))
        if not files:
            return None
        tot, n = 0.0, 0
        for r in csv.DictReader(open(files[-1])):
            if r["kernel"].startswith(kernel_prefix):
                tot += float(r["total"])
                n += int(r["launches"])
        if n == 0:
            return None
        out[kind] = (tot * 1024.0 / n, os.path.basename(files[-1]))
    return {"bytes_per_launch": 2.0 * out["fetch"][0] + out["write"][0],
            "fetch_bytes_raw": out["fetch"][0], "write_bytes": out["write"][0],
            "source": "profiles/%s + profiles/%s" % (out["fetch"][1], out["write"][1])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bs", type=int, default=4, help="pairs per GPU (BASELINE cfg2: 4)")
    ap.add_argument("--proposals", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["f32", "bf16", "bf16x3"], default="f32",
                    help="matmul arithmetic of the AIT GEMMs.  f32 (default) is the headline / parity "
                         "configuration; bf16 is the BASELINE cfg-5 arithmetic (operands rounded to bf16, "
                         "fp32 accumulate) and is reported as such, never as the headline number")
    args = ap.parse_args()

    from ait_amd import distributed as D
    from ait_amd import _lib, ops, tuning
    rank, local_rank, world = D.init()
    tuned = tuning.use_tuned_miopen_db(rank)      # MIOpen solver picks for the torch-side convs
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    _lib.lib()                                   # fail loudly if libait_hip.so is missing
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)

    ops.set_matmul_dtype(args.dtype)
    model = build_model(args.proposals, device)
    opt = make_optimizer(model)
    ddp = D.wrap(model, local_rank)
    np.random.seed(3 + rank)                     # reference RNG_SEED, one stream per rank
    batch = synth_batch(args.bs, 1000 + rank, device)

    def step():
        opt.zero_grad(set_to_none=True)
        out = ddp(*batch)
        total_cost(out).backward()
        opt.step()

    for _ in range(args.warmup):
        step()
    ops.GEMM_PROFILE = []
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    D.barrier()
    elapsed = time.perf_counter() - t0
    prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
    elapsed = D.max_over_ranks(elapsed, device)

    if rank != 0:
        return
    flops = sum(p[0] for p in prof)
    gemm_ms = sum(p[1].elapsed_time(p[2]) for p in prof)
    achieved = flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    pairs = world * args.bs * args.steps
    # dense MFMA peak for the arithmetic: fp32, bf16, or bf16 / 3 MFMAs per product
    peak = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16": 2500.0, "bf16x3": 2500.0 / 3}[args.dtype]
    pmc = pmc_traffic_per_launch() if args.dtype == "f32" else None
    # algorithmic bytes of the same launches: each operand read once, the output written once
    alg = sum(4.0 * (p[3][0] * p[3][2] + p[3][1] * p[3][2] + p[3][0] * p[3][1]) for p in prof) / max(1, len(prof))
    line = {
        "metric": METRIC, "value": pairs / elapsed, "unit": "pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32", "bf16": "bf16 (AIT GEMM operands; fp32 accumulate, fp32 elsewhere)",
                  "bf16x3": "f32 emulated as 3 bf16 MFMAs per product (experimental; fp32 accumulate)"}[args.dtype],
        "data": "synthetic",
        "config": {"workload": "ResNet50 VOC seen-classes, %d proposals, bs=%d per GPU, "
                               "fwd+bwd+SGD step (BASELINE.json configs[1])" % (args.proposals, args.bs),
                   "pairs_per_gpu": args.bs, "global_batch": world * args.bs,
                   "proposals": args.proposals, "target": "600x1000", "query": "128x128",
                   "parallelism": "dp%d" % world, "miopen_find_db": bool(tuned)},
        "roofline": {"bound": "mfma",
                     "kernel": "gemm_f32_kernel (v_mfma_f32_32x32x2_f32)" if args.dtype == "f32"
                               else "gemm_bf16_kernel (v_mfma_f32_32x32x16_bf16%s)" % (", 3 per product" if args.dtype == "bf16x3" else ""),
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved / peak,
                     "traffic": pmc["bytes_per_launch"] if pmc else None,
                     "traffic_detail": pmc, "algorithmic_bytes_per_launch": alg,
                     "launches_per_step": len(prof) // max(1, args.steps),
                     "gemm_ms_per_step": gemm_ms / max(1, args.steps),
                     "gemm_gflop_per_step": flops / max(1, args.steps) / 1e9},
    }
    if os.environ.get("AIT_BENCH_GEMM_TABLE"):
        import collections
        tab = collections.OrderedDict()
        for p in prof:
            t = tab.setdefault(p[3], [0.0, 0])
            t[0] += p[1].elapsed_time(p[2])
            t[1] += 1
        for k, (ms, n) in sorted(tab.items(), key=lambda kv: -kv[1][0]):
            fl = 2.0 * k[0] * k[1] * k[2]
            print("gemm M=%6d N=%5d K=%6d ta=%d tb=%d sk=%2d colblk=%3d : %2d/step %8.1f us  %6.1f TF/s  %5.2f ms/step"
                  % (k + (n // args.steps, 1e3 * ms / n, fl / (ms / n) / 1e9, ms / args.steps)), file=sys.stderr)
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.proposals)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
