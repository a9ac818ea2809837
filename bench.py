"""bench.py -- (target, query) pairs/sec, fwd+bwd+SGD step, ResNet50, 300 proposals, on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N --steps K --warmup W          (starts its own N ranks, one child process per GPU)
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
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured float4 copy)
AIT_TFLOP_PER_PAIR = 0.911       # AIT forward + backward at P = 300, as computed by the reference


def measured_peaks(device):
    """What this box attains on the two rooflines, next to the spec peaks: a register-only
    v_mfma_f32_32x32x2_f32 loop (scripts/mfma_peak.hip, built by __graft_entry__.build(), run as a child
    process) and a 1-GiB device-to-device copy."""
    import subprocess
    out = {"mfma_f32_tflops": None, "mfma_bf16_tflops": None, "hbm_copy_gbs": None}
    exe = os.path.join(ROOT, "scripts", "_mfma_peak")
    if os.path.exists(exe):
        try:
            r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=120)
            vals = [float(l.split("ms")[1].split("TFLOP/s")[0]) for l in r.stdout.splitlines() if "fp32 MFMA" in l]
            out["mfma_f32_tflops"] = max(vals) if vals else None
            # the bf16 pipe UNDER LOAD: per waves-per-SIMD setting the LAST (sustained) window, random operand bits
            per = {}
            for l in r.stdout.splitlines():
                if "bf16 MFMA" in l:
                    per[l.split(":")[0]] = float(l.split("ms")[1].split("TFLOP/s")[0])
            out["mfma_bf16_tflops"] = max(per.values()) if per else None
        except Exception:
            pass
    x = torch.empty(1 << 28, dtype=torch.float32, device=device)
    y = torch.empty_like(x)
    for _ in range(2):
        y.copy_(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    out["hbm_copy_gbs"] = 5 * 2 * x.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9     # read + write
    del x, y
    return out


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


# BASELINE.json configs the driver's command can select (bs pairs PER GPU, P proposals):
#   cfg2 (default, the headline)  ResNet50 VOC variant, 9 anchors, MAX_NUM_GT_BOXES 20, 4 pairs x 300 proposals
#   cfg3 / cfg4                   ResNet50 COCO variant (non-local co-attention, trainval_net_coco.py:34), 12 anchors
#                                 (ANCHOR_SCALES [4,8,16,32], trainval_net_coco.py:196-204), 50 gt boxes, 8 pairs x 300;
#                                 cfg3 = 2 GPUs (global batch 16), cfg4 = 8 GPUs (global batch 64): --gpus decides
#   cfg5                          ResNet101, COCO variant, 8 pairs x 512 proposals, bf16 AIT matmuls
CONFIGS = {
    "cfg2": dict(variant="voc", layers=50, bs=4, proposals=300, dtype="f32",
                 workload="ResNet50 VOC seen-classes, %(P)d proposals, bs=%(bs)d per GPU, fwd+bwd+SGD step (BASELINE.json configs[1])"),
    "cfg3": dict(variant="coco", layers=50, bs=8, proposals=300, dtype="f32",
                 workload="ResNet50 COCO --g 1, %(P)d proposals, bs=%(bs)d per GPU, fwd+bwd+SGD step (BASELINE.json configs[2]: bs=16 on 2 GPUs)"),
    "cfg4": dict(variant="coco", layers=50, bs=8, proposals=300, dtype="f32",
                 workload="ResNet50 COCO --g 0, 1000x600 targets, %(P)d proposals, bs=%(bs)d per GPU, fwd+bwd+SGD step (BASELINE.json configs[3]: bs=64 on 8 GPUs)"),
    "cfg5": dict(variant="coco", layers=101, bs=8, proposals=512, dtype="bf16",
                 workload="ResNet101 COCO bf16, %(P)d proposals, bs=%(bs)d per GPU, fwd+bwd+SGD step (BASELINE.json configs[4])"),
}


def build_model(P, device, variant="voc", layers=50):
    from ait_amd.config import cfg_from_list
    from ait_amd.faster_rcnn import resnet, resnet_coco
    cfg_from_list(['TRAIN.BATCH_SIZE', P])      # P RoIs per image reach RoIAlign / AIT
    if variant == "coco":                       # trainval_net_coco.py:196-204
        cfg_from_list(['ANCHOR_SCALES', [4, 8, 16, 32], 'MAX_NUM_GT_BOXES', 50])
    torch.manual_seed(1234)                     # identical initial weights on every rank
    cls = resnet_coco if variant == "coco" else resnet
    m = cls(('__background__', 'fg'), layers, pretrained=False, class_agnostic=True, num_K=3)
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


def cpu_baseline(P, seconds_budget=16.0):
    """The CPU oracle (oracle/detector_ref.py, a port of the reference path; the reference itself
    cannot run backward on CPU) timed on this box's host cores on ONE pair at a time.  `value` is
    SURVEY 8d's figure (ii), the full forward + backward; (i) the eval forward and (iii) the AIT alone
    ride along."""
    try:
        from oracle import detector_ref
    except Exception as e:          # the checker is optional for the timing run
        return {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port",
                "sample": "oracle unavailable: %r" % (e,)}
    cores = min(64, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    out = detector_ref.time_train_step(P=P, cores=cores, seconds_budget=seconds_budget)
    ev = detector_ref.time_eval_forward(P=P, cores=cores, seconds_budget=5.0)
    ao = detector_ref.time_ait_only(P=P, cores=cores, seconds_budget=6.0)
    out["eval_forward"] = {"value": ev["value"], "unit": "pairs/s", "sample": ev["sample"]}
    out["ait_only_fwd_bwd"] = {"value": ao["value"], "unit": "pairs/s", "sample": ao["sample"]}
    return out


def pmc_clock_and_util():
    """Clock-independent matrix-pipe utilisation and the effective shader clock of the GEMM launches inside
    this step, from the committed `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace`
    pass over this same command (scripts/profile_bench.sh, scripts/pmc_clock.py): the step runs its
    GEMMs below the 2.4 GHz the 157.3 TFLOP/s peak assumes."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_mfma_util_and_clock.txt")))
    if not files:
        return None
    util, clk, n_clk, in_clock = None, 0.0, 0, False
    for l in open(files[-1]):
        m = re.match(r"\s*([\d.]+) % mean over all (\d+) launches", l)
        if m:
            util = float(m.group(1)) / 100.0
        if l.startswith("effective clock"):
            in_clock = True
            continue
        m = re.match(r"\s*([\d.]+) median\s+[\d.]+ min\s+[\d.]+ max\s+n=(\d+)", l)
        if in_clock and m:
            clk += float(m.group(1)) * int(m.group(2))
            n_clk += int(m.group(2))
    if util is None or n_clk == 0:
        return None
    return {"mfma_util": util, "effective_clock_ghz": clk / n_clk, "nominal_clock_ghz": 2.4,
            "source": "profiles/" + os.path.basename(files[-1])}


def pmc_traffic_per_launch(kernel_prefix="gemm_f32_"):
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


def spawn_ranks(n, argv, deadline_s=None, grace_s=10.0, script=None):
    """`python bench.py --gpus N` outside a torch.distributed.run environment: this parent starts N fresh child
    processes of this same script, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torchrun would:
    the reference's `python trainval_net_voc.py --mGPUs`, trainval_net_voc.py:83,321-326, is one command too),
    relays rank 0's JSON line and returns non-zero if any rank fails.  The parent never initialises the GPU
    (torch.cuda.device_count() only) and nothing is exec'ed.

    It cannot hang: when a rank fails the others get SIGTERM and, `grace_s` later, SIGKILL (a rank stuck inside an
    RCCL collective or a kernel does not act on SIGTERM); the whole job has a deadline (AIT_BENCH_DEADLINE_S, default
    3600 s) behind which everything is killed and the exit code is 124.  The rendezvous port is picked by binding port 0
    and closing it, so another process can take it before rank 0 listens: a job whose rank 0 dies of EADDRINUSE is
    started again on a new port (three tries)."""
    import socket
    import subprocess
    import tempfile
    have = torch.cuda.device_count()
    gloo = os.environ.get("AIT_DIST_BACKEND") == "gloo"          # test hook: N ranks share the GPUs there are
    if have < n and not gloo:
        sys.stderr.write("bench.py: --gpus %d but %d GPU(s) visible\n" % (n, have))
        return 2
    deadline_s = float(os.environ.get("AIT_BENCH_DEADLINE_S", "3600")) if deadline_s is None else deadline_s
    t_end = time.monotonic() + deadline_s

    def stop(procs):
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + grace_s
        while any(p.poll() is None for p in procs) and time.monotonic() < t_kill:
            time.sleep(0.05)
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()

    for attempt in range(3):
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        procs, errs = [], []
        with tempfile.TemporaryFile(mode="w+") as out0:
            for r in range(n):
                # (HSA_ENABLE_IPC_MODE_LEGACY: see ait_amd/distributed.py init())
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                           MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
                env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                if gloo:
                    env.setdefault("GLOO_SOCKET_IFNAME", "lo")
                errs.append(tempfile.TemporaryFile(mode="w+"))
                procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env,
                                              stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=errs[-1]))
            # a rank that dies leaves the others inside a collective: stop them (by their exact PIDs) instead of
            # waiting for the collective's timeout
            failed, timed_out = None, False
            while any(p.poll() is None for p in procs):
                bad = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
                if bad:
                    failed = bad[0]
                    time.sleep(min(2.0, grace_s))       # let the others fail by themselves and say why
                    stop(procs)
                    break
                if time.monotonic() > t_end:
                    timed_out = True
                    stop(procs)
                    break
                time.sleep(0.05)
            codes = [p.wait() for p in procs]
            texts = []
            for e in errs:
                e.seek(0)
                texts.append(e.read())
                e.close()
            in_use = any(k in texts[0] for k in ("EADDRINUSE", "Address already in use", "address already in use"))
            if codes[0] != 0 and in_use and attempt < 2 and not timed_out:
                sys.stderr.write("bench.py: port %d was taken before rank 0 could listen: starting the ranks again\n" % port)
                continue
            for t in texts:
                sys.stderr.write(t)
            out0.seek(0)
            sys.stdout.write(out0.read() if not (any(codes) or timed_out) else "")
            sys.stdout.flush()
        if timed_out:
            sys.stderr.write("bench.py: the %d ranks did not finish within %.0f s: killed\n" % (n, deadline_s))
            return 124
        if any(codes):
            sys.stderr.write("bench.py: rank exit codes %r%s\n" % (codes, "" if failed is None else " (rank %d failed first)" % failed))
            return 1
        return 0
    return 1


def cpu_model():
    """the host CPU's model string (core count AND CPU model printed beside the CPU baseline)"""
    try:
        for l in open("/proc/cpuinfo"):
            if l.lower().startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2",
                    help="BASELINE.json configuration (cfg2 = configs[1], the headline; cfg3/cfg4 = the COCO-variant "
                         "8-pairs-per-GPU workload of configs[2]/[3]; cfg5 = configs[4], ResNet101 bf16)")
    ap.add_argument("--variant", choices=["voc", "coco"], default=None, help="override the configuration's detector variant")
    ap.add_argument("--bs", type=int, default=None, help="pairs per GPU (override; cfg2: 4, cfg3-5: 8)")
    ap.add_argument("--proposals", type=int, default=None)
    ap.add_argument("--exchange", choices=["allreduce", "rs_ag"], default="allreduce",
                    help="N > 1: the gradient buckets' exchange (default DDP's all-reduce; rs_ag is experimental and is "
                         "selected by this argument only)")
    ap.add_argument("--force-ddp", action="store_true",
                    help="--gpus 1 only: build a real one-rank process group (RCCL) and the DDP wrapper anyway "
                         "(same as AIT_FORCE_DDP=1): runs the N > 1 code path on a one-GPU box; not the headline mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ab", action="store_true", help="skip the f32_native A/B steps behind the timed region")
    ap.add_argument("--no-probe-pass", action="store_true",
                    help="skip the probe-overhead steps behind the timed region (profiling runs: the trace then holds warm-up + "
                         "timed steps only); probe_overhead_ms_per_step is null")
    ap.add_argument("--gemm-table", default=None, metavar="PATH",
                    help="also write the timed region's GEMM launches grouped by shape (launches/step, ms/step, TFLOP/s)")
    ap.add_argument("--dtype", choices=["f32", "f32_native", "bf16"], default=None,
                    help="matmul arithmetic of the AIT GEMMs.  f32 is the headline / parity "
                         "configuration; bf16 is the BASELINE cfg-5 arithmetic (operands rounded to bf16, "
                         "fp32 accumulate) and is reported as such, never as the headline number")
    args = ap.parse_args()
    conf = CONFIGS[args.config]
    args.variant = args.variant or conf["variant"]
    args.bs = args.bs or conf["bs"]
    args.proposals = args.proposals or conf["proposals"]
    args.dtype = args.dtype or conf["dtype"]

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    from ait_amd import distributed as D
    if args.force_ddp:
        os.environ["AIT_FORCE_DDP"] = "1"
    # a rank of an N > 1 job: next to its GPU's NUMA cores, BEFORE anything initialises the GPU (sysfs only).  N = 1 stays
    # unbound: its CPU baseline is timed on all the box's cores
    try:
        affinity = D.bind_rank_to_gpu_numa(D.env_world()[1], int(os.environ.get("LOCAL_WORLD_SIZE", "1"))) \
            if D.env_world()[2] > 1 else {"bound": False, "why": "one rank"}
    except Exception as e:              # (placement is an optimisation: never a reason not to run)
        affinity = {"bound": False, "why": "bind_rank_to_gpu_numa raised %r" % (e,)}
    from ait_amd import _lib, ops, tuning
    rank, local_rank, world = D.init()
    # MIOpen solver picks for the torch-side convolutions: the committed find-db holds the shapes of cfg2 .. cfg5 as
    # configured (scripts/exp_miopen_db.py); any other batch size / variant would start a search (minutes) on the
    # first step, so those run MIOpen's immediate mode
    tuned = tuning.use_tuned_miopen_db(rank) if (args.bs == conf["bs"] and args.variant == conf["variant"]) else False
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d: refusing to report a line for another job size"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    _lib.lib()                                   # fail loudly if libait_hip.so is missing
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)

    ops.set_matmul_dtype(args.dtype)
    model = build_model(args.proposals, device, args.variant, conf["layers"])
    opt = make_optimizer(model)
    ddp = D.wrap(model, local_rank)
    np.random.seed(3 + rank)                     # reference RNG_SEED, one stream per rank
    batch = synth_batch(args.bs, 1000 + rank, device, max_gt=50 if args.variant == "coco" else 20)

    # (N > 1: the buckets' exchange -- DDP's all-reduce, or with --exchange rs_ag reduce-scatter +
    # all-gather -- under a clock that notes when the reducer hands each bucket over)
    exchange = args.exchange
    clock = D.BucketClock(ddp, exchange) if D.active() else None
    if D.active():
        ddp._ait_exchange = exchange

    def step():
        opt.zero_grad(set_to_none=True)
        out = ddp(*batch)
        cost = total_cost(out)
        if clock is not None:
            clock.start()
        cost.backward()
        if clock is not None:
            clock.stop()
        opt.step()

    peaks = measured_peaks(device) if rank == 0 else None
    for _ in range(args.warmup):
        step()
    # live roofline measurement: the library brackets every GEMM / RoIAlign launch of the timed region with
    # HIP events on its launch stream (include/ait_hip.h "Measurement"), wherever the launch comes from
    probe = _lib.Probe(1024 * max(1, args.steps))
    # every step between two HIP events on the step's stream; the line reports their MEDIAN as ms_per_step and
    # the wall-clock mean of the whole region (what `value` is computed from, per the bench contract) beside it
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with probe:
        for i in range(args.steps):
            marks[i].record()
            step()
        marks[args.steps].record()
    torch.cuda.synchronize()
    D.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = D.max_over_ranks(elapsed, device)
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    med = lambda v: 0.5 * (v[(len(v) - 1) // 2] + v[len(v) // 2])
    step_stats = {"median": D.max_over_ranks(med(step_ms), device), "mean": sum(step_ms) / len(step_ms),
                  "min": step_ms[0], "max": step_ms[-1],
                  "clock": "HIP events on the step's stream, one pair per step%s" % ("; median = max over ranks of the per-rank medians" if world > 1 else "")}

    # what the library's event brackets around every GEMM / RoIAlign launch cost the timed region: further steps outside it,
    # ALTERNATING with and without the brackets (the chip's clock drifts over seconds, so two back-to-back blocks of steps
    # would measure the drift), each between two events; the difference of the two medians (every rank runs them: the steps
    # hold collectives)
    n_np = 0 if args.no_probe_pass else max(2, min(args.steps, 10))
    probe2 = _lib.Probe(1024 * max(1, n_np))
    marks2 = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n_np + 1)]
    D.barrier()
    with probe2:
        for i in range(2 * n_np):
            _lib._ACTIVE_PROBE = probe2 if i % 2 == 0 else None
            marks2[i].record()
            step()
        marks2[2 * n_np].record()
    torch.cuda.synchronize()
    D.barrier()
    t_on = sorted(marks2[i].elapsed_time(marks2[i + 1]) for i in range(0, 2 * n_np, 2))
    t_off = sorted(marks2[i].elapsed_time(marks2[i + 1]) for i in range(1, 2 * n_np, 2))
    probe_overhead_ms = (med(t_on) - med(t_off)) if n_np else None

    # A/B beside the headline, OUTSIDE its timed region (every rank runs it: the steps hold collectives): the same step
    # with the AIT's products on the instruction that multiplies f32 operands (v_mfma_f32_32x32x2_f32)
    ab_native_ms = None
    if args.dtype == "f32" and not args.no_ab:
        ops.set_matmul_dtype("f32_native")
        try:
            for _ in range(2):
                step()
            D.barrier()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            n_ab = max(1, min(args.steps, 10))
            for _ in range(n_ab):
                step()
            torch.cuda.synchronize()
            D.barrier()
            ab_native_ms = D.max_over_ranks(time.perf_counter() - ta, device) / n_ab * 1e3
        finally:
            ops.set_matmul_dtype(args.dtype)

    # on a GPU box the product path is the library's kernels: not one torch stand-in may have run
    if ops.fallback_count() != 0:
        raise SystemExit("bench.py: torch fallbacks ran inside the step: %r" % (dict(ops.FALLBACKS),))
    if rank != 0:
        return
    entries = probe.entries()
    probe_overflow = probe.overflowed()
    # (flops, ms, (M, N, K, trans_a, trans_b, splits))
    prof = [(e[1], e[2], e[3]) for e in entries if e[0] == _lib.PROBE_GEMM]
    roi_prof = [(e[1], e[2], "fwd" if e[0] == _lib.PROBE_ROI_FWD else "bwd") for e in entries
                if e[0] in (_lib.PROBE_ROI_FWD, _lib.PROBE_ROI_BWD)]
    flops = sum(p[0] for p in prof)
    gemm_ms = sum(p[1] for p in prof)
    if args.gemm_table:
        by = {}
        for f, ms, dims in prof:
            a = by.setdefault(tuple(dims), [0, 0.0, 0.0])
            a[0] += 1; a[1] += ms; a[2] += f
        with open(args.gemm_table, "w") as fh:
            fh.write("# GEMM launches of the timed region by shape (M, N, K, trans_a, trans_b, splits); convolutions as their implicit GEMM\n")
            fh.write("# %8s %6s %7s ta tb spl | launches/step   ms/step   TFLOP/s\n" % ("M", "N", "K"))
            for dims, (n, ms, f) in sorted(by.items(), key=lambda kv: -kv[1][1]):
                fh.write("%10d %6d %7d %2d %2d %3d | %13.1f %9.3f %9.1f\n" % (dims + (n / args.steps, ms / args.steps, f / (ms * 1e-3) / 1e12)))
            fh.write("# total %.3f ms/step, %.1f TFLOP/s\n" % (gemm_ms / args.steps, flops / (gemm_ms * 1e-3) / 1e12))
    achieved = flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    pairs = world * args.bs * args.steps
    # dense MFMA peak for the arithmetic: fp32, bf16, or bf16 / 3 MFMAs per product
    # (f32: the bf16 pipe's dense peak over the SIX MFMAs an f32 block product costs on it -- the ceiling of the split
    # form; the f32 instruction's own peak is reported beside it)
    peak = {"f32": 2500.0 / 6, "f32_native": PEAK_F32_MFMA_TFLOPS, "bf16": 2500.0}[args.dtype]
    is_f32 = args.dtype in ("f32", "f32_native")
    pmc = pmc_traffic_per_launch() if is_f32 else None
    # algorithmic bytes of the same launches: each operand read once, the output written once
    alg = sum(4.0 * (p[2][0] * p[2][2] + p[2][1] * p[2][2] + p[2][0] * p[2][1]) for p in prof) / max(1, len(prof))
    def roi_entry(tag):
        ev = [p for p in roi_prof if p[2] == tag]
        if not ev:
            return None
        ms = sum(p[1] for p in ev) / len(ev)
        b = sum(p[0] for p in ev) / len(ev)
        gbs = b / (ms * 1e-3) / 1e9
        return {"kernel": "roi_align_nhwc_fwd_sliced_kernel" if tag == "fwd" else "roi_align_nhwc_bwd_tile_kernel",
                "bound": "hbm", "achieved": gbs,
                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "ms_per_launch": ms,
                "algorithmic_bytes_per_launch": b, "launches_per_step": len(ev) // max(1, args.steps),
                "traffic": pmc_traffic_per_launch("roi_align_nhwc_%s" % tag)}

    value = pairs / elapsed
    line = {
        "metric": METRIC, "value": value, "unit": "pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup,
        # the median of the event-timed steps; `value` = pairs / wall-clock of the whole region (mean)
        "ms_per_step": step_stats["median"], "ms_per_step_mean": 1e3 * elapsed / args.steps, "step_ms": step_stats,
        "probe_overhead_ms_per_step": probe_overhead_ms,
        "probe_overhead_is": "median of %d event-timed steps WITH the library's two HIP events around every GEMM / RoIAlign launch (as in "
                             "the timed region: they feed `roofline`) minus the median of %d steps without them, alternating, run "
                             "right after the timed region" % (n_np, n_np),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32 (operands, results, accumulation and storage f32; products: every operand value split exactly into "
                         "three bf16 values (round to nearest), six v_mfma_f32_32x32x16_bf16 per block, dropped terms <= 2^-23 |a b|)",
                  "f32_native": "f32 (AIT products on v_mfma_f32_32x32x2_f32; the proposal tail's and the RPN head's convolutions "
                                "keep the split-bf16 form)",
                  "bf16": "bf16 (BASELINE configs[4]): C4 trunk on MIOpen with bf16 tensors (f32 master weights, f32 accumulate, "
                          "frozen-BN passes on bf16); AIT: every linear on v_mfma_f32_32x32x16_bf16 with f32 accumulate -- feed-forward "
                          "hidden tensors, q / k / v and the attention blocks' gradients STORED in bf16 (bf16 operands from memory), the "
                          "embeddings' operands rounded to bf16 in registers; proposal tail in the library (ait_tail_*): layer4's "
                          "activations and gradients STORED in bf16, its 1x1 / 3x3 convolutions on the bf16-storage kernels (folded "
                          "bf16 weight copies per call), the SK blocks' operands rounded in registers; f32 residual stream, LayerNorm, "
                          "softmax and attention-tile arithmetic (on widened bf16 q / k / v), RPN, losses"}[args.dtype],
        "data": "synthetic",
        "config": {"workload": conf["workload"] % {"P": args.proposals, "bs": args.bs},
                   "name": args.config,
                   "variant": "voc (MultiHeadAttention co-attention, 9 anchors, 20 gt boxes)" if args.variant == "voc"
                              else "coco (non-local co-attention, 12 anchors, 50 gt boxes)",
                   "backbone": "ResNet%d" % conf["layers"],
                   "pairs_per_gpu": args.bs, "global_batch": world * args.bs,
                   "collective": D.collective_description(ddp), "host_affinity": affinity,
                   "gradient_buckets": clock.summary() if clock is not None else None,
                   "proposals": args.proposals, "target": "600x1000", "query": "128x128",
                   "parallelism": "dp%d" % world, "miopen_find_db": bool(tuned)},
        "roofline": {"bound": "mfma",
                     "kernel": "gemm_f32_stream_kernel: f32 operands, each split exactly into three bf16 values in registers; "
                               "six v_mfma_f32_32x32x16_bf16 (f32 accumulate) per 32x32x16 block = f32-equivalent products "
                               "(few-tile launches: gemm_f32_kernel on v_mfma_f32_32x32x2_f32)" if args.dtype == "f32"
                               else "gemm_f32_stream_kernel / gemm_f32_kernel (v_mfma_f32_32x32x2_f32)" if args.dtype == "f32_native"
                               else "the AIT's products: gemm_bf16s_kernel / gemm_bf16s_tn_kernel (bf16 operands stored in memory) and gemm_f32_stream_kernel with KNOB_BF16 (f32 operands rounded to bf16 in registers), one v_mfma_f32_32x32x16_bf16 per block",
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved / peak,
                     "peak_is": {"f32": "2500 TFLOP/s dense bf16 MFMA / 6 MFMAs per f32 block product",
                                 "f32_native": "v_mfma_f32_32x32x2_f32 dense peak", "bf16": "dense bf16 MFMA"}[args.dtype],
                     "f32_instruction_peak": PEAK_F32_MFMA_TFLOPS,
                     "achieved_over_f32_instruction_peak": achieved / PEAK_F32_MFMA_TFLOPS if is_f32 else None,
                     # PMC counters cannot be read from inside this process: `traffic` (HBM bytes per launch of the dominant
                     # kernel) comes from the COMMITTED rocprofv3 passes of this same command, taken on another box at the
                     # end of the round (profiles/README_r05.md); everything else in this object is measured live in this run
                     "traffic": pmc["bytes_per_launch"] if pmc else None,
                     "from_committed_profile": {"traffic_detail": pmc,
                                                "clock_and_mfma_util": pmc_clock_and_util() if is_f32 else None,
                                                "note": "not measured by this run: parsed from profiles/*, boxes differ by ~3 %"},
                     "algorithmic_bytes_per_launch": alg,
                     "launches_per_step": len(prof) // max(1, args.steps),
                     "probe_overflow": probe_overflow,      # launches beyond the probe's capacity (0: none truncated)
                     "gemm_ms_per_step": gemm_ms / max(1, args.steps),
                     "gemm_gflop_per_step": flops / max(1, args.steps) / 1e9,
                     # BASELINE.md 4 / SURVEY 8d: the whole AIT path end to end against the matrix peak --
                     # pairs/s x 0.911 TFLOP (as-computed-by-reference FLOPs of one pair) per GPU
                     "e2e_ait_frac": value / world * AIT_TFLOP_PER_PAIR / peak if is_f32 else None,
                     # what THIS box's matrix pipe sustains for the product form in use: a register-only loop of the same
                     # instruction on random operand bits (scripts/mfma_peak.hip), per f32-equivalent product
                     "measured_peak": (peaks["mfma_bf16_tflops"] / 6 if args.dtype == "f32" and peaks.get("mfma_bf16_tflops")
                                       else peaks["mfma_f32_tflops"] if args.dtype == "f32_native" else None) if peaks else None,
                     "measured_peak_is": "register-only v_mfma_f32_32x32x16_bf16 loop on random operands, sustained, / 6"
                                         if args.dtype == "f32" else "register-only v_mfma_f32_32x32x2_f32 loop",
                     "measured_bf16_pipe_tflops": peaks.get("mfma_bf16_tflops") if peaks else None,
                     "measured_f32_instruction_tflops": peaks.get("mfma_f32_tflops") if peaks else None},
        "roofline_roi_align": {"fwd": roi_entry("fwd"), "bwd": roi_entry("bwd"),
                               "measured_peak_gbs": peaks["hbm_copy_gbs"] if peaks else None},
    }
    if os.environ.get("AIT_BENCH_GEMM_TABLE"):
        import collections
        tab = collections.OrderedDict()
        for p in prof:
            t = tab.setdefault(p[2], [0.0, 0])
            t[0] += p[1]
            t[1] += 1
        for k, (ms, n) in sorted(tab.items(), key=lambda kv: -kv[1][0]):
            fl = 2.0 * k[0] * k[1] * k[2]
            print("gemm M=%6d N=%5d K=%6d ta=%d tb=%d splits=%2d : %2d/step %8.1f us  %6.1f TF/s  %5.2f ms/step"
                  % (k + (n // args.steps, 1e3 * ms / n, fl / (ms / n) / 1e9, ms / args.steps)), file=sys.stderr)
    r = line["roofline"]
    if ab_native_ms is not None:
        line["ab"] = {"f32_native_ms_per_step": ab_native_ms, "f32_native_pairs_per_s": world * args.bs / (ab_native_ms * 1e-3),
                      "what": "the same step with the AIT's products formed by v_mfma_f32_32x32x2_f32 (--dtype f32_native), "
                              "run right after the timed region; the tail's and the RPN head's convolutions keep the split form"}
    r["frac_of_measured_peak"] = (achieved / r["measured_peak"]) if r.get("measured_peak") else None
    if world == 1 and not args.no_cpu_baseline:
        # (the CPU port is timed on the headline workload's pair: VOC variant, ResNet50)
        line["cpu_baseline"] = cpu_baseline(args.proposals)
        line["cpu_baseline"]["cpu_model"] = cpu_model()
    # (RCCL writes its version banner through C stdio, which flushes at exit -- behind this line unless flushed first)
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
