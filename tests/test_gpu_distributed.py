"""The N>1 path on the REAL detector: two fresh child processes share the one GPU of the test box
(AIT_DIST_BACKEND=gloo: RCCL refuses two ranks on one device), each builds the product detector with
identical weights, trains on its own pair under DDP (ait_amd.distributed.wrap), and checks that
  (1) after backward every rank holds the same gradients, and
  (2) they equal the mean over ranks of the gradients each rank computes alone on its shard --
      the data-parallel contract of SURVEY 8e / trainval_net_voc.py:391-395 (mean of replica losses).
"""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import faulthandler, os, sys
    faulthandler.dump_traceback_later(150, exit=True)      # a rank that hangs says where (all threads), then exits
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from ait_amd import distributed as D, _lib
    from ait_amd.config import cfg_from_list
    from ait_amd.faster_rcnn import resnet
    from oracle import detector_ref as R          # (synthetic inputs only)
    rank, local_rank, world = D.init()
    assert world == 2 and dist.get_backend() == "gloo"
    _lib.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg_from_list(['TRAIN.BATCH_SIZE', 32])
    torch.manual_seed(1234)                        # identical initial weights on every rank
    m = resnet(('__background__', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3)
    m.create_architecture()
    m = m.to(dev).train()
    for mod in m.modules():                        # deterministic arithmetic: no dropout masks
        if hasattr(mod, "p") and isinstance(mod.p, float):
            mod.p = 0.0
    ins = [t.to(dev) for t in R.synth_inputs(1, 500 + rank, im_hw=(320, 480))]
    watch = ["transformer.encoder.layer_stack.0.slf_attn.w_qs.weight", "transformer.dec_trans.0.bias",
             "transformer.enc_emb.0.weight", "RCNN_cls_score.1.weight", "RCNN_bbox_pred.weight",
             "coattention.img_trans.0.weight", "RCNN_rpn.RPN_Conv.bias", "RCNN_base.backbone.layer4.2.conv3.weight",
             "RCNN_base.backbone.layer3.5.conv3.weight", "sk.sk_props.convs.1.0.weight"]
    params = dict(m.named_parameters())

    def step(model):
        m.zero_grad(set_to_none=True)
        np.random.seed(3 + rank)
        out = model(*ins)
        (out[3] + out[4] + out[5] + out[6] + out[7]).backward()
        return {k: params[k].grad.detach().clone() for k in watch}

    alone = step(m)                                # this rank's own gradient, no exchange
    ddp = D.wrap(m, local_rank, bucket_mb=8)
    assert ddp is not m
    # ---- when do the reducer's buckets become ready?  (SURVEY 8e: buckets launched as backward produces them.)  A
    # comm hook notes every bucket the reducer hands to the all-reduce, the AIT notes its three backward parts, a
    # gradient hook on the LAST convolution of the trunk (the first trunk layer the backward reaches) notes when the
    # trunk's backward has started.
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    from ait_amd import system
    events = []
    system.PART_TRACE = events
    def hook(state, bucket):
        events.append(("bucket", bucket.index()))
        return default_hooks.allreduce_hook(None, bucket)
    ddp.register_comm_hook(None, hook)
    params["RCNN_base.backbone.layer3.5.conv3.weight"].register_post_accumulate_grad_hook(lambda p_: events.append(("trunk",)))
    for _ in range(3):                             # static_graph settles (and rebuilds its buckets) in the first iterations
        del events[:]
        got = step(ddp)
    kinds = [e[0] for e in events]
    first_trunk = kinds.index("trunk")
    n_before = sum(k == "bucket" for k in kinds[:first_trunk])
    print("rank", rank, "buckets ready before the trunk backward:", n_before, "of", kinds.count("bucket"), flush=True)
    assert n_before >= 4, events
    # the AIT's gradients arrive in three bursts, and the reducer launches buckets BETWEEN them (while the rest of the
    # AIT backward is still being enqueued), not only after the whole operator's backward
    p0, p2 = events.index(("ait_part", 0)), events.index(("ait_part", 2))
    assert events.index(("ait_part", 1)) in range(p0, p2)
    assert any(k == "bucket" for k in kinds[p0:p2]), events
    assert params["RCNN_base.backbone.fc.weight"].grad is None      # never used: dropped from the buckets
    for k in watch:
        both = [torch.zeros_like(got[k]) for _ in range(world)]
        dist.all_gather(both, got[k])
        assert torch.equal(both[0], both[1]), "rank gradients differ: " + k
        mine = [torch.zeros_like(alone[k]) for _ in range(world)]
        dist.all_gather(mine, alone[k])
        mean = (mine[0] + mine[1]) / world
        rel = float((got[k] - mean).norm() / (mean.norm() + 1e-20))
        print("rank", rank, k, "rel", rel, flush=True)
        # (split-K weight-gradient kernels -- MIOpen's and this library's -- sum with atomics: two runs of the
        # same step differ by ~1e-4 relative; gradients of DIFFERENT shards differ by O(1))
        assert rel < 1e-3, (k, rel)
        assert float((mine[0] - mine[1]).norm()) > 0, "shards were not different: " + k
    t = D.max_over_ranks(1.0 + rank, dev)
    assert t == 2.0
    D.barrier()
    print("rank", rank, "ok")
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_train_the_real_detector(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), AIT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
                   GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append(p.communicate()[0])
    report = "\n".join("---- rank %d (exit %s) ----\n%s" % (r, p.returncode, o[-4000:]) for r, (p, o) in enumerate(zip(procs, outs)))
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed; both ranks' output:\n%s" % (r, report)
        assert "rank %d ok" % r in o


def test_bench_gpus_2_starts_its_own_two_ranks():
    """`python bench.py --gpus 2` outside a torchrun environment is ONE command that runs two ranks (the reference's
    `trainval_net_voc.py --mGPUs`, trainval_net_voc.py:83,321-326): the parent starts two fresh children before any
    GPU call and relays rank 0's line.  Here the two ranks share the box's one GPU over gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(AIT_DIST_BACKEND="gloo", GLOO_SOCKET_IFNAME="lo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-ab"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["config"]["parallelism"] == "dp2"
    assert "gloo" in line["config"]["collective"]
    gb = line["config"]["gradient_buckets"]
    assert gb["buckets"] >= 4
    # on the DEVICE (events on the gradients' stream): the first bucket is ready with most of the backward still ahead
    dev = gb["device"]
    assert 0 <= dev["first_ready_ms"] <= dev["last_ready_ms"] <= dev["backward_end_ms"] + 1e-3
    assert dev["first_ready_ms"] < 0.6 * dev["backward_end_ms"], dev
    assert line["value"] > 0 and line["roofline"]["achieved"] > 0
