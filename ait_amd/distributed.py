"""Data-parallel plumbing for the AIT path on one MI355X node: one process per GPU, RCCL
(`backend="nccl"` on ROCm) over xGMI.

The (target, query) pairs of a batch are independent, so the only exchange per
step is the gradient all-reduce; it is bucketed (~25 MB) and overlapped with backward by
torch's DistributedDataParallel reducer.  The reference's single-process nn.DataParallel
(trainval_net_voc.py:321-326: per-step parameter broadcast + gather through device 0) is not
reproduced.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None):
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # AIT_DIST_BACKEND=gloo lets two ranks share ONE GPU (RCCL refuses duplicate devices):
            # used to exercise the N>1 path on a single-GPU box
            backend = os.environ.get("AIT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def wrap(model, local_rank, bucket_mb=25):
    """DDP with a static graph: parameters that never receive a gradient (the never-used
    RCNN_base.backbone.fc and the SKBlock fc/sk that the reference computes but does not use)
    are found on the first iteration and dropped from the reduction."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return model
    from torch.nn.parallel import DistributedDataParallel as DDP
    ids = [next(model.parameters()).device.index] if next(model.parameters()).is_cuda else None
    # buffers (frozen BatchNorm statistics, positional tables) never change: no per-step broadcast
    return DDP(model, device_ids=ids, bucket_cap_mb=bucket_mb, gradient_as_bucket_view=True,
               static_graph=True, broadcast_buffers=False)


def collective_description(ddp):
    """what the gradient exchange of this run is, for the bench line: backend, bucket size, bucket count (DDP's
    reducer rebuilds its buckets in gradient-arrival order after the first iteration) and the RCCL algorithm /
    protocol the environment pins, if any"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return "none"
    kind = getattr(ddp, "_ait_exchange", "allreduce")
    desc = ("DDP gradient all-reduce (sum / world)" if kind == "allreduce" else
            "DDP buckets exchanged by reduce-scatter + all-gather (sum of 1/N shards, averaged, gathered)") + ", backend %s" % dist.get_backend()
    if dist.get_backend() == "nccl":
        desc += " = RCCL over xGMI"
    try:
        n_buckets = len(ddp.reducer._get_zeros_like_grad_buckets())
        desc += ", %d buckets" % n_buckets
    except Exception:
        pass
    desc += ", bucket cap %d MB, overlapped with backward" % int(getattr(ddp, "bucket_bytes_cap", 25 << 20) >> 20)
    algo, proto = os.environ.get("NCCL_ALGO"), os.environ.get("NCCL_PROTO")
    desc += ", NCCL_ALGO=%s NCCL_PROTO=%s" % (algo or "default (RCCL tuner)", proto or "default")
    return desc


# ---- the gradient exchange of a bucket ----------------------------------------------------------------------------------
# "allreduce" (default): DDP's own hook -- RCCL picks the algorithm.
# "rs_ag" (opt-in: AIT_DDP_EXCHANGE=rs_ag or bench.py --exchange rs_ag): reduce-scatter + all-gather, the exchange SURVEY 8e
# sketches for a fully connected xGMI node -- every rank sums ONE 1/N shard of the bucket (N - 1 direct peer transfers per
# rank, no ring), averages it, and the shards are gathered back.  The same sum as the all-reduce up to the order of the
# additions.  NEVER RUN ON RCCL HERE (no multi-GPU box in this round): its tensor logic -- padding to a multiple of N, the
# average, the copy back -- is held against the default hook by a two-rank gloo test with the two collectives emulated
# (tests/test_distributed_cpu.py); the two RCCL calls themselves are the untested lines.
EXCHANGES = ("allreduce", "rs_ag")


def _rs_future(shard, flat, group):
    return dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=group, async_op=True).get_future()


def _ag_future(flat, shard, group):
    return dist.all_gather_into_tensor(flat, shard, group=group, async_op=True).get_future()


def make_exchange_hook(kind="allreduce", reduce_scatter=_rs_future, all_gather=_ag_future):
    """a DDP comm hook (state = the process group or None) for the named exchange; the two collectives of "rs_ag" are
    injectable (callables returning futures) so that the hook's own logic can be tested on a backend without them"""
    if kind not in EXCHANGES:
        raise ValueError("exchange must be one of %s" % (EXCHANGES,))
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    if kind == "allreduce":
        return default_hooks.allreduce_hook

    def hook(state, bucket):
        group = state if state is not None else dist.group.WORLD
        world = dist.get_world_size(group)
        buf = bucket.buffer()
        n = buf.numel()
        pad = (-n) % world
        flat = buf if pad == 0 else torch.cat([buf, buf.new_zeros(pad)])
        shard = torch.empty(flat.numel() // world, dtype=flat.dtype, device=flat.device)

        def gather(_):
            shard.div_(world)
            all_gather(flat, shard, group).wait()
            if pad:
                buf.copy_(flat[:n])
            return buf
        return reduce_scatter(shard, flat, group).then(gather)
    return hook


def exchange_from_env():
    kind = os.environ.get("AIT_DDP_EXCHANGE", "allreduce")
    if kind not in EXCHANGES:
        raise ValueError("AIT_DDP_EXCHANGE must be one of %s" % (EXCHANGES,))
    return kind


class BucketClock:
    """When does the reducer hand its buckets to the all-reduce?  A comm hook (the default all-reduce, plus two notes per
    bucket) over a DDP-wrapped model: `start()` before backward(), `stop()` behind it, then `summary()` says when the
    first and the last bucket of that backward became ready, relative to start --
      * on the HOST clock: when the reducer ENQUEUED the exchange, and
      * on the DEVICE: an event recorded on the gradients' stream at that moment, i.e. behind the kernels that produced
        the bucket and in front of everything the backward enqueues later; against the event `stop()` records at the end
        of the backward this shows how much of the backward's device time was still ahead when the bucket could go
        (CUDA tensors only; needs a synchronize before summary(), which bench.py's timed region ends with)."""

    def __init__(self, ddp, exchange="allreduce", inner=None):
        import time
        inner = inner if inner is not None else make_exchange_hook(exchange)
        self.exchange = exchange
        self._time, self.t0, self.stamps, self.events = time, None, [], []
        self.ev0 = self.ev1 = None
        self._cuda = next(ddp.parameters()).is_cuda

        def hook(state, bucket):
            if self.t0 is not None:
                self.stamps.append((self._time.perf_counter() - self.t0) * 1e3)
                if self._cuda:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    self.events.append(ev)
            return inner(None, bucket)
        ddp.register_comm_hook(None, hook)

    def start(self):
        self.t0, self.stamps, self.events, self.ev1 = self._time.perf_counter(), [], [], None
        if self._cuda:
            self.ev0 = torch.cuda.Event(enable_timing=True)
            self.ev0.record()

    def stop(self):
        if self._cuda and self.ev0 is not None:
            self.ev1 = torch.cuda.Event(enable_timing=True)
            self.ev1.record()

    def summary(self):
        if not self.stamps:
            return None
        out = {"buckets": len(self.stamps), "first_ready_ms": self.stamps[0], "last_ready_ms": self.stamps[-1],
               "clock": "host, from the call of backward()"}
        if self._cuda and self.events and self.ev1 is not None:
            try:
                self.ev1.synchronize()
                out["device"] = {"first_ready_ms": self.ev0.elapsed_time(self.events[0]),
                                 "last_ready_ms": self.ev0.elapsed_time(self.events[-1]),
                                 "backward_end_ms": self.ev0.elapsed_time(self.ev1),
                                 "clock": "device events on the gradients' stream, from the start of backward()"}
            except RuntimeError:
                pass
        return out


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_slice(n_items, rank, world):
    """Contiguous even split of n_items work units (pairs) over ranks."""
    per = n_items // world
    extra = n_items % world
    start = rank * per + min(rank, extra)
    return start, start + per + (1 if rank < extra else 0)
