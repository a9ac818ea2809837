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


def forced():
    """AIT_FORCE_DDP=1: build the process group and the DDP wrapper at WORLD_SIZE 1 as well, so that the code every
    rank of an N > 1 job runs -- init_process_group on RCCL, DDP's reducer, the bucket hooks, reduce_scatter_tensor /
    all_gather_into_tensor -- executes on a box with ONE GPU (tests/test_gpu_distributed.py).  Not a product mode:
    at world size 1 the exchange moves nothing."""
    return os.environ.get("AIT_FORCE_DDP") == "1"


def init(backend=None):
    rank, local_rank, world = env_world()
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # ROCr shares device memory between the ranks of a node (RCCL's intra-node transport, torch's CUDA-tensor
        # sharing) through IPC handles; its legacy handle kind is not supported by this pool's host driver
        # (hipIpcGetMemHandle: invalid argument) -- "0" selects the dmabuf kind, which is.  It costs nothing on the
        # data path (a handle is exchanged once per buffer at communicator setup); setdefault: the caller's choice wins.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # AIT_DIST_BACKEND=gloo lets two ranks share ONE GPU (RCCL refuses duplicate devices):
            # used to exercise the N>1 path on a single-GPU box
            backend = os.environ.get("AIT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def active():
    """is there a gradient exchange in this process (N > 1, or the forced world-size-1 group)?"""
    return dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def wrap(model, local_rank, bucket_mb=25):
    """DDP with a static graph: parameters that never receive a gradient (the never-used
    RCNN_base.backbone.fc and the SKBlock fc/sk that the reference computes but does not use)
    are found on the first iteration and dropped from the reduction."""
    if not active():
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
    if not active():
        return "none"
    kind = getattr(ddp, "_ait_exchange", "allreduce")
    desc = ("DDP gradient all-reduce (sum / world)" if kind == "allreduce" else
            "DDP buckets exchanged by reduce-scatter + all-gather (sum of 1/N shards, averaged, gathered; "
            + RS_AG_STATUS + ")") + ", backend %s" % dist.get_backend()
    if dist.get_backend() == "nccl":
        desc += " = RCCL over xGMI"
    if dist.get_world_size() == 1:
        desc += ", WORLD SIZE 1 (AIT_FORCE_DDP: the exchange runs but moves nothing)"
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
# "rs_ag" (opt-in, ONLY by an explicit argument: bench.py --exchange rs_ag; no environment variable selects it):
# reduce-scatter + all-gather, the exchange SURVEY 8e sketches for a fully connected xGMI node -- every rank sums ONE 1/N
# shard of the bucket (N - 1 direct peer transfers per rank, no ring), averages it, and the shards are gathered back.  The
# same sum as the all-reduce up to the order of the additions.
# What has run where: the hook's tensor logic (padding to a multiple of N, the average, the copy back, the future DDP
# waits on) against the default hook on two gloo ranks with the two collectives emulated (tests/test_distributed_cpu.py);
# the two RCCL calls and the stream ordering of the chained futures on a real `nccl` group at WORLD SIZE 1 -- the one GPU
# of the test boxes (tests/test_gpu_distributed.py: gradients equal to the all-reduce hook's and to the unwrapped
# model's); on N >= 2 GPUs only where the test box has them (the same file's device_count() >= 2 test).  Hence experimental.
EXCHANGES = ("allreduce", "rs_ag")
RS_AG_STATUS = "EXPERIMENTAL: verified on RCCL at world size 1 and on two gloo ranks only"


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

    # per bucket index: the padded flat copy (only when the bucket's length is not a multiple of the world size) and the
    # 1/N shard, allocated once -- the reducer's buckets keep their sizes after its one rebuild, and a bucket's buffers
    # are not touched again before DDP has waited on the future of its previous exchange
    cache = {}

    def buffers(bucket, buf, world):
        n = buf.numel()
        pad = (-n) % world
        key = (bucket.index(), n, world, buf.dtype, buf.device)
        hit = cache.get(bucket.index())
        if hit is None or hit[0] != key:
            flat = buf.new_zeros(n + pad) if pad else None          # (the tail stays zero: sums and gathers of zeros)
            shard = buf.new_empty((n + pad) // world)
            hit = cache[bucket.index()] = (key, flat, shard)
        return hit[1], hit[2], n, pad

    def hook(state, bucket):
        group = state if state is not None else dist.group.WORLD
        world = dist.get_world_size(group)
        buf = bucket.buffer()
        padded, shard, n, pad = buffers(bucket, buf, world)
        if pad:
            padded[:n].copy_(buf)
        flat = padded if pad else buf
        # the future DDP waits on: completed by the all-gather's callback.  No wait() anywhere: each callback runs when
        # its collective has been ENQUEUED (CUDA futures: on a stream ordered behind it), issues the next piece on that
        # stream and returns -- the reducer's thread is never held inside a callback
        done = torch.futures.Future(devices=[buf.device]) if buf.is_cuda else torch.futures.Future()

        def gathered(f):
            try:
                f.value()
                if pad:
                    buf.copy_(flat[:n])
                done.set_result(buf)
            except Exception as e:          # surface a failed collective through the future DDP holds
                done.set_exception(e)

        def scattered(f):
            try:
                f.value()
                shard.div_(world)
                all_gather(flat, shard, group).add_done_callback(gathered)
            except Exception as e:
                done.set_exception(e)
        reduce_scatter(shard, flat, group).add_done_callback(scattered)
        return done
    hook.cache = cache
    return hook


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


# ---- host placement: a rank's threads next to its GPU --------------------------------------------------------------------
# The host side of a step is not idle (NumPy sampling per image, ~700 kernel launches): a rank whose threads sit on the
# other socket pays a cross-socket hop per launch.  bind_rank_to_gpu_numa() pins the CALLING process to the cores of the NUMA
# node its GPU hangs off, read from sysfs only -- it must run BEFORE anything initialises the GPU, and it is a plain
# sched_setaffinity in this process: never numactl / taskset around the program (under rocprofv3 such a wrapper is an exec
# behind an initialised GPU, which this pool forbids).

def _parse_cpulist(s):
    cpus = set()
    for part in s.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def _visible_gpu_order(n_nodes):
    """indices into the KFD GPU-node list that HIP numbers 0, 1, ... (ROCR_VISIBLE_DEVICES filters the runtime's agents,
    then HIP_/CUDA_VISIBLE_DEVICES filter again); None when a list is not plain integers (UUIDs: no guess)"""
    order = list(range(n_nodes))
    hip = os.environ.get("HIP_VISIBLE_DEVICES")
    for v in (os.environ.get("ROCR_VISIBLE_DEVICES"), hip if hip is not None else os.environ.get("CUDA_VISIBLE_DEVICES")):
        if v is None:
            continue
        try:
            pick = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            return None
        order = [order[i] for i in pick if 0 <= i < len(order)]
    return order


def gpu_numa_cpus(local_rank, sysfs="/sys"):
    """(numa_node, set of cpus) of HIP device `local_rank`, or (None, None) when sysfs does not say.  KFD lists the
    topology nodes in the order the runtime enumerates its agents; a GPU node has simd_count > 0 and names its PCI
    function (domain, location_id = bus << 8 | devfn), whose sysfs entry carries numa_node and local_cpulist."""
    top = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        nodes = sorted((int(d) for d in os.listdir(top) if d.isdigit()))
    except OSError:
        return None, None
    gpus = []
    for nd in nodes:
        props = {}
        try:
            with open(os.path.join(top, str(nd), "properties")) as fh:
                for line in fh:
                    k, _, v = line.strip().partition(" ")
                    props[k] = v
        except OSError:
            continue
        if int(props.get("simd_count", "0") or 0) > 0:
            gpus.append(props)
    order = _visible_gpu_order(len(gpus))
    if order is None or not (0 <= local_rank < len(order)):
        return None, None
    p = gpus[order[local_rank]]
    try:
        loc, dom = int(p["location_id"]), int(p.get("domain", "0") or 0)
    except (KeyError, ValueError):
        return None, None
    bdf = "%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
    dev = os.path.join(sysfs, "bus", "pci", "devices", bdf)
    try:
        with open(os.path.join(dev, "numa_node")) as fh:
            node = int(fh.read().strip())
        with open(os.path.join(dev, "local_cpulist")) as fh:
            cpus = _parse_cpulist(fh.read())
    except (OSError, ValueError):
        return None, None
    if node < 0:                                       # a single-node host reports -1: nothing to choose
        return None, None
    return node, cpus


def bind_rank_to_gpu_numa(local_rank, local_world=1, sysfs="/sys", setaffinity=None):
    """pin this process to its GPU's NUMA cores (intersected with what it may already run on); when several ranks share
    the node each takes an even contiguous share.  Returns a description for the bench line; never raises: a box whose
    sysfs does not answer is left as it is."""
    setaffinity = setaffinity or (lambda cpus: os.sched_setaffinity(0, cpus))
    try:
        allowed = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        return {"bound": False, "why": "no sched_getaffinity on this host"}
    node, cpus = gpu_numa_cpus(local_rank, sysfs)
    if node is None:
        return {"bound": False, "why": "sysfs names no NUMA node for GPU %d" % local_rank}
    cpus = sorted(cpus & allowed)
    if not cpus:
        return {"bound": False, "why": "none of NUMA node %d's cores is in this process's affinity mask" % node}
    peers = [r for r in range(local_world) if gpu_numa_cpus(r, sysfs)[0] == node] or [local_rank]
    if len(peers) > 1 and len(cpus) >= len(peers):
        k = peers.index(local_rank) if local_rank in peers else 0
        per = len(cpus) // len(peers)
        cpus = cpus[k * per:(k + 1) * per]
    try:
        setaffinity(set(cpus))
    except OSError as e:
        return {"bound": False, "why": "sched_setaffinity: %s" % e}
    return {"bound": True, "numa_node": node, "cpus": len(cpus), "first_cpu": cpus[0], "last_cpu": cpus[-1],
            "ranks_on_node": len(peers)}
