"""HIP-graph replay of a fixed-shape sub-network, forward and backward.

The C4 trunk on the query patches (faster_rcnn._fasterRCNN._query_trunk) is ~130 forward and ~260 backward
launches of a few microseconds each on ONE input size: the GPU finishes them faster than the host can issue
them, so they are captured once into two HIP graphs and replayed (MI355X playbook: "capture launch-bound inner
loops in hipGraphs").  Same kernels, same order, same values as the eager launches.

torch.cuda.make_graphed_callables does the same job but differentiates with respect to the module's own
parameter tensors; their gradient-accumulator nodes carry the stream they were first used on, and whenever such a
node is alive while the backward is captured (the target image went through the same trunk a moment ago; DDP
stashes every accumulator at construction) the autograd engine inserts a wait on THAT stream into the capture, which
pulls the legacy default stream into it -- hipStreamEndCapture then takes the process down (ROCm 7.2).  Here the
captured region differentiates with respect to fresh ALIASES of the parameters (same storage, so the optimizer's
in-place updates are seen; no history, so no foreign stream), and an autograd node hands the replayed gradients to
the real parameters.
"""
import contextlib

import torch
from torch.autograd.function import once_differentiable


@contextlib.contextmanager
def _parameters_replaced(module, by_id):
    """every submodule's parameter slots hold by_id[id(parameter)] inside the block and the parameters themselves
    again after it.  (torch.func.functional_call leaves its substitutes inside modules that are reachable under two
    names -- RCNN_base.layer1 is RCNN_base.backbone.layer1 -- so the swap is done by hand, per owning module.)"""
    undo = []
    try:
        for sub in module.modules():
            for name, p in list(sub._parameters.items()):
                if p is not None and id(p) in by_id:
                    undo.append((sub, name, p))
                    sub._parameters[name] = by_id[id(p)]
        yield
    finally:
        for sub, name, p in undo:
            sub._parameters[name] = p


class GraphedModule:
    """`module(x)` for one input shape, replayed from a forward and a backward HIP graph.

    Limits (those of any static capture): one replay in flight -- the output, the saved activations and the
    returned gradients are overwritten by the next call, so call -> backward -> call; the parameters must keep their
    storage and requires_grad flags (faster_rcnn keys its graphs on both and re-captures otherwise); no double
    backward."""

    def __init__(self, module, sample, warmup=3):
        if not sample.is_cuda:
            raise ValueError("GraphedModule: a GPU tensor is required")
        self.module = module
        self.real = list(module.parameters())
        leaves = [p.detach().requires_grad_(p.requires_grad) for p in self.real]
        by_id = {id(p): a for p, a in zip(self.real, leaves)}

        def run(x):
            with _parameters_replaced(module, by_id):
                return module(x)

        self.static_in = torch.zeros_like(sample)
        cur = torch.cuda.current_stream(sample.device)
        side = torch.cuda.Stream(sample.device)
        side.wait_stream(cur)
        used = [i for i, p in enumerate(leaves) if p.requires_grad]
        with torch.enable_grad(), torch.cuda.stream(side):
            for _ in range(warmup):                      # MIOpen picks its kernels, the allocator its blocks
                out = run(self.static_in)
                grads = torch.autograd.grad(out, [leaves[i] for i in used], torch.ones_like(out), allow_unused=True)
                used = [i for i, g in zip(used, grads) if g is not None]
            del out, grads
        cur.wait_stream(side)
        torch.cuda.synchronize(sample.device)
        self.used = used
        pool = torch.cuda.graph_pool_handle()
        self.fwd = torch.cuda.CUDAGraph()
        self.bwd = torch.cuda.CUDAGraph()
        with torch.enable_grad():
            with torch.cuda.graph(self.fwd, pool=pool):
                out = run(self.static_in)
            self.static_gout = torch.zeros_like(out)
            with torch.cuda.graph(self.bwd, pool=pool):
                grads = torch.autograd.grad(out, [leaves[i] for i in used], self.static_gout)
        self.static_out = out.detach()
        self.static_grads = tuple(grads)
        self._leaves = leaves                             # keeps the captured autograd graph's leaves alive

    def __call__(self, x):
        return _Replay.apply(self, x, *[self.real[i] for i in self.used])


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, unit, x, *params):
        x.record_stream(torch.cuda.current_stream(x.device))      # (the caller may replay on a side stream)
        unit.static_in.copy_(x)
        unit.fwd.replay()
        ctx.unit = unit
        return unit.static_out.detach()

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        unit = ctx.unit
        g.record_stream(torch.cuda.current_stream(g.device))
        unit.static_gout.copy_(g)
        unit.bwd.replay()
        # the static tensors themselves (the unit keeps a reference): autograd then never adopts or accumulates
        # into them in place, so a parameter's .grad cannot end up aliasing memory the next replay overwrites
        return (None, None) + unit.static_grads
