"""Thin tensor-level wrappers over the C ABI (no autograd here; see ait_amd/system.py)."""
import ctypes

import torch

from . import _lib


def _p(t, dtype=torch.float32):
    return None if t is None else _lib.dev_ptr(t, dtype)


def gemm(a, b, trans_a=False, trans_b=True, out=None, bias=None, residual=None, relu=False,
         accumulate=False, alpha=1.0, split_k=1, bias_row=False, c_colblk=0, c_batch_stride=0,
         out_shape=None):
    """out (op)= alpha * op(a) @ op(b) (+bias)(+residual)(relu) on the fp32 matrix cores.

    a: [M,K] (or [K,M] if trans_a);  b: [N,K] if trans_b (nn.Linear weight layout) else [K,N].
    2-D, last dim contiguous; row pitch taken from stride(0).
    """
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    N = b.shape[0] if trans_b else b.shape[1]
    Kb = b.shape[1] if trans_b else b.shape[0]
    if K != Kb:
        raise ValueError("gemm: reduction dims differ (%d vs %d)" % (K, Kb))
    if out is None:
        assert c_colblk == 0 or out_shape is not None
        out = torch.empty(out_shape or (M, N), dtype=torch.float32, device=a.device)
        if accumulate or split_k > 1:
            out.zero_()
    ldc = c_colblk if c_colblk > 0 else out.stride(0)
    flags = (_lib.GEMM_RELU if relu else 0) \
        | (_lib.GEMM_ACCUMULATE if accumulate and split_k == 1 else 0) \
        | (_lib.GEMM_ATOMIC if split_k > 1 else 0) | (_lib.GEMM_BIAS_ROW if bias_row else 0)
    with torch.cuda.device(a.device):
        rc = _lib.lib().ait_gemm_f32(
            int(trans_a), int(trans_b), M, N, K, float(alpha), _lib.dev_ptr(a), a.stride(0),
            _lib.dev_ptr(b), b.stride(0), ctypes.c_void_p(out.data_ptr()), ldc, _p(bias),
            _p(residual), flags, int(split_k), int(c_colblk), int(c_batch_stride),
            _lib.cur_stream(a.device))
    _lib.check(rc, "ait_gemm_f32")
    return out
