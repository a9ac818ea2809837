"""Thin tensor-level wrappers over the C ABI (no autograd here; see ait_amd/system.py)."""
import collections
import ctypes

import torch

from . import _lib


def _p(t, dtype=torch.float32):
    return None if t is None else _lib.dev_ptr(t, dtype)


# Torch compositions that stand in for a kernel of the library exist for ONE reason: CPU tensors (the host-logic
# tests, which run without a GPU).  Every time one runs it is counted here, by site; bench.py and the full-size GPU
# tests assert that the count stays zero -- on a GPU box the product path is the library's kernels or an error.
FALLBACKS = collections.Counter()


def note_fallback(site, tensor=None):
    FALLBACKS[site] += 1


def fallback_count():
    return sum(FALLBACKS.values())


def reset_fallbacks():
    FALLBACKS.clear()


# the product form of the dense contractions; see set_matmul_dtype for what each name means.  "f32" (the default, the
# parity / headline path) is f32 in and out with the products formed on the bf16 pipe from an exact three-way split.
MATMUL_DTYPE = "f32"


def set_matmul_dtype(dtype):
    """"f32" (default): f32 operands, f32 results; products formed on the bf16 matrix pipe from an exact 3-way bf16
    split of every operand value (include/ait_hip.h, ait_launch_ctx::flags) -- f32-equivalent accuracy.
    "f32_native": the same kernels on v_mfma_f32_32x32x2_f32 (the A/B for the above).
    "bf16": the same kernels with every operand value rounded to bf16 in registers, one MFMA per block, f32
    accumulate, f32 tensors in memory (torch.autocast semantics; BASELINE configs[4]) -- NOT f32 accuracy.
    "bf16_lds": the round-1 bf16 kernel (operands converted on their way into LDS)."""
    global MATMUL_DTYPE
    if dtype not in ("f32", "f32_native", "bf16", "bf16_lds"):
        raise ValueError(dtype)
    _lib.NATIVE_F32 = dtype == "f32_native"
    _lib.BF16_PRODUCTS = dtype == "bf16"
    MATMUL_DTYPE = {"f32_native": "f32", "bf16": "f32", "bf16_lds": "bf16"}.get(dtype, dtype)


def _gemm_fn(exact=False):
    L = _lib.lib()
    if exact:
        return L.ait_gemm_f32
    return {"f32": L.ait_gemm_f32, "bf16": L.ait_gemm_bf16}[MATMUL_DTYPE]


def gemm(a, b, trans_a=False, trans_b=True, out=None, bias=None, residual=None, relu=False,
         accumulate=False, alpha=1.0, split_k=1, bias_row=False, c_colblk=0, c_batch_stride=0,
         out_shape=None, exact=False):
    """out (op)= alpha * op(a) @ op(b) (+bias)(+residual)(relu) on the fp32 matrix cores.

    a: [M,K] (or [K,M] if trans_a);  b: [N,K] if trans_b (nn.Linear weight layout) else [K,N].
    2-D, last dim contiguous; row pitch taken from stride(0).  exact=True pins the launch to the
    fp32 kernel whatever set_matmul_dtype says (the convolution-side products).
    """
    assert a.dim() == 2 and b.dim() == 2
    if a.stride(1) != 1 or a.stride(0) % 4 or a.stride(0) < a.shape[1]:
        a = a.contiguous()
    if b.stride(1) != 1 or b.stride(0) % 4 or b.stride(0) < b.shape[1]:
        b = b.contiguous()
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
    else:
        # a caller-supplied destination is written in place: it must be exactly what the kernel
        # addresses.  With split_k > 1 the partial tiles are ADDED to `out` with atomics: the caller
        # has zeroed it (or wants the product accumulated into what it holds).
        if not out.is_cuda or out.device != a.device or out.dtype != torch.float32:
            raise _lib.AitHipError("gemm: `out` must be a float32 tensor on %s" % (a.device,))
        if c_colblk == 0:
            if out.dim() != 2 or tuple(out.shape) != (M, N) or out.stride(1) != 1 or out.stride(0) < N:
                raise _lib.AitHipError("gemm: `out` must be [%d, %d] with unit column stride (got %s, strides %s)"
                                       % (M, N, tuple(out.shape), out.stride()))
        elif not out.is_contiguous() or (out_shape is not None and tuple(out.shape) != tuple(out_shape)):
            raise _lib.AitHipError("gemm: column-blocked `out` must be contiguous and of shape out_shape")
    ldc = c_colblk if c_colblk > 0 else out.stride(0)
    # (an empty reduction, K == 0, never reads A or B: any legal pitch will do)
    lda, ldb = (a.stride(0), b.stride(0)) if K > 0 else (4, 4)
    flags = (_lib.GEMM_RELU if relu else 0) \
        | (_lib.GEMM_ACCUMULATE if accumulate and split_k == 1 else 0) \
        | (_lib.GEMM_ATOMIC if split_k > 1 else 0) | (_lib.GEMM_BIAS_ROW if bias_row else 0)
    with torch.cuda.device(a.device):
        rc = _gemm_fn(exact)(
            int(trans_a), int(trans_b), M, N, K, float(alpha), _lib.dev_ptr(a), lda,
            _lib.dev_ptr(b), ldb, ctypes.c_void_p(out.data_ptr()), ldc, _p(bias),
            _p(residual), flags, int(split_k), int(c_colblk), int(c_batch_stride),
            _lib.launch_ctx(a.device), _lib.cur_stream(a.device))
    _lib.check(rc, "ait_gemm_f32")
    return out


def _pad4(n):
    return (n + 3) & ~3


def bmat(batch, rows, cols, device):
    """zero-filled [*batch, rows, cols] whose row pitch is a multiple of 4 floats (the layout bgemm operands
    need when cols % 4 != 0: the 2394-token image side of the co-attention); `batch` an int or a tuple"""
    batch = (batch,) if isinstance(batch, int) else tuple(batch)
    return torch.zeros(batch + (rows, _pad4(cols)), dtype=torch.float32, device=device)[..., :cols]


def bgemm(a, b, trans_a=False, trans_b=True, alpha=1.0, out=None, accumulate=False):
    """Batched out[i(,j)] (op)= alpha * op(a[i(,j)]) @ op(b[i(,j)]) in ONE launch (ait_gemm_f32_batched).
    a, b, out: 3-D [batch, rows, cols] or 4-D [batch, batch2, rows, cols] float32 GPU tensors (any batch
    strides -- e.g. the per-head views of a [tokens, 8*64] projection) with unit column stride and row / batch
    pitches that are multiples of 4 floats; an operand that is not is copied into such a layout."""
    nb = a.dim() - 2
    if nb not in (1, 2) or b.dim() != a.dim() or a.shape[:nb] != b.shape[:nb]:
        raise ValueError("bgemm: operand batch shapes differ")
    M, K = (a.shape[-1], a.shape[-2]) if trans_a else (a.shape[-2], a.shape[-1])
    N = b.shape[-2] if trans_b else b.shape[-1]
    Kb = b.shape[-1] if trans_b else b.shape[-2]
    if K != Kb:
        raise ValueError("bgemm: reduction dims differ (%d vs %d)" % (K, Kb))

    def ok(t):
        return t.stride(-1) == 1 and t.stride(-2) % 4 == 0 and t.stride(-2) >= t.shape[-1] and \
            all(st % 4 == 0 for st in t.stride()[:nb]) and t.data_ptr() % 16 == 0

    def fix(t):
        if ok(t):
            return t
        t2 = bmat(tuple(t.shape[:nb]), t.shape[-2], t.shape[-1], t.device)
        t2.copy_(t)
        return t2
    a, b = fix(a), fix(b)
    # a long reduction of few small products (the 2394-token side as the reduction dim): split-K, partial sums
    # combined with atomics into a zero-filled result
    n_prob = a.shape[0] * (a.shape[1] if nb == 2 else 1)
    split_k = 8 if (K >= 1024 and n_prob * ((M + 63) // 64) * ((N + 63) // 64) <= 256 and not accumulate) else 1
    if out is None:
        out = bmat(tuple(a.shape[:nb]), M, N, a.device)
    elif split_k > 1:
        out.zero_()
    for t in (a, b, out):
        if not t.is_cuda or t.dtype != torch.float32:
            raise _lib.AitHipError("bgemm: float32 GPU tensors only (no CPU fallback)")
    if tuple(out.shape) != tuple(a.shape[:nb]) + (M, N) or out.stride(-1) != 1:
        raise _lib.AitHipError("bgemm: bad `out`")
    s1 = lambda t: t.stride(0)
    s2 = lambda t: t.stride(1) if nb == 2 else 0
    with torch.cuda.device(a.device):
        rc = _lib.lib().ait_gemm_f32_batched(
            int(trans_a), int(trans_b), M, N, K, float(alpha), ctypes.c_void_p(a.data_ptr()), a.stride(-2), s1(a), s2(a),
            ctypes.c_void_p(b.data_ptr()), b.stride(-2), s1(b), s2(b), ctypes.c_void_p(out.data_ptr()), out.stride(-2),
            s1(out), s2(out), a.shape[0], a.shape[1] if nb == 2 else 1,
            _lib.GEMM_ATOMIC if split_k > 1 else (_lib.GEMM_ACCUMULATE if accumulate else 0), split_k,
            _lib.launch_ctx(a.device), _lib.cur_stream(a.device))
    _lib.check(rc, "ait_gemm_f32_batched")
    return out


def softmax_rows(x, p, seed):
    """x [..., rows, cols] (row pitch ld = x.stride(-2), dense batch dims) -> (y, y_drop) of the same layout"""
    cols, ld = x.shape[-1], x.stride(-2)
    rows = x.numel() // cols
    base = x.as_strided((rows, cols), (ld, 1))
    y = torch.zeros((rows, ld), dtype=torch.float32, device=x.device)[:, :cols]        # (zero row padding)
    yd = torch.zeros((rows, ld), dtype=torch.float32, device=x.device)[:, :cols] if p > 0 else y
    with torch.cuda.device(x.device):
        rc = _lib.lib().ait_softmax_rows_fwd(ctypes.c_void_p(base.data_ptr()), rows, cols, ld, float(p), int(seed),
                                             ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(yd.data_ptr()),
                                             _lib.cur_stream(x.device))
    _lib.check(rc, "ait_softmax_rows_fwd")
    shape = tuple(x.shape)
    view = lambda t: t.as_strided(shape, x.stride())
    return view(y), view(yd)


def softmax_rows_bwd(dyd, y, p, seed):
    cols, ld = y.shape[-1], y.stride(-2)
    rows = y.numel() // cols
    if dyd.stride() != y.stride():
        d2 = torch.zeros((rows, ld), dtype=torch.float32, device=y.device)[:, :cols].as_strided(tuple(y.shape), y.stride())
        d2.copy_(dyd)
        dyd = d2
    dx = torch.zeros((rows, ld), dtype=torch.float32, device=y.device)[:, :cols]
    with torch.cuda.device(y.device):
        rc = _lib.lib().ait_softmax_rows_bwd(ctypes.c_void_p(dyd.data_ptr()), ctypes.c_void_p(y.data_ptr()), rows, cols, ld,
                                             float(p), int(seed), ctypes.c_void_p(dx.data_ptr()), _lib.cur_stream(y.device))
    _lib.check(rc, "ait_softmax_rows_bwd")
    return dx.as_strided(tuple(y.shape), y.stride())


def sh_general_fwd(O, sk_w, sk_b):
    n, H, T, dv = O.shape
    u = torch.empty((n, T, dv), dtype=torch.float32, device=O.device)
    gate = torch.empty((n, H * dv), dtype=torch.float32, device=O.device)
    s = torch.empty((n, dv), dtype=torch.float32, device=O.device)
    with torch.cuda.device(O.device):
        rc = _lib.lib().ait_sh_general_fwd(_p(O), _p(sk_w), _p(sk_b), n, H, T, dv, _p(u), _p(gate), _p(s),
                                           _lib.cur_stream(O.device))
    _lib.check(rc, "ait_sh_general_fwd")
    return u, gate, s


def sh_general_bwd(du, O, gate, sk_w):
    n, H, T, dv = O.shape
    dO = torch.empty_like(O)
    dg = torch.empty((n, H * dv), dtype=torch.float32, device=O.device)
    ws = torch.empty((n * (H * dv + dv),), dtype=torch.float32, device=O.device)
    with torch.cuda.device(O.device):
        rc = _lib.lib().ait_sh_general_bwd(_p(du), _p(O), _p(gate), _p(sk_w), n, H, T, dv, _p(dO), _p(dg), _p(ws),
                                           _lib.cur_stream(O.device))
    _lib.check(rc, "ait_sh_general_bwd")
    return dO, dg


def gemm_relu_bwd(dy, w, act, out=None):
    """dh = (dy @ w) masked by act > 0   (dy [M,N], w [N,K] row-major, act [M,K])."""
    M, N = dy.shape
    K = w.shape[1]
    if out is None:
        out = torch.empty((M, K), dtype=torch.float32, device=dy.device)
    with torch.cuda.device(dy.device):
        rc = _gemm_fn()(
            0, 0, M, K, N, 1.0, _lib.dev_ptr(dy), dy.stride(0), _lib.dev_ptr(w), w.stride(0),
            _lib.dev_ptr(out), out.stride(0), None, _lib.dev_ptr(act), _lib.GEMM_MASK_POS, 1, 0, 0,
            _lib.launch_ctx(dy.device), _lib.cur_stream(dy.device))
    _lib.check(rc, "ait_gemm_f32(mask)")
    return out


def to_bf16(x, transpose=False, out=None):
    """ait_f32_to_bf16: x [rows, cols] f32 -> bf16 [rows, cols], or with transpose=True [cols, rows]"""
    assert x.dim() == 2 and x.dtype == torch.float32
    if x.stride(1) != 1:
        x = x.contiguous()
    rows, cols = x.shape
    if out is None:
        out = torch.empty((cols, rows) if transpose else (rows, cols), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().ait_f32_to_bf16(_lib.dev_ptr(x) if x.is_contiguous() else ctypes.c_void_p(x.data_ptr()), rows, cols,
                                        x.stride(0), ctypes.c_void_p(out.data_ptr()), out.stride(0), int(bool(transpose)),
                                        _lib.cur_stream(x.device))
    _lib.check(rc, "ait_f32_to_bf16")
    return out


def gemm_bf16s(a16, b16, bias=None, residual=None, gate16=None, relu=False, mask_pos=False, out32=None, out16=None,
               want32=True, want16=False):
    """ait_gemm_bf16s: (f32 and / or bf16) = a16 [M, K] @ b16 [N, K]^T (+bias)(+residual | gated)(relu), bf16 operands
    stored in memory.  Returns (out32 or None, out16 or None)."""
    assert a16.dtype == torch.bfloat16 and b16.dtype == torch.bfloat16 and a16.dim() == 2 and b16.dim() == 2
    assert a16.stride(1) == 1 and b16.stride(1) == 1
    M, K = a16.shape
    N = b16.shape[0]
    assert b16.shape[1] == K
    dev = a16.device
    if out32 is None and want32:
        out32 = torch.empty((M, N), dtype=torch.float32, device=dev)
    if out16 is None and want16:
        out16 = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    g = gate16 if gate16 is not None else residual
    flags = (_lib.GEMM_RELU if relu else 0) | (_lib.GEMM_MASK_POS if mask_pos else 0)
    vp = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        rc = _lib.lib().ait_gemm_bf16s(M, N, K, vp(a16), a16.stride(0), vp(b16), b16.stride(0), vp(out32),
                                       0 if out32 is None else out32.stride(0), vp(out16), 0 if out16 is None else out16.stride(0),
                                       _p(bias), _p(residual), vp(gate16), 0 if g is None else g.stride(0), flags,
                                       _lib.launch_ctx(dev), _lib.cur_stream(dev))
    _lib.check(rc, "ait_gemm_bf16s")
    return out32, out16


def colsum_bf16(x16, out=None):
    """ait_colsum_bf16: out [cols] f32 += column sums of a bf16 [rows, cols] matrix (row pitch from stride(0))"""
    assert x16.dtype == torch.bfloat16 and x16.dim() == 2 and x16.stride(1) == 1
    rows, cols = x16.shape
    if out is None:
        out = torch.zeros(cols, dtype=torch.float32, device=x16.device)
    with torch.cuda.device(x16.device):
        rc = _lib.lib().ait_colsum_bf16(ctypes.c_void_p(x16.data_ptr()), rows, cols, x16.stride(0), _p(out), _lib.cur_stream(x16.device))
    _lib.check(rc, "ait_colsum_bf16")
    return out


def gemm_bf16s_tn(dy16, x16, out=None, split_k=None, partials=True):
    """ait_gemm_bf16s_tn: out [Mo, No] f32 (+)= dy16^T @ x16 over the R token rows, bf16 operands [R, Mo] / [R, No]"""
    assert dy16.dtype == torch.bfloat16 and x16.dtype == torch.bfloat16 and dy16.shape[0] == x16.shape[0]
    R, Mo = dy16.shape
    No = x16.shape[1]
    if split_k is None:
        split_k = 1
        tiles = max(1, (Mo // 256) * (No // 128))
        for cand in (64, 32, 16, 8, 4, 2):
            if R % cand == 0 and (R // cand) % 32 == 0 and R // cand >= 512 and tiles * cand <= 2048:
                split_k = cand
                break
    if out is None:
        out = torch.zeros((Mo, No), dtype=torch.float32, device=dy16.device)
    ws = torch.empty(int(split_k) * Mo * No, dtype=torch.float32, device=dy16.device) if (partials and split_k > 1) else None
    with torch.cuda.device(dy16.device):
        rc = _lib.lib().ait_gemm_bf16s_tn(Mo, No, R, ctypes.c_void_p(dy16.data_ptr()), dy16.stride(0),
                                          ctypes.c_void_p(x16.data_ptr()), x16.stride(0), _p(out), out.stride(0), int(split_k),
                                          None if ws is None else ctypes.c_void_p(ws.data_ptr()), 0 if ws is None else ws.numel() * 4,
                                          _lib.launch_ctx(dy16.device), _lib.cur_stream(dy16.device))
    _lib.check(rc, "ait_gemm_bf16s_tn")
    return out


def p3_split(w, transpose=False):
    """The pre-split form of a weight (include/ait_hip.h "P3"): w [rows, cols] f32 -> bf16 [rows, cols/8, 3, 8] (planes h,
    m, l of every value, x = h + m + l exactly), or with transpose=True the same of w.t(): [cols, rows/8, 3, 8]."""
    assert w.dim() == 2 and w.dtype == torch.float32
    if w.stride(1) != 1 or w.stride(0) % 4:
        w = w.contiguous()
    rows, cols = w.shape
    shape = (cols, rows // 8, 3, 8) if transpose else (rows, cols // 8, 3, 8)
    out = torch.empty(shape, dtype=torch.bfloat16, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.lib().ait_p3_split(_lib.dev_ptr(w), rows, cols, w.stride(0), int(bool(transpose)),
                                     ctypes.c_void_p(out.data_ptr()), _lib.cur_stream(w.device))
    _lib.check(rc, "ait_p3_split")
    return out


def gemm_p3(a, w_p3, bias=None, residual=None, relu=False, mask_pos=False, colsum=None, alpha=1.0, out=None):
    """out = alpha * a @ W^T (+bias)(+residual | masked by residual > 0)(relu), W given pre-split: w_p3 = p3_split(W)
    ([N, K/8, 3, 8]; for the input gradient dy @ W pass p3_split(W, transpose=True)).  colsum: float[N] into which
    the column sums of the result are added (ait_gemm_f32_p3)."""
    M, K = a.shape
    N = w_p3.shape[0]
    assert w_p3.dtype == torch.bfloat16 and w_p3.shape[1] * 8 == K and w_p3.is_contiguous()
    if a.stride(1) != 1 or a.stride(0) % 4:
        a = a.contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    flags = (_lib.GEMM_RELU if relu else 0) | (_lib.GEMM_MASK_POS if mask_pos else 0) | (_lib.GEMM_COLSUM if colsum is not None else 0)
    with torch.cuda.device(a.device):
        rc = _lib.lib().ait_gemm_f32_p3(M, N, K, float(alpha), _lib.dev_ptr(a), a.stride(0),
                                        ctypes.c_void_p(w_p3.data_ptr()), K, ctypes.c_void_p(out.data_ptr()),
                                        out.stride(0), _p(colsum if colsum is not None else bias), _p(residual), flags, 0, 0,
                                        _lib.launch_ctx(a.device), _lib.cur_stream(a.device))
    _lib.check(rc, "ait_gemm_f32_p3")
    return out


D_MODEL = 512


def ln_fwd(a, pos, residual, gamma, beta, rows, seq_len, src_rows, rep, eps, p, seed,
           save_stats=True):
    y = torch.empty((rows, D_MODEL), dtype=torch.float32, device=a.device)
    mean = torch.empty((rows,), dtype=torch.float32, device=a.device) if save_stats else None
    rstd = torch.empty((rows,), dtype=torch.float32, device=a.device) if save_stats else None
    with torch.cuda.device(a.device):
        rc = _lib.lib().ait_ln_fwd(_p(a), _p(pos), _p(residual), _p(gamma), _p(beta), rows, D_MODEL,
                                   seq_len, src_rows, rep, float(eps), float(p), int(seed), _p(y),
                                   _p(mean), _p(rstd), _lib.cur_stream(a.device))
    _lib.check(rc, "ait_ln_fwd")
    return y, mean, rstd


def ln_bwd(dy, a, pos, residual, gamma, mean, rstd, rows, seq_len, src_rows, rep, p, seed,
           need_da=True, need_dres=True, dy_rows=None, colsum=False):
    """colsum=True also returns the column sums of da (the bias gradient of the linear layer that
    produced `a`) as a fifth value; dy_rows: dy holds only that many rows per sequence."""
    dev = dy.device
    da = None
    if need_da:
        n_src = a.shape[0] if rep == 1 else rows
        da = torch.empty((n_src, D_MODEL), dtype=torch.float32, device=dev)
    dres = torch.empty((rows, D_MODEL), dtype=torch.float32, device=dev) if need_dres else None
    dgb = torch.zeros((3, D_MODEL), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().ait_ln_bwd(_p(dy), _p(a), _p(pos), _p(residual), _p(gamma), _p(mean),
                                   _p(rstd), rows, D_MODEL, seq_len, src_rows, rep,
                                   seq_len if dy_rows is None else int(dy_rows), float(p),
                                   int(seed), _p(da), _p(dres), _p(dgb[0]), _p(dgb[1]),
                                   _p(dgb[2]) if colsum else None, _lib.cur_stream(dev))
    _lib.check(rc, "ait_ln_bwd")
    if colsum:
        return da, dres, dgb[0], dgb[1], dgb[2]
    return da, dres, dgb[0], dgb[1]


def colsum(x):
    """sum over the rows of a 2-D tensor (bias gradients) -> [cols]"""
    assert x.dim() == 2 and x.stride(1) == 1
    out = torch.zeros((x.shape[1],), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().ait_colsum_f32(ctypes.c_void_p(x.data_ptr()), x.shape[0], x.shape[1], x.stride(0),
                                       _p(out), _lib.cur_stream(x.device))
    _lib.check(rc, "ait_colsum_f32")
    return out


def rep_sum(x, groups, rep, E):
    """x [groups*rep*E] -> out [groups*E]: sum over the rep copies of each E-float block"""
    out = torch.zeros((groups * E,), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().ait_rep_sum_f32(_p(x), int(groups), int(rep), int(E), _p(out), _lib.cur_stream(x.device))
    _lib.check(rc, "ait_rep_sum_f32")
    return out


def dropout_seed(base, site):
    return int(_lib.lib().ait_dropout_seed(int(base), int(site)))


def dropout_mask(site_seed, first_index, count, p, device):
    """ait_dropout_mask: the factors (1 / (1 - p) or 0) the kernels apply to elements first_index .. + count of a site"""
    out = torch.empty(int(count), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().ait_dropout_mask(int(site_seed), int(first_index), int(count), float(p), _lib.dev_ptr(out),
                                               _lib.cur_stream(device)), "ait_dropout_mask")
    return out


def heads_fwd(props, query, w_bbox, b_bbox, w1, b1, w2, b2):
    """ait_heads_fwd: bbox_pred [R, n_bbox], hidden [R, 8], score [R, 2] of the detector's two heads"""
    R, F = props.shape
    bs = query.shape[0]
    nb = w_bbox.shape[0]
    dev = props.device
    bbox = torch.empty((R, nb), dtype=torch.float32, device=dev)
    hidden = torch.empty((R, 8), dtype=torch.float32, device=dev)
    score = torch.empty((R, 2), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().ait_heads_fwd(_p(props), _p(query), R, bs, F, _p(w_bbox), _p(b_bbox), nb, _p(w1), _p(b1), _p(w2),
                                      _p(b2), _p(bbox), _p(hidden), _p(score), _lib.cur_stream(dev))
    _lib.check(rc, "ait_heads_fwd")
    return bbox, hidden, score


def heads_bwd(d_bbox, d_score, props, query, w_bbox, w1, w2, hidden, need_props=True, need_query=True):
    """ait_heads_bwd: (d_props, d_query, one zero-filled flat buffer holding d w_bbox | d b_bbox | d w1 | d b1 | d w2 | d b2)"""
    R, F = props.shape
    bs = query.shape[0]
    nb = w_bbox.shape[0]
    dev = props.device
    L = _lib.lib()
    sizes = [nb * F, nb, 8 * 2 * F, 8, 16, 2]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    parts, o = [], 0
    for n in sizes:
        parts.append(flat[o:o + n])
        o += n
    ws = torch.empty(int(L.ait_heads_bwd_workspace_bytes(R, bs)), dtype=torch.uint8, device=dev)
    d_props = torch.empty_like(props) if need_props else None
    d_query = torch.empty_like(query) if need_query else None
    with torch.cuda.device(dev):
        rc = L.ait_heads_bwd(_p(d_bbox), _p(d_score), _p(props), _p(query), R, bs, F, _p(w_bbox), nb, _p(w1), _p(w2), _p(hidden),
                             ctypes.c_void_p(ws.data_ptr()), ws.numel(), _p(d_props), _p(d_query), *[_p(t) for t in parts],
                             _lib.cur_stream(dev))
    _lib.check(rc, "ait_heads_bwd")
    return d_props, d_query, parts


def sh_fwd(O, sk_w, sk_b):
    n, H, T, dv = O.shape
    u = torch.empty((n, T, dv), dtype=torch.float32, device=O.device)
    gate = torch.empty((n, H * dv), dtype=torch.float32, device=O.device)
    s = torch.empty((n, dv), dtype=torch.float32, device=O.device)
    with torch.cuda.device(O.device):
        rc = _lib.lib().ait_sh_fwd(_p(O), _p(sk_w), _p(sk_b), n, H, T, dv, _p(u), _p(gate), _p(s),
                                   _lib.cur_stream(O.device))
    _lib.check(rc, "ait_sh_fwd")
    return u, gate, s


def sh_bwd(du, O, gate, sk_w):
    n, H, T, dv = O.shape
    dO = torch.empty_like(O)
    dg = torch.empty((n, H * dv), dtype=torch.float32, device=O.device)
    with torch.cuda.device(O.device):
        rc = _lib.lib().ait_sh_bwd(_p(du), _p(O), _p(gate), _p(sk_w), n, H, T, dv, _p(dO), _p(dg),
                                   _lib.cur_stream(O.device))
    _lib.check(rc, "ait_sh_bwd")
    return dO, dg


def _col(t, off):
    """device pointer of column `off` of a 2-D row-major tensor"""
    _lib.dev_ptr(t)
    return ctypes.c_void_p(t.data_ptr() + 4 * off)


def attn_fwd(q, qoff, k, koff, v, voff, n_seq, H, T, d, mask_mode, n_valid, scale, p, seed,
             save_p=True, kv_rows=None):
    kv_rows = T if kv_rows is None else kv_rows
    dev = q.device
    O = torch.empty((n_seq, H, T, d), dtype=torch.float32, device=dev)
    P = torch.empty((n_seq, H, T, T), dtype=torch.float32, device=dev) if save_p else None
    with torch.cuda.device(dev):
        rc = _lib.lib().ait_attn_fwd(_col(q, qoff), q.stride(0), _col(k, koff), k.stride(0),
                                     _col(v, voff), v.stride(0), n_seq, H, T, d, int(kv_rows), mask_mode,
                                     n_valid, float(scale), float(p), int(seed), _p(P), _p(O),
                                     _lib.cur_stream(dev))
    _lib.check(rc, "ait_attn_fwd")
    return O, P


def mha_core_fwd(q, qoff, k, koff, v, voff, n_seq, mask_mode, n_valid, p_attn, seed_attn, sk_w, sk_b, fc_w, residual,
                 ln_g, ln_b, eps, p_fc, seed_fc, kv_rows=64, out_rows=64, save=True, q_rep=1):
    """ait_mha_core_fwd: attention tiles + selective heads + fc + dropout + residual + LayerNorm of one
    MultiHeadAttention block in one launch.  Returns y and, with save=True, the dict of tensors the backward reads."""
    dev = q.device
    e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    y = e(n_seq * out_rows, D_MODEL)
    sv = dict(P=e(n_seq, 8, 64, 64), O=e(n_seq, 8, 64, 64), u=e(n_seq * 64, 64), gate=e(n_seq, D_MODEL), s=e(n_seq, 64),
              f=e(n_seq * 64, D_MODEL), mean=e(n_seq * 64), rstd=e(n_seq * 64)) if save else {}
    g = lambda name: _p(sv.get(name))
    with torch.cuda.device(dev):
        rc = _lib.lib().ait_mha_core_fwd(_col(q, qoff), q.stride(0), _col(k, koff), k.stride(0), _col(v, voff), v.stride(0),
                                         n_seq, int(kv_rows), mask_mode, n_valid, 0.125, float(p_attn), int(seed_attn),
                                         _p(sk_w), _p(sk_b), _p(fc_w), _p(residual), _p(ln_g), _p(ln_b), float(eps),
                                         float(p_fc), int(seed_fc), int(out_rows), int(q_rep), g("P"), g("O"), g("u"), g("gate"), g("s"),
                                         g("f"), _p(y), g("mean"), g("rstd"), _lib.cur_stream(dev))
    _lib.check(rc, "ait_mha_core_fwd")
    return y, sv


def mha_core_bwd(df, fc_w, O, gate, sk_w, q, qoff, k, koff, v, voff, P, n_seq, p_attn, seed_attn, dq, dqoff, dk, dkoff,
                 dv, dvoff, kv_rows=64):
    """ait_mha_core_bwd: fc's input gradient + selective heads + attention tiles backwards, one launch.  Writes dq / dk / dv
    (columns dqoff / dkoff / dvoff .. + 512 of the given tensors) and returns dg [n_seq, 512]."""
    dev = q.device
    dg = torch.empty(n_seq, D_MODEL, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().ait_mha_core_bwd(_p(df), _p(fc_w), _p(O), _p(gate), _p(sk_w), _col(q, qoff), q.stride(0), _col(k, koff),
                                         k.stride(0), _col(v, voff), v.stride(0), _p(P), n_seq, int(kv_rows), 0.125,
                                         float(p_attn), int(seed_attn), _col(dq, dqoff), dq.stride(0), _col(dk, dkoff),
                                         dk.stride(0), _col(dv, dvoff), dv.stride(0), _p(dg), _lib.cur_stream(dev))
    _lib.check(rc, "ait_mha_core_bwd")
    return dg


def attn_bwd(q, qoff, k, koff, v, voff, P, dO, n_seq, H, T, d, scale, p, seed, dq, dqoff, dk,
             dkoff, dv, dvoff, kv_rows=None):
    dev = q.device
    kv_rows = T if kv_rows is None else kv_rows
    with torch.cuda.device(dev):
        rc = _lib.lib().ait_attn_bwd(_col(q, qoff), q.stride(0), _col(k, koff), k.stride(0),
                                     _col(v, voff), v.stride(0), _p(P), _p(dO), n_seq, H, T, d,
                                     int(kv_rows), float(scale), float(p), int(seed), _col(dq, dqoff),
                                     dq.stride(0), _col(dk, dkoff), dk.stride(0),
                                     _col(dv, dvoff), dv.stride(0), _lib.cur_stream(dev))
    _lib.check(rc, "ait_attn_bwd")


def _bn_dims(x):
    """(n, C, HW) as the kernel sees the memory: NCHW planes, or -- for a channels-last tensor --
    n*H*W rows of C channels (HW = 1)."""
    n, C = x.shape[0], x.shape[1]
    if x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last):
        return x.numel() // max(1, C), C, 1
    return n, C, x.numel() // max(1, n * C)


def _pd(t):
    return None if t is None else _lib.dev_ptr(t, torch.float32, dense_any_format=True)


def _same_format(x, *others):
    for t in others:
        if t is not None and (t.shape != x.shape or t.stride() != x.stride()):
            raise _lib.AitHipError("bn_act operands must share shape and memory format")


def bn_act_fwd(x, scale, shift, residual, relu):
    n, C, HW = _bn_dims(x)
    y = torch.empty_like(x)
    _same_format(x, residual, y)
    with torch.cuda.device(x.device):
        rc = _lib.lib().ait_bn_act_fwd(_pd(x), _p(scale), _p(shift), _pd(residual), int(relu), n, C, HW,
                                       _pd(y), _lib.cur_stream(x.device))
    _lib.check(rc, "ait_bn_act_fwd")
    return y


def bn_act_bwd(dy, y, scale, relu, need_dres, dy2=None):
    """dx, dres of ait_bn_act_fwd for the gradient dy (+ dy2: a second addend, summed in the same pass)"""
    n, C, HW = _bn_dims(dy)
    dx = torch.empty_like(dy)
    dres = torch.empty_like(dy) if need_dres else None
    _same_format(dy, y, dx, dres, dy2)
    with torch.cuda.device(dy.device):
        rc = _lib.lib().ait_bn_act_bwd(_pd(dy), _pd(dy2), _pd(y), _p(scale), int(relu), n, C, HW, _pd(dx), _pd(dres),
                                       _lib.cur_stream(dy.device))
    _lib.check(rc, "ait_bn_act_bwd")
    return dx, dres


def _cl_rows(x):
    """(rows, C) of a 4-d bf16 tensor in channels-last memory"""
    if not (x.dim() == 4 and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.AitHipError("bf16 bn_act: a 4-d bfloat16 tensor in channels-last memory expected")
    return x.numel() // x.shape[1], x.shape[1]


def bn_act_fwd_bf16(x, scale, shift, residual, relu):
    """ait_bn_act_fwd_bf16: relu(x * scale[c] + shift[c] + residual) over bf16 channels-last tensors"""
    rows, C = _cl_rows(x)
    if residual is not None:
        _cl_rows(residual)
    y = torch.empty_like(x)
    _same_format(x, residual, y)
    vp = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    with torch.cuda.device(x.device):
        rc = _lib.lib().ait_bn_act_fwd_bf16(vp(x), _p(scale), _p(shift), vp(residual), int(relu), rows, C, vp(y),
                                            _lib.cur_stream(x.device))
    _lib.check(rc, "ait_bn_act_fwd_bf16")
    return y


def bn_act_bwd_bf16(dy, y, scale, relu, need_dres, dy2=None):
    rows, C = _cl_rows(dy)
    if dy2 is not None:
        _cl_rows(dy2)
    dx = torch.empty_like(dy)
    dres = torch.empty_like(dy) if need_dres else None
    _same_format(dy, y, dx, dres, dy2)
    vp = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    with torch.cuda.device(dy.device):
        rc = _lib.lib().ait_bn_act_bwd_bf16(vp(dy), vp(dy2), vp(y), _p(scale), int(relu), rows, C, vp(dx), vp(dres),
                                            _lib.cur_stream(dy.device))
    _lib.check(rc, "ait_bn_act_bwd_bf16")
    return dx, dres


def sk_sqsum_fwd(a, b):
    y = torch.empty_like(a)
    _same_format(a, b, y)
    with torch.cuda.device(a.device):
        rc = _lib.lib().ait_sk_sqsum_fwd(_pd(a), _pd(b), a.numel(), _pd(y), _lib.cur_stream(a.device))
    _lib.check(rc, "ait_sk_sqsum_fwd")
    return y


def sk_sqsum_bwd(dy, a, b):
    da, db = torch.empty_like(a), torch.empty_like(b)
    _same_format(a, b, dy, da, db)
    with torch.cuda.device(a.device):
        rc = _lib.lib().ait_sk_sqsum_bwd(_pd(dy), _pd(a), _pd(b), a.numel(), _pd(da), _pd(db),
                                         _lib.cur_stream(a.device))
    _lib.check(rc, "ait_sk_sqsum_bwd")
    return da, db


# ------------------------------------------------------------------------------------------
# convolutions over channels-last maps as implicit GEMMs (ait_conv_*_f32)
# ------------------------------------------------------------------------------------------
_ZEROS = {}


def _zeros(device, n=8192):
    """the row of zeros a window position outside the map reads (one per device, never written)"""
    key = (str(device), n)
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros(n, dtype=torch.float32, device=device)
    return _ZEROS[key]


def conv_geom(n, in_hw, out_hw, k, stride, pad, groups=1):
    g = _lib.ConvGeom()
    g.n, g.in_h, g.in_w, g.out_h, g.out_w = int(n), int(in_hw[0]), int(in_hw[1]), int(out_hw[0]), int(out_hw[1])
    g.kh, g.kw, g.stride, g.pad, g.groups = int(k[0]), int(k[1]), int(stride), int(pad), int(groups)
    return g


def conv_supported(in_hw, out_hw, stride, cin, cout):
    """what ait_conv_*_f32 take (forward, data and weight gradient all together): maps of any size (csrc/conv_f32.hip
    decomposes a row index by shifts for power-of-two maps and by exactly corrected f32 quotients otherwise), stride 1
    or 2, channel counts the 16-value slabs and the 128-channel groups divide.  (Maps whose sides are not powers of two:
    ungrouped only and fewer than 2^24 output rows -- the library returns AIT_EUNSUPPORTED otherwise.)"""
    return (min(in_hw) > 0 and min(out_hw) > 0 and stride in (1, 2)
            and cin % 128 == 0 and cout % 16 == 0 and max(cin, cout) + 144 <= 8192)


def conv_fwd(x, w, geom, bias=None, residual=None, relu=False):
    """x [rows_in, cin] token rows of a channels-last map, w [cout, kh, kw, cin / groups] -> y [rows_out, cout]"""
    cout, cin = w.shape[0], w.shape[3] * max(1, geom.groups)
    rows = geom.n * geom.out_h * geom.out_w
    y = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    z = _zeros(x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().ait_conv_fwd_f32(_lib.dev_ptr(x), x.stride(0), _lib.dev_ptr(w), ctypes.byref(geom), cin, cout,
                                         _p(bias), _p(residual), _lib.GEMM_RELU if relu else 0, _p(y), cout, _p(z),
                                         z.numel(), _lib.launch_ctx(x.device), _lib.cur_stream(x.device))
    _lib.check(rc, "ait_conv_fwd_f32")
    return y


def conv_bwd_data(dy, w, geom, residual=None, mask_pos=False, out=None):
    """dx = conv_transpose(dy, w) (+ residual, or gated by residual > 0 with mask_pos).  out: write into this
    [rows_in, cin] tensor; with residual is out the product is ACCUMULATED into it (the second branch of a block
    whose branches share their input: SKBlock)."""
    cout, cin = w.shape[0], w.shape[3] * max(1, geom.groups)
    rows = geom.n * geom.in_h * geom.in_w
    dx = torch.empty((rows, cin), dtype=torch.float32, device=dy.device) if out is None else out
    if tuple(dx.shape) != (rows, cin) or not dx.is_contiguous():
        raise _lib.AitHipError("conv_bwd_data: `out` must be a contiguous [%d, %d] tensor" % (rows, cin))
    z = _zeros(dy.device)
    with torch.cuda.device(dy.device):
        rc = _lib.lib().ait_conv_bwd_data_f32(_lib.dev_ptr(dy), dy.stride(0), _lib.dev_ptr(w), ctypes.byref(geom), cin, cout,
                                              _p(residual), _lib.GEMM_MASK_POS if mask_pos else 0, _p(dx), cin, _p(z),
                                              z.numel(), _lib.launch_ctx(dy.device), _lib.cur_stream(dy.device))
    _lib.check(rc, "ait_conv_bwd_data_f32")
    return dx


def conv_bwd_weight(dy, x, geom, kh, kw, split_k=8):
    cout, cin = dy.shape[1], x.shape[1]
    dw = torch.zeros((cout, kh, kw, cin // max(1, geom.groups)), dtype=torch.float32, device=dy.device)
    z = _zeros(dy.device)
    with torch.cuda.device(dy.device):
        rc = _lib.lib().ait_conv_bwd_weight_f32(_lib.dev_ptr(dy), dy.stride(0), _lib.dev_ptr(x), x.stride(0),
                                                ctypes.byref(geom), cin, cout, _p(dw), int(split_k), _p(z), z.numel(),
                                                _lib.launch_ctx(dy.device), _lib.cur_stream(dy.device))
    _lib.check(rc, "ait_conv_bwd_weight_f32")
    return dw


# ------------------------------------------------------------------------------------------
# ... and over bf16 channels-last maps (ait_conv_*_bf16s: the bf16 configuration's layer4)
# ------------------------------------------------------------------------------------------
def conv_weight_to_bf16(w, row_scale=None, dgrad=False):
    """w f32 [cout, kh, kw, cin] (x row_scale[cout]) -> bf16 [cout, kh*kw*cin]; dgrad=True: the data gradient's operand
    [cin, kh*kw*cout] with the window mirrored (ait_conv_weight_to_bf16)"""
    cout, kh, kw, cin = w.shape
    taps = kh * kw
    out = torch.empty((cin, taps * cout) if dgrad else (cout, taps * cin), dtype=torch.bfloat16, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.lib().ait_conv_weight_to_bf16(_lib.dev_ptr(w), _p(row_scale), cout, taps, cin,
                                                None if dgrad else ctypes.c_void_p(out.data_ptr()),
                                                ctypes.c_void_p(out.data_ptr()) if dgrad else None, _lib.cur_stream(w.device))
    _lib.check(rc, "ait_conv_weight_to_bf16")
    return out


def conv_fwd_bf16s(x16, w16, geom, cin, cout, bias=None, res16=None, gate16=None, relu=False, out_f32=False):
    """x16 bf16 [rows, cin] rows of a channels-last map, w16 bf16 [cout, taps*cin] -> y [rows, cout] (bf16, or f32 with
    out_f32); + bias, + res16, kept where gate16 > 0, ReLU.  The data gradient: the same call with conv_weight_to_bf16(...,
    dgrad=True) and cin / cout swapped."""
    rows = geom.n * geom.out_h * geom.out_w
    y = torch.empty((rows, cout), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x16.device)
    z = _zeros(x16.device)
    ldr = (res16 if res16 is not None else gate16).stride(0) if (res16 is not None or gate16 is not None) else 0
    if res16 is not None and gate16 is not None and res16.stride(0) != gate16.stride(0):
        raise _lib.AitHipError("conv_fwd_bf16s: res16 and gate16 share one pitch")
    vp = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    flags = (_lib.GEMM_RELU if relu else 0) | (_lib.GEMM_MASK_POS if gate16 is not None else 0)
    with torch.cuda.device(x16.device):
        rc = _lib.lib().ait_conv_fwd_bf16s(vp(x16), x16.stride(0), vp(w16), ctypes.byref(geom), cin, cout, _p(bias), vp(res16),
                                           vp(gate16), ldr, flags, _p(y) if out_f32 else None, cout,
                                           None if out_f32 else vp(y), cout, _p(z), z.numel() * 4,
                                           _lib.launch_ctx(x16.device), _lib.cur_stream(x16.device))
    _lib.check(rc, "ait_conv_fwd_bf16s")
    return y


def conv_bwd_weight_bf16s(dy16, x16, geom, kh, kw, split_k=8, partials=False):
    cout, cin = dy16.shape[1], x16.shape[1]
    dw = torch.zeros((cout, kh, kw, cin), dtype=torch.float32, device=dy16.device)
    z = _zeros(dy16.device)
    ws = torch.empty(int(split_k) * dw.numel(), dtype=torch.float32, device=dy16.device) if (partials and split_k > 1) else None
    with torch.cuda.device(dy16.device):
        rc = _lib.lib().ait_conv_bwd_weight_bf16s(ctypes.c_void_p(dy16.data_ptr()), dy16.stride(0), ctypes.c_void_p(x16.data_ptr()),
                                                  x16.stride(0), ctypes.byref(geom), cin, cout, _p(dw), int(split_k), _p(z),
                                                  z.numel() * 4, _p(ws), 0 if ws is None else ws.numel() * 4,
                                                  _lib.launch_ctx(dy16.device), _lib.cur_stream(dy16.device))
    _lib.check(rc, "ait_conv_bwd_weight_bf16s")
    return dw
