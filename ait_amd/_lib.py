"""ctypes binding of libait_hip.so (include/ait_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a tensor is not on a
GPU, the call raises.  PyTorch is used only for device memory and streams.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libait_hip.so")

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_ll, _ull = ctypes.c_longlong, ctypes.c_ulonglong

# name -> (restype, argtypes); mirrors include/ait_hip.h one to one
SIGNATURES = {
    "ait_abi_version": (_i, []),
    "ait_lab_build": (_i, []),
    "ait_strerror": (ctypes.c_char_p, [_i]),
    "ait_gemm_workspace_bytes": (_sz, []),
    "ait_gemm_workspace_init": (_i, [_vp, _sz, _vp]),
    "ait_probe_create": (_vp, [_i]),
    "ait_probe_destroy": (None, [_vp]),
    "ait_probe_reset": (_i, [_vp]),
    "ait_probe_count": (_i, [_vp]),
    "ait_probe_capacity": (_i, [_vp]),
    "ait_probe_get": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "ait_roi_align_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "ait_roi_align_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "ait_roi_align_nhwc_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ait_roi_align_nhwc_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp, _vp, _vp]),
    "ait_roi_align_nhwc_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp, _vp, _vp]),
    "ait_nms_workspace_bytes": (_sz, [_i]),
    "ait_nms": (_i, [_vp, _vp, _i, _f, _i, _vp, _sz, _vp, _vp, _vp]),
    "ait_nms_batched_workspace_bytes": (_sz, [_i, _i]),
    "ait_nms_batched": (_i, [_vp, _i, _i, _f, _i, _vp, _sz, _vp, _ll, _vp, _vp]),
    "ait_rpn_decode": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ait_proposals_assemble": (_i, [_vp, _i, _vp, _ll, _i, _vp, _i, _i, _vp, _vp]),
    "ait_anchor_classify": (_i, [_vp, _i, _vp, _i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ait_anchor_targets": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp,
                                _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "ait_roi_classify_workspace_bytes": (_sz, [_i, _i, _i]),
    "ait_roi_classify": (_i, [_vp, _i, _i, _vp, _i, _i, _f, _f, _f, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ait_roi_sample_gather": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i,
                                   ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                   ctypes.POINTER(ctypes.c_float), _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ait_bn_act_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _ll, _i, _i, _vp, _vp]),
    "ait_bn_act_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _ll, _i, _i, _vp, _vp, _vp]),
    "ait_bn_act_fwd_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _ll, _i, _vp, _vp]),
    "ait_bn_act_bwd_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp]),
    "ait_sk_sqsum_fwd": (_i, [_vp, _vp, _ll, _vp, _vp]),
    "ait_sk_sqsum_bwd": (_i, [_vp, _vp, _vp, _ll, _vp, _vp, _vp]),
    "ait_gemm_f32": (_i, [_i, _i, _i, _i, _i, _f, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i,
                          _i, ctypes.c_longlong, _vp, _vp]),
    "ait_gemm_bf16s": (_i, [_i, _i, _i, _vp, _ll, _vp, _ll, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _ll, _i, _vp, _vp]),
    "ait_gemm_bf16s_tn": (_i, [_i, _i, _i, _vp, _ll, _vp, _ll, _vp, _ll, _i, _vp, _sz, _vp, _vp]),
    "ait_colsum_bf16": (_i, [_vp, _ll, _i, _ll, _vp, _vp]),
    "ait_f32_to_bf16": (_i, [_vp, _ll, _i, _ll, _vp, _ll, _i, _vp]),
    "ait_conv_fwd_bf16s": (_i, [_vp, _ll, _vp, _vp, _i, _i, _vp, _vp, _vp, _ll, _i, _vp, _ll, _vp, _ll, _vp, _sz, _vp, _vp]),
    "ait_conv_bwd_weight_bf16s": (_i, [_vp, _ll, _vp, _ll, _vp, _i, _i, _vp, _i, _vp, _sz, _vp, _sz, _vp, _vp]),
    "ait_conv_weight_to_bf16": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ait_p3_bytes": (_sz, [_ll, _ll]),
    "ait_p3_split": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "ait_gemm_f32_p3": (_i, [_i, _i, _i, _f, _vp, _i, _vp, _ll, _vp, _i, _vp, _vp, _i, _i, _ll, _vp, _vp]),
    "ait_gemm_f32_batched": (_i, [_i, _i, _i, _i, _i, _f, _vp, _i, _ll, _ll, _vp, _i, _ll, _ll, _vp, _i, _ll, _ll, _i, _i,
                                  _i, _i, _vp, _vp]),
    "ait_softmax_rows_fwd": (_i, [_vp, _ll, _i, _ll, _f, _ull, _vp, _vp, _vp]),
    "ait_softmax_rows_bwd": (_i, [_vp, _vp, _ll, _i, _ll, _f, _ull, _vp, _vp]),
    "ait_sh_general_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ait_sh_general_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ait_conv_fwd_f32": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _sz, _vp, _vp]),
    "ait_conv_bwd_data_f32": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _sz, _vp, _vp]),
    "ait_conv_bwd_weight_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _sz, _vp, _vp]),
    "ait_gemm_bf16": (_i, [_i, _i, _i, _i, _i, _f, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i,
                           _i, ctypes.c_longlong, _vp, _vp]),
    "ait_mha_block_workspace_bytes": (_sz, [_i, _i]),
    "ait_mha_block_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp]),
    "ait_ffn_workspace_bytes": (_sz, [_ll]),
    "ait_ffn_fwd": (_i, [_vp, _ll, _vp, _vp, _sz, _vp, _vp, _vp]),
    "ait_transformer_workspace_bytes": (_sz, [_i, _i, _i]),
    "ait_transformer_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp]),
    "ait_ln_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _f, _f, _ull, _vp, _vp, _vp, _vp]),
    "ait_ln_fwd_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _f, _f, _ull, _vp, _vp, _vp, _vp]),
    "ait_ln_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _f, _ull, _vp, _vp,
                        _vp, _vp, _vp, _vp]),
    "ait_colsum_f32": (_i, [_vp, _ll, _i, _ll, _vp, _vp]),
    "ait_rep_sum_f32": (_i, [_vp, _i, _i, _ll, _vp, _vp]),
    "ait_dropout_seed": (_ull, [_ull, _i]),
    "ait_dropout_mask": (_i, [_ull, _ull, _ll, _f, _vp, _vp]),
    "ait_mha_block_saved_bytes": (_sz, [_i, _i]),
    "ait_mha_block_fwd_train": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _f, _f, _ull, _vp, _sz, _vp, _vp, _vp]),
    "ait_mha_block_bwd_workspace_bytes": (_sz, [_i, _i]),
    "ait_mha_block_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _f, _f, _ull, _vp, _sz, _vp, _sz, _vp, _vp,
                               _vp, _vp, _vp]),
    "ait_ffn_saved_bytes": (_sz, [_ll]),
    "ait_ffn_fwd_train": (_i, [_vp, _ll, _vp, _f, _ull, _vp, _sz, _vp, _vp, _vp]),
    "ait_ffn_bwd_workspace_bytes": (_sz, [_ll]),
    "ait_ffn_bwd": (_i, [_vp, _vp, _ll, _vp, _f, _ull, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp]),
    "ait_transformer_saved_bytes": (_sz, [_i, _i, _i]),
    "ait_transformer_io_bf16_ok": (_i, [_i, _i, _i]),
    "ait_transformer_fwd_train": (_i, [_vp, _vp, _i, _i, _i, _vp, _f, _f, _ull, _vp, _sz, _vp, _vp, _vp, _vp]),
    "ait_transformer_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "ait_transformer_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _f, _f, _ull, _vp, _sz, ctypes.c_uint, _vp, _sz, _vp, _vp,
                                 _vp, _vp, _vp]),
    "ait_transformer_bwd_part": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _vp, _f, _f, _ull, _vp, _sz, ctypes.c_uint, _vp, _sz,
                                      _vp, _vp, _vp, _vp, _vp]),
    "ait_tail_saved_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ait_tail_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "ait_tail_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ait_tail_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, ctypes.c_uint, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "ait_heads_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ait_heads_bwd_workspace_bytes": (_sz, [_i, _i]),
    "ait_heads_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _sz] + [_vp] * 8 + [_vp]),
    "ait_sh_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ait_sh_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ait_mha_core_fwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _f, _ull, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f,
                               _ull, _i, _i] + [_vp] * 10),
    "ait_mha_core_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _f, _f, _ull, _vp, _i, _vp, _i,
                               _vp, _i, _vp, _vp]),
    "ait_attn_fwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _ull, _vp, _vp,
                          _vp]),
    "ait_attn_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _ull, _vp, _i,
                          _vp, _i, _vp, _i, _vp]),
}

GEMM_RELU, GEMM_ACCUMULATE, GEMM_ATOMIC, GEMM_BIAS_ROW, GEMM_MASK_POS, GEMM_COLSUM = 1, 2, 4, 8, 16, 32

_lib = None


class AitHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AitHipError(
                "libait_hip.so is missing (%s): run `python -m ait_amd.build` -- there is no "
                "CPU fallback for the hot path" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class MhaWeights(ctypes.Structure):
    """ait_mha_weights of include/ait_hip.h."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("w_qkv", "sk_w", "sk_b", "fc_w", "ln_g", "ln_b")]


class FfnWeights(ctypes.Structure):
    """ait_ffn_weights."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("w1", "b1", "w2", "b2", "ln_g", "ln_b")]


class ConvGeom(ctypes.Structure):
    """ait_conv_geom."""
    _fields_ = [(n, ctypes.c_int) for n in ("n", "in_h", "in_w", "out_h", "out_w", "kh", "kw", "stride", "pad", "groups")]


class SkWeights(ctypes.Structure):
    """ait_sk_weights (and ait_sk_grads: the same members)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("w1", "b1", "w3", "b3")]


class BottleneckWeights(ctypes.Structure):
    """ait_bottleneck_weights."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("conv1", "conv2", "conv3", "down", "bn1_scale", "bn1_shift", "bn2_scale",
                                               "bn2_shift", "bn3_scale", "bn3_shift", "bnd_scale", "bnd_shift")]


class BottleneckGrads(ctypes.Structure):
    """ait_bottleneck_grads."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("conv1", "conv2", "conv3", "down")]


class TailWeights(ctypes.Structure):
    """ait_tail_weights."""
    _fields_ = [("sk_props", SkWeights), ("sk_query", SkWeights), ("block", BottleneckWeights * 4)]


class TailGrads(ctypes.Structure):
    """ait_tail_grads."""
    _fields_ = [("sk_props", SkWeights), ("sk_query", SkWeights), ("block", BottleneckGrads * 4)]


class TransformerWeights(ctypes.Structure):
    """ait_transformer_weights."""
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "enc_emb_w", "enc_emb_b", "dec_emb_w", "dec_emb_b", "dec_trans_w", "dec_trans_b",
        "enc_ln_g", "enc_ln_b", "dec_ln_g", "dec_ln_b", "pos_table")] + \
        [("enc_slf", MhaWeights), ("dec_slf", MhaWeights), ("dec_enc", MhaWeights),
         ("enc_ffn", FfnWeights), ("dec_ffn", FfnWeights)]


# the gradient structs have the members of the weight structs (ait_mha_grads / ait_ffn_grads), and
# ait_transformer_grads those of ait_transformer_weights without pos_table
MhaGrads, FfnGrads = MhaWeights, FfnWeights


class TransformerGrads(ctypes.Structure):
    """ait_transformer_grads."""
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "enc_emb_w", "enc_emb_b", "dec_emb_w", "dec_emb_b", "dec_trans_w", "dec_trans_b",
        "enc_ln_g", "enc_ln_b", "dec_ln_g", "dec_ln_b")] + \
        [("enc_slf", MhaGrads), ("dec_slf", MhaGrads), ("dec_enc", MhaGrads),
         ("enc_ffn", FfnGrads), ("dec_ffn", FfnGrads)]


PROBE_GEMM, PROBE_ROI_FWD, PROBE_ROI_BWD = 1, 2, 3


class LaunchCtx(ctypes.Structure):
    """ait_launch_ctx of include/ait_hip.h: caller-owned scheduler scratch of the persistent GEMM + probe."""
    _fields_ = [("sched_ws", ctypes.c_void_p), ("sched_ws_bytes", ctypes.c_size_t), ("probe", ctypes.c_void_p),
                ("flags", ctypes.c_uint)]


# ---- the host side's launch contexts: one scheduler workspace per (device, stream), allocated with torch and
# initialised once; the probe in force (a Python-level setting of THIS host layer: the library keeps none) ----
_SCHED = {}          # (device index, stream handle) -> (uint8 tensor, LaunchCtx without probe)
_ACTIVE_PROBE = None
USE_SCHED_WS = True  # test hook: False = launch without scheduler scratch (static work lists, whole tiles)
CTX_NATIVE_F32 = 1   # ait_launch_ctx::flags
CTX_BF16 = 2
CTX_IO_BF16 = 4
NATIVE_F32 = False   # ops.set_matmul_dtype("f32_native"): dense products on v_mfma_f32_32x32x2_f32 (default: bf16 3-way split)
BF16_PRODUCTS = False   # ops.set_matmul_dtype("bf16"): operands rounded to bf16 in registers, one MFMA per block


def current_flags():
    """ait_launch_ctx::flags of the product form in force (ops.set_matmul_dtype)"""
    return CTX_BF16 if BF16_PRODUCTS else (CTX_NATIVE_F32 if NATIVE_F32 else 0)


def launch_ctx(device=None, flags=None):
    """byref(ait_launch_ctx) for a launch on the current stream of `device`.  flags: None = the product form in force;
    an int = exactly these flags (a backward pass gives the flags of ITS forward: the saved activations of the bf16-storage
    feed-forward are bf16 only if the forward ran in that mode)"""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = torch.cuda.current_stream(dev)
    ctx = LaunchCtx()
    if USE_SCHED_WS:
        key = (idx, st.cuda_stream)
        ent = _SCHED.get(key)
        if ent is None:
            if torch.cuda.is_current_stream_capturing():
                # a first use inside a stream capture would take the workspace from the graph's private pool and CAPTURE
                # the zeroing of its control words instead of running it: later launches would read garbage tickets
                raise AitHipError("the GEMM scheduler workspace of this stream is created on first use, which must not happen "
                                  "inside a stream capture: call ait_amd._lib.prewarm(stream) (or run one product on the "
                                  "stream) before capturing")
            with torch.cuda.device(dev):
                L = lib()
                nbytes = int(L.ait_gemm_workspace_bytes())
                if nbytes <= 0:
                    raise AitHipError("ait_gemm_workspace_bytes failed")
                buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                check(L.ait_gemm_workspace_init(ctypes.c_void_p(buf.data_ptr()), nbytes, ctypes.c_void_p(st.cuda_stream)),
                      "ait_gemm_workspace_init")
            ent = _SCHED[key] = (buf, nbytes)
        ctx.sched_ws, ctx.sched_ws_bytes = ent[0].data_ptr(), ent[1]
    pr = _ACTIVE_PROBE
    if pr is not None and pr._p and pr.device_index == idx:
        ctx.probe = pr._p
    ctx.flags = current_flags() if flags is None else int(flags)
    return ctypes.byref(ctx)


def prewarm(stream=None, device=None):
    """create and initialise the scheduler workspace of `stream` (default: the current one) -- to be called before a
    stream capture whose first product would otherwise do it inside the capture"""
    if stream is None:
        launch_ctx(device)
    else:
        with torch.cuda.stream(stream):
            launch_ctx(device if device is not None else stream.device)


def release_sched_workspaces():
    """drop the cached scheduler workspaces (their streams must be idle)"""
    _SCHED.clear()


class Probe:
    """Measurement probe of include/ait_hip.h: while active (`with probe:`), every launch context this host layer
    builds carries it, so the library brackets each GEMM / RoIAlign launch with HIP events on its launch stream.

        with Probe(20000) as pr:  ...run...
        torch.cuda.synchronize(); rows = pr.entries()   # [(kind, work, ms, dims6), ...]
    """

    def __init__(self, capacity=65536):
        self._L = lib()
        self.device_index = torch.cuda.current_device()
        self._p = self._L.ait_probe_create(int(capacity))
        if not self._p:
            raise AitHipError("ait_probe_create failed")

    def __enter__(self):
        global _ACTIVE_PROBE
        torch.cuda.synchronize()                 # nothing in flight holds the probe while it is reset
        self._L.ait_probe_reset(self._p)
        _ACTIVE_PROBE = self
        return self

    def __exit__(self, *exc):
        global _ACTIVE_PROBE
        _ACTIVE_PROBE = None
        return False

    def overflowed(self):
        """launches the probe saw beyond its capacity (their timings were not recorded)"""
        return max(0, self._L.ait_probe_count(self._p) - self._L.ait_probe_capacity(self._p))

    def entries(self):
        out = []
        kind, work, ms = ctypes.c_int(), ctypes.c_double(), ctypes.c_float()
        dims = (ctypes.c_int * 6)()
        n = min(self._L.ait_probe_count(self._p), self._L.ait_probe_capacity(self._p))
        for i in range(n):
            rc = self._L.ait_probe_get(self._p, i, ctypes.byref(kind), ctypes.byref(work), ctypes.byref(ms), dims)
            check(rc, "ait_probe_get (synchronise the stream first)")
            out.append((kind.value, work.value, ms.value, tuple(dims)))
        return out

    def close(self):
        global _ACTIVE_PROBE
        if self._p:
            if _ACTIVE_PROBE is self:
                _ACTIVE_PROBE = None
            torch.cuda.synchronize()
            self._L.ait_probe_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def check(rc: int, what: str):
    if rc != 0:
        raise AitHipError("%s failed: %s (%d)" % (what, lib().ait_strerror(rc).decode(), rc))


def dev_ptr(t: torch.Tensor, dtype=torch.float32, dense_any_format=False):
    """data_ptr of a contiguous GPU tensor of the given dtype (else raise).  dense_any_format
    also accepts a channels-last-contiguous 4-d tensor (elementwise kernels that take the memory
    order from their (n, C, HW) arguments)."""
    if not t.is_cuda:
        raise AitHipError("tensor must live on a GPU (got %s): the hot path has no CPU fallback"
                          % t.device)
    if t.dtype != dtype:
        raise AitHipError("expected %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous() and not (dense_any_format and t.dim() == 4
                                      and t.is_contiguous(memory_format=torch.channels_last)):
        raise AitHipError("tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def cur_stream(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
