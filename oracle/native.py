"""oracle/native.py -- TEST INFRASTRUCTURE ONLY (see oracle/native.c header).

Builds oracle/native.c with gcc (-O2 -ffp-contract=off, single thread) into
oracle/_build/liboracle.so and exposes numpy-level wrappers.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liboracle.so")

_lib = None


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(
            ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
             "-o", LIB, SRC, "-lm"])
    return LIB


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int64)
        i, f = ctypes.c_int, ctypes.c_float
        L.orc_roi_align_fwd.argtypes = [fp, fp, i, i, i, i, i, i, i, f, i, fp]
        L.orc_roi_align_fwd.restype = i
        L.orc_roi_align_bwd.argtypes = [fp, fp, i, i, i, i, i, i, i, f, i, fp]
        L.orc_roi_align_bwd.restype = i
        L.orc_nms.argtypes = [fp, ip, ctypes.c_int64, f, ip]
        L.orc_nms.restype = ctypes.c_int64
        _lib = L
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def roi_align_fwd(feat, rois, pooled=(7, 7), scale=1.0 / 16.0, sampling_ratio=0):
    feat, pf = _f(feat)
    rois, pr = _f(rois)
    B, C, H, W = feat.shape
    n = rois.shape[0]
    out = np.empty((n, C, pooled[0], pooled[1]), np.float32)
    rc = lib().orc_roi_align_fwd(pf, pr, n, B, C, H, W, pooled[0], pooled[1], scale,
                                 sampling_ratio, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    if rc != 0:
        raise ValueError("roi batch index out of range")
    return out


def roi_align_bwd(grad_out, rois, in_shape, scale=1.0 / 16.0, sampling_ratio=0):
    grad_out, pg = _f(grad_out)
    rois, pr = _f(rois)
    B, C, H, W = in_shape
    n, _, PH, PW = grad_out.shape
    gin = np.zeros((B, C, H, W), np.float32)
    rc = lib().orc_roi_align_bwd(pg, pr, n, B, C, H, W, PH, PW, scale, sampling_ratio,
                                 gin.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    if rc != 0:
        raise ValueError("roi batch index out of range")
    return gin


def nms(dets, scores, thr):
    """Reference semantics of model._C.nms on CPU: returns int64 kept indices, ascending."""
    dets, pd = _f(dets)
    n = dets.shape[0]
    if n == 0:
        return np.empty((0,), np.int64)
    scores = np.asarray(scores, np.float32)
    order = np.ascontiguousarray(np.argsort(-scores, kind="stable").astype(np.int64))
    keep = np.empty((n,), np.int64)
    ip = ctypes.POINTER(ctypes.c_int64)
    k = lib().orc_nms(pd, order.ctypes.data_as(ip), n, float(thr), keep.ctypes.data_as(ip))
    return keep[:k].copy()
