"""Host-side mirror of the reference's lib/model/roi_layers package on top of libait_hip.so.

Same names, argument meaning and error behaviour as the reference:
  ROIAlign(output_size, spatial_scale, sampling_ratio)(input, rois)   roi_layers/roi_align.py:49-67
  roi_align(input, rois, output_size, spatial_scale, sampling_ratio)  roi_layers/roi_align.py:12-46
  nms(dets, scores, threshold) -> int64 kept indices, ascending       roi_layers/nms.py:5
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair
import ctypes

from . import _lib

class _ROIAlign(Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio):
        ph, pw = _pair(output_size)
        input = input.contiguous()
        roi = roi.contiguous().float()
        if roi.dim() != 2 or roi.size(1) != 5:
            raise ValueError("rois must be [K,5] (batch_index, x1, y1, x2, y2)")
        B, C, H, W = input.shape
        _lib.dev_ptr(input)  # raises unless this is a GPU fp32 tensor: no CPU fallback
        out = torch.empty((roi.size(0), C, ph, pw), dtype=input.dtype, device=input.device)
        with torch.cuda.device(input.device):
            rc = _lib.lib().ait_roi_align_fwd(
                _lib.dev_ptr(input), _lib.dev_ptr(roi), roi.size(0), B, C, H, W, ph, pw,
                float(spatial_scale), int(sampling_ratio), _lib.dev_ptr(out),
                _lib.cur_stream(input.device))
        _lib.check(rc, "ait_roi_align_fwd")
        ctx.save_for_backward(roi)
        ctx.geom = (B, C, H, W, ph, pw, float(spatial_scale), int(sampling_ratio))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (roi,) = ctx.saved_tensors
        B, C, H, W, ph, pw, scale, sr = ctx.geom
        grad_output = grad_output.contiguous()
        grad_input = torch.empty((B, C, H, W), dtype=grad_output.dtype, device=grad_output.device)
        with torch.cuda.device(grad_output.device):
            rc = _lib.lib().ait_roi_align_bwd(
                _lib.dev_ptr(grad_output), _lib.dev_ptr(roi), roi.size(0), B, C, H, W, ph, pw,
                scale, sr, _lib.dev_ptr(grad_input), _lib.cur_stream(grad_output.device))
        _lib.check(rc, "ait_roi_align_bwd")
        return grad_input, None, None, None, None


roi_align = _ROIAlign.apply


def _nhwc_workspace(n_rois, H, W, ph, pw, device):
    nbytes = _lib.lib().ait_roi_align_nhwc_workspace_bytes(n_rois, H, W, ph, pw)
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device), int(nbytes)


class _ROIAlignChannelsLast(Function):
    """Same operator, same logical shapes ([B,C,H,W] in, [K,C,ph,pw] out), but computed on
    channels-last memory by ait_roi_align_nhwc_*: the input is taken as [B,H,W,C] and the result IS
    the token-major [K, ph*pw, C] matrix (returned as its [K,C,ph,pw] view), so a channels-last
    producer and a token-major consumer meet without a transpose."""

    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio):
        ph, pw = _pair(output_size)
        roi = roi.contiguous().float()
        if roi.dim() != 2 or roi.size(1) != 5:
            raise ValueError("rois must be [K,5] (batch_index, x1, y1, x2, y2)")
        _lib.dev_ptr(roi)
        B, C, H, W = input.shape
        x = input.contiguous(memory_format=torch.channels_last)
        K = roi.size(0)
        out = torch.empty((K, ph, pw, C), dtype=torch.float32, device=input.device)
        ws, nbytes = _nhwc_workspace(K, H, W, ph, pw, input.device)
        with torch.cuda.device(input.device):
            rc = _lib.lib().ait_roi_align_nhwc_fwd(
                _lib.dev_ptr(x, torch.float32, True), _lib.dev_ptr(roi), K, B, C, H, W, ph, pw,
                float(spatial_scale), int(sampling_ratio), ctypes.c_void_p(ws.data_ptr()), nbytes,
                _lib.dev_ptr(out), _lib.launch_ctx(input.device), _lib.cur_stream(input.device))
        _lib.check(rc, "ait_roi_align_nhwc_fwd")
        ctx.save_for_backward(roi)
        ctx.geom = (B, C, H, W, ph, pw, float(spatial_scale), int(sampling_ratio))
        return out.permute(0, 3, 1, 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (roi,) = ctx.saved_tensors
        B, C, H, W, ph, pw, scale, sr = ctx.geom
        g = grad_output.permute(0, 2, 3, 1).contiguous()          # [K, ph, pw, C]; free for a token-major grad
        K = roi.size(0)
        grad_input = torch.empty((B, H, W, C), dtype=torch.float32, device=g.device)
        ws, nbytes = _nhwc_workspace(K, H, W, ph, pw, g.device)
        with torch.cuda.device(g.device):
            rc = _lib.lib().ait_roi_align_nhwc_bwd(
                _lib.dev_ptr(g), _lib.dev_ptr(roi), K, B, C, H, W, ph, pw, scale, sr,
                ctypes.c_void_p(ws.data_ptr()), nbytes, _lib.dev_ptr(grad_input), _lib.launch_ctx(g.device),
                _lib.cur_stream(g.device))
        _lib.check(rc, "ait_roi_align_nhwc_bwd")
        return grad_input.permute(0, 3, 1, 2), None, None, None, None


def roi_align_channels_last_supported(C, output_size):
    ph, pw = _pair(output_size)
    return pw == 7 and ph <= 7 and C % 4 == 0 and C <= 4096


class ROIAlign(nn.Module):
    """roi_layers/roi_align.py:49-67.  channels_last=True selects the channels-last / token-major
    kernels (same values up to fp32 summation order; see include/ait_hip.h); the default is the
    bit-exact NCHW operator."""

    def __init__(self, output_size, spatial_scale, sampling_ratio, channels_last=False):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio
        self.channels_last = channels_last

    def forward(self, input, rois):
        if self.channels_last and input.is_cuda and roi_align_channels_last_supported(input.size(1), self.output_size):
            return _ROIAlignChannelsLast.apply(input, rois, self.output_size, self.spatial_scale,
                                               self.sampling_ratio)
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio)

    def __repr__(self):
        return "%s(output_size=%s, spatial_scale=%s, sampling_ratio=%s)" % (
            self.__class__.__name__, self.output_size, self.spatial_scale, self.sampling_ratio)


def nms_sorted(dets, threshold, max_keep=0):
    """NMS over boxes that are already sorted by descending score (what
    rpn/proposal_layer.py:153 passes).  Returns (keep[int64, n], n_keep[int32 device scalar]):
    no host synchronisation; keep[:n_keep] are the survivors in ascending index order."""
    dets = dets.contiguous().float()
    _lib.dev_ptr(dets)
    n = dets.size(0)
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=dets.device)
    n_keep = torch.zeros((1,), dtype=torch.int32, device=dets.device)
    L = _lib.lib()
    ws_bytes = L.ait_nms_workspace_bytes(n)
    ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=dets.device)
    with torch.cuda.device(dets.device):
        rc = L.ait_nms(_lib.dev_ptr(dets) if n else None, None, n, float(threshold),
                       int(max_keep), _lib.dev_ptr(ws, torch.uint8), ws_bytes,
                       _lib.dev_ptr(keep, torch.int64), _lib.dev_ptr(n_keep, torch.int32),
                       _lib.cur_stream(dets.device))
    _lib.check(rc, "ait_nms")
    return keep, n_keep


def nms_sorted_batched(dets, threshold, max_keep=0):
    """dets [b, n, 4], each image's boxes sorted by descending score.  Returns
    (keep [b, n] int64, n_keep [b] int32); one launch pair for the whole batch, no host sync."""
    dets = dets.contiguous().float()
    _lib.dev_ptr(dets)
    b, n = dets.size(0), dets.size(1)
    keep = torch.empty((b, max(n, 1)), dtype=torch.int64, device=dets.device)
    n_keep = torch.zeros((b,), dtype=torch.int32, device=dets.device)
    L = _lib.lib()
    ws_bytes = L.ait_nms_batched_workspace_bytes(b, n)
    ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=dets.device)
    with torch.cuda.device(dets.device):
        rc = L.ait_nms_batched(_lib.dev_ptr(dets) if n else None, b, n, float(threshold),
                               int(max_keep), _lib.dev_ptr(ws, torch.uint8), ws_bytes,
                               _lib.dev_ptr(keep, torch.int64), keep.stride(0),
                               _lib.dev_ptr(n_keep, torch.int32), _lib.cur_stream(dets.device))
    _lib.check(rc, "ait_nms_batched")
    return keep, n_keep


def nms(dets, scores, threshold):
    """Drop-in for model._C.nms: int64 indices of kept boxes, ascending (nms_cpu.cpp:64)."""
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=dets.device)
    dets = dets.contiguous().float()
    _lib.dev_ptr(dets)
    n = dets.size(0)
    order = torch.sort(scores.float(), dim=0, descending=True, stable=True)[1].contiguous()
    keep = torch.empty((n,), dtype=torch.int64, device=dets.device)
    n_keep = torch.zeros((1,), dtype=torch.int32, device=dets.device)
    L = _lib.lib()
    ws_bytes = L.ait_nms_workspace_bytes(n)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dets.device)
    with torch.cuda.device(dets.device):
        rc = L.ait_nms(_lib.dev_ptr(dets), _lib.dev_ptr(order, torch.int64), n, float(threshold),
                       0, _lib.dev_ptr(ws, torch.uint8), ws_bytes,
                       _lib.dev_ptr(keep, torch.int64), _lib.dev_ptr(n_keep, torch.int32),
                       _lib.cur_stream(dets.device))
    _lib.check(rc, "ait_nms")
    return keep[: int(n_keep.item())]
