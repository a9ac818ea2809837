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


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois):
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
    ws_bytes = max(L.ait_nms_batched_workspace_bytes(b, n), L.ait_nms_workspace_bytes(n))
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
