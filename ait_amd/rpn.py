"""Proposal machinery around the hot path: host-side mirror of the reference's lib/model/rpn
package (same class names, constructor arguments, forward signatures and outputs).

  generate_anchors          rpn/generate_anchors.py:45-105
  bbox_transform_inv / clip_boxes / bbox_transform_batch / bbox_overlaps_batch
                            rpn/bbox_transform.py:77-103,125-133,38-75,168-257
  _ProposalLayer            rpn/proposal_layer.py:28-166   (NMS = libait_hip.so, no host sync)
  _AnchorTargetLayer        rpn/anchor_target_layer.py:27-199
  _ProposalTargetLayer      rpn/proposal_target_layer_cascade.py:17-220
  _RPN                      rpn/rpn.py:18-128
  _smooth_l1_loss           lib/model/utils/net_utils.py:75-89

MI355X notes: box decode / clip / IoU are small batched tensor ops on the device; NMS is the
on-device HIP kernel with early exit at post_nms_topN (the reference's GPU path copies an
N x N/64 bitmask to the host and scans it there); the two samplers keep the reference's exact
NumPy global-RNG call sequence (index parity) but move ONE small tensor per layer to the host
instead of synchronising per image.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import cfg
from . import _lib, ops
from .roi_layers import nms_sorted_batched


# ------------------------------------------------------------------------------------------
# anchors
# ------------------------------------------------------------------------------------------
def _centered(ws, hs, cx, cy):
    ws, hs = np.asarray(ws, np.float64)[:, None], np.asarray(hs, np.float64)[:, None]
    return np.hstack((cx - 0.5 * (ws - 1), cy - 0.5 * (hs - 1), cx + 0.5 * (ws - 1), cy + 0.5 * (hs - 1)))


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=2 ** np.arange(3, 6)):
    """Anchor windows (x1,y1,x2,y2) around a (0,0,base-1,base-1) cell: for every aspect ratio
    (area preserved, sides rounded), every scale.  Row order = ratio-major, scale-minor."""
    ratios, scales = np.asarray(ratios, np.float64), np.asarray(scales, np.float64)
    ctr = 0.5 * (base_size - 1)
    area = float(base_size * base_size)
    ws = np.round(np.sqrt(area / ratios))
    hs = np.round(ws * ratios)
    return np.vstack([_centered(w * scales, h * scales, ctr, ctr) for w, h in zip(ws, hs)])


# The box arithmetic of the three layers below runs on the library's kernels (csrc/boxes.hip) for GPU tensors.
# The tensor expressions remain for CPU tensors (the host-logic tests) and are COUNTED (ops.fallback_count);
# a GPU tensor of the wrong dtype raises instead of silently leaving the kernels.  _BOX_KERNELS is a test hook
# (tests/test_gpu_ops.py compares the kernels with the tensor expressions), set from Python only.
_BOX_KERNELS = True


def _lib_kernels(t, site="box arithmetic"):
    if t.is_cuda and _BOX_KERNELS:
        if t.dtype != torch.float32:
            raise _lib.AitHipError("%s: float32 tensors expected on the GPU, got %s" % (site, t.dtype))
        return True
    ops.note_fallback(site, t)
    return False


class _AnchorGrid:
    """Anchors shifted over an H x W feature map (stride 16), cached per (H, W, device)."""

    def __init__(self, feat_stride, scales, ratios):
        self.stride = feat_stride
        self.base = torch.from_numpy(generate_anchors(scales=np.array(scales), ratios=np.array(ratios))).float()
        self._cache = {}

    @property
    def A(self):
        return self.base.size(0)

    def get(self, H, W, device):
        key = (H, W, str(device))
        if key not in self._cache:
            sx = torch.arange(W, dtype=torch.float32) * self.stride
            sy = torch.arange(H, dtype=torch.float32) * self.stride
            shifts = torch.stack([sx.repeat(H), sy.repeat_interleave(W), sx.repeat(H), sy.repeat_interleave(W)], 1)
            self._cache[key] = (self.base.view(1, -1, 4) + shifts.view(-1, 1, 4)).view(-1, 4).to(device)
        return self._cache[key]


# ------------------------------------------------------------------------------------------
# box arithmetic (+1 pixel convention throughout)
# ------------------------------------------------------------------------------------------
def bbox_transform_inv(boxes, deltas, batch_size=None):
    """boxes [b,N,4] (or [N,4] broadcast), deltas [b,N,4] -> decoded boxes [b,N,4]."""
    w = boxes[..., 2] - boxes[..., 0] + 1.0
    h = boxes[..., 3] - boxes[..., 1] + 1.0
    cx = boxes[..., 0] + 0.5 * w
    cy = boxes[..., 1] + 0.5 * h
    pcx = deltas[..., 0] * w + cx
    pcy = deltas[..., 1] * h + cy
    pw = torch.exp(deltas[..., 2]) * w
    ph = torch.exp(deltas[..., 3]) * h
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), -1)


def clip_boxes(boxes, im_shape, batch_size=None):
    """Clamp [b,N,4] boxes to [0, W-1] x [0, H-1] with im_shape[b] = (H, W, scale)."""
    hi = torch.stack((im_shape[:, 1], im_shape[:, 0], im_shape[:, 1], im_shape[:, 0]), 1) - 1
    hi = hi.to(boxes.dtype).unsqueeze(1)
    return torch.minimum(boxes.clamp(min=0), hi)


def bbox_transform_batch(ex_rois, gt_rois):
    """Regression targets of gt w.r.t. ex; ex [N,4] or [b,N,4], gt [b,N,4]."""
    if ex_rois.dim() == 2:
        ex_rois = ex_rois.unsqueeze(0)
    ew = ex_rois[..., 2] - ex_rois[..., 0] + 1.0
    eh = ex_rois[..., 3] - ex_rois[..., 1] + 1.0
    ecx = ex_rois[..., 0] + 0.5 * ew
    ecy = ex_rois[..., 1] + 0.5 * eh
    gw = gt_rois[..., 2] - gt_rois[..., 0] + 1.0
    gh = gt_rois[..., 3] - gt_rois[..., 1] + 1.0
    gcx = gt_rois[..., 0] + 0.5 * gw
    gcy = gt_rois[..., 1] + 0.5 * gh
    return torch.stack(((gcx - ecx) / ew, (gcy - ecy) / eh, torch.log(gw / ew), torch.log(gh / eh)), 2)


def bbox_overlaps_batch(anchors, gt_boxes):
    """IoU [b,N,K] of anchors ([N,4], [b,N,4] or [b,N,5] with a leading batch column) against
    gt_boxes [b,K,>=4].  Zero-area gt columns -> 0, zero-area anchor rows -> -1."""
    b = gt_boxes.size(0)
    if anchors.dim() == 2:
        anchors = anchors.unsqueeze(0).expand(b, -1, 4)
    elif anchors.size(2) == 5:
        anchors = anchors[:, :, 1:5]
    gt = gt_boxes[:, :, :4]
    gw = gt[:, :, 2] - gt[:, :, 0] + 1
    gh = gt[:, :, 3] - gt[:, :, 1] + 1
    aw = anchors[:, :, 2] - anchors[:, :, 0] + 1
    ah = anchors[:, :, 3] - anchors[:, :, 1] + 1
    g_area = (gw * gh).unsqueeze(1)
    a_area = (aw * ah).unsqueeze(2)
    A, G = anchors.unsqueeze(2), gt.unsqueeze(1)
    iw = (torch.min(A[..., 2], G[..., 2]) - torch.max(A[..., 0], G[..., 0]) + 1).clamp(min=0)
    ih = (torch.min(A[..., 3], G[..., 3]) - torch.max(A[..., 1], G[..., 1]) + 1).clamp(min=0)
    inter = iw * ih
    ov = inter / (a_area + g_area - inter)
    ov = ov.masked_fill(((gw == 1) & (gh == 1)).unsqueeze(1), 0)
    ov = ov.masked_fill(((aw == 1) & (ah == 1)).unsqueeze(2), -1)
    return ov


def _smooth_l1_loss(bbox_pred, bbox_targets, bbox_inside_weights, bbox_outside_weights, sigma=1.0,
                    dim=[1]):
    s2 = sigma ** 2
    d = bbox_inside_weights * (bbox_pred - bbox_targets)
    ad = d.abs()
    quad = (ad < 1.0 / s2).detach().float()
    loss = bbox_outside_weights * (d * d * (s2 / 2.0) * quad + (ad - 0.5 / s2) * (1.0 - quad))
    for i in sorted(dim, reverse=True):
        loss = loss.sum(i)
    return loss.mean()


# ------------------------------------------------------------------------------------------
# proposal layer
# ------------------------------------------------------------------------------------------
class _ProposalLayer(nn.Module):
    """(rpn_cls_prob, rpn_bbox_pred, im_info, 'TRAIN'|'TEST') -> rois [b, post_nms_topN, 5]
    = (batch index, x1, y1, x2, y2), zero rows after the last survivor."""

    def __init__(self, feat_stride, scales, ratios):
        super().__init__()
        self._grid = _AnchorGrid(feat_stride, scales, ratios)
        self._num_anchors = self._grid.A

    def forward(self, input):
        probs, deltas, im_info, cfg_key = input
        return self._run(probs, deltas, im_info, cfg_key)

    def _run(self, probs, deltas, im_info, cfg_key):
        if _lib_kernels(probs, "proposal_layer"):
            return self._run_hip(probs, deltas, im_info, cfg_key)
        A = self._num_anchors
        pre_n = cfg[cfg_key].RPN_PRE_NMS_TOP_N
        post_n = cfg[cfg_key].RPN_POST_NMS_TOP_N
        thr = cfg[cfg_key].RPN_NMS_THRESH
        b, _, H, W = deltas.shape
        scores = probs[:, A:].permute(0, 2, 3, 1).reshape(b, -1)        # fg probabilities
        deltas = deltas.permute(0, 2, 3, 1).reshape(b, -1, 4)
        anchors = self._grid.get(H, W, deltas.device)
        boxes = clip_boxes(bbox_transform_inv(anchors.unsqueeze(0), deltas), im_info)
        order = torch.sort(scores, 1, True)[1]
        if 0 < pre_n < scores.numel():          # (the reference compares with numel of the batch)
            order = order[:, :pre_n]
        cand = torch.gather(boxes, 1, order.unsqueeze(2).expand(-1, -1, 4)).contiguous()   # [b, n, 4]
        n = cand.size(1)
        keep, n_keep = nms_sorted_batched(cand, thr, post_n)          # one launch pair, no sync
        idx = keep[:, :post_n].clamp_(0, n - 1)
        if idx.size(1) < post_n:
            idx = F.pad(idx, (0, post_n - idx.size(1)))
        sel = torch.gather(cand, 1, idx.unsqueeze(2).expand(-1, -1, 4))
        live = torch.arange(post_n, device=scores.device).unsqueeze(0) < n_keep.unsqueeze(1)
        out = scores.new_zeros(b, post_n, 5)
        out[:, :, 1:] = torch.where(live.unsqueeze(2), sel, torch.zeros_like(sel))
        out[:, :, 0] = torch.arange(b, device=scores.device, dtype=scores.dtype).unsqueeze(1)
        return out


    def _run_hip(self, probs, deltas, im_info, cfg_key):
        """the same layer on the library's kernels: decode + clip + score re-layout in one launch, sort,
        gather, batched NMS, assembly in one launch (include/ait_hip.h "Box arithmetic")"""
        A = self._num_anchors
        pre_n = cfg[cfg_key].RPN_PRE_NMS_TOP_N
        post_n = cfg[cfg_key].RPN_POST_NMS_TOP_N
        thr = cfg[cfg_key].RPN_NMS_THRESH
        b, _, H, W = deltas.shape
        dev = probs.device
        probs, deltas = probs.contiguous(), deltas.contiguous()      # NCHW, as the RPN head's convolutions leave them
        im_info = im_info.to(torch.float32).contiguous()
        anchors = self._grid.get(H, W, dev)
        N = H * W * A
        boxes = torch.empty((b, N, 4), dtype=torch.float32, device=dev)
        scores = torch.empty((b, N), dtype=torch.float32, device=dev)
        L = _lib.lib()
        with torch.cuda.device(dev):
            st = _lib.cur_stream(dev)
            _lib.check(L.ait_rpn_decode(_lib.dev_ptr(probs), _lib.dev_ptr(deltas), _lib.dev_ptr(anchors),
                                        _lib.dev_ptr(im_info), b, A, H, W, _lib.dev_ptr(boxes), _lib.dev_ptr(scores), st),
                       "ait_rpn_decode")
            order = torch.sort(scores, 1, True)[1]
            if 0 < pre_n < scores.numel():          # (the reference compares with numel of the batch)
                order = order[:, :pre_n]
            cand = torch.gather(boxes, 1, order.unsqueeze(2).expand(-1, -1, 4)).contiguous()   # [b, n, 4]
            n = cand.size(1)
            keep, n_keep = nms_sorted_batched(cand, thr, post_n)          # one launch pair, no sync
            out = torch.empty((b, post_n, 5), dtype=torch.float32, device=dev)
            _lib.check(L.ait_proposals_assemble(_lib.dev_ptr(cand), n, _lib.dev_ptr(keep, torch.int64), keep.stride(0),
                                                keep.size(1), _lib.dev_ptr(n_keep, torch.int32), b, post_n,
                                                _lib.dev_ptr(out), st), "ait_proposals_assemble")
        return out.to(probs.dtype)


# ------------------------------------------------------------------------------------------
# RPN training targets
# ------------------------------------------------------------------------------------------
class _AnchorTargetLayer(nn.Module):
    """(rpn_cls_score, gt_boxes, im_info, num_boxes) -> [labels [b,1,A*H,W], bbox_targets
    [b,4A,H,W], inside weights, outside weights]."""

    def __init__(self, feat_stride, scales, ratios):
        super().__init__()
        self._grid = _AnchorGrid(feat_stride, scales, ratios)
        self._num_anchors = self._grid.A
        self._allowed_border = 0
        self._inside = {}

    def forward(self, input):
        rpn_cls_score, gt_boxes, im_info, num_boxes = input
        H, W = rpn_cls_score.size(2), rpn_cls_score.size(3)
        pending = self._pending
        self._pending = None
        if pending is None or pending[0] != (H, W, gt_boxes.data_ptr()):
            pending = self.prepare(gt_boxes, im_info, H, W)
        return self.finish(pending)

    _pending = None

    def _inside_set(self, all_anchors, H, W, im_h, im_w):
        """the anchors inside an im_h x im_w image (anchor_target_layer.py:84-90) and what derives from them,
        cached per (H, W, im_h, im_w, device): the set depends on the IMAGE size, not only on the feature size --
        about 16 image widths share one W"""
        dev = all_anchors.device
        key = (H, W, int(im_h), int(im_w), str(dev))
        ent = self._inside.get(key)
        if ent is None:
            bd = self._allowed_border
            inside = ((all_anchors[:, 0] >= -bd) & (all_anchors[:, 1] >= -bd) &
                      (all_anchors[:, 2] < int(im_w) + bd) & (all_anchors[:, 3] < int(im_h) + bd))
            inds = torch.nonzero(inside).view(-1)
            if len(self._inside) >= 64:          # (bounded: a loader produces a few dozen padded sizes)
                self._inside.pop(next(iter(self._inside)))
            ent = self._inside[key] = {"inds": inds, "hw": torch.tensor([int(im_h), int(im_w)], device=dev)}
        return key, ent

    def prepare(self, gt_boxes, im_info, H, W, im_hw_hint=None):
        """Everything this layer computes depends only on the INPUTS (anchors, gt boxes, image size), not on
        the network: the device part (IoU, arg-max, labels) and the two per-image counts the host-side
        sampling needs are therefore enqueued at the very start of the detector's forward, and the counts
        travel to the host while the GPU runs the backbone (no GPU-idle gap at the sampler).
        Returns the pending state consumed by finish() / forward().

        The inside-image anchor set is a function of im_info[0] = (im_h, im_w) (like the reference, row 0
        decides for the batch), which lives on the device.  im_hw_hint: the host's PREDICTION of it (the
        detector passes the image tensor's height and width, which is what the reference's loader writes
        there, roibatchLoader.py:228-255); the set for the predicted size is used at once and a device-side
        comparison with the real im_info travels to the host with the counts -- finish() redoes the layer on
        a mismatch.  Without a hint im_info is read back (one host synchronisation)."""
        b = gt_boxes.size(0)
        dev = gt_boxes.device
        all_anchors = self._grid.get(H, W, dev)
        if im_hw_hint is None:
            im_h, im_w = float(im_info[0][0]), float(im_info[0][1])        # D2H: synchronises
            key, ent = self._inside_set(all_anchors, H, W, int(im_h), int(im_w))
            mismatch = None
        else:
            key, ent = self._inside_set(all_anchors, H, W, int(im_hw_hint[0]), int(im_hw_hint[1]))
            # (int(): the reference truncates with long(), anchor_target_layer.py:86-87)
            mismatch = (im_info[0, :2].to(torch.int64) != ent["hw"]).any().to(torch.int64).view(1)
        inds_inside = ent["inds"]
        redo = (gt_boxes, im_info, H, W)
        if _lib_kernels(gt_boxes, "anchor_target_layer"):
            return self._prepare_hip(gt_boxes, all_anchors, inds_inside, ent, H, W, mismatch, redo)
        anchors = all_anchors[inds_inside]

        overlaps = bbox_overlaps_batch(anchors, gt_boxes)                 # [b, n_in, G]
        max_ov, argmax_ov = overlaps.max(2)
        gt_max = overlaps.max(1)[0]
        n_in = anchors.size(0)
        labels = gt_boxes.new_full((b, n_in), -1)
        if not cfg.TRAIN.RPN_CLOBBER_POSITIVES:
            labels[max_ov < cfg.TRAIN.RPN_NEGATIVE_OVERLAP] = 0
        gt_max = torch.where(gt_max == 0, torch.full_like(gt_max, 1e-5), gt_max)
        is_best = (overlaps == gt_max.unsqueeze(1)).sum(2) > 0
        labels[is_best] = 1
        labels[max_ov >= cfg.TRAIN.RPN_POSITIVE_OVERLAP] = 1
        if cfg.TRAIN.RPN_CLOBBER_POSITIVES:
            labels[max_ov < cfg.TRAIN.RPN_NEGATIVE_OVERLAP] = 0
        counts = torch.stack(((labels == 1).sum(1), (labels == 0).sum(1)), 1)       # [b, 2] int64
        host, ev = self._to_host(counts, mismatch)
        return ((H, W, gt_boxes.data_ptr()), gt_boxes, inds_inside, anchors, argmax_ov, labels, host, ev,
                all_anchors.size(0), None, redo)

    @staticmethod
    def _to_host(counts, mismatch):
        """the 2b class sizes (+ the image-size check) -> pinned host memory, asynchronously; an event marks arrival"""
        flat = counts.reshape(-1) if mismatch is None else torch.cat([counts.reshape(-1), mismatch])
        if flat.is_cuda:
            host = torch.empty((flat.numel(),), dtype=torch.int64, pin_memory=True)
            host.copy_(flat, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            return host, ev
        return flat, None

    def _prepare_hip(self, gt_boxes, all_anchors, inds_inside, ent, H, W, mismatch, redo):
        """the device part of prepare() in three launches (ait_anchor_classify)"""
        dev = gt_boxes.device
        if "anchors" not in ent:
            ent["anchors"] = all_anchors[inds_inside].float().contiguous()
            inside_pos = torch.full((all_anchors.size(0),), -1, dtype=torch.int32, device=dev)
            inside_pos[inds_inside] = torch.arange(inds_inside.numel(), dtype=torch.int32, device=dev)
            ent["inside_pos"] = inside_pos
        anchors, inside_pos = ent["anchors"], ent["inside_pos"]
        gt = gt_boxes.contiguous()
        b, G, n_in = gt.size(0), gt.size(1), anchors.size(0)
        max_ov = torch.empty((b, n_in), dtype=torch.float32, device=dev)
        argmax = torch.empty((b, n_in), dtype=torch.int64, device=dev)
        gt_max = torch.empty((b, G), dtype=torch.int32, device=dev)
        labels = torch.empty((b, n_in), dtype=torch.float32, device=dev)
        counts = torch.empty((b, 2), dtype=torch.int64, device=dev)
        fg_members = torch.empty((b, n_in), dtype=torch.int32, device=dev)
        bg_members = torch.empty((b, n_in), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().ait_anchor_classify(
                _lib.dev_ptr(anchors), n_in, _lib.dev_ptr(gt), b, G, gt.size(2), float(cfg.TRAIN.RPN_NEGATIVE_OVERLAP),
                float(cfg.TRAIN.RPN_POSITIVE_OVERLAP), int(bool(cfg.TRAIN.RPN_CLOBBER_POSITIVES)), _lib.dev_ptr(max_ov),
                _lib.dev_ptr(argmax, torch.int64), _lib.dev_ptr(gt_max, torch.int32), _lib.dev_ptr(labels),
                _lib.dev_ptr(counts, torch.int64), _lib.dev_ptr(fg_members, torch.int32),
                _lib.dev_ptr(bg_members, torch.int32), _lib.cur_stream(dev))
        _lib.check(rc, "ait_anchor_classify")
        host, ev = self._to_host(counts, mismatch)
        return ((H, W, gt_boxes.data_ptr()), gt, inds_inside, anchors, argmax, labels, host, ev, all_anchors.size(0),
                (inside_pos, fg_members, bg_members), redo)

    def begin(self, gt_boxes, im_info, H, W, im_hw_hint=None):
        """called by the detector before the backbone is enqueued (see prepare)"""
        self._pending = self.prepare(gt_boxes, im_info, H, W, im_hw_hint)

    def finish(self, pending):
        (H, W, _), gt_boxes, inds_inside, anchors, argmax_ov, labels, host, ev, total, hip, redo = pending
        b, n_in = labels.shape
        A = self._num_anchors
        dev = gt_boxes.device
        # ---- subsampling: the reference's NumPy global-RNG call sequence (index parity), driven by the two
        # counts per image; WHICH anchors the drawn positions denote is resolved on the device ----------
        if ev is not None:
            ev.synchronize()
        flat = host.numpy()
        if flat.size > 2 * b and flat[2 * b] != 0:
            # the image-size prediction was wrong (im_info[0] is not the image tensor's size): redo the layer for
            # the real size, read back from the device.  Nothing random has been drawn yet.
            return self.finish(self.prepare(*redo, im_hw_hint=None))
        cnt = flat[:2 * b].reshape(b, 2)
        num_fg = int(cfg.TRAIN.RPN_FG_FRACTION * cfg.TRAIN.RPN_BATCHSIZE)
        dis_fg, dis_bg, n_fg_after, n_bg_after = [], [], [], []
        for i in range(b):
            n_f, n_b = int(cnt[i, 0]), int(cnt[i, 1])
            d = np.empty((0,), np.int64)
            if n_f > num_fg:
                d = np.random.permutation(n_f)[:n_f - num_fg]
            dis_fg.append(d)
            n_f = min(n_f, num_fg)
            num_bg = cfg.TRAIN.RPN_BATCHSIZE - n_f
            d = np.empty((0,), np.int64)
            if n_b > num_bg:
                d = np.random.permutation(n_b)[:n_b - num_bg]
            dis_bg.append(d)
            n_fg_after.append(n_f)
            n_bg_after.append(min(n_b, num_bg))
        # uniform example weighting; like the reference, the count comes from the LAST image
        assert cfg.TRAIN.RPN_POSITIVE_WEIGHT < 0
        num_examples = n_fg_after[b - 1] + n_bg_after[b - 1]

        if hip is not None:
            return self._finish_hip(hip, H, W, gt_boxes, anchors, argmax_ov, labels, dis_fg, dis_bg, num_examples)

        def disable(lab, want, drawn):
            """lab[i, (want-th class member list)[drawn[i]]] = -1: the k-th member of a class in ascending
            anchor order is found with a stable sort of the class mask (no data-dependent shapes)."""
            m = max(d.size for d in drawn)
            if m == 0:
                return lab
            pos = np.zeros((b, m), np.int64)
            ok = np.zeros((b, m), np.bool_)
            for i, d in enumerate(drawn):
                pos[i, :d.size] = d
                ok[i, :d.size] = True
            pos_t = torch.from_numpy(pos).to(dev, non_blocking=True)
            ok_t = torch.from_numpy(ok).to(dev, non_blocking=True)
            members = torch.sort((lab != want).to(torch.uint8), dim=1, stable=True)[1]      # class members first, ascending
            idx = torch.gather(members, 1, pos_t)
            # padding entries rewrite an already disabled slot: route them to column n_in (a scratch column)
            idx = torch.where(ok_t, idx, torch.full_like(idx, n_in))
            ext = torch.cat([lab, lab.new_zeros((b, 1))], 1)
            ext.scatter_(1, idx, -1.0)
            return ext[:, :n_in]

        labels = disable(labels, 1, dis_fg)
        labels = disable(labels, 0, dis_bg)

        gt_for_anchor = torch.gather(gt_boxes[:, :, :4], 1, argmax_ov.unsqueeze(2).expand(-1, -1, 4))
        targets = bbox_transform_batch(anchors, gt_for_anchor)            # [b, n_in, 4]
        inside_w = (labels == 1).to(gt_boxes.dtype) * cfg.TRAIN.RPN_BBOX_INSIDE_WEIGHTS[0]
        outside_w = (labels >= 0).to(gt_boxes.dtype) * (1.0 / num_examples)

        def unmap(x, fill):
            full = x.new_full((b, total) + tuple(x.shape[2:]), fill)
            full[:, inds_inside] = x
            return full

        labels = unmap(labels, -1).view(b, H, W, A).permute(0, 3, 1, 2).reshape(b, 1, A * H, W)
        targets = unmap(targets, 0).view(b, H, W, A * 4).permute(0, 3, 1, 2).contiguous()

        def spread(wt):
            return unmap(wt, 0).unsqueeze(2).expand(b, total, 4).reshape(b, H, W, 4 * A) \
                .permute(0, 3, 1, 2).contiguous()

        return [labels, targets, spread(inside_w), spread(outside_w)]


def _anchor_finish_hip(layer, hip, H, W, gt, anchors, argmax, labels, dis_fg, dis_bg, num_examples):
    """finish() behind the host's draws in two or three launches (ait_anchor_targets)"""
    inside_pos, fg_members, bg_members = hip
    dev = gt.device
    b, n_in = labels.shape
    A = layer._num_anchors

    def pack(drawn):
        m = max(d.size for d in drawn)
        pos = np.zeros((b, max(m, 1)), np.int64)
        cnt = np.zeros((b,), np.int32)
        for i, d in enumerate(drawn):
            pos[i, :d.size] = d
            cnt[i] = d.size
        return m, torch.from_numpy(pos).to(dev, non_blocking=True), torch.from_numpy(cnt).to(dev, non_blocking=True)

    m_fg, fg_pos, fg_cnt = pack(dis_fg)
    m_bg, bg_pos, bg_cnt = pack(dis_bg)
    out_l = torch.empty((b, 1, A * H, W), dtype=torch.float32, device=dev)
    out_t = torch.empty((b, 4 * A, H, W), dtype=torch.float32, device=dev)
    out_i = torch.empty_like(out_t)
    out_o = torch.empty_like(out_t)
    with torch.cuda.device(dev):
        rc = _lib.lib().ait_anchor_targets(
            _lib.dev_ptr(anchors), _lib.dev_ptr(inside_pos, torch.int32), n_in, A, H, W, _lib.dev_ptr(gt), b, gt.size(1),
            gt.size(2), _lib.dev_ptr(argmax, torch.int64), _lib.dev_ptr(labels),
            _lib.dev_ptr(fg_pos, torch.int64), _lib.dev_ptr(fg_cnt, torch.int32), m_fg, _lib.dev_ptr(fg_members, torch.int32),
            _lib.dev_ptr(bg_pos, torch.int64), _lib.dev_ptr(bg_cnt, torch.int32), m_bg, _lib.dev_ptr(bg_members, torch.int32),
            float(cfg.TRAIN.RPN_BBOX_INSIDE_WEIGHTS[0]), float(1.0 / num_examples), _lib.dev_ptr(out_l), _lib.dev_ptr(out_t),
            _lib.dev_ptr(out_i), _lib.dev_ptr(out_o), _lib.cur_stream(dev))
    _lib.check(rc, "ait_anchor_targets")
    return [out_l, out_t, out_i, out_o]


_AnchorTargetLayer._finish_hip = _anchor_finish_hip


class _ProposalTargetLayer(nn.Module):
    """(all_rois [b,R,5], gt_boxes [b,G,5], num_boxes) -> rois [b,P,5], labels [b,P],
    bbox_targets [b,P,4], inside weights, outside weights with P = cfg.TRAIN.BATCH_SIZE."""

    def __init__(self, nclasses):
        super().__init__()
        self._num_classes = nclasses

    def forward(self, all_rois, gt_boxes, num_boxes):
        dev, dt = gt_boxes.device, gt_boxes.dtype
        b = gt_boxes.size(0)
        P = int(cfg.TRAIN.BATCH_SIZE)
        fg_per_image = int(np.round(cfg.TRAIN.FG_FRACTION * P)) or 1
        gpu = dev.type == "cuda"
        hip = _lib_kernels(gt_boxes, "proposal_target_layer")
        if hip and all_rois.dtype != torch.float32:
            raise _lib.AitHipError("proposal_target_layer: float32 RoIs expected on the GPU, got %s" % all_rois.dtype)
        if hip:
            all_rois, assign, labels, counts, fg_members, bg_members = self._classify_hip(all_rois, gt_boxes)
        else:
            all_rois, assign, labels, counts, fg_members, bg_members = self._classify(all_rois, gt_boxes)

        # ---- sampling: the host needs only the two class sizes per image to make the reference's RNG calls
        # (index parity); which RoIs the drawn positions denote is resolved on the device -------------
        if gpu:
            host = torch.empty((b, 2), dtype=torch.int64, pin_memory=True)
            host.copy_(counts, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            ev.synchronize()
            cnt = host.numpy()
        else:
            cnt = counts.numpy()
        pos = np.zeros((b, P), np.int64)
        n_fg = np.zeros((b,), np.int64)
        for i in range(b):
            n_f, n_b = int(cnt[i, 0]), int(cnt[i, 1])
            if n_f > 0 and n_b > 0:
                k = min(fg_per_image, n_f)
                pos[i, :k] = np.random.permutation(n_f)[:k]
                pos[i, k:] = np.floor(np.random.rand(P - k) * n_b).astype(np.int64)
            elif n_f > 0:
                pos[i] = np.floor(np.random.rand(P) * n_f).astype(np.int64)
                k = P
            elif n_b > 0:
                pos[i] = np.floor(np.random.rand(P) * n_b).astype(np.int64)
                k = 0
            else:
                raise ValueError("bg_num_rois = 0 and fg_num_rois = 0, this should not happen!")
            n_fg[i] = k
        pos_t = torch.from_numpy(pos).to(dev, non_blocking=True)
        n_fg_t = torch.from_numpy(n_fg).to(dev, non_blocking=True)
        if hip:
            return self._gather_hip(pos_t, n_fg_t, fg_members, bg_members, labels, all_rois, assign, gt_boxes)
        return self._gather(pos_t, n_fg_t, fg_members, bg_members, labels, all_rois, assign, gt_boxes)

    def _classify_hip(self, rois, gt_boxes):
        """_classify in one launch (ait_roi_classify)"""
        dev = gt_boxes.device
        rois, gt = rois.contiguous(), gt_boxes.contiguous()
        b, R0, G = gt.size(0), rois.size(1), gt.size(1)
        R = R0 + G
        all_rois = torch.empty((b, R, 5), dtype=torch.float32, device=dev)
        assign = torch.empty((b, R), dtype=torch.int64, device=dev)
        labels = torch.empty((b, R), dtype=torch.float32, device=dev)
        counts = torch.empty((b, 2), dtype=torch.int64, device=dev)
        fg_members = torch.empty((b, R), dtype=torch.int64, device=dev)
        bg_members = torch.empty((b, R), dtype=torch.int64, device=dev)
        L = _lib.lib()
        ws_bytes = L.ait_roi_classify_workspace_bytes(b, R0, G)
        ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = L.ait_roi_classify(_lib.dev_ptr(rois), b, R0, _lib.dev_ptr(gt), G, gt.size(2),
                                    float(cfg.TRAIN.FG_THRESH), float(cfg.TRAIN.BG_THRESH_HI), float(cfg.TRAIN.BG_THRESH_LO),
                                    _lib.dev_ptr(ws, torch.uint8), ws_bytes, _lib.dev_ptr(all_rois),
                                    _lib.dev_ptr(assign, torch.int64), _lib.dev_ptr(labels), _lib.dev_ptr(counts, torch.int64),
                                    _lib.dev_ptr(fg_members, torch.int64), _lib.dev_ptr(bg_members, torch.int64),
                                    _lib.cur_stream(dev))
        _lib.check(rc, "ait_roi_classify")
        return all_rois, assign, labels, counts, fg_members, bg_members

    def _gather_hip(self, pos_t, n_fg_t, fg_members, bg_members, labels, all_rois, assign, gt_boxes):
        """_gather in one launch (ait_roi_sample_gather)"""
        import ctypes
        dev = gt_boxes.device
        gt = gt_boxes.contiguous()
        b, P = pos_t.shape
        R, G = all_rois.size(1), gt.size(1)
        rois_b = torch.empty((b, P, 5), dtype=torch.float32, device=dev)
        labels_b = torch.empty((b, P), dtype=torch.float32, device=dev)
        targets = torch.empty((b, P, 4), dtype=torch.float32, device=dev)
        inside_w = torch.empty((b, P, 4), dtype=torch.float32, device=dev)
        outside_w = torch.empty((b, P, 4), dtype=torch.float32, device=dev)
        f4 = ctypes.c_float * 4
        with torch.cuda.device(dev):
            rc = _lib.lib().ait_roi_sample_gather(
                _lib.dev_ptr(pos_t.contiguous(), torch.int64), _lib.dev_ptr(n_fg_t.contiguous(), torch.int64), b, P, R,
                _lib.dev_ptr(fg_members, torch.int64), _lib.dev_ptr(bg_members, torch.int64), _lib.dev_ptr(labels),
                _lib.dev_ptr(all_rois), _lib.dev_ptr(assign, torch.int64), _lib.dev_ptr(gt), G, gt.size(2),
                f4(*[float(v) for v in cfg.TRAIN.BBOX_NORMALIZE_MEANS]), f4(*[float(v) for v in cfg.TRAIN.BBOX_NORMALIZE_STDS]),
                f4(*[float(v) for v in cfg.TRAIN.BBOX_INSIDE_WEIGHTS]), int(bool(cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED)),
                _lib.dev_ptr(rois_b), _lib.dev_ptr(labels_b), _lib.dev_ptr(targets), _lib.dev_ptr(inside_w),
                _lib.dev_ptr(outside_w), _lib.cur_stream(dev))
        _lib.check(rc, "ait_roi_sample_gather")
        return rois_b, labels_b, targets, inside_w, outside_w

    def _classify(self, all_rois, gt_boxes):
        """RoIs (+ the gt boxes as RoIs) -> best gt per RoI, its class, the fg / bg class sizes per image and
        the class member lists in ascending RoI order (stable sort of the class mask: no data-dependent shape)"""
        gt_as_rois = torch.zeros_like(gt_boxes)
        gt_as_rois[:, :, 1:5] = gt_boxes[:, :, :4]
        all_rois = torch.cat([all_rois, gt_as_rois], 1)
        overlaps = bbox_overlaps_batch(all_rois, gt_boxes)
        max_ov, assign = overlaps.max(2)
        labels = torch.gather(gt_boxes[:, :, 4], 1, assign)
        fg_mask = max_ov >= cfg.TRAIN.FG_THRESH
        bg_mask = (max_ov < cfg.TRAIN.BG_THRESH_HI) & (max_ov >= cfg.TRAIN.BG_THRESH_LO)
        counts = torch.stack((fg_mask.sum(1), bg_mask.sum(1)), 1)
        fg_members = torch.sort((~fg_mask).to(torch.uint8), dim=1, stable=True)[1]
        bg_members = torch.sort((~bg_mask).to(torch.uint8), dim=1, stable=True)[1]
        return all_rois, assign, labels, counts, fg_members, bg_members

    def _gather(self, pos_t, n_fg_t, fg_members, bg_members, labels, all_rois, assign, gt_boxes):
        dev, dt = gt_boxes.device, gt_boxes.dtype
        b, P = pos_t.shape
        is_fg = torch.arange(P, device=dev).unsqueeze(0) < n_fg_t.unsqueeze(1)
        keep_t = torch.where(is_fg, torch.gather(fg_members, 1, pos_t), torch.gather(bg_members, 1, pos_t))

        labels_b = torch.gather(labels, 1, keep_t) * is_fg.to(dt)
        rois_b = torch.gather(all_rois, 1, keep_t.unsqueeze(2).expand(-1, -1, 5)).clone()
        rois_b[:, :, 0] = torch.arange(b, device=dev, dtype=dt).unsqueeze(1)
        gt_idx = torch.gather(assign, 1, keep_t)
        gt_b = torch.gather(gt_boxes, 1, gt_idx.unsqueeze(2).expand(-1, -1, gt_boxes.size(2)))

        targets = bbox_transform_batch(rois_b[:, :, 1:5], gt_b[:, :, :4])
        if cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED:
            targets = (targets - _const(cfg.TRAIN.BBOX_NORMALIZE_MEANS, dev, dt)) / _const(cfg.TRAIN.BBOX_NORMALIZE_STDS, dev, dt)
        pos = (labels_b > 0).unsqueeze(2).to(dt)
        # an image whose sampled labels sum to zero gets no regression targets at all
        pos = pos * (labels_b.sum(1) != 0).view(b, 1, 1).to(dt)
        # (assigned at the foreground rows only, as proposal_target_layer_cascade.py:101-107: a select, so that a
        # non-finite target of a degenerate box cannot leak as NaN * 0)
        bbox_targets = torch.where(pos > 0, targets, torch.zeros_like(targets))
        inside_w = pos * _const(cfg.TRAIN.BBOX_INSIDE_WEIGHTS, dev, dt)
        outside_w = (inside_w > 0).to(dt)
        return rois_b, labels_b, bbox_targets, inside_w, outside_w


_CONSTS = {}


def _const(values, dev, dt):
    """small config vectors as device tensors, uploaded once (an upload inside a graph capture is not allowed)"""
    key = (tuple(float(v) for v in values), str(dev), dt)
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(key[0], device=dev, dtype=dt)
    return t


class _Conv3x3BiasRelu(torch.autograd.Function):
    """relu(conv3x3(x) + bias), stride 1, pad 1, on a channels-last map of ANY size as the library's implicit GEMM
    (ait_conv_fwd_f32 / ait_conv_bwd_data_f32 / ait_conv_bwd_weight_f32: f32 products on the bf16 matrix pipe, bias +
    ReLU in the forward epilogue).  The RPN's 1024 -> 512 convolution (lib/model/rpn/rpn.py:32,53) is 270 GFLOP per
    bench step forward + backward: the largest single convolution outside the proposal tail."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        n, cin, h, w = x.shape
        cout = weight.shape[0]
        xm = x.permute(0, 2, 3, 1).reshape(n * h * w, cin)                   # view of channels-last x
        wm = weight.permute(0, 2, 3, 1)                                      # [cout, kh, kw, cin]: channels-last memory
        if not wm.is_contiguous():
            wm = wm.contiguous()
        geom = ops.conv_geom(n, (h, w), (h, w), (3, 3), 1, 1)
        ym = ops.conv_fwd(xm, wm, geom, bias=bias, relu=True)
        y = ym.view(n, h, w, cout).permute(0, 3, 1, 2)
        ctx.save_for_backward(xm, wm, y)
        ctx.geom, ctx.xshape = geom, (n, cin, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        xm, wm, y = ctx.saved_tensors
        n, cin, h, w = ctx.xshape
        cout = wm.shape[0]
        dz = torch.where(y > 0, dy, torch.zeros((), dtype=dy.dtype, device=dy.device))
        dzm = dz.permute(0, 2, 3, 1).reshape(n * h * w, cout)
        if not dzm.is_contiguous():
            dzm = dzm.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv_bwd_data(dzm, wm, ctx.geom).view(n, h, w, cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            dw = ops.conv_bwd_weight(dzm, xm, ctx.geom, 3, 3, split_k=8).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[2]:
            db = ops.colsum(dzm)
        return dx, dw, db


class _Heads1x1(torch.autograd.Function):
    """The RPN's two 1x1 heads (lib/model/rpn/rpn.py:34-43: 512 -> 2A objectness scores, 512 -> 4A box deltas) as ONE
    product on the library's GEMM: the two weight matrices stacked and padded to a multiple of 64 rows (64 for 9 anchors,
    128 for the COCO variant's 12), y [tokens, 64 | 128] = x W^T + b on the token rows of the channels-last feature; the two
    results are its column ranges.  Backward: the stacked gradient -> dx = dy W, dW = dy^T x (split-K), db = column sums.  (MIOpen runs each head as its own rocBLAS call
    with N = 18 / 36.)"""

    @staticmethod
    def forward(ctx, xm, w1, b1, w2, b2):
        n1, n2, k = w1.shape[0], w2.shape[0], xm.shape[1]
        npad = (n1 + n2 + 63) // 64 * 64
        W = torch.zeros((npad, k), dtype=torch.float32, device=xm.device)
        W[:n1] = w1.reshape(n1, k)
        W[n1:n1 + n2] = w2.reshape(n2, k)
        bias = torch.zeros((npad,), dtype=torch.float32, device=xm.device)
        bias[:n1] = b1
        bias[n1:n1 + n2] = b2
        y = ops.gemm(xm, W, bias=bias)
        ctx.save_for_backward(xm, W)
        ctx.sizes = (n1, n2, tuple(w1.shape), tuple(w2.shape))
        return y[:, :n1], y[:, n1:n1 + n2]

    @staticmethod
    def backward(ctx, d1, d2):
        xm, W = ctx.saved_tensors
        n1, n2, s1, s2 = ctx.sizes
        dy = torch.zeros((xm.shape[0], W.shape[0]), dtype=torch.float32, device=xm.device)
        dy[:, :n1] = d1
        dy[:, n1:n1 + n2] = d2
        dx = ops.gemm(dy, W, trans_b=False) if ctx.needs_input_grad[0] else None
        # (16 K-ranges: 22 us against 39 with 8 at 9576 rows, 42 against 77 at 19152; 32 and more lose again --
        # profiles/r06_gemm_tail_experiments.txt)
        dW = ops.gemm(dy, xm, trans_a=True, trans_b=False, split_k=16 if xm.shape[0] >= 2048 else 1)
        db = ops.colsum(dy)
        return dx, dW[:n1].reshape(s1), db[:n1], dW[n1:n1 + n2].reshape(s2), db[n1:n1 + n2]


_RPN_CONV_KERNEL = True     # test hook: False = PyTorch-ROCm's convolution + ReLU
_RPN_HEADS_KERNEL = True    # test hook: False = the two 1x1 heads as PyTorch-ROCm convolutions


class _RPN(nn.Module):
    """Region proposal network head: 3x3 conv -> {2A objectness, 4A box deltas}."""

    def __init__(self, din):
        super().__init__()
        self.din = din
        self.anchor_scales = cfg.ANCHOR_SCALES
        self.anchor_ratios = cfg.ANCHOR_RATIOS
        self.feat_stride = cfg.FEAT_STRIDE[0]
        A = len(self.anchor_scales) * len(self.anchor_ratios)
        self.RPN_Conv = nn.Conv2d(self.din, 512, 3, 1, 1, bias=True)
        self.nc_score_out = A * 2
        self.RPN_cls_score = nn.Conv2d(512, self.nc_score_out, 1, 1, 0)
        self.nc_bbox_out = A * 4
        self.RPN_bbox_pred = nn.Conv2d(512, self.nc_bbox_out, 1, 1, 0)
        self.RPN_proposal = _ProposalLayer(self.feat_stride, self.anchor_scales, self.anchor_ratios)
        self.RPN_anchor_target = _AnchorTargetLayer(self.feat_stride, self.anchor_scales, self.anchor_ratios)
        self.rpn_loss_cls = 0
        self.rpn_loss_box = 0

    @staticmethod
    def reshape(x, d):
        s = x.size()
        return x.contiguous().view(s[0], int(d), int(float(s[1] * s[2]) / float(d)), s[3])

    def forward(self, base_feat, im_info, gt_boxes, num_boxes):
        b = base_feat.size(0)
        c = self.RPN_Conv
        if _RPN_CONV_KERNEL and base_feat.is_cuda and base_feat.dtype == torch.float32:
            if (c.in_channels % 128 == 0 and c.out_channels % 16 == 0
                    and base_feat.shape[0] * base_feat.shape[2] * base_feat.shape[3] >= 16):
                if not base_feat.is_contiguous(memory_format=torch.channels_last):
                    base_feat = base_feat.contiguous(memory_format=torch.channels_last)    # (an NCHW caller: one re-layout, same kernel)
                conv = _Conv3x3BiasRelu.apply(base_feat, c.weight, c.bias)
            else:
                # a head the implicit-GEMM kernel does not take: PyTorch-ROCm's convolution, COUNTED (bench.py refuses
                # to print a line if any stand-in ran; DESIGN.md 1: on a GPU there is no silent fallback)
                ops.note_fallback("rpn.RPN_Conv", base_feat)
                conv = F.relu(c(base_feat), inplace=True)
        else:
            conv = F.relu(c(base_feat), inplace=True)
        n_out = self.nc_score_out + self.nc_bbox_out
        if (_RPN_HEADS_KERNEL and conv.is_cuda and conv.dtype == torch.float32 and n_out <= 256 and conv.shape[1] % 16 == 0
                and conv.is_contiguous(memory_format=torch.channels_last)):
            n, ch, fh, fw = conv.shape
            ys, yb = _Heads1x1.apply(conv.permute(0, 2, 3, 1).reshape(n * fh * fw, ch), self.RPN_cls_score.weight,
                                     self.RPN_cls_score.bias, self.RPN_bbox_pred.weight, self.RPN_bbox_pred.bias)
            # (the proposal layer's decode kernel and the losses read the reference's NCHW layout)
            cls_score = ys.reshape(n, fh, fw, self.nc_score_out).permute(0, 3, 1, 2).contiguous()
            bbox_pred = yb.reshape(n, fh, fw, self.nc_bbox_out).permute(0, 3, 1, 2).contiguous()
        else:
            if conv.is_cuda and _RPN_HEADS_KERNEL:       # heads the stacked product does not take: COUNTED
                ops.note_fallback("rpn.heads", conv)
            cls_score = self.RPN_cls_score(conv)
            bbox_pred = self.RPN_bbox_pred(conv)
        score_2 = self.reshape(cls_score, 2)
        cls_prob = self.reshape(F.softmax(score_2, 1), self.nc_score_out)
        rois = self.RPN_proposal((cls_prob.data, bbox_pred.data, im_info,
                                  'TRAIN' if self.training else 'TEST'))
        self.rpn_loss_cls = 0
        self.rpn_loss_box = 0
        if self.training:
            assert gt_boxes is not None
            labels, targets, w_in, w_out = self.RPN_anchor_target((cls_score.data, gt_boxes, im_info, num_boxes))
            logits = score_2.permute(0, 2, 3, 1).reshape(-1, 2)
            labels = labels.view(-1)
            # (the reference selects the labelled anchors with nonzero() + index_select, a host synchronisation;
            # the mean over the same anchors without one)
            self.rpn_loss_cls = F.cross_entropy(logits, labels.long(), ignore_index=-1)
            self.rpn_loss_box = _smooth_l1_loss(bbox_pred, targets, w_in, w_out, sigma=3, dim=[1, 2, 3])
        return rois, self.rpn_loss_cls, self.rpn_loss_box
