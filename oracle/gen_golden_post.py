"""oracle/gen_golden_post.py -- TEST INFRASTRUCTURE ONLY; runs only where /root/reference is.

g14: the detection post-processing of the reference's evaluation drivers (test_net_coco.py:381-449, the same
block in test_net_voc.py) recorded from the imported, unmodified reference: its own bbox_transform_inv /
clip_boxes (lib/model/rpn/bbox_transform.py:77-133) and its own compiled NMS (model._C.nms ->
lib/model/csrc/cpu/nms_cpu.cpp) are CALLED in the order the driver calls them; the statements between those
calls (de-normalisation, rescale, threshold, sort, top max_per_image) are the driver's own torch / numpy
expressions on the same operands.  The driver's block is not a function (it sits inline in its main loop), so
this script is the call sequence, not a copy of it; cfg values come from the reference's cfgs/res50.yml.

    python -m oracle.gen_golden_post

Inputs (regenerated identically by the test): the reference's eval outputs of golden g9 (rois, cls_prob,
bbox_pred) with the regression scaled x300 and distinct seeded scores (random-weight probabilities are tied to
~1e-7), and a second, fully seeded case of 300 boxes that exercises clipping, the score threshold and the
max_per_image cut.
"""
import os

import numpy as np
import torch

from . import ref_import
from .gen_golden import _save

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def case_a():
    g = np.load(os.path.join(GOLD, "g9_detector_eval.npz"))
    rois, prob, bbox = (torch.from_numpy(g[k]) for k in ("rois", "cls_prob", "bbox_pred"))
    bbox = bbox * 300.0
    n = prob.numel()
    prob = torch.from_numpy(np.random.RandomState(5).permutation(n).astype(np.float32) / n * 0.9 + 0.05).view_as(prob)
    return rois, prob, bbox, torch.tensor([[600.0, 1000.0, 1.6]]), 0.0, 100


def case_b():
    rs = np.random.RandomState(77)
    n = 300
    cx, cy = rs.uniform(0, 1000, n), rs.uniform(0, 600, n)
    w, h = rs.uniform(20, 500, n), rs.uniform(20, 400, n)
    boxes = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)          # many reach outside the image
    rois = np.concatenate([np.zeros((n, 1)), boxes], 1).astype(np.float32)[None]
    bbox = rs.normal(0, 1.5, (1, n, 4)).astype(np.float32)
    prob = (rs.permutation(n).astype(np.float32) / n).reshape(1, n, 1)
    return torch.from_numpy(rois), torch.from_numpy(prob), torch.from_numpy(bbox), torch.tensor([[600.0, 1000.0, 2.0]]), 0.05, 100


def reference_postprocess(cfg, rois, cls_prob, bbox_pred, im_info, thresh, max_per_image):
    from model.roi_layers import nms
    from model.rpn.bbox_transform import bbox_transform_inv, clip_boxes
    scores = cls_prob.data
    boxes = rois.data[:, :, 1:5]
    box_deltas = bbox_pred.data
    box_deltas = box_deltas.view(-1, 4) * torch.FloatTensor(cfg.TRAIN.BBOX_NORMALIZE_STDS) \
        + torch.FloatTensor(cfg.TRAIN.BBOX_NORMALIZE_MEANS)
    pred_boxes = clip_boxes(bbox_transform_inv(boxes, box_deltas.view(1, -1, 4), 1), im_info.data, 1)
    pred_boxes = pred_boxes / im_info[0][2].item()
    scores, pred_boxes = scores.squeeze(), pred_boxes.squeeze()
    inds = torch.nonzero(scores > thresh).view(-1)
    cls_scores, cls_boxes = scores[inds], pred_boxes[inds, :]
    cls_dets = torch.cat((cls_boxes, cls_scores.unsqueeze(1)), 1)
    _, order = torch.sort(cls_scores, 0, True)
    cls_dets = cls_dets[order]
    keep = nms(cls_boxes[order, :], cls_scores[order], cfg.TEST.NMS)
    dets = cls_dets[keep.view(-1).long()].numpy()
    if max_per_image > 0 and len(dets) > max_per_image:
        image_thresh = np.sort(dets[:, -1])[-max_per_image]
        dets = dets[np.where(dets[:, -1] >= image_thresh)[0], :]
    return dets, keep.numpy()


def main():
    ref_import.setup()
    from model.utils.config import cfg, cfg_from_file
    cfg_from_file(os.path.join(ref_import.REF, "cfgs", "res50.yml"))
    assert cfg.TEST.BBOX_REG and cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED
    out = {"nms_thr": np.float32(cfg.TEST.NMS)}
    for name, case in (("a", case_a), ("b", case_b)):
        rois, prob, bbox, info, thresh, mpi = case()
        dets, keep = reference_postprocess(cfg, rois, prob, bbox, info, thresh, mpi)
        out[name + "_dets"] = dets.astype(np.float32)
        out[name + "_keep"] = keep.astype(np.int64)
        print("g14 case %s: %d boxes -> %d after NMS -> %d detections" % (name, prob.numel(), keep.size, dets.shape[0]))
    _save("g14_postprocess", out)


if __name__ == "__main__":
    main()
