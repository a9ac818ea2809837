"""Detection post-processing of the reference's evaluation drivers (§8f rank 4), on the device:
test_net_coco.py:381-449 / test_net_voc.py (identical block): de-normalise the class-agnostic box
deltas, decode + clip, rescale to the original image, score threshold, sort, NMS (cfg.TEST.NMS,
libait_hip.so), keep the max_per_image best.

    dets = detections(rois, cls_prob, bbox_pred, im_info, scale)     # [K, 5] = x1,y1,x2,y2,score
"""
import torch

from .config import cfg
from .roi_layers import nms
from .rpn import bbox_transform_inv, clip_boxes


def detections(rois, cls_prob, bbox_pred, im_info, im_scale, thresh=0.0, max_per_image=100):
    """One image (batch dimension 1, as the reference's DataLoader(batch_size=1) feeds it).
    Returns a [K,5] device tensor sorted by descending score."""
    scores = cls_prob.data
    boxes = rois.data[:, :, 1:5]
    deltas = bbox_pred.data
    dev, dt = deltas.device, deltas.dtype
    if cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED:
        deltas = deltas.view(-1, 4) * torch.tensor(cfg.TRAIN.BBOX_NORMALIZE_STDS, device=dev, dtype=dt) \
            + torch.tensor(cfg.TRAIN.BBOX_NORMALIZE_MEANS, device=dev, dtype=dt)
        deltas = deltas.view(1, -1, 4)
    pred = clip_boxes(bbox_transform_inv(boxes, deltas, 1), im_info.data, 1)
    pred = (pred / float(im_scale)).squeeze(0)
    scores = scores.reshape(-1)
    inds = torch.nonzero(scores > thresh).view(-1)
    if inds.numel() == 0:
        return pred.new_zeros((0, 5))
    cls_scores, cls_boxes = scores[inds], pred[inds]
    order = torch.sort(cls_scores, 0, True)[1]
    dets = torch.cat((cls_boxes, cls_scores.unsqueeze(1)), 1)[order]
    keep = nms(cls_boxes[order], cls_scores[order], cfg.TEST.NMS)
    dets = dets[keep.view(-1).long()]
    if max_per_image > 0 and dets.size(0) > max_per_image:
        # the reference keeps every detection whose score reaches the max_per_image-th best
        kth = torch.sort(dets[:, 4])[0][-max_per_image]
        dets = dets[dets[:, 4] >= kth]
    return dets
