"""oracle/gen_golden_detector.py -- TEST INFRASTRUCTURE ONLY; runs only where /root/reference is.

Detector-level golden vectors recorded by running the imported, unmodified
reference (VOC variant, cfgs/res50.yml) on seeded inputs and the deterministic weights of
oracle/detector_ref.make_detector_state_dict:

  g7   _ProposalLayer rois for seeded scores/deltas, TRAIN and TEST keys
  g8   _AnchorTargetLayer / _ProposalTargetLayer outputs under np.random.seed(3)
  g9   whole-model eval forward, BASELINE cfg1 (1 pair, 600x1000 target, 128 proposals):
       rois, cls_prob, bbox_pred, similarity logits `score`, stage features
  g10  whole-model train forward losses at P = 128 and P = 300 (np.random.seed(3))
"""
import contextlib
import io
import os

import numpy as np
import torch

from . import cases, detector_ref, ref_import
from .digest import pack, seeded
from .gen_golden import _save

REF_CFG = os.path.join(ref_import.REF, "cfgs", "res50.yml")


def _cfg():
    ref_import.setup()
    from model.utils.config import cfg, cfg_from_file
    cfg_from_file(REF_CFG)
    return cfg


rpn_case, gt_case = cases.rpn_case, cases.gt_case


def g7():
    cfg = _cfg()
    from model.rpn.proposal_layer import _ProposalLayer
    out = {}
    prob, deltas, info = rpn_case()
    layer = _ProposalLayer(16, cfg.ANCHOR_SCALES, cfg.ANCHOR_RATIOS)
    for key in ("TRAIN", "TEST"):
        rois = layer((torch.from_numpy(prob), torch.from_numpy(deltas), torch.from_numpy(info), key))
        out["rois_" + key] = rois.numpy()
    _save("g7_proposal_layer", out)


def g8():
    cfg = _cfg()
    from model.rpn.anchor_target_layer import _AnchorTargetLayer
    from model.rpn.proposal_target_layer_cascade import _ProposalTargetLayer
    out = {}
    prob, deltas, info = rpn_case()
    gt = gt_case()
    nb = torch.tensor([3, 3])
    np.random.seed(3)
    atl = _AnchorTargetLayer(16, cfg.ANCHOR_SCALES, cfg.ANCHOR_RATIOS)
    labels, targets, w_in, w_out = atl((torch.from_numpy(prob), torch.from_numpy(gt), torch.from_numpy(info), nb))
    out["atl_labels"] = labels.numpy().astype(np.int8)
    pack("atl_targets", targets, out)
    pack("atl_w_in", w_in, out)
    pack("atl_w_out", w_out, out)
    rois = torch.from_numpy(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                 "tests", "golden", "g7_proposal_layer.npz"))["rois_TRAIN"])
    for P in (128, 300):
        cfg.TRAIN.BATCH_SIZE = P
        ptl = _ProposalTargetLayer(2)
        r, lab, tg, wi, wo = ptl(rois, torch.from_numpy(gt), nb)
        out["ptl%d_rois" % P] = r.numpy()
        out["ptl%d_labels" % P] = lab.numpy()
        out["ptl%d_targets" % P] = tg.numpy()
        out["ptl%d_w_in" % P] = wi.numpy()
        out["ptl%d_w_out" % P] = wo.numpy()
    cfg.TRAIN.BATCH_SIZE = 128
    _save("g8_target_layers", out)


class _ContiguousOps:
    """The reference's CPU RoIAlign reads `input.data<T>()` as if it were contiguous NCHW
    (csrc/cpu/ROIAlign_cpu.cpp:242-253) while its CUDA path calls `.contiguous()` first
    (csrc/cuda/ROIAlign_cuda.cu:286,294).  In the detector the feature map handed to RoIAlign
    is a transposed VIEW (faster_rcnn_sys_transformer_sk_dilat.py:96-100), so the unwrapped CPU
    operator would pool from re-interpreted memory -- a defect of the CPU path, not behaviour
    the GPU-trained model ever had.  The goldens are therefore recorded with the CUDA path's
    semantics: inputs made contiguous, then the reference's own compiled operator."""

    def __init__(self, c):
        self._c = c
        self.nms = c.nms

    def roi_align_forward(self, input, rois, *a):
        return self._c.roi_align_forward(input.contiguous(), rois.contiguous(), *a)


def _ref_model():
    cfg = _cfg()
    import sys
    import model.roi_layers  # noqa: F401
    ra = sys.modules['model.roi_layers.roi_align']
    if not isinstance(ra._C, _ContiguousOps):
        ra._C = _ContiguousOps(ra._C)
    from model.faster_rcnn.resnet_sys_transformer_sk_dilat import resnet
    with contextlib.redirect_stdout(io.StringIO()):
        m = resnet(('__background__', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3)
        m.create_architecture()
    sd = detector_ref.make_detector_state_dict(9, detector_ref.reference_shapes())
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.startswith(("RCNN_base.stem.", "RCNN_base.layer")) for k in res.missing_keys), res.missing_keys
    return m, cfg


def g9():
    m, cfg = _ref_model()
    m.eval()
    cfg.TEST.RPN_POST_NMS_TOP_N = 128                  # BASELINE cfg1: 128 proposals
    im, qr, info, gt, nb = detector_ref.synth_inputs(1, 901)
    feats = {}
    hooks = [m.RCNN_cls_score.register_forward_hook(lambda mod, i, o: feats.__setitem__("score", o)),
             m.coattention.register_forward_hook(lambda mod, i, o: feats.__setitem__("co", o)),
             m.transformer.register_forward_hook(lambda mod, i, o: feats.__setitem__("ait", o)),
             m.RCNN_roi_align.register_forward_hook(lambda mod, i, o: feats.__setitem__("props", o))]
    with torch.no_grad():
        rois, cls_prob, bbox_pred, *_ = m(im, qr, info, gt, nb)
    for h in hooks:
        h.remove()
    cfg.TEST.RPN_POST_NMS_TOP_N = 300
    out = {"rois": rois.numpy(), "cls_prob": cls_prob.numpy(), "bbox_pred": bbox_pred.numpy(),
           "score": feats["score"].numpy()}
    pack("non_img", feats["co"][0], out)
    pack("non_qry", feats["co"][1], out)
    pack("props", feats["props"], out)
    pack("ait_out", feats["ait"], out)
    _save("g9_detector_eval", out)


def g10():
    m, cfg = _ref_model()
    m.train()
    for mod in m.modules():                                # parity is defined at dropout p = 0
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    out = {}
    for P in (128, 300):
        cfg.TRAIN.BATCH_SIZE = P
        np.random.seed(3)
        im, qr, info, gt, nb = detector_ref.synth_inputs(1, 1001)
        with torch.no_grad():
            res = m(im, qr, info, gt, nb)
        out["P%d_rois" % P] = res[0].numpy()
        out["P%d_labels" % P] = res[8].numpy()
        out["P%d_losses" % P] = np.array([float(x) for x in res[3:8]], np.float64)
        out["P%d_cls_prob" % P] = res[1].numpy()
    cfg.TRAIN.BATCH_SIZE = 128
    _save("g10_detector_train", out)


def g11():
    """COCO variant (faster_rcnn_coatt_transformer_sk.py): non-local co-attention, A = 12 anchors,
    50 GT slots; eval forward (1 pair, 128 proposals) and one train forward (P = 128)."""
    cfg = _cfg()
    import sys
    import model.roi_layers  # noqa: F401
    ra = sys.modules['model.roi_layers.roi_align']
    if not isinstance(ra._C, _ContiguousOps):
        ra._C = _ContiguousOps(ra._C)
    saved = (list(cfg.ANCHOR_SCALES), cfg.MAX_NUM_GT_BOXES)
    cfg.ANCHOR_SCALES = [4, 8, 16, 32]
    cfg.MAX_NUM_GT_BOXES = 50
    from model.faster_rcnn.resnet_coatt_transformer_sk import resnet
    with contextlib.redirect_stdout(io.StringIO()):
        m = resnet(('__background__', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3)
        m.create_architecture()
    sd = detector_ref.make_detector_state_dict(11, detector_ref.reference_shapes(A=12, variant="coco"))
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.startswith(("RCNN_base.stem.", "RCNN_base.layer")) for k in res.missing_keys), res.missing_keys
    out = {}
    m.eval()
    cfg.TEST.RPN_POST_NMS_TOP_N = 128
    im, qr, info, gt, nb = detector_ref.synth_inputs(1, 1101, max_gt=50)
    feats = {}
    h = [m.RCNN_cls_score.register_forward_hook(lambda mod, i, o: feats.__setitem__("score", o)),
         m.coattention_module.register_forward_hook(lambda mod, i, o: feats.__setitem__("co", o))]
    with torch.no_grad():
        rois, cls_prob, bbox_pred, *_ = m(im, qr, info, gt, nb)
    for x in h:
        x.remove()
    cfg.TEST.RPN_POST_NMS_TOP_N = 300
    out.update({"rois": rois.numpy(), "cls_prob": cls_prob.numpy(), "bbox_pred": bbox_pred.numpy(),
                "score": feats["score"].numpy()})
    pack("non_img", feats["co"][0], out)
    pack("non_qry", feats["co"][1], out)
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    cfg.TRAIN.BATCH_SIZE = 128
    np.random.seed(3)
    with torch.no_grad():
        res = m(im, qr, info, gt, nb)
    out["train_rois"] = res[0].numpy()
    out["train_labels"] = res[8].numpy()
    out["train_losses"] = np.array([float(x) for x in res[3:8]], np.float64)
    cfg.ANCHOR_SCALES, cfg.MAX_NUM_GT_BOXES = saved
    _save("g11_detector_coco", out)


def g12():
    """What the reference's proposal layer hands to its NMS and what comes back (proposal_layer.py:
    134-157): the decoded, clipped, score-sorted candidate boxes of every image and the kept indices
    (first post_nms_topN, -1 padded).  Recorded by wrapping the module-level `nms` name."""
    cfg = _cfg()
    import model.rpn.proposal_layer as PL
    out = {}
    prob, deltas, info = rpn_case()
    layer = PL._ProposalLayer(16, cfg.ANCHOR_SCALES, cfg.ANCHOR_RATIOS)
    real = PL.nms
    for key in ("TRAIN", "TEST"):
        rec = []

        def spy(boxes, scores, thr, _rec=rec):
            keep = real(boxes, scores, thr)
            _rec.append((boxes.clone().numpy(), keep.clone().numpy().reshape(-1), float(thr)))
            return keep
        PL.nms = spy
        try:
            layer((torch.from_numpy(prob), torch.from_numpy(deltas), torch.from_numpy(info), key))
        finally:
            PL.nms = real
        post_n = cfg[key].RPN_POST_NMS_TOP_N
        out["cand_" + key] = np.stack([r[0] for r in rec]).astype(np.float32)
        keep = np.full((len(rec), post_n), -1, np.int32)
        for i, r in enumerate(rec):
            k = r[1][:post_n]
            keep[i, :k.size] = k
        out["keep_" + key] = keep
        out["nkeep_" + key] = np.array([min(post_n, r[1].size) for r in rec], np.int32)
        out["thr_" + key] = np.float32(rec[0][2])
    _save("g12_proposal_nms", out)


def g13():
    """The proposal-layer outputs inside the g10 / g11 training forwards (seeds as there), so that
    the GPU tests can hand the product's sampler / RoIAlign / AIT / heads the reference's OWN
    proposals and compare labels and losses unconditionally."""
    out = {}
    m, cfg = _ref_model()
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    cfg.TRAIN.BATCH_SIZE = 128
    np.random.seed(3)
    ins = detector_ref.synth_inputs(1, 1001)
    got = {}
    h = m.RCNN_rpn.RPN_proposal.register_forward_hook(lambda mod, i, o: got.__setitem__("rois", o))
    with torch.no_grad():
        m(*ins)
    h.remove()
    out["voc_prop_rois"] = got["rois"].numpy()
    # COCO variant (as g11)
    import sys
    saved = (list(cfg.ANCHOR_SCALES), cfg.MAX_NUM_GT_BOXES)
    cfg.ANCHOR_SCALES = [4, 8, 16, 32]
    cfg.MAX_NUM_GT_BOXES = 50
    from model.faster_rcnn.resnet_coatt_transformer_sk import resnet
    with contextlib.redirect_stdout(io.StringIO()):
        mc = resnet(('__background__', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3)
        mc.create_architecture()
    sd = detector_ref.make_detector_state_dict(11, detector_ref.reference_shapes(A=12, variant="coco"))
    mc.load_state_dict(sd, strict=False)
    mc.train()
    for mod in mc.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    ins = detector_ref.synth_inputs(1, 1101, max_gt=50)
    np.random.seed(3)
    h = mc.RCNN_rpn.RPN_proposal.register_forward_hook(lambda mod, i, o: got.__setitem__("rois", o))
    with torch.no_grad():
        mc(*ins)
    h.remove()
    out["coco_prop_rois"] = got["rois"].numpy()
    cfg.ANCHOR_SCALES, cfg.MAX_NUM_GT_BOXES = saved
    _save("g13_detector_proposals", out)


GROUPS = {"g7": g7, "g8": g8, "g9": g9, "g10": g10, "g11": g11, "g12": g12, "g13": g13}
