"""GPU parity of the detector assembly (ait_amd.faster_rcnn / ait_amd.rpn over libait_hip.so)
against the reference's golden vectors g7..g10 and the CPU oracle."""
import contextlib
import warnings

import numpy as np
import pytest
import torch

from oracle import cases, detector_ref as D
from oracle.digest import compare

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(autouse=True)
def _reset_cfg():
    from ait_amd import config
    saved = (config.cfg.TRAIN.BATCH_SIZE, config.cfg.TEST.RPN_POST_NMS_TOP_N)
    yield
    config.cfg.TRAIN.BATCH_SIZE, config.cfg.TEST.RPN_POST_NMS_TOP_N = saved


def _rows_match(got, want, atol=2e-3):
    """fraction of RoI rows equal within atol (decode uses exp(): GPU vs CPU libm differ in the
    last ulp, which can move a box across an NMS threshold once in a while)"""
    return float((np.abs(got - want).max(-1) <= atol).mean())


@pytest.mark.parametrize("key", ["TRAIN", "TEST"])
def test_proposal_layer_vs_reference_golden(golden, key, record_property):
    from ait_amd.rpn import _ProposalLayer
    g = golden("g7_proposal_layer")
    prob, deltas, info = cases.rpn_case()
    layer = _ProposalLayer(16, [8, 16, 32], [0.5, 1, 2])
    rois = layer((_dev(prob), _dev(deltas), _dev(info), key)).cpu().numpy()
    want = g["rois_" + key]
    assert rois.shape == want.shape
    n_diff = int((np.abs(rois - want).max(-1) > 2e-3).sum())
    record_property("rows_differing_from_reference", n_diff)
    if n_diff:
        warnings.warn("proposal layer (%s): %d of %d rows differ from the reference (GPU exp() ulp moved a "
                      "box across the NMS threshold)" % (key, n_diff, want.shape[0] * want.shape[1]))
    assert _rows_match(rois, want) >= 0.995


@pytest.mark.parametrize("key", ["TRAIN", "TEST"])
def test_nms_on_reference_candidates_index_identity(golden, key):
    """The boxes the REFERENCE's proposal layer hands to its NMS (decoded, clipped, score-sorted:
    golden g12, proposal_layer.py:134-153) through ait_nms_batched: kept indices identical, image by
    image -- index identity through the layer without the GPU-vs-CPU exp() of the decode in between."""
    from ait_amd.roi_layers import nms_sorted, nms_sorted_batched
    g = golden("g12_proposal_nms")
    cand, want, n_want = g["cand_" + key], g["keep_" + key], g["nkeep_" + key]
    post_n = want.shape[1]
    keep, cnt = nms_sorted_batched(_dev(cand), float(g["thr_" + key]), post_n)
    for b in range(cand.shape[0]):
        c = int(cnt[b].item())
        assert c == int(n_want[b])
        assert np.array_equal(keep[b, :c].cpu().numpy(), want[b, :c].astype(np.int64))
        k1, c1 = nms_sorted(_dev(cand[b]), float(g["thr_" + key]), post_n)      # per-image entry point
        assert int(c1.item()) == c and np.array_equal(k1[:c].cpu().numpy(), want[b, :c].astype(np.int64))


def test_anchor_grid_bit_exact(golden):
    from ait_amd.rpn import _AnchorGrid, generate_anchors
    g = golden("g6_anchors")
    assert np.array_equal(generate_anchors(scales=np.array([8, 16, 32])), g["anchors_voc"])
    assert np.array_equal(generate_anchors(scales=np.array([4, 8, 16, 32])), g["anchors_coco"])
    for name, scales in (("voc", [8, 16, 32]), ("coco", [4, 8, 16, 32])):
        grid = _AnchorGrid(16, scales, [0.5, 1, 2]).get(cases.FEAT_H, cases.FEAT_W, torch.device("cuda")).cpu()
        assert np.array_equal(grid.double().sum(0).numpy(), g["grid_%s_sum" % name])
        assert np.array_equal(grid[[0, 1, 8, 9, 1000, 12345, grid.shape[0] - 1]].numpy(), g["grid_%s_rows" % name])


def test_target_layers_index_parity(golden):
    from ait_amd.config import cfg
    from ait_amd.rpn import _AnchorTargetLayer, _ProposalTargetLayer
    g = golden("g8_target_layers")
    prob, deltas, info = cases.rpn_case()
    gt = _dev(cases.gt_case())
    nb = torch.tensor([3, 3]).cuda()
    np.random.seed(3)
    labels, targets, w_in, w_out = _AnchorTargetLayer(16, [8, 16, 32], [0.5, 1, 2])((_dev(prob), gt, _dev(info), nb))
    assert np.array_equal(labels.cpu().numpy().astype(np.int8), g["atl_labels"])
    for name, t in (("atl_targets", targets), ("atl_w_in", w_in), ("atl_w_out", w_out)):
        ok, msg = compare(name, t.contiguous(), g, 1e-5, 1e-5)
        assert ok, msg
    rois = _dev(golden("g7_proposal_layer")["rois_TRAIN"])
    for P in (128, 300):
        cfg.TRAIN.BATCH_SIZE = P
        r, lab, tg, wi, wo = _ProposalTargetLayer(2)(rois, gt, nb)
        assert np.array_equal(r.cpu().numpy(), g["ptl%d_rois" % P])
        assert np.array_equal(lab.cpu().numpy(), g["ptl%d_labels" % P])
        np.testing.assert_allclose(tg.cpu().numpy(), g["ptl%d_targets" % P], rtol=1e-5, atol=1e-5)
        assert np.array_equal(wi.cpu().numpy(), g["ptl%d_w_in" % P])


@pytest.fixture(scope="module")
def model():
    from ait_amd.faster_rcnn import resnet
    m = resnet(('__background__', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3)
    m.create_architecture()
    res = m.load_state_dict(D.make_detector_state_dict(9, D.reference_shapes()), strict=False)
    assert not res.unexpected_keys
    assert all(k.startswith(("RCNN_base.stem.", "RCNN_base.layer")) for k in res.missing_keys)
    return m.cuda()


def test_detector_eval_forward_cfg1(golden, model):
    """BASELINE cfg1 shape: 1 pair, 600x1000 target, 128 proposals.  (1) The product's own forward: the
    proposals are the reference's up to GPU-vs-CPU exp() ulps in the box decode (>= 98 % of the rows within
    2e-3 px), logits compared on the matching rows.  (2) On the REFERENCE's proposals (g9's `rois`, injected
    behind the proposal layer): EVERY one of the 128 similarity logits within 1e-4 relative (+2e-6 abs) of the
    reference -- the bar north_star states, unconditionally."""
    from ait_amd import ops
    from ait_amd.config import cfg
    g = golden("g9_detector_eval")
    cfg.TEST.RPN_POST_NMS_TOP_N = 128
    model.eval()
    im, qr, info, gt, nb = [t.cuda() for t in D.synth_inputs(1, 901)]
    feats = {}
    hooks = [model.RCNN_cls_score.register_forward_hook(lambda m, i, o: feats.__setitem__("score", o)),
             model.coattention.register_forward_hook(lambda m, i, o: feats.__setitem__("co", o)),
             model.transformer.register_forward_hook(lambda m, i, o: feats.__setitem__("ait", o))]
    ops.reset_fallbacks()
    try:
        with torch.no_grad():
            out = model(im, qr, info, gt, nb)
        own_score = feats["score"].cpu().numpy()
        ok, msg = compare("non_img", feats["co"][0], g, 1e-4, 5e-5)
        assert ok, msg
        with torch.no_grad(), _reference_proposals(model, g["rois"]):
            ref_in = model(im, qr, info, gt, nb)
        ref_score = feats["score"].cpu().numpy()
    finally:
        for h in hooks:
            h.remove()
    assert ops.fallback_count() == 0, dict(ops.FALLBACKS)        # nothing left the library's kernels
    assert out[3] == 0 and out[4] == 0 and out[8] is None and out[9] is None
    # (1) own proposals
    rois = out[0].cpu().numpy()
    same = np.abs(rois - g["rois"]).max(-1)[0] <= 2e-3
    assert same.mean() >= 0.98, same.mean()
    np.testing.assert_allclose(own_score[same], g["score"][same], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(out[1].cpu().numpy()[0][same], g["cls_prob"][0][same], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out[2].cpu().numpy()[0][same], g["bbox_pred"][0][same], rtol=1e-3, atol=2e-6)
    # (2) the reference's proposals: all 128 rows
    assert np.array_equal(ref_in[0].cpu().numpy(), g["rois"])
    assert ref_score.shape == g["score"].shape == (128, 2)
    np.testing.assert_allclose(ref_score, g["score"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(ref_in[1].cpu().numpy(), g["cls_prob"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ref_in[2].cpu().numpy(), g["bbox_pred"], rtol=1e-3, atol=2e-6)


@contextlib.contextmanager
def _dropout_off(model):
    """parity is defined at dropout p = 0"""
    saved = [(m, m.p) for m in model.modules() if hasattr(m, "p") and isinstance(m.p, float)]
    for m, _ in saved:
        m.p = 0.0
    try:
        yield
    finally:
        for m, p in saved:
            m.p = p


@contextlib.contextmanager
def _reference_proposals(model, rois):
    """Replace the proposal layer's output by the REFERENCE's own proposals of the same forward
    (golden g13): everything downstream -- both samplers under the reference's RNG order, RoIAlign,
    the AIT, SK, layer4, heads, losses -- then runs on exactly the reference's boxes, so labels and
    losses compare unconditionally (no dependence on a GPU-vs-CPU exp() ulp in the box decode)."""
    layer = model.RCNN_rpn.RPN_proposal
    fixed = torch.from_numpy(np.ascontiguousarray(rois)).cuda()
    h = layer.register_forward_hook(lambda mod, i, o: fixed.to(o.dtype))
    try:
        yield
    finally:
        h.remove()


@pytest.mark.parametrize("P", [128, 300])
def test_detector_train_forward_losses(golden, model, P, record_property):
    from ait_amd.config import cfg
    g = golden("g10_detector_train")
    props = golden("g13_detector_proposals")["voc_prop_rois"]
    cfg.TRAIN.BATCH_SIZE = P
    model.train()
    ins = [t.cuda() for t in D.synth_inputs(1, 1001)]
    with _dropout_off(model):
        np.random.seed(3)
        with torch.no_grad():
            out = model(*ins)
        np.random.seed(3)
        with torch.no_grad(), _reference_proposals(model, props):
            ref_in = model(*ins)
    # (1) the product's own proposals: the sampled RoI set is the reference's up to boundary cases
    frac = _rows_match(out[0].cpu().numpy(), g["P%d_rois" % P])
    record_property("sampled_roi_rows_matching_reference", frac)
    assert frac >= 0.98, frac
    # the RPN losses do not depend on the proposals: always comparable
    np.testing.assert_allclose(np.array([float(out[3]), float(out[4])]), g["P%d_losses" % P][:2], rtol=2e-4, atol=2e-6)
    # (2) on the reference's proposals: identical sampled RoIs, labels and all five losses -- unconditional
    np.testing.assert_allclose(ref_in[0].cpu().numpy(), g["P%d_rois" % P], rtol=0, atol=1e-4)
    assert np.array_equal(ref_in[8].cpu().numpy(), g["P%d_labels" % P])
    losses = np.array([float(x) for x in ref_in[3:8]])
    np.testing.assert_allclose(losses, g["P%d_losses" % P], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(ref_in[1].cpu().numpy(), g["P%d_cls_prob" % P], rtol=1e-4, atol=1e-6)


def test_detector_train_step_gradients_vs_oracle(model):
    """One small training step (P=16) forward+backward on the GPU vs the CPU oracle, same
    weights / inputs / NumPy RNG stream: losses and a few parameter gradients."""
    from ait_amd.config import cfg
    cfg.TRAIN.BATCH_SIZE = 16
    cfgd = D.default_config()
    cfgd["TRAIN"]["BATCH_SIZE"] = 16
    sd = {k: v.clone() for k, v in D.make_detector_state_dict(9, D.reference_shapes()).items()}
    watch = ["transformer.encoder.layer_stack.0.slf_attn.w_qs.weight", "transformer.dec_trans.0.bias",
             "RCNN_cls_score.1.weight", "coattention.img_trans.0.weight", "RCNN_rpn.RPN_Conv.bias",
             "RCNN_base.backbone.layer3.5.conv3.weight", "transformer.enc_emb.0.weight",
             # the proposal tail (ait_tail_fwd / ait_tail_bwd: both SK blocks + layer4 as one node): every convolution of
             # the first and last bottleneck, the stride-2 shortcut, both SK branches
             "RCNN_top.0.0.conv1.weight", "RCNN_top.0.0.conv2.weight", "RCNN_top.0.0.conv3.weight",
             "RCNN_top.0.2.conv1.weight", "RCNN_top.0.2.conv2.weight", "RCNN_top.0.2.conv3.weight",
             "RCNN_top.0.0.downsample.0.weight",
             "sk.sk_props.convs.0.0.weight", "sk.sk_props.convs.0.0.bias",
             "sk.sk_props.convs.1.0.weight", "sk.sk_props.convs.1.0.bias"]
    watch = [k for k in watch if k in sd]
    assert len(watch) >= 16, watch
    for k in watch:
        sd[k].requires_grad_(True)
    ins = D.synth_inputs(1, 1101)
    np.random.seed(3)
    out, _aux = D.detector_forward(sd, cfgd, *ins, True)
    (out[3] + out[4] + out[5] + out[6] + out[7]).backward()
    model.train()
    # the oracle's own proposals are injected into the product (as _reference_proposals does with the
    # reference's): the comparison below never depends on a GPU-vs-CPU ulp in the box decode
    with _dropout_off(model), _reference_proposals(model, _aux["rpn_rois"].numpy()):
        model.zero_grad(set_to_none=True)
        np.random.seed(3)
        res = model(*[t.cuda() for t in ins])
        (res[3] + res[4] + res[5] + res[6] + res[7]).backward()
    np.testing.assert_allclose(res[0].cpu().numpy(), out[0].numpy(), rtol=0, atol=1e-4)
    params = dict(model.named_parameters())
    for i in range(3, 8):
        assert abs(float(res[i]) - float(out[i])) <= 2e-4 * abs(float(out[i])) + 2e-6
    for k in watch:
        # (RCNN_top.0 IS RCNN_base.backbone.layer4 -- one module under two names, resnet_sys_transformer_sk_dilat.py:230,422;
        # named_parameters() lists it once)
        pk = k if k in params else "RCNN_base.backbone.layer4." + k[len("RCNN_top.0."):]
        got, want = params[pk].grad.cpu(), sd[k].grad
        rel = float((got - want).norm() / (want.norm() + 1e-12))
        assert rel < 5e-3, (k, rel)


WATCH18 = ["transformer.encoder.layer_stack.0.slf_attn.w_qs.weight", "transformer.dec_trans.0.bias",
           "RCNN_cls_score.1.weight", "coattention.img_trans.0.weight", "RCNN_rpn.RPN_Conv.bias",
           "RCNN_base.backbone.layer3.5.conv3.weight", "transformer.enc_emb.0.weight",
           "RCNN_top.0.0.conv1.weight", "RCNN_top.0.0.conv2.weight", "RCNN_top.0.0.conv3.weight",
           "RCNN_top.0.2.conv1.weight", "RCNN_top.0.2.conv2.weight", "RCNN_top.0.2.conv3.weight",
           "RCNN_top.0.0.downsample.0.weight",
           "sk.sk_props.convs.0.0.weight", "sk.sk_props.convs.0.0.bias",
           "sk.sk_props.convs.1.0.weight", "sk.sk_props.convs.1.0.bias"]


def test_detector_headline_size_eval_logits_vs_oracle(model):
    """The headline shape -- one 600x1000 target, one 128x128 query, 300 proposals (BASELINE configs[1] per pair) -- in
    eval(): EVERY one of the 300 similarity logits (faster_rcnn_sys_transformer_sk_dilat.py:277-290) within 1e-4 relative
    (+2e-6) of the CPU oracle's, on the oracle's own proposals injected behind the proposal layer (the cfg1 test's recipe:
    no dependence on a GPU-vs-CPU exp() ulp in the box decode); cls_prob and bbox_pred with them."""
    from ait_amd import ops
    from ait_amd.config import cfg
    cfg.TEST.RPN_POST_NMS_TOP_N = 300
    cfgd = D.default_config()
    cfgd["TEST"]["RPN_POST_NMS_TOP_N"] = 300
    sd = D.make_detector_state_dict(9, D.reference_shapes())
    ins = D.synth_inputs(1, 2201)
    with torch.no_grad():
        want, aux = D.detector_forward(sd, cfgd, *ins, False)
    assert tuple(want[0].shape) == (1, 300, 5) and tuple(aux["score"].shape) == (300, 2)
    model.eval()
    feats = {}
    h = model.RCNN_cls_score.register_forward_hook(lambda m, i, o: feats.__setitem__("score", o))
    ops.reset_fallbacks()
    try:
        with torch.no_grad(), _reference_proposals(model, want[0].numpy()):
            got = model(*[t.cuda() for t in ins])
    finally:
        h.remove()
    assert ops.fallback_count() == 0, dict(ops.FALLBACKS)
    assert np.array_equal(got[0].cpu().numpy(), want[0].numpy())
    score, ref = feats["score"].cpu().numpy(), aux["score"].numpy()
    err = np.abs(score - ref) / (1e-4 * np.abs(ref) + 2e-6)
    print("P=300 logits: worst error / tolerance", float(err.max()), "logit range", float(ref.min()), float(ref.max()))
    np.testing.assert_allclose(score, ref, rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(got[1].cpu().numpy(), want[1].numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got[2].cpu().numpy(), want[2].numpy(), rtol=1e-3, atol=2e-6)


def test_detector_headline_size_train_step_vs_oracle(model):
    """The same pair in train() (TRAIN.BATCH_SIZE = 300 sampled RoIs, dropout off): the sampled RoIs and labels identical,
    the five losses within 2e-4, and the gradients of the 18 watched parameters (AIT, heads, co-attention, RPN, trunk,
    every convolution of layer4's first and last bottleneck, both SK branches) within 1e-3 relative L2 of the oracle's
    backward -- the step the timed region runs, at its size (faster_rcnn_sys_transformer_sk_dilat.py:277-314)."""
    from ait_amd.config import cfg
    cfg.TRAIN.BATCH_SIZE = 300
    cfgd = D.default_config()
    cfgd["TRAIN"]["BATCH_SIZE"] = 300
    sd = {k: v.clone() for k, v in D.make_detector_state_dict(9, D.reference_shapes()).items()}
    watch = [k for k in WATCH18 if k in sd]
    assert len(watch) == 18, watch
    for k in watch:
        sd[k].requires_grad_(True)
    ins = D.synth_inputs(1, 2201)
    np.random.seed(3)
    out, aux = D.detector_forward(sd, cfgd, *ins, True)
    assert tuple(out[0].shape) == (1, 300, 5)
    (out[3] + out[4] + out[5] + out[6] + out[7]).backward()
    model.train()
    with _dropout_off(model), _reference_proposals(model, aux["rpn_rois"].numpy()):
        model.zero_grad(set_to_none=True)
        np.random.seed(3)
        res = model(*[t.cuda() for t in ins])
        (res[3] + res[4] + res[5] + res[6] + res[7]).backward()
    np.testing.assert_allclose(res[0].cpu().numpy(), out[0].numpy(), rtol=0, atol=1e-4)
    assert np.array_equal(res[8].cpu().numpy(), out[8].numpy())
    for i in range(3, 8):
        assert abs(float(res[i]) - float(out[i])) <= 2e-4 * abs(float(out[i])) + 2e-6, (i, float(res[i]), float(out[i]))
    np.testing.assert_allclose(res[1].detach().cpu().numpy(), out[1].detach().numpy(), rtol=1e-4, atol=1e-6)
    params = dict(model.named_parameters())
    rels = {}
    for k in watch:
        pk = k if k in params else "RCNN_base.backbone.layer4." + k[len("RCNN_top.0."):]
        got, want = params[pk].grad.cpu(), sd[k].grad
        rels[k] = float((got - want).norm() / (want.norm() + 1e-12))
    print("P=300 train step, relative L2 of the watched gradients:", {k: round(v, 6) for k, v in rels.items()})
    # (VERDICT r5 asked for <= 1e-3; measured <= 3.4e-5 -- at P = 300 no ReLU of this pair sits within rounding of its kink)
    assert max(rels.values()) < 2e-4, rels


def test_detector_coco_variant(golden):
    """COCO variant (faster_rcnn_coatt_transformer_sk.py): non-local co-attention, A = 12."""
    from ait_amd import config
    from ait_amd.faster_rcnn import resnet_coco
    g = golden("g11_detector_coco")
    saved = (config.cfg.ANCHOR_SCALES, config.cfg.MAX_NUM_GT_BOXES)
    config.cfg_from_list(['ANCHOR_SCALES', [4, 8, 16, 32], 'MAX_NUM_GT_BOXES', 50])
    try:
        m = resnet_coco(('__background__', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3)
        m.create_architecture()
        res = m.load_state_dict(D.make_detector_state_dict(11, D.reference_shapes(A=12, variant="coco")), strict=False)
        assert not res.unexpected_keys
        assert all(k.startswith(("RCNN_base.stem.", "RCNN_base.layer")) for k in res.missing_keys)
        m = m.cuda().eval()
        config.cfg.TEST.RPN_POST_NMS_TOP_N = 128
        ins = [t.cuda() for t in D.synth_inputs(1, 1101, max_gt=50)]
        feats = {}
        h = m.RCNN_cls_score.register_forward_hook(lambda mod, i, o: feats.__setitem__("score", o))
        with torch.no_grad():
            out = m(*ins)
        h.remove()
        same = np.abs(out[0].cpu().numpy() - g["rois"]).max(-1)[0] <= 2e-3
        assert same.mean() >= 0.98
        np.testing.assert_allclose(feats["score"].cpu().numpy()[same], g["score"][same], rtol=1e-4, atol=2e-6)
        # training forward: sampled RoIs / labels / losses, on the product's own proposals (RoI set
        # up to boundary cases, RPN losses exact) and on the reference's proposals (everything)
        m.train()
        config.cfg.TRAIN.BATCH_SIZE = 128
        with _dropout_off(m):
            np.random.seed(3)
            with torch.no_grad():
                out = m(*ins)
            np.random.seed(3)
            with torch.no_grad(), _reference_proposals(m, golden("g13_detector_proposals")["coco_prop_rois"]):
                ref_in = m(*ins)
        # (own proposals: a proposal whose score ties within fp32 summation-order noise of its neighbour's can
        # swap NMS / top-k places, and one swap moves a few of the 128 sampled rows; the injected run below is
        # the exact check of everything behind the proposals)
        assert _rows_match(out[0].cpu().numpy(), g["train_rois"]) >= 0.95
        np.testing.assert_allclose(np.array([float(out[3]), float(out[4])]), g["train_losses"][:2], rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(ref_in[0].cpu().numpy(), g["train_rois"], rtol=0, atol=1e-4)
        assert np.array_equal(ref_in[8].cpu().numpy(), g["train_labels"])
        np.testing.assert_allclose(np.array([float(x) for x in ref_in[3:8]]), g["train_losses"], rtol=2e-4, atol=2e-6)
    finally:
        config.cfg.ANCHOR_SCALES, config.cfg.MAX_NUM_GT_BOXES = saved


def test_resnet101_variant_runs():
    """BASELINE cfg5 backbone (ResNet101, cfgs/res101.yml) in fp32: builds with the reference's
    key set and trains one small step (the bf16 arithmetic of cfg5 is a later round)."""
    from ait_amd import config
    from ait_amd.faster_rcnn import resnet
    m = resnet(('__background__', 'fg'), 101, pretrained=False, class_agnostic=True, num_K=3)
    m.create_architecture()
    keys = set(m.state_dict())
    assert set(D.reference_shapes(n_layers=101)) <= keys
    assert "RCNN_base.backbone.layer3.22.conv3.weight" in keys
    m = m.cuda().train()
    config.cfg.TRAIN.BATCH_SIZE = 32
    np.random.seed(3)
    ins = [t.cuda() for t in D.synth_inputs(1, 7, im_hw=(320, 480))]
    out = m(*ins)
    loss = out[3] + out[4] + out[5] + out[6] + out[7]
    loss.backward()
    assert torch.isfinite(loss)
    assert tuple(out[0].shape) == (1, 32, 5)


def test_eval_postprocessing_vs_reference_golden(golden):
    """test_net_*.py post-processing (de-normalise, decode, clip, rescale, threshold, sort, second NMS at
    cfg.TEST.NMS, top-100) against golden g14: the imported reference's own bbox_transform_inv / clip_boxes /
    _C.nms called in the driver's order (oracle/gen_golden_post.py, test_net_coco.py:381-449).  Case a: the
    reference's own eval outputs (g9); case b: 300 seeded boxes with clipping, a score threshold and the
    max_per_image cut.  Boxes within 1e-3 px (GPU exp()), scores exact, the same detections in the same order."""
    from oracle import gen_golden_post as G
    from ait_amd.postprocess import detections
    g = golden("g14_postprocess")
    for name, case in (("a", G.case_a), ("b", G.case_b)):
        rois, prob, bbox, info, thresh, mpi = case()
        got = detections(rois.cuda(), prob.cuda(), bbox.cuda(), info.cuda(), float(info[0, 2]), thresh=thresh,
                         max_per_image=mpi).cpu().numpy()
        want = g[name + "_dets"]
        assert got.shape == want.shape, (name, got.shape, want.shape)
        assert np.array_equal(got[:, 4], want[:, 4]), name            # the same boxes survive, in the same order
        np.testing.assert_allclose(got[:, :4], want[:, :4], rtol=1e-5, atol=1e-3)


def test_detector_full_size_step_matches_reference_work_and_is_reproducible(model):
    """BASELINE cfg2 size (4 pairs, 600x1000, 300 proposals), training mode.  Size-independent
    properties: (1) the default fast path (dead SK / trunk positions skipped, channels-last stages,
    channels-last RoIAlign) and the path that does all of the reference's work in NCHW activations agree on every
    loss and on the similarity probabilities; (2) the same seeds reproduce the same step."""
    import ait_amd.faster_rcnn as fr
    from ait_amd.config import cfg
    from ait_amd.roi_layers import ROIAlign
    cfg.TRAIN.BATCH_SIZE = 300
    model.train()
    for m in model.modules():                                        # dropout off: compare arithmetic, not masks
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    ins = [t.cuda() for t in D.synth_inputs(4, 77)]

    def run():
        np.random.seed(3)
        torch.manual_seed(5)
        with torch.no_grad():
            out = model(*ins)
        return torch.stack([out[3], out[4], out[5], out[6], out[7]]).double().cpu(), out[1].double().cpu(), out[0].cpu()

    saved = (fr._SK_FULL, fr._TOP_NHWC, fr._BASE_NHWC, model.RCNN_roi_align.channels_last, model.transformer.channels_last_out)
    try:
        losses, prob, rois = run()
        losses2, prob2, rois2 = run()
        fr._SK_FULL, fr._TOP_NHWC, fr._BASE_NHWC = True, False, False
        model.RCNN_roi_align.channels_last = False
        model.transformer.channels_last_out = False
        losses_ref, prob_ref, rois_ref = run()
    finally:
        fr._SK_FULL, fr._TOP_NHWC, fr._BASE_NHWC = saved[:3]
        model.RCNN_roi_align.channels_last, model.transformer.channels_last_out = saved[3], saved[4]
        model.train()
    assert bool(torch.isfinite(losses).all())
    # MIOpen's split-K convolution kernels accumulate with atomics: two runs differ in the last bits
    assert float(((rois - rois2).abs().amax(-1) <= 2e-3).float().mean()) >= 0.98
    assert float((losses - losses2).abs().max()) <= 2e-4 * float(losses.abs().max()) + 1e-6
    same = (rois - rois_ref).abs().amax(-1) <= 2e-3                    # proposals can swap at near-tied scores
    assert float(same.float().mean()) >= 0.98
    assert float((losses - losses_ref).abs().max()) <= 2e-4 * float(losses_ref.abs().max()) + 1e-6
    assert float((prob - prob_ref)[same].abs().max()) <= 1e-4


