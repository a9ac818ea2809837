"""Pins oracle/detector_ref.py (CPU restatement of the detector around the hot path) against the
golden vectors recorded from the imported reference (tests/golden/g6..g10) -- CPU only."""
import numpy as np
import pytest
import torch

from oracle import cases, detector_ref as D
from oracle.digest import compare


def test_anchor_table_known_answer(golden):
    """The only known-answer vector in the reference tree is the comment block at
    rpn/generate_anchors.py:12-37.  It is the 1-BASED Matlab table; the Python function (0-based
    base anchor, :51) returns it shifted by -1, which is what the golden recorded."""
    want = np.array([[-83, -39, 100, 56], [-175, -87, 192, 104], [-359, -183, 376, 200],
                     [-55, -55, 72, 72], [-119, -119, 136, 136], [-247, -247, 264, 264],
                     [-35, -79, 52, 96], [-79, -167, 96, 184], [-167, -343, 184, 360]], np.float64)
    assert np.array_equal(D.generate_anchors(), want - 1)
    g = golden("g6_anchors")
    assert np.array_equal(D.generate_anchors(scales=(8, 16, 32)), g["anchors_voc"])
    assert np.array_equal(D.generate_anchors(scales=(4, 8, 16, 32)), g["anchors_coco"])
    for name, scales in (("voc", (8, 16, 32)), ("coco", (4, 8, 16, 32))):
        grid = D.anchor_grid(cases.FEAT_H, cases.FEAT_W, 16, scales, (0.5, 1, 2))
        assert list(grid.shape) == list(g["grid_%s_shape" % name])
        assert np.array_equal(grid.double().sum(0).numpy(), g["grid_%s_sum" % name])
        rows = [0, 1, 8, 9, 1000, 12345, grid.shape[0] - 1]
        assert np.array_equal(grid[rows].numpy(), g["grid_%s_rows" % name])


@pytest.mark.parametrize("key", ["TRAIN", "TEST"])
def test_proposal_layer(golden, key):
    g = golden("g7_proposal_layer")
    prob, deltas, info = cases.rpn_case()
    rois = D.proposal_layer(D.default_config(), key, torch.from_numpy(prob), torch.from_numpy(deltas),
                            torch.from_numpy(info))
    want = g["rois_" + key]
    assert rois.shape == want.shape
    np.testing.assert_allclose(rois.numpy(), want, rtol=0, atol=1e-3)


def test_target_layers_index_parity(golden):
    """np.random.seed(3) + the reference's call order => identical sampled indices."""
    g = golden("g8_target_layers")
    cfgd = D.default_config()
    prob, deltas, info = cases.rpn_case()
    gt = torch.from_numpy(cases.gt_case())
    np.random.seed(3)
    labels, targets, w_in, w_out = D.anchor_target_layer(cfgd, torch.from_numpy(prob), gt, torch.from_numpy(info))
    assert np.array_equal(labels.numpy().astype(np.int8), g["atl_labels"])
    for name, t in (("atl_targets", targets), ("atl_w_in", w_in), ("atl_w_out", w_out)):
        ok, msg = compare(name, t.contiguous(), g, 1e-5, 1e-6)
        assert ok, msg
    rois = torch.from_numpy(golden("g7_proposal_layer")["rois_TRAIN"])
    for P in (128, 300):
        cfgd["TRAIN"]["BATCH_SIZE"] = P
        r, lab, tg, wi, wo = D.proposal_target_layer(cfgd, rois, gt)
        assert np.array_equal(r.numpy(), g["ptl%d_rois" % P])
        assert np.array_equal(lab.numpy(), g["ptl%d_labels" % P])
        np.testing.assert_allclose(tg.numpy(), g["ptl%d_targets" % P], rtol=1e-5, atol=1e-6)
        assert np.array_equal(wi.numpy(), g["ptl%d_w_in" % P])
        assert np.array_equal(wo.numpy(), g["ptl%d_w_out" % P])


@pytest.fixture(scope="module")
def det_sd():
    return D.make_detector_state_dict(9, D.reference_shapes())


def test_detector_eval_forward_cfg1(golden, det_sd):
    """BASELINE cfg1: one (600x1000, 128x128) pair, 128 proposals, CPU forward."""
    g = golden("g9_detector_eval")
    cfgd = D.default_config()
    cfgd["TEST"]["RPN_POST_NMS_TOP_N"] = 128
    im, qr, info, gt, nb = D.synth_inputs(1, 901)
    with torch.no_grad():
        out, aux = D.detector_forward(det_sd, cfgd, im, qr, info, gt, nb, False)
    for name in ("non_img", "non_qry", "props", "ait_out"):
        ok, msg = compare(name, aux[name], g, 1e-4, 2e-5)
        assert ok, msg
    np.testing.assert_allclose(out[0].numpy(), g["rois"], rtol=0, atol=1e-3)
    # similarity logits: north_star tolerance 1e-4 relative (+ a small absolute floor)
    np.testing.assert_allclose(aux["score"].numpy(), g["score"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out[1].numpy(), g["cls_prob"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out[2].numpy(), g["bbox_pred"], rtol=1e-4, atol=1e-6)
    assert out[3] == 0 and out[8] is None


@pytest.mark.parametrize("P", [128, 300])
def test_detector_train_forward_losses(golden, det_sd, P):
    g = golden("g10_detector_train")
    cfgd = D.default_config()
    cfgd["TRAIN"]["BATCH_SIZE"] = P
    np.random.seed(3)
    im, qr, info, gt, nb = D.synth_inputs(1, 1001)
    with torch.no_grad():
        out, _ = D.detector_forward(det_sd, cfgd, im, qr, info, gt, nb, True)
    np.testing.assert_allclose(out[0].numpy(), g["P%d_rois" % P], rtol=0, atol=1e-3)
    assert np.array_equal(out[8].numpy(), g["P%d_labels" % P])
    losses = np.array([float(x) for x in out[3:8]])
    np.testing.assert_allclose(losses, g["P%d_losses" % P], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out[1].numpy(), g["P%d_cls_prob" % P], rtol=1e-4, atol=1e-6)


def test_detector_coco_variant(golden):
    """COCO variant (non-local co-attention, 12 anchors, 50 GT slots): eval logits + train losses."""
    g = golden("g11_detector_coco")
    sd = D.make_detector_state_dict(11, D.reference_shapes(A=12, variant="coco"))
    cfgd = D.default_config()
    cfgd["ANCHOR_SCALES"] = [4, 8, 16, 32]
    cfgd["TEST"]["RPN_POST_NMS_TOP_N"] = 128
    im, qr, info, gt, nb = D.synth_inputs(1, 1101, max_gt=50)
    with torch.no_grad():
        out, aux = D.detector_forward(sd, cfgd, im, qr, info, gt, nb, False)
    for name in ("non_img", "non_qry"):
        ok, msg = compare(name, aux[name], g, 1e-4, 2e-5)
        assert ok, msg
    np.testing.assert_allclose(out[0].numpy(), g["rois"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(aux["score"].numpy(), g["score"], rtol=1e-4, atol=1e-6)
    cfgd["TRAIN"]["BATCH_SIZE"] = 128
    np.random.seed(3)
    with torch.no_grad():
        out, _ = D.detector_forward(sd, cfgd, im, qr, info, gt, nb, True)
    np.testing.assert_allclose(out[0].numpy(), g["train_rois"], rtol=0, atol=1e-3)
    assert np.array_equal(out[8].numpy(), g["train_labels"])
    np.testing.assert_allclose(np.array([float(x) for x in out[3:8]]), g["train_losses"], rtol=1e-4, atol=1e-6)


def test_oracle_postprocessing_vs_reference_golden(golden):
    """pins oracle/detector_ref.postprocess_detections (the checker of the eval post-processing, test_net_coco.py:
    381-449) on golden g14 = the imported reference's own functions in the driver's call order"""
    from oracle import gen_golden_post as G
    g = golden("g14_postprocess")
    for name, case in (("a", G.case_a), ("b", G.case_b)):
        rois, prob, bbox, info, thresh, mpi = case()
        got = D.postprocess_detections(D.default_config(), rois, prob, bbox, info, float(info[0, 2]),
                                       nms_thr=float(g["nms_thr"]), thresh=thresh, max_per_image=mpi).numpy()
        want = g[name + "_dets"]
        assert got.shape == want.shape
        assert np.array_equal(got[:, 4], want[:, 4])
        np.testing.assert_allclose(got[:, :4], want[:, :4], rtol=1e-6, atol=1e-4)
