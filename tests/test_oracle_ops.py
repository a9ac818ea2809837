"""Pins oracle/native.c (RoIAlign fwd/bwd, NMS) and the anchor restatement against the golden
vectors recorded from the reference (tests/golden/g4..g6) -- CPU only."""
import numpy as np
import pytest

from oracle import cases, native


def test_roi_align_fwd_matches_reference_golden(golden):
    g = golden("g4_roi_align")
    feat, rois = cases.roi_align_case()
    # same arithmetic, same order, no FMA contraction: bit-exact with the reference C++
    assert np.array_equal(native.roi_align_fwd(feat, rois), g["y"])
    assert np.array_equal(native.roi_align_fwd(feat, rois, sampling_ratio=2), g["y_sr2"])
    from oracle.digest import seeded
    feat_r = seeded(402, (2, 8, cases.FEAT_H, cases.FEAT_W))
    rois_r = cases.random_rois(403, 64, 2)
    assert np.array_equal(native.roi_align_fwd(feat_r, rois_r), g["y_rand"])


def test_roi_align_fwd_edge_semantics(golden):
    y = golden("g4_roi_align")["y"]
    assert np.all(y[8] == 0.0)                      # entirely out of range -> zeros
    assert np.all(np.isfinite(y))


def test_roi_align_bwd_is_adjoint_of_fwd():
    """The reference has no CPU backward (ROIAlign.h:44).  RoIAlign is linear in the feature
    map, so backward must be its exact adjoint: <fwd(x), g> == <x, bwd(g)>, and bwd(g) must
    equal the finite-difference gradient of the pinned forward."""
    from oracle.digest import seeded
    feat, rois = cases.roi_align_case()
    feat = feat[:, :3].copy()
    g = seeded(41, (rois.shape[0], 3, 7, 7))
    y = native.roi_align_fwd(feat, rois)
    gx = native.roi_align_bwd(g, rois, feat.shape)
    lhs = float((y.astype(np.float64) * g).sum())
    rhs = float((feat.astype(np.float64) * gx).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))
    # finite differences on a handful of feature cells (linearity => exact up to rounding)
    rs = np.random.RandomState(0)
    for _ in range(12):
        idx = tuple(rs.randint(0, s) for s in feat.shape)
        e = np.zeros_like(feat)
        e[idx] = 1.0
        col = native.roi_align_fwd(e, rois)
        assert abs(float((col.astype(np.float64) * g).sum()) - gx[idx]) <= 1e-4 * max(1.0, abs(gx[idx]))


def test_roi_align_empty():
    feat, _ = cases.roi_align_case()
    y = native.roi_align_fwd(feat, np.zeros((0, 5), np.float32))
    assert y.shape == (0, 16, 7, 7)


@pytest.mark.parametrize("n", cases.NMS_SIZES)
@pytest.mark.parametrize("thr", cases.NMS_THRESHOLDS)
def test_nms_matches_reference_golden(golden, n, thr):
    g = golden("g5_nms")
    box, sc = cases.nms_boxes(500 + n, n)
    keep = native.nms(box, sc, thr)
    assert keep.dtype == np.int64
    assert np.array_equal(keep, g["keep_n%d_t%02d" % (n, int(thr * 10))])


def test_nms_ties_suppress_on_equal(golden):
    g = golden("g5_nms")
    box, sc = cases.nms_tie_case()
    for thr in (0.7, 0.5, 0.3):
        assert np.array_equal(native.nms(box, sc, thr), g["keep_tie_t%02d" % int(thr * 10)])
    # the CPU reference suppresses when ovr == thr (nms_cpu.cpp:60)
    assert 1 not in set(native.nms(box, sc, 0.7).tolist())
    box2, sc2 = cases.nms_boxes(777, 2000, integer=True)
    assert np.array_equal(native.nms(box2, sc2, 0.7), g["keep_int2000_t07"])
    assert native.nms(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.7).shape == (0,)
    assert g["keep_empty"].shape == (0,)


@pytest.mark.parametrize("key", ["TRAIN", "TEST"])
def test_nms_on_the_reference_proposal_layer_candidates(golden, key):
    """golden g12: the boxes the reference's proposal layer hands to its NMS (proposal_layer.py:134-153)
    and the indices it keeps; the C restatement reproduces them image by image."""
    g = golden("g12_proposal_nms")
    cand, want, n_want = g["cand_" + key], g["keep_" + key], g["nkeep_" + key]
    for b in range(cand.shape[0]):
        n = cand.shape[1]
        scores = np.linspace(1.0, 0.0, n, dtype=np.float32)            # already sorted by descending score
        keep = native.nms(cand[b], scores, float(g["thr_" + key]))[:want.shape[1]]
        assert keep.size == int(n_want[b])
        assert np.array_equal(keep, want[b, :keep.size].astype(np.int64))
