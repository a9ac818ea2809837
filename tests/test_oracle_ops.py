"""Pins oracle/native.c (RoIAlign fwd/bwd, NMS) and the anchor restatement against the golden
vectors recorded from the reference (tests/golden/g4..g6) -- CPU only."""
import os

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


def test_c_restatement_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """`-fsanitize=address` on the host C side.  oracle/native.c (198 lines of index arithmetic restating
    ROIAlign_cpu.cpp:17-219 / nms_cpu.cpp:5-65) is built WITH oracle/native_san_driver.c under ASan + UBSan (no recovery) and
    run on RoIs inside the map, touching and crossing every border, narrower than a pixel, the whole image and out of range,
    on an empty RoI list, and on NMS inputs of 1 / 64 / 65 / 3000 boxes with ties: the run must be clean, and its outputs
    bit-equal to the ordinary -O2 build's (the one every other test and the GPU parity tests use)."""
    import struct
    import subprocess
    import sys
    from oracle import cases, native
    from oracle.digest import seeded
    here = os.path.dirname(native.SRC)
    exe = tmp_path / "native_san"
    subprocess.check_call(["gcc", "-O1", "-g", "-ffp-contract=off", "-fno-fast-math", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-o", str(exe), native.SRC, os.path.join(here, "native_san_driver.c"), "-lm"])
    blob, want = [], []
    roi_cases = []
    feat = seeded(5, (2, 3, cases.FEAT_H, cases.FEAT_W))
    rois = np.concatenate([cases.random_rois(40, 6, 2), cases.roi_align_case()[1]]).astype(np.float32)
    roi_cases.append((feat, rois, 0))
    roi_cases.append((feat, rois[:7], 2))                                        # a fixed sampling grid
    roi_cases.append((seeded(6, (1, 2, 5, 4)), np.array([[0, -50, -50, 500, 500], [0, 3, 3, 3.2, 3.1]], np.float32), 0))
    roi_cases.append((seeded(7, (1, 1, 3, 3)), np.zeros((0, 5), np.float32), 0))     # no RoIs
    for f, r, sr in roi_cases:
        n, (B, C, H, W) = r.shape[0], f.shape
        go = seeded(8 + n, (n, C, 7, 7))
        blob.append(struct.pack("<i8if", 0, n, B, C, H, W, 7, 7, sr, 1.0 / 16.0) + f.tobytes() + r.tobytes() + go.tobytes())
        want.append((native.roi_align_fwd(f, r, sampling_ratio=sr) if n else np.zeros((0, C, 7, 7), np.float32),
                     native.roi_align_bwd(go, r, f.shape, sampling_ratio=sr)))
    nms_cases = []
    for n, thr in ((1, 0.7), (64, 0.7), (65, 0.3), (3000, 0.7)):
        box, sc = cases.nms_boxes(100 + n, n)
        nms_cases.append((box, sc, thr))
    for thr in (0.7, 0.5, 0.3):                                                  # IoU exactly == thr
        nms_cases.append(cases.nms_tie_case() + (thr,))
    for box, sc, thr in nms_cases:
        order = np.argsort(-np.asarray(sc, np.float32), kind="stable").astype(np.int64)
        blob.append(struct.pack("<iif", 1, box.shape[0], thr) + np.ascontiguousarray(box, np.float32).tobytes() + order.tobytes())
        want.append(native.nms(box, sc, thr))
    fin, fout = tmp_path / "cases.bin", tmp_path / "out.bin"
    fin.write_bytes(struct.pack("<i", len(blob)) + b"".join(blob))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(exe), str(fin), str(fout)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok" and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    got = fout.read_bytes()
    o = 0
    for (f, r_, sr), (y, gi) in zip(roi_cases, want[:len(roi_cases)]):
        rc = struct.unpack_from("<2i", got, o); o += 8
        assert rc == (0, 0)
        yy = np.frombuffer(got, np.float32, y.size, o).reshape(y.shape); o += 4 * y.size
        gg = np.frombuffer(got, np.float32, gi.size, o).reshape(gi.shape); o += 4 * gi.size
        assert np.array_equal(yy, y) and np.array_equal(gg, gi)
    for keep in want[len(roi_cases):]:
        k = struct.unpack_from("<q", got, o)[0]; o += 8
        kk = np.frombuffer(got, np.int64, k, o); o += 8 * k
        assert k == keep.size and np.array_equal(kk, keep)
    assert o == len(got)
