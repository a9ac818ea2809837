"""oracle/cases.py -- TEST INFRASTRUCTURE ONLY.

Seeded input definitions shared by oracle/gen_golden.py (which runs the reference on them)
and tests/ (which run the oracle and the HIP path on them).  Inputs are regenerated from
seeds; only the reference OUTPUTS are stored under tests/golden/.
"""
import numpy as np

from .digest import seeded

FEAT_H, FEAT_W = 38, 63          # C4 feature of a 600x1000 target
IM_H, IM_W = 600, 1000


def roi_align_case():
    """G4: feature [2,16,38,63] + RoIs covering the edge cases of SURVEY.md 8c."""
    feat = seeded(401, (2, 16, FEAT_H, FEAT_W))
    rois = np.array([
        [0, 100.0, 80.0, 420.0, 360.0],        # inside
        [1, 0.0, 0.0, 999.0, 599.0],           # whole image (max adaptive grid 6x9)
        [0, 0.0, 0.0, 15.0, 15.0],             # touches top-left border, exactly one cell
        [1, 983.0, 583.0, 999.0, 599.0],       # touches bottom-right border
        [0, 333.3, 222.2, 333.9, 222.7],       # < 1 px  -> forced 1x1
        [1, 500.0, 300.0, 500.0, 300.0],       # degenerate point
        [0, -64.0, -48.0, 90.0, 70.0],         # partly outside (negative coords)
        [1, 900.0, 500.0, 1200.0, 800.0],      # beyond the image: out-of-range samples = 0
        [0, 1500.0, 900.0, 1600.0, 1000.0],    # entirely out of range -> zeros
        [1, 17.3, 101.9, 611.7, 207.2],        # wide, fractional
        [0, 640.5, 33.25, 700.75, 590.0],      # tall, fractional
        [1, 420.0, 360.0, 100.0, 80.0],        # malformed (x2 < x1): forced 1x1
    ], np.float32)
    return feat, rois


def random_rois(seed, n, batch, im_h=IM_H, im_w=IM_W, min_side=8.0, max_side=480.0):
    rs = np.random.RandomState(seed)
    w = rs.uniform(min_side, max_side, n)
    h = rs.uniform(min_side, max_side, n)
    x1 = rs.uniform(0, im_w - 1, n)
    y1 = rs.uniform(0, im_h - 1, n)
    x2 = np.minimum(x1 + w, im_w - 1)
    y2 = np.minimum(y1 + h, im_h - 1)
    b = rs.randint(0, batch, n)
    return np.stack([b, x1, y1, x2, y2], 1).astype(np.float32)


def nms_boxes(seed, n, im_h=IM_H, im_w=IM_W, integer=False):
    """n boxes, already sorted by descending (distinct) score, like proposal_layer.py:129-153
    hands them to nms.  Clustered so that a realistic fraction gets suppressed."""
    rs = np.random.RandomState(seed)
    n_ctr = max(1, n // 12)
    cx = rs.uniform(0, im_w, n_ctr)
    cy = rs.uniform(0, im_h, n_ctr)
    which = rs.randint(0, n_ctr, n)
    w = rs.uniform(16, 400, n)
    h = rs.uniform(16, 400, n)
    x = cx[which] + rs.normal(0, 12, n)
    y = cy[which] + rs.normal(0, 12, n)
    box = np.stack([x - w / 2, y - h / 2, x + w / 2, y + h / 2], 1)
    box[:, 0::2] = np.clip(box[:, 0::2], 0, im_w - 1)
    box[:, 1::2] = np.clip(box[:, 1::2], 0, im_h - 1)
    if integer:
        box = np.round(box)
    scores = np.sort(rs.permutation(n).astype(np.float64) / n + 1e-3)[::-1]
    return box.astype(np.float32), np.ascontiguousarray(scores.astype(np.float32))


def nms_tie_case():
    """Boxes whose IoU is EXACTLY 0.7f / 0.5f / 0.3f in fp32 (inter/union = 70/100, 50/100,
    30/100 with the +1 convention), so `>=` (CPU reference, nms_cpu.cpp:60) and `>` (CUDA
    reference, nms.cu:60) give different answers."""
    box = np.array([
        [0, 0, 9, 9],          # area 100
        [0, 0, 6, 9],          # inter 70, union 100 -> 0.7
        [100, 0, 109, 9],
        [100, 0, 104, 9],      # 0.5
        [200, 0, 209, 9],
        [200, 0, 202, 9],      # 0.3
        [300, 300, 330, 330],  # isolated
    ], np.float32)
    scores = np.linspace(0.9, 0.3, len(box)).astype(np.float32)
    return box, scores


NMS_SIZES = (1, 64, 65, 1000, 6000, 12000)
NMS_THRESHOLDS = (0.7, 0.3)


def rpn_case(A=9, b=2):
    """seeded RPN outputs at the 38x63 feature size"""
    H, W = FEAT_H, FEAT_W
    rs = np.random.RandomState(701)
    fg = rs.uniform(0.0, 1.0, (b, A, H, W)).astype(np.float32)
    prob = np.concatenate([1 - fg, fg], 1)
    deltas = (rs.standard_normal((b, 4 * A, H, W)) * 0.25).astype(np.float32)
    im_info = np.array([[600, 1000, 1.0]] * b, np.float32)
    return prob, deltas, im_info


def gt_case(b=2, G=20, n=3, seed=702):
    rs = np.random.RandomState(seed)
    gt = np.zeros((b, G, 5), np.float32)
    for i in range(b):
        for k in range(n):
            w, h = rs.uniform(64, 400, 2)
            x1, y1 = rs.uniform(0, 1000 - w), rs.uniform(0, 600 - h)
            gt[i, k] = (x1, y1, x1 + w, y1 + h, 1.0)
    return gt
