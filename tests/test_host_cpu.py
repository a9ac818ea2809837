"""Host-side logic of the product that needs no GPU: buffers, configuration, launch heuristics."""
import numpy as np
import torch


def test_positional_tables_equal_the_reference(golden):
    """the sinusoid table is built in float64 and cast (Models.py:33-45); the
    product's buffers must equal the reference's bit for bit (golden g1)."""
    from ait_amd.system import PositionalEncoding, Transformer
    g = golden("g1_pos_table")
    assert np.array_equal(PositionalEncoding(512, n_position=64).pos_table[0].numpy(), g["pos_table_64_512"])
    assert np.array_equal(PositionalEncoding(64, n_position=200).pos_table[0].numpy(), g["pos_table_200_64"])
    t = Transformer(d_k=64, d_v=64, d_model=512, d_word_vec=512, d_inner=2048, n_position=64,
                    n_layers=1, n_head=8, dropout=0.1)
    for coder in (t.encoder, t.decoder):
        assert np.array_equal(coder.position_enc.pos_table[0].numpy(), g["pos_table_64_512"])
    assert "encoder.position_enc.pos_table" in dict(t.named_buffers())
    assert "encoder.position_enc.pos_table" not in dict(t.named_parameters())


def test_split_k_heuristic_is_a_multiple_of_the_xcd_count():
    from ait_amd.system import _split_k
    for (m, n, k) in [(512, 2048, 76800), (2048, 512, 76800), (1536, 512, 76800), (512, 64, 76800),
                      (1024, 512, 58800), (512, 512, 256), (2048, 512, 19200)]:
        s = _split_k(m, n, k)
        assert s % 8 == 0 and 8 <= s <= 128
        assert k // s >= 256 or s == 8


def test_cfg_from_list_follows_the_reference_convention():
    from ait_amd import config
    saved = config.cfg.TRAIN.BATCH_SIZE
    try:
        config.cfg_from_list(['TRAIN.BATCH_SIZE', 300])
        assert config.cfg.TRAIN.BATCH_SIZE == 300 and config.cfg['TRAIN'].BATCH_SIZE == 300
        try:
            config.cfg_from_list(['TRAIN.NO_SUCH_KEY', 1])
            raise AssertionError("unknown key accepted")
        except KeyError:
            pass
    finally:
        config.cfg.TRAIN.BATCH_SIZE = saved


def test_product_refuses_cpu_tensors():
    """No CPU fallback: the hot path raises on a non-GPU tensor instead of computing somewhere else."""
    import pytest
    from ait_amd import _lib
    with pytest.raises(_lib.AitHipError):
        _lib.dev_ptr(torch.zeros(4))


def test_samplers_index_parity_on_cpu_tensors(golden):
    """The two training samplers (rpn/anchor_target_layer.py:128-175, rpn/proposal_target_layer_cascade.py:
    148-190) are host logic over tensor ops: on CPU tensors they must reproduce the reference's sampled
    indices under its NumPy RNG call order exactly as on the GPU (goldens g7 / g8)."""
    from oracle import cases
    from oracle.digest import compare
    from ait_amd.config import cfg
    from ait_amd.rpn import _AnchorTargetLayer, _ProposalTargetLayer
    g = golden("g8_target_layers")
    prob, deltas, info = cases.rpn_case()
    gt, nb = torch.from_numpy(cases.gt_case()), torch.tensor([3, 3])
    np.random.seed(3)
    labels, targets, w_in, w_out = _AnchorTargetLayer(16, [8, 16, 32], [0.5, 1, 2])(
        (torch.from_numpy(prob), gt, torch.from_numpy(info), nb))
    assert np.array_equal(labels.numpy().astype(np.int8), g["atl_labels"])
    for name, t in (("atl_targets", targets), ("atl_w_in", w_in), ("atl_w_out", w_out)):
        ok, msg = compare(name, t.contiguous(), g, 1e-5, 1e-5)
        assert ok, msg
    rois = torch.from_numpy(golden("g7_proposal_layer")["rois_TRAIN"])
    saved = cfg.TRAIN.BATCH_SIZE
    try:
        for P in (128, 300):
            cfg.TRAIN.BATCH_SIZE = P
            r, lab, tg, wi, wo = _ProposalTargetLayer(2)(rois, gt, nb)
            assert np.array_equal(r.numpy(), g["ptl%d_rois" % P])
            assert np.array_equal(lab.numpy(), g["ptl%d_labels" % P])
            np.testing.assert_allclose(tg.numpy(), g["ptl%d_targets" % P], rtol=1e-5, atol=1e-5)
            assert np.array_equal(wi.numpy(), g["ptl%d_w_in" % P])
    finally:
        cfg.TRAIN.BATCH_SIZE = saved


def test_anchor_target_inside_set_follows_the_image_size():
    """anchor_target_layer.py:84-88 recomputes the inside-image anchor set from im_info on every call, and about 16
    image widths share one feature width: one layer object fed two image sizes that map to the same (H, W) must
    label each with ITS inside set -- also when the host's prediction of the size (im_hw_hint) is wrong, and the
    result must equal a fresh layer's."""
    from oracle import cases
    from ait_amd.rpn import _AnchorTargetLayer
    prob, _, _ = cases.rpn_case()
    H, W = prob.shape[2], prob.shape[3]
    gt, nb = torch.from_numpy(cases.gt_case()), torch.tensor([3, 3])
    score = torch.from_numpy(prob)
    sizes = [(16 * H - 1, 16 * W - 1), (16 * H - 9, 16 * W - 13), (16 * H - 1, 16 * W - 1)]

    def run(layer, hw, hint):
        np.random.seed(3)
        info = torch.tensor([[hw[0], hw[1], 1.0]] * gt.size(0))
        if hint is not None:
            layer.begin(gt, info, H, W, im_hw_hint=hint)
        return [t.clone() for t in layer((score, gt, info, nb))]

    shared = _AnchorTargetLayer(16, [8, 16, 32], [0.5, 1, 2])
    outs = []
    for i, hw in enumerate(sizes):
        fresh = run(_AnchorTargetLayer(16, [8, 16, 32], [0.5, 1, 2]), hw, None)
        # the shared layer: no hint, the right hint, a WRONG hint (the previous image's size)
        for hint in (None, hw, sizes[i - 1]):
            got = run(shared, hw, hint)
            for a, b in zip(got, fresh):
                assert torch.equal(a, b), "stale inside-anchor set for image size %r (hint %r)" % (hw, hint)
        outs.append(fresh)
    # the two sizes really differ in their border anchors (otherwise the test pins nothing)
    assert not torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][0], outs[2][0])


def test_row_decomposition_by_corrected_f32_quotients_is_exact():
    """gemm_f32_impl.h div_small (maps of any size in the implicit-GEMM convolutions): q = int(float(n) * (1/d)) is off
    by at most one for 0 <= n < 2^24, and the remainder test puts it right -- restated in numpy float32 (same
    rounding as the device: round-to-nearest multiply, truncating conversion) and checked against integer division
    for the divisors the C4 maps produce and for adversarial ones."""
    import numpy as np
    rs = np.random.RandomState(4)
    divisors = [63, 38 * 63, 125, 75 * 125, 250, 150 * 250, 3, 7, 8191, 65535, 1 << 12, (1 << 16) + 1, 9576, 16777215]
    for d in divisors:
        inv = np.float32(1.0) / np.float32(d)
        n = np.concatenate([rs.randint(0, 1 << 24, 200000), np.arange(0, min(1 << 24, 40 * d), max(1, d // 7)),
                            np.array([0, 1, d - 1, d, d + 1, (1 << 24) - 1]),
                            (np.arange(1, 4000) * d - 1) % (1 << 24), (np.arange(1, 4000) * d) % (1 << 24)]).astype(np.int64)
        q = (n.astype(np.float32) * inv).astype(np.int64)          # truncation: values are non-negative
        assert np.abs(q - n // d).max() <= 1
        rem = n - q * d
        lo = rem < 0
        q, rem = np.where(lo, q - 1, q), np.where(lo, rem + d, rem)
        hi = rem >= d
        q, rem = np.where(hi, q + 1, q), np.where(hi, rem - d, rem)
        assert np.array_equal(q, n // d) and np.array_equal(rem, n % d), d


def test_three_way_bf16_split_is_exact_and_six_terms_reach_f32_accuracy():
    """gemm_f32_impl.h split2 / mfma_split restated in numpy, both forms: every plane by truncation (round 3's) and the
    product form since round 4 (KNOB_RNE): h = bf16(x) rounded to NEAREST, m = the top 8 bits of the exact remainder
    r = x - h, l = r - m.  (1) x == h + m + l exactly and every plane is a bf16 value, for random, tiny, huge and
    all-bits-set inputs; (2) the six partial products kept (hh, hm, mh, hl, lh, mm) reproduce a * b to
    <= 2^-21 |a b| (truncation) / <= 2^-23 |a b| (nearest); (3) on POSITIVE inputs the truncated form's m and l planes are
    all positive (the one-signed planes that drift on long same-signed sums, profiles/r04_split_bias.txt) while the
    nearest form's are zero-mean."""
    import numpy as np
    rs = np.random.RandomState(7)

    def rne_bf16(x):
        u = x.view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)

    def trunc_bf16(x):
        return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)

    def split(x, first):
        h = first(x)
        r = (x - h).astype(np.float32)
        m = trunc_bf16(r)
        l = (r - m).astype(np.float32)
        return h, m, l

    x = np.concatenate([rs.randn(200000).astype(np.float32) * np.exp(rs.uniform(-30, 30, 200000)).astype(np.float32),
                        np.array([0.0, 1.0, -1.0, 16777215.0, 1.0 / 3.0, 3e38, -3e38, 1e-30, 2.0 ** -100, 1.9999999],
                                 dtype=np.float32)])
    y = rs.permutation(x)
    f = np.float64
    worst = {}
    for name, first in (("nearest", rne_bf16), ("truncate", trunc_bf16)):
        h, m, l = split(x, first)
        assert np.array_equal(trunc_bf16(l), l)                                 # l needs no more than bf16's 8 bits
        assert np.array_equal(h.astype(f) + m.astype(f) + l.astype(f), x.astype(f))
        hy, my, ly = split(y, first)
        six = h.astype(f) * hy + h.astype(f) * my + m.astype(f) * hy + h.astype(f) * ly + l.astype(f) * hy + m.astype(f) * my
        exact = x.astype(f) * y.astype(f)
        ok = np.isfinite(exact) & (np.abs(exact) > 1e-300) & (np.abs(exact) < 1e300)
        worst[name] = float((np.abs(six - exact)[ok] / np.abs(exact)[ok]).max())
        # magnitudes of the planes
        nz = x != 0
        lim_m, lim_l = (2.0 ** -8, 2.0 ** -16) if name == "nearest" else (2.0 ** -7, 2.0 ** -15)
        assert (np.abs(m[nz].astype(f)) <= lim_m * np.abs(x[nz].astype(f))).all()
        assert (np.abs(l[nz].astype(f)) <= lim_l * np.abs(x[nz].astype(f))).all()
    assert worst["nearest"] <= 2.0 ** -23
    assert 2.0 ** -23 < worst["truncate"] <= 2.0 ** -21
    pos = (rs.rand(200000).astype(np.float32) + 0.5)
    _, m_t, l_t = split(pos, trunc_bf16)
    _, m_n, l_n = split(pos, rne_bf16)
    assert (m_t >= 0).all() and (l_t >= 0).all() and float(m_t.mean()) > 1e-3
    assert abs(float(m_n.astype(f).mean())) < 2e-5 and 0.4 < float((m_n > 0).mean()) < 0.6 and 0.4 < float((l_n > 0).mean()) < 0.6


def test_fused_heads_path_defers_to_torch_for_hooks_it_cannot_call():
    """ait_amd.faster_rcnn._only_plain_forward_hooks: the fused heads kernel calls plain forward hooks on the
    two head modules by hand; anything else -- pre-hooks, kwargs / always-call hooks, backward hooks, a hook on a child of
    RCNN_cls_score -- must send the heads through their modules"""
    import torch.nn as nn
    from ait_amd.faster_rcnn import _only_plain_forward_hooks as plain
    mk = lambda: nn.Sequential(nn.Linear(4, 3), nn.Linear(3, 2))
    m = mk()
    assert plain(m)
    h = m.register_forward_hook(lambda mod, a, o: None)
    assert plain(m)
    h.remove()
    for reg in (lambda m: m.register_forward_pre_hook(lambda mod, a: None),
                lambda m: m.register_forward_hook(lambda mod, a, k, o: None, with_kwargs=True),
                lambda m: m.register_forward_hook(lambda mod, a, o: None, always_call=True),
                lambda m: m.register_full_backward_hook(lambda mod, gi, go: None),
                lambda m: m[1].register_forward_hook(lambda mod, a, o: None)):
        m = mk()
        h = reg(m)
        assert not plain(m)
        h.remove()
        assert plain(m)
