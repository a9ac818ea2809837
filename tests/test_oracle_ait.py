"""Pins oracle/ait_ref.py (the CPU restatement of the AIT transformer) against the golden
vectors recorded from the imported reference (tests/golden/g1..g3) -- CPU only.

Tolerance: both sides are fp32 torch-CPU programs that differ only in summation order, so
elementwise |err| <= ATOL + RTOL*|ref| with RTOL = 1e-4 (north_star's logit tolerance) and a
small absolute floor for values that are differences of O(1) terms."""
import numpy as np
import pytest
import torch

from oracle import ait_ref
from oracle.digest import compare, seeded

RTOL, ATOL = 1e-4, 2e-5
GRTOL, GATOL = 1e-4, 2e-4     # gradients accumulate over bp*64 rows


def _check(prefix, t, g, rtol=RTOL, atol=ATOL):
    ok, msg = compare(prefix, t, g, rtol, atol)
    assert ok, msg


def test_pos_table_bit_exact(golden):
    g = golden("g1_pos_table")
    assert np.array_equal(ait_ref.pos_table(64, 512)[0].numpy(), g["pos_table_64_512"])
    assert np.array_equal(ait_ref.pos_table(200, 64)[0].numpy(), g["pos_table_200_64"])


def test_selective_heads(golden):
    g = golden("g2_sublayers")
    sd = ait_ref.make_ait_state_dict(seed=2)
    pre = "encoder.layer_stack.0.slf_attn."
    w = sd[pre + "sh.sk.weight"].clone().requires_grad_(True)
    b = sd[pre + "sh.sk.bias"].clone().requires_grad_(True)
    x = torch.from_numpy(seeded(201, (2, 8, 64, 64))).requires_grad_(True)
    y = ait_ref.selective_heads(x, w, b)
    cot = torch.from_numpy(seeded(202, tuple(y.shape)))
    gx, gw, gb = torch.autograd.grad(y, [x, w, b], cot)
    _check("shblock/y", y, g)
    _check("shblock/gx", gx, g, GRTOL, GATOL)
    _check("shblock/gw", gw, g, GRTOL, GATOL)
    _check("shblock/gb", gb, g, GRTOL, GATOL)


@pytest.mark.parametrize("mname", ["none", "pad49", "causal"])
def test_scaled_dot_attention(golden, mname):
    g = golden("g2_sublayers")
    src, trg = ait_ref.build_masks(2, 49, 64)
    m = {"none": None, "pad49": src.unsqueeze(1), "causal": trg.unsqueeze(1)}[mname]
    q, k, v = (torch.from_numpy(seeded(s, (2, 8, 64, 64))).requires_grad_(True) for s in (211, 212, 213))
    o, attn = ait_ref.scaled_dot_attention(q, k, v, m, 8.0)
    cot = torch.from_numpy(seeded(214, tuple(o.shape)))
    gq, gk, gv = torch.autograd.grad(o, [q, k, v], cot)
    for name, t in (("o", o), ("attn", attn)):
        _check("sdpa_%s/%s" % (mname, name), t, g)
    for name, t in (("gq", gq), ("gk", gk), ("gv", gv)):
        _check("sdpa_%s/%s" % (mname, name), t, g, GRTOL, GATOL)


@pytest.mark.parametrize("mname", ["none", "pad49", "causal", "cross_pad49"])
def test_multi_head_attention(golden, mname):
    g = golden("g2_sublayers")
    pre = "encoder.layer_stack.0.slf_attn."
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point)
          for k, v in ait_ref.make_ait_state_dict(seed=2).items() if k.startswith(pre)}
    src, trg = ait_ref.build_masks(2, 49, 64)
    m = {"none": None, "pad49": src, "causal": trg, "cross_pad49": src}[mname]
    xq = torch.from_numpy(seeded(221, (2, 64, 512))).requires_grad_(True)
    wrt = [xq]
    if mname.startswith("cross"):
        xk = torch.from_numpy(seeded(222, (2, 64, 512))).requires_grad_(True)
        wrt.append(xk)
        y, _ = ait_ref.multi_head_attention(sd, pre, xq, xk, xk, m)
    else:
        y, _ = ait_ref.multi_head_attention(sd, pre, xq, xq, xq, m)
    cot = torch.from_numpy(seeded(223, tuple(y.shape)))
    names = sorted(sd)
    gs = torch.autograd.grad(y, wrt + [sd[n] for n in names], cot)
    _check("mha_%s/y" % mname, y, g)
    _check("mha_%s/gx" % mname, gs[0], g, GRTOL, GATOL)
    if mname.startswith("cross"):
        _check("mha_%s/gkv" % mname, gs[1], g, GRTOL, GATOL)
    for n, gr in zip(names, gs[len(wrt):]):
        _check("mha_%s/g_%s" % (mname, n[len(pre):]), gr, g, GRTOL, GATOL)


def test_feed_forward(golden):
    g = golden("g2_sublayers")
    pre = "encoder.layer_stack.0.pos_ffn."
    sd = {k: v.clone().requires_grad_(True)
          for k, v in ait_ref.make_ait_state_dict(seed=2).items() if k.startswith(pre)}
    x = torch.from_numpy(seeded(231, (2, 64, 512))).requires_grad_(True)
    y = ait_ref.feed_forward(sd, pre, x)
    cot = torch.from_numpy(seeded(232, tuple(y.shape)))
    names = sorted(sd)
    gs = torch.autograd.grad(y, [x] + [sd[n] for n in names], cot)
    _check("ffn/y", y, g)
    _check("ffn/gx", gs[0], g, GRTOL, GATOL)
    for n, gr in zip(names, gs[1:]):
        _check("ffn/g_" + n[len(pre):], gr, g, GRTOL, GATOL)


def test_transformer_forward_and_grads(golden):
    g = golden("g3_transformer")
    sd = {k: (v.clone().requires_grad_(True) if "pos_table" not in k else v)
          for k, v in ait_ref.make_ait_state_dict(seed=3).items()}
    with torch.no_grad():
        y23 = ait_ref.transformer_forward(sd, torch.from_numpy(seeded(301, (6, 1024, 7, 7))),
                                          torch.from_numpy(seeded(302, (2, 1024, 8, 8))))
    assert tuple(y23.shape) == (6, 1024, 8, 8)            # adaptive_image_transformer.py:35
    _check("t23/y", y23, g)
    # backward fixture at (bs,P)=(1,2): seed chosen by gen_golden so that no ReLU pre-activation
    # is within rounding noise of zero (see oracle/gen_golden.py g3)
    seed = int(g["t12/seed"])
    assert float(g["t12/relu_margin"]) >= 5e-6
    xp = torch.from_numpy(seeded(seed, (2, 1024, 7, 7))).requires_grad_(True)
    xq = torch.from_numpy(seeded(seed + 1000, (1, 1024, 8, 8))).requires_grad_(True)
    y = ait_ref.transformer_forward(sd, xp, xq)
    cot = torch.from_numpy(seeded(303, tuple(y.shape)))
    names = [k for k in sd if "pos_table" not in k]
    assert len(names) == 46 and sum(sd[n].numel() for n in names) == 8338944
    gs = torch.autograd.grad(y, [xp, xq] + [sd[n] for n in names], cot)
    _check("t12/y", y, g)
    _check("t12/g_x_props", gs[0], g, GRTOL, GATOL)
    _check("t12/g_x_query", gs[1], g, GRTOL, GATOL)
    for n, gr in zip(names, gs[2:]):
        _check("t12/g_" + n, gr, g, GRTOL, GATOL)


def golden_dropout_masks(g):
    """the reference's own keep decisions of g15 as the oracle's factor tensors"""
    inv = np.float32(1.0) / (np.float32(1.0) - np.float32(float(g["p"])))
    masks = {}
    for k in ait_ref.DROPOUT_SITES:
        shape = tuple(int(v) for v in g["keep/%s/shape" % k])
        keep = np.unpackbits(g["keep/%s/bits" % k])[:int(np.prod(shape))].reshape(shape)
        masks[k] = torch.from_numpy(keep.astype(np.float32) * inv)
    return masks


def test_transformer_train_mode_dropout_masks_vs_reference(golden):
    """Dropout ON (p = 0.1 at all ten sites, Modules.py:24, SubLayers.py:98,184, Models.py:98,155): the oracle, given
    the decisions the reference's own nn.Dropout modules drew in train() mode (g15, recorded by forward hooks), reproduces
    the reference's output and every gradient -- which pins WHERE the ten sites sit and the 1 / (1 - p) factor."""
    g = golden("g15_transformer_dropout")
    assert float(g["relu_margin"]) >= 5e-6
    masks = golden_dropout_masks(g)
    for k in ait_ref.DROPOUT_SITES:           # the fixture really is a p = 0.1 draw at every site
        frac = float(g["keep/%s/kept_fraction" % k])
        assert 0.88 < frac < 0.97, (k, frac)       # (attention sites: masked probabilities read as kept)
    sd = {k: (v.clone().requires_grad_(True) if "pos_table" not in k else v)
          for k, v in ait_ref.make_ait_state_dict(seed=3).items()}
    seed = int(g["seed"])
    xp = torch.from_numpy(seeded(seed, (2, 1024, 7, 7))).requires_grad_(True)
    xq = torch.from_numpy(seeded(seed + 1000, (1, 1024, 8, 8))).requires_grad_(True)
    y = ait_ref.transformer_forward(sd, xp, xq, masks=masks)
    cot = torch.from_numpy(seeded(1503, tuple(y.shape)))
    names = [k for k in sd if "pos_table" not in k]
    gs = torch.autograd.grad(y, [xp, xq] + [sd[n] for n in names], cot)
    _check("y", y, g)
    _check("g_x_props", gs[0], g, GRTOL, GATOL)
    _check("g_x_query", gs[1], g, GRTOL, GATOL)
    for n, gr in zip(names, gs[2:]):
        _check("g_" + n, gr, g, GRTOL, GATOL)
    # and the masks matter: without them the same inputs give another output
    with torch.no_grad():
        y0 = ait_ref.transformer_forward(sd, xp, xq)
    assert float((y0 - y).abs().max()) > 1e-2


def test_transformer_cfg1_shape(golden):
    """cfg1 of BASELINE.json: one pair, 128 proposals, CPU forward."""
    g = golden("g3_transformer")
    sd = ait_ref.make_ait_state_dict(seed=3)
    with torch.no_grad():
        y = ait_ref.transformer_forward(sd, torch.from_numpy(seeded(311, (128, 1024, 7, 7))),
                                        torch.from_numpy(seeded(312, (1, 1024, 8, 8))))
    _check("t1_128/y", y, g)


def test_padding_rows_are_not_inert_but_dead_after_encoder():
    """SURVEY.md hard parts: encoder rows 49..63 feed the SHBlock mean, yet clobbering
    enc_output[:, 49:] cannot change the result (masked as keys in cross-attention)."""
    sd = ait_ref.make_ait_state_dict(seed=5)
    xp = torch.from_numpy(seeded(51, (2, 1024, 7, 7)))
    xq = torch.from_numpy(seeded(52, (1, 1024, 8, 8)))
    with torch.no_grad():
        y, inter = ait_ref.transformer_forward(sd, xp, xq, return_intermediates=True)
    assert inter["enc"].shape == (2, 64, 512)
