"""oracle/gen_golden.py -- TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists.

Imports the UNMODIFIED reference (oracle/ref_import.py), feeds it the seeded inputs of
oracle/cases.py and the deterministic weights of oracle/ait_ref.make_ait_state_dict, and
writes the reference's OUTPUTS (data only) to tests/golden/*.npz.

    python -m oracle.gen_golden            # all groups
    python -m oracle.gen_golden g4 g5      # some groups

Groups:
  g1  PositionalEncoding table                         (system/Models.py:26-51)
  g2  SHBlock / ScaledDotProductAttention / MultiHeadAttention{none,pad49,causal} /
      PositionwiseFeedForward at bp=2, outputs + input/weight grads (system/SubLayers.py)
  g3  Transformer (2,3) and (1,128) outputs; (1,2) output + grads wrt inputs and all 46 params
  g4  model._C.roi_align_forward (csrc/cpu/ROIAlign_cpu.cpp)
  g5  model._C.nms kept indices (csrc/cpu/nms_cpu.cpp), incl. exact ovr == thr ties
  g6  generate_anchors tables + shifted-grid checksum (rpn/generate_anchors.py)
  g7+ detector-level groups are appended by oracle/gen_golden_detector.py
  g15 Transformer in train() mode (dropout p = 0.1 ON) at (1,2): the keep decisions its ten nn.Dropout modules drew
      (recorded by forward hooks, bit-packed), output, grads wrt inputs and all 46 params
"""
import os
import sys

import numpy as np
import torch

from . import ait_ref, cases, ref_import
from .digest import pack, seeded

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _save(name, d):
    os.makedirs(GOLDEN, exist_ok=True)
    d["_meta/torch"] = np.asarray(torch.__version__)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez(path, **d)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def g1():
    from model.system.Models import PositionalEncoding
    out = {}
    for n_pos, d in ((64, 512), (200, 64)):
        out["pos_table_%d_%d" % (n_pos, d)] = PositionalEncoding(d, n_position=n_pos).pos_table[0].numpy()
    _save("g1_pos_table", out)


def _grads(y, wrt, cot):
    g = torch.autograd.grad(y, wrt, grad_outputs=cot, allow_unused=False)
    return g


def g2():
    from model.system.Modules import ScaledDotProductAttention
    from model.system.SubLayers import MultiHeadAttention, PositionwiseFeedForward, SHBlock
    out = {}
    bp, T, d, H, dv = 2, 64, 512, 8, 64
    sd = ait_ref.make_ait_state_dict(seed=2)

    # SHBlock
    sh = SHBlock(n_head=H, d_v=dv).eval()
    pre = "encoder.layer_stack.0.slf_attn."
    sh.load_state_dict({"sk.weight": sd[pre + "sh.sk.weight"], "sk.bias": sd[pre + "sh.sk.bias"]})
    x = torch.from_numpy(seeded(201, (bp, H, T, dv))).requires_grad_(True)
    y = sh(x)
    cot = torch.from_numpy(seeded(202, tuple(y.shape)))
    gx, gw, gb = _grads(y, [x, sh.sk.weight, sh.sk.bias], cot)
    for k, v in (("y", y), ("gx", gx), ("gw", gw), ("gb", gb)):
        pack("shblock/" + k, v, out)

    # ScaledDotProductAttention under the three masks
    src_mask, trg_mask = ait_ref.build_masks(bp, 49, T)
    masks = {"none": None, "pad49": src_mask.unsqueeze(1), "causal": trg_mask.unsqueeze(1)}
    sdpa = ScaledDotProductAttention(temperature=dv ** 0.5).eval()
    for mname, m in masks.items():
        q = torch.from_numpy(seeded(211, (bp, H, T, dv))).requires_grad_(True)
        k = torch.from_numpy(seeded(212, (bp, H, T, dv))).requires_grad_(True)
        v = torch.from_numpy(seeded(213, (bp, H, T, dv))).requires_grad_(True)
        o, attn = sdpa(q, k, v, mask=m)
        cot = torch.from_numpy(seeded(214, tuple(o.shape)))
        gq, gk, gv = _grads(o, [q, k, v], cot)
        for kk, vv in (("o", o), ("attn", attn), ("gq", gq), ("gk", gk), ("gv", gv)):
            pack("sdpa_%s/%s" % (mname, kk), vv, out)

    # MultiHeadAttention: self-attention under each mask + a cross-attention call
    for mname, m in (("none", None), ("pad49", src_mask), ("causal", trg_mask), ("cross_pad49", src_mask)):
        mha = MultiHeadAttention(H, d, dv, dv, dropout=0.1).eval()
        mha.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
        xq = torch.from_numpy(seeded(221, (bp, T, d))).requires_grad_(True)
        if mname.startswith("cross"):
            xk = torch.from_numpy(seeded(222, (bp, T, d))).requires_grad_(True)
            y, _ = mha(xq, xk, xk, mask=m)
        else:
            xk = None
            y, _ = mha(xq, xq, xq, mask=m)
        cot = torch.from_numpy(seeded(223, tuple(y.shape)))
        params = dict(mha.named_parameters())
        wrt = [xq] + ([xk] if xk is not None else []) + list(params.values())
        gs = _grads(y, wrt, cot)
        pack("mha_%s/y" % mname, y, out)
        pack("mha_%s/gx" % mname, gs[0], out)
        off = 1
        if xk is not None:
            pack("mha_%s/gkv" % mname, gs[1], out)
            off = 2
        for (pn, _), g in zip(params.items(), gs[off:]):
            pack("mha_%s/g_%s" % (mname, pn), g, out)

    # PositionwiseFeedForward
    fp = "encoder.layer_stack.0.pos_ffn."
    ffn = PositionwiseFeedForward(d, 2048, dropout=0.1).eval()
    ffn.load_state_dict({k[len(fp):]: v for k, v in sd.items() if k.startswith(fp)})
    x = torch.from_numpy(seeded(231, (bp, T, d))).requires_grad_(True)
    y = ffn(x.clone())          # the reference adds the residual in place (SubLayers.py:183)
    cot = torch.from_numpy(seeded(232, tuple(y.shape)))
    params = dict(ffn.named_parameters())
    gs = _grads(y, [x] + list(params.values()), cot)
    pack("ffn/y", y, out)
    pack("ffn/gx", gs[0], out)
    for (pn, _), g in zip(params.items(), gs[1:]):
        pack("ffn/g_" + pn, g, out)
    _save("g2_sublayers", out)


def _ref_transformer(sd):
    from model.system.Models import Transformer
    t = Transformer(d_k=64, d_v=64, d_model=512, d_word_vec=512, d_inner=2048,
                    n_position=64, n_layers=1, n_head=8, dropout=0.1)
    t.load_state_dict(sd, strict=True)
    return t.eval()


def g3():
    """Forward at (bs,P)=(2,3) and cfg1 (1,128); forward+backward at (1,2).

    ReLU makes the gradient discontinuous where a hidden pre-activation crosses zero, and with
    ~1e5..1e6 hidden activations per call one of them is routinely within fp32 rounding noise
    (~3e-7) of zero: its sign, and with it ~1e-3 of the gradient norm, then depends on the
    summation order of whoever computes it (observed: the reference itself at 1 vs 8 CPU
    threads).  The backward fixture therefore searches for an input seed whose smallest
    |pre-activation| in the REFERENCE's own fp32 forward is >= 5e-6, so that every correct
    implementation sees the same ReLU mask."""
    out = {}
    sd = ait_ref.make_ait_state_dict(seed=3)
    t = _ref_transformer(sd)
    with torch.no_grad():
        y = t(x_props=torch.from_numpy(seeded(301, (6, 1024, 7, 7))),
              x_query=torch.from_numpy(seeded(302, (2, 1024, 8, 8))))
    pack("t23/y", y, out)
    pre_min = []
    hooks = [m.register_forward_hook(lambda mod, i, o: pre_min.append(float(o.abs().min())))
             for m in (t.encoder.layer_stack[0].pos_ffn.w_1, t.decoder.layer_stack[0].pos_ffn.w_1)]
    bs, P = 1, 2
    for seed in range(3200, 3400):
        del pre_min[:]
        xp = torch.from_numpy(seeded(seed, (bs * P, 1024, 7, 7))).requires_grad_(True)
        xq = torch.from_numpy(seeded(seed + 1000, (bs, 1024, 8, 8))).requires_grad_(True)
        y = t(x_props=xp, x_query=xq)
        if min(pre_min) >= 5e-6:
            break
    else:
        raise RuntimeError("no seed with a safe ReLU margin found")
    for h in hooks:
        h.remove()
    print("g3 backward fixture: seed", seed, "min |pre-activation|", min(pre_min))
    out["t12/seed"] = np.asarray(seed)
    out["t12/relu_margin"] = np.asarray(min(pre_min))
    cot = torch.from_numpy(seeded(303, tuple(y.shape)))
    params = dict(t.named_parameters())
    gs = _grads(y, [xp, xq] + list(params.values()), cot)
    pack("t12/y", y, out)
    pack("t12/g_x_props", gs[0], out)
    pack("t12/g_x_query", gs[1], out)
    for (pn, _), g in zip(params.items(), gs[2:]):
        pack("t12/g_" + pn, g, out)
    # cfg1 shape (1,128), forward only
    with torch.no_grad():
        y = t(x_props=torch.from_numpy(seeded(311, (128, 1024, 7, 7))),
              x_query=torch.from_numpy(seeded(312, (1, 1024, 8, 8))))
    pack("t1_128/y", y, out)
    _save("g3_transformer", out)


# reference module path of each dropout site -> key of oracle.ait_ref.DROPOUT_SITES
_DROPOUT_MODULES = {
    "encoder.dropout": "enc_pro",                                          # Models.py:98
    "encoder.layer_stack.0.slf_attn.attention.dropout": "enc_slf_attn",    # Modules.py:24
    "encoder.layer_stack.0.slf_attn.dropout": "enc_slf_fc",                # SubLayers.py:98
    "encoder.layer_stack.0.pos_ffn.dropout": "enc_ffn",                    # SubLayers.py:184
    "decoder.dropout": "dec_pro",                                          # Models.py:155
    "decoder.layer_stack.0.slf_attn.attention.dropout": "dec_slf_attn",
    "decoder.layer_stack.0.slf_attn.dropout": "dec_slf_fc",
    "decoder.layer_stack.0.enc_attn.attention.dropout": "dec_enc_attn",
    "decoder.layer_stack.0.enc_attn.dropout": "dec_enc_fc",
    "decoder.layer_stack.0.pos_ffn.dropout": "dec_ffn",
}


def g15():
    """The reference in TRAIN mode (its ten nn.Dropout modules at p = 0.1 active, torch's own generator), with a forward
    hook on every one of them noting which elements it kept: the fixture that pins oracle/ait_ref.transformer_forward's
    `masks` argument -- placement of the ten sites and the 1 / (1 - p) factor -- to the reference's own arithmetic.
    (An element whose input is exactly 0 -- a masked probability -- reads as kept; its decision cannot matter.)
    Same ReLU-margin seed search as g3."""
    out = {}
    sd = ait_ref.make_ait_state_dict(seed=3)
    t = _ref_transformer(sd).train()
    mods = dict(t.named_modules())
    assert all(isinstance(mods[k], torch.nn.Dropout) and mods[k].p == 0.1 for k in _DROPOUT_MODULES)
    assert sum(isinstance(m, torch.nn.Dropout) for m in mods.values()) == len(_DROPOUT_MODULES)
    keeps, pre_min = {}, []

    def note(key):
        def hook(mod, inp, o):
            assert key not in keeps, key
            keeps[key] = ((o != 0) | (inp[0] == 0)).detach()
        return hook
    hooks = [mods[k].register_forward_hook(note(v)) for k, v in _DROPOUT_MODULES.items()]
    hooks += [m.register_forward_hook(lambda mod, i, o: pre_min.append(float(o.abs().min())))
              for m in (t.encoder.layer_stack[0].pos_ffn.w_1, t.decoder.layer_stack[0].pos_ffn.w_1)]
    bs, P = 1, 2
    for seed in range(15200, 15400):
        del pre_min[:]
        keeps.clear()
        xp = torch.from_numpy(seeded(seed, (bs * P, 1024, 7, 7))).requires_grad_(True)
        xq = torch.from_numpy(seeded(seed + 1000, (bs, 1024, 8, 8))).requires_grad_(True)
        torch.manual_seed(seed)
        y = t(x_props=xp, x_query=xq)
        if min(pre_min) >= 5e-6:
            break
    else:
        raise RuntimeError("no seed with a safe ReLU margin found")
    for h in hooks:
        h.remove()
    print("g15 fixture: seed", seed, "min |pre-activation|", min(pre_min))
    out["seed"] = np.asarray(seed)
    out["p"] = np.asarray(0.1)
    out["relu_margin"] = np.asarray(min(pre_min))
    for k in ait_ref.DROPOUT_SITES:
        m = keeps[k].numpy()
        out["keep/%s/shape" % k] = np.asarray(m.shape)
        out["keep/%s/bits" % k] = np.packbits(m.reshape(-1))
        out["keep/%s/kept_fraction" % k] = np.asarray(m.mean())
    cot = torch.from_numpy(seeded(1503, tuple(y.shape)))
    params = dict(t.named_parameters())
    gs = _grads(y, [xp, xq] + list(params.values()), cot)
    pack("y", y, out)
    pack("g_x_props", gs[0], out)
    pack("g_x_query", gs[1], out)
    for (pn, _), g in zip(params.items(), gs[2:]):
        pack("g_" + pn, g, out)
    _save("g15_transformer_dropout", out)


def g4():
    import model
    feat, rois = cases.roi_align_case()
    y = model._C.roi_align_forward(torch.from_numpy(feat), torch.from_numpy(rois), 1.0 / 16.0, 7, 7, 0)
    out = {"y": y.numpy()}
    # a second call with a fixed sampling ratio (the API allows it; every cfg uses 0)
    y2 = model._C.roi_align_forward(torch.from_numpy(feat), torch.from_numpy(rois), 1.0 / 16.0, 7, 7, 2)
    out["y_sr2"] = y2.numpy()
    # realistic random RoIs at C=8
    feat_r = seeded(402, (2, 8, cases.FEAT_H, cases.FEAT_W))
    rois_r = cases.random_rois(403, 64, 2)
    out["y_rand"] = model._C.roi_align_forward(torch.from_numpy(feat_r), torch.from_numpy(rois_r),
                                               1.0 / 16.0, 7, 7, 0).numpy()
    _save("g4_roi_align", out)


def g5():
    import model
    out = {}
    for n in cases.NMS_SIZES:
        box, sc = cases.nms_boxes(500 + n, n)
        for thr in cases.NMS_THRESHOLDS:
            keep = model._C.nms(torch.from_numpy(box), torch.from_numpy(sc), thr)
            out["keep_n%d_t%02d" % (n, int(thr * 10))] = keep.numpy().astype(np.int64)
    box, sc = cases.nms_boxes(777, 2000, integer=True)
    out["keep_int2000_t07"] = model._C.nms(torch.from_numpy(box), torch.from_numpy(sc), 0.7).numpy()
    box, sc = cases.nms_tie_case()
    for thr in (0.7, 0.5, 0.3):
        out["keep_tie_t%02d" % int(thr * 10)] = model._C.nms(
            torch.from_numpy(box), torch.from_numpy(sc), thr).numpy().astype(np.int64)
    out["keep_empty"] = model._C.nms(torch.zeros(0, 4), torch.zeros(0), 0.7).numpy().astype(np.int64)
    _save("g5_nms", out)


def g6():
    from model.rpn.generate_anchors import generate_anchors
    out = {}
    for name, scales in (("voc", [8, 16, 32]), ("coco", [4, 8, 16, 32])):
        a = generate_anchors(scales=np.array(scales), ratios=np.array([0.5, 1, 2]))
        out["anchors_" + name] = a
        # shifted grid exactly as proposal_layer.py:82-95 builds it, at 38x63, stride 16
        sx, sy = np.meshgrid(np.arange(0, cases.FEAT_W) * 16, np.arange(0, cases.FEAT_H) * 16)
        shifts = np.ascontiguousarray(
            np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose())
        A = torch.from_numpy(a).float()
        S = torch.from_numpy(shifts).float()
        grid = (A.view(1, -1, 4) + S.view(-1, 1, 4)).view(-1, 4)
        out["grid_%s_shape" % name] = np.asarray(grid.shape)
        out["grid_%s_sum" % name] = grid.double().sum(0).numpy()
        out["grid_%s_rows" % name] = grid[[0, 1, 8, 9, 1000, 12345, grid.shape[0] - 1]].numpy()
    _save("g6_anchors", out)


GROUPS = {"g1": g1, "g2": g2, "g3": g3, "g4": g4, "g5": g5, "g6": g6, "g15": g15}


def main(argv):
    ref_import.setup()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    todo = argv or list(GROUPS)
    for g in todo:
        if g in GROUPS:
            GROUPS[g]()
        else:
            from . import gen_golden_detector
            gen_golden_detector.GROUPS[g]()


if __name__ == "__main__":
    main(sys.argv[1:])
