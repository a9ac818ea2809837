"""GPU parity tests for the HIP RoIAlign / NMS kernels, called through the C ABI, against the
CPU oracle (oracle/native.c) and the reference's golden vectors."""
import numpy as np
import pytest
import torch

from oracle import cases, native
from oracle.digest import seeded

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_roi_align_fwd_golden(golden):
    from ait_amd.roi_layers import ROIAlign
    g = golden("g4_roi_align")
    feat, rois = cases.roi_align_case()
    y = ROIAlign((7, 7), 1.0 / 16.0, 0)(_dev(feat), _dev(rois)).cpu().numpy()
    # identical fp32 operation sequence (no FMA contraction) -> expected bit-exact; the stated
    # tolerance for the contract is 1e-6 relative
    np.testing.assert_allclose(y, g["y"], rtol=1e-6, atol=1e-7)
    y2 = ROIAlign((7, 7), 1.0 / 16.0, 2)(_dev(feat), _dev(rois)).cpu().numpy()
    np.testing.assert_allclose(y2, g["y_sr2"], rtol=1e-6, atol=1e-7)
    feat_r = seeded(402, (2, 8, cases.FEAT_H, cases.FEAT_W))
    rois_r = cases.random_rois(403, 64, 2)
    y3 = ROIAlign((7, 7), 1.0 / 16.0, 0)(_dev(feat_r), _dev(rois_r)).cpu().numpy()
    np.testing.assert_allclose(y3, g["y_rand"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("C,n", [(1024, 300), (40, 17), (3, 1)])
def test_roi_align_fwd_bwd_vs_oracle(C, n):
    from ait_amd.roi_layers import roi_align
    feat = seeded(7, (2, C, cases.FEAT_H, cases.FEAT_W))
    rois = cases.random_rois(8, n, 2)
    x = _dev(feat).requires_grad_(True)
    y = roi_align(x, _dev(rois), (7, 7), 1.0 / 16.0, 0)
    want = native.roi_align_fwd(feat, rois)
    np.testing.assert_allclose(y.detach().cpu().numpy(), want, rtol=1e-6, atol=1e-7)
    g = seeded(9, tuple(y.shape))
    y.backward(_dev(g))
    gwant = native.roi_align_bwd(g, rois, feat.shape)
    # atomics: order-dependent last-ulp differences; sums of up to ~n*gh*gw terms
    np.testing.assert_allclose(x.grad.cpu().numpy(), gwant, rtol=1e-4, atol=1e-4)


def test_roi_align_channels_last_golden_and_oracle(golden):
    """ait_roi_align_nhwc_fwd/bwd (channels-last features, token-major result, separable weights):
    the same operator up to fp32 summation order -- 1e-5 relative against the reference's golden
    vectors (edge cases: inside, touching borders, < 1 px, whole image, out of range) and against
    the oracle at C = 1024; the result's memory is the [K, 49, C] token matrix."""
    from ait_amd.roi_layers import ROIAlign
    g = golden("g4_roi_align")
    feat, rois = cases.roi_align_case()
    op = ROIAlign((7, 7), 1.0 / 16.0, 0, channels_last=True)
    if feat.shape[1] % 4 == 0:
        y = op(_dev(feat), _dev(rois))
        assert y.permute(0, 2, 3, 1).is_contiguous()
        np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=1e-5, atol=1e-6)
        y2 = ROIAlign((7, 7), 1.0 / 16.0, 2, channels_last=True)(_dev(feat), _dev(rois)).cpu().numpy()
        np.testing.assert_allclose(y2, g["y_sr2"], rtol=1e-5, atol=1e-6)
    feat_r = seeded(402, (2, 8, cases.FEAT_H, cases.FEAT_W))
    rois_r = cases.random_rois(403, 64, 2)
    y3 = op(_dev(feat_r), _dev(rois_r)).cpu().numpy()
    np.testing.assert_allclose(y3, g["y_rand"], rtol=1e-5, atol=1e-6)
    for C, n in ((1024, 300), (40, 17), (2052, 5), (256, 33), (96, 9)):      # (tiled backward: C % 32 == 0, C <= 1024)
        feat = seeded(7, (2, C, cases.FEAT_H, cases.FEAT_W))
        rois = cases.random_rois(8, n, 2)
        x = _dev(feat).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = op(x, _dev(rois))
        want = native.roi_align_fwd(feat, rois)
        np.testing.assert_allclose(y.detach().cpu().numpy(), want, rtol=1e-5, atol=1e-6)
        gr = seeded(9, tuple(y.shape))
        y.backward(_dev(gr))
        gwant = native.roi_align_bwd(gr, rois, feat.shape)
        np.testing.assert_allclose(x.grad.cpu().numpy(), gwant, rtol=1e-4, atol=1e-4)
        # the gather backward is bitwise reproducible
        x2 = _dev(feat).requires_grad_(True)
        op(x2, _dev(rois)).backward(_dev(gr))
        assert torch.equal(x.grad, x2.grad)
    small = _dev(seeded(1, (1, 8, 10, 12)))
    empty = op(small, torch.zeros((0, 5), device="cuda"))
    assert tuple(empty.shape) == (0, 8, 7, 7)
    # a batch index outside [0, B): zeros out, no gradient in (as the NCHW operator)
    xs = small.clone().requires_grad_(True)
    yb = op(xs, torch.tensor([[3, 0, 0, 50, 50]], dtype=torch.float32, device="cuda"))
    assert float(yb.abs().sum()) == 0.0
    yb.backward(torch.ones_like(yb))
    assert float(xs.grad.abs().sum()) == 0.0


def test_roi_align_empty_and_bad_batch_index():
    from ait_amd.roi_layers import roi_align
    feat = _dev(seeded(1, (1, 8, 10, 12)))
    y = roi_align(feat, torch.zeros((0, 5), device="cuda"), (7, 7), 1.0 / 16.0, 0)
    assert tuple(y.shape) == (0, 8, 7, 7)
    rois = torch.tensor([[3, 0, 0, 50, 50]], dtype=torch.float32, device="cuda")
    assert float(roi_align(feat, rois, (7, 7), 1.0 / 16.0, 0).abs().sum()) == 0.0


def test_roi_align_requires_gpu_tensor():
    from ait_amd import _lib
    from ait_amd.roi_layers import roi_align
    with pytest.raises(_lib.AitHipError):
        roi_align(torch.zeros(1, 1, 4, 4), torch.zeros(1, 5), (7, 7), 1.0, 0)


@pytest.mark.parametrize("n", cases.NMS_SIZES)
@pytest.mark.parametrize("thr", cases.NMS_THRESHOLDS)
def test_nms_bit_exact_vs_golden(golden, n, thr):
    from ait_amd.roi_layers import nms, nms_sorted
    g = golden("g5_nms")
    box, sc = cases.nms_boxes(500 + n, n)
    want = g["keep_n%d_t%02d" % (n, int(thr * 10))]
    keep = nms(_dev(box), _dev(sc), thr)
    assert keep.dtype == torch.int64
    assert np.array_equal(keep.cpu().numpy(), want)
    # pre-sorted entry point with early exit after post_nms_topN survivors
    for topn in (0, 300, 2000):
        k, cnt = nms_sorted(_dev(box), thr, topn)
        cnt = int(cnt.item())
        ref = want if topn == 0 else want[:topn]
        assert cnt == len(ref)
        assert np.array_equal(k[:cnt].cpu().numpy(), ref)


def test_nms_ties_and_unsorted_scores(golden):
    from ait_amd.roi_layers import nms
    g = golden("g5_nms")
    box, sc = cases.nms_tie_case()
    for thr in (0.7, 0.5, 0.3):
        assert np.array_equal(nms(_dev(box), _dev(sc), thr).cpu().numpy(),
                              g["keep_tie_t%02d" % int(thr * 10)])
    box2, sc2 = cases.nms_boxes(777, 2000, integer=True)
    assert np.array_equal(nms(_dev(box2), _dev(sc2), 0.7).cpu().numpy(), g["keep_int2000_t07"])
    # scores in arbitrary order: survivors still reported as ascending original indices
    rs = np.random.RandomState(3)
    perm = rs.permutation(len(sc2))
    got = nms(_dev(box2[perm]), _dev(sc2[perm]), 0.7).cpu().numpy()
    want = native.nms(box2[perm], sc2[perm], 0.7)
    assert np.array_equal(got, want)
    assert nms(torch.zeros((0, 4), device="cuda"), torch.zeros((0,), device="cuda"), 0.7).numel() == 0


def test_nms_idempotent_at_full_size():
    """Size-independent property at the TRAIN pre-NMS size: NMS of the survivors keeps all."""
    from ait_amd.roi_layers import nms
    box, sc = cases.nms_boxes(4242, 12000)
    b, s = _dev(box), _dev(sc)
    keep = nms(b, s, 0.7)
    again = nms(b[keep], s[keep], 0.7)
    assert again.numel() == keep.numel()
    assert np.array_equal(again.cpu().numpy(), np.arange(keep.numel()))


def test_nms_batched_matches_per_image(golden):
    from ait_amd.roi_layers import nms_sorted_batched
    g = golden("g5_nms")
    for n in (65, 1000, 6000):
        boxes = np.stack([cases.nms_boxes(500 + n, n)[0], cases.nms_boxes(900 + n, n)[0],
                          cases.nms_boxes(500 + n, n)[0]])
        for topn in (0, 300):
            keep, cnt = nms_sorted_batched(_dev(boxes), 0.7, topn)
            want0 = g["keep_n%d_t07" % n]
            want1 = native.nms(boxes[1], cases.nms_boxes(900 + n, n)[1], 0.7)
            for b, want in ((0, want0), (1, want1), (2, want0)):
                ref = want if topn == 0 else want[:topn]
                c = int(cnt[b].item())
                assert c == len(ref)
                assert np.array_equal(keep[b, :c].cpu().numpy(), ref)


def test_frozen_bn_residual_relu_matches_torch():
    """ait_bn_act_fwd/bwd vs eval-mode BatchNorm2d + add + ReLU in torch (fp32 tolerance 1e-5)."""
    from ait_amd.faster_rcnn import bn_act
    torch.manual_seed(0)
    # HW % 4 == 0 and != 0, then the same in channels-last memory (C % 4 == 0 and != 0)
    for shape, fmt in (((3, 16, 4, 4), torch.contiguous_format), ((2, 8, 75, 125), torch.contiguous_format),
                       ((1, 5, 7, 9), torch.contiguous_format), ((3, 16, 4, 4), torch.channels_last),
                       ((2, 6, 5, 3), torch.channels_last)):
        bn = torch.nn.BatchNorm2d(shape[1]).cuda().eval()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
            bn.running_mean.uniform_(-0.5, 0.5); bn.running_var.uniform_(0.5, 1.5)
        for p in bn.parameters():
            p.requires_grad = False
        for use_res in (False, True):
            for relu in (True, False):
                x = torch.randn(shape, device="cuda").contiguous(memory_format=fmt).requires_grad_(True)
                r = torch.randn(shape, device="cuda", requires_grad=True) if use_res else None
                y = bn_act(x, bn, residual=r, relu=relu)
                assert y.is_contiguous(memory_format=fmt)
                ref = bn(x) + (r if use_res else 0)
                ref = torch.relu(ref) if relu else ref
                assert torch.allclose(y, ref, rtol=1e-5, atol=1e-6)
                g = torch.randn(shape, device="cuda")
                wrt = [x] + ([r] if use_res else [])
                got = torch.autograd.grad(y, wrt, g)
                want = torch.autograd.grad(ref, wrt, g)
                for a, b in zip(got, want):
                    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_bottleneck_gradients_summed_inside_the_frozen_bn_backward():
    """The pair form of a bottleneck's closing pass (faster_rcnn._BnActPair, ait_bn_act_bwd's second addend): the block's
    result goes to the next block as two tensors over one storage, the two gradients come back separately and are summed
    inside the backward pass.  (1) the kernel: dx / dres for g = dy + dy2 equal those of the pre-summed gradient bit for bit,
    f32 and bf16, both memory formats; (2) a stack of bottlenecks (resnet_sys_transformer_sk_dilat.py:89-107) through
    run_stages: outputs identical and every parameter / input gradient equal to the plain composition's, with no add
    kernel's worth of difference -- the sums are the same two-operand f32 additions."""
    import ait_amd.faster_rcnn as fr
    from ait_amd import ops
    torch.manual_seed(3)
    for shape, fmt in (((2, 16, 6, 4), torch.contiguous_format), ((2, 16, 5, 3), torch.channels_last), ((1, 5, 7, 9), torch.contiguous_format)):
        y = torch.randn(shape, device="cuda").contiguous(memory_format=fmt)
        g1, g2 = (torch.randn(shape, device="cuda").contiguous(memory_format=fmt) for _ in range(2))
        sc = torch.rand(shape[1], device="cuda") + 0.5
        for relu in (True, False):
            a = ops.bn_act_bwd(g1 + g2, y, sc, relu, True)
            b = ops.bn_act_bwd(g1, y, sc, relu, True, dy2=g2)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    y = torch.randn(2, 16, 5, 3, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g1, g2 = (torch.randn(2, 16, 5, 3, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(2))
    sc = torch.rand(16, device="cuda") + 0.5
    b = ops.bn_act_bwd_bf16(g1, y, sc, True, True, dy2=g2)
    want = ((g1.float() + g2.float()) * (y.float() > 0))
    assert torch.equal(b[1], want.to(torch.bfloat16)) and torch.equal(b[0], (want * sc.view(1, -1, 1, 1)).to(torch.bfloat16))
    # (2) three stages of bottlenecks, with a stride-2 stage boundary (the dead-position subsampling path) in the middle
    net = fr.ResNet(fr.Bottleneck, [2, 3, 2, 1]).cuda().eval()          # eval(): every BatchNorm frozen, as the detector's
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            for p in m.parameters():
                p.requires_grad = False
            with torch.no_grad():
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.6, 1.4); m.weight.uniform_(0.7, 1.3); m.bias.uniform_(-0.2, 0.2)
    stages = [net.layer1, net.layer2, net.layer3]
    x0 = torch.randn(2, 64, 24, 20, device="cuda").contiguous(memory_format=torch.channels_last)
    cot = None
    res = {}
    for mode in (True, False):
        fr._PAIR_GRADS = mode
        try:
            net.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            out = fr.run_stages(stages, x)
            cot = torch.randn_like(out) if cot is None else cot
            out.backward(cot)
            res[mode] = (out.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
        finally:
            fr._PAIR_GRADS = True
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    # (the forward is the same kernels either way; MIOpen may pick another solver on its second call of a shape)
    assert rel(res[True][0], res[False][0]) < 1e-5
    # (MIOpen's split-K weight-gradient kernels sum with atomics: two runs of ONE configuration differ by up to ~1e-4
    # relative -- tests/test_gpu_distributed.py; the sums themselves are proven bit-equal in (1))
    assert rel(res[True][1], res[False][1]) < 2e-4
    assert set(res[True][2]) == set(res[False][2]) and len(res[True][2]) >= 20
    for k in res[True][2]:
        assert rel(res[True][2][k], res[False][2][k]) < 5e-4, k


def test_roi_align_channels_last_fuzz_vs_oracle():
    """Seeded fuzz of the channels-last RoIAlign pair against the C oracle: odd feature sizes, C not a
    multiple of the workgroup's channel span, RoIs from sub-pixel to larger than the image, explicit
    sampling ratios, several images, RoI counts across the backward's 256-RoI scan chunks."""
    from ait_amd.roi_layers import ROIAlign
    rs = np.random.RandomState(1234)
    for case in range(12):
        # (C = 128 / 256 / 1024: the channel-sliced forward, one 128-channel-multiple slice per XCD)
        B = int(rs.randint(1, 4)); C = int([4, 12, 64, 260, 1028, 128, 256, 1024, 128, 256, 64, 1024][case])
        H = int(rs.randint(3, 41)); W = int(rs.randint(3, 70))
        n = int(rs.choice([1, 7, 255, 256, 257, 600])); sr = int(rs.choice([0, 0, 1, 3]))
        feat = rs.standard_normal((B, C, H, W)).astype(np.float32)
        x1 = rs.uniform(-40, W * 16, n); y1 = rs.uniform(-40, H * 16, n)
        w = np.where(rs.rand(n) < 0.2, rs.uniform(0.01, 4, n), rs.uniform(4, W * 20, n))
        h = np.where(rs.rand(n) < 0.2, rs.uniform(0.01, 4, n), rs.uniform(4, H * 20, n))
        rois = np.stack([rs.randint(0, B, n).astype(np.float32), x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
        op = ROIAlign((7, 7), 1.0 / 16.0, sr, channels_last=True)
        x = _dev(feat).requires_grad_(True)
        y = op(x, _dev(rois))
        want = native.roi_align_fwd(feat, rois, sampling_ratio=sr) if sr else native.roi_align_fwd(feat, rois)
        scale = max(1.0, float(np.abs(want).max()))
        assert float(np.abs(y.detach().cpu().numpy() - want).max()) <= 2e-5 * scale, (case, B, C, H, W, n, sr)
        g = rs.standard_normal(want.shape).astype(np.float32)
        y.backward(_dev(g))
        gwant = native.roi_align_bwd(g, rois, feat.shape, sampling_ratio=sr) if sr else native.roi_align_bwd(g, rois, feat.shape)
        gs = max(1.0, float(np.abs(gwant).max()))
        assert float(np.abs(x.grad.cpu().numpy() - gwant).max()) <= 1e-4 * gs, (case, B, C, H, W, n, sr)


def test_nms_fuzz_bit_exact_vs_oracle():
    """Seeded fuzz of the device NMS (mask + single-workgroup scan with the prefetched critical
    column and the deferred ORs) against the C oracle: box counts around the 64-row block and
    512-thread boundaries, clustered boxes (long suppression chains), every max_keep regime, the
    unsorted-scores path, and the batched entry.  Integer output: bit-exact."""
    from ait_amd.roi_layers import nms, nms_sorted, nms_sorted_batched
    rs = np.random.RandomState(4321)
    for case in range(24):
        n = int(rs.choice([1, 2, 63, 64, 65, 127, 128, 129, 191, 193, 511, 513, 700, 1500]))
        thr = float(rs.choice([0.3, 0.5, 0.7]))
        centers = rs.uniform(0, 400, (max(1, n // int(rs.choice([1, 4, 16]))), 2))
        c = centers[rs.randint(0, len(centers), n)] + rs.uniform(-12, 12, (n, 2))
        wh = rs.uniform(8, 120, (n, 2))
        boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        scores = np.sort(rs.rand(n).astype(np.float32))[::-1].copy()          # strictly sorted input
        want = native.nms(boxes, scores, thr)
        for max_keep in (0, 1, max(1, len(want) // 2), len(want), len(want) + 5):
            keep, cnt = nms_sorted(_dev(boxes), thr, max_keep)
            k = int(cnt.item())
            ref = want if max_keep == 0 else want[:max_keep]
            assert k == len(ref) and np.array_equal(keep[:k].cpu().numpy(), ref), (case, n, thr, max_keep)
        perm = rs.permutation(n)
        got = nms(_dev(boxes[perm]), _dev(scores[perm]), thr).cpu().numpy()
        assert np.array_equal(got, native.nms(boxes[perm], scores[perm], thr)), (case, n, thr, "unsorted")
    # batched: four images of different content, same n
    n = 900
    bx = []
    for b in range(4):
        c = rs.uniform(0, 300 + 100 * b, (n, 2)); wh = rs.uniform(10, 150, (n, 2))
        bx.append(np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32))
    keep, cnt = nms_sorted_batched(_dev(np.stack(bx)), 0.7, 300)
    sc = np.arange(n, 0, -1).astype(np.float32)
    for b in range(4):
        ref = native.nms(bx[b], sc, 0.7)[:300]
        assert int(cnt[b]) == len(ref) and np.array_equal(keep[b, :len(ref)].cpu().numpy(), ref)


@pytest.mark.parametrize("n,cin,cout,hw,k,stride,pad", [
    (37, 512, 512, 4, 3, 1, 1),      # layer4.conv2 on the proposal tail (4x4 maps); 592 rows: 128x128 tiles
    (300, 256, 128, 4, 3, 1, 1),     # 4800 rows: both tile shapes' edges
    (40, 128, 64, 8, 3, 2, 1),       # stride 2: 8x8 -> 4x4 (the SK block's window geometry)
    (16, 128, 256, 2, 1, 1, 0),      # 1x1 on 2x2 maps (degenerate window)
    (1300, 512, 512, 4, 3, 1, 1),    # 20800 rows: 256x128 tiles (>= 512 of them in the data gradient)
])
def test_implicit_gemm_convolutions_vs_torch(n, cin, cout, hw, k, stride, pad):
    """ait_conv_fwd_f32 / ait_conv_bwd_data_f32 / ait_conv_bwd_weight_f32 (channels-last implicit GEMMs, no
    im2col) against torch's convolution in float64: forward with bias + residual + ReLU, both gradients."""
    from ait_amd import ops
    torch.manual_seed(n + cin + hw)
    oh = (hw + 2 * pad - k) // stride + 1
    x = torch.randn(n, cin, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device="cuda") * (1.0 / (cin * k * k) ** 0.5)).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device="cuda")
    res = torch.randn(n * oh * oh, cout, device="cuda")
    xm = x.permute(0, 2, 3, 1).reshape(n * hw * hw, cin)
    wm = w.permute(0, 2, 3, 1).contiguous()
    geom = ops.conv_geom(n, (hw, hw), (oh, oh), (k, k), stride, pad)
    assert ops.conv_supported((hw, hw), (oh, oh), stride, cin, cout)
    y = ops.conv_fwd(xm, wm, geom, bias=bias, residual=res, relu=True)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), stride, pad)
    ref = torch.relu(ref.permute(0, 2, 3, 1).reshape(-1, cout) + res.double())
    assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6
    y0 = ops.conv_fwd(xm, wm, geom)
    ref0 = torch.nn.functional.conv2d(x.double(), w.double(), None, stride, pad).permute(0, 2, 3, 1).reshape(-1, cout)
    assert float((y0.double() - ref0).abs().max()) <= 2e-5 * float(ref0.abs().max()) + 1e-6
    # gradients
    dy = torch.randn(n * oh * oh, cout, device="cuda")
    dyn = dy.view(n, oh, oh, cout).permute(0, 3, 1, 2).double()
    dx_ref = torch.nn.grad.conv2d_input((n, cin, hw, hw), w.double(), dyn, stride, pad).permute(0, 2, 3, 1).reshape(-1, cin)
    dx = ops.conv_bwd_data(dy, wm, geom)
    assert float((dx.double() - dx_ref).abs().max()) <= 2e-5 * float(dx_ref.abs().max()) + 1e-6
    gate = torch.randn(n * hw * hw, cin, device="cuda")
    dxm = ops.conv_bwd_data(dy, wm, geom, residual=gate, mask_pos=True)
    assert float((dxm.double() - dx_ref * (gate > 0)).abs().max()) <= 2e-5 * float(dx_ref.abs().max()) + 1e-6
    dw_ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, k, k), dyn, stride, pad).permute(0, 2, 3, 1)
    dw = ops.conv_bwd_weight(dy, xm, geom, k, k, split_k=8)
    assert float((dw.double() - dw_ref).abs().max()) <= 5e-5 * float(dw_ref.abs().max()) + 1e-6


def test_box_kernels_equal_the_tensor_expressions():
    """ait_rpn_decode / ait_proposals_assemble / ait_roi_classify / ait_roi_sample_gather against the tensor
    expressions they replace (ait_amd/rpn.py: the reference's formulas op by op): bit-identical boxes, scores,
    class sizes, member lists, sampled RoIs and weights; regression targets to the last ulp of log()."""
    from ait_amd.config import cfg
    from ait_amd.rpn import _ProposalLayer, _ProposalTargetLayer
    torch.manual_seed(21)
    b, A, H, W = 3, 9, 19, 25
    pl = _ProposalLayer(16, [8, 16, 32], [0.5, 1, 2])
    probs = torch.softmax(torch.randn(b, 2, A * H, W, device="cuda"), 1).view(b, 2 * A, H, W)
    deltas = (torch.randn(b, 4 * A, H, W, device="cuda") * 0.5).contiguous(memory_format=torch.channels_last)
    info = torch.tensor([[300.0, 400.0, 1.0]] * b, device="cuda")
    for key in ("TRAIN", "TEST"):
        got = pl._run_hip(probs, deltas, info, key)
        # the tensor-expression path (what _run does for non-fp32 / CPU tensors), forced on the same inputs
        A_ = pl._num_anchors
        import ait_amd.rpn as R
        scores = probs[:, A_:].permute(0, 2, 3, 1).reshape(b, -1)
        d = deltas.permute(0, 2, 3, 1).reshape(b, -1, 4)
        boxes = R.clip_boxes(R.bbox_transform_inv(pl._grid.get(H, W, probs.device).unsqueeze(0), d), info)
        order = torch.sort(scores, 1, True)[1][:, :cfg[key].RPN_PRE_NMS_TOP_N]
        cand = torch.gather(boxes, 1, order.unsqueeze(2).expand(-1, -1, 4)).contiguous()
        from ait_amd.roi_layers import nms_sorted_batched
        keep, n_keep = nms_sorted_batched(cand, cfg[key].RPN_NMS_THRESH, cfg[key].RPN_POST_NMS_TOP_N)
        post_n = cfg[key].RPN_POST_NMS_TOP_N
        want = torch.zeros(b, post_n, 5, device="cuda")
        for i in range(b):
            k = int(n_keep[i])
            want[i, :, 0] = i
            want[i, :k, 1:] = cand[i, keep[i, :k]]
        assert torch.equal(got, want), key

    ptl = _ProposalTargetLayer(2)
    R0, G = 500, 20
    rois = torch.zeros(b, R0, 5, device="cuda")
    xy = torch.rand(b, R0, 2, device="cuda") * 300
    wh = torch.rand(b, R0, 2, device="cuda") * 120
    rois[:, :, 1:3] = xy.round()
    rois[:, :, 3:5] = (xy + wh).round()
    rois[:, ::37, 3:5] = rois[:, ::37, 1:3]                 # zero-area RoIs (IoU -1)
    rois[:, :, 0] = torch.arange(b, device="cuda").view(b, 1)
    gt = torch.zeros(b, G, 5, device="cuda")                # rows past the real boxes stay zero (zero area)
    for i in range(b):
        n = 2 + i
        gt[i, :n, :4] = rois[i, 5:5 + n, 1:5] + torch.tensor([3.0, -2.0, 4.0, 1.0], device="cuda")
        gt[i, :n, 4] = 1
    rois[:, 100, 1:5] = gt[:, 0, :4]                        # an exact hit (IoU 1) and ...
    rois[:, 101, 1:5] = gt[:, 0, :4]                        # ... a duplicate of it
    want = ptl._classify(rois, gt)
    got = ptl._classify_hip(rois, gt)
    names = ["all_rois", "assign", "labels", "counts", "fg_members", "bg_members"]
    for nme, a, w in zip(names, got, want):
        assert torch.equal(a, w.to(a.dtype)), nme
    assert int(got[3][:, 0].min()) > 0
    for P in (128, 300):
        pos = torch.stack([torch.randint(0, int(got[3][i].min()), (P,), device="cuda") for i in range(b)])
        n_fg = torch.tensor([32, 0, P], device="cuda")[:b]
        w = ptl._gather(pos, n_fg, *got[4:6], got[2], got[0], got[1], gt)
        h = ptl._gather_hip(pos, n_fg, *got[4:6], got[2], got[0], got[1], gt)
        for nme, a, ww in zip(["rois", "labels", "targets", "inside", "outside"], h, w):
            if nme == "targets":
                assert torch.allclose(a, ww, rtol=2e-6, atol=1e-7), nme
            else:
                assert torch.equal(a, ww), nme


def test_anchor_target_kernels_equal_the_tensor_expressions(monkeypatch):
    """ait_anchor_classify / ait_anchor_targets (+ the host's RNG draws between them) against the tensor
    expressions of the same layer under the same NumPy seed: identical labels and weights on the full anchor
    grid, regression targets to the last ulp of log(); batches with and without over-full classes."""
    import numpy as np
    import ait_amd.rpn as R
    from ait_amd.config import cfg
    torch.manual_seed(4)
    b, H, W, G = 3, 38, 63, 20
    info = torch.tensor([[600.0, 1000.0, 1.0]] * b, device="cuda")
    score = torch.zeros(b, 18, H, W, device="cuda")
    for trial, n_gt in enumerate((1, 3, 12)):
        gt = torch.zeros(b, G, 5, device="cuda")
        for i in range(b):
            xy = torch.rand(n_gt, 2, device="cuda") * torch.tensor([700.0, 350.0], device="cuda")
            wh = 40 + torch.rand(n_gt, 2, device="cuda") * torch.tensor([280.0, 230.0], device="cuda")
            gt[i, :n_gt, :2] = xy.round()
            gt[i, :n_gt, 2:4] = (xy + wh).round()
            gt[i, :n_gt, 4] = 1
        nb = torch.full((b,), n_gt, device="cuda")
        outs = {}
        for hip in (True, False):
            monkeypatch.setattr(R, "_BOX_KERNELS", hip)
            layer = R._AnchorTargetLayer(16, [8, 16, 32], [0.5, 1, 2])
            np.random.seed(11 + trial)
            outs[hip] = layer((score, gt, info, nb))
        for name, a, w in zip(["labels", "targets", "inside", "outside"], outs[True], outs[False]):
            assert a.shape == w.shape, name
            if name == "targets":
                assert torch.allclose(a, w, rtol=2e-6, atol=1e-7), (name, trial)
            else:
                assert torch.equal(a, w.to(a.dtype)), (name, trial)
        assert int((outs[True][0] == 1).sum()) > 0 and int((outs[True][0] == 0).sum()) > 0


@pytest.mark.parametrize("n,k,stride,pad", [(300, 3, 2, 1), (300, 1, 2, 0), (80, 3, 1, 1), (4, 3, 2, 1), (8, 1, 2, 0), (16, 3, 2, 1)])
def test_grouped_implicit_gemm_convolutions_vs_torch(n, k, stride, pad):
    """The SK block's grouped convolutions (8 groups of 128 channels, blocks_sys_transformer_sk_dilat.py:938-947)
    on the implicit-GEMM kernels -- forward with bias, data and weight gradients -- against torch's grouped
    convolution in float64, at the bench geometry (8x8 maps evaluated at stride 2) and at stride 1."""
    from ait_amd import ops
    torch.manual_seed(n + k)
    G, C, hw = 8, 1024, 8
    oh = (hw + 2 * pad - k) // stride + 1
    x = torch.randn(n, C, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, C // G, k, k, device="cuda") * (1.0 / (C // G * k * k) ** 0.5)
    bias = torch.randn(C, device="cuda")
    xm = x.permute(0, 2, 3, 1).reshape(n * hw * hw, C)
    wm = w.permute(0, 2, 3, 1).contiguous()
    geom = ops.conv_geom(n, (hw, hw), (oh, oh), (k, k), stride, pad, groups=G)
    y = ops.conv_fwd(xm, wm, geom, bias=bias)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), stride, pad, 1, G).permute(0, 2, 3, 1).reshape(-1, C)
    assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6
    dy = torch.randn(n * oh * oh, C, device="cuda")
    dyn = dy.view(n, oh, oh, C).permute(0, 3, 1, 2).double()
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    torch.nn.functional.conv2d(xd, wd, None, stride, pad, 1, G).backward(dyn)
    dx = ops.conv_bwd_data(dy, wm, geom)
    dx_ref = xd.grad.permute(0, 2, 3, 1).reshape(-1, C)
    assert float((dx.double() - dx_ref).abs().max()) <= 2e-5 * float(dx_ref.abs().max()) + 1e-6
    dw = ops.conv_bwd_weight(dy, xm, geom, k, k, split_k=16)
    dw_ref = wd.grad.permute(0, 2, 3, 1)
    assert float((dw.double() - dw_ref).abs().max()) <= 5e-5 * float(dw_ref.abs().max()) + 1e-6
    if stride != 2:
        return
    # stride 2 (the parity-class launches): accumulating into a gradient in place (residual is dx itself: the second
    # branch of the SK block), and the ReLU-backward gate
    base = torch.randn_like(dx)
    acc = base.clone()
    ops.conv_bwd_data(dy, wm, geom, residual=acc, out=acc)
    assert float((acc.double() - (dx_ref + base.double())).abs().max()) <= 2e-5 * float(dx_ref.abs().max()) + 1e-5
    gate = torch.randn_like(dx)
    dxm = ops.conv_bwd_data(dy, wm, geom, residual=gate, mask_pos=True)
    assert float((dxm.double() - dx_ref * (gate > 0)).abs().max()) <= 2e-5 * float(dx_ref.abs().max()) + 1e-6


def test_box_kernel_edge_cases():
    """Degenerate inputs of the box kernels: a single gt box, RoIs that all miss it (no foreground: every RoI is
    background or nothing), one image in the batch, and proposal counts that are not a multiple of the
    workgroup's per-thread share -- against the tensor expressions."""
    from ait_amd.rpn import _ProposalTargetLayer
    torch.manual_seed(2)
    ptl = _ProposalTargetLayer(2)
    for b, R0, G in ((1, 37, 1), (2, 1, 3), (3, 2000, 20)):
        rois = torch.zeros(b, R0, 5, device="cuda")
        xy = torch.rand(b, R0, 2, device="cuda") * 100
        rois[:, :, 1:3] = xy.round()
        rois[:, :, 3:5] = (xy + 5 + torch.rand(b, R0, 2, device="cuda") * 40).round()
        gt = torch.zeros(b, G, 5, device="cuda")
        gt[:, 0, :4] = torch.tensor([400.0, 300.0, 480.0, 390.0], device="cuda")      # far from every RoI
        gt[:, 0, 4] = 1
        want = ptl._classify(rois, gt)
        got = ptl._classify_hip(rois, gt)
        for name, a, w in zip(["all_rois", "assign", "labels", "counts", "fg_members", "bg_members"], got, want):
            assert torch.equal(a, w.to(a.dtype)), (name, b, R0, G)
        # the only foreground RoI is the gt box itself (appended as a RoI: IoU 1)
        assert bool((got[3][:, 0] == 1).all())


# (300, 4): one pair's worth; (37, 2): below every fast path's threshold; (1200, 4): the headline size -- layer4's 3x3 convolutions
# with the out-of-map taps skipped, one launch; (2400, 8): BASELINE configs[2] / [3] per GPU -- the same in two slices
# (513, 3): 520 maps -- the third row tile of every position holds 8 real rows; (255, 1): exactly one row tile per position, every
# tile cut over five or more workgroups
@pytest.mark.parametrize("bp,bs", [(300, 4), (37, 2), (1200, 4), (2400, 8), (513, 3), (255, 1)])
def test_proposal_tail_node_matches_the_module_composition(monkeypatch, bp, bs):
    """ait_tail_fwd / ait_tail_bwd (both SK blocks + layer4 + the mean over positions as ONE autograd node: grouped
    implicit GEMMs at stride 2, per-parity data gradients, frozen BN folded into the weights, every ReLU mask and
    residual add in a GEMM epilogue, proposals and queries through layer4 together) against the nn.Module
    composition of the same parameters on PyTorch-ROCm's convolutions (test hook _TAIL_FUSED = False): pooled
    features, both input gradients and all 18 parameter gradients."""
    import ait_amd.faster_rcnn as fr
    from ait_amd import ops
    torch.manual_seed(5)
    m = fr.resnet(('__background__', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3)
    m.create_architecture()
    for mod in m.RCNN_top.modules():            # non-trivial frozen statistics
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.5, 1.5)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.1)
    for blk in (m.sk.sk_props, m.sk.sk_query):
        for c in blk.convs:
            c[0].bias.data.normal_(0, 0.1)
    m = m.cuda().train()
    x0 = torch.randn(bp, 1024, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last)
    q0 = torch.randn(bs, 1024, 8, 8, device="cuda")
    cot_p, cot_q = torch.randn(bp, 2048, device="cuda"), torch.randn(bs, 2048, device="cuda")
    # (layer4 is registered twice -- RCNN_base.backbone.layer4 and RCNN_top.0 are the same modules, as in the
    # reference -- and named_parameters() reports it under its first name)
    names = [n for n, _ in m.named_parameters()
             if n.startswith(("sk.sk_props.convs", "sk.sk_query.convs", "RCNN_base.backbone.layer4.", "RCNN_top."))
             and "bn" not in n and "downsample.1" not in n]
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(fr, "_TAIL_FUSED", fused)
        m.zero_grad(set_to_none=True)
        x, q = x0.clone().requires_grad_(True), q0.clone().requires_grad_(True)
        ops.reset_fallbacks()
        if m._tail_on_library(x, q, 2):
            yp, yq = m._tail(x, q)
        else:
            a, b = m.sk(x_props=x, x_query=q, stride=2)
            yp, yq = m._head_to_tail(a, subsampled=True), m._head_to_tail(b, subsampled=True)
        assert ops.fallback_count() == 0
        ((yp * cot_p).sum() + (yq * cot_q).sum()).backward()
        params = dict(m.named_parameters())
        res[fused] = [yp.detach(), yq.detach(), x.grad.clone(), q.grad.clone()] + [params[n].grad.clone() for n in names]
    assert len(names) == 18, names
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    labels = ["pooled_props", "pooled_query", "d_x_props", "d_x_query"] + names
    for lab, a, b in zip(labels, res[True], res[False]):
        assert a.shape == b.shape, lab
        # forward to rounding; gradients cross up to nine ReLU masks whose bits flip on pre-activations within
        # rounding of zero between two implementations
        # (which solver MIOpen picks for the composition differs between boxes and runs, and with it the set of
        # flipped bits: one flip on the 4-image query side is ~1e-3 of that gradient's norm)
        tol = 2e-5 if lab.startswith("pooled") else 5e-3
        assert rel(a, b) < tol, (lab, rel(a, b))


def test_tail_backward_is_held_to_its_forwards_layout(monkeypatch):
    """ABI v7: layer4's saved activations are position-major in the split product form and map-major otherwise.  (1) The
    autograd node runs its backward in the product form of ITS forward whatever the global switch says by then: gradients
    equal those of an undisturbed step.  (2) The C entry refuses (AIT_EINVAL) a layout word that is not the one the forward
    reported -- the other row order, or no word of the forward's at all -- instead of reading the buffer in the wrong order."""
    import ait_amd.faster_rcnn as fr
    from ait_amd import _lib, ops
    torch.manual_seed(11)
    m = fr.resnet(('__background__', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3)
    m.create_architecture()
    m = m.cuda().train()
    bp, bs = 37, 2
    x0 = torch.randn(bp, 1024, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last)
    q0 = torch.randn(bs, 1024, 8, 8, device="cuda")
    cot = torch.randn(bp + bs, 2048, device="cuda")
    L = _lib.lib()
    real = L.ait_tail_bwd
    seen = []

    def spy(*a):
        word = a[11]
        seen.append(word)
        assert word >> 20 == 0xA18 and (word & 1) == 1            # position-major: the f32 split form
        assert real(*(a[:11] + (word ^ 1,) + a[12:])) == -1       # the other row order
        assert real(*(a[:11] + (0x12345,) + a[12:])) == -1        # not a word of the forward's
        return real(*a)

    grads = {}
    for switch in (False, True):
        m.zero_grad(set_to_none=True)
        x, q = x0.clone().requires_grad_(True), q0.clone().requires_grad_(True)
        assert m._tail_on_library(x, q, 2)
        yp, yq = m._tail(x, q)
        try:
            if switch:
                ops.set_matmul_dtype("f32_native")                  # between forward and backward
                monkeypatch.setattr(L, "ait_tail_bwd", spy, raising=False)
            (torch.cat([yp, yq]) * cot).sum().backward()
        finally:
            ops.set_matmul_dtype("f32")
            if switch:
                monkeypatch.undo()
        grads[switch] = [x.grad.clone(), q.grad.clone(), m.sk.sk_props.convs[1][0].weight.grad.clone(),
                         m.RCNN_top[0][1].conv2.weight.grad.clone()]
    assert len(seen) == 1
    for a, b in zip(grads[True], grads[False]):
        assert float((a - b).norm() / (b.norm() + 1e-30)) < 1e-4     # (atomics in the weight gradients: not bit-equal)


def _tail_bf16_emulation(m, dp, x, q):
    """The proposal tail in float64 on the host, rounded to bf16 (straight-through for autograd) exactly where the library's
    bf16-storage form rounds: the SK blocks' operands, their result, the folded weights, every activation of layer4."""
    import torch.nn.functional as F
    import ait_amd.faster_rcnn as fr

    def r(t):
        return t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()

    def sk(prefix, blk, x):
        fs = []
        for i in range(2):
            conv = blk.convs[i][0]
            w, b = dp["%s.convs.%d.0.weight" % (prefix, i)], dp["%s.convs.%d.0.bias" % (prefix, i)]
            fs.append(F.relu(F.conv2d(r(x), r(w), b, 2, conv.padding, 1, 8)))
        return r(fs[0] ** 2 + fs[1] ** 2)
    xt = torch.cat([sk("sk.sk_props", m.sk.sk_props, x), sk("sk.sk_query", m.sk.sk_query, q)])
    for k, b in enumerate(m.RCNN_top[0]):
        def fold(name, bn):
            scale, shift, _ = fr._bn_affine(bn)
            w = dp["RCNN_base.backbone.layer4.%d.%s.weight" % (k, name)]
            return r(w * scale.cpu().double()[:, None, None, None]), shift.cpu().double()[None, :, None, None]
        w1, s1 = fold("conv1", b.bn1)
        a1 = r(F.relu(F.conv2d(xt, w1) + s1))
        w2, s2 = fold("conv2", b.bn2)
        a2 = r(F.relu(F.conv2d(a1, w2, padding=1) + s2))
        idn = xt
        if k == 0:
            wd, sd = fold("downsample.0", b.downsample[1])
            idn = r(F.conv2d(xt, wd) + sd)
        w3, s3 = fold("conv3", b.bn3)
        xt = r(F.relu(F.conv2d(a2, w3) + s3 + idn))
    return xt.mean((2, 3))


@pytest.mark.parametrize("bp,bs,positive", [(100, 4, True), (100, 4, False), (300, 5, True)])
def test_proposal_tail_on_bf16_storage(monkeypatch, bp, bs, positive):
    """ABI v8: under AIT_CTX_BF16, from 1024 rows, ait_tail_* keeps layer4's activations and gradients in bf16 and runs its
    products -- the 3x3 convolutions through the window gather -- on the bf16-storage kernels.  Against a float64 host
    emulation that rounds to bf16 at the same points (_tail_bf16_emulation; gradients by autograd, un-rounded).
    positive: frozen statistics and biases that keep every pre-activation above zero -- no ReLU masks, so none can flip:
    pooled features to 2e-4 and all 22 gradients to 8e-3 in relative L2 (what is left is the bf16 rounding of the stored
    GRADIENTS: measured 1.6e-3 .. 2.9e-3).  Otherwise (random statistics: half of every mask is zero) the two chains
    decorrelate at the level of one bf16 ulp per element within a few layers -- a perturbation d of a value flips its rounding
    with probability d / 2^-8 -- and every flipped mask bit moves a gradient: pooled 4e-3, gradients 0.15 (measured 9e-4 /
    1.2e-2 at the last convolution .. 6.1e-2 at the inputs; the f32-tensor form of the same call measures 1.3e-3 / 9.9e-2).
    The forward must report the bf16 layout."""
    import ait_amd.faster_rcnn as fr
    from ait_amd import _lib, ops
    torch.manual_seed(7)
    m = fr.resnet(('__background__', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3)
    m.create_architecture()
    for mod in m.RCNN_top.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.5, 1.5)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.1)
            if positive:
                mod.weight.data.mul_(0.002)
                mod.running_mean.zero_()
                mod.bias.data.fill_(1.0)
    if positive:
        for blk in (m.sk.sk_props, m.sk.sk_query):
            for c in blk.convs:
                c[0].bias.data.fill_(8.0)
    m = m.cuda().train()
    x0 = torch.randn(bp, 1024, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last)
    q0 = torch.randn(bs, 1024, 8, 8, device="cuda")
    cot = torch.randn(bp + bs, 2048, device="cuda")
    names = [n for n, _ in m.named_parameters()
             if n.startswith(("sk.sk_props.convs", "sk.sk_query.convs", "RCNN_base.backbone.layer4.", "RCNN_top."))
             and "bn" not in n and "downsample.1" not in n]
    assert len(names) == 18
    L = _lib.lib()
    real = L.ait_tail_bwd
    words = []

    def spy(*a):
        words.append(a[11])
        return real(*a)

    x, q = x0.clone().requires_grad_(True), q0.clone().requires_grad_(True)
    ops.set_matmul_dtype("bf16")
    try:
        assert m._tail_on_library(x, q, 2)
        yp, yq = m._tail(x, q)
        monkeypatch.setattr(L, "ait_tail_bwd", spy, raising=False)
        (torch.cat([yp, yq]) * cot).sum().backward()
    finally:
        monkeypatch.undo()
        ops.set_matmul_dtype("f32")
    assert len(words) == 1 and words[0] >> 20 == 0xA18 and (words[0] & 3) == 2, [hex(w) for w in words]    # bf16, map-major
    params = dict(m.named_parameters())
    got = [torch.cat([yp, yq]).detach(), x.grad, q.grad] + [params[n].grad for n in names]
    dp = {n: params[n].detach().cpu().double().requires_grad_(True) for n in names}
    xe, qe = x0.cpu().double().requires_grad_(True), q0.cpu().double().requires_grad_(True)
    ye = _tail_bf16_emulation(m, dp, xe, qe)
    (ye * cot.cpu().double()).sum().backward()
    want = [ye.detach(), xe.grad, qe.grad] + [dp[n].grad for n in names]
    rel = lambda a, b: float((a.double().cpu() - b).norm() / (b.norm() + 1e-30))
    for lab, a, b in zip(["pooled", "d_x_props", "d_x_query"] + names, got, want):
        assert bool(torch.isfinite(a).all()), lab
        tol = (2e-4 if positive else 4e-3) if lab == "pooled" else (8e-3 if positive else 0.15)
        assert rel(a, b) < tol, (lab, rel(a, b))
    if positive:
        assert bool((yp > 0).all())                        # (the configuration did what it is for)


def test_bf16_tail_rows_do_not_depend_on_the_maps_around_them():
    """A map's pooled feature and input gradient depend on that map alone, and an output element's sum runs over the reduction
    in the same order wherever its row tile lies and however many row tiles there are: the bf16-storage tail at BASELINE
    configs[4]'s size -- 4096 + 8 maps, 257 row tiles of the 256 x 256 x 64 tile: two whole rounds and two tiles -- against
    2047 + 8 on the first 2047 proposals (another row count, another padding, another tile for every query map): the
    proposals' pooled features bit for bit, the queries' and the input gradients to rounding."""
    import ait_amd.faster_rcnn as fr
    from ait_amd import ops
    torch.manual_seed(13)
    m = fr.resnet(('__background__', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3)
    m.create_architecture()
    m = m.cuda().train()
    n_big, n_small, nq = 4096, 2047, 8
    x0 = torch.randn(n_big, 1024, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last)
    q0 = torch.randn(nq, 1024, 8, 8, device="cuda")
    cot = torch.randn(n_big + nq, 2048, device="cuda")
    out = {}
    ops.set_matmul_dtype("bf16")
    try:
        for bp in (n_big, n_small):
            m.zero_grad(set_to_none=True)
            x, q = x0[:bp].clone().requires_grad_(True), q0.clone().requires_grad_(True)
            yp, yq = m._tail(x, q)
            ((yp * cot[:bp]).sum() + (yq * cot[n_big:]).sum()).backward()
            out[bp] = (yp.detach(), yq.detach(), x.grad.clone())
            del x, q, yp, yq
    finally:
        ops.set_matmul_dtype("f32")
    assert torch.equal(out[n_big][0][:n_small], out[n_small][0])
    # (the query maps sit in the thin last round of row tiles, which the 4608-deep convolutions cut along K: which of their rows
    # share a cut tile differs between the two runs, and with it the order of eight partial sums -- f32 rounding under a bf16
    # store: a rare last-bit flip of a stored activation, not more)
    qa, qb = out[n_big][1], out[n_small][1]
    assert float((qa - qb).norm() / qb.norm()) < 1e-3, float((qa - qb).norm() / qb.norm())
    # (the input gradient leaves through the SK blocks' f32 products, whose stream-K cut of the last round of tiles -- and with
    # it the order of a few partial sums -- follows the number of rows: to f32 rounding, not to the bit)
    a, b = out[n_big][2][:n_small], out[n_small][2]
    assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), float((a - b).abs().max() / b.abs().max())
    assert float(out[n_big][0][n_small:].abs().sum()) > 0 and float(out[n_big][2][n_small:].abs().sum()) > 0
    assert bool(torch.isfinite(out[n_big][0]).all()) and bool(torch.isfinite(out[n_big][2]).all())


@pytest.mark.parametrize("n,h,w,cin,cout,k,pad", [(4, 38, 63, 1024, 512, 3, 1), (1, 20, 30, 256, 128, 3, 1), (2, 19, 31, 128, 256, 3, 1),
                                                   (3, 7, 5, 128, 64, 1, 0), (2, 38, 63, 256, 256, 3, 1)])
def test_implicit_gemm_convolutions_on_maps_of_any_size(n, h, w, cin, cout, k, pad):
    """Maps whose sides are not powers of two (the C4 feature maps: 38 x 63 for a 600 x 1000 image) -- forward with
    bias + ReLU, data gradient, weight gradient (rows not a multiple of the 16-row reduction slab: 4 * 38 * 63 = 9576)
    against torch's convolution in float64; and the RPN's convolution node built on them against nn.Conv2d + ReLU."""
    from ait_amd import ops
    torch.manual_seed(n * h + w)
    x = torch.randn(n, cin, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, k, k, device="cuda") * (1.0 / (cin * k * k) ** 0.5)
    bias = torch.randn(cout, device="cuda")
    xm = x.permute(0, 2, 3, 1).reshape(n * h * w, cin)
    wm = wt.permute(0, 2, 3, 1).contiguous()
    geom = ops.conv_geom(n, (h, w), (h, w), (k, k), 1, pad)
    y = ops.conv_fwd(xm, wm, geom, bias=bias, relu=True)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), 1, pad)).permute(0, 2, 3, 1).reshape(-1, cout)
    assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6
    dy = torch.randn(n * h * w, cout, device="cuda")
    dyn = dy.view(n, h, w, cout).permute(0, 3, 1, 2).double()
    xd = x.double().requires_grad_(True)
    wd = wt.double().requires_grad_(True)
    torch.nn.functional.conv2d(xd, wd, None, 1, pad).backward(dyn)
    dx = ops.conv_bwd_data(dy, wm, geom)
    dx_ref = xd.grad.permute(0, 2, 3, 1).reshape(-1, cin)
    assert float((dx.double() - dx_ref).abs().max()) <= 2e-5 * float(dx_ref.abs().max()) + 1e-6
    if cin % 128 == 0:
        dw = ops.conv_bwd_weight(dy, xm, geom, k, k, split_k=8)
        dw_ref = wd.grad.permute(0, 2, 3, 1)
        assert float((dw.double() - dw_ref).abs().max()) <= 5e-5 * float(dw_ref.abs().max()) + 1e-6
    if k != 3 or cin % 128:
        return
    from ait_amd.rpn import _Conv3x3BiasRelu
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).cuda()
    with torch.no_grad():
        conv.weight.copy_(wt)
        conv.bias.copy_(bias)
    xa = x.clone().requires_grad_(True)
    ya = _Conv3x3BiasRelu.apply(xa, conv.weight, conv.bias)
    g = torch.randn_like(ya)
    ya.backward(g)
    assert float((ya.double() - ref.view(n, h, w, cout).permute(0, 3, 1, 2)).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6
    # the node's gradients against float64 with the node's OWN ReLU mask (a second f32 implementation flips the mask
    # bit of every pre-activation within rounding of zero, and each flip moves a gradient by a whole |g w|)
    dz = (g * (ya > 0)).double()
    xd2 = x.double().requires_grad_(True)
    wd2 = wt.double().requires_grad_(True)
    bd2 = bias.double().requires_grad_(True)
    (torch.nn.functional.conv2d(xd2, wd2, bd2, 1, 1) * dz).sum().backward()
    for got, want, tol in ((xa.grad, xd2.grad, 2e-5), (conv.weight.grad, wd2.grad, 5e-5), (conv.bias.grad, bd2.grad, 5e-5)):
        assert tuple(got.shape) == tuple(want.shape)
        assert float((got.double() - want).abs().max()) <= tol * float(want.abs().max()) + 1e-6


def test_convolution_epilogues_are_reproducible_launch_to_launch():
    """Every non-atomic product of the proposal tail -- forward with bias + residual + ReLU, data gradient with the
    residual add and the ReLU gate, the per-parity stride-2 data gradient -- twice on the same inputs: bit-identical
    (what caught the packed-epilogue bug of ait_amd/build.py on the dense products)."""
    from ait_amd import ops
    torch.manual_seed(9)
    n, C, hw = 1200, 512, 4
    x = torch.randn(n * hw * hw, C, device="cuda")
    w = (torch.randn(C, 3, 3, C, device="cuda") / (9 * C) ** 0.5).contiguous()
    bias = torch.randn(C, device="cuda")
    res = torch.randn(n * hw * hw, C, device="cuda")
    geom = ops.conv_geom(n, (hw, hw), (hw, hw), (3, 3), 1, 1)
    y1 = ops.conv_fwd(x, w, geom, bias=bias, residual=res, relu=True)
    y2 = ops.conv_fwd(x, w, geom, bias=bias, residual=res, relu=True)
    assert torch.equal(y1, y2)
    dy = torch.randn_like(y1)
    d1 = ops.conv_bwd_data(dy, w, geom, residual=res, mask_pos=True)
    d2 = ops.conv_bwd_data(dy, w, geom, residual=res, mask_pos=True)
    assert torch.equal(d1, d2)
    # the SK block's grouped 3x3 at stride 2 on 8x8 maps: parity-class launch with the in-place accumulation
    G, Cg = 8, 1024
    geom2 = ops.conv_geom(300, (8, 8), (4, 4), (3, 3), 2, 1, groups=G)
    wg = (torch.randn(Cg, 3, 3, Cg // G, device="cuda") / (9 * Cg // G) ** 0.5).contiguous()
    dyg = torch.randn(300 * 16, Cg, device="cuda")
    base = torch.randn(300 * 64, Cg, device="cuda")
    a1, a2 = base.clone(), base.clone()
    ops.conv_bwd_data(dyg, wg, geom2, residual=a1, out=a1)
    ops.conv_bwd_data(dyg, wg, geom2, residual=a2, out=a2)
    assert torch.equal(a1, a2)


@pytest.mark.gpu
@pytest.mark.parametrize("A", [9, 12])
def test_rpn_heads_as_one_product_match_the_two_convolutions(A):
    """rpn._Heads1x1 (the RPN's objectness and box-delta 1x1 heads stacked into one 64-row product on the library's
    GEMM, lib/model/rpn/rpn.py:34-43) against the two convolutions in float64: outputs and every gradient, on a map
    whose side is not a power of two; 9 anchors (VOC: 18 + 36 outputs in 64 rows) and 12 (COCO: 24 + 48 in 128)."""
    from ait_amd import rpn
    torch.manual_seed(5)
    n, c, h, w = 2, 512, 38, 63
    x = torch.randn(n, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    n1, n2 = 2 * A, 4 * A
    w1, b1 = (torch.randn(n1, c, 1, 1, device="cuda") * 0.05).requires_grad_(True), torch.randn(n1, device="cuda").requires_grad_(True)
    w2, b2 = (torch.randn(n2, c, 1, 1, device="cuda") * 0.05).requires_grad_(True), torch.randn(n2, device="cuda").requires_grad_(True)
    ys, yb = rpn._Heads1x1.apply(x.permute(0, 2, 3, 1).reshape(n * h * w, c), w1, b1, w2, b2)
    ys = ys.reshape(n, h, w, n1).permute(0, 3, 1, 2)
    yb = yb.reshape(n, h, w, n2).permute(0, 3, 1, 2)
    g1, g2 = torch.randn(n, n1, h, w, device="cuda"), torch.randn(n, n2, h, w, device="cuda")
    grads = torch.autograd.grad([ys, yb], [x, w1, b1, w2, b2], [g1, g2])
    xd, w1d, b1d, w2d, b2d = (t.detach().double().requires_grad_(True) for t in (x, w1, b1, w2, b2))
    rs, rb = torch.nn.functional.conv2d(xd, w1d, b1d), torch.nn.functional.conv2d(xd, w2d, b2d)
    ref = torch.autograd.grad([rs, rb], [xd, w1d, b1d, w2d, b2d], [g1.double(), g2.double()])
    rel = lambda a, b: float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))
    assert rel(ys, rs) < 1e-5 and rel(yb, rb) < 1e-5
    for got, want, name in zip(grads, ref, ("dx", "dw1", "db1", "dw2", "db2")):
        assert rel(got, want) < 2e-5, name


@pytest.mark.parametrize("bs,P,F,nb", [(1, 1, 2048, 4), (4, 300, 2048, 4), (2, 37, 512, 8), (3, 64, 2048, 0)])
def test_heads_kernel_vs_float64_linear_layers(bs, P, F, nb):
    """ait_heads_fwd / ait_heads_bwd (csrc/heads.hip) against the reference's composition in float64:
    bbox_pred = Linear(F, nb)(props); score = Linear(8, 2)(Linear(2F, 8)(cat(props, repeat_P(query))))
    (faster_rcnn_sys_transformer_sk_dilat.py:283-288) -- outputs, input gradients and all six parameter gradients.
    Weights at the reference's init scale (N(0, 0.01) / N(0, 0.001)), where the logits are small sums of many terms."""
    from ait_amd import ops
    torch.manual_seed(5 + P)
    R = bs * P
    props, query = torch.randn(R, F, device="cuda").relu_(), torch.randn(bs, F, device="cuda").relu_()
    wb, bb = torch.randn(nb, F, device="cuda") * 1e-3, torch.randn(nb, device="cuda") * 0.1
    w1, b1 = torch.randn(8, 2 * F, device="cuda") * 1e-2, torch.randn(8, device="cuda") * 0.1
    w2, b2 = torch.randn(2, 8, device="cuda") * 1e-2, torch.randn(2, device="cuda") * 0.1
    d_bbox, d_score = torch.randn(R, nb, device="cuda"), torch.randn(R, 2, device="cuda")
    bbox, hidden, score = ops.heads_fwd(props, query, wb, bb, w1, b1, w2, b2)
    dp, dq, g = ops.heads_bwd(d_bbox if nb else None, d_score, props, query, wb, w1, w2, hidden)
    leaves = [t.double().requires_grad_(True) for t in (props, query, wb, bb, w1, b1, w2, b2)]
    p64, q64, wb64, bb64, w164, b164, w264, b264 = leaves
    stack = torch.cat((p64.view(bs, P, F), q64.unsqueeze(1).expand(-1, P, -1)), 2).reshape(R, 2 * F)
    h64 = torch.nn.functional.linear(stack, w164, b164)
    s64 = torch.nn.functional.linear(h64, w264, b264)
    bx64 = torch.nn.functional.linear(p64, wb64, bb64)
    want = torch.autograd.grad([s64, bx64], leaves, [d_score.double(), d_bbox.double()], allow_unused=True)

    def close(name, got, ref, rtol=1e-4):
        if ref is None or ref.numel() == 0:
            return
        err = float((got.double() - ref).abs().max())
        assert err <= rtol * float(ref.abs().max()) + 1e-9, (name, err, float(ref.abs().max()))
    close("score", score, s64.detach(), 1e-5)
    close("hidden", hidden, h64.detach(), 1e-5)
    close("bbox", bbox, bx64.detach(), 1e-5)
    close("d_props", dp, want[0])
    close("d_query", dq, want[1])
    for name, got, ref in zip(("d_w_bbox", "d_b_bbox", "d_w1", "d_b1", "d_w2", "d_b2"), g, want[2:]):
        close(name, got.view(ref.shape) if ref is not None else got, ref)


def test_detector_heads_run_in_the_library_and_keep_module_hooks():
    """the detector's forward takes both heads through _HeadsFn (no torch Linear under the logits) and a forward hook on
    RCNN_cls_score -- how the reference's users (and g9) read the logits -- still sees them"""
    from ait_amd import faster_rcnn as FR
    props, query = torch.randn(6, 2048, device="cuda"), torch.randn(2, 2048, device="cuda")
    lin_b = torch.nn.Linear(2048, 4).cuda()
    lin_s = torch.nn.Sequential(torch.nn.Linear(4096, 8), torch.nn.Linear(8, 2)).cuda()
    bbox, score = FR._HeadsFn.apply(props, query, lin_b.weight, lin_b.bias, lin_s[0].weight, lin_s[0].bias, lin_s[1].weight,
                                    lin_s[1].bias)
    stack = torch.cat((props.view(2, 3, -1), query.unsqueeze(1).expand(-1, 3, -1)), 2).reshape(-1, 4096)
    assert torch.allclose(score, lin_s(stack), rtol=1e-4, atol=1e-5) and torch.allclose(bbox, lin_b(props), rtol=1e-4, atol=1e-5)
    (score.sum() + bbox.sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in list(lin_b.parameters()) + list(lin_s.parameters()))


def test_round5_entry_points_check_their_arguments():
    """ABI v6 additions refuse what they do not support with the documented codes (include/ait_hip.h) instead of
    launching: the bf16-storage products, the conversions, the heads, the dropout-mask read-back."""
    import ctypes
    from ait_amd import _lib, ops
    L = _lib.lib()
    st = _lib.cur_stream(torch.device("cuda"))
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    OK, EINVAL, EUNSUPPORTED = 0, -1, -4                 # include/ait_hip.h
    a16 = torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16)
    c32 = torch.zeros(256, 256, device="cuda")
    # weight-gradient product: rows must split into whole 32-row slabs; outputs in whole 256 x 128 tiles
    big = torch.zeros(96, 256, device="cuda", dtype=torch.bfloat16)
    assert L.ait_gemm_bf16s_tn(256, 256, 80, vp(big), 256, vp(big), 256, vp(c32), 256, 1, None, 0, None, st) == EUNSUPPORTED   # not whole slabs
    assert L.ait_gemm_bf16s_tn(256, 256, 96, vp(big), 256, vp(big), 256, vp(c32), 256, 4, None, 0, None, st) == EUNSUPPORTED   # 3 slabs, 4 ranges
    assert L.ait_gemm_bf16s_tn(256, 256, 96, vp(big), 256, vp(big), 256, vp(c32), 256, 2, None, 0, None, st) == OK             # ranges of 2 + 1 slabs
    assert L.ait_gemm_bf16s_tn(200, 256, 96, vp(big), 256, vp(big), 256, vp(c32), 256, 1, None, 0, None, st) == EUNSUPPORTED
    assert L.ait_gemm_bf16s_tn(256, 256, 96, vp(big), 256, vp(big), 256, vp(c32), 256, 1, None, 0, None, st) == OK
    # NT product: N in whole 128-column tiles, K in whole 32-deep slabs, at least one result
    assert L.ait_gemm_bf16s(64, 64, 64, vp(a16), 64, vp(a16), 64, vp(c32), 64, None, 0, None, None, None, 0, 0, None, st) == EUNSUPPORTED
    w16 = torch.zeros(128, 64, device="cuda", dtype=torch.bfloat16)
    assert L.ait_gemm_bf16s(64, 128, 64, vp(a16), 64, vp(w16), 64, None, 0, None, 0, None, None, None, 0, 0, None, st) == EINVAL
    assert L.ait_gemm_bf16s(0, 128, 64, vp(a16), 64, vp(w16), 64, vp(c32), 128, None, 0, None, None, None, 0, 0, None, st) == OK
    # conversions: odd shapes through the transposing form are fine, the straight form wants columns % 4 == 0
    x = torch.randn(37, 10, device="cuda")
    assert torch.equal(ops.to_bf16(x, transpose=True), x.t().contiguous().to(torch.bfloat16))
    dst = torch.zeros(37, 12, device="cuda", dtype=torch.bfloat16)
    assert L.ait_f32_to_bf16(vp(x), 37, 10, 10, vp(dst), 12, 0, st) == EUNSUPPORTED
    # heads: feature width must be one of the register-resident sizes
    p = torch.zeros(4, 100, device="cuda")
    z = torch.zeros(8, 200, device="cuda")
    o = torch.zeros(4, 8, device="cuda")
    assert L.ait_heads_fwd(vp(p), vp(p), 4, 2, 100, vp(p), vp(p), 4, vp(z), vp(z), vp(z), vp(z), vp(o), vp(o), vp(o), st) == EUNSUPPORTED
    assert L.ait_heads_fwd(vp(p), vp(p), 0, 2, 256, vp(p), vp(p), 4, vp(z), vp(z), vp(z), vp(z), vp(o), vp(o), vp(o), st) == OK
    # dropout mask: p = 0 keeps everything at factor 1; p outside [0, 1) is refused
    assert bool((ops.dropout_mask(123, 0, 1000, 0.0, "cuda") == 1).all())
    m = ops.dropout_mask(123, 5, 4096, 0.25, "cuda")
    assert set(torch.unique(m).tolist()) <= {0.0, float(np.float32(1) / np.float32(0.75))} and 0.70 < float((m != 0).float().mean()) < 0.80
    assert torch.equal(m[7:100], ops.dropout_mask(123, 12, 93, 0.25, "cuda"))            # a function of (seed, absolute index)
    assert L.ait_dropout_mask(1, 0, 10, ctypes.c_float(1.0), vp(o), st) == EINVAL
    # the fused attention backward: nothing to do for zero sequences; NULL tensors, kv_rows outside 1..64 and p = 1 are refused
    t = torch.zeros(64 * 1536, device="cuda")
    f = ctypes.c_float
    core = lambda n, kv, p, df: L.ait_mha_core_bwd(df, vp(t), vp(t), vp(t), vp(t), vp(t), 1536, vp(t), 1536, vp(t), 1536, vp(t), n, kv,
                                                   f(0.125), f(p), 1, vp(t), 1536, vp(t), 1536, vp(t), 1536, vp(t), st)
    assert core(0, 64, 0.0, None) == OK
    assert core(1, 64, 0.0, None) == EINVAL and core(1, 0, 0.0, vp(t)) == EINVAL and core(1, 65, 0.0, vp(t)) == EINVAL
    assert core(1, 64, 1.0, vp(t)) == EINVAL and core(-1, 64, 0.0, vp(t)) == EINVAL
    # ... and so are rows narrower than the eight heads' 512 columns, pitches that are not whole 16-byte vectors and
    # misaligned bases, on the inputs and on the outputs (EUNSUPPORTED, not a fault)
    core2 = lambda q, ldq, dq, lddq: L.ait_mha_core_bwd(vp(t), vp(t), vp(t), vp(t), vp(t), q, ldq, vp(t), 1536, vp(t), 1536, vp(t), 1, 64,
                                                        f(0.125), f(0.0), 1, dq, lddq, vp(t), 1536, vp(t), 1536, vp(t), st)
    off = ctypes.c_void_p(t.data_ptr() + 4)
    assert core2(vp(t), 1536, vp(t), 1536) == OK
    assert core2(vp(t), 256, vp(t), 1536) == EUNSUPPORTED and core2(vp(t), 1538, vp(t), 1536) == EUNSUPPORTED
    assert core2(off, 1536, vp(t), 1536) == EUNSUPPORTED
    assert core2(vp(t), 1536, vp(t), 510) == EUNSUPPORTED and core2(vp(t), 1536, vp(t), 1537) == EUNSUPPORTED
    assert core2(vp(t), 1536, off, 1536) == EUNSUPPORTED


def test_frozen_bn_residual_relu_on_bf16_tensors_matches_float_arithmetic():
    """ait_bn_act_fwd_bf16 / bwd_bf16 (the C4 trunk of the bf16 configuration): the same affine + residual + ReLU pass on
    bf16 channels-last tensors -- f32 arithmetic on the bf16 values, ONE rounding on the way out: against torch computing
    in f32 from the same bf16 inputs and rounding once (bit-exact up to the fused multiply-add's last bit: 1 bf16 ulp)."""
    from ait_amd.faster_rcnn import bn_act
    torch.manual_seed(1)
    for shape in ((2, 64, 9, 7), (3, 256, 5, 5), (1, 1024, 3, 4)):
        bn = torch.nn.BatchNorm2d(shape[1]).cuda().eval()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
            bn.running_mean.uniform_(-0.5, 0.5); bn.running_var.uniform_(0.5, 1.5)
        for p in bn.parameters():
            p.requires_grad = False
        for use_res in (False, True):
            for relu in (True, False):
                x = torch.randn(shape, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                r = torch.randn(shape, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True) if use_res else None
                y = bn_act(x, bn, residual=r, relu=relu)
                assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
                ref = bn(x.float()) + (r.float() if use_res else 0)
                ref = torch.relu(ref) if relu else ref
                assert float((y.float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max()) + 1e-6
                g = torch.randn(shape, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                wrt = [x] + ([r] if use_res else [])
                got = torch.autograd.grad(y, wrt, g)
                mask = (y.float() > 0).float() if relu else torch.ones_like(ref)
                scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).view(1, -1, 1, 1)
                want = [g.float() * mask * scale] + ([g.float() * mask] if use_res else [])
                for a, b in zip(got, want):
                    assert a.dtype == torch.bfloat16
                    assert float((a.float() - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max()) + 1e-6


def test_trunk_in_bf16_stays_close_to_the_f32_trunk():
    """the C4 trunk of the bf16 configuration (MIOpen convolutions on bf16 tensors under autocast, bf16 frozen-BN passes)
    against the f32 trunk on the same weights: feature and input-side gradient within bf16's reach (stated: 3e-2 relative
    L2 through the 40 convolutions of ResNet50-C4; measured ~1e-2), f32 out, f32 weight gradients."""
    from ait_amd import ops
    from ait_amd.faster_rcnn import RCNNBackbone, resnet50
    from ait_amd.config import cfg
    torch.manual_seed(3)
    base = RCNNBackbone(cfg, resnet50()).cuda().eval()
    for m in base.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.running_var.uniform_(0.8, 1.2); m.running_mean.uniform_(-0.1, 0.1)
            for p in m.parameters():
                p.requires_grad = False
    x = torch.randn(2, 3, 160, 224, device="cuda")
    outs = {}
    for mode in ("f32", "bf16"):
        ops.set_matmul_dtype(mode)
        try:
            for p in base.parameters():
                p.grad = None
            y = base(x)[0]
            y.square().mean().backward()
            outs[mode] = (y.detach().float(), base.layer3[0].conv1.weight.grad.clone())
        finally:
            ops.set_matmul_dtype("f32")
    assert outs["bf16"][0].dtype == torch.float32 and outs["bf16"][1].dtype == torch.float32
    rel = lambda a, b: float((a - b).norm() / b.norm())
    ry, rg = rel(outs["bf16"][0], outs["f32"][0]), rel(outs["bf16"][1], outs["f32"][1])
    print("bf16 trunk vs f32 trunk: feature rel L2 %.3g, a layer3 weight gradient %.3g" % (ry, rg))
    assert 1e-5 < ry < 3e-2 and rg < 6e-2, (ry, rg)
