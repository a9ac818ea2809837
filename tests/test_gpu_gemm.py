"""fp32 MFMA GEMM through the C ABI vs a float64 torch reference (GPU)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(a, b, ta, tb):
    A = a.double().t() if ta else a.double()
    B = b.double().t() if tb else b.double()
    return A @ B


@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (256, 512, 512), (300, 200, 64), (58800 // 8, 512, 1024),
                                   (64, 64, 2048), (1, 1, 4), (130, 4, 20)])
def test_gemm_layouts(ta, tb, M, N, K):
    from ait_amd import ops
    torch.manual_seed(M * 7 + N * 3 + K)
    if ta and M % 4:
        pytest.skip("lda must be a multiple of 4")
    if not tb and N % 4:
        pytest.skip("ldb must be a multiple of 4")
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    c = ops.gemm(a, b, trans_a=ta, trans_b=tb)
    want = _ref(a, b, ta, tb)
    # exact-fp32 products, fp32 accumulate: error <= ~1e-7 * sum|a||b|
    bound = 4e-7 * (a.double().abs().t() if ta else a.double().abs()) @ \
        (b.double().abs().t() if tb else b.double().abs()) + 1e-6
    assert bool(((c.double() - want).abs() <= bound).all())


def test_gemm_epilogues():
    from ait_amd import ops
    torch.manual_seed(0)
    M, N, K = 384, 256, 128
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda")
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda")
    want = torch.relu((a.double() @ w.double().t()) + bias.double() + res.double())
    got = ops.gemm(a, w, bias=bias, residual=res, relu=True)
    assert torch.allclose(got.double(), want, rtol=1e-5, atol=1e-4)
    # accumulate into an existing C
    c = res.clone()
    ops.gemm(a, w, out=c, accumulate=True, alpha=0.5)
    assert torch.allclose(c.double(), res.double() + 0.5 * (a.double() @ w.double().t()), rtol=1e-5, atol=1e-4)
    # split-K weight-gradient shape: dW[N,K] = dy[Mtok,N]^T x[Mtok,K]
    Mtok = 4096
    dy = torch.randn(Mtok, N, device="cuda")
    x = torch.randn(Mtok, K, device="cuda")
    dw = ops.gemm(dy, x, trans_a=True, trans_b=False, split_k=8)
    assert torch.allclose(dw.double(), dy.double().t() @ x.double(), rtol=1e-5, atol=2e-3)
    # column-blocked C: [channel, token] product written into NCHW [p, ch, 64]
    P, CH = 5, 96
    dec = torch.randn(P * 64, K, device="cuda")
    wt = torch.randn(CH, K, device="cuda")
    bch = torch.randn(CH, device="cuda")
    out = ops.gemm(wt, dec, bias=bch, bias_row=True, c_colblk=64, c_batch_stride=CH * 64,
                   out_shape=(P, CH, 64))
    want = (dec.double() @ wt.double().t() + bch.double()).view(P, 64, CH).transpose(1, 2)
    assert torch.allclose(out.double(), want, rtol=1e-5, atol=1e-4)


def test_gemm_rejects_unaligned():
    from ait_amd import _lib, ops
    a = torch.randn(8, 6, device="cuda")
    b = torch.randn(8, 6, device="cuda")
    with pytest.raises(_lib.AitHipError):
        ops.gemm(a, b)
