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
                                   (64, 64, 2048), (1, 1, 4), (130, 4, 20),
                                   (40000, 64, 128),     # 256x64 tile (narrow outputs)
                                   (33000, 512, 64),     # 256x128 tile, slabs direct to LDS (K % 16 == 0)
                                   (33000, 512, 72),     # 256x128 tile, register-staged slabs (K tail)
                                   (33002, 520, 64),     # ragged M / N edges on both
                                   (70000, 1536, 32),    # persistent kernel: 3288 tiles, several per workgroup, 2 slabs each
                                   (66000, 132, 16)])    # ... one slab per tile (the ring crosses a tile every iteration)
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
    # exact-fp32 products, fp32 accumulate.  Worst case K * 2^-24 * sum|a||b|; observed maxima over
    # 30 seeds are 0.33..0.40e-6 * sum|a||b| for every tile / slab path, so 6e-7 is a tight fence
    bound = 6e-7 * (a.double().abs().t() if ta else a.double().abs()) @ \
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
    # the same on the 256x64 tile: SHBlock / fc weight gradient, 64 columns, 128 K-splits
    dy64, x64 = torch.randn(40000, 512, device="cuda"), torch.randn(40000, 64, device="cuda")
    dw64 = ops.gemm(dy64, x64, trans_a=True, trans_b=False, split_k=128)
    assert torch.allclose(dw64.double(), dy64.double().t() @ x64.double(), rtol=1e-5, atol=5e-3)
    # column-blocked C: [channel, token] product written into NCHW [p, ch, 64]
    P, CH = 5, 96
    dec = torch.randn(P * 64, K, device="cuda")
    wt = torch.randn(CH, K, device="cuda")
    bch = torch.randn(CH, device="cuda")
    out = ops.gemm(wt, dec, bias=bch, bias_row=True, c_colblk=64, c_batch_stride=CH * 64,
                   out_shape=(P, CH, 64))
    want = (dec.double() @ wt.double().t() + bch.double()).view(P, 64, CH).transpose(1, 2)
    assert torch.allclose(out.double(), want, rtol=1e-5, atol=1e-4)


def test_gemm_empty_reduction_is_the_epilogue_of_zero():
    """K == 0 (an empty token batch in a weight gradient, an empty feature in a forward): the product
    is zero, so C = bias / residual / zero -- not a crash (round-1 advisor finding)."""
    from ait_amd import ops
    M, N = 300, 136
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda")
    a, w = torch.zeros(M, 0, device="cuda"), torch.zeros(N, 0, device="cuda")
    assert torch.equal(ops.gemm(a, w), torch.zeros(M, N, device="cuda"))
    assert torch.equal(ops.gemm(a, w, bias=bias, residual=res, relu=True), torch.relu(res + bias))
    # weight-gradient layout with no tokens: dW = dy[0,N]^T x[0,K]
    dw = ops.gemm(torch.zeros(0, 512, device="cuda"), torch.zeros(0, 2048, device="cuda"), trans_a=True,
                  trans_b=False, split_k=8)
    assert tuple(dw.shape) == (512, 2048) and float(dw.abs().max()) == 0.0


def test_gemm_out_argument_is_validated():
    from ait_amd import _lib, ops
    a, w = torch.randn(64, 32, device="cuda"), torch.randn(48, 32, device="cuda")
    good = torch.empty(64, 48, device="cuda")
    ops.gemm(a, w, out=good)
    assert torch.allclose(good, a @ w.t(), atol=1e-4)
    wide = torch.empty(64, 96, device="cuda")
    ops.gemm(a, w, out=wide[:, :48])                       # row pitch 96: legal
    assert torch.allclose(wide[:, :48], a @ w.t(), atol=1e-4)
    for bad in (torch.empty(48, 64, device="cuda").t(), torch.empty(64, 47, device="cuda"),
                torch.empty(64, 48, device="cuda", dtype=torch.float64), torch.empty(64, 48)):
        with pytest.raises(_lib.AitHipError):
            ops.gemm(a, w, out=bad)


def test_gemm_rejects_unaligned():
    from ait_amd import _lib, ops
    a = torch.randn(8, 6, device="cuda")
    b = torch.randn(8, 6, device="cuda")
    with pytest.raises(_lib.AitHipError):
        ops.gemm(a, b)


@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("mode", ["bf16", "bf16_lds"])
@pytest.mark.parametrize("M,N,K", [(256, 128, 32), (512, 512, 512), (300, 200, 64), (7350, 512, 1024), (64, 64, 2048),
                                   (130, 4, 20), (33000, 1536, 512)])      # (the last: the persistent tile, rounding in registers)
def test_gemm_bf16_layouts(ta, tb, M, N, K, mode):
    """bf16 matrix-core variant: equals an fp64 product of the bf16-ROUNDED operands up to fp32
    accumulation error, and the fp32 product within bf16 input rounding (2^-8 relative per operand)."""
    from ait_amd import ops
    torch.manual_seed(M + N + K)
    if (ta and M % 4) or (not tb and N % 4):
        pytest.skip("leading dimension must be a multiple of 4")
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    ops.set_matmul_dtype(mode)
    try:
        c = ops.gemm(a, b, trans_a=ta, trans_b=tb)
    finally:
        ops.set_matmul_dtype("f32")
    ar, br = a.bfloat16().double(), b.bfloat16().double()
    want = (ar.t() if ta else ar) @ (br.t() if tb else br)
    bound = 4e-7 * (ar.abs().t() if ta else ar.abs()) @ (br.abs().t() if tb else br.abs()) + 1e-5
    assert bool(((c.double() - want).abs() <= bound).all())
    full = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    assert float((c.double() - full).norm() / full.norm()) < 6e-3


def test_gemm_bf16_epilogues_and_splitk():
    from ait_amd import ops
    torch.manual_seed(1)
    ops.set_matmul_dtype("bf16")
    try:
        M, N, K = 512, 256, 128
        a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        bias, res = torch.randn(N, device="cuda"), torch.randn(M, N, device="cuda")
        ref = a.bfloat16().double() @ w.bfloat16().double().t()
        got = ops.gemm(a, w, bias=bias, residual=res, relu=True)
        assert torch.allclose(got.double(), torch.relu(ref + bias.double() + res.double()), rtol=1e-5, atol=1e-4)
        dy, x = torch.randn(4096, N, device="cuda"), torch.randn(4096, K, device="cuda")
        dw = ops.gemm(dy, x, trans_a=True, trans_b=False, split_k=8)
        assert torch.allclose(dw.double(), dy.bfloat16().double().t() @ x.bfloat16().double(), rtol=1e-5, atol=2e-3)
    finally:
        ops.set_matmul_dtype("f32")


def test_gemm_large_tile_direct_to_lds_epilogues():
    """The 256x128 tile with direct-to-LDS slabs (K % 16 == 0, >= 512 workgroups) under every
    epilogue the AIT uses: bias + residual + ReLU, accumulate, ReLU-backward gate, column-blocked
    (NCHW) store, split-K atomics -- against fp64 products."""
    from ait_amd import ops
    torch.manual_seed(11)
    M, N, K = 33000, 512, 256
    a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    bias, res = torch.randn(N, device="cuda"), torch.randn(M, N, device="cuda")
    ref = a.double() @ w.double().t()
    tol = dict(rtol=1e-5, atol=2e-4)
    assert torch.allclose(ops.gemm(a, w, bias=bias, residual=res, relu=True).double(),
                          torch.relu(ref + bias.double() + res.double()), **tol)
    c = res.clone()
    ops.gemm(a, w, out=c, accumulate=True, alpha=0.5)
    assert torch.allclose(c.double(), res.double() + 0.5 * ref, **tol)
    # dgrad layout (B stored [K, N]) with the ReLU-backward gate
    wk = torch.randn(K, N, device="cuda")
    act = torch.randn(M, N, device="cuda")
    got = ops.gemm_relu_bwd(a, wk, act)
    assert torch.allclose(got.double(), (a.double() @ wk.double()) * (act > 0), **tol)
    # weight-gradient layout, split-K atomics: dW[N2, K2] = dy[T, N2]^T x[T, K2]
    T_, N2, K2 = 65536, 2048, 512
    dy, x = torch.randn(T_, N2, device="cuda"), torch.randn(T_, K2, device="cuda")
    dw = ops.gemm(dy, x, trans_a=True, trans_b=False, split_k=16)
    assert torch.allclose(dw.double(), dy.double().t() @ x.double(), rtol=1e-5, atol=3e-2)
    # column-blocked C: [channel, token] product written into NCHW [p, ch, 64]
    P, CH = 520, 1024
    dec, wt, bch = torch.randn(P * 64, K, device="cuda"), torch.randn(CH, K, device="cuda"), torch.randn(CH, device="cuda")
    out = ops.gemm(wt, dec, bias=bch, bias_row=True, c_colblk=64, c_batch_stride=CH * 64, out_shape=(P, CH, 64))
    want = (dec.double() @ wt.double().t() + bch.double()).view(P, 64, CH).transpose(1, 2)
    assert torch.allclose(out.double(), want, **tol)


def _abs_bound(a, b, ta, tb):
    return 6e-7 * (a.double().abs().t() if ta else a.double().abs()) @ \
        (b.double().abs().t() if tb else b.double().abs()) + 1e-6


@pytest.mark.parametrize("M,N,K,ta,tb", [
    (19200, 512, 2048, False, True),      # layer4 1x1: 300 tiles < one round of 512 -> every tile cut (stream-K only)
    (9576, 512, 1024, False, False),      # co-attention token product: 152 tiles
    (76800, 1536, 512, False, True),      # 3600 tiles = 7 rounds + 16 tiles: the 16 are cut 8 ways
    (58800, 1024, 512, False, True),      # 49-row sequences: 230 x 8 tiles, 38 left per XCD
    (40000, 512, 1024, False, False),     # ragged last M tile inside a cut tile
    (2048, 4096, 3072, True, False),      # K-outer A without split-K: 256 tiles of 192 slabs
])
def test_gemm_stream_k_work_list(M, N, K, ta, tb):
    """Tile counts that leave the persistent kernel's last round badly filled: the slabs of those tiles
    are spread over all workgroups and the owner of each tile adds the published partial tiles in a
    fixed order (gemm_f32_impl.h) -- same fp32 error fence as whole tiles, bias + ReLU applied once,
    and bit-identical from launch to launch (no atomics)."""
    from ait_amd import ops
    torch.manual_seed(M + N + K)
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    bias = torch.randn(N, device="cuda")
    c = ops.gemm(a, b, trans_a=ta, trans_b=tb)
    assert bool(((c.double() - _ref(a, b, ta, tb)).abs() <= _abs_bound(a, b, ta, tb)).all())
    c2 = ops.gemm(a, b, trans_a=ta, trans_b=tb)
    assert torch.equal(c, c2)
    r = ops.gemm(a, b, trans_a=ta, trans_b=tb, bias=bias, relu=True)
    want = torch.relu(_ref(a, b, ta, tb) + bias.double())
    assert bool(((r.double() - want).abs() <= _abs_bound(a, b, ta, tb)).all())
    if not ta:
        res = torch.randn(M, N, device="cuda")
        r = ops.gemm(a, b, trans_a=ta, trans_b=tb, residual=res)
        assert bool(((r.double() - _ref(a, b, ta, tb) - res.double()).abs() <= _abs_bound(a, b, ta, tb)).all())


def test_gemm_stream_k_back_to_back_launches_reuse_the_flags():
    """Every flag a stream-K launch sets is cleared by the workgroup that consumes it: many launches in a
    row on one stream (different shapes sharing the scratch) stay correct."""
    from ait_amd import ops
    torch.manual_seed(5)
    shapes = [(19200, 512, 2048), (9576, 1024, 512), (19200, 2048, 512), (76800, 1536, 512)]
    ops_in = [(torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")) for M, N, K in shapes]
    outs = []
    for rep in range(3):
        for a, w in ops_in:
            outs.append(ops.gemm(a, w, trans_b=True))
    for i, (a, w) in enumerate(ops_in):
        want = a.double() @ w.double().t()
        for rep in range(3):
            got = outs[rep * len(shapes) + i]
            assert bool(((got.double() - want).abs() <= _abs_bound(a, w, False, True)).all())
            assert torch.equal(got, outs[i])


def test_gemm_persistent_kernels_share_the_chip():
    """Two streams run persistent GEMMs at the same time: neither gets the whole chip, so part of each grid
    becomes resident only when other workgroups exit (what RCCL's kernels do to the backward GEMMs of a
    data-parallel step).  Tiles are handed out dynamically and stream-K waits only point at higher workgroup
    ids, so this is a speed matter, never a correctness or liveness one; each stream has its own scratch."""
    from ait_amd import ops
    torch.manual_seed(9)
    shapes = [(76800, 512, 2048), (19200, 2048, 512), (58800, 1024, 512), (33000, 512, 1024)]
    data = [(torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")) for M, N, K in shapes]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    outs = []
    for rep in range(3):
        for i, (a, w) in enumerate(data):
            with torch.cuda.stream(streams[(i + rep) % 2]):
                outs.append((i, ops.gemm(a, w, trans_b=True)))
    torch.cuda.synchronize()
    for i, got in outs:
        a, w = data[i]
        want = a.double() @ w.double().t()
        assert bool(((got.double() - want).abs() <= _abs_bound(a, w, False, True)).all()), shapes[i]


def test_batched_gemm_layouts_and_autograd():
    """ait_gemm_f32_batched on the COCO co-attention shapes (blocks_coatt_transformer_sk.py:86-110: 64 query
    tokens x 2394 image tokens x 512 channels) against torch.bmm in float64, forward and through _Bmm's backward;
    2394 % 4 != 0 exercises the padded row pitch."""
    from ait_amd import ops
    from ait_amd.faster_rcnn import _Bmm
    torch.manual_seed(5)
    bz, nq, ni, ch = 3, 64, 2394, 512
    rho = torch.randn(bz, nq, ch, device="cuda", requires_grad=True)
    phi = torch.randn(bz, ni, ch, device="cuda", requires_grad=True)
    eq = torch.randn(bz, nq, ch, device="cuda", requires_grad=True)
    ei = torch.randn(bz, ni, ch, device="cuda", requires_grad=True)
    rel = _Bmm.apply(rho, phi, False, True, 1.0)
    a = _Bmm.apply(rel, eq, True, False, 1.0 / nq)
    b = _Bmm.apply(rel, ei, False, False, 1.0 / ni)
    ca, cb = torch.randn_like(a), torch.randn_like(b)
    g = torch.autograd.grad([a, b], [rho, phi, eq, ei], [ca, cb])
    R, P, Q, I = (t.detach().double().requires_grad_(True) for t in (rho, phi, eq, ei))
    rel_r = R @ P.transpose(1, 2)
    a_r = rel_r.transpose(1, 2) @ Q / nq
    b_r = rel_r @ I / ni
    g_r = torch.autograd.grad([a_r, b_r], [R, P, Q, I], [ca.double(), cb.double()])
    rel_err = lambda x, y: float((x.double() - y).norm() / y.norm())
    assert tuple(rel.shape) == (bz, nq, ni) and rel.stride(1) % 4 == 0
    assert rel_err(rel, rel_r) < 1e-6 and rel_err(a, a_r) < 1e-6 and rel_err(b, b_r) < 1e-6
    for x, y in zip(g, g_r):
        assert rel_err(x, y) < 2e-6
    # accumulate into an existing output; a non-padded operand is re-laid-out, not misread
    out = ops.bgemm(rho.detach(), phi.detach().contiguous(), False, True)
    ops.bgemm(rho.detach(), phi.detach(), False, True, alpha=0.5, out=out, accumulate=True)
    assert rel_err(out, 1.5 * rel_r) < 1e-6


def test_gemm_whole_tile_mode_without_scratch():
    """No scheduler workspace in the launch context (ait_launch_ctx.sched_ws == NULL): static work lists, whole
    tiles only -- the same products within the fp32 bound, and the library touches no memory but its operands."""
    from ait_amd import _lib, ops
    torch.manual_seed(3)
    saved = _lib.USE_SCHED_WS
    try:
        for M, N, K in ((19200, 512, 2048), (76800, 1536, 512), (33000, 512, 64)):
            a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
            _lib.USE_SCHED_WS = False
            c0 = ops.gemm(a, w, trans_b=True)
            _lib.USE_SCHED_WS = True
            c1 = ops.gemm(a, w, trans_b=True)
            want = a.double() @ w.double().t()
            bound = 6e-7 * (a.double().abs() @ w.double().abs().t()) + 1e-6
            for c in (c0, c1):
                assert bool(((c.double() - want).abs() <= bound).all()), (M, N, K)
    finally:
        _lib.USE_SCHED_WS = saved


def test_gemm_scheduler_workspace_contract():
    """include/ait_hip.h, ait_launch_ctx: the scheduler scratch is the CALLER's -- sized by
    ait_gemm_workspace_bytes(), prepared once by ait_gemm_workspace_init(), reusable by stream-ordered launches
    (the control words are self-cleaning: a second launch on the same workspace gives the same bits), and a
    workspace that is too small is refused with AIT_EWORKSPACE, never silently ignored."""
    import ctypes
    from ait_amd import _lib
    L = _lib.lib()
    nbytes = int(L.ait_gemm_workspace_bytes())
    assert nbytes > 16384
    M, N, K = 19200, 512, 2048                 # 300 tiles of 256x128: a stream-K launch
    torch.manual_seed(11)
    a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    ws.fill_(0xAB)                             # (a fresh allocation holds anything)
    st = _lib.cur_stream(a.device)
    assert L.ait_gemm_workspace_init(ctypes.c_void_p(ws.data_ptr()), nbytes, st) == 0
    p = lambda t: ctypes.c_void_p(t.data_ptr())

    def run(ctx):
        out = torch.empty(M, N, device="cuda")
        rc = L.ait_gemm_f32(0, 1, M, N, K, 1.0, p(a), K, p(w), K, p(out), N, None, None, 0, 1, 0, 0,
                            None if ctx is None else ctypes.byref(ctx), st)
        return rc, out

    ctx = _lib.LaunchCtx()
    ctx.sched_ws, ctx.sched_ws_bytes = ws.data_ptr(), nbytes
    rc, c1 = run(ctx)
    assert rc == 0
    rc, c2 = run(ctx)
    assert rc == 0 and torch.equal(c1, c2)
    want = a.double() @ w.double().t()
    assert bool(((c1.double() - want).abs() <= 6e-7 * (a.double().abs() @ w.double().abs().t()) + 1e-6).all())
    rc, c0 = run(None)                          # no context at all: legal
    assert rc == 0 and bool(((c0.double() - want).abs() <= 6e-7 * (a.double().abs() @ w.double().abs().t()) + 1e-6).all())
    small = _lib.LaunchCtx()
    small.sched_ws, small.sched_ws_bytes = ws.data_ptr(), 1 << 20
    rc, _ = run(small)
    assert rc == -2
    assert L.ait_gemm_workspace_init(None, nbytes, st) == -1


def test_gemm_stream_k_inside_graph_capture():
    """A stream-K-sized product launched inside hipGraph capture: nothing in the launch path allocates,
    memsets synchronously or synchronises (the scheduler scratch was the caller's before the capture began), and
    the replayed graph reproduces the eager result bit for bit."""
    from ait_amd import _lib, ops
    M, N, K = 19200, 512, 2048
    torch.manual_seed(12)
    a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    out = torch.empty(M, N, device="cuda")
    eager = ops.gemm(a, w, trans_b=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.gemm(a, w, trans_b=True, out=out)            # warm-up on the capture stream: its workspace now exists
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    out.zero_()
    with torch.cuda.graph(g, stream=side):
        ops.gemm(a, w, trans_b=True, out=out)
    out.zero_()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    del g
    _lib.release_sched_workspaces()


@pytest.mark.parametrize("M,N,K", [(76800, 2048, 512), (19200, 512, 2048), (300, 200, 64), (33002, 520, 64)])
def test_gemm_column_sums_in_the_epilogue(M, N, K):
    """AIT_GEMM_COLSUM: the bias gradient of the layer an input gradient flows into (SubLayers.py:181), formed in
    the product's epilogue -- with the ReLU-backward gate, on whole and stream-K tiles, ragged edges included --
    equals the column sums of the stored result."""
    import ctypes
    from ait_amd import _lib
    torch.manual_seed(M + K)
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(K, N, device="cuda")
    act = torch.randn(M, N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    cs = torch.zeros(N, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    with torch.cuda.device(a.device):
        rc = _lib.lib().ait_gemm_f32(0, 0, M, N, K, 1.0, p(a), K, p(w), N, p(out), N, p(cs), p(act),
                                     _lib.GEMM_MASK_POS | _lib.GEMM_COLSUM, 1, 0, 0, _lib.launch_ctx(a.device),
                                     _lib.cur_stream(a.device))
    assert rc == 0
    want = (a.double() @ w.double()) * (act > 0)
    assert bool(((out.double() - want).abs() <= _abs_bound(a, w, False, False)).all())
    ref = out.double().sum(0)
    assert float((cs.double() - ref).abs().max()) <= 1e-5 * float(out.double().abs().sum(0).max()) + 1e-4
    # a second call ADDS
    with torch.cuda.device(a.device):
        rc = _lib.lib().ait_gemm_f32(0, 0, M, N, K, 1.0, p(a), K, p(w), N, p(out), N, p(cs), None,
                                     _lib.GEMM_COLSUM, 1, 0, 0, _lib.launch_ctx(a.device), _lib.cur_stream(a.device))
    assert rc == 0
    ref2 = ref + (a.double() @ w.double()).sum(0)
    assert float((cs.double() - ref2).abs().max()) <= 1e-5 * float((a.double().abs() @ w.double().abs()).sum(0).max()) + 1e-4


def _native(flag):
    """context: dense products on v_mfma_f32_32x32x2_f32 (ait_launch_ctx::flags = AIT_CTX_NATIVE_F32)"""
    import contextlib
    from ait_amd import _lib

    @contextlib.contextmanager
    def cm():
        old = _lib.NATIVE_F32
        _lib.NATIVE_F32 = flag
        try:
            yield
        finally:
            _lib.NATIVE_F32 = old
    return cm()


@pytest.mark.parametrize("M,N,K,ta,tb", [(33000, 1536, 512, False, True), (33000, 512, 2048, False, False),
                                         (512, 2048, 33024, True, False)])
def test_split_bf16_products_are_as_close_to_float64_as_the_f32_instruction(M, N, K, ta, tb, record_property):
    """The default product form (every f32 operand value = three bf16 values exactly, six bf16 MFMAs per block,
    f32 accumulate; include/ait_hip.h ait_launch_ctx::flags) against float64, beside the same launch on
    v_mfma_f32_32x32x2_f32: the split form's error is the f32 instruction's (the dropped partial products are
    < 2^-23 |a b|).  Operands with a wide spread of magnitudes and signs."""
    from ait_amd import ops
    torch.manual_seed(M + N + K)
    a = torch.randn((K, M) if ta else (M, K), device="cuda") * torch.exp(2 * torch.randn((K, 1) if ta else (1, K), device="cuda"))
    b = torch.randn((N, K) if tb else (K, N), device="cuda") * torch.exp(2 * torch.randn((1, K) if tb else (K, 1), device="cuda"))
    sk = 16 if ta else 1
    want = _ref(a, b, ta, tb)
    mag = (a.double().abs().t() if ta else a.double().abs()) @ (b.double().abs().t() if tb else b.double().abs())
    got = ops.gemm(a, b, trans_a=ta, trans_b=tb, split_k=sk)
    with _native(True):
        nat = ops.gemm(a, b, trans_a=ta, trans_b=tb, split_k=sk)
    e_split = float(((got.double() - want).abs() / mag).max())
    e_nat = float(((nat.double() - want).abs() / mag).max())
    rms_split = float(((got.double() - want) / mag).square().mean().sqrt())
    rms_nat = float(((nat.double() - want) / mag).square().mean().sqrt())
    record_property("max_err_over_sum_abs_split", e_split)
    record_property("max_err_over_sum_abs_f32_instruction", e_nat)
    assert e_nat <= 6e-6 and e_split <= 6e-6            # (K * 2^-24 worst case; these operands span e^+-6)
    assert e_split <= 1.5 * e_nat + 2e-8 and rms_split <= 1.25 * rms_nat + 1e-9
    assert float(((got - nat).double().abs() / mag).max()) <= e_split + e_nat + 1e-9


def test_split_bf16_products_special_values():
    """zeros, denormal-sized, huge and tiny magnitudes, exact powers of two and values with all 24 significant bits
    set go through the 3-way split exactly; a non-finite operand gives a non-finite result"""
    from ait_amd import ops
    torch.manual_seed(5)
    M, N, K = 33024, 512, 64
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(N, K, device="cuda")
    a[:, 0] = 0.0
    a[:, 1] = 1e-30
    b[:, 1] = 1e30
    a[:, 2] = 16777215.0                 # 2^24 - 1: all significant bits set
    b[:, 2] = 1.0 / 3.0
    a[:, 3] = -0.5
    a[:, 4] = 3e38
    b[:, 4] = 1e-38
    a[:, 5] = 1e-41                      # denormal (flushed by both forms or carried: compared with a tolerance)
    got = ops.gemm(a, b)
    want = a.double() @ b.double().t()
    mag = a.double().abs() @ b.double().abs().t()
    assert float(((got.double() - want).abs() / mag).max()) <= 6e-7
    # a FINITE, overflow-free product never becomes NaN / inf: operands of either sign up to the largest bf16 value
    # (3.3895e38: the nearest-rounded high plane of anything larger would be an infinity -- the documented limit of the
    # split form, include/ait_hip.h) against operands that keep every partial product and the sum in range
    a2 = torch.randn(512, 64, device="cuda")
    b2 = torch.randn(512, 64, device="cuda") * 1e-3
    a2[:, 0], a2[:, 1], a2[:, 2] = 3e38, -3e38, 3.3895e38
    b2[:, 0], b2[:, 1], b2[:, 2] = 1e-38, 1e-38, -2.5e-39
    got2 = ops.gemm(a2, b2)
    want2 = a2.double() @ b2.double().t()
    assert bool(torch.isfinite(got2).all())
    assert float(((got2.double() - want2).abs() / (a2.double().abs() @ b2.double().abs().t())).max()) <= 6e-7
    a[7, 9] = float("inf")
    got = ops.gemm(a, b)
    assert not bool(torch.isfinite(got[7]).any()) and bool(torch.isfinite(got[8]).all())


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_products_are_reproducible_launch_to_launch(mode):
    """Regression: built with hipcc's SLP vectoriser on, the bias epilogue of the persistent tile became packed
    v_pk_fma_f32 (bias broadcast from the high half of a register pair through op_sel) and returned, on some launches
    only, 16-element row segments off by the difference of two bias values -- lanes 48-63 of four accumulator
    registers, a handful of the 4800 tiles of a launch, on launches WITH a bias only (ait_amd/build.py,
    -fno-slp-vectorize).  A launch is compared with a second launch of itself (bit for bit: these launches have no
    atomics) and with the f32 instruction's result."""
    from ait_amd import ops
    torch.manual_seed(2)
    for (M, N, K) in [(19200, 2048, 512), (76800, 1024, 512), (58800, 2048, 512)]:
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda")
        bias = torch.randn(N, device="cuda")
        with _native(True):
            nat = ops.gemm(a, w, bias=bias, relu=True)
        scale = float(nat.abs().max())
        ops.set_matmul_dtype(mode)
        try:
            first = ops.gemm(a, w, bias=bias, relu=True)
            for _ in range(3):
                again = ops.gemm(a, w, bias=bias, relu=True)
                assert torch.equal(first, again)
        finally:
            ops.set_matmul_dtype("f32")
        assert float((first - nat).abs().max()) <= (2e-6 if mode == "f32" else 2e-2) * scale
        del a, w, nat, first, again


@pytest.mark.parametrize("rows,cols", [(512, 1024), (1536, 512), (2048, 520)])
def test_p3_planes_are_an_exact_re_encoding(rows, cols):
    """ait_p3_split (include/ait_hip.h "P3"): the three bf16 planes of every value sum to it EXACTLY, in the
    group-of-eight interleaved layout, for a weight and for its transpose; wide magnitude spread, all 24
    significant bits set, zeros."""
    from ait_amd import ops
    torch.manual_seed(rows + cols)
    w = torch.randn(rows, cols, device="cuda") * torch.exp(3 * torch.randn(rows, 1, device="cuda"))
    w[0, :8] = torch.tensor([0.0, 16777215.0, -16777215.0, 1.0, -0.5, 3e38, 1e-30, 1.0 / 3.0], device="cuda")
    p = ops.p3_split(w)
    assert p.shape == (rows, cols // 8, 3, 8)
    back = p.float().sum(dim=2).reshape(rows, cols)        # h + m, then + l: every partial sum is exact in f32
    assert torch.equal(back, w)
    pt = ops.p3_split(w, transpose=True)
    assert pt.shape == (cols, rows // 8, 3, 8)
    assert torch.equal(pt.float().sum(dim=2).reshape(cols, rows), w.t())
    # planes are ordered by magnitude: |m| <= 2^-8 |h|, |l| <= 2^-16 |h| (round to nearest)
    h, m, l = p.float().unbind(dim=2)
    assert bool((m.abs() <= h.abs() * 2.0 ** -8 + 1e-45).all()) and bool((l.abs() <= h.abs() * 2.0 ** -16 + 1e-45).all())


@pytest.mark.parametrize("M,N,K", [(76800, 1536, 512), (58800, 2048, 512), (33000, 512, 2048), (76800, 1024, 512),
                                   (40100, 768, 1024)])
def test_gemm_with_pre_split_weight_vs_float64(M, N, K):
    """ait_gemm_f32_p3 (the 256x256 tile whose weight operand arrives as P3 planes; csrc/gemm_p3.hip): forward and
    input-gradient forms, every epilogue the transformer uses on it, against float64 with the f32 error fence of the
    other product tests; bit-identical launch to launch; and as close to float64 as the product that splits both
    operands in registers."""
    from ait_amd import ops
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda") * torch.exp(2 * torch.randn(1, K, device="cuda"))
    w = torch.randn(N, K, device="cuda") * torch.exp(2 * torch.randn(1, K, device="cuda"))
    bias = torch.randn(N, device="cuda")
    wp = ops.p3_split(w)
    ref = a.double() @ w.double().t()
    mag = a.double().abs() @ w.double().abs().t()
    bound = 3e-6 * mag + 1e-6        # (K * 2^-24 worst case of the f32 accumulation; these operands span e^+-6)
    c = ops.gemm_p3(a, wp)
    assert bool(((c.double() - ref).abs() <= bound).all())
    assert torch.equal(c, ops.gemm_p3(a, wp))
    raw = ops.gemm(a, w)
    e_p3 = float(((c.double() - ref).abs() / mag).max())
    e_raw = float(((raw.double() - ref).abs() / mag).max())
    assert e_p3 <= 1.5 * e_raw + 2e-8
    r = ops.gemm_p3(a, wp, bias=bias, relu=True)
    assert bool(((r.double() - torch.relu(ref + bias.double())).abs() <= bound).all())
    res = torch.randn(M, N, device="cuda")
    r = ops.gemm_p3(a, wp, residual=res)
    assert bool(((r.double() - ref - res.double()).abs() <= bound).all())
    del r, raw
    # input gradient: dy [M, N] @ W [N, K] with W^T pre-split, gated by a saved activation, column sums in the epilogue
    dy = torch.randn(M, N, device="cuda")
    wtp = ops.p3_split(w, transpose=True)
    act = torch.randn(M, K, device="cuda")
    cs = torch.zeros(K, device="cuda")
    dx = ops.gemm_p3(dy, wtp, residual=act, mask_pos=True, colsum=cs)
    want = (dy.double() @ w.double()) * (act > 0)
    mag2 = dy.double().abs() @ w.double().abs()
    assert bool(((dx.double() - want).abs() <= 3e-6 * mag2 + 1e-6).all())
    assert torch.allclose(cs.double(), want.sum(0), rtol=1e-4, atol=1e-3 * float(want.abs().sum(0).max()))


def _all_ones(shape, gen, lo=-2, hi=2):
    """positive f32 values whose 24 significant bits are all set, times random powers of two: the operands for which a
    TRUNCATED three-plane split drops the most (every plane has all its bits set, every dropped term has the same sign)"""
    e = torch.randint(lo, hi + 1, shape, device="cuda", generator=gen).float()
    return (2.0 - 2.0 ** -23) * torch.exp2(e)


@pytest.mark.parametrize("kind", ["all_ones", "positive_random"])
@pytest.mark.parametrize("M,N,K,ta,sk", [(33000, 512, 512, False, 1), (33000, 512, 2048, False, 1), (512, 2048, 76800, True, 16)])
def test_split_products_same_signed_operands_have_no_bias(M, N, K, ta, sk, kind, record_property):
    """The adversarial case for a split-product form, not random data: every operand value POSITIVE (all 24 significant
    bits set, or uniform in [0.5, 1.5)), so that the partial products the six-term form drops and whatever the bf16
    pipe's accumulator chops all have one sign, over reductions of 512, 2048 and 76800 (16 splits of 4800) terms.
    Measured history (profiles/r04_split_bias.txt): with the planes formed by TRUNCATION the result lay below the exact
    sum by up to 1.4e-5 of it at K = 76800 (the small same-signed planes are chopped when the pipe aligns them to a
    large running sum) -- 30x the 2^-21 the dropped terms alone explain; with the planes rounded to NEAREST (the product
    form since round 4: zero-mean m and l planes, dropped terms <= 2^-23 |a b|) the mean error is <= 1e-7 and the
    scatter is below the f32 instruction's own.  Asserted: mean error within 2^-21, worst error within 1.5x the f32
    instruction's on the same data."""
    from ait_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(M + K)
    def make(shape):
        if kind == "all_ones":
            return _all_ones(shape, gen)
        return torch.rand(shape, device="cuda", generator=gen) + 0.5
    a = make((K, M) if ta else (M, K))
    b = make((K, N) if ta else (N, K))
    ref = (a.double().t() @ b.double()) if ta else (a.double() @ b.double().t())      # all terms positive: ref = sum |a||b|
    got = ops.gemm(a, b, trans_a=ta, trans_b=not ta, split_k=sk)
    with _native(True):
        nat = ops.gemm(a, b, trans_a=ta, trans_b=not ta, split_k=sk)
    e_split = (got.double() - ref) / ref
    e_nat = (nat.double() - ref) / ref
    record_property("mean_error_split", float(e_split.mean()))
    record_property("mean_error_f32_instruction", float(e_nat.mean()))
    record_property("worst_error_split", float(e_split.abs().max()))
    record_property("worst_error_f32_instruction", float(e_nat.abs().max()))
    bound = 2.0 ** -21
    assert abs(float(e_split.mean())) <= bound
    assert float(e_split.abs().max()) <= 1.5 * float(e_nat.abs().max()) + 2.0 ** -22
    if not ta:
        c = ops.gemm_p3(a, ops.p3_split(b))         # the pre-split-weight product: same planes, B's formed by the conversion pass
        e3 = (c.double() - ref) / ref
        assert abs(float(e3.mean())) <= bound and float(e3.abs().max()) <= 1.5 * float(e_nat.abs().max()) + 2.0 ** -22


def test_every_epilogue_and_tile_family_the_step_launches_is_reproducible():
    """Widened regression of test_products_are_reproducible_launch_to_launch (timing-dependent wrong lanes are also what
    a race looks like): every (tile family x epilogue) form the training step launches -- the persistent 256x128 split
    tile with store / bias+ReLU / residual / ReLU-mask+column-sum / split-K atomics (compared against the f32
    instruction; atomics reorder sums), the 256x256 pre-split-weight tile with the same epilogues, the 256x64 tile, the
    register-staged 128x128 and 64x64 tiles -- a launch against three more launches of itself, bit for bit, and against
    the f32 instruction's result."""
    from ait_amd import ops
    torch.manual_seed(9)
    cases = [(76800, 1536, 512), (58800, 2048, 512), (19200, 512, 2048), (76800, 64, 512), (3000, 640, 256), (256, 512, 1024)]
    for (M, N, K) in cases:
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda")
        bias = torch.randn(N, device="cuda")
        res = torch.randn(M, N, device="cuda")
        forms = {
            "store": lambda: ops.gemm(a, w),
            "bias_relu": lambda: ops.gemm(a, w, bias=bias, relu=True),
            "residual": lambda: ops.gemm(a, w, residual=res),
        }
        if K % 16 == 0 and N >= 256 and M >= 512:
            wp = ops.p3_split(w)
            wtp = ops.p3_split(w, transpose=True)
            dy = torch.randn(M, N, device="cuda")
            act = torch.randn(M, K, device="cuda")
            forms["p3_store"] = lambda: ops.gemm_p3(a, wp)
            forms["p3_bias_relu"] = lambda: ops.gemm_p3(a, wp, bias=bias, relu=True)
            forms["p3_residual"] = lambda: ops.gemm_p3(a, wp, residual=res)
            forms["p3_mask"] = lambda: ops.gemm_p3(dy, wtp, residual=act, mask_pos=True)
            forms["mask"] = lambda: ops.gemm_relu_bwd(dy, w, act)
        for name, f in forms.items():
            first = f()
            for _ in range(3):
                assert torch.equal(first, f()), (M, N, K, name)
            with _native(True):
                nat = f() if not name.startswith("p3") else None
            if nat is not None:
                assert float((first - nat).abs().max()) <= 3e-6 * float(nat.abs().max()) + 1e-6, (M, N, K, name)
        del a, w, res, forms


@pytest.mark.gpu
# (12290 ..., 50000 ...: the 256 x 256 x 64 tile; then thin last rounds of it -- 257 row tiles x 2 column tiles = two rounds of
# 256 + 2 tiles; 258 x 8 = eight rounds + 16 tiles, ragged rows; 291 x 4 = four rounds + 140 tiles -- and, the last two, long
# reductions with such a round: the K-cut (pieces first, a finishing launch that adds them in piece order: hence the
# reproducibility check), 2 leftover tiles in 8 pieces each and 28 leftover tiles in 8, ragged rows)
@pytest.mark.parametrize("M,N,K", [(256, 128, 32), (1000, 256, 64), (4097, 512, 512), (19200, 2048, 512), (777, 1536, 2048),
                                   (12290, 1024, 1024), (50000, 512, 2048), (65792, 512, 1024), (65900, 2048, 512),
                                   (74400, 1024, 512), (65792, 512, 4096), (34500, 1024, 4608)])
def test_bf16_storage_gemm_against_float64(M, N, K):
    """ait_gemm_bf16s (csrc/gemm_bf16s.hip): bf16 operands stored in memory, f32 accumulate; against the float64 product of
    the SAME bf16 values (the only error left is the f32 accumulation), every epilogue: bias, ReLU, residual, the two
    gates, f32 and bf16 outputs; ragged M (rows past M are neither read as results nor written)."""
    from ait_amd import ops
    torch.manual_seed(M + N + K)
    a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    a16, b16 = ops.to_bf16(a), ops.to_bf16(b)
    assert torch.equal(a16, a.to(torch.bfloat16)) and torch.equal(ops.to_bf16(b, transpose=True), b.t().contiguous().to(torch.bfloat16))
    bias, res = torch.randn(N, device="cuda"), torch.randn(M, N, device="cuda")
    ref = a16.double() @ b16.double().t()
    mag = a16.double().abs() @ b16.double().abs().t()
    guard = torch.full((M + 3, N), 7.0, device="cuda")                      # rows past M must stay untouched
    y32, y16 = ops.gemm_bf16s(a16, b16, out32=guard[:M], want16=True)
    assert float(((y32.double() - ref).abs() / mag).max()) < 2e-6 and bool((guard[M:] == 7.0).all())
    assert torch.equal(y16, y32.to(torch.bfloat16))
    again, _ = ops.gemm_bf16s(a16, b16)
    assert torch.equal(again, y32)                                           # (also where a tile's pieces meet: fixed order)
    y, _ = ops.gemm_bf16s(a16, b16, bias=bias, relu=True)
    assert float(((y.double() - (ref + bias.double()).clamp_min(0)).abs() / (mag + 1)).max()) < 2e-6
    y, _ = ops.gemm_bf16s(a16, b16, bias=bias, residual=res)
    assert float(((y.double() - (ref + bias.double() + res.double())).abs() / (mag + 1)).max()) < 2e-6
    y, _ = ops.gemm_bf16s(a16, b16, residual=res, mask_pos=True)
    assert float(((y.double() - ref * (res > 0)).abs() / mag).max()) < 2e-6
    g16 = res.to(torch.bfloat16)
    _, y = ops.gemm_bf16s(a16, b16, gate16=g16, mask_pos=True, want32=False, want16=True)
    want = (ref * (g16.float() > 0)).float().to(torch.bfloat16)
    assert float((y.float() - want.float()).abs().max()) <= 2.0 ** -7 * float(want.float().abs().max())
    # (zero exactly where the gate is shut; a sum whose float64 value is ~1e-9 of its terms may also be an exact 0 in f32 -- a
    # handful of elements among 1e8)
    assert int(((y.float() == 0) != ((g16.float() <= 0) | (want.float() == 0))).sum()) <= 4
    assert bool((y.float()[g16.float() <= 0] == 0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,pitch", [(1, 4, 4), (130, 12, 12), (1000, 2048, 2048), (76800, 1024, 1024), (5000, 512, 1536),
                                             (4097, 20, 28)])
def test_bf16_column_sums_against_float64(rows, cols, pitch):
    """ait_colsum_bf16 (the bias gradients of the bf16-storage mode): out += column sums of a bf16 matrix, any row count,
    pitched rows, both load widths (16-byte when columns, pitch and base allow, else 8-byte); ACCUMULATES into out."""
    from ait_amd import ops
    torch.manual_seed(rows + cols)
    buf = torch.randn(rows, pitch, device="cuda").to(torch.bfloat16)
    x = buf[:, :cols]
    out = torch.full((cols,), 2.0, device="cuda")
    ops.colsum_bf16(x, out)
    want = x.double().sum(0) + 2.0
    scale = x.double().abs().sum(0) + 2.0
    assert float(((out.double() - want).abs() / scale).max()) < 2e-6
    if cols >= 8 and pitch > cols:      # a base that is only 8-byte aligned takes the narrow form
        y = buf[:, 4:4 + (cols // 8) * 8 - 4] if (cols // 8) * 8 - 4 > 0 and ((cols // 8) * 8 - 4) % 4 == 0 else None
        if y is not None and y.shape[1] % 4 == 0 and y.shape[1] > 0:
            assert float(((ops.colsum_bf16(y).double() - y.double().sum(0)).abs() / (y.double().abs().sum(0) + 1e-9)).max()) < 2e-6


@pytest.mark.gpu
# (4800, ..., 7): 150 slabs of 32 rows in ranges of 22 and 21; (66560, ..., 7): 1040 slabs of 64 in ranges of 149 and 148 -- one
# round of the 256 x 256 tile for the 36 tiles of layer4's 3x3 gradient
@pytest.mark.parametrize("R,Mo,No,sk", [(64, 256, 128, 1), (2048, 512, 2048, 4), (4800, 2048, 512, 5), (76800, 512, 2048, 16),
                                        (4800, 512, 512, 7), (66560, 512, 4608, 7), (2080, 256, 256, 65),
                                        (76800, 512, 64, 128), (4800, 256, 64, 5)])      # (the 256 x 64 tile: d fc_w = df^T u)
def test_bf16_storage_weight_gradient_product_against_float64(R, Mo, No, sk):
    """ait_gemm_bf16s_tn: dW[Mo, No] += dy^T x over the token rows, bf16 operands row-major with the reduction index
    outermost (the transposing LDS read ds_read_b64_tr_b16 builds the MFMA operands); split-K partials added with f32
    atomics: against the float64 product of the same bf16 values, and accumulation into a non-zero destination."""
    from ait_amd import ops
    torch.manual_seed(R + Mo)
    dy16 = torch.randn(R, Mo, device="cuda").to(torch.bfloat16)
    x16 = torch.randn(R, No, device="cuda").to(torch.bfloat16)
    ref = dy16.double().t() @ x16.double()
    mag = dy16.double().abs().t() @ x16.double().abs()
    got = ops.gemm_bf16s_tn(dy16, x16, split_k=sk)                          # K-ranges stored as partial tiles, then reduced
    assert float(((got.double() - ref).abs() / mag).max()) < 4e-6, float(((got.double() - ref).abs() / mag).max())
    assert torch.equal(got, ops.gemm_bf16s_tn(dy16, x16, split_k=sk))       # ... in range order: bit-reproducible
    got_a = ops.gemm_bf16s_tn(dy16, x16, split_k=sk, partials=False)        # no scratch: f32 atomics
    assert float(((got_a.double() - ref).abs() / mag).max()) < 4e-6
    base = torch.randn(Mo, No, device="cuda")
    got2 = ops.gemm_bf16s_tn(dy16, x16, out=base.clone(), split_k=sk)
    assert float(((got2.double() - ref - base.double()).abs() / (mag + 1)).max()) < 4e-6


@pytest.mark.parametrize("n,hw,k,cin,cout", [(8, 4, 3, 512, 512),       # layer4's 3x3 on 4x4 maps (256 x 128 x 32 tile: one row tile)
                                              (520, 4, 3, 512, 512),     # ... several row tiles per workgroup
                                              (3100, 4, 3, 512, 512),    # ... enough of them for the 256 x 256 x 64 tile
                                              (4112, 4, 3, 512, 512),    # ... 257 row tiles: the third round's two tiles cut along K, 8 pieces each
                                              (37, 8, 3, 128, 256),      # ragged row count, 8x8 maps
                                              (16, 4, 1, 256, 128),      # a 1x1 window is the plain product
                                              (6, 8, 5, 64, 128)])       # 5x5 window, 64-channel taps
def test_bf16_storage_convolution_against_float64(n, hw, k, cin, cout):
    """ait_conv_fwd_bf16s / ait_conv_bwd_weight_bf16s / ait_conv_weight_to_bf16: forward (+ bias, bf16 residual, ReLU), data
    gradient (the forward on the mirrored-transposed weight copy, + residual, gated by a stored ReLU output) and weight
    gradient of a stride-1 "same" convolution over bf16 channels-last maps, against a float64 convolution of the SAME bf16
    values written as an explicit window gather + matrix product, its gradients taken by autograd (products of bf16 values are
    exact in f32: what is left is the f32 summation order)."""
    from ait_amd import ops
    torch.manual_seed(n * 31 + hw + k + cin)
    dev = "cuda"
    rows, pad = n * hw * hw, k // 2
    x16 = torch.randn(rows, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, k, k, cin, device=dev) * (1.0 / (k * k * cin) ** 0.5)
    scale = torch.rand(cout, device=dev) + 0.5
    bias = torch.randn(cout, device=dev)
    res16 = torch.randn(rows, cout, device=dev).to(torch.bfloat16)
    w16 = ops.conv_weight_to_bf16(w, scale)
    w16d = ops.conv_weight_to_bf16(w, scale, dgrad=True)
    assert torch.equal(w16.view(cout, k, k, cin), (w * scale[:, None, None, None]).to(torch.bfloat16))
    assert torch.equal(w16d.view(cin, k, k, cout), w16.view(cout, k, k, cin).flip(1, 2).permute(3, 1, 2, 0))
    geom = ops.conv_geom(n, (hw, hw), (hw, hw), (k, k), 1, pad)

    def conv64(x, wm):                    # x [rows, cin] float64, wm [cout, k*k*cin] float64 -> [rows, cout]
        xp = torch.nn.functional.pad(x.view(n, hw, hw, cin), (0, 0, pad, pad, pad, pad))
        cols = torch.cat([xp[:, dy:dy + hw, dx:dx + hw, :] for dy in range(k) for dx in range(k)], dim=3)
        return cols.reshape(rows, k * k * cin) @ wm.t()
    xd = x16.double().requires_grad_(True)
    wd = w16.double().requires_grad_(True)
    lin = conv64(xd, wd)
    want = torch.relu(lin + bias.double() + res16.double()).detach()
    mag = (conv64(xd.detach().abs(), wd.detach().abs()) + bias.double().abs() + res16.double().abs())
    got = ops.conv_fwd_bf16s(x16, w16, geom, cin, cout, bias=bias, res16=res16, relu=True, out_f32=True)
    err = (got.double() - want).abs() / mag
    assert float(err.max()) < 2e-6, float(err.max())
    got16 = ops.conv_fwd_bf16s(x16, w16, geom, cin, cout, bias=bias, res16=res16, relu=True)
    assert torch.equal(got16, got.to(torch.bfloat16))                        # the bf16 result is the f32 one, rounded

    dy16 = torch.randn(rows, cout, device=dev).to(torch.bfloat16)
    want_dx, want_dw = torch.autograd.grad(lin, (xd, wd), dy16.double())
    # data gradient: dx = conv_transpose(dy, w) + residual, kept where the stored activation is positive
    act16 = torch.randn(rows, cin, device=dev).to(torch.bfloat16)
    radd16 = torch.randn(rows, cin, device=dev).to(torch.bfloat16)
    if (cout & (cout - 1)) == 0 and cin % 128 == 0:
        got_dx = ops.conv_fwd_bf16s(dy16, w16d, geom, cout, cin, res16=radd16, gate16=act16, out_f32=True)
        ref = (want_dx + radd16.double()) * (act16.double() > 0)
        err = (got_dx.double() - ref).abs() / (want_dx.abs().mean() * 8 + radd16.double().abs())
        assert float(err.max()) < 2e-5, float(err.max())
        assert float((got_dx.double() - ref).norm() / ref.norm()) < 1e-6
    # weight gradient: dw[o][tap][c] = sum_rows dy[row, o] x[row + tap, c]
    if cout % 256 == 0 and cin >= 128 and rows % 32 == 0:
        for sk, partials in ((1, False), (2 if rows % 64 == 0 else 1, True), (4 if rows % 128 == 0 else 1, False)):
            got_dw = ops.conv_bwd_weight_bf16s(dy16, x16, geom, k, k, split_k=sk, partials=partials).view(cout, -1)
            assert float((got_dw.double() - want_dw).norm() / want_dw.norm()) < 2e-6, (sk, partials)
            assert float((got_dw.double() - want_dw).abs().max() / want_dw.abs().max()) < 2e-5, (sk, partials)
