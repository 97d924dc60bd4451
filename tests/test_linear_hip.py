"""dgn_linear_* (tall-skinny fp32 Linear on the streaming MFMA kernels) against torch on the GPU, anchored on fp64:
forward, input gradient, weight gradient and bias gradient; ragged row counts (not a multiple of the 16-row strips),
one row, widths that are / are not multiples of 4 and 16, batched towers, and the dispatcher's library fall-back."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(T, M, k, n, bias, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(T, M, k, device="cuda", generator=g)
    w = torch.randn(T, n, k, device="cuda", generator=g) / k ** 0.5
    b = torch.randn(T, n, device="cuda", generator=g) if bias else None
    gy = torch.randn(T, M, n, device="cuda", generator=g)
    return x, w, b, gy


def _ref64(x, w, b, gy):
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True) if b is not None else None
    y = torch.bmm(x64, w64.transpose(1, 2)) + (b64.unsqueeze(1) if b64 is not None else 0)
    y.backward(gy.double())
    return y.detach(), x64.grad, w64.grad, (b64.grad if b64 is not None else None)


def _close(a, r, tol=1e-5):
    scale = float(r.abs().max()) + 1e-30
    assert float((a.double() - r).abs().max()) <= tol * scale, float((a.double() - r).abs().max()) / scale


@pytest.mark.parametrize("T,M,k,n,bias", [(1, 1000, 70, 140, True), (5, 777, 84, 42, False), (1, 1, 70, 70, True), (1, 16, 2, 2, True),
                                          (3, 4097, 16, 160, True), (2, 333, 160, 16, False), (1, 50000, 42, 84, False),
                                          (1, 15, 64, 64, True), (4, 31, 6, 10, True)])
def test_linear_matches_fp64(T, M, k, n, bias):
    from dgn_amd import ops
    x, w, b, gy = _case(T, M, k, n, bias)
    assert ops.linear_supported(k, n)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if b is not None else None
    y = ops.linear(xr, wr, br)
    y.backward(gy)
    y64, gx64, gw64, gb64 = _ref64(x, w, b, gy)
    _close(y, y64)
    _close(xr.grad, gx64)
    _close(wr.grad, gw64)
    if b is not None:
        _close(br.grad, gb64)


@pytest.mark.parametrize("M,k,n", [(100, 70, 70), (33, 6, 10), (17, 42, 84)])
def test_operands_at_8_byte_alignment(M, k, n):
    """Row views whose first element is only 8-byte aligned (the strips are read with 16-byte buffer lanes from the strip's own base):
    forward, input gradient and weight gradient against fp64."""
    from dgn_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    xs = torch.randn(M + 1, k, device="cuda", generator=g)
    gs = torch.randn(M + 1, n, device="cuda", generator=g)
    x, gy = xs[1:], gs[1:]                              # storage offsets of k / n floats: k = 70, 6, 42 -> 8 bytes past a 16-byte boundary
    assert x.data_ptr() % 16 == 8 and x.is_contiguous()
    w = torch.randn(n, k, device="cuda", generator=g) / k ** 0.5
    xr, wr = x.detach().requires_grad_(True), w.clone().requires_grad_(True)
    y = ops.linear(xr, wr, None)
    y.backward(gy)
    y64, gx64, gw64, _ = _ref64(x.unsqueeze(0), w.unsqueeze(0), None, gy.unsqueeze(0))
    _close(y, y64[0])
    _close(xr.grad, gx64[0])
    _close(wr.grad, gw64[0])


def test_linear_2d_equals_torch_and_is_reproducible():
    import torch.nn.functional as F
    from dgn_amd import ops
    x, w, b, gy = _case(1, 12345, 70, 140, True)
    x, w, b, gy = x[0], w[0], b[0], gy[0]
    outs = []
    for _ in range(2):
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = ops.linear(xr, wr, br)
        y.backward(gy)
        outs.append((y.detach(), xr.grad, wr.grad, br.grad))
    for a, c in zip(*outs):
        assert torch.equal(a, c)                       # no atomics anywhere: bitwise reproducible
    yt = F.linear(x, w, b)
    assert torch.allclose(outs[0][0], yt, rtol=1e-5, atol=1e-5)


def test_non_finite_rows_stay_in_their_rows():
    """The strip's padding columns alias the next row in LDS: an inf / nan there must not leak into the row before."""
    from dgn_amd import ops
    x, w, b, _ = _case(1, 64, 70, 70, True)
    x[0, 5, 0] = float("inf")
    x[0, 21, 3] = float("nan")
    y = ops.linear(x, w, b)[0]
    bad = ~torch.isfinite(y).all(dim=1)
    assert bad.nonzero().flatten().tolist() == [5, 21]


def test_dispatcher_falls_back_to_the_library():
    from dgn_amd import ops
    assert not ops.linear_supported(228, 75) and not ops.linear_supported(70, 71)
    x, w = torch.randn(100, 228, device="cuda"), torch.randn(75, 228, device="cuda")
    assert torch.allclose(ops.node_linear(x, w), torch.nn.functional.linear(x, w), rtol=1e-5, atol=1e-5)
    with pytest.raises(Exception):
        ops.linear(x, w)                                # the kernels refuse unsupported widths loudly


@pytest.mark.parametrize("T,N,k,S,fo,with_scale_rows", [(5, 3001, 84, 3, 14, True), (2, 100, 20, 1, 6, False), (3, 17, 36, 2, 8, True)])
def test_linear_combine_bn_tail_equals_the_unfused_nodes(T, N, k, S, fo, with_scale_rows):
    """posttrans + scale-combine + BatchNorm tail as one node (the product never written) == bmm -> combine_bn_tail."""
    from dgn_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    aggx = torch.randn(T, N, k, device="cuda", generator=g)
    w = torch.randn(T, S * fo, k, device="cuda", generator=g) / k ** 0.5
    sc = (torch.rand(N, S, device="cuda", generator=g) + 0.5) if S > 1 else None
    bias = torch.randn(T * fo, device="cuda", generator=g)
    rs = (torch.rand(N, device="cuda", generator=g) + 0.5) if with_scale_rows else None
    gamma, beta = torch.rand(T * fo, device="cuda", generator=g) + 0.5, torch.randn(T * fo, device="cuda", generator=g)
    ct = torch.randn(N, T * fo, device="cuda", generator=g)
    res = []
    for fused in (True, False):
        a, ww, b = aggx.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        ga, be = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        rm, rv = torch.zeros(T * fo, device="cuda"), torch.ones(T * fo, device="cuda")
        if fused:
            y = ops.linear_combine_bn_tail(a, ww, sc, b, rs, ga, be, rm, rv, None, 0.1, 1e-5)
        else:
            y = ops.combine_bn_tail(torch.bmm(a, ww.transpose(1, 2)), sc, b, rs, ga, be, rm, rv, None, 0.1, 1e-5)
        y.backward(ct)
        res.append((y.detach(), a.grad, ww.grad, b.grad, ga.grad, be.grad, rm, rv))
    for name, x, r in zip("y g_aggx g_w g_bias g_gamma g_beta running_mean running_var".split(), *res):
        # (the bias feeds a BatchNorm: its exact gradient is zero, both paths return rounding noise of the column sums)
        scale = float(ct.abs().sum(0).max()) if name == "g_bias" else float(r.abs().max()) + 1e-30
        assert float((x - r).abs().max()) <= 2e-5 * scale, (name, float((x - r).abs().max()) / scale)


@pytest.mark.parametrize("tile", ["0", "1"])        # 1: the 128 x 128 tile kernel (both operands through LDS) for the nn.Linear-layout products
@pytest.mark.parametrize("M,k,n", [(5000, 152, 225), (4097, 350, 210), (129, 198, 65), (1, 7, 5), (3, 4, 4), (300, 420, 300), (2500, 75, 75), (640, 16, 16)])
def test_wide_gemm_vs_fp64(monkeypatch, M, k, n, tile):
    """dgn_gemm_* (the simple / complex layers' posttrans shapes: odd widths, k and n beyond the streaming kernels): forward with
    bias, input gradient (both weight layouts), weight gradient against an fp64 evaluation; strided input rows; bitwise repeatability."""
    import ctypes as C
    from dgn_amd import _lib
    lib = _lib.load()
    monkeypatch.setattr(_lib.options, "tile_gemm", int(tile))
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(M + k + n)
    xbig = torch.randn(M, k + 3, device=dev, generator=gen)
    x = xbig[:, 1:k + 1]                                        # row stride k + 3, 4-byte aligned rows
    w = torch.randn(n, k, device=dev, generator=gen) / k ** 0.5
    b = torch.randn(n, device=dev, generator=gen)
    g = torch.randn(M, n, device=dev, generator=gen)
    st = torch.cuda.current_stream().cuda_stream
    c = torch.full((M, n), float("nan"), device=dev)
    _lib.check(lib.dgn_gemm_forward(M, k, n, x.data_ptr(), x.stride(0), w.data_ptr(), k, 0, b.data_ptr(), c.data_ptr(), n, st), "fwd")
    ref = x.double() @ w.double().T + b.double()
    tol = lambda r: 2e-6 * float(r.abs().max()) + 1e-6
    assert float((c.double() - ref).abs().max()) <= tol(ref)
    gx1 = torch.full((M, k), float("nan"), device=dev)
    gx2 = torch.full((M, k), float("nan"), device=dev)
    wt = w.t().contiguous()
    _lib.check(lib.dgn_gemm_forward(M, n, k, g.data_ptr(), n, w.data_ptr(), k, 1, None, gx1.data_ptr(), k, st), "dgrad kn")
    _lib.check(lib.dgn_gemm_forward(M, n, k, g.data_ptr(), n, wt.data_ptr(), n, 0, None, gx2.data_ptr(), k, st), "dgrad t")
    rgx = g.double() @ w.double()
    assert float((gx1.double() - rgx).abs().max()) <= tol(rgx) and float((gx2.double() - rgx).abs().max()) <= tol(rgx)
    # weight gradient: both kernels (DGN_TILE_WGRAD=1: 32 x 32 x 2 MFMA tile kernel, with and without the bias-gradient ones column;
    # 0: the 16-row strip kernel, n <= 256), against fp64; fixed summation order
    rgw, rgb = g.double().T @ x.double(), g.double().sum(0)
    wtol = 2e-6 * float(rgw.abs().max()) * max(1.0, (M / 4096) ** 0.5) + 1e-6
    nb = lib.dgn_gemm_wgrad_workspace_bytes(M, k, n)
    assert nb > 0
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    for tw, with_bias in (("1", True), ("1", False), ("0", False), ("0", True)):
        if tw == "0" and n > 256:
            continue
        monkeypatch.setattr(_lib.options, "tile_wgrad", int(tw))
        outs = []
        for _ in range(2):
            gw = torch.full((n, k + 2), float("nan"), device=dev)                 # (row stride k + 2: the padding columns must stay untouched)
            gb = torch.full((n,), float("nan"), device=dev) if with_bias else None
            _lib.check(lib.dgn_gemm_wgrad(M, k, n, g.data_ptr(), n, x.data_ptr(), x.stride(0), gw.data_ptr(), k + 2, gb.data_ptr() if with_bias else None,
                                          ws.data_ptr(), nb, st), "wgrad")
            outs.append((gw, gb))
        assert float((outs[0][0][:, :k].double() - rgw).abs().max()) <= wtol, (tw, with_bias)
        assert bool(torch.isnan(outs[0][0][:, k:]).all())
        assert torch.equal(outs[0][0][:, :k], outs[1][0][:, :k])
        if with_bias:
            assert float((outs[0][1].double() - rgb).abs().max()) <= 2e-6 * float(g.abs().sum(0).max()) + 1e-6
            assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("M,k,n", [(4100, 96, 330), (9000, 880, 1320), (33, 64, 32), (70000, 421, 211)])
def test_tile_wgrad_wide_and_aligned_shapes(M, k, n):
    """widths beyond one 256-column block either way (PCBA's complex layers: hidden 440 x 3 scalers, README.md:100-103), widths that are
    multiples of 32 (the ones column then opens a tile of its own), a row count that leaves ragged strips"""
    from dgn_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(M + k)
    x, g = torch.randn(M, k, device=dev, generator=gen), torch.randn(M, n, device=dev, generator=gen)
    st = torch.cuda.current_stream().cuda_stream
    nb = lib.dgn_gemm_wgrad_workspace_bytes(M, k, n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    gw, gb = torch.empty(n, k, device=dev), torch.empty(n, device=dev)
    _lib.check(lib.dgn_gemm_wgrad(M, k, n, g.data_ptr(), n, x.data_ptr(), k, gw.data_ptr(), k, gb.data_ptr(), ws.data_ptr(), nb, st), "wgrad")
    rgw, rgb = g.double().T @ x.double(), g.double().sum(0)
    assert float((gw.double() - rgw).abs().max()) <= 2e-6 * float(rgw.abs().max()) * max(1.0, (M / 4096) ** 0.5) + 1e-6
    assert float((gb.double() - rgb).abs().max()) <= 2e-6 * float(g.abs().sum(0).max()) + 1e-6


@pytest.mark.parametrize("M,k,n,tile", [(4096, 256, 192, "1"), (8192, 512, 192, "1"), (4096, 256, 192, "0"), (5000, 768, 320, None)])
@pytest.mark.parametrize("with_bias", [False, True])
def test_gemm_wgrad_never_writes_past_its_workspace(M, k, n, tile, with_bias, monkeypatch):
    """the workspace query planned with k + 1 while a bias-less call plans with k, and the plans are not monotone in
    k (k = 256, n = 192: 50 MB vs 31 MB).  A poisoned guard region behind the queried workspace must survive both kinds of call."""
    from dgn_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda")
    if tile is not None:
        monkeypatch.setattr(_lib.options, "tile_wgrad", int(tile))
    gen = torch.Generator(device=dev).manual_seed(k + n)
    x, g = torch.randn(M, k, device=dev, generator=gen), torch.randn(M, n, device=dev, generator=gen)
    st = torch.cuda.current_stream().cuda_stream
    nb = lib.dgn_gemm_wgrad_workspace_bytes(M, k, n)
    guard = 64 << 20
    buf = torch.full((nb + guard,), 0xA5, dtype=torch.uint8, device=dev)
    gw = torch.empty(n, k, device=dev)
    gb = torch.empty(n, device=dev) if with_bias else None
    _lib.check(lib.dgn_gemm_wgrad(M, k, n, g.data_ptr(), n, x.data_ptr(), k, gw.data_ptr(), k, gb.data_ptr() if with_bias else None,
                                  buf.data_ptr(), nb, st), "wgrad")
    torch.cuda.synchronize()
    assert bool((buf[nb:] == 0xA5).all()), "dgn_gemm_wgrad wrote behind the workspace it asked for"
    rgw = g.double().T @ x.double()
    assert float((gw.double() - rgw).abs().max()) <= 2e-6 * float(rgw.abs().max()) * max(1.0, (M / 4096) ** 0.5) + 1e-6


def test_wide_linear_autograd_matches_library(monkeypatch):
    from dgn_amd import ops
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(0)
    M, k, n = 6000, 152, 225
    x0, w0, b0 = torch.randn(M, k, device=dev, generator=gen), torch.randn(n, k, device=dev, generator=gen) / 12, torch.randn(n, device=dev, generator=gen)
    ct = torch.randn(M, n, device=dev, generator=gen)
    res = []
    for f in (ops.node_linear, torch.nn.functional.linear):
        x, w, b = (t.clone().requires_grad_(True) for t in (x0, w0, b0))
        assert f is not ops.node_linear or ops.wide_linear_supported(x, w)
        y = f(x, w, b)
        res.append((y, *torch.autograd.grad(y, [x, w, b], ct)))
    for a, r in zip(*res):
        torch.testing.assert_close(a, r, rtol=2e-5, atol=2e-5 * float(r.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4099, 70, 70), (275, 70, 70), (16, 84, 42), (1000, 140, 70), (5, 6, 2)])
@pytest.mark.parametrize("affine", [True, False])
def test_linear_with_batchnorm_prologue_is_bitwise_bn_apply_then_linear(shape, affine):
    """dgn_linear_forward_bn / dgn_linear_wgrad_bn (the operand normalised while its strips are staged) against dgn_bn_tail_forward's
    apply pass followed by dgn_linear_forward / dgn_linear_wgrad: bit-identical products and bias gradient."""
    import ctypes as C
    from dgn_amd import _lib
    lib = _lib.load()
    M, k, n = shape
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(M + k)
    x = torch.randn(M, k, device=dev, generator=gen) * 2 + 0.5
    w = torch.randn(n, k, device=dev, generator=gen)
    b = torch.randn(n, device=dev, generator=gen)
    g = torch.randn(M, n, device=dev, generator=gen)
    gamma = (torch.rand(k, device=dev, generator=gen) + 0.5) if affine else None
    beta = torch.randn(k, device=dev, generator=gen) if affine else None
    rm, rv = torch.zeros(k, device=dev), torch.ones(k, device=dev)
    mean, invstd = torch.empty(k, device=dev), torch.empty(k, device=dev)
    y1 = torch.empty_like(x)
    nb = lib.dgn_bn_tail_workspace_bytes(M, k)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: None if t is None else t.data_ptr()
    _lib.check(lib.dgn_bn_tail_forward(M, k, x.data_ptr(), k, P(gamma), P(beta), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, 1, 0, None, y1.data_ptr(),
                                       mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), nb, None, st), "bn")
    # statistics-only call gives the same statistics
    mean2, invstd2 = torch.empty_like(mean), torch.empty_like(invstd)
    _lib.check(lib.dgn_bn_tail_forward(M, k, x.data_ptr(), k, P(gamma), P(beta), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, 1, 0, None, None,
                                       mean2.data_ptr(), invstd2.data_ptr(), ws.data_ptr(), nb, None, st), "bn stats")
    assert torch.equal(mean, mean2) and torch.equal(invstd, invstd2)
    c_ref, c_fus = torch.empty(M, n, device=dev), torch.empty(M, n, device=dev)
    _lib.check(lib.dgn_linear_forward(M, k, n, 1, y1.data_ptr(), k, 0, w.data_ptr(), k, 0, 0, b.data_ptr(), 0, c_ref.data_ptr(), n, 0, st), "lin")
    _lib.check(lib.dgn_linear_forward_bn(M, k, n, x.data_ptr(), w.data_ptr(), k, 0, b.data_ptr(), c_fus.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                         P(gamma), P(beta), st), "lin bn")
    assert torch.equal(c_ref, c_fus)
    if lib.dgn_linear_supported(k, n, 1):
        nbw = lib.dgn_linear_wgrad_workspace_bytes(M, k, n, 1)
        wsw = torch.empty(max(nbw, 1), dtype=torch.uint8, device=dev)
        dw_ref, dw_fus = torch.empty(n, k, device=dev), torch.empty(n, k, device=dev)
        db_ref, db_fus = (torch.empty(n, device=dev), torch.empty(n, device=dev)) if k % 16 else (None, None)
        _lib.check(lib.dgn_linear_wgrad(M, k, n, 1, g.data_ptr(), n, 0, y1.data_ptr(), k, 0, dw_ref.data_ptr(), k, 0, P(db_ref), 0, wsw.data_ptr(), nbw, st), "wg")
        _lib.check(lib.dgn_linear_wgrad_bn(M, k, n, g.data_ptr(), x.data_ptr(), dw_fus.data_ptr(), k, P(db_fus), mean.data_ptr(), invstd.data_ptr(),
                                           P(gamma), P(beta), wsw.data_ptr(), nbw, st), "wg bn")
        assert torch.equal(dw_ref, dw_fus)
        if db_ref is not None:
            assert torch.equal(db_ref, db_fus)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4099, 70, 70), (275, 70, 70), (19, 84, 42), (1000, 140, 70), (3, 6, 2)])
@pytest.mark.parametrize("act", [2, 1, 0])
def test_linear_with_activation_gradient_prologue_is_bitwise(shape, act):
    """dgn_linear_forward_act (g * act'(z + b) formed while the strips are staged, side output included) against
    dgn_bias_act_backward followed by dgn_linear_forward: bit-identical operand and product."""
    from dgn_amd import _lib
    lib = _lib.load()
    M, k, n = shape
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(M + 7 * k + act)
    g, z = torch.randn(M, k, device=dev, generator=gen), torch.randn(M, k, device=dev, generator=gen)
    b = torch.randn(k, device=dev, generator=gen)
    w = torch.randn(k, n, device=dev, generator=gen)                                   # [k, n]: c = g_z . w  (w_is_kn = 1)
    st = torch.cuda.current_stream().cuda_stream
    gz_ref, gb = torch.empty_like(g), torch.empty(k, device=dev)
    nb = lib.dgn_bn_tail_workspace_bytes(M, k)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    _lib.check(lib.dgn_bias_act_backward(M, k, g.data_ptr(), z.data_ptr(), k, b.data_ptr(), act, 0.01, gz_ref.data_ptr(), gb.data_ptr(), ws.data_ptr(), nb, st), "bab")
    c_ref = torch.empty(M, n, device=dev)
    _lib.check(lib.dgn_linear_forward(M, k, n, 1, gz_ref.data_ptr(), k, 0, w.data_ptr(), n, 0, 1, None, 0, c_ref.data_ptr(), n, 0, st), "lin")
    c_fus, gz_fus = torch.empty(M, n, device=dev), torch.full((M, k), float("nan"), device=dev)
    _lib.check(lib.dgn_linear_forward_act(M, k, n, g.data_ptr(), z.data_ptr(), b.data_ptr(), act, 0.01, w.data_ptr(), n, 1, c_fus.data_ptr(), gz_fus.data_ptr(), st), "lin act")
    assert torch.equal(gz_ref, gz_fus)
    assert torch.equal(c_ref, c_fus)
    c_no = torch.empty(M, n, device=dev)                                                # without the side output
    _lib.check(lib.dgn_linear_forward_act(M, k, n, g.data_ptr(), z.data_ptr(), b.data_ptr(), act, 0.01, w.data_ptr(), n, 1, c_no.data_ptr(), None, st), "lin act")
    assert torch.equal(c_ref, c_no)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4099, 140, 70), (275, 140, 70), (19, 84, 42), (1000, 70, 70), (3, 6, 2)])
@pytest.mark.parametrize("two", [True, False])
def test_linear_with_add_epilogue_is_bitwise(shape, two):
    """dgn_linear_forward_add: (add1 + a w) + add2 in the product's epilogue against the product followed by the sums in that order."""
    from dgn_amd import _lib
    lib = _lib.load()
    M, k, n = shape
    if not lib.dgn_linear_add_supported(k, n):
        pytest.skip("shape outside the add-epilogue set")
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(M + k + int(two))
    a, w = torch.randn(M, k, device=dev, generator=gen), torch.randn(k, n, device=dev, generator=gen)
    e1, e2 = torch.randn(M, n, device=dev, generator=gen), torch.randn(M, n, device=dev, generator=gen)
    st = torch.cuda.current_stream().cuda_stream
    c = torch.empty(M, n, device=dev)
    _lib.check(lib.dgn_linear_forward(M, k, n, 1, a.data_ptr(), k, 0, w.data_ptr(), n, 0, 1, None, 0, c.data_ptr(), n, 0, st), "lin")
    want = (e1 + c) + e2 if two else e1 + c
    got = torch.full((M, n), float("nan"), device=dev)
    _lib.check(lib.dgn_linear_forward_add(M, k, n, a.data_ptr(), w.data_ptr(), n, 1, e1.data_ptr(), e2.data_ptr() if two else None, got.data_ptr(), st), "lin add")
    assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4099, 70, 70), (275, 70, 70), (19, 84, 42), (3, 6, 2)])
@pytest.mark.parametrize("residual", [True, False])
def test_bn_linear_act_residual_in_one_pass_is_bitwise(shape, residual):
    """dgn_linear_forward_bn_act against dgn_bn_tail_forward (apply) -> dgn_linear_forward -> dgn_bias_act_forward: both outputs bit-identical."""
    from dgn_amd import _lib
    lib = _lib.load()
    M, k, n = shape
    if not lib.dgn_linear_add_supported(k, n):
        pytest.skip("shape outside the supported set")
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(M + 3 * k)
    x = torch.randn(M, k, device=dev, generator=gen) + 0.3
    w, b = torch.randn(n, k, device=dev, generator=gen), torch.randn(n, device=dev, generator=gen)
    gamma, beta = torch.rand(k, device=dev, generator=gen) + 0.5, torch.randn(k, device=dev, generator=gen)
    res = torch.randn(M, n, device=dev, generator=gen) if residual else None
    rm, rv = torch.zeros(k, device=dev), torch.ones(k, device=dev)
    mean, invstd, y1 = torch.empty(k, device=dev), torch.empty(k, device=dev), torch.empty_like(x)
    nb = lib.dgn_bn_tail_workspace_bytes(M, k)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: None if t is None else t.data_ptr()
    _lib.check(lib.dgn_bn_tail_forward(M, k, x.data_ptr(), k, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, 1, 0, None,
                                       y1.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), nb, None, st), "bn")
    z_ref, o_ref = torch.empty(M, n, device=dev), torch.empty(M, n, device=dev)
    _lib.check(lib.dgn_linear_forward(M, k, n, 1, y1.data_ptr(), k, 0, w.data_ptr(), k, 0, 0, None, 0, z_ref.data_ptr(), n, 0, st), "lin")
    _lib.check(lib.dgn_bias_act_forward(M, n, z_ref.data_ptr(), n, b.data_ptr(), 2, 0.01, P(res), o_ref.data_ptr(), st), "act")
    z_f, o_f = torch.full((M, n), float("nan"), device=dev), torch.full((M, n), float("nan"), device=dev)
    _lib.check(lib.dgn_linear_forward_bn_act(M, k, n, x.data_ptr(), w.data_ptr(), k, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                             b.data_ptr(), 2, 0.01, P(res), z_f.data_ptr(), o_f.data_ptr(), st), "fused")
    assert torch.equal(z_ref, z_f)
    assert torch.equal(o_ref, o_f)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4099, 70, 70), (275, 70, 70), (19, 84, 42), (16, 70, 70), (3, 6, 2), (1000, 112, 112)])
@pytest.mark.parametrize("act,residual", [(2, True), (1, True), (2, False)])
def test_activation_mask_round_trip_is_bitwise_the_z_path(shape, act, residual):
    """dgn_linear_forward_bn_act_mask writes the sign mask of (z + b) instead of z; dgn_linear_forward_act_mask reads the activation's
    derivative from it: `out`, the input-gradient product and its side output must be the bits of the z-based pair, and the mask itself
    must be exactly (z + b > 0), two bits per byte."""
    from dgn_amd import _lib
    lib = _lib.load()
    M, k, n = shape
    if not (lib.dgn_linear_add_supported(k, n) and lib.dgn_linear_act_supported(n, k)):
        pytest.skip("shape outside the supported set")
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(M + 3 * k + act)
    x = torch.randn(M, k, device=dev, generator=gen) + 0.3
    w, b = torch.randn(n, k, device=dev, generator=gen) / k ** 0.5, torch.randn(n, device=dev, generator=gen)
    gamma, beta = torch.rand(k, device=dev, generator=gen) + 0.5, torch.randn(k, device=dev, generator=gen)
    mean, var = x.mean(0), x.var(0, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    res = torch.randn(M, n, device=dev, generator=gen) if residual else None
    rp = res.data_ptr() if residual else None
    st = torch.cuda.current_stream().cuda_stream
    z, o = torch.empty(M, n, device=dev), torch.empty(M, n, device=dev)
    _lib.check(lib.dgn_linear_forward_bn_act(M, k, n, x.data_ptr(), w.data_ptr(), k, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                             b.data_ptr(), act, 0.01, rp, z.data_ptr(), o.data_ptr(), st), "bn_act")
    nb = lib.dgn_linear_act_mask_bytes(M, n)
    assert nb >= M * n // 2 and nb % 16 == 0
    mask = torch.full((nb,), 0xAA, dtype=torch.uint8, device=dev)
    o_m = torch.full((M, n), float("nan"), device=dev)
    _lib.check(lib.dgn_linear_forward_bn_act_mask(M, k, n, x.data_ptr(), w.data_ptr(), k, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                                  beta.data_ptr(), b.data_ptr(), act, 0.01, rp, mask.data_ptr(), o_m.data_ptr(), st), "bn_act_mask")
    assert torch.equal(o, o_m)
    pos = ((z + b) > 0).reshape(-1, 2)
    want = (pos[:, 0].to(torch.uint8) | (pos[:, 1].to(torch.uint8) << 1))
    assert torch.equal(mask[:M * n // 2], want)
    assert bool((mask[M * n // 2:] == 0xAA).all())                                    # nothing written past the tensor's bytes
    # backward: g * act'(z + b) and the product with the mixing weight ([n, k] read as the reduction-major operand)
    g = torch.randn(M, n, device=dev, generator=gen)
    wt = torch.randn(n, k, device=dev, generator=gen)
    c_ref, gz_ref = torch.empty(M, k, device=dev), torch.empty(M, n, device=dev)
    _lib.check(lib.dgn_linear_forward_act(M, n, k, g.data_ptr(), z.data_ptr(), b.data_ptr(), act, 0.01, wt.data_ptr(), k, 1, c_ref.data_ptr(), gz_ref.data_ptr(), st), "act")
    c_m, gz_m = torch.full((M, k), float("nan"), device=dev), torch.full((M, n), float("nan"), device=dev)
    _lib.check(lib.dgn_linear_forward_act_mask(M, n, k, g.data_ptr(), mask.data_ptr(), act, 0.01, wt.data_ptr(), k, 1, c_m.data_ptr(), gz_m.data_ptr(), st), "act_mask")
    assert torch.equal(gz_ref, gz_m)
    assert torch.equal(c_ref, c_m)


@pytest.mark.parametrize("M,n,fo,with_rs", [(4099, 70, 14, True), (16, 70, 14, False), (7, 70, 14, True), (1033, 64, 16, True), (515, 40, 10, False),
                                           (2050, 80, 16, True), (300, 20, 10, True)])
def test_mixing_backward_kernels_of_round_6_vs_the_separate_passes(M, n, fo, with_rs):
    """dgn_linear_wgrad_bn_act_mask (the G operand of the weight gradient formed from g and the activation's byte mask, the mask through a
    direct-to-LDS load) = dgn_linear_forward_act_mask's side output fed to dgn_linear_wgrad_bn, bit for bit; and
    dgn_linear_forward_act_mask_bnb (BatchNorm's backward + row scale in the input-gradient product's epilogue, tower-major) = the product
    followed by the elementwise formula in combine_bwd's order, bit for bit.  Row counts with partial last strips and single strips."""
    from dgn_amd import _lib
    lib = _lib.load()
    k = n
    if not lib.dgn_linear_bnb_supported(k, n):
        pytest.skip("shape outside the supported set")
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(M + 7 * n)
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: None if t is None else t.data_ptr()
    g = torch.randn(M, n, device=dev, generator=gen)
    z = torch.randn(M, n, device=dev, generator=gen)
    y0 = torch.randn(M, k, device=dev, generator=gen) * 1.5 + 0.2
    w = torch.randn(n, k, device=dev, generator=gen) / k ** 0.5            # mixing weight [out, in]
    gamma = torch.rand(k, device=dev, generator=gen) + 0.5
    mean, invstd = y0.mean(0), (y0.var(0, unbiased=False) + 1e-5).rsqrt()
    rs = (torch.rand(M, device=dev, generator=gen) + 0.5) if with_rs else None
    nb = lib.dgn_linear_act_mask_bytes(M, n)
    pos = (z > 0).reshape(-1, 2)
    mask = torch.zeros(nb, dtype=torch.uint8, device=dev)
    mask[:M * n // 2] = pos[:, 0].to(torch.uint8) | (pos[:, 1].to(torch.uint8) << 1)
    slope = 0.01
    # the separate passes: g_z as a tensor, the input-gradient product, the weight gradient on the normalised operand (no affine part)
    g_y1, g_z = torch.empty(M, k, device=dev), torch.empty(M, n, device=dev)
    _lib.check(lib.dgn_linear_forward_act_mask(M, n, k, g.data_ptr(), mask.data_ptr(), 2, slope, w.data_ptr(), k, 1, g_y1.data_ptr(), g_z.data_ptr(), st), "act_mask")
    nbw = lib.dgn_linear_wgrad_workspace_bytes(M, k, n, 1)
    ws = torch.empty(max(nbw, 1), dtype=torch.uint8, device=dev)
    with_db = k % 16 != 0                                                   # (the bias gradient rides in the operand's padding column)
    dw_ref, db_ref = torch.empty(n, k, device=dev), torch.zeros(n, device=dev)
    _lib.check(lib.dgn_linear_wgrad_bn(M, k, n, g_z.data_ptr(), y0.data_ptr(), dw_ref.data_ptr(), k, db_ref.data_ptr() if with_db else None, mean.data_ptr(),
                                       invstd.data_ptr(), None, None, ws.data_ptr(), nbw, st), "wgrad_bn")
    dw, db = torch.full((n, k), float("nan"), device=dev), torch.zeros(n, device=dev)
    for _ in range(2):      # (twice: bitwise reproducible)
        _lib.check(lib.dgn_linear_wgrad_bn_act_mask(M, k, n, g.data_ptr(), mask.data_ptr(), 2, slope, y0.data_ptr(), dw.data_ptr(), k, db.data_ptr() if with_db else None,
                                                    mean.data_ptr(), invstd.data_ptr(), None, None, ws.data_ptr(), nbw, st), "wgrad_bn_act_mask")
        assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
    # BatchNorm's backward on the product's rows, combine_bwd's order: ga * is * (g - s0 / M - xh * s1 / M), then the row scale
    xh = (y0 - mean) * invstd
    sums = torch.cat([g_y1.sum(0), (g_y1 * xh).sum(0)]).contiguous()
    inv_n = torch.tensor(1.0, device=dev) / torch.tensor(float(M), device=dev)
    c4, c5 = sums[:k] * inv_n, sums[k:] * inv_n
    want = (gamma * invstd) * ((g_y1 - c4) - xh * c5)
    if rs is not None:
        want = want * rs.unsqueeze(1)
    T = k // fo
    gz = torch.full((T, M, fo), float("nan"), device=dev)
    _lib.check(lib.dgn_linear_forward_act_mask_bnb(M, n, k, g.data_ptr(), mask.data_ptr(), 2, slope, w.data_ptr(), k, 1, y0.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                   gamma.data_ptr(), sums.data_ptr(), P(rs), fo, gz.data_ptr(), M * fo, st), "act_mask_bnb")
    got = gz.permute(1, 0, 2).reshape(M, k)
    assert torch.isfinite(got).all()
    assert torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.parametrize("T,fi,M", [(5, 14, 4099), (5, 14, 16), (5, 14, 1), (5, 10, 700), (5, 20, 1033), (5, 30, 515), (4, 14, 2000), (2, 14, 333)])
def test_block_diagonal_pair_linear_vs_fp64(T, fi, M):
    """dgn_linear_bd_*: the towers' block-diagonal P|Q product, its input gradient (with the two-operand add epilogue) and its weight /
    bias gradient against an fp64 evaluation of the dense formula; off-diagonal weight-gradient blocks are exactly zero; the dense
    streaming kernels agree; run-to-run bitwise equality."""
    import ctypes as C
    import dgn_amd
    from dgn_amd import _lib, ops
    dev = torch.device("cuda")
    lib = _lib.load()
    assert lib.dgn_linear_bd_supported(T, fi)
    Fm = T * fi
    gen = torch.Generator(device=dev).manual_seed(T * 100 + fi)
    x = torch.randn(M, Fm, device=dev, generator=gen)
    blocks = torch.randn(2, T, fi, fi, device=dev, generator=gen)
    w = torch.zeros(2, T, fi, T, fi, device=dev)
    idx = torch.arange(T, device=dev)
    w[:, idx, :, idx, :] = blocks.transpose(0, 1)
    w = w.view(2 * Fm, Fm)
    bias = torch.randn(2 * Fm, device=dev, generator=gen)
    g = torch.randn(M, 2 * Fm, device=dev, generator=gen)
    a1, a2 = torch.randn(M, Fm, device=dev, generator=gen), torch.randn(M, Fm, device=dev, generator=gen)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    monkey_rows = ops.LINEAR_MIN_ROWS
    ops.LINEAR_MIN_ROWS = 0
    try:
        y = ops.pair_linear(xr, wr, br, T, fi)
        gx, gw, gb = torch.autograd.grad(y, [xr, wr, br], g)
        y2 = ops.pair_linear(xr, wr, br, T, fi)
        gx2, gw2, gb2 = torch.autograd.grad(y2, [xr, wr, br], g)
    finally:
        ops.LINEAR_MIN_ROWS = monkey_rows
    assert torch.equal(y, y2) and torch.equal(gx, gx2) and torch.equal(gw, gw2) and torch.equal(gb, gb2)
    x64, w64, b64, g64 = x.double(), w.double(), bias.double(), g.double()
    y_ref = x64 @ w64.t() + b64
    tol = lambda ref: 2e-6 * float(ref.abs().max()) + 1e-30
    assert float((y.double() - y_ref).abs().max()) <= tol(y_ref)
    gx_ref = g64 @ w64
    assert float((gx.double() - gx_ref).abs().max()) <= tol(gx_ref)
    mask = (w != 0).double()                       # (random blocks: no exact zeros inside a block)
    gw_ref = (g64.t() @ x64) * mask
    assert float((gw.double() - gw_ref).abs().max()) <= 1e-5 * float(gw_ref.abs().max()) + 1e-30
    assert float((gw * (1 - mask.float())).abs().max()) == 0.0
    gb_ref = g64.sum(0)
    assert float((gb.double() - gb_ref).abs().max()) <= 1e-5 * max(1.0, float(gb_ref.abs().max()))
    # the add epilogue: (add1 + product) + add2, through the C entry point
    out = torch.empty(M, Fm, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for add2 in (a2, None):
        _lib.check(lib.dgn_linear_bd_backward_input(M, T, fi, g.data_ptr(), w.data_ptr(), Fm, a1.data_ptr(), add2.data_ptr() if add2 is not None else None,
                                                    out.data_ptr(), stream), "dgn_linear_bd_backward_input")
        ref = (a1 + gx) + (add2 if add2 is not None else 0)
        assert torch.equal(out, ref) if add2 is not None else torch.equal(out, a1 + gx)
    # inf / nan stay in their rows and in their towers' columns
    if M > 3:
        xb = x.clone()
        xb[2, 3] = float("inf")
        yb = torch.empty(M, 2 * Fm, device=dev)
        _lib.check(lib.dgn_linear_bd_forward(M, T, fi, xb.data_ptr(), w.data_ptr(), Fm, bias.data_ptr(), yb.data_ptr(), stream), "dgn_linear_bd_forward")
        bad = ~torch.isfinite(yb)
        assert bool(bad[2].any()) and not bool(bad[torch.arange(M, device=dev) != 2].any())
        cols = torch.nonzero(bad[2]).flatten()
        assert bool(((cols % Fm) < fi).all())                                           # tower 0's outputs only (P and Q halves)
