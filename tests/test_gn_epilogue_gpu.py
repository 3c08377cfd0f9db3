"""GroupNorm statistics out of the producing contraction's epilogue (round 6; include/syn3r_hip.h
syn3r_gemm_set_gn_partials / syn3r_groupnorm_pre_f16, csrc/gemm_wide.h gn_tile_stats, csrc/norm.hip k_gn_finalize_parts).

Reference semantics: GroupNorm(32) of resnet.py:272,286,574,588 and transformer_temporal.py:235 on the output of the
contraction in front of it.  Checked here: the partial sums the kernels leave behind equal the sums over the fp16 output
as stored (per 32-row block and 10-column unit), on every kernel family with the lean epilogue and every epilogue form
(plain, row vector, residual, residual + aux); GroupNorm from the partial sums equals torch's fp32 group_norm of that
output within the bar of the statistics-pass form; two-source groups that straddle the sources; bitwise repeatability;
shapes / kernels that cannot serve the request fall back silently to the statistics pass."""
import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu
H = torch.float16


def rnd(gen, *shape, scale=1.0, dev=None):
    return (torch.randn(*shape, generator=gen) * scale).to(H).to(dev)


def close(a, b, tol=3e-3):
    a, b = a.float(), b.float()
    err = (a - b).abs().max().item()
    assert err <= tol * (b.abs().max().item() + 1e-6), f"max err {err} vs scale {b.abs().max().item()}"


def expected_partials(out2d: torch.Tensor) -> torch.Tensor:
    """[M, N] fp16 -> [M/32, 2, N/10] fp64 sums of x and x^2 per 32-row block and 10-column unit."""
    M, N = out2d.shape
    x = out2d.double().view(M // 32, 32, N // 10, 10)
    return torch.stack([x.sum((1, 3)), (x * x).sum((1, 3))], dim=1)


def check_partials(out: torch.Tensor, N: int):
    part = getattr(out, "gn_part", None)
    assert part is not None, "the kernel did not write the partial sums"
    out2d = out.reshape(-1, N)
    exp = expected_partials(out2d)
    got = part.view(out2d.shape[0] // 32, 2, N // 10).double()
    scale = exp[:, 1].abs().max().item() + 1e-6
    assert (got[:, 0] - exp[:, 0]).abs().max().item() <= 2e-5 * (exp[:, 0].abs().max().item() + 32 * 10)
    assert (got[:, 1] - exp[:, 1]).abs().max().item() <= 2e-5 * scale


def gn_ref(x2d: torch.Tensor, samples: int, ga, be, eps, silu):
    M, C = x2d.shape
    xr = x2d.float().reshape(samples, M // samples, C).permute(0, 2, 1)
    ref = Fn.group_norm(xr, 32, ga.float(), be.float(), eps)
    if silu:
        ref = Fn.silu(ref)
    return ref.permute(0, 2, 1).reshape(M, C)


@pytest.mark.parametrize("tile", [0, -320, -322, -256])
@pytest.mark.parametrize("epi", ["plain", "rowvec", "residual", "blend"])
def test_dense_partials_every_lean_kernel(tile, epi, gpu):
    """syn3r_gemm_f16 on the persistent kernels (k_gemm_widep / k_gemm_z / k_gemm_dmap, forced and by shape)."""
    from syn3r_amd import _lib
    from syn3r_amd.unet import ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    # by shape (tile 0) the 256-row persistent kernel needs >= 256 tiles of 256 x 160; forced kernels take a smaller matrix
    M, N, K = (16384 + 512 if tile == 0 else 4096 + 512), 640, 192
    x, w, b = rnd(g, M, K, dev=gpu), rnd(g, N, K, scale=K ** -0.5, dev=gpu), rnd(g, N, dev=gpu)
    kw = {}
    if epi == "rowvec":
        kw = dict(rowvec=rnd(g, 4, N, dev=gpu), rows_per_vec=M // 4)
    elif epi == "residual":
        kw = dict(residual=rnd(g, M, N, dev=gpu))
    elif epi == "blend":
        kw = dict(residual=rnd(g, M, N, dev=gpu), aux=rnd(g, M, N, dev=gpu), s_acc=0.4, s_res=0.4, s_aux=0.6)
    try:
        _lib.check(lib.syn3r_gemm_set_tile(tile), "set_tile")
        plain = ops.linear(x, w, b, **kw)
        out = ops.linear(x, w, b, gn_stats=True, **kw)
    finally:
        lib.syn3r_gemm_set_tile(0)
    assert torch.equal(out, plain), "asking for the partial sums changed the output"
    check_partials(out, N)


@pytest.mark.parametrize("NB,Hi,Wi,Cin,Cout,stride,ups", [(4, 32, 32, 64, 320, 1, False), (2, 32, 64, 128, 640, 1, False),
                                                          (4, 32, 32, 64, 320, 2, False), (2, 16, 32, 64, 320, 1, True)])
def test_conv3x3_partials(NB, Hi, Wi, Cin, Cout, stride, ups, gpu):
    """The implicit-GEMM convolutions (k_gemm_z<conv2d>, also the stride-2 and fused-upsample forms) with a row vector and with
    a residual."""
    from syn3r_amd import _lib
    from syn3r_amd.unet import ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(Cin + Cout + stride)
    x = rnd(g, NB, Hi, Wi, Cin, dev=gpu)
    w, b = rnd(g, Cout, 3, 3, Cin, scale=(9 * Cin) ** -0.5, dev=gpu), rnd(g, Cout, dev=gpu)
    try:
        _lib.check(lib.syn3r_gemm_set_tile(-322), "set_tile")
        y0 = ops.conv3x3(x, w, b, stride=stride, upsample=ups)
        M = y0.numel() // Cout
        rv = rnd(g, NB, Cout, dev=gpu)
        y1 = ops.conv3x3(x, w, b, stride=stride, upsample=ups, rowvec=rv, rows_per_vec=M // NB, gn_stats=True)
        res = rnd(g, *y0.shape, dev=gpu)
        y2 = ops.conv3x3(x, w, b, stride=stride, upsample=ups, residual=res, gn_stats=True)
        y2b = ops.conv3x3(x, w, b, stride=stride, upsample=ups, residual=res)
    finally:
        lib.syn3r_gemm_set_tile(0)
    check_partials(y1, Cout)
    check_partials(y2, Cout)
    assert torch.equal(y2, y2b)


@pytest.mark.parametrize("tile", [0, -322])
def test_tconv3_partials_and_temporal_groupnorm(tile, gpu):
    """k_gemm_dmap<tconv> (frame-minor tile order) and k_gemm_z<tconv>; then the 3D GroupNorm form (one sample = F*HW rows)
    and the 2D form (one sample = HW rows) from the same partial sums."""
    from syn3r_amd import _lib
    from syn3r_amd.unet import ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    B, F, HW, Cin, Cout = 2, 5, 1024, 64, 320
    x = rnd(g, B * F * HW, Cin, dev=gpu)
    w, b = rnd(g, Cout, 3, Cin, scale=(3 * Cin) ** -0.5, dev=gpu), rnd(g, Cout, dev=gpu)
    res = rnd(g, B * F * HW, Cout, dev=gpu)
    try:
        _lib.check(lib.syn3r_gemm_set_tile(tile), "set_tile")
        y = ops.tconv3(x, w, b, B, F, HW, residual=res, s_acc=0.3, s_res=1.0, gn_stats=True)
    finally:
        lib.syn3r_gemm_set_tile(0)
    check_partials(y, Cout)
    ga, be = rnd(g, Cout, dev=gpu), rnd(g, Cout, dev=gpu)
    for samples in (B, B * F):
        got = ops.groupnorm(y, ga, be, samples, 1e-5, True)
        close(got, gn_ref(y, samples, ga, be, 1e-5, True))
        again = ops.groupnorm(y, ga, be, samples, 1e-5, True)
        assert torch.equal(got, again)
        # against the statistics-pass form: the same normalisation up to the summation order of the statistics
        close(got, ops.groupnorm(y, ga, be, samples, 1e-5, True, use_partials=False), tol=2e-3)


@pytest.mark.parametrize("C1,C2", [(640, 320), (1280, 640), (320, 320)])
def test_groupnorm_two_source_from_partials(C1, C2, gpu):
    """Groups that straddle the two sources (640 + 320: 30 channels per group; 1280 + 640: 60), statistics folded from the two
    producers' partial sums."""
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(C1 + C2)
    M, K, samples = 2048, 64, 4
    a = rnd(g, M, K, dev=gpu)
    w1, w2 = rnd(g, C1, K, scale=0.2, dev=gpu), rnd(g, C2, K, scale=0.4, dev=gpu)
    b1, b2 = rnd(g, C1, dev=gpu), rnd(g, C2, dev=gpu)
    from syn3r_amd import _lib
    lib = _lib.load()
    try:
        _lib.check(lib.syn3r_gemm_set_tile(-256), "set_tile")      # (2 048 rows: by shape these go to the 128-row blocks)
        x1 = ops.linear(a, w1, b1, gn_stats=True)
        x2 = ops.linear(a, w2, b2, gn_stats=True)
    finally:
        lib.syn3r_gemm_set_tile(0)
    assert getattr(x1, "gn_part", None) is not None and getattr(x2, "gn_part", None) is not None
    ga, be = rnd(g, C1 + C2, dev=gpu), rnd(g, C1 + C2, dev=gpu)
    got = ops.groupnorm(x1, ga, be, samples, 1e-5, True, x2=x2)
    close(got, gn_ref(torch.cat([x1, x2], 1), samples, ga, be, 1e-5, True))
    close(got, ops.groupnorm(x1, ga, be, samples, 1e-5, True, x2=x2, use_partials=False), tol=2e-3)
    # one source with partial sums, the other without: the statistics pass
    x2n = x2.clone()
    got2 = ops.groupnorm(x1, ga, be, samples, 1e-5, True, x2=x2n)
    assert torch.equal(got2, ops.groupnorm(x1, ga, be, samples, 1e-5, True, x2=x2, use_partials=False))


def test_partials_bitwise_repeatable(gpu):
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(3)
    x, w = rnd(g, 8192, 256, dev=gpu), rnd(g, 320, 256, scale=1 / 16, dev=gpu)
    res = rnd(g, 8192, 320, dev=gpu)
    from syn3r_amd import _lib
    lib = _lib.load()
    try:
        _lib.check(lib.syn3r_gemm_set_tile(-320), "set_tile")
        a = ops.linear(x, w, residual=res, gn_stats=True)
        b = ops.linear(x, w, residual=res, gn_stats=True)
    finally:
        lib.syn3r_gemm_set_tile(0)
    assert torch.equal(a.gn_part, b.gn_part) and torch.equal(a, b)


def test_requests_that_cannot_be_served_fall_back(gpu):
    """Ragged rows / widths, kernels without the lean epilogue (128-row LDS-DMA blocks, the skinny kernel): no partial sums,
    nothing pending afterwards, GroupNorm runs its statistics pass."""
    from syn3r_amd import _lib
    from syn3r_amd.unet import ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    x, w = rnd(g, 1000, 64, dev=gpu), rnd(g, 320, 64, scale=0.125, dev=gpu)
    y = ops.linear(x, w, gn_stats=True)                                   # M % 32 != 0
    assert getattr(y, "gn_part", None) is None
    x = rnd(g, 1024, 64, dev=gpu)
    w2 = rnd(g, 328, 64, scale=0.125, dev=gpu)
    assert getattr(ops.linear(x, w2, gn_stats=True), "gn_part", None) is None     # N % 80 != 0
    try:
        _lib.check(lib.syn3r_gemm_set_tile(-128), "set_tile")
        y = ops.linear(x, w, gn_stats=True)
    finally:
        lib.syn3r_gemm_set_tile(0)
    assert getattr(y, "gn_part", None) is None
    assert lib.syn3r_gemm_gn_partials_written() == 0
    # the request did not leak into the next launch
    z = ops.linear(x, w)
    assert lib.syn3r_gemm_gn_partials_written() == 0 and getattr(z, "gn_part", None) is None
    # a request followed by an entry point that never writes partial sums (the gated projection): dropped there, not kept for the
    # next contraction of the thread
    part = torch.zeros(lib.syn3r_gn_partials_bytes(1024, 320) // 4, dtype=torch.float32, device=gpu)
    _lib.check(lib.syn3r_gemm_set_gn_partials(part.data_ptr(), part.numel() * 4), "set_gn_partials")
    wg, bg = rnd(g, 2 * 160, 64, scale=0.125, dev=gpu), rnd(g, 2 * 160, dev=gpu)
    wp, bp, _ = ops.pack_geglu(wg, bg)
    ops.linear_geglu(x, wp, bp, 160)
    try:
        _lib.check(lib.syn3r_gemm_set_tile(-256), "set_tile")
        ops.linear(x, w)
    finally:
        lib.syn3r_gemm_set_tile(0)
    torch.cuda.synchronize()
    assert lib.syn3r_gemm_gn_partials_written() == 0 and float(part.abs().sum()) == 0.0
    ga, be = rnd(g, 320, dev=gpu), rnd(g, 320, dev=gpu)
    close(ops.groupnorm(y, ga, be, 4, 1e-5, False), gn_ref(y, 4, ga, be, 1e-5, False))
    # rows per sample not a multiple of 32 with partial sums present: the statistics pass
    try:
        _lib.check(lib.syn3r_gemm_set_tile(-256), "set_tile")
        y = ops.linear(x, w, gn_stats=True)
    finally:
        lib.syn3r_gemm_set_tile(0)
    assert getattr(y, "gn_part", None) is not None
    close(ops.groupnorm(y, ga, be, 64, 1e-5, False), gn_ref(y, 64, ga, be, 1e-5, False))


def test_groupnorm_pre_rejects_bad_input(gpu):
    from syn3r_amd import _lib
    lib = _lib.load()
    x = torch.zeros(64, 320, dtype=H, device=gpu)
    part = torch.zeros(2 * 2 * 32, dtype=torch.float32, device=gpu)
    ws = torch.zeros(1 << 16, dtype=torch.uint8, device=gpu)
    ga = torch.ones(320, dtype=H, device=gpu)
    args = lambda rows, samples, p1: (x.data_ptr(), 320, p1, None, 0, None, x.data_ptr(), samples, rows, ga.data_ptr(), ga.data_ptr(), 1e-5, 0,
                                      ws.data_ptr(), ws.numel(), None)
    assert lib.syn3r_groupnorm_pre_f16(*args(32, 2, None)) != 0 and b"partial sums" in lib.syn3r_last_error()
    assert lib.syn3r_groupnorm_pre_f16(*args(16, 4, part.data_ptr())) != 0 and b"rows" in lib.syn3r_last_error()
    assert lib.syn3r_gemm_set_gn_partials(part.data_ptr(), 0) != 0
    assert lib.syn3r_gn_partials_bytes(1000, 320) == 0 and lib.syn3r_gn_partials_bytes(1024, 320) == 32 * 2 * 32 * 4
    torch.cuda.synchronize()
