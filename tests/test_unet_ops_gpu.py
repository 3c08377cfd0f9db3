"""Each hand-written UNet operator vs a plain PyTorch fp32 reference of the same op (inputs are the
same fp16-rounded values; fp32 accumulation on both sides; tolerance = fp16 output rounding)."""
import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu

H = torch.float16


def rnd(gen, *shape, scale=1.0, dev=None):
    return (torch.randn(*shape, generator=gen) * scale).to(H).to(dev)


def close(a, b, tol=2e-3):
    a, b = a.float(), b.float()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-6
    assert err <= tol * ref + 1e-3, (err, ref)


@pytest.mark.parametrize("M,N,K", [(256, 160, 64), (300, 320, 128), (1000, 2560, 320), (37, 48, 192), (513, 960, 640)])
def test_gemm_plain(M, N, K, gpu):
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(M + N)
    x, w, b = rnd(g, M, K, dev=gpu), rnd(g, N, K, scale=K ** -0.5, dev=gpu), rnd(g, N, dev=gpu)
    close(ops.linear(x, w, b), x.float() @ w.float().T + b.float())
    close(ops.linear(x, w), x.float() @ w.float().T)


def test_gemm_asymmetric_identity(gpu):
    """A = I with an asymmetric B catches a transposed accumulator map."""
    from syn3r_amd.unet import ops
    K = 128
    x = torch.eye(K, dtype=H, device=gpu)
    w = (torch.arange(160 * K, device=gpu).reshape(160, K) % 61).to(H)
    assert torch.equal(ops.linear(x, w), w.T.contiguous())


def test_gemm_epilogue(gpu):
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(5)
    M, N, K, rpv = 600, 320, 320, 100
    x, w, b = rnd(g, M, K, dev=gpu), rnd(g, N, K, scale=K ** -0.5, dev=gpu), rnd(g, N, dev=gpu)
    rv, res, aux = rnd(g, M // rpv, N, dev=gpu), rnd(g, M, N, dev=gpu), rnd(g, M, N, dev=gpu)
    out = ops.linear(x, w, b, rowvec=rv, rows_per_vec=rpv, residual=res, aux=aux, s_acc=0.3, s_res=1.0, s_aux=0.7)
    ref = 0.3 * (x.float() @ w.float().T + b.float() + rv.float().repeat_interleave(rpv, 0)) + res.float() + 0.7 * aux.float()
    close(out, ref)
    # strided A (column slice of a wider matrix) and strided output
    wide = rnd(g, M, 3 * K, dev=gpu)
    close(ops.linear(wide[:, K:2 * K], w), wide[:, K:2 * K].float() @ w.float().T)


@pytest.mark.parametrize("NB,Hi,Wi,Cin,Cout,stride,ups", [(2, 9, 16, 64, 160, 1, False), (3, 18, 32, 128, 64, 2, False),
                                                          (2, 5, 7, 64, 320, 1, True), (1, 40, 72, 320, 320, 1, False),
                                                          (2, 9, 7, 64, 64, 2, False)])
def test_conv3x3(NB, Hi, Wi, Cin, Cout, stride, ups, gpu):
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(NB * Hi + Cout)
    x = rnd(g, NB, Hi, Wi, Cin, dev=gpu)
    w = rnd(g, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, dev=gpu)
    b = rnd(g, Cout, dev=gpu)
    xin = x.permute(0, 3, 1, 2).float()
    if ups:
        xin = Fn.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = Fn.conv2d(xin, w.float(), b.float(), stride=stride, padding=1).permute(0, 2, 3, 1)
    out = ops.conv3x3(x, w.permute(0, 2, 3, 1).contiguous(), b, stride=stride, upsample=ups)
    assert out.shape == ref.shape
    close(out, ref)


@pytest.mark.parametrize("NB,Hi,Wi,Cin,Cout", [(2, 5, 7, 64, 320), (3, 9, 16, 128, 328), (1, 18, 32, 192, 640)])
def test_conv3x3_upsample_every_kernel(NB, Hi, Wi, Cin, Cout, gpu, every_contraction_kernel):
    """upsampling.py:172-183 (nearest 2x, then the 3 x 3 convolution) fused into the gather, through every kernel family:
    odd and even image sizes (both parities of the output rows / columns at the borders), several images, a residual."""
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(NB * Hi + Cout)
    x = rnd(g, NB, Hi, Wi, Cin, dev=gpu)
    w = rnd(g, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, dev=gpu)
    b = rnd(g, Cout, dev=gpu)
    res = rnd(g, NB * 4 * Hi * Wi, Cout, dev=gpu)
    xin = Fn.interpolate(x.permute(0, 3, 1, 2).float(), scale_factor=2.0, mode="nearest")
    ref = Fn.conv2d(xin, w.float(), b.float(), padding=1).permute(0, 2, 3, 1)
    wk = w.permute(0, 2, 3, 1).contiguous()
    outs = {}

    def body(tile):
        out = ops.conv3x3(x, wk, b, upsample=True)
        assert out.shape == ref.shape
        close(out, ref)
        out_r = ops.conv3x3(x, wk, b, upsample=True, residual=res, s_acc=0.5, s_res=1.0)
        close(out_r.reshape(-1, Cout), 0.5 * ref.reshape(-1, Cout) + res.float())
        outs[tile] = out
    every_contraction_kernel(body)
    for tile, out in outs.items():                      # same products, same k order inside a 64-channel chunk: tight agreement
        assert (out.float() - outs[0].float()).abs().max().item() <= 2e-3 * ref.abs().max().item(), tile


def test_conv3x3_epilogue(gpu):
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(11)
    NB, Hh, Ww, C = 4, 8, 12, 64
    x, w, b = rnd(g, NB, Hh, Ww, C, dev=gpu), rnd(g, 128, C, 3, 3, scale=0.05, dev=gpu), rnd(g, 128, dev=gpu)
    temb, res = rnd(g, NB, 128, dev=gpu), rnd(g, NB, Hh, Ww, 128, dev=gpu)
    out = ops.conv3x3(x, w.permute(0, 2, 3, 1).contiguous(), b, rowvec=temb, rows_per_vec=Hh * Ww,
                      residual=res.reshape(-1, 128), s_acc=1.0, s_res=1.0)
    ref = Fn.conv2d(x.permute(0, 3, 1, 2).float(), w.float(), b.float(), padding=1).permute(0, 2, 3, 1)
    ref = ref + temb.float()[:, None, None, :] + res.float()
    close(out, ref)


@pytest.mark.parametrize("B,F,HW,Cin,Cout", [(2, 14, 24, 64, 64), (1, 25, 30, 128, 320), (2, 5, 7, 64, 160)])
def test_tconv3(B, F, HW, Cin, Cout, gpu):
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(F + HW)
    x = rnd(g, B * F * HW, Cin, dev=gpu)
    w = rnd(g, Cout, Cin, 3, scale=(3 * Cin) ** -0.5, dev=gpu)   # Conv3d weight [Cout,Cin,3,1,1] squeezed
    b = rnd(g, Cout, dev=gpu)
    res = rnd(g, B * F * HW, Cout, dev=gpu)
    x5 = x.float().reshape(B, F, HW, Cin).permute(0, 3, 1, 2)[..., None]    # [B,C,F,HW,1]
    ref = Fn.conv3d(x5, w.float()[..., None, None], b.float(), padding=(1, 0, 0))
    ref = ref[..., 0].permute(0, 2, 3, 1).reshape(B * F * HW, Cout)
    out = ops.tconv3(x, w.permute(0, 2, 1).contiguous(), b, B, F, HW, residual=res, s_acc=0.6, s_res=1.0)
    close(out, 0.6 * ref + res.float())


# 128-query blocks, 64-key tiles: sizes with a partial last query block and a partial last key tile (45, 130, 216, 520, 700,
# 1000), whole tiles only (128, 512, 576, 2304) and the benchmark's sequence length (9216)
@pytest.mark.parametrize("nseq,S,heads", [(2, 128, 1), (3, 576, 2), (1, 2304, 5), (2, 45, 2), (2, 216, 1), (1, 130, 3),
                                          (2, 512, 1), (1, 520, 2), (2, 700, 1), (1, 1000, 3), (1, 9216, 1)])
def test_attention_spatial(nseq, S, heads, gpu):
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(S)
    C = heads * 64
    qkv = rnd(g, nseq * S, 3 * C, dev=gpu)
    out = ops.attention(qkv, nseq, S, heads)
    q, k, v = [t.float().reshape(nseq, S, heads, 64).transpose(1, 2) for t in qkv.split(C, dim=1)]
    ref = Fn.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(nseq * S, C)
    close(out, ref, tol=3e-3)


def test_attention_spatial_peaked_softmax(gpu):
    """Force the online-softmax rescale: one key row dominates late in the sequence."""
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(3)
    S, C = 512, 64
    qkv = rnd(g, S, 3 * C, dev=gpu)
    qkv[400, C:2 * C] = qkv[7, :C] * 6.0      # key 400 aligned with query 7, far larger logit
    out = ops.attention(qkv, 1, S, 1)
    q, k, v = [t.float().reshape(1, S, 1, 64).transpose(1, 2) for t in qkv.split(C, dim=1)]
    ref = Fn.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(S, C)
    close(out, ref, tol=3e-3)


@pytest.mark.parametrize("growth", [0.5, 3.0, 12.0])
def test_attention_spatial_moving_reference(growth, gpu):
    """The softmax reference the kernel subtracts is moved lazily (only past a threshold): keys whose logits grow along
    the sequence, slowly (below the threshold per tile), faster and in jumps larger than fp16 could hold without the move;
    and large negative logits in the first tile (the reference starts below zero)."""
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(11)
    S, C = 1000, 64
    qkv = rnd(g, S, 3 * C, dev=gpu)
    q = qkv[:, :C].float()
    # key j = a copy of the mean query direction scaled so that its logit grows ~linearly along the sequence
    d = q.mean(0)
    d = d / d.norm()
    qkv[:, :C] = (q + 2.0 * d).half()                                    # every query has a positive component along d
    ramp = torch.linspace(-growth * 8, growth * 8, S, device=gpu)[:, None]
    qkv[:, C:2 * C] = (qkv[:, C:2 * C].float() + ramp * d).half()
    out = ops.attention(qkv, 1, S, 1)
    qq, k, v = [t.float().reshape(1, S, 1, 64).transpose(1, 2) for t in qkv.split(C, dim=1)]
    ref = Fn.scaled_dot_product_attention(qq, k, v).transpose(1, 2).reshape(S, C)
    assert torch.isfinite(out.float()).all()
    close(out, ref, tol=3e-3)


@pytest.mark.parametrize("B,F,HW,heads", [(2, 14, 50, 2), (1, 25, 33, 5), (2, 3, 7, 1), (1, 32, 9, 1)])
def test_attention_temporal(B, F, HW, heads, gpu):
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(F * HW)
    C = heads * 64
    qkv = rnd(g, B * F * HW, 3 * C, dev=gpu)
    out = ops.attention_temporal(qkv, B, F, HW, heads)
    def seqs(t):   # [B,F,HW,heads,64] -> [B*HW, heads, F, 64]
        return t.float().reshape(B, F, HW, heads, 64).permute(0, 2, 3, 1, 4).reshape(B * HW, heads, F, 64)
    q, k, v = [seqs(t) for t in qkv.split(C, dim=1)]
    ref = Fn.scaled_dot_product_attention(q, k, v)            # [B*HW, heads, F, 64]
    ref = ref.reshape(B, HW, heads, F, 64).permute(0, 3, 1, 2, 4).reshape(B * F * HW, C)
    close(out, ref, tol=3e-3)


@pytest.mark.parametrize("samples,rows,C,silu", [(4, 200, 320, True), (2, 1000, 64, False), (3, 77, 2560, True),
                                                 (2, 512, 960, True), (28, 144, 1280, False)])
def test_groupnorm(samples, rows, C, silu, gpu):
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(rows + C)
    x = (rnd(g, samples * rows, C, dev=gpu).float() * 2 + 0.5).to(H)
    ga, be = rnd(g, C, dev=gpu), rnd(g, C, dev=gpu)
    out = ops.groupnorm(x, ga, be, samples, 1e-5, silu)
    xr = x.float().reshape(samples, rows, C).permute(0, 2, 1)
    ref = Fn.group_norm(xr, 32, ga.float(), be.float(), 1e-5)
    if silu:
        ref = Fn.silu(ref)
    close(out, ref.permute(0, 2, 1).reshape(samples * rows, C), tol=3e-3)


@pytest.mark.parametrize("M,C", [(1000, 320), (37, 1280), (513, 64), (64, 640), (257, 200), (70, 2560), (33, 1288)])
def test_layernorm(M, C, gpu):
    """(C = 320 / 640 / 1280 / 2560: the exact five-vectors-per-lane kernels of round 6; 64, 200, 1288: the guarded one, 200 and 1288
    with a ragged last vector group)"""
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(M)
    x, ga, be = rnd(g, M, C, dev=gpu), rnd(g, C, dev=gpu), rnd(g, C, dev=gpu)
    close(ops.layernorm(x, ga, be), Fn.layer_norm(x.float(), (C,), ga.float(), be.float(), 1e-5), tol=3e-3)
    rpv = 8 if M % 8 == 0 else 1
    add = rnd(g, M // rpv, C, dev=gpu)
    y, s = ops.layernorm(x, ga, be, addvec=add, rows_per_vec=rpv, want_sum=True)
    xs = x + add.repeat_interleave(rpv, 0)            # fp16 add, as the reference
    assert torch.equal(s, xs)
    close(y, Fn.layer_norm(xs.float(), (C,), ga.float(), be.float(), 1e-5), tol=3e-3)


def test_geglu(gpu):
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(2)
    x = rnd(g, 333, 2 * 640, scale=2.0, dev=gpu)
    hid, gate = x.float().chunk(2, dim=-1)
    close(ops.geglu(x), hid * Fn.gelu(gate))


@pytest.mark.parametrize("M,D,K", [(500, 1280, 320), (300, 256, 64), (77, 80, 128), (1000, 2560, 640)])
def test_gemm_geglu_fused(M, D, K, gpu):
    """Fused projection + gate vs the two-step fp32 reference; also bit-equal to the unfused HIP kernels."""
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(D)
    x, w, b = rnd(g, M, K, dev=gpu), rnd(g, 2 * D, K, scale=K ** -0.5, dev=gpu), rnd(g, 2 * D, dev=gpu)
    wp, bp, d = ops.pack_geglu(w, b)
    assert d == D and wp.shape[0] == ((D + 79) // 80) * 160
    out = ops.linear_geglu(x, wp, bp, D)
    proj = (x.float() @ w.float().T + b.float()).to(H).float()      # projection rounded to fp16, as the reference
    hid, gate = proj.chunk(2, dim=-1)
    close(out, hid * Fn.gelu(gate))
    assert torch.equal(out, ops.geglu(ops.linear(x, w, b)))


def test_ops_reject_bad_input(gpu):
    from syn3r_amd import _lib
    from syn3r_amd.unet import ops
    x = torch.zeros(8, 100, dtype=H, device=gpu)     # K not a multiple of 64
    w = torch.zeros(16, 100, dtype=H, device=gpu)
    with pytest.raises(_lib.Syn3rError):
        ops.linear(x, w)
    with pytest.raises(_lib.Syn3rError):
        ops.linear(torch.zeros(8, 64, dtype=H), torch.zeros(16, 64, dtype=H))   # CPU tensors
    with pytest.raises(_lib.Syn3rError):
        ops.attention_temporal(torch.zeros(40 * 2, 192, dtype=H, device=gpu), 1, 40, 2, 1)   # F > 32


@pytest.fixture
def every_contraction_kernel():
    """Run a test body once per contraction kernel: the per-shape default and every forced variant of
    `syn3r_gemm_set_tile` (LDS-DMA 128/256 and the two persistent 256x320 kernels)."""
    from syn3r_amd import _lib
    lib = _lib.load()

    def run(body):
        try:
            for tile in (0, -128, -256, -320, -322):
                _lib.check(lib.syn3r_gemm_set_tile(tile), "set_tile")
                body(tile)
        finally:
            lib.syn3r_gemm_set_tile(0)
    return run


def test_every_contraction_kernel_agrees(gpu, every_contraction_kernel):
    """Dense / conv3x3 / tconv3 / fused GEGLU with the full epilogue on ragged sizes (M, N not multiples of the
    tiles, N crossing the 320-column wide tile), through every kernel variant."""
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(77)
    M, N, K, rpv = 700, 400, 192, 100
    x, w, b = rnd(g, M, K, dev=gpu), rnd(g, N, K, scale=K ** -0.5, dev=gpu), rnd(g, N, dev=gpu)
    rv, res, aux = rnd(g, M // rpv, N, dev=gpu), rnd(g, M, N, dev=gpu), rnd(g, M, N, dev=gpu)
    ref_lin = 0.3 * (x.float() @ w.float().T + b.float() + rv.float().repeat_interleave(rpv, 0)) + res.float() + 0.7 * aux.float()
    xc = rnd(g, 2, 9, 11, 64, dev=gpu)
    wc, bc = rnd(g, 328, 3, 3, 64, scale=(9 * 64) ** -0.5, dev=gpu), rnd(g, 328, dev=gpu)
    ref_conv = Fn.conv2d(xc.float().permute(0, 3, 1, 2), wc.float().permute(0, 3, 1, 2), bc.float(), padding=1).permute(0, 2, 3, 1)
    B, F, HW = 2, 5, 13
    xt = rnd(g, B * F * HW, 64, dev=gpu)
    wt, bt = rnd(g, 336, 3, 64, scale=(3 * 64) ** -0.5, dev=gpu), rnd(g, 336, dev=gpu)
    ref_t = Fn.conv3d(xt.float().view(B, F, HW, 64).permute(0, 3, 1, 2)[..., None], wt.float().permute(0, 2, 1)[..., None, None],
                      bt.float(), padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(-1, 336)
    D = 200
    wg, bg = rnd(g, 2 * D, K, scale=K ** -0.5, dev=gpu), rnd(g, 2 * D, dev=gpu)
    wp, bp, _ = ops.pack_geglu(wg, bg)
    y = (x.float() @ wg.float().T + bg.float()).half().float()
    ref_g = y[:, :D] * Fn.gelu(y[:, D:])

    # more than one 64-channel chunk per tap (k_gemm_z walks K chunk-major with the taps innermost, the others tap-major) and
    # the stride-2 / (0,1,0,1)-padded downsampling form (downsampling.py:116-148)
    xc2 = rnd(g, 2, 10, 12, 192, dev=gpu)
    wc2, bc2 = rnd(g, 328, 3, 3, 192, scale=(9 * 192) ** -0.5, dev=gpu), rnd(g, 328, dev=gpu)
    nchw = lambda t: t.float().permute(0, 3, 1, 2)
    ref_conv2 = Fn.conv2d(nchw(xc2), nchw(wc2), bc2.float(), padding=1).permute(0, 2, 3, 1)
    ref_conv2s = Fn.conv2d(Fn.pad(nchw(xc2), (0, 1, 0, 1)), nchw(wc2), bc2.float(), stride=2).permute(0, 2, 3, 1)

    def body(tile):
        close(ops.linear(x, w, b, rowvec=rv, rows_per_vec=rpv, residual=res, aux=aux, s_acc=0.3, s_res=1.0, s_aux=0.7), ref_lin)
        close(ops.conv3x3(xc, wc, bc), ref_conv)
        close(ops.conv3x3(xc2, wc2, bc2), ref_conv2)
        close(ops.conv3x3(xc2, wc2, bc2, stride=2, pad_lo=0), ref_conv2s)
        close(ops.tconv3(xt, wt, bt, B, F, HW), ref_t)
        close(ops.linear_geglu(x, wp, bp, D), ref_g, tol=4e-3)
    every_contraction_kernel(body)


@pytest.mark.parametrize("M,N,K", [(5120, 4800, 128), (5000, 4808, 192), (8192, 1920, 256), (66000, 320, 64)])
def test_gemm_persistent_wide_tile(M, N, K, gpu):
    """k_gemm_widep (persistent 256 x 320 tile) with MORE tiles than CUs, so blocks walk several tiles: even k-tile counts
    (next tile's stage 0 prefetched during the last k-tile, epilogue staged behind ring slot 0) and odd ones, ragged
    M / N (multiples of 8), a narrow last column band, bias / scale / residual / aux epilogues and the GEGLU gate."""
    from syn3r_amd.unet import ops
    from syn3r_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = rnd(g, M, K, dev=gpu), rnd(g, N, K, scale=K ** -0.5, dev=gpu), rnd(g, N, dev=gpu)
    res, aux = rnd(g, M, N, dev=gpu), rnd(g, M, N, dev=gpu)
    y = x.float() @ w.float().T
    try:
      for forced in (-320, -322):                                        # both persistent 256 x 320 kernels
        _lib.check(lib.syn3r_gemm_set_tile(forced), "set_tile")
        close(ops.linear(x, w), y)
        close(ops.linear(x, w, b, s_acc=0.5), 0.5 * (y + b.float()))
        close(ops.linear(x, w, b, residual=res), y + b.float() + res.float())
        close(ops.linear(x, w, b, residual=res, aux=aux, s_acc=0.3, s_res=1.0, s_aux=0.7), 0.3 * (y + b.float()) + res.float() + 0.7 * aux.float())
        rpv = M // 8                                                    # per-sample row vectors, both indexings
        rv = rnd(g, 8, N, dev=gpu)
        close(ops.linear(x, w, b, rowvec=rv, rows_per_vec=rpv, residual=res),
              y + b.float() + rv.float().repeat_interleave(rpv, 0)[:M] + res.float())
        close(ops.linear(x, w, b, rowvec=rv, rows_per_vec=-8), y + b.float() + rv.float().repeat(M // 8 + 1, 1)[:M])
        if N % 16 == 0:
            D = N // 2
            wp, bp, _ = ops.pack_geglu(w, b)
            yh = (y + b.float()).half().float()
            close(ops.linear_geglu(x, wp, bp, D), yh[:, :D] * Fn.gelu(yh[:, D:]), tol=4e-3)
    finally:
        lib.syn3r_gemm_set_tile(0)


@pytest.mark.parametrize("M,C,D", [(700, 320, 1280), (129, 64, 128), (1000, 640, 2560), (4032, 1280, 5120), (77, 128, 192)])
def test_feedforward_tiled_intermediate(M, C, D, gpu):
    """`feedforward` (gated hidden activation in the tiled workspace) equals the two separate launches with the
    row-major intermediate bit for bit (same kernels, same arithmetic; only where the intermediate lives differs), with
    the full epilogue, on ragged M and through the default and the forced tile kernels."""
    from syn3r_amd import _lib
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(M + D)
    x = rnd(g, M, C, dev=gpu)
    w1, b1 = rnd(g, 2 * D, C, scale=C ** -0.5, dev=gpu), rnd(g, 2 * D, dev=gpu)
    w2, b2 = rnd(g, C, D, scale=D ** -0.5, dev=gpu), rnd(g, C, dev=gpu)
    res, aux = rnd(g, M, C, dev=gpu), rnd(g, M, C, dev=gpu)
    wp, bp, _ = ops.pack_geglu(w1, b1)
    lib = _lib.load()
    try:
        for tile in (0, -128, -256, -320, -322):
            _lib.check(lib.syn3r_gemm_set_tile(tile), "set_tile")
            ref = ops.linear(ops.linear_geglu(x, wp, bp, D), w2, b2, residual=res, aux=aux, s_acc=0.4, s_res=0.6, s_aux=0.25)
            out = ops.feedforward(x, wp, bp, D, w2, b2, residual=res, aux=aux, s_acc=0.4, s_res=0.6, s_aux=0.25)
            assert torch.equal(out, ref), (tile, (out.float() - ref.float()).abs().max().item())
            assert torch.equal(ops.feedforward(x, wp, bp, D, w2, b2), ops.linear(ops.linear_geglu(x, wp, bp, D), w2, b2))
    finally:
        lib.syn3r_gemm_set_tile(0)
    # against the fp32 restatement of FeedForward (attention.py:608-665): projection and gate rounded to fp16
    y = (x.float() @ w1.float().T + b1.float()).half().float()
    h = (y[:, :D] * Fn.gelu(y[:, D:])).half().float()
    close(ops.feedforward(x, wp, bp, D, w2, b2), h @ w2.float().T + b2.float(), tol=4e-3)


@pytest.mark.parametrize("M,C,D", [(256, 128, 128), (384, 128, 128), (2048, 640, 2560), (16128, 1280, 5120), (28800, 1280, 5120),
                                   (70144, 192, 384), (768, 320, 1280)])
def test_feedforward_g256_equals_packed80(M, C, D, gpu):
    """net.0 on the persistent 256 x 256 tile (k_gemm_g256, syn3r_feedforward_p64_f16: [64 hidden | 64 gate] packing, three A and
    two B ring slots, LDS-free epilogue into the tiled hidden activation) reproduces syn3r_feedforward_f16 bit for bit - one tile,
    fewer tiles than CUs, several tiles per block (the stage cursors cross tile boundaries; 70144 rows x 3 column tiles = 822
    tiles), a HALF last row tile (M = 384; 28 800 = the level-2 rows of an F = 25 unit), K of 2 to 20 k-tiles - and both match the fp32 restatement of FeedForward (attention.py:608-665)."""
    from syn3r_amd import _lib
    from syn3r_amd.unet import ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + C + D)
    x = rnd(g, M, C, dev=gpu)
    w1, b1 = rnd(g, 2 * D, C, scale=C ** -0.5, dev=gpu), rnd(g, 2 * D, dev=gpu)
    w2, b2 = rnd(g, C, D, scale=D ** -0.5, dev=gpu), rnd(g, C, dev=gpu)
    res = rnd(g, M, C, dev=gpu)
    wp, bp, _ = ops.pack_geglu(w1, b1)
    w64, b64, _ = ops.pack_geglu64(w1, b1)
    assert lib.syn3r_feedforward_p64_supported(M, D, C) == 1
    ref = ops.feedforward(x, wp, bp, D, w2, b2, residual=res)
    out = ops.feedforward(x, wp, bp, D, w2, b2, residual=res, packed64=(w64, b64))
    assert torch.equal(out, ref), (out.float() - ref.float()).abs().max().item()
    again = ops.feedforward(x, wp, bp, D, w2, b2, residual=res, packed64=(w64, b64))
    assert torch.equal(out, again)
    if M <= 4096:
        y = (x.float() @ w1.float().T + b1.float()).half().float()
        h = (y[:, :D] * Fn.gelu(y[:, D:])).half().float()
        close(out, h @ w2.float().T + b2.float() + res.float(), tol=4e-3)
    # shapes without whole tiles are refused by the entry point and not offered by the wrapper
    assert lib.syn3r_feedforward_p64_supported(M + 8, D, C) == 0 and lib.syn3r_feedforward_p64_supported(M, D + 64, C) == 0 and lib.syn3r_feedforward_p64_supported(128, D, C) == 0
    xr = rnd(g, M + 8, C, dev=gpu)
    assert torch.equal(ops.feedforward(xr, wp, bp, D, w2, b2, packed64=(w64, b64)), ops.feedforward(xr, wp, bp, D, w2, b2))


@pytest.mark.parametrize("M,D", [(128, 64), (700, 1280), (4097, 1280), (129, 192)])
def test_feedforward_fused_layernorm_c320(M, D, gpu):
    """norm3 -> ff in one kernel (syn3r_feedforward_fused_ln_f16): bit for bit the LayerNorm launch followed by the fused
    feed-forward (same statistics, same order of additions), and the fp32 restatement of attention.py:376-392 within the
    feed-forward's tolerance; ragged M, rows with a large offset (the two-pass variance), strided x, the full epilogue."""
    from syn3r_amd.unet import ops
    C = 320
    g = torch.Generator().manual_seed(M * 3 + D)
    wide = rnd(g, M, 2 * C, dev=gpu)
    x = wide[:, C // 2:C // 2 + C]                                      # row stride 640: a column slice
    x[::7] += 6.0                                                        # some rows far from zero mean
    ga, be = (1.0 + 0.2 * rnd(g, C, dev=gpu).float()).half(), (0.1 * rnd(g, C, dev=gpu).float()).half()
    w1, b1 = rnd(g, 2 * D, C, scale=C ** -0.5, dev=gpu), rnd(g, 2 * D, dev=gpu)
    w2, b2 = rnd(g, C, D, scale=D ** -0.5, dev=gpu), rnd(g, C, dev=gpu)
    aux = rnd(g, M, C, dev=gpu)
    wc, bc, _ = ops.pack_geglu_chunked(w1, b1)
    n = ops.layernorm(x.contiguous(), ga, be)
    for kw in ({}, {"residual": x.contiguous(), "aux": aux, "s_acc": 0.4, "s_res": 0.6, "s_aux": 0.25}):
        two = ops.feedforward_fused(n, wc, bc, D, w2, b2, **kw)
        one = ops.feedforward_fused(x, wc, bc, D, w2, b2, ln=(ga, be, 1e-5), **kw)
        assert torch.equal(one, two), (one.float() - two.float()).abs().max().item()
    nf = Fn.layer_norm(x.float(), (C,), ga.float(), be.float(), 1e-5).half().float()
    y = (nf @ w1.float().T + b1.float()).half().float()
    h = (y[:, :D] * Fn.gelu(y[:, D:])).half().float()
    close(ops.feedforward_fused(x, wc, bc, D, w2, b2, ln=(ga, be, 1e-5)), h @ w2.float().T + b2.float(), tol=6e-3)
    with pytest.raises(ValueError):
        ops.feedforward_fused(x, wc, bc, D, w2, b2, ln=(ga[:64], be[:64], 1e-5))


@pytest.mark.parametrize("M,D", [(128, 64), (700, 1280), (4097, 1280), (129, 192)])
def test_feedforward_fused_c320(M, D, gpu):
    """`feedforward_fused` (syn3r_feedforward_fused_f16: x -> net.0 -> GEGLU -> net.2 in one kernel, C = 320) against the
    two-kernel path (same arithmetic: fp32 accumulation in k order, projection and gate rounded to fp16) and the fp32
    restatement of FeedForward (attention.py:608-665); ragged M, one and many hidden chunks, the full epilogue."""
    from syn3r_amd.unet import ops
    C = 320
    g = torch.Generator().manual_seed(M + D)
    x = rnd(g, M, C, dev=gpu)
    w1, b1 = rnd(g, 2 * D, C, scale=C ** -0.5, dev=gpu), rnd(g, 2 * D, dev=gpu)
    w2, b2 = rnd(g, C, D, scale=D ** -0.5, dev=gpu), rnd(g, C, dev=gpu)
    res, aux = rnd(g, M, C, dev=gpu), rnd(g, M, C, dev=gpu)
    wc, bc, d_ = ops.pack_geglu_chunked(w1, b1)
    assert d_ == D and wc.shape == (2 * D, C)
    wp, bp, _ = ops.pack_geglu(w1, b1)
    out = ops.feedforward_fused(x, wc, bc, D, w2, b2, residual=res, aux=aux, s_acc=0.4, s_res=0.6, s_aux=0.25)
    ref2 = ops.feedforward(x, wp, bp, D, w2, b2, residual=res, aux=aux, s_acc=0.4, s_res=0.6, s_aux=0.25)
    close(out, ref2.float(), tol=2e-3)
    y = (x.float() @ w1.float().T + b1.float()).half().float()
    h = (y[:, :D] * Fn.gelu(y[:, D:])).half().float()
    close(ops.feedforward_fused(x, wc, bc, D, w2, b2), h @ w2.float().T + b2.float(), tol=4e-3)
    close(ops.feedforward_fused(x, wc, bc, D, w2, None, residual=res), h @ w2.float().T + res.float(), tol=4e-3)
    # a column slice of a wider matrix as the input (row stride > C)
    wide = rnd(g, M, 2 * C, dev=gpu)
    close(ops.feedforward_fused(wide[:, C:], wc, bc, D, w2, b2),
          ((wide[:, C:].float() @ w1.float().T + b1.float()).half().float()[:, :D]
           * Fn.gelu((wide[:, C:].float() @ w1.float().T + b1.float()).half().float()[:, D:])).half().float() @ w2.float().T + b2.float(),
          tol=4e-3)
    assert torch.equal(out, ops.feedforward_fused(x, wc, bc, D, w2, b2, residual=res, aux=aux, s_acc=0.4, s_res=0.6, s_aux=0.25))


def test_feedforward_fused_rejects_other_widths(gpu):
    from syn3r_amd import _lib
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(2)
    x = rnd(g, 64, 640, dev=gpu)
    w1, b1 = rnd(g, 2 * 128, 640, dev=gpu), rnd(g, 2 * 128, dev=gpu)
    wc, bc, _ = ops.pack_geglu_chunked(w1, b1)
    with pytest.raises(_lib.Syn3rError, match="C = 320"):
        ops.feedforward_fused(x, wc, bc, 128, rnd(g, 640, 128, dev=gpu))


def test_feedforward_rejects_bad_input(gpu):
    from syn3r_amd import _lib
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(1)
    x = rnd(g, 64, 64, dev=gpu)
    w1, b1 = rnd(g, 2 * 96, 64, dev=gpu), rnd(g, 2 * 96, dev=gpu)      # D = 96 is not a multiple of 64
    wp, bp, _ = ops.pack_geglu(w1, b1)
    with pytest.raises(_lib.Syn3rError, match="multiple of 64"):
        ops.feedforward(x, wp, bp, 96, rnd(g, 64, 96, dev=gpu))


@pytest.mark.parametrize("M,K1,K2,N", [(5120, 128, 64, 4800), (4032, 1280, 1280, 1280), (1000, 64, 192, 320), (36, 64, 64, 48)])
def test_linear_cat_two_source(M, K1, K2, N, gpu):
    """[x1 | x2] @ W^T + b with the concatenation read in place by the persistent kernel (first k-tiles from x1, the rest
    from x2; tiles that prefetch their successor's first stage switch back to x1), against the concatenated fp32
    reference and, bit for bit, against the one-source kernel on the materialised concatenation.  (36 rows: not served
    by the two-source kernel -> the wrapper concatenates.)"""
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(M + N)
    x1, x2 = rnd(g, M, K1, dev=gpu), rnd(g, M, K2, dev=gpu)
    w, b = rnd(g, N, K1 + K2, scale=(K1 + K2) ** -0.5, dev=gpu), rnd(g, N, dev=gpu)
    got = ops.linear_cat(x1, x2, w, b)
    cat = torch.cat([x1, x2], 1)
    close(got, cat.float() @ w.float().T + b.float())
    # strided sources (column slices of wider tensors)
    wide1, wide2 = rnd(g, M, K1 + 64, dev=gpu), rnd(g, M, K2 + 128, dev=gpu)
    a, c = wide1[:, 64:], wide2[:, :K2]
    close(ops.linear_cat(a, c, w, b), torch.cat([a, c], 1).float() @ w.float().T + b.float())
    if M % 8 == 0:
        from syn3r_amd import _lib
        lib = _lib.load()
        try:                                                    # same kernel, same arithmetic order: identical bits
            _lib.check(lib.syn3r_gemm_set_tile(-320), "set_tile")
            assert torch.equal(got, ops.linear(cat, w, b))
            _lib.check(lib.syn3r_gemm_set_tile(-322), "set_tile")       # the software-pipelined kernel's two-source form
            assert torch.equal(ops.linear_cat(x1, x2, w, b), ops.linear(cat, w, b))
        finally:
            lib.syn3r_gemm_set_tile(0)


@pytest.mark.parametrize("samples,rows,C1,C2,silu", [(4, 100, 64, 64, True), (3, 64, 640, 320, True), (2, 37, 320, 640, False),
                                                     (2, 50, 1280, 1280, True)])
def test_groupnorm_two_source(samples, rows, C1, C2, silu, gpu):
    """GroupNorm over the channel concatenation [x1 | x2] read in place == GroupNorm of the materialised concatenation,
    bit for bit (640 + 320 channels: a group of 30 channels straddles the boundary between the sources)."""
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(C1 + C2 + rows)
    x1, x2 = rnd(g, samples * rows, C1, dev=gpu), rnd(g, samples * rows, C2, dev=gpu)
    ga, be = rnd(g, C1 + C2, dev=gpu), rnd(g, C1 + C2, dev=gpu)
    got = ops.groupnorm(x1, ga, be, samples, 1e-5, silu, x2=x2)
    exp = ops.groupnorm(torch.cat([x1, x2], 1), ga, be, samples, 1e-5, silu)
    assert got.shape == (samples * rows, C1 + C2) and torch.equal(got, exp)


def test_persistent_lds_dma_kernel_many_tiles(gpu):
    """k_gemm_dmap (persistent 256 x 160 tile): more tiles than CUs, so the DMA issue cursor crosses tile boundaries while
    the previous tile is still being multiplied and the epilogue stages through ONE ring slot in two passes - dense
    (ragged M, bias / row vector / residual / aux) and temporal convolution, against fp32 references and, bit for bit,
    against the one-tile-per-block kernel the same shapes used before."""
    import os
    from syn3r_amd.unet import ops
    from syn3r_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(21)
    M, N, K = 70000 + 8, 320, 192                        # 274 row tiles x 2 column tiles = 548 tiles, M not a tile multiple
    x, w, b = rnd(g, M, K, dev=gpu), rnd(g, N, K, scale=K ** -0.5, dev=gpu), rnd(g, N, dev=gpu)
    res, aux, rv = rnd(g, M, N, dev=gpu), rnd(g, M, N, dev=gpu), rnd(g, 4, N, dev=gpu)
    rpv = (M + 3) // 4
    y = x.float() @ w.float().T + b.float()
    B, F, HW, C = 2, 7, 2560, 64                         # 35840 rows x 320 columns: 280 tiles
    xt = rnd(g, B * F * HW, C, dev=gpu)
    wt, bt = rnd(g, 320, 3, C, scale=(3 * C) ** -0.5, dev=gpu), rnd(g, 320, dev=gpu)
    ref_t = Fn.conv3d(xt.float().view(B, F, HW, C).permute(0, 3, 1, 2)[..., None], wt.float().permute(0, 2, 1)[..., None, None],
                      bt.float(), padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(-1, 320)
    try:
        _lib.check(lib.syn3r_gemm_set_tile(-256), "set_tile")
        o1 = ops.linear(x, w, b, rowvec=rv, rows_per_vec=rpv, residual=res)
        close(o1, y + rv.float().repeat_interleave(rpv, 0)[:M] + res.float())
        o2 = ops.linear(x, w, b, residual=res, aux=aux, s_acc=0.3, s_res=1.0, s_aux=0.7)
        close(o2, 0.3 * y + res.float() + 0.7 * aux.float())
        o3 = ops.tconv3(xt, wt, bt, B, F, HW, residual=xt.repeat(1, 5).contiguous(), s_acc=0.5, s_res=1.5)
        close(o3, 0.5 * ref_t + 1.5 * xt.float().repeat(1, 5))
    finally:
        lib.syn3r_gemm_set_tile(0)


@pytest.mark.parametrize("NB,Hi,Wi,Cin,Cout", [(4, 8, 8, 128, 160), (28, 9, 16, 1280, 1280), (3, 6, 10, 128, 328), (2, 8, 8, 256, 160)])
def test_conv3x3_split_k(NB, Hi, Wi, Cin, Cout, gpu):
    """syn3r_gemm_set_splitk_workspace: the small-grid convolutions as two or four K parts (cut inside a filter tap too) + the finishing
    launch (bias, per-sample row vector, residual: the one-pass epilogue's arithmetic on the fp32 sum of the parts): equal to the
    one-pass launch up to the association of the fp32 sum, and to the fp32 reference within the usual tolerance; the split path
    is really taken (kernel trace), and a workspace that is too small falls back to one pass."""
    from syn3r_amd import _lib
    from syn3r_amd.unet import ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(NB * Hi + Cout)
    x = rnd(g, NB, Hi, Wi, Cin, dev=gpu)
    w = rnd(g, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, dev=gpu)
    b = rnd(g, Cout, dev=gpu)
    temb = rnd(g, NB, Cout, dev=gpu)
    res = rnd(g, NB * Hi * Wi, Cout, dev=gpu)
    wk = w.permute(0, 2, 3, 1).contiguous()
    kw = dict(rowvec=temb, rows_per_vec=Hi * Wi, residual=res, s_acc=0.7, s_res=1.0)
    ref = 0.7 * (Fn.conv2d(x.permute(0, 3, 1, 2).float(), w.float(), b.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
                 + temb.float().repeat_interleave(Hi * Wi, 0)) + res.float()
    one = ops.conv3x3(x, wk, b, **kw).reshape(-1, Cout)
    M = NB * Hi * Wi
    ws = torch.empty(4 * M * Cout * 4, dtype=torch.uint8, device=gpu)
    try:
        _lib.check(lib.syn3r_gemm_set_splitk_workspace(ws.data_ptr(), ws.numel()), "set_splitk")
        with _lib.kernel_trace() as tr:
            split = ops.conv3x3(x, wk, b, **kw).reshape(-1, Cout)
            torch.cuda.synchronize()
        assert any("k_splitk_finish" in k for k in tr.result), list(tr.result)
        _lib.check(lib.syn3r_gemm_set_splitk_workspace(ws.data_ptr(), 1024), "set_splitk")          # too small: one pass
        with _lib.kernel_trace() as tr2:
            small = ops.conv3x3(x, wk, b, **kw).reshape(-1, Cout)
            torch.cuda.synchronize()
        assert not any("k_splitk_finish" in k for k in tr2.result)
        assert torch.equal(small, one)
    finally:
        lib.syn3r_gemm_set_splitk_workspace(None, 0)
    close(split, ref)
    assert (split.float() - one.float()).abs().max().item() <= 2e-3 * ref.abs().max().item()
    assert lib.syn3r_gemm_set_splitk_workspace(ws.data_ptr(), 0) != 0            # pointer without a size


def test_tconv3_split_k(gpu):
    from syn3r_amd import _lib
    from syn3r_amd.unet import ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    B, F, HW, Cin, Cout = 2, 6, 20, 256, 192
    x = rnd(g, B * F * HW, Cin, dev=gpu)
    w = rnd(g, Cout, Cin, 3, scale=(3 * Cin) ** -0.5, dev=gpu)
    b, res = rnd(g, Cout, dev=gpu), rnd(g, B * F * HW, Cout, dev=gpu)
    x5 = x.float().reshape(B, F, HW, Cin).permute(0, 3, 1, 2)[..., None]
    ref = Fn.conv3d(x5, w.float()[..., None, None], b.float(), padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(B * F * HW, Cout)
    wk = w.permute(0, 2, 1).contiguous()
    one = ops.tconv3(x, wk, b, B, F, HW, residual=res, s_acc=0.6, s_res=1.0)
    ws = torch.empty(4 * B * F * HW * Cout * 4, dtype=torch.uint8, device=gpu)
    try:
        _lib.check(lib.syn3r_gemm_set_splitk_workspace(ws.data_ptr(), ws.numel()), "set_splitk")
        with _lib.kernel_trace() as tr:
            split = ops.tconv3(x, wk, b, B, F, HW, residual=res, s_acc=0.6, s_res=1.0)
            torch.cuda.synchronize()
        assert any("k_splitk_finish" in k for k in tr.result), list(tr.result)
    finally:
        lib.syn3r_gemm_set_splitk_workspace(None, 0)
    close(split, 0.6 * ref + res.float())
    assert (split.float() - one.float()).abs().max().item() <= 2e-3 * ref.abs().max().item()


def test_linear_split_k(gpu):
    """The dense contraction on the split-K path (a small grid, long K) with the full epilogue."""
    from syn3r_amd import _lib
    from syn3r_amd.unet import ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(21)
    M, N, K, rpv = 600, 320, 2560, 100
    x, w, b = rnd(g, M, K, dev=gpu), rnd(g, N, K, scale=K ** -0.5, dev=gpu), rnd(g, N, dev=gpu)
    rv, res, aux = rnd(g, M // rpv, N, dev=gpu), rnd(g, M, N, dev=gpu), rnd(g, M, N, dev=gpu)
    kw = dict(rowvec=rv, rows_per_vec=rpv, residual=res, aux=aux, s_acc=0.3, s_res=1.0, s_aux=0.7)
    ref = 0.3 * (x.float() @ w.float().T + b.float() + rv.float().repeat_interleave(rpv, 0)) + res.float() + 0.7 * aux.float()
    one = ops.linear(x, w, b, **kw)
    ws = torch.empty(4 * M * N * 4, dtype=torch.uint8, device=gpu)
    try:
        _lib.check(lib.syn3r_gemm_set_splitk_workspace(ws.data_ptr(), ws.numel()), "set_splitk")
        with _lib.kernel_trace() as tr:
            split = ops.linear(x, w, b, **kw)
            torch.cuda.synchronize()
        assert any("k_splitk_finish" in k for k in tr.result), list(tr.result)
    finally:
        lib.syn3r_gemm_set_splitk_workspace(None, 0)
    close(split, ref)
    assert (split.float() - one.float()).abs().max().item() <= 2e-3 * ref.abs().max().item()


def test_feedforward_split_k(gpu):
    """net.2 of the two-launch FeedForward reads its A operand from the tiled workspace; at level 3 of the UNet (4 032 rows,
    K = 5 120) its grid is small enough for the split-K path: same result as the unsplit launch within fp16 rounding."""
    from syn3r_amd import _lib
    from syn3r_amd.unet import ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(22)
    M, C, D = 4032, 1280, 5120
    x = rnd(g, M, C, dev=gpu)
    w1, b1 = rnd(g, 2 * D, C, scale=C ** -0.5, dev=gpu), rnd(g, 2 * D, dev=gpu)
    w2, b2, res = rnd(g, C, D, scale=D ** -0.5, dev=gpu), rnd(g, C, dev=gpu), rnd(g, M, C, dev=gpu)
    wp, bp, _ = ops.pack_geglu(w1, b1)
    one = ops.feedforward(x, wp, bp, D, w2, b2, residual=res)
    ws = torch.empty(4 * M * C * 4, dtype=torch.uint8, device=gpu)
    try:
        _lib.check(lib.syn3r_gemm_set_splitk_workspace(ws.data_ptr(), ws.numel()), "set_splitk")
        with _lib.kernel_trace() as tr:
            split = ops.feedforward(x, wp, bp, D, w2, b2, residual=res)
            torch.cuda.synchronize()
        assert any("k_splitk_finish" in k for k in tr.result), list(tr.result)
    finally:
        lib.syn3r_gemm_set_splitk_workspace(None, 0)
    y = (x.float() @ w1.float().T + b1.float()).half().float()
    h = (y[:, :D] * Fn.gelu(y[:, D:])).half().float()
    close(split, h @ w2.float().T + b2.float() + res.float(), tol=4e-3)
    assert (split.float() - one.float()).abs().max().item() <= 2e-3 * one.float().abs().max().item()


@pytest.mark.parametrize("M,K1,K2,N", [(4032, 1280, 1280, 1280), (1000, 256, 256, 320), (520, 1280, 640, 320)])
def test_linear_cat_split_k(M, K1, K2, N, gpu):
    """The two-source shortcut projection on the split-K path: the K parts of the second source read A2 (no part may
    straddle the boundary: K1 = 1280, K2 = 640 cannot be split in 2 or 4 equal parts that respect it and stays unsplit)."""
    from syn3r_amd import _lib
    from syn3r_amd.unet import ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + K2)
    big1, big2 = rnd(g, M, K1 + 64, dev=gpu), rnd(g, M, K2 + 8, dev=gpu)
    x1, x2 = big1[:, 32:32 + K1], big2[:, 8:]                    # strided sources
    w, b = rnd(g, N, K1 + K2, scale=(K1 + K2) ** -0.5, dev=gpu), rnd(g, N, dev=gpu)
    one = ops.linear_cat(x1, x2, w, b)
    ws = torch.empty(4 * M * N * 4, dtype=torch.uint8, device=gpu)
    try:
        _lib.check(lib.syn3r_gemm_set_splitk_workspace(ws.data_ptr(), ws.numel()), "set_splitk")
        with _lib.kernel_trace() as tr:
            split = ops.linear_cat(x1, x2, w, b)
            torch.cuda.synchronize()
        assert any("k_splitk_finish" in k for k in tr.result) == (K1 == K2), list(tr.result)
    finally:
        lib.syn3r_gemm_set_splitk_workspace(None, 0)
    ref = torch.cat([x1, x2], 1).float() @ w.float().T + b.float()
    close(split, ref)
    assert (split.float() - one.float()).abs().max().item() <= 2e-3 * ref.abs().max().item()


@pytest.mark.parametrize("M,D,rpv", [(128, 64, 32), (700, 1280, 144), (4097, 1280, 1000), (9216 * 2 + 5, 1280, 9216)])
def test_feedforward_fused_add_layernorm_c320(M, D, rpv, gpu):
    """(hidden + frame embedding) -> norm_in -> ff_in -> + (hidden + embedding) in one kernel
    (syn3r_feedforward_fused_addln_f16; attention.py:500-517): bit for bit the LayerNorm launch with the add vector (which
    also writes the sum) followed by the fused feed-forward with that sum as its residual; ragged M, strided x, aux + scales."""
    from syn3r_amd.unet import ops
    C = 320
    g = torch.Generator().manual_seed(M + D + rpv)
    wide = rnd(g, M, 2 * C, dev=gpu)
    x = wide[:, C // 2:C // 2 + C]
    vec = rnd(g, (M + rpv - 1) // rpv, C, dev=gpu)
    ga, be = (1.0 + 0.2 * rnd(g, C, dev=gpu).float()).half(), (0.1 * rnd(g, C, dev=gpu).float()).half()
    w1, b1 = rnd(g, 2 * D, C, scale=C ** -0.5, dev=gpu), rnd(g, 2 * D, dev=gpu)
    w2, b2 = rnd(g, C, D, scale=D ** -0.5, dev=gpu), rnd(g, C, dev=gpu)
    aux = rnd(g, M, C, dev=gpu)
    wc, bc, _ = ops.pack_geglu_chunked(w1, b1)
    n, xs = ops.layernorm(x.contiguous(), ga, be, addvec=vec, rows_per_vec=rpv, want_sum=True)
    assert torch.equal(xs, x + vec.repeat_interleave(rpv, 0)[:M])
    for kw in ({}, {"aux": aux, "s_acc": 0.4, "s_res": 0.6, "s_aux": 0.25}):
        two = ops.feedforward_fused(n, wc, bc, D, w2, b2, residual=xs, **kw)
        one = ops.feedforward_fused(x, wc, bc, D, w2, b2, ln=(ga, be, 1e-5), addvec=(vec, rpv), **kw)
        assert torch.equal(one, two), (one.float() - two.float()).abs().max().item()
    with pytest.raises(ValueError):
        ops.feedforward_fused(x, wc, bc, D, w2, b2, addvec=(vec, rpv))                                  # no LayerNorm
    with pytest.raises(ValueError):
        ops.feedforward_fused(x, wc, bc, D, w2, b2, ln=(ga, be, 1e-5), addvec=(vec, rpv), residual=aux)   # the sum IS the residual
    with pytest.raises(ValueError):
        ops.feedforward_fused(x, wc, bc, D, w2, b2, ln=(ga, be, 1e-5), addvec=(vec[:-1], rpv))          # too few vectors


@pytest.mark.parametrize("M,N", [(128, 320), (700, 960), (4097, 960), (129, 640), (9216 * 3, 960)])
def test_layernorm_linear_c320(M, N, gpu):
    """norm1 -> stacked q / k / v projection in one kernel (syn3r_layernorm_linear320_f16; attention.py:340-352): against the
    LayerNorm launch followed by `linear` (same normalised fp16 values, same fp32 accumulation order along K: equal up to the
    order the 16-wide k groups reach the accumulator, i.e. within a few fp32 ulps before the fp16 rounding), and against the fp32
    restatement; ragged M (partial last block), strided x, rows far from zero mean."""
    from syn3r_amd.unet import ops
    C = 320
    g = torch.Generator().manual_seed(M + N)
    wide = rnd(g, M, 2 * C, dev=gpu)
    x = wide[:, C // 2:C // 2 + C]
    x[::5] += 4.0
    ga, be = (1.0 + 0.2 * rnd(g, C, dev=gpu).float()).half(), (0.1 * rnd(g, C, dev=gpu).float()).half()
    w = rnd(g, N, C, scale=C ** -0.5, dev=gpu)
    one = ops.layernorm_linear(x, ga, be, w)
    n = ops.layernorm(x.contiguous(), ga, be)
    two = ops.linear(n, w)
    assert one.shape == (M, N)
    ref = n.float() @ w.float().T                                        # on the SAME normalised fp16 values
    assert (one.float() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    assert (one.float() - two.float()).abs().max().item() <= 2e-3 * ref.abs().max().item()
    assert (one != two).float().mean().item() < 0.02                     # fp16 outputs: all but a few last-bit roundings equal
    nf = Fn.layer_norm(x.float(), (C,), ga.float(), be.float(), 1e-5).half().float()
    close(one, nf @ w.float().T, tol=6e-3)


def test_layernorm_linear_other_widths(gpu):
    """Widths the fused kernel is not built for take the two launches (Python host) / are rejected at the C-ABI."""
    from syn3r_amd import _lib
    from syn3r_amd.unet import ops
    g = torch.Generator().manual_seed(3)
    for C, N in ((640, 1920), (320, 328)):
        x, ga, be, w = rnd(g, 200, C, dev=gpu), rnd(g, C, dev=gpu), rnd(g, C, dev=gpu), rnd(g, N, C, scale=C ** -0.5, dev=gpu)
        assert torch.equal(ops.layernorm_linear(x, ga, be, w), ops.linear(ops.layernorm(x, ga, be), w))
    lib = _lib.load()
    x, ga, be, w = rnd(g, 200, 640, dev=gpu), rnd(g, 640, dev=gpu), rnd(g, 640, dev=gpu), rnd(g, 640, 640, dev=gpu)
    out = torch.empty(200, 640, dtype=torch.float16, device=gpu)
    rc = lib.syn3r_layernorm_linear320_f16(x.data_ptr(), 640, ga.data_ptr(), be.data_ptr(), 1e-5, w.data_ptr(), out.data_ptr(), 640, 200, 640, 640, None)
    assert rc != 0 and b"320" in lib.syn3r_last_error()


@pytest.mark.parametrize("M,D", [(128, 64), (700, 1280), (4097, 1280), (129, 192), (9216 * 3 + 40, 1280)])
def test_feedforward_fused_in_kernel_layernorm_equals_two_launches(M, D, gpu):
    """k_ffn320r (x tile as register fragments) normalises in registers with k_layernorm<8>'s arithmetic and summation order: the
    in-kernel LayerNorm (norm3 -> ff) and the add-vector form (norm_in / ff_in) equal the LayerNorm LAUNCH followed by the same
    kernel bit for bit, with residual, aux and scales; ragged M, strided x.  And the plain call against the fp32 restatement."""
    from syn3r_amd.unet import ops
    C = 320
    g = torch.Generator().manual_seed(M + 7 * D)
    wide = rnd(g, M, 2 * C, dev=gpu)
    x = wide[:, C // 2:C // 2 + C]
    x[::7] += 3.0
    ga, be = (1.0 + 0.2 * rnd(g, C, dev=gpu).float()).half(), (0.1 * rnd(g, C, dev=gpu).float()).half()
    w1, b1 = rnd(g, 2 * D, C, scale=C ** -0.5, dev=gpu), rnd(g, 2 * D, dev=gpu)
    w2, b2 = rnd(g, C, D, scale=D ** -0.5, dev=gpu), rnd(g, C, dev=gpu)
    aux, res = rnd(g, M, C, dev=gpu), rnd(g, M, C, dev=gpu)
    rpv = max(1, M // 3)
    vec = rnd(g, (M + rpv - 1) // rpv, C, dev=gpu)
    wc, bc, _ = ops.pack_geglu_chunked(w1, b1)
    xc = x.contiguous()
    # norm3 -> ff
    a = ops.feedforward_fused(x, wc, bc, D, w2, b2, ln=(ga, be, 1e-5), residual=res, aux=aux, s_acc=0.4, s_res=0.6, s_aux=0.25)
    b = ops.feedforward_fused(ops.layernorm(xc, ga, be, 1e-5), wc, bc, D, w2, b2, residual=res, aux=aux, s_acc=0.4, s_res=0.6, s_aux=0.25)
    assert torch.equal(a, b), (a.float() - b.float()).abs().max().item()
    # norm_in -> ff_in: x + vec is normalised AND is the residual
    a = ops.feedforward_fused(x, wc, bc, D, w2, b2, ln=(ga, be, 1e-5), addvec=(vec, rpv), aux=aux, s_acc=0.5, s_res=0.5, s_aux=0.5)
    y, xsum = ops.layernorm(xc, ga, be, 1e-5, addvec=vec, rows_per_vec=rpv, want_sum=True)
    b = ops.feedforward_fused(y, wc, bc, D, w2, b2, residual=xsum, aux=aux, s_acc=0.5, s_res=0.5, s_aux=0.5)
    assert torch.equal(a, b), (a.float() - b.float()).abs().max().item()
    # no bias on the second projection
    a = ops.feedforward_fused(x, wc, bc, D, w2, None)
    b = ops.feedforward_fused(x, wc, bc, D, w2, torch.zeros_like(b2))
    assert torch.equal(a, b)
    # and against the fp32 restatement
    y = (x.float() @ w1.float().T + b1.float()).half().float()
    h = (y[:, :D] * Fn.gelu(y[:, D:])).half().float()
    close(ops.feedforward_fused(x, wc, bc, D, w2, b2), h @ w2.float().T + b2.float(), tol=4e-3)
