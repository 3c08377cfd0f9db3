"""Fused trainer-loop operators (L1 loss, Adam) vs torch on the CPU (the published 3DGS step)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(3, 37, 53), (3, 270, 480), (1, 5), (3, 1080, 1920)])
def test_l1_loss_forward_backward(shape, gpu):
    from syn3r_amd.gs.train_ops import l1_loss
    g = torch.Generator().manual_seed(sum(shape))
    a = torch.rand(shape, generator=g)
    b = torch.rand(shape, generator=g)
    b.view(-1)[::7] = a.view(-1)[::7]                 # exact ties: sign(0) = 0 as torch.sign
    ad = a.double().requires_grad_(True)
    ref = 0.3 * (ad - b.double()).abs().mean()
    (ref * 2.5).backward()
    x = a.to(gpu).requires_grad_(True)
    loss = l1_loss(x, b.to(gpu), weight=0.3)
    (loss * 2.5).backward()
    assert loss.shape == () and loss.dtype == torch.float32
    assert abs(float(loss.detach()) - float(ref.detach())) <= 2e-6 * abs(float(ref.detach()))
    torch.testing.assert_close(x.grad.cpu(), ad.grad.float(), rtol=1e-6, atol=0)
    # fixed-order reduction: bitwise reproducible
    again = l1_loss(x.detach(), b.to(gpu), weight=0.3)
    assert float(again) == float(loss.detach())


def test_l1_loss_rejects_bad_input(gpu):
    from syn3r_amd import _lib
    from syn3r_amd.gs.train_ops import l1_loss
    a = torch.rand(3, 8, 8)
    with pytest.raises(_lib.Syn3rError):
        l1_loss(a, a.to(gpu))
    with pytest.raises(ValueError):
        l1_loss(a.to(gpu), a[:, :4].to(gpu))
    with pytest.raises(ValueError):
        l1_loss(a.to(gpu).half(), a.to(gpu).half())


def test_fused_adam_matches_torch_adam(gpu):
    from syn3r_amd.gs.train_ops import FusedAdam
    g = torch.Generator().manual_seed(5)
    shapes = [(1000, 3), (1000, 16, 3), (1000, 1), (777,)]
    lrs = [1.6e-4, 2.5e-3, 5e-2, 1e-3]
    ref_p = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    hip_p = [p.detach().clone().to(gpu).requires_grad_(True) for p in ref_p]
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref_p, lrs)], eps=1e-15)
    hip = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(hip_p, lrs)], eps=1e-15)
    for it in range(25):
        for rp, hp in zip(ref_p, hip_p):
            gr = torch.randn(rp.shape, generator=g) * (10.0 ** (-(it % 4)))
            if it == 3:
                gr[::5] = 0.0                         # zero gradients with eps = 1e-15 (0/eps path)
            rp.grad = gr
            hp.grad = gr.to(gpu)
        ref.step()
        hip.step()
    for rp, hp in zip(ref_p, hip_p):
        torch.testing.assert_close(hp.detach().cpu(), rp.detach(), rtol=2e-6, atol=1e-7)
        st = hip.state[hp]
        # moments: one fp32 rounding of the O(1) gradient terms survives cancellation -> absolute floor ~1 ulp(1)
        torch.testing.assert_close(st["exp_avg"].cpu(), ref.state[rp]["exp_avg"], rtol=1e-5, atol=2e-7)
        torch.testing.assert_close(st["exp_avg_sq"].cpu(), ref.state[rp]["exp_avg_sq"], rtol=1e-5, atol=2e-7)
    hip.zero_grad()
    assert all(p.grad is None for p in hip_p)


def test_adam_multi_tensor_equals_per_tensor_launches(gpu):
    """`syn3r_adam_step_multi` (one launch over a descriptor table; round 6) against `syn3r_adam_step` per tensor: the same bits in the
    parameters and both moments, for ragged sizes, different learning rates / eps / steps, and more tensors than one table holds
    (FusedAdam splits them); bad tables are refused."""
    import ctypes as C
    from syn3r_amd import _lib as L
    from syn3r_amd.gs.train_ops import FusedAdam
    lib = L.load()
    g = torch.Generator().manual_seed(8)
    sizes = [3 * 1000, 45 * 1000 + 7, 1, 255, 257, 4 * 1000, 1000, 513, 12345, 31]      # 10 tensors: two tables
    mk = lambda n: torch.randn(n, generator=g).to(gpu)
    P = [mk(n) for n in sizes]; G = [mk(n) for n in sizes]; M1 = [mk(n).abs() * 0.1 for n in sizes]; V1 = [mk(n).abs() * 0.01 for n in sizes]
    lrs = [10.0 ** (-2 - (k % 3)) for k in range(len(sizes))]
    epss = [1e-15 if k % 2 else 1e-8 for k in range(len(sizes))]
    steps = [1 + 3 * k for k in range(len(sizes))]
    ref = [(p.clone(), m.clone(), v.clone()) for p, m, v in zip(P, M1, V1)]
    for (p, m, v), gr, lr, ep, stp in zip(ref, G, lrs, epss, steps):
        L.check(lib.syn3r_adam_step(L.ptr(p), L.ptr(gr), L.ptr(m), L.ptr(v), p.numel(), lr, 0.9, 0.999, ep, stp, L.stream_ptr(gpu)), "adam_step")
    for k0 in (0, 8):
        idx = list(range(k0, min(k0 + 8, len(sizes))))
        n = len(idx)
        arr = lambda ts: (C.c_void_p * n)(*[ts[i].data_ptr() for i in idx])
        rc = lib.syn3r_adam_step_multi(n, arr(P), arr(G), arr(M1), arr(V1), (C.c_longlong * n)(*[sizes[i] for i in idx]),
                                       (C.c_float * n)(*[lrs[i] for i in idx]), 0.9, 0.999, (C.c_float * n)(*[epss[i] for i in idx]),
                                       (C.c_int * n)(*[steps[i] for i in idx]), L.stream_ptr(gpu))
        L.check(rc, "adam_step_multi")
    torch.cuda.synchronize()
    for (rp, rm, rv), p, m, v in zip(ref, P, M1, V1):
        assert torch.equal(rp, p) and torch.equal(rm, m) and torch.equal(rv, v)
    one = (C.c_void_p * 1)(P[0].data_ptr())
    ll, ff, ii = (C.c_longlong * 1)(sizes[0]), (C.c_float * 1)(1e-3), (C.c_int * 1)(1)
    assert lib.syn3r_adam_step_multi(0, one, one, one, one, ll, ff, 0.9, 0.999, ff, ii, None) != 0
    assert lib.syn3r_adam_step_multi(9, one, one, one, one, ll, ff, 0.9, 0.999, ff, ii, None) != 0
    assert lib.syn3r_adam_step_multi(1, one, one, one, one, ll, ff, 0.9, 0.999, ff, (C.c_int * 1)(0), None) != 0 and b"1-based" in lib.syn3r_last_error()
    # FusedAdam over ten groups (two launches) == ten torch.optim.Adam groups
    ref_p = [torch.randn(n, generator=g).requires_grad_(True) for n in sizes]
    hip_p = [p.detach().clone().to(gpu).requires_grad_(True) for p in ref_p]
    ro = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref_p, lrs)], eps=1e-15)
    ho = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(hip_p, lrs)], eps=1e-15)
    for _ in range(3):
        for rp, hp in zip(ref_p, hip_p):
            gr = torch.randn(rp.shape, generator=g)
            rp.grad, hp.grad = gr, gr.to(gpu)
        ro.step(); ho.step()
    for rp, hp in zip(ref_p, hip_p):
        torch.testing.assert_close(hp.detach().cpu(), rp.detach(), rtol=2e-6, atol=1e-7)


def test_fused_adam_skips_params_without_grad_and_rejects_cpu(gpu):
    from syn3r_amd import _lib
    from syn3r_amd.gs.train_ops import FusedAdam
    p = torch.ones(10, device=gpu, requires_grad=True)
    opt = FusedAdam([{"params": [p]}], lr=0.1)
    opt.step()                                        # no grad: untouched, no state
    assert bool((p == 1).all()) and p not in opt.state
    q = torch.ones(10, requires_grad=True)
    q.grad = torch.ones(10)
    with pytest.raises(_lib.Syn3rError):
        FusedAdam([{"params": [q]}]).step()


def _published_ssim(img1, img2):
    """The published 3DGS `ssim()` (utils/loss_utils.py of the 3DGS code base the FSGS trainer builds on):
    11x11 Gaussian window, sigma 1.5, conv2d padding 5, groups = channels, mean of the map."""
    import math
    import torch.nn.functional as Fn
    g = torch.tensor([math.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)], dtype=img1.dtype)
    g = (g / g.sum())[:, None]
    C = img1.shape[0]
    win = (g @ g.t())[None, None].expand(C, 1, 11, 11).contiguous()
    conv = lambda t: Fn.conv2d(t[None], win, padding=5, groups=C)[0]
    mu1, mu2 = conv(img1), conv(img2)
    s1, s2, s12 = conv(img1 * img1) - mu1 * mu1, conv(img2 * img2) - mu2 * mu2, conv(img1 * img2) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))).mean()


@pytest.mark.parametrize("shape,lam", [((3, 37, 53), 0.2), ((3, 64, 48), 0.2), ((1, 16, 16), 1.0), ((3, 20, 70), 0.0)])
def test_photometric_loss_vs_published_formula(shape, lam, gpu):
    from syn3r_amd.gs.train_ops import photometric_loss
    g = torch.Generator().manual_seed(11 + shape[1])
    a = torch.rand(shape, generator=g)
    b = (a + 0.2 * torch.randn(shape, generator=g)).clamp(0, 1)      # correlated target: SSIM well away from 0
    ad = a.double().requires_grad_(True)
    l1 = (ad - b.double()).abs().mean()
    ssim = _published_ssim(ad, b.double())
    ref = 0.7 * ((1 - lam) * l1 + lam * (1 - ssim))
    (ref * 1.5).backward()
    x = a.to(gpu).requires_grad_(True)
    loss, parts = photometric_loss(x, b.to(gpu), lambda_dssim=lam, weight=0.7, return_parts=True)
    (loss * 1.5).backward()
    assert abs(float(parts[1]) - float(l1.detach())) < 2e-6 and abs(float(parts[2]) - float(ssim.detach())) < 2e-5
    assert abs(float(loss.detach()) - float(ref.detach())) < 2e-5
    torch.testing.assert_close(x.grad.cpu(), ad.grad.float(), rtol=2e-3, atol=2e-7 + 2e-4 * float(ad.grad.abs().max()))


def test_photometric_loss_full_frame_and_errors(gpu):
    from syn3r_amd.gs.train_ops import photometric_loss
    g = torch.Generator().manual_seed(2)
    a = torch.rand(3, 1080, 1920, generator=g).to(gpu).requires_grad_(True)
    loss, parts = photometric_loss(a, a.detach().clone(), return_parts=True)       # identical images: SSIM = 1, L1 = 0
    loss.backward()
    assert abs(float(parts[2]) - 1.0) < 1e-5 and float(parts[1]) == 0.0 and abs(float(loss.detach())) < 1e-5
    assert float(a.grad.abs().max()) < 1e-6
    with pytest.raises(ValueError):
        photometric_loss(a.detach()[0], a.detach()[0])


@pytest.mark.parametrize("shape,lam", [((3, 37, 53), 0.2), ((3, 270, 480), 0.2), ((1, 16, 16), 1.0), ((3, 20, 70), 0.0)])
def test_photometric_loss_step_equals_forward_then_backward(shape, lam, gpu):
    """`syn3r_photo_loss_step` (value and gradient, the scalar sums formed by the gradient pass' first block) leaves the bits of
    `syn3r_photo_loss` + `syn3r_photo_loss_backward` in the three scalars and in the image gradient - with and without an
    upstream gradient."""
    from syn3r_amd.gs.train_ops import photometric_loss, photometric_loss_step
    g = torch.Generator().manual_seed(3 + shape[1])
    a = torch.rand(shape, generator=g).to(gpu)
    b = (a + 0.2 * torch.randn(shape, generator=g).to(gpu)).clamp(0, 1)
    for up in (None, 1.5):
        x = a.clone().requires_grad_(True)
        loss, parts = photometric_loss(x, b, lambda_dssim=lam, weight=0.7, return_parts=True)
        (loss if up is None else loss * up).backward()
        go = None if up is None else torch.tensor(up, device=gpu)
        l2, p2, grad = photometric_loss_step(a, b, lam, 0.7, grad_loss=go)
        assert torch.equal(p2, parts) and torch.equal(l2, loss.detach())
        assert torch.equal(grad, x.grad)
    with pytest.raises(ValueError):
        photometric_loss_step(a[0], b[0])


def test_densification_stats_vs_masked_torch(gpu):
    """syn3r_densification_stats == the published GaussianModel.add_densification_stats (boolean-mask indexing) bit for bit."""
    from syn3r_amd import _lib as L
    g = torch.Generator().manual_seed(11)
    n = 10_007
    radii = torch.randint(-2, 40, (n,), generator=g, dtype=torch.int32).to(gpu)
    vgrad = torch.randn(n, 3, generator=g).to(gpu)
    accum = torch.rand(n, 1, generator=g).to(gpu); denom = torch.randint(0, 5, (n, 1), generator=g).float().to(gpu)
    maxr = (torch.rand(n, generator=g) * 30).to(gpu)
    vis = radii > 0
    ea, ed, em = accum.clone(), denom.clone(), maxr.clone()
    ea[vis] += torch.norm(vgrad[vis, :2], dim=-1, keepdim=True)
    ed[vis] += 1
    em[vis] = torch.max(em[vis], radii[vis].to(em.dtype))
    L.check(L.load().syn3r_densification_stats(n, L.ptr(radii), L.ptr(vgrad), L.ptr(accum), L.ptr(denom), L.ptr(maxr),
                                               L.stream_ptr(gpu)), "densification_stats")
    assert torch.equal(denom, ed) and torch.equal(maxr, em)
    assert torch.allclose(accum, ea, rtol=2e-7, atol=0)          # torch.norm may fuse the two products differently: 1 ulp
