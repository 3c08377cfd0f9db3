"""LPIPS (VGG16) loss term on the HIP path against the torch oracle (oracle/lpips_oracle.py): value and
gradient wrt the rendered image, shared seeded weights; PARITY UNPINNED (the `lpips` package and its weights are absent)."""
import numpy as np
import pytest
import torch

from oracle import lpips_oracle as LO

pytestmark = pytest.mark.gpu


def _images(H, W, seed):
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    base = torch.stack([0.5 + 0.4 * torch.sin(xs / 7 + c) * torch.cos(ys / 5 - c) for c in range(3)])
    a = (base + 0.08 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    b = (base.roll(2, 2) + 0.08 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    return a, b


@pytest.mark.parametrize("H,W", [(64, 96), (50, 70), (136, 240)])
def test_lpips_value_and_gradient_vs_oracle(gpu, H, W, measurements):
    from syn3r_amd.gs.lpips import LPIPS
    m = LPIPS().init_random(gpu, seed=3)
    sd = {}
    g = torch.Generator().manual_seed(3)                      # the same draws as init_random
    import math
    for k, shape in m.parameter_shapes().items():
        if k.startswith("lin"):
            sd[k] = torch.rand(shape, generator=g) * 0.2 + 0.01
        elif k.endswith(".bias"):
            sd[k] = 0.05 * torch.randn(shape, generator=g)
        else:
            sd[k] = torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[1] * 9))
    a, b = _images(H, W, H)
    pred = a.to(gpu).requires_grad_(True)
    target = b.to(gpu)
    loss = m(pred, target)
    (3.0 * loss).backward()
    ao = a.double().requires_grad_(True)
    ref = LO.lpips(ao, b.double(), sd)
    (3.0 * ref).backward()
    gh, go = pred.grad.double().cpu(), ao.grad
    rel = float((gh - go).norm() / go.norm())
    cos = float((gh * go).sum() / (gh.norm() * go.norm()))
    measurements(f"lpips:{H}x{W}", value_rel=abs(float(loss) - float(ref)) / abs(float(ref)), grad_rel=rel, grad_cos=cos)
    # measured (round 5, gpurun_out/test_measurements.jsonl): value 9e-7 .. 2.5e-5 relative, gradient 5.4-5.7e-2 relative, cosine
    # 0.9983-0.9986 (the gradient's error is the 13 fp16 backward-data convolutions, profiles/r04/lpips_layers.txt)
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-4 * abs(float(ref.detach())), (float(loss.detach()), float(ref.detach()))
    # The gradient bar is NOT the suite's "2x the measurement" rule, on purpose: 2 x 5.7e-2 would accept a gradient that is 11 % off, i.e.
    # say nothing.  The error is the fp16 rounding of the 13 backward-data convolutions' activations and gradients against a float64
    # chain (per-layer table: profiles/r04/lpips_layers.txt); the bars sit ~5 % above the worst recorded value (5.7e-2, cosine 0.9983),
    # so a change of accumulation order may trip them - re-measure (gpurun_out/test_measurements.jsonl) before moving them, and the LPIPS
    # weights are unpinned anyway.
    assert rel < 6e-2 and cos > 0.998, (rel, cos)              # fp16 activations / gradients against float64
    # a second call with the same target object reuses its cached features and gives the same number
    assert float(m(pred.detach(), target)) == float(loss)
    assert float(m(target, target)) < 1e-6 * abs(float(ref)) + 1e-9


def test_lpips_state_dict_names_and_errors(gpu):
    from syn3r_amd import _lib as L
    from syn3r_amd.gs.lpips import LPIPS
    m = LPIPS()
    names = m.parameter_shapes()
    assert names["net.slice1.0.weight"] == (64, 3, 3, 3) and names["net.slice5.28.bias"] == (512,) and names["lin3.model.1.weight"] == (1, 512, 1, 1)
    assert len(names) == 26 + 5
    with pytest.raises(L.Syn3rError):
        m(torch.zeros(3, 32, 32, device=gpu), torch.zeros(3, 32, 32, device=gpu))           # weights not loaded
    m.init_random(gpu)
    with pytest.raises(ValueError):
        m(torch.zeros(3, 8, 8, device=gpu), torch.zeros(3, 8, 8, device=gpu))               # too small for four poolings
    with pytest.raises(KeyError):
        LPIPS().load_state_dict({"net.slice1.0.weight": torch.zeros(64, 3, 3, 3)}, gpu)


def test_trainer_uses_lpips_when_switched_on(gpu, tmp_path):
    """`opt.use_lpips_loss` (raised by DiffusionGS.run around refine_GS) x `opt.lpips_weight`: the term enters train_step and
    evaluate() reports the metric; off -> the loss is the photometric one alone."""
    import math
    from oracle import raster_oracle as RO
    from syn3r_amd.gs import Camera, GaussianModel, GSTrainer, OptimizationParams
    from syn3r_amd.gs.lpips import LPIPS
    N, H, W = 600, 48, 64
    mm, s, q, o, sh = RO.synthetic_gaussians(N, seed=2, log_scale_mean=np.log(0.08))
    logit = torch.log(o.clamp(1e-3, 1 - 1e-3) / (1 - o.clamp(1e-3, 1 - 1e-3)))
    f = W / (2 * math.tan(math.radians(30)))
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=np.float32)
    gt = GSTrainer(GaussianModel(mm, torch.log(s), q, logit, sh, device=gpu), [Camera.from_w2c(np.eye(4, dtype=np.float32), K, H, W, data_device=gpu)])
    img = gt.render_view(gt.scene.getTrainCameras()[0])["render"].detach().clamp(0, 1)
    cam = Camera.from_w2c(np.eye(4, dtype=np.float32), K, H, W, image=img, data_device=gpu)
    gm = GaussianModel(mm + 0.02 * torch.randn_like(mm), torch.log(s), q, logit, sh, device=gpu)
    tr = GSTrainer(gm, [cam], OptimizationParams(iterations=3, lpips_weight=1.0))
    tr.lpips = LPIPS().init_random(gpu, seed=1)
    base = float(tr.train_step(cam))
    tr.opt.use_lpips_loss = True
    with_term = float(tr.train_step(cam))
    assert with_term > base * 1.02                            # the perceptual term is in the loss
    ev = tr.evaluate([cam])
    assert np.isfinite(ev["lpips"]) and ev["lpips"] > 0
    tr.lpips = None
    assert np.isnan(tr.evaluate([cam])["lpips"])


@pytest.mark.gpu
def test_lpips_layer_backward_saturates_instead_of_overflowing(gpu):
    """the layer backward multiplies by 1 / |a| and by the loss scale (~2.6e5 at relu5_3): a pixel whose feature norm
    is tiny but positive used to overflow the fp16 gradient to inf, which the backward convolutions then spread into the image
    gradient and Adam.  The kernel clamps to the fp16 range: every gradient is finite, the ordinary pixels are untouched."""
    from syn3r_amd import _lib as L
    lib = L.load()
    P, C = 256, 512
    g = torch.Generator().manual_seed(5)
    a = torch.rand(P, C, generator=g).to(torch.float16).to(gpu)
    b = torch.rand(P, C, generator=g).to(torch.float16).to(gpu)
    a[::16] = 0.0
    a[::16, 3] = 6e-5                                   # tiny but positive norm: 1 / |a| ~ 1.7e4
    a[8::16] = 0.0                                      # an all-zero pixel (norm 0 + 1e-10)
    w = torch.rand(C, generator=g).to(gpu)
    grad = torch.empty(P, C, dtype=torch.float16, device=gpu)
    L.check(lib.syn3r_lpips_layer_bwd_f16(L.ptr(a), L.ptr(b), L.ptr(w), P, C, 2.6e5, 0, L.ptr(grad), L.stream_ptr(gpu)), "lpips_layer_bwd")
    torch.cuda.synchronize()
    gf = grad.float()
    assert bool(torch.isfinite(gf).all())
    assert float(gf.abs().max()) <= 65504.0
    ordinary = gf[1::16].clone()
    assert float(ordinary.abs().max()) > 0.0
    # the same call at a scale no fp16 gradient survives: saturated at the largest finite fp16, never inf / NaN
    L.check(lib.syn3r_lpips_layer_bwd_f16(L.ptr(a), L.ptr(b), L.ptr(w), P, C, 1.0e12, 0, L.ptr(grad), L.stream_ptr(gpu)), "lpips_layer_bwd")
    torch.cuda.synchronize()
    gf = grad.float()
    assert bool(torch.isfinite(gf).all()) and float(gf.abs().max()) == 65504.0
