"""Config 1 "plumbing": the minimal trainer drives the HIP rasteriser; one optimisation step equals the same step
taken with the CPU oracle rasteriser, and a short run fits a target view."""
import math

import numpy as np
import pytest
import torch

from oracle import raster_oracle as RO

pytestmark = pytest.mark.gpu


def make_scene(N, H, W, seed, dev):
    from syn3r_amd.gs import Camera, GaussianModel
    m, s, q, o, sh = RO.synthetic_gaussians(N, seed=seed, log_scale_mean=np.log(0.08))
    logit = torch.log(o.clamp(1e-3, 1 - 1e-3) / (1 - o.clamp(1e-3, 1 - 1e-3)))
    gm = GaussianModel(m, torch.log(s), q, logit, sh, device=dev)
    K = np.array([[W / (2 * math.tan(math.radians(30))), 0, W / 2], [0, W / (2 * math.tan(math.radians(30))), H / 2],
                  [0, 0, 1]], dtype=np.float32)
    return gm, K


def test_render_view_and_one_adam_step_match_oracle(gpu):
    from syn3r_amd.gs import Camera, GSTrainer, OptimizationParams
    N, H, W = 400, 40, 56
    gm, K = make_scene(N, H, W, 3, gpu)
    w2c = np.eye(4, dtype=np.float32)
    g = torch.Generator().manual_seed(1)
    target = torch.rand(3, H, W, generator=g)
    cam = Camera.from_w2c(w2c, K, H, W, image=target, data_device=gpu, cam_confidence=0.5)
    tr = GSTrainer(gm, [cam], OptimizationParams(iterations=1))
    out = tr.render_view(cam)
    assert set(("render", "depth", "alpha")) <= set(out) and out["render"].shape == (3, H, W)
    # the same step with the oracle rasteriser on the CPU
    P = [p.detach().cpu().double().clone().requires_grad_(True) for p in gm.parameters()]   # xyz, sh, opac, scale, rot
    oc, _, _, _, _ = RO.rasterize(P[0], torch.exp(P[3]), torch.nn.functional.normalize(P[4]), torch.sigmoid(P[2]), P[1],
                                  torch.ones(N, dtype=torch.float64), cam.world_view_transform.cpu().double(),
                                  cam.full_proj_transform.cpu().double(), cam.camera_center.cpu().double(),
                                  math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), H, W, torch.zeros(3, dtype=torch.float64), 3)
    np.testing.assert_allclose(out["render"].detach().cpu().numpy(), oc.detach().numpy(), atol=3e-4)
    from tests.test_train_ops_gpu import _published_ssim
    loss_o = 0.5 * (0.8 * (oc - target.double()).abs().mean() + 0.2 * (1.0 - _published_ssim(oc, target.double())))
    opt = torch.optim.Adam([{"params": [P[0]], "lr": 1.6e-4}, {"params": [P[1]], "lr": 2.5e-3}, {"params": [P[2]], "lr": 5e-2},
                            {"params": [P[3]], "lr": 5e-3}, {"params": [P[4]], "lr": 1e-3}], eps=1e-15)
    loss_o.backward()
    opt.step()
    loss_h = tr.train_step(cam)
    assert torch.is_tensor(loss_h) and loss_h.is_cuda          # no host synchronisation inside the step
    assert abs(float(loss_h) - float(loss_o)) < 1e-4
    for a, b in zip(gm.parameters(), P):
        # Adam's first step is lr * sign(grad): identical wherever the gradient sign is unambiguous
        d = (a.detach().cpu().double() - b.detach()).abs()
        assert (d > 1e-6).double().mean() < 2e-2


@pytest.mark.parametrize("lambda_dssim", [0.2, 0.0])
def test_explicit_step_equals_autograd_step(lambda_dssim, gpu):
    """The training loop's default step calls the activations (one HIP launch forward, one for the chain rule), the rasteriser
    and the loss directly - no autograd graph.  Its raw-parameter gradients equal autograd's through torch.exp / sigmoid /
    normalize + the same HIP operators, and so does the screen-space gradient the density control reads."""
    from syn3r_amd.gs import Camera, GSTrainer, OptimizationParams
    N, H, W = 3000, 72, 104
    w2c = np.eye(4, dtype=np.float32)
    target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(5))
    grads = {}
    for explicit in (False, True):
        gm, K = make_scene(N, H, W, 7, gpu)
        cam = Camera.from_w2c(w2c, K, H, W, image=target, data_device=gpu, cam_confidence=0.7)
        tr = GSTrainer(gm, [cam], OptimizationParams(iterations=1, lambda_dssim=lambda_dssim))
        if explicit:
            loss, out = tr._explicit_step(cam)
            vs = out["viewspace_grad"]
        else:
            out = tr.render_view(cam)
            from syn3r_amd.gs.train_ops import l1_loss, photometric_loss
            loss = (photometric_loss(out["render"], cam.original_image, lambda_dssim, 0.7) if lambda_dssim > 0
                    else l1_loss(out["render"], cam.original_image, weight=0.7))
            loss.backward()
            vs = out["viewspace_points"].grad
        grads[explicit] = [float(loss)] + [p.grad.detach().clone() for p in gm.parameters()] + [vs.detach().clone()]
    assert abs(grads[True][0] - grads[False][0]) < 1e-6
    for a, b in zip(grads[True][1:], grads[False][1:]):
        assert a.shape == b.shape
        scale = float(b.abs().max()) + 1e-20
        assert float((a - b).abs().max()) <= 2e-5 * scale, (float((a - b).abs().max()), scale)
    # and the public entry takes the explicit step by default, the autograd step on request: the same update
    finals = []
    for explicit in (None, False):
        gm, K = make_scene(N, H, W, 7, gpu)
        cam = Camera.from_w2c(w2c, K, H, W, image=target, data_device=gpu, cam_confidence=0.7)
        tr = GSTrainer(gm, [cam], OptimizationParams(iterations=1, lambda_dssim=lambda_dssim))
        tr.train_step(cam, explicit=explicit)
        finals.append([p.detach().clone() for p in gm.parameters()])
    for a, b in zip(*finals):
        assert ((a - b).abs() > 1e-6).double().mean() < 1e-3          # Adam's first step is lr * sign(grad)


@pytest.mark.parametrize("looks_away", [False, True])
def test_explicit_step_density_statistics_equal_autograd_path(looks_away, gpu):
    """With the density control on: the explicit step hands `add_densification_stats` the radii (its visible set is
    radii > 0, one HIP launch) where the autograd path hands it a boolean filter - the accumulators agree; and a view that sees
    no Gaussian at all (camera turned away: N_visible = 0) steps without error, with zero gradients and untouched statistics.
    A confidence tensor that asks for a gradient is refused by the explicit step (it discards that gradient)."""
    from syn3r_amd.gs import Camera, GSTrainer, OptimizationParams
    N, H, W = 2000, 64, 96
    w2c = np.eye(4, dtype=np.float32)
    if looks_away:
        w2c = np.diag([-1.0, 1.0, -1.0, 1.0]).astype(np.float32)      # half a turn about y: the cloud is behind the camera
    target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(9))
    stats = []
    for explicit in (True, False):
        gm, K = make_scene(N, H, W, 13, gpu)
        cam = Camera.from_w2c(w2c, K, H, W, image=target, data_device=gpu, cam_confidence=1.0)
        tr = GSTrainer(gm, [cam], OptimizationParams(iterations=10, lambda_dssim=0.2))
        tr.densify = True
        tr.train_step(cam, explicit=explicit)
        torch.cuda.synchronize()
        stats.append((gm.xyz_gradient_accum.clone(), gm.denom.clone(), gm.max_radii2D.clone(), [p.detach().clone() for p in gm.parameters()]))
    (a_acc, a_den, a_rad, a_par), (b_acc, b_den, b_rad, b_par) = stats
    assert torch.equal(a_den, b_den) and torch.equal(a_rad, b_rad)
    assert float((a_acc - b_acc).abs().max()) <= 2e-5 * (float(b_acc.abs().max()) + 1e-20)
    if looks_away:
        assert float(a_den.sum()) == 0.0 and float(a_acc.abs().sum()) == 0.0
    for x, y in zip(a_par, b_par):
        assert ((x - y).abs() > 1e-6).double().mean() < 1e-3
    gm, K = make_scene(N, H, W, 13, gpu)
    gm.confidence = torch.ones(N, device=gpu, requires_grad=True)
    cam = Camera.from_w2c(np.eye(4, dtype=np.float32), K, H, W, image=target, data_device=gpu)
    with pytest.raises(ValueError, match="confidence"):
        GSTrainer(gm, [cam], OptimizationParams(iterations=1))._explicit_step(cam)


def test_training_loop_fits_a_view(gpu):
    from syn3r_amd.gs import Camera, GSTrainer, OptimizationParams
    N, H, W = 1500, 64, 96
    gt, K = make_scene(N, H, W, 11, gpu)
    cam0 = Camera.from_w2c(np.eye(4, dtype=np.float32), K, H, W, data_device=gpu)
    target = GSTrainer(gt, [cam0]).render_view(cam0)["render"].detach()
    gm, _ = make_scene(N, H, W, 12, gpu)            # different Gaussians
    cam = Camera.from_w2c(np.eye(4, dtype=np.float32), K, H, W, image=target, data_device=gpu)
    tr = GSTrainer(gm, [cam], OptimizationParams(iterations=150, position_lr=2e-3))
    first = float(tr.train_step(cam))
    last = tr.training(0, 0)
    assert last < 0.6 * first, (first, last)
    k, w2c = cam.get_calib_matrix_nerf()
    assert k.shape == (3, 3) and torch.allclose(w2c, torch.eye(4))
    tr.update_cameras([target], [np.eye(4, dtype=np.float32)], K, [0.05])
    assert len(tr.pseudo_cameras) == 1 and tr.pseudo_cameras[0].cam_confidence == 0.05
    tr.finetune(0, 1, iterations=3, pseudo_cam_sampling_rate=1.0)


def test_checkpoints_pcd_reset_metrics_and_neighbours(gpu, tmp_path):
    """Trainer surface the orchestrator consumes beyond render/train (diffusionGS.py:1611-1625,1685-1687,475-477)."""
    from syn3r_amd.gs import Camera, GSTrainer, OptimizationParams
    from syn3r_amd.gs.train_ops import image_metrics
    N, H, W = 600, 48, 64
    gm, K = make_scene(N, H, W, 21, gpu)
    cam0 = Camera.from_w2c(np.eye(4, dtype=np.float32), K, H, W, data_device=gpu)
    target = GSTrainer(gm, [cam0]).render_view(cam0)["render"].detach().clamp(0, 1)
    p1 = np.eye(4, dtype=np.float32)
    p1[0, 3] = 0.2
    cams = [Camera.from_w2c(np.eye(4, dtype=np.float32), K, H, W, image=target, data_device=gpu),
            Camera.from_w2c(p1, K, H, W, image=target, data_device=gpu)]
    tr = GSTrainer(gm, cams, OptimizationParams(iterations=3), model_path=str(tmp_path), checkpoint_iterations=[3])
    tr.training(0, 0)
    assert sorted(p.name for p in tmp_path.iterdir()) == ["chkpnt3.pth", "chkpnt_latest.pth"]
    saved = gm._xyz.detach().clone()
    tr.training(0, 0, iterations=2)
    assert not torch.equal(saved, gm._xyz.detach())
    tr.load_checkpoint(str(tmp_path / "chkpnt3.pth"))
    assert torch.equal(saved, tr.gaussians._xyz.detach()) and tr.iteration == 3 and tr.optimizer.state == {}
    assert tr.finetune(0, 1, iterations=2) >= 0 and (tmp_path / "refine_1_chkpnt2.pth").exists()
    # metrics on the device vs their definitions
    g = torch.Generator().manual_seed(4)
    a, b = torch.rand(3, H, W, generator=g).to(gpu), torch.rand(3, H, W, generator=g).to(gpu)
    m = image_metrics(a, b)
    from tests.test_train_ops_gpu import _published_ssim
    assert abs(float(m[0]) - float(-10 * torch.log10(((a - b) ** 2).mean()))) < 1e-4
    assert abs(float(m[1]) - float(_published_ssim(a.cpu().double(), b.cpu().double()))) < 1e-5
    ev = tr.evaluate()
    assert ev["n"] == 2 and np.isfinite(ev["psnr"]) and 0 < ev["ssim"] <= 1 and np.isnan(ev["lpips"])
    # point-cloud re-initialisation (published create_from_pcd): replace, then append
    rng = np.random.default_rng(0)
    pts, col = rng.uniform(-1, 1, (500, 3)).astype(np.float32) + [0, 0, 4], rng.uniform(0, 1, (500, 3)).astype(np.float32)
    tr.reset_gaussians_from_pcd((pts, col), append_to_old_gaussians=False)
    g2 = tr.gaussians
    assert g2._xyz.shape == (500, 3) and g2._features.shape == (500, 16, 3) and g2.confidence.shape == (500,)
    d = np.sort(((pts[:, None] - pts[None]) ** 2).sum(-1), axis=1)[:, 1:4].mean(1)
    np.testing.assert_allclose(g2.get_scaling[:, 0].detach().cpu().numpy(), np.sqrt(d), rtol=2e-3)
    np.testing.assert_allclose(g2._features[:, 0].detach().cpu().numpy(), (col - 0.5) / 0.28209479177387814, atol=1e-5)
    assert float(g2.get_opacity[0]) == pytest.approx(0.1, abs=1e-6) and float(g2._features[:, 1:].abs().max()) == 0
    tr.reset_gaussians_from_pcd((pts[:100], col[:100]), append_to_old_gaussians=True)
    assert tr.gaussians._xyz.shape == (600, 3)
    out = tr.render_view(cams[0])
    assert torch.isfinite(out["render"]).all()
    tr.train_step(cams[0])                                   # the optimiser was rebuilt for the new tensors
    nn = tr.find_nearest_cam([cams[0]], cams, multi_view_max_angle=30, multi_view_min_dis=0.01, multi_view_max_dis=1.5)
    assert nn == [[1]] and cams[0].nearest_id == [1]


def test_training_with_density_control_changes_the_set_and_keeps_fitting(gpu):
    """training() with the adaptive density control on a compressed schedule: the Gaussian count changes (clone / split /
    prune on the device, optimiser moments carried over), the rasteriser re-sizes its pair buffer for every new count
    (first render of a new N is exact, then asynchronous again), and the loss still goes down."""
    from syn3r_amd.gs import Camera, GSTrainer, OptimizationParams
    N, H, W = 1500, 64, 96
    gt, K = make_scene(N, H, W, 11, gpu)
    cam0 = Camera.from_w2c(np.eye(4, dtype=np.float32), K, H, W, data_device=gpu)
    target = GSTrainer(gt, [cam0]).render_view(cam0)["render"].detach()
    gm, _ = make_scene(N, H, W, 12, gpu)
    cam = Camera.from_w2c(np.eye(4, dtype=np.float32), K, H, W, image=target, data_device=gpu)
    opt = OptimizationParams(iterations=120, position_lr=2e-3, densify_from_iter=10, densification_interval=20,
                             opacity_reset_interval=1000, densify_grad_threshold=3e-4, prune_min_opacity=0.02)
    tr = GSTrainer(gm, [cam], opt)
    first = float(tr.train_step(cam))
    counts = []
    orig = tr.densify_and_prune
    tr.densify_and_prune = lambda *a: counts.append(orig(*a)) or counts[-1]
    last = tr.training(0, 0)
    assert len(counts) == 6 and sum(c[0] + c[1] for c in counts) > 0 and sum(c[2] for c in counts) > 0, counts
    n_now = gm._xyz.shape[0]
    assert n_now != N and gm._features.shape[0] == n_now and gm.confidence.shape[0] == n_now
    for grp, p in zip(tr.optimizer.param_groups, gm.parameters()):
        assert grp["params"][0] is p and tr.optimizer.state[p]["exp_avg"].shape == p.shape
    assert np.isfinite(last) and last < 0.8 * first, (first, last)
    assert tr.truncated_renders == 0
    # six changes of the Gaussian count: the async pair capacity followed them (carried over, scaled by N_new / N_old) and
    # the keys of the counts that no longer exist are gone, every queued check has been read
    from syn3r_amd import raster
    assert not raster._pending and raster.capacity_key(gpu, n_now, H, W) in raster._capacity
    # the flag the orchestrator forwards (refine_GS, diffusionGS.py:1610,1640) switches it off
    counts.clear()
    tr.finetune(0, 1, iterations=40, disable_densification=True)
    assert counts == [] and gm._xyz.shape[0] == n_now


@pytest.mark.parametrize("with_conf", [False, True])
def test_raw_parameter_rasteriser_entries_equal_activate_then_rasterise(with_conf, gpu):
    """`syn3r_raster_preprocess_raw` / `syn3r_raster_backward_raw` (the activations and their chain rule inside the projection
    kernels: what the explicit training step calls) leave BIT FOR BIT what `syn3r_gaussian_activate` -> `syn3r_raster_preprocess`
    ... `syn3r_raster_backward` -> `syn3r_gaussian_activate_backward` leave in the image, depth, alpha and radii; the parameter
    gradients agree to the noise of the blend backward's floating-point atomics (their order differs from launch to launch:
    2e-5 of the largest entry, the bar of the explicit-vs-autograd test above), culled rows are exactly zero on both routes."""
    from syn3r_amd import _lib as L
    from syn3r_amd.gs import Camera
    from syn3r_amd.raster import GaussianRasterizationSettings, rasterize_backward, rasterize_forward
    N, H, W = 5000, 88, 120
    gm, K = make_scene(N, H, W, 11, gpu)
    w2c = np.eye(4, dtype=np.float32)
    w2c[2, 3] = -3.0                                  # depths 2..6 -> -1..3: a quarter of the Gaussians are behind the near plane (culled rows on both routes)
    cam = Camera.from_w2c(w2c, K, H, W, image=torch.rand(3, H, W), data_device=gpu)
    conf = torch.rand(N, device=gpu) if with_conf else None
    st = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.zeros(3, device=gpu), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center, prefiltered=False, debug=False)
    lib, stream = L.load(), L.stream_ptr(gpu)
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=gpu)
    g_color = torch.randn(3, H, W, device=gpu)
    g_depth, g_alpha = torch.randn(1, H, W, device=gpu), torch.randn(1, H, W, device=gpu)
    with torch.no_grad():
        # two launches around the activated-tensor entries
        sc, ro, op = new(N, 3), new(N, 4), new(N, 1)
        L.check(lib.syn3r_gaussian_activate(N, L.ptr(gm._scaling), L.ptr(gm._rotation), L.ptr(gm._opacity), L.ptr(sc), L.ptr(ro),
                                            L.ptr(op), stream), "gaussian_activate")
        *out0, s0 = rasterize_forward(gm._xyz, gm._features, op, sc, ro, conf, st)
        d_m3, d_m2, d_sh, d_op, d_sc, d_ro, d_cf = rasterize_backward(s0, g_color, g_depth, g_alpha)
        d_ls, d_rr, d_lg = new(N, 3), new(N, 4), new(N, 1)
        L.check(lib.syn3r_gaussian_activate_backward(N, L.ptr(gm._rotation), L.ptr(sc), L.ptr(ro), L.ptr(op), L.ptr(d_sc),
                                                     L.ptr(d_ro), L.ptr(d_op), L.ptr(d_ls), L.ptr(d_rr), L.ptr(d_lg), stream),
                "gaussian_activate_backward")
        ref = [d_m3, d_m2, d_sh, d_lg.reshape(-1), d_ls, d_rr] + ([d_cf] if with_conf else [])
        # the raw-parameter entries
        *out1, s1 = rasterize_forward(gm._xyz, gm._features, gm._opacity, gm._scaling, gm._rotation, conf, st, raw_params=True)
        r_m3, r_m2, r_sh, r_lg, r_ls, r_rr, r_cf = rasterize_backward(s1, g_color, g_depth, g_alpha)
        got = [r_m3, r_m2, r_sh, r_lg.reshape(-1), r_ls, r_rr] + ([r_cf] if with_conf else [])
    assert int((out0[1] > 0).sum()) > N // 4 and int((out0[1] == 0).sum()) > 0          # visible and culled Gaussians both
    for a, b in zip(out0, out1):
        assert torch.equal(a, b)
    culled = out0[1] == 0
    for name, a, b in zip(("xyz", "means2D", "sh", "opacity logit", "log scale", "raw rotation", "confidence"), ref, got):
        assert a.shape == b.shape, name
        scale = float(a.abs().max()) + 1e-20
        assert float((a - b).abs().max()) <= 2e-5 * scale, (name, float((a - b).abs().max()), scale)
        assert float(a[culled].abs().max()) == 0.0 and float(b[culled].abs().max()) == 0.0, name
    assert float(d_ls.abs().max()) > 0 and float(d_rr.abs().max()) > 0 and float(d_lg.abs().max()) > 0
