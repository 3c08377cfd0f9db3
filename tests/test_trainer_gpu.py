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
    assert abs(loss_h - float(loss_o)) < 1e-4
    for a, b in zip(gm.parameters(), P):
        # Adam's first step is lr * sign(grad): identical wherever the gradient sign is unambiguous
        d = (a.detach().cpu().double() - b.detach()).abs()
        assert (d > 1e-6).double().mean() < 2e-2


def test_training_loop_fits_a_view(gpu):
    from syn3r_amd.gs import Camera, GSTrainer, OptimizationParams
    N, H, W = 1500, 64, 96
    gt, K = make_scene(N, H, W, 11, gpu)
    cam0 = Camera.from_w2c(np.eye(4, dtype=np.float32), K, H, W, data_device=gpu)
    target = GSTrainer(gt, [cam0]).render_view(cam0)["render"].detach()
    gm, _ = make_scene(N, H, W, 12, gpu)            # different Gaussians
    cam = Camera.from_w2c(np.eye(4, dtype=np.float32), K, H, W, image=target, data_device=gpu)
    tr = GSTrainer(gm, [cam], OptimizationParams(iterations=150, position_lr=2e-3))
    first = tr.train_step(cam)
    last = tr.training(0, 0)
    assert last < 0.6 * first, (first, last)
    k, w2c = cam.get_calib_matrix_nerf()
    assert k.shape == (3, 3) and torch.allclose(w2c, torch.eye(4))
    tr.update_cameras([target], [np.eye(4, dtype=np.float32)], K, [0.05])
    assert len(tr.pseudo_cameras) == 1 and tr.pseudo_cameras[0].cam_confidence == 0.05
    tr.finetune(0, 1, iterations=3, pseudo_cam_sampling_rate=1.0)
