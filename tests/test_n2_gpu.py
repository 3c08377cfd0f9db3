"""Point-cloud densification slice on the GPU: the device-side cloud filter and the flow cycle mask
against `oracle/pcd_oracle.py`, and BASELINE config 5 end to end — `DiffusionGS(num_input_views=9,
"2PassProbUncertainPost", fps_keyframe_sampling=1).run(1)` with `num_views_for_pcd_densification=4` — with the two absent
networks (GMFlow, dust3r) injected as stand-ins on the trainer, where the reference keeps them."""
import math
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import pcd_oracle as PO
from oracle import pipeline_mocks as PM
from oracle import raster_oracle as RO

pytestmark = pytest.mark.gpu


def _cloud(n, seed, strays=40):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-2, 2, (6, 3))
    pts = centres[rng.integers(0, 6, n)] + 0.25 * rng.standard_normal((n, 3)) * rng.uniform(0.3, 1.5, (n, 1))
    pts[rng.choice(n, strays, replace=False)] += rng.uniform(5, 9, (strays, 3))
    return pts


@pytest.mark.parametrize("n,seed", [(30000, 0), (4097, 1), (13, 2)])
def test_statistical_outlier_vs_oracle(gpu, n, seed):
    """avg distance (k = 20, self included, float64) bit for bit against the k-d tree oracle; mean / std / threshold to
    float64 rounding of a different summation order; the SAME inliers."""
    from syn3r_amd import pcd as P
    pts = _cloud(n, seed, strays=min(40, n // 4))
    keep, avg, stats = P.statistical_outlier(torch.from_numpy(pts).to(gpu), 20, 3.0)
    ind, oavg, (mean, std, thr) = PO.remove_statistical_outlier(pts, 20, 3.0)
    np.testing.assert_array_equal(avg.cpu().numpy(), oavg)
    np.testing.assert_allclose(stats.cpu().numpy(), [mean, std, thr, float(n)], rtol=1e-12)
    np.testing.assert_array_equal(np.nonzero(keep.cpu().numpy())[0], ind)
    assert 0 < len(ind) < n or n < 50


def test_statistical_outlier_with_coincident_points(gpu):
    """More than nb_neighbors coincident points have average neighbour distance 0: open3d leaves them out of the SUMS but keeps
    them in the divisors (`valid_distances` counts every point whose query returned neighbours), and drops them from the
    inliers (0 < avg is required).  Kernel and oracle follow that rule: the statistics divide by n, not by the positive count."""
    from syn3r_amd import pcd as P
    pts = _cloud(4000, 5, strays=20)
    pts[100:130] = pts[100]                                    # 30 copies of one point
    keep, avg, stats = P.statistical_outlier(torch.from_numpy(pts).to(gpu), 20, 3.0)
    ind, oavg, (mean, std, thr) = PO.remove_statistical_outlier(pts, 20, 3.0)
    a = avg.cpu().numpy()
    assert (a[100:130] == 0).all() and (oavg[100:130] == 0).all()
    pos = oavg > 0
    assert abs(mean - oavg[pos].sum() / 4000) < 1e-12 and mean < oavg[pos].mean()        # divisor n, not the positive count
    np.testing.assert_allclose(stats.cpu().numpy(), [mean, std, thr, 4000.0], rtol=1e-12)
    np.testing.assert_array_equal(np.nonzero(keep.cpu().numpy())[0], ind)
    assert not set(range(100, 130)) & set(ind.tolist())


def test_filter_dense_cloud_vs_oracle_and_ply(gpu, tmp_path):
    """model/diffusionGS.py:314-336 at the reference's size: 230 000 dust3r-like points -> stride 2 -> outlier removal;
    the written .ply holds the inliers."""
    from syn3r_amd import pcd as P
    n = 230_000
    pts = _cloud(n, 7, strays=300).astype(np.float32)                 # dust3r vertices are float32
    rgba = np.random.default_rng(8).integers(0, 256, (n, 4), dtype=np.uint8)
    got = P.filter_dense_cloud(pts, rgba, gpu)
    op, oc = PO.filter_dense_cloud(pts, rgba)
    np.testing.assert_array_equal(got.points.cpu().numpy(), op)
    np.testing.assert_array_equal(got.colors.cpu().numpy(), oc)
    assert 100_000 < len(got) < 115_000
    path = tmp_path / "dense.ply"
    P.write_point_cloud(str(path), got)
    rp, rc = PO.read_ply(str(path))
    np.testing.assert_array_equal(rp, op)
    np.testing.assert_array_equal(rc, np.rint(oc * 255).astype(np.uint8))
    with pytest.raises(ValueError):                                   # fewer than 100 000 points: open3d rejects k = 0 too
        P.filter_dense_cloud(pts[:5000], rgba[:5000], gpu)
    from syn3r_amd import _lib as L
    with pytest.raises(L.Syn3rError):
        P.statistical_outlier(torch.from_numpy(pts[:100].astype(np.float64)).to(gpu), 10, 3.0)


@pytest.mark.parametrize("H,W,seed", [(72, 128, 0), (135, 240, 1), (5, 7, 2)])
def test_flow_cycle_mask_vs_oracle(gpu, H, W, seed):
    from syn3r_amd import pcd as P
    rng = np.random.default_rng(seed)
    n = 3
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    fw = np.stack([np.stack([4 * np.sin(xs / 17 + b) + rng.standard_normal((H, W)), 3 * np.cos(ys / 11 - b) + rng.standard_normal((H, W))])
                   for b in range(n)]).astype(np.float32)
    bw = (-fw + 1.5 * rng.standard_normal(fw.shape) * (rng.random((n, 1, H, W)) < 0.5)).astype(np.float32)
    mask, dist = P.flow_cycle_mask(torch.from_numpy(fw).to(gpu), torch.from_numpy(bw).to(gpu), 3.0, want_dist=True)
    for b in range(n):
        om, od = PO.flow_cycle_mask(fw[b], bw[b], 3.0)
        fin = np.isfinite(od)
        np.testing.assert_array_equal(np.isfinite(dist[b].cpu().numpy()), fin)
        np.testing.assert_allclose(dist[b].cpu().numpy()[fin], od[fin], rtol=2e-6, atol=2e-6)
        near = fin & (np.abs(od - 3.0) < 1e-4)                       # the threshold itself: rounding may fall either way
        np.testing.assert_array_equal(mask[b].cpu().numpy()[~near], om[~near])
    assert 0.05 < float(mask.mean()) < 0.95


# ------------------------------------------------------------------------------------------------ config 5 end to end
def _scene(gpu, tmp_path, V=9, iterations=6):
    from syn3r_amd.gs import Camera, GaussianModel, GSTrainer, OptimizationParams
    N, H, W = 1500, 72, 128
    m, s, q, o, sh = RO.synthetic_gaussians(N, seed=11, log_scale_mean=np.log(0.08))
    logit = torch.log(o.clamp(1e-3, 1 - 1e-3) / (1 - o.clamp(1e-3, 1 - 1e-3)))
    gt = GaussianModel(m, torch.log(s), q, logit, sh, device=gpu)
    f = W / (2 * math.tan(math.radians(30)))
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=np.float32)
    poses = []
    for v in range(V):
        p = np.eye(4, dtype=np.float32)
        p[0, 3] = -0.3 + 0.6 * v / (V - 1) + 0.01 * ((v * 7) % 3)
        p[2, 3] = 0.02 * math.sin(v)
        poses.append(p)
    tr_gt = GSTrainer(gt, [Camera.from_w2c(poses[0], K, H, W, data_device=gpu)])
    views = [tr_gt.render_view(Camera.from_w2c(p, K, H, W, data_device=gpu))["render"].detach() for p in poses]
    cams = [Camera.from_w2c(p, K, H, W, image=v, data_device=gpu) for p, v in zip(poses, views)]
    gm = GaussianModel(m + 0.01 * torch.randn_like(m), torch.log(s), q, logit, sh, device=gpu)
    trainer = GSTrainer(gm, cams, OptimizationParams(iterations=iterations), model_path=str(tmp_path / "model"),
                        checkpoint_iterations=[iterations])
    args = SimpleNamespace(cam_confidence=0.2, pseudo_cam_sampling_rate=0.5, fps_keyframe_sampling=1,
                           densify_type="interpolate_gs_v2", num_views_for_pcd_densification=4)
    return trainer, args, m.numpy()


class _FlowNet:
    """GMFlow's role: a -> b flow [2,H,W].  Zero flow, except that frames whose mean differs a lot get an inconsistent one."""
    def __init__(self):
        self.calls = 0

    def __call__(self, a, b):
        self.calls += 1
        return torch.zeros((2,) + tuple(a.shape[-2:]), device=a.device)


class _Dust3r:
    """dust3r's role: (frames, c2w poses, intrinsics, pairs) -> a trimesh-like scene holding one coloured cloud."""
    def __init__(self, centres):
        self.centres, self.moves, self.runs = centres, [], []

    def to(self, dev):
        self.moves.append(dev)

    def run(self, frames, c2w_poses, intrinsics, preset_pairs):
        self.runs.append(dict(n=len(frames), pairs=list(preset_pairs), K=np.array(intrinsics), c2w=np.array(c2w_poses)))
        rng = np.random.default_rng(5)
        n = 150_000
        v = (self.centres[rng.integers(0, len(self.centres), n)] + 0.03 * rng.standard_normal((n, 3))).astype(np.float32)
        v[rng.choice(n, 200, replace=False)] += rng.uniform(3, 6, (200, 3)).astype(np.float32)
        col = rng.integers(0, 256, (n, 4), dtype=np.uint8)
        return None, SimpleNamespace(geometry={"geometry_0": SimpleNamespace(vertices=v, colors=col)})


def test_config5_nine_views_post_with_pcd_densification(gpu, tmp_path):
    """BASELINE config 5: DL3DV-style 9 views, Post pipeline, FPS key frames, dust3r / GMFlow densification (stand-ins)."""
    from syn3r_amd.diffusionGS import DiffusionGS
    trainer, args, centres = _scene(gpu, tmp_path)
    trainer.flow_net, trainer.dust3r = _FlowNet(), _Dust3r(centres)
    comps = dict(vae=PM.MockVAE(), image_encoder=PM.MockImageEncoder(), unet=PM.MockUNet().to(gpu), dtype=torch.float32)
    d = DiffusionGS(trainer, num_input_views=9, save_dir=str(tmp_path), diffusion_type="2PassProbUncertainPost",
                    interp_type="backward_warp", input_args=args, svd_components=comps, num_inference_steps=2)
    assert d.dust3r is trainer.dust3r and d.fps_keyframe_sampling == 1
    np.random.seed(3)
    seen = {}
    orig = trainer.finetune

    def finetune(*a, **k):
        seen.update(n=trainer.gaussians._xyz.shape[0], pseudo=len(trainer.pseudo_cameras), lpips=trainer.opt.use_lpips_loss)
        return orig(*a, **k)

    trainer.finetune = finetune
    d.run(refine_cycles=1)
    # 9 closed-loop pairs x 3 key frames each (4 picked, the pair's last one dropped) go through the correspondence test
    run = trainer.dust3r.runs[0]
    assert len(trainer.dust3r.runs) == 1 and trainer.flow_net.calls == 2 * 27 and run["n"] == 27
    assert trainer.dust3r.moves == ["cuda", "cpu"]
    assert len(run["pairs"]) == 27 * 26 // 2 and run["pairs"][0] == (0, 1)              # complete graph over the kept key frames
    K0 = d.gs_intrinsics.copy()
    K0[:2] *= 512 / d.gs_width
    np.testing.assert_allclose(run["K"][0], K0)
    np.testing.assert_allclose(run["c2w"][0], np.linalg.inv(trainer.scene.getTrainCameras()[0].get_calib_matrix_nerf()[1].numpy()), atol=1e-6)
    # the filtered cloud re-initialised the Gaussians (cycle 0: not appended) and was written where the reference writes it
    ply = tmp_path / "dense_views" / "dense_views_cyc0.ply"
    pts, _ = PO.read_ply(str(ply))
    assert 140_000 < pts.shape[0] < 150_000 and seen["n"] == pts.shape[0]
    lo, hi = centres.min(0) - 1.0, centres.max(0) + 1.0
    assert ((pts >= lo) & (pts <= hi)).all() and pts.shape[0] >= 150_000 - 260          # the 200 strays are gone, little else
    assert seen["pseudo"] == 9 * 24 and seen["lpips"] is True and trainer.truncated_renders == 0
    files = sorted(p.name for p in tmp_path.iterdir() if p.suffix == ".pt")
    assert files == [f"dense_viewsinterpolated_dense_views_cyc0_view{i}.pt" for i in range(9)]
    assert (tmp_path / "model" / "refine_0_chkpnt6.pth").exists()


def test_generate_corresp_mask_surface(gpu, tmp_path):
    """the trainer method the orchestrator calls (diffusionGS.py:377): (masks, flows), masks[0][0] a [H,W] map"""
    trainer, _, _ = _scene(gpu, tmp_path, V=3, iterations=1)
    a, b = torch.rand(3, 72, 128), torch.rand(3, 72, 128) * 255
    with pytest.raises(RuntimeError):
        trainer.generate_corresp_mask([a], [b], dist_thresh=3, desc_only=False)

    def net(x, y):                                  # forward flow +4 px in x, backward flow -4 px except on the right half
        f = torch.zeros(2, 72, 128, device=x.device)
        f[0] = 4.0 if float(x.max()) <= 1.0 else -4.0
        if float(x.max()) > 1.0:
            f[0, :, 64:] = 1.0
        return f

    trainer.flow_net = net
    masks, flows = trainer.generate_corresp_mask([a], [b], dist_thresh=3, desc_only=False)
    m = masks[0][0]
    assert tuple(m.shape) == (72, 128) and len(flows) == 1
    assert bool(m[:, :60].all()) and not bool(m[:, 60:124].any()) and not bool(m[:, 124:].any())
    with pytest.raises(NotImplementedError):
        trainer.generate_corresp_mask([a], [b], dist_thresh=3, desc_only=True)
