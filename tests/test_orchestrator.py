"""Host-side orchestrator numerics vs the reference's own methods (golden from model/diffusionGS.py)."""
import numpy as np
import torch

from oracle import golden_inputs as GI
from syn3r_amd import orchestrator as O


def test_pose_interpolation_and_dists_match_reference(golden_dir):
    g = np.load(golden_dir / "orchestrator.npz")
    for k, (a, b) in enumerate(GI.orch_pose_pairs()):
        poses = O.pose_interpolation(a, b)
        assert poses.shape == (25, 4, 4) and poses.dtype == np.float32
        np.testing.assert_allclose(poses, g[f"poses{k}"], atol=1e-6)
        np.testing.assert_allclose(poses[0], a, atol=1e-6)
        np.testing.assert_allclose(poses[-1], b, atol=1e-6)
        d, idx = O.compute_dists(poses)
        assert idx == int(g[f"idx{k}"])
        np.testing.assert_allclose(d, g[f"dists{k}"], atol=1e-6)


def test_search_hypers_v2_matches_reference(golden_dir):
    g = np.load(golden_dir / "orchestrator.npz")
    for k, m in enumerate(GI.orch_masks()):
        lam = O.search_hypers_v2(torch.from_numpy(m))
        assert lam.dtype == torch.float64 and lam.shape == (100, 25)
        np.testing.assert_array_equal(lam.numpy(), g[f"lambda{k}"])
        assert (lam[:, 0] == 1).all() and (lam[:, -1] == 1).all()


def test_mask_pooling_and_dilate():
    rng = np.random.default_rng(0)
    x = rng.random((576, 1024))
    p = O.block_mean_pool(x)
    assert p.shape == (72, 128)
    np.testing.assert_allclose(p[3, 5], x[24:32, 40:48].mean())
    m = np.zeros((20, 30)); m[10, 12] = 255.0; m[0, 0] = 255.0
    d = O.dilate5x5(m)
    assert d[8:13, 10:15].min() == 255.0 and d[7, 12] == 0 and d[2, 2] == 255.0 and d[3, 3] == 0


def test_fuse_uncertainty_shapes_and_limits():
    rng = np.random.default_rng(1)
    warped = rng.random((23, 576, 1024, 3)).astype(np.float32)
    warped[:, :50] = 0.0                                   # unwarped (black) band -> fully uncertain
    gs = warped.copy()
    soft = np.zeros((23, 576, 1024), np.float32)
    masks, cond, unc = O.fuse_uncertainty(warped, gs, soft)
    assert masks.shape == (23, 72, 128) and masks.dtype == torch.float32
    assert float(masks[:, :6].min()) == 1.0 and float(masks[:, 8:].max()) < 1e-6
    assert len(cond) == 23 and cond[0].shape == (576, 1024, 3)


def test_perturb_candidates_match_reference(golden_dir):
    """Same global-np.random draws as the reference method when seeded identically."""
    g = np.load(golden_dir / "orchestrator.npz")
    a, b = GI.orch_pose_pairs()[0]
    anchors = O.pose_interpolation(a, b)[::4]
    np.random.seed(1234)
    cands = np.array(O._perturb_interp_pose_candidates(anchors, perturb_num=5))
    assert cands.shape == (7, 6, 4, 4) and cands.dtype == np.float32
    np.testing.assert_allclose(cands, g["perturbed"], atol=1e-6)
    np.testing.assert_array_equal(cands[:, 0], anchors)          # candidate 0 is the unperturbed anchor


def test_host_helpers_match_independent_oracle():
    """O3/O5 host numerics (dilate5x5, block_mean_pool, fuse_uncertainty) against oracle/orchestrator_oracle.py:
    scipy.ndimage maximum filter / explicit block loops — a second implementation that shares no code with the product."""
    from oracle import orchestrator_oracle as OO
    rng = np.random.default_rng(5)
    for shape in ((48, 64), (40, 72, 3), (17, 23)):
        m = (rng.uniform(size=shape) > 0.93).astype(np.float64) * 255.0
        m[0] = 255.0
        m[:, -1] = 0.0
        assert np.array_equal(O.dilate5x5(m), OO.dilate5x5(m))
    x = rng.uniform(size=(48, 64)).astype(np.float32)
    assert np.allclose(O.block_mean_pool(x, 6, 8), OO.block_mean_pool(x, 6, 8), rtol=0, atol=1e-7)
    n, H, W, h, w = 3, 48, 64, 6, 8
    warped = rng.uniform(0, 1, (n, H, W, 3)).astype(np.float32)
    warped[:, :4] = 0.0
    gs = np.clip(warped + rng.normal(0, 0.25, warped.shape), -0.1, 1.1).astype(np.float32)
    soft = rng.uniform(0, 1, (n, H, W)).astype(np.float32)
    m1, c1, u1 = O.fuse_uncertainty(warped, gs, soft, h=h, w=w)
    m2, c2, u2 = OO.fuse_uncertainty(warped, gs, soft, h=h, w=w)
    assert np.abs(m1.numpy() - m2).max() < 1e-6 and np.array_equal(u1, u2)
    assert all(np.array_equal(a, b) for a, b in zip(c1, c2))


def test_oracle_resize_semantics():
    """cv2.resize restatements used by the oracle: INTER_LINEAR is pixel-centre aligned (identity at equal size, exact on
    linear ramps in the interior), INTER_NEAREST picks floor(dst * scale)."""
    from oracle import orchestrator_oracle as OO
    x = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert np.array_equal(OO.resize_linear(x, 3, 4), x)
    ramp = np.tile(np.arange(8, dtype=np.float32), (4, 1))
    up = OO.resize_linear(ramp, 4, 16)
    assert np.allclose(up[0, 1:-1], (np.arange(16)[1:-1] + 0.5) * 0.5 - 0.5)
    assert np.array_equal(OO.resize_nearest(ramp, 4, 16)[0], np.arange(16) // 2)
    assert np.array_equal(OO.resize_nearest(ramp, 2, 4)[0], np.array([0, 2, 4, 6], dtype=np.float32))


def test_knn_oracle_bruteforce_agrees_with_kdtree():
    """oracle/knn_oracle.py: the fp32 brute-force restatement (what the HIP kernel must match bit for bit) against
    scipy's k-d tree in float64 (independent algorithm), including duplicate points."""
    from oracle import knn_oracle as KO
    g = np.random.default_rng(0)
    p = g.random((1500, 3), dtype=np.float32)
    p[:50] = p[50:100]
    a, b = KO.mean_dist2_bruteforce(p), KO.mean_dist2_kdtree(p)
    assert np.allclose(a, b, rtol=2e-5, atol=1e-9)


def test_frames_to_gs_matches_pil():
    """diffusionGS.py:909-916 done with PIL itself: interior frames are ROUNDED to uint8 (tensor2vid / numpy_to_pil), the
    two replaced end frames are truncated, all are resized by PIL's default (bicubic) filter."""
    import PIL.Image
    from types import SimpleNamespace
    from syn3r_amd.diffusionGS import DiffusionGS
    rng = np.random.default_rng(3)
    frames = [rng.random((48, 64, 3), dtype=np.float32) for _ in range(4)]
    image_o, image_o2 = rng.random((48, 64, 3)), rng.random((48, 64, 3))
    for gh, gw in ((48, 64), (30, 40)):
        me = SimpleNamespace(gs_height=gh, gs_width=gw)
        got = DiffusionGS._frames_to_gs(me, list(frames), image_o, image_o2, True)
        pil = [PIL.Image.fromarray((f * 255).round().astype("uint8")) for f in frames]
        pil[0] = PIL.Image.fromarray((image_o * 255).astype(np.uint8))
        pil[-1] = PIL.Image.fromarray((image_o2 * 255).astype(np.uint8))
        ref = [torch.from_numpy(np.asarray(fr.resize((gw, gh)))).permute([2, 0, 1]) / 255. for fr in pil]
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
