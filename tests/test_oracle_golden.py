"""Pin the CPU oracle against golden vectors produced by the reference itself
(oracle/gen_golden.py).  CPU-only."""
import numpy as np
import pytest

from oracle import golden_inputs as GI
from oracle import scheduler_oracle as SO
from oracle import warp_oracle as WO


def frac_mismatch(a, b):
    return float(np.mean(a != b))


def assert_mostly_close(a, b, atol, rtol, max_frac=1e-3, hard=None):
    """All but `max_frac` of the elements within tolerance (fp32 chains amplify a rounding
    difference where a bilinear tap crosses the image border); optional hard bound on the rest."""
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    bad = d > (atol + rtol * np.abs(b))
    assert bad.mean() <= max_frac, f"{bad.mean():.2e} of elements out of tolerance, max diff {d.max():.3e}"
    if hard is not None:
        assert d.max() <= hard, d.max()


@pytest.mark.parametrize("name", list(GI.WARP_CASES))
def test_inverse_warp_oracle_matches_reference(name, golden_dir):
    c = GI.warp_case(name)
    g = np.load(golden_dir / f"warp_{name}.npz")
    sy, sx = c["stride"]
    o = WO.inverse_warp(c["img"], c["depth"], c["depth_pseudo"], c["pose1"], c["pose2"], c["K"], c["bandwidth"])
    err_ref = g["reproj_error"]
    err = o["reproj_error"][::sy, ::sx]
    fin = np.isfinite(err_ref)
    assert np.array_equal(fin, np.isfinite(err))
    # fp32 projection chains: tolerance 2e-3 px (+1e-4 relative)
    assert_mostly_close(err[fin], err_ref[fin], atol=2e-3, rtol=1e-4, hard=0.5)
    for k in ("warped_img", "warped_depth", "warped_masked_img"):
        a, b = o[k][..., ::sy, ::sx], g["iw_" + k]
        # nearest-neighbour picks may flip on exact half-pixel ties only
        assert frac_mismatch(a, b) < 2e-3, k
    np.testing.assert_allclose(o["soft_mask_reproj"][::sy, ::sx], g["iw_soft_mask_reproj"], atol=2e-4)
    for k in ("mask_warp", "mask_depth", "mask", "mask_inv", "mask_depth_strict", "mask_reproj"):
        assert frac_mismatch(o[k][::sy, ::sx], g["iw_" + k]) < 2e-3, k


@pytest.mark.parametrize("name", list(GI.WARP_CASES))
def test_forward_warp_oracle_matches_reference(name, golden_dir):
    c = GI.warp_case(name)
    g = np.load(golden_dir / f"warp_{name}.npz")
    sy, sx = c["stride"]
    frame = (c["img"].transpose(1, 2, 0) * 255.0).astype(np.float64)
    warped, mask2, flow = WO.forward_warp(frame, None, c["depth"].astype(np.float64), c["pose1"].astype(np.float64),
                                          c["pose2"].astype(np.float64), c["K"].astype(np.float64), None)
    np.testing.assert_allclose(flow[::sy, ::sx], g["fw_flow"], atol=1e-9, rtol=1e-12)
    assert np.array_equal(mask2[::sy, ::sx], g["fw_mask"])
    d = np.abs(warped[::sy, ::sx].astype(int) - g["fw_warped"].astype(int))
    assert d.max() <= 1 and np.mean(d > 0) < 1e-3


def test_nearby_consistency_oracle_matches_reference(golden_dir):
    """O4: oracle/orchestrator_oracle.nearby_consistency against the reference's own
    consistency_check_from_nearby_images_bw (model/diffusionGS.py:1300-1361) run on five seeded 576 x 1024 frames
    (oracle/gen_golden.py orch_nearby).  Tolerances are the inverse warp's (W2): soft masks 2e-4, nearest-neighbour picks
    may flip on exact half-pixel ties on < 0.2 % of the pixels."""
    from oracle import orchestrator_oracle as OO
    K, poses, images, depths = GI.orch_nearby_case()
    g = np.load(golden_dir / "orch_nearby.npz")
    sy, sx = GI.ORCH_NEARBY_STRIDE
    um, im = OO.nearby_consistency(K, poses, images, depths)
    assert len(um) == len(im) == 5
    for i in range(5):
        assert_mostly_close(um[i][::sy, ::sx], g["uncertainty"][i], atol=2e-4, rtol=0, max_frac=2e-3, hard=0.6)
        # the mean of nearest-sampled colours moves by a whole texel where the rounding of the sample position differs
        assert_mostly_close(im[i][::sy, ::sx], g["intensity_uncertainty"][i], atol=2e-3, rtol=0, max_frac=2e-3)
        assert abs(float(um[i].astype(np.float64).mean()) - g["uncertainty_mean"][i]) < 2e-4
        assert abs(float(im[i].astype(np.float64).mean()) - g["intensity_uncertainty_mean"][i]) < 1e-3


def test_uncertainty_fusion_oracle_matches_reference(golden_dir):
    """O5: oracle/orchestrator_oracle.fuse_uncertainty (and the product's numpy host form) against the reference's own statements
    (model/diffusionGS.py:821-867, executed on four seeded 576 x 1024 frames by oracle/gen_golden.py orch_fusion): pooled masks
    [n,72,128], the condition images and which pixels took the GS render."""
    from oracle import orchestrator_oracle as OO
    c = GI.orch_fusion_case()
    g = np.load(golden_dir / "orch_fusion.npz")
    sy, sx = GI.ORCH_NEARBY_STRIDE
    cond_ori, gs, soft = np.stack(c["cond_images_ori"]), np.stack(c["pseudo_images"][1:-1]), np.stack(c["soft_masks_reproj_ori"])
    masks, cond, unc = OO.fuse_uncertainty(cond_ori, gs, soft, 72, 128)
    np.testing.assert_allclose(masks, g["masks"], atol=2e-6)
    cond = np.stack(cond)
    np.testing.assert_allclose(cond[:, ::sy, ::sx], g["cond_image"], atol=1e-6)
    assert np.array_equal(np.all(cond == gs, axis=-1)[:, ::sy, ::sx], g["took_gs"])
    np.testing.assert_allclose(cond.astype(np.float64).mean(axis=(1, 2, 3)), g["cond_image_mean"], atol=1e-7)
    # the numpy host form the product keeps beside the kernel (no device needed)
    from syn3r_amd import orchestrator as O
    m2, c2, _ = O.fuse_uncertainty(cond_ori, gs, soft, 72, 128)
    np.testing.assert_allclose(m2.numpy(), g["masks"], atol=2e-6)
    assert np.array_equal(np.all(np.stack(c2) == gs, axis=-1)[:, ::sy, ::sx], g["took_gs"])


def test_sigma_schedule_matches_reference(golden_dir):
    g = np.load(golden_dir / "sched_sigmas.npz")
    s = GI.karras_sigmas(100)
    np.testing.assert_array_equal(s, g["sigmas"])
    assert g["sigmas"][0] == np.float32(700.0) and abs(float(g["timesteps"][0]) - 1.6378) < 1e-4


@pytest.mark.parametrize("name", list(GI.SCHED_CASES))
def test_scheduler_oracle_matches_reference(name, golden_dir):
    c = GI.sched_case(name)
    g = np.load(golden_dir / f"sched_{name}.npz")
    sig = np.load(golden_dir / "sched_sigmas.npz")["sigmas"]
    s = c["stride"]
    lam = c["lambda_ts"][c["step_i"]]
    half = c["model_output"].dtype == np.float16
    tol = dict(atol=2e-3, rtol=2e-3) if half else dict(atol=1e-5, rtol=1e-5)
    for cg in (True, False):
        o = SO.step_interp(c["model_output"], c["sample"], c["temp_cond"], c["mask"], lam, sig, c["step_i"], lr=0.02,
                           compute_grad=cg)
        tag = "g1" if cg else "g0"
        np.testing.assert_allclose(o["pred_original_sample"][..., ::s, ::s], g[f"interp_{tag}_x0"], atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(o["prev_sample"][..., ::s, ::s].astype(np.float32),
                                   g[f"interp_{tag}_prev"].astype(np.float32), **tol)
        if cg:
            a, b = o["grad"][..., ::s, ::s], g["interp_g1_grad"]
            # selection flips only at exact ties with the cutoff; everything else agrees to fp32 rounding
            bad = np.abs(a - b) > (1e-5 + 1e-4 * np.abs(b))
            assert bad.mean() < 1e-4, bad.mean()
    o = SO.step_interp_prob_uncertain(c["model_output"], c["sample"], c["temp_cond"], c["mask"], lam, sig, c["step_i"])
    bad = np.abs(o["pred_original_sample"][..., ::s, ::s] - g["replace_x0"]) > 1e-5
    assert bad.mean() < 1e-4
    np.testing.assert_allclose(o["prev_sample"][..., ::s, ::s].astype(np.float32), g["replace_prev"].astype(np.float32),
                               **tol)


def test_scheduler_mirror_schedule_matches_golden(golden_dir):
    """Host-side schedule of the product scheduler (no GPU needed).  sigmas are exact; timesteps go
    through the host libm's log (0.25*ln sigma), which differs by 1 ulp between CPU models."""
    from syn3r_amd.schedulers.scheduling_euler_discrete import SVD_XT_SCHEDULER_CONFIG, EulerDiscreteScheduler
    s = EulerDiscreteScheduler(**SVD_XT_SCHEDULER_CONFIG)
    s.set_timesteps(100)
    g = np.load(golden_dir / "sched_sigmas.npz")
    np.testing.assert_array_equal(s.sigmas.numpy(), g["sigmas"])
    np.testing.assert_allclose(s.timesteps.numpy(), g["timesteps"], rtol=3e-7, atol=0)
    assert np.float32(s.init_noise_sigma) == g["init_noise_sigma"]
    x = __import__("torch").ones(2, 3)
    np.testing.assert_allclose(s.scale_model_input(x, s.timesteps[4], step_i=4).numpy(),
                               (x / ((s.sigmas[4] ** 2 + 1) ** 0.5)).numpy())
    assert s.step_index == 4
    s.set_timesteps(25)
    g = np.load(golden_dir / "sched_sigmas25.npz")
    np.testing.assert_array_equal(s.sigmas.numpy(), g["sigmas"])
    np.testing.assert_allclose(s.timesteps.numpy(), g["timesteps"], rtol=3e-7, atol=0)
