"""Device path of the orchestrator post-processing against the per-frame host restatement of
diffusionGS.py:1447-1483 / 821-862 (numpy + the cv2 semantics restated in `orchestrator.dilate5x5`)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _host_post(mask_reproj, warped, soft_in, h, w):
    """One frame of the reference loop body (diffusionGS.py:1447-1483) on numpy arrays."""
    from syn3r_amd import orchestrator as O
    mask = (1 - mask_reproj >= 0.5).astype(np.float64)
    mask = np.repeat(mask[:, :, None] * 255.0, 3, axis=2)
    ero = np.uint8(O.dilate5x5(mask)) / 255.0
    ero = (ero >= 0.5).astype(np.float64)
    wimg = warped.transpose([1, 2, 0])
    cond = np.asarray(np.uint8(wimg * (1 - ero)), dtype=np.float32) / 255.0
    pooled = O.block_mean_pool(np.mean(ero, axis=-1), h, w)
    soft = 1 - soft_in
    return dict(ero=ero[..., 0].astype(np.uint8), cond=cond, cond_ori=wimg / 255.0,
                masks=(pooled >= 0.2).astype(np.float32), soft=soft, soft_pool=O.block_mean_pool(soft, h, w))


@pytest.mark.parametrize("n,H,W,h,w", [(3, 48, 64, 6, 8), (2, 40, 72, 5, 9), (1, 576, 1024, 72, 128)])
def test_warp_post_matches_host_restatement(n, H, W, h, w):
    from syn3r_amd import _lib as L
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(n * 1000 + H)
    # sparse holes so the dilation matters, plus full rows/columns at the borders
    mr = torch.rand((n, H, W), generator=g) > 0.03
    mr[:, 0, :] = False
    mr[:, :, -1] = False
    mr[0, H // 2:H // 2 + 9, W // 3:W // 3 + 11] = False
    warped = torch.rand((n, 3, H, W), generator=g) * 255.0
    soft_in = torch.rand((n, H, W), generator=g)
    d = dict(mr=mr.to(dev), warped=warped.to(dev).contiguous(), soft=soft_in.to(dev).contiguous())
    new = lambda shape, dt=torch.float32: torch.empty(shape, dtype=dt, device=dev)
    ero, cond, cond_ori, soft = new((n, H, W), torch.uint8), new((n, H, W, 3)), new((n, H, W, 3)), new((n, H, W))
    masks, soft_pool = new((n, h, w)), new((n, h, w))
    rc = L.load().syn3r_warp_post(L.ptr(d["mr"]), L.ptr(d["warped"]), L.ptr(d["soft"]), n, H, W, h, w, L.ptr(ero), L.ptr(cond),
                                  L.ptr(cond_ori), L.ptr(soft), L.ptr(masks), L.ptr(soft_pool), L.stream_ptr(dev))
    L.check(rc, "syn3r_warp_post")
    torch.cuda.synchronize()
    for f in range(n):
        ref = _host_post(mr[f].numpy(), warped[f].numpy(), soft_in[f].numpy(), h, w)
        assert np.array_equal(ero[f].cpu().numpy(), ref["ero"])
        assert np.array_equal(masks[f].cpu().numpy(), ref["masks"])
        assert np.array_equal(cond[f].cpu().numpy(), ref["cond"])
        assert np.array_equal(cond_ori[f].cpu().numpy(), ref["cond_ori"])
        assert np.array_equal(soft[f].cpu().numpy(), ref["soft"])
        assert np.abs(soft_pool[f].cpu().numpy() - ref["soft_pool"]).max() <= 1.2e-7     # fp32 summation order


def test_warp_post_rejects_bad_pooling():
    from syn3r_amd import _lib as L
    dev = torch.device("cuda", 0)
    z = torch.zeros(16, device=dev)
    rc = L.load().syn3r_warp_post(L.ptr(z), L.ptr(z), L.ptr(z), 1, 50, 64, 6, 8, L.ptr(z), L.ptr(z), L.ptr(z), L.ptr(z), L.ptr(z),
                                  L.ptr(z), L.stream_ptr(dev))
    assert rc != 0 and b"bad shape" in L.load().syn3r_last_error()


def _scene(H=96, W=128, n_pose=9):
    from syn3r_amd import orchestrator as O
    K = np.array([[110.0, 0, W / 2], [0, 110.0, H / 2], [0, 0, 1]])
    p0 = np.eye(4)
    p1 = np.eye(4)
    p1[:3, 3] = [0.25, 0.02, 0.05]
    poses = list(O.pose_interpolation(p0, p1, num=n_pose))
    ys, xs = np.mgrid[0:H, 0:W]
    depth = lambda p: (2.0 + 0.4 * np.sin(xs / 17.0 + p[0, 3]) + 0.3 * np.cos(ys / 11.0)).astype(np.float32)
    rng = np.random.default_rng(7)
    img_l, img_r = rng.uniform(0, 255, (H, W, 3)), rng.uniform(0, 255, (H, W, 3))
    return K, poses, img_l, img_r, depth(p0), depth(p1), depth


def test_warp_images_bw_device_vs_per_frame_host_loop():
    """The batched device path equals the reference-shaped loop (one inverse_warp + numpy post-processing per frame)."""
    from syn3r_amd import orchestrator as O
    from syn3r_amd.solver_utils.forward_warp import inverse_warp
    dev = torch.device("cuda", 0)
    K, poses, img_l, img_r, dl, dr, depth = _scene()
    h, w = 12, 16
    d = O.warp_images_bw_device(K, poses, img_l, img_r, dl, dr, render_depth=depth, device="cuda:0", h=h, w=w)
    n = len(poses) - 2
    assert d["masks"].shape == (n, h, w) and d["cond_image"].shape == (n, 96, 128, 3)
    Kt = torch.tensor(K, device=dev, dtype=torch.float32)
    for i in range(n):
        side = (img_l, dl, poses[0]) if i < 12 else (img_r, dr, poses[-1])
        wd = inverse_warp(torch.tensor(side[0], device=dev, dtype=torch.float32).permute(2, 0, 1).contiguous(),
                          torch.tensor(side[1][None], device=dev, dtype=torch.float32),
                          torch.tensor(depth(poses[i + 1])[None], device=dev, dtype=torch.float32),
                          torch.tensor(side[2], device=dev, dtype=torch.float32),
                          torch.tensor(poses[i + 1], device=dev, dtype=torch.float32), Kt)
        ref = _host_post(wd["mask_reproj"].cpu().numpy(), wd["warped_img"].cpu().numpy(),
                         wd["soft_mask_reproj"].cpu().numpy(), h, w)
        assert np.array_equal(d["masks_ero"][i].cpu().numpy(), ref["ero"])
        assert np.array_equal(d["masks"][i].cpu().numpy(), ref["masks"])
        assert np.array_equal(d["cond_image"][i].cpu().numpy(), ref["cond"])
        assert np.array_equal(d["soft_masks_reproj_ori"][i].cpu().numpy(), ref["soft"])
    # the reference-shaped wrapper: types and shapes of diffusionGS.py:1507-1510
    il, ir, masks, cond, aux = O.warp_images_bw(K, poses, img_l, img_r, dl, dr, render_depth=depth, device="cuda:0", h=h, w=w)
    assert masks.dtype == torch.float64 and tuple(masks.shape) == (n, h, w) and len(cond) == n
    assert aux["masks_ero"].shape == (n, 96, 128, 3) and aux["masks_ero"].dtype == np.uint8
    assert np.array_equal(cond[0], d["cond_image"][0].cpu().numpy()) and il.max() <= 1.0


def test_fuse_uncertainty_device_vs_numpy():
    from syn3r_amd import orchestrator as O
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    n, H, W, h, w = 4, 48, 64, 6, 8
    warped = rng.uniform(0, 1, (n, H, W, 3)).astype(np.float32)
    warped[:, :5] = 0.0                                               # unknown pixels (sum == 0)
    gs = np.clip(warped + rng.normal(0, 0.25, warped.shape), -0.1, 1.1).astype(np.float32)
    soft = rng.uniform(0, 1, (n, H, W)).astype(np.float32)
    m_ref, c_ref, u_ref = O.fuse_uncertainty(warped, gs, soft, h=h, w=w)
    m, c, u = O.fuse_uncertainty_device(torch.tensor(warped, device=dev), gs, torch.tensor(soft, device=dev), h=h, w=w)
    u, c, m = u.cpu().numpy(), c.cpu().numpy(), m.cpu().numpy()
    assert np.abs(u - u_ref[..., 0]).max() < 2e-6
    assert np.abs(m - m_ref.numpy()).max() < 2e-6
    decided = np.abs(u_ref[..., 0] - 0.5) > 1e-5                      # away from the selection threshold
    assert decided.mean() > 0.99
    assert np.abs(c - np.stack(c_ref))[decided].max() == 0.0
    assert (u[:, :5] == 1.0).all()                                    # unknown pixels are fully uncertain
