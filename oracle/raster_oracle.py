"""CPU restatement of the 3D Gaussian Splatting rasteriser with depth / alpha outputs and a
per-Gaussian confidence factor (test oracle, not product code).

PARITY UNPINNED: the reference's rasteriser (diff-gaussian-rasterization-confidence, inside the
un-vendored submodule thirdparty/FSGS -> DecaYale/FSGS_dev@dev, pinned SHA unknown; call sites
model/diffusionGS.py:154,166,139,1640) is not in /root/reference and the reference holds no test
or golden vector for it.  This file restates the PUBLISHED algorithm (Kerbl et al., SIGGRAPH 2023,
"3D Gaussian Splatting for Real-Time Radiance Field Rendering": EWA projection with the 0.3 px
low-pass, 3-sigma tile binning on 16x16 tiles, (tile<<32 | depth) key sort, front-to-back blend
with alpha < 1/255 skip, alpha <= 0.99 clamp, stop at T < 1e-4).

Written in differentiable torch (float64 by default) so that autograd through this file is the
checker for the hand-written HIP backward.  The 0.99 clamp and the 1.3*tan(fov) guard pass
gradients the way the published backward does (straight-through / constant).
"""
from __future__ import annotations

import numpy as np
import torch

TILE = 16
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def eval_sh(deg, sh, dirs):
    """sh [N,M,3], dirs [N,3] unit -> [N,3] (before the +0.5 and clamp)."""
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


def preprocess(means3D, scales, rotations, opacities, shs, confidence, viewmatrix, projmatrix, campos, tanfovx,
               tanfovy, H, W, sh_degree, scale_modifier=1.0):
    """Per-Gaussian projection.  viewmatrix / projmatrix: [4,4] tensors holding the TRANSPOSED
    matrices (FSGS camera convention), i.e. p_view = [p,1] @ viewmatrix."""
    dt = means3D.dtype
    N = means3D.shape[0]
    V = viewmatrix.to(dt)
    Pm = projmatrix.to(dt)
    ones = torch.ones(N, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    t = (ph @ V)[:, :3]
    hom = ph @ Pm
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None]
    in_front = t[:, 2] > 0.2
    tz = torch.where(in_front, t[:, 2], torch.ones_like(t[:, 2]))
    R = quat_to_rot(rotations)
    Mm = R * (scale_modifier * scales)[:, None, :]
    Sigma = Mm @ Mm.transpose(1, 2)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = t[:, 0] / tz, t[:, 1] / tz
    # outside the guard band the published backward treats the clamped t.x as a constant
    tx = torch.where((txtz < -limx) | (txtz > limx), (txtz.clamp(-limx, limx) * tz).detach(), t[:, 0])
    ty = torch.where((tytz < -limy) | (tytz > limy), (tytz.clamp(-limy, limy) * tz).detach(), t[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], 1).reshape(N, 2, 3)
    Wm = V[:3, :3].T
    T = J @ Wm
    cov2 = T @ Sigma @ T.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    det_safe = torch.where(det == 0, torch.ones_like(det), det)
    conic = torch.stack([c / det_safe, -b / det_safe, a / det_safe], 1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    def tl(v, g):
        return torch.clamp(torch.trunc(v / TILE), 0, g).to(torch.int64)

    x0, y0 = tl(px.detach() - radius, gx), tl(py.detach() - radius, gy)
    x1, y1 = tl(px.detach() + radius + TILE - 1, gx), tl(py.detach() + radius + TILE - 1, gy)
    area = (x1 - x0) * (y1 - y0)
    valid = in_front & (det != 0) & (area > 0)
    dirs = means3D - campos.to(dt)[None]
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    rgb_raw = eval_sh(sh_degree, shs, dirs) + 0.5
    rgb = torch.clamp(rgb_raw, min=0.0)
    conf = confidence if confidence is not None else torch.ones(N, dtype=dt)
    return dict(depth=t[:, 2], px=px, py=py, conic=conic, rgb=rgb, clamped=(rgb_raw < 0), opacity=opacities * conf,
                radius=torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int64), rect=(x0, y0, x1, y1),
                valid=valid, tiles_touched=torch.where(valid, area, torch.zeros_like(area)), grid=(gx, gy))


def build_tile_lists(pre):
    """(tile<<32 | float32 depth bits) keys in duplication order, stable sort -> (keys, point_list, ranges)."""
    gx, gy = pre["grid"]
    x0, y0, x1, y1 = [v.numpy() for v in pre["rect"]]
    valid = pre["valid"].numpy()
    dbits = pre["depth"].detach().to(torch.float32).numpy().view(np.uint32).astype(np.uint64)
    ids = np.nonzero(valid)[0]
    wd = (x1 - x0)[ids].astype(np.int64)
    area = wd * (y1 - y0)[ids].astype(np.int64)
    total = int(area.sum())
    rep = np.repeat(np.arange(len(ids)), area)                      # duplication order = Gaussian index order
    start = np.repeat(np.cumsum(area) - area, area)
    off = np.arange(total, dtype=np.int64) - start                  # row-major inside the tile rectangle
    ty = y0[ids][rep] + off // wd[rep]
    tx = x0[ids][rep] + off % wd[rep]
    keys = ((ty * gx + tx).astype(np.uint64) << np.uint64(32)) | dbits[ids][rep]
    vals = ids[rep].astype(np.int64)
    order = np.argsort(keys, kind="stable")
    keys, vals = keys[order], vals[order]
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    tid = np.arange(gx * gy)
    ranges = np.stack([np.searchsorted(tiles, tid, "left"), np.searchsorted(tiles, tid, "right")], 1).astype(np.int64)
    ranges[ranges[:, 0] == ranges[:, 1]] = 0
    return keys, vals, ranges


def blend_tile(px, py, conic, rgb, zdepth, opacity, pxf, pyf, bg):
    """Front-to-back blend of one tile's depth-sorted splats (rows) over its pixels (columns).
    Returns colour [npix,3] (background composited), depth [npix], alpha [npix], n_contrib [npix]."""
    dt = px.dtype
    dx = px[:, None] - pxf[None]
    dy = py[:, None] - pyf[None]
    power = -0.5 * (conic[:, 0:1] * dx * dx + conic[:, 2:3] * dy * dy) - conic[:, 1:2] * dx * dy
    raw = opacity[:, None] * torch.exp(power)
    alpha = raw - torch.clamp(raw - 0.99, min=0).detach()     # min(0.99, raw), straight-through
    ok = (power <= 0) & (alpha >= 1.0 / 255.0)
    a_eff = torch.where(ok, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - a_eff
    T_incl = torch.cumprod(one_m, 0)
    T_excl = torch.cat([torch.ones(1, T_incl.shape[1], dtype=dt), T_incl[:-1]], 0)
    stop = ok & (T_incl < 1e-4)
    stopped = torch.cummax(stop.to(torch.int8), 0)[0].bool()
    live = ok & ~stopped
    w = torch.where(live, a_eff * T_excl, torch.zeros_like(a_eff))
    T_final = torch.prod(torch.where(live, one_m, torch.ones_like(one_m)), 0)
    c = (w[:, :, None] * rgb[:, None, :]).sum(0) + T_final[:, None] * bg.to(dt)[None]
    d = (w * zdepth[:, None]).sum(0)
    idx = torch.arange(1, live.shape[0] + 1)[:, None] * live
    return c, d, 1.0 - T_final, idx.max(0)[0]


BLEND_KEYS = ("px", "py", "conic", "rgb", "depth", "opacity")


def _tile_pixels(ty, tx, H, W, dt):
    ys = torch.arange(ty * TILE, min((ty + 1) * TILE, H))
    xs = torch.arange(tx * TILE, min((tx + 1) * TILE, W))
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    return len(ys), len(xs), xx.reshape(-1).to(dt), yy.reshape(-1).to(dt)


def render(pre, point_list, ranges, bg, H, W):
    dt = pre["px"].dtype
    gx, gy = pre["grid"]
    color = torch.zeros(3, H, W, dtype=dt) + bg.to(dt)[:, None, None]
    depth = torch.zeros(1, H, W, dtype=dt)
    alpha_out = torch.zeros(1, H, W, dtype=dt)
    n_contrib = np.zeros((H, W), dtype=np.int64)
    for ty in range(gy):
        for tx in range(gx):
            s, e = ranges[ty * gx + tx]
            hh, ww, pxf, pyf = _tile_pixels(ty, tx, H, W, dt)
            if e <= s or hh == 0 or ww == 0:
                continue
            ids = torch.from_numpy(point_list[s:e])
            c, d, a, nc = blend_tile(*[pre[k][ids] for k in BLEND_KEYS], pxf, pyf, bg)
            y0_, x0_ = ty * TILE, tx * TILE
            color[:, y0_:y0_ + hh, x0_:x0_ + ww] = c.T.reshape(3, hh, ww)
            depth[0, y0_:y0_ + hh, x0_:x0_ + ww] = d.reshape(hh, ww)
            alpha_out[0, y0_:y0_ + hh, x0_:x0_ + ww] = a.reshape(hh, ww)
            n_contrib[y0_:y0_ + hh, x0_:x0_ + ww] = nc.reshape(hh, ww).numpy()
    return color, depth, alpha_out, n_contrib


def render_with_grads(pre, point_list, ranges, bg, H, W, wc, wd, wa):
    """Forward AND the gradient of  L = sum(color*wc) + sum(depth*wd) + sum(alpha*wa)  with respect to the blend
    inputs pre[BLEND_KEYS], one tile at a time (the autograd graph of a whole 1080p render of 2.6 M (splat, tile)
    pairs does not fit in memory).  Same arithmetic as render(): blend_tile is the one implementation.
    Returns (color, depth, alpha, n_contrib, grads) with grads[k] shaped like pre[k]."""
    dt = pre["px"].dtype
    gx, gy = pre["grid"]
    color = torch.zeros(3, H, W, dtype=dt) + bg.to(dt)[:, None, None]
    depth = torch.zeros(1, H, W, dtype=dt)
    alpha_out = torch.zeros(1, H, W, dtype=dt)
    n_contrib = np.zeros((H, W), dtype=np.int64)
    grads = {k: torch.zeros_like(pre[k]) for k in BLEND_KEYS}
    for ty in range(gy):
        for tx in range(gx):
            s, e = ranges[ty * gx + tx]
            hh, ww, pxf, pyf = _tile_pixels(ty, tx, H, W, dt)
            if e <= s or hh == 0 or ww == 0:
                continue
            ids = torch.from_numpy(point_list[s:e])
            loc = [pre[k][ids].detach().requires_grad_(True) for k in BLEND_KEYS]
            c, d, a, nc = blend_tile(*loc, pxf, pyf, bg)
            y0_, x0_ = ty * TILE, tx * TILE
            sl = (slice(y0_, y0_ + hh), slice(x0_, x0_ + ww))
            loss = ((c.T.reshape(3, hh, ww) * wc[(slice(None),) + sl]).sum() + (d.reshape(hh, ww) * wd[(0,) + sl]).sum()
                    + (a.reshape(hh, ww) * wa[(0,) + sl]).sum())
            g = torch.autograd.grad(loss, loc, allow_unused=True)
            for k, gk in zip(BLEND_KEYS, g):
                if gk is not None:
                    grads[k].index_add_(0, ids, gk)
            color[(slice(None),) + sl] = c.detach().T.reshape(3, hh, ww)
            depth[(0,) + sl] = d.detach().reshape(hh, ww)
            alpha_out[(0,) + sl] = a.detach().reshape(hh, ww)
            n_contrib[sl] = nc.reshape(hh, ww).numpy()
    return color, depth, alpha_out, n_contrib, grads


def rasterize(means3D, scales, rotations, opacities, shs, confidence, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
              H, W, bg, sh_degree, scale_modifier=1.0):
    """Full forward.  Returns (color [3,H,W], radii [N], depth [1,H,W], alpha [1,H,W], aux)."""
    pre = preprocess(means3D, scales, rotations, opacities, shs, confidence, viewmatrix, projmatrix, campos, tanfovx,
                     tanfovy, H, W, sh_degree, scale_modifier)
    keys, plist, ranges = build_tile_lists(pre)
    color, depth, alpha, n_contrib = render(pre, plist, ranges, bg, H, W)
    return color, pre["radius"], depth, alpha, dict(pre=pre, keys=keys, point_list=plist, ranges=ranges,
                                                    n_contrib=n_contrib)


# synthetic scene recipe shared with bench.py lives with the product's data helpers
from syn3r_amd.synthetic import look_at_camera, synthetic_gaussians  # noqa: E402,F401
