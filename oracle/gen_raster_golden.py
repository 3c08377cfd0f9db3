"""Generate tests/golden/raster_200k_1080p.npz: the CPU restatement (oracle/raster_oracle.py, float64) run ONCE
on bench.py's scene — 200 000 Gaussians, 1920x1080, SH degree 3, P ~ 2.6 M (Gaussian, tile) pairs — forward and
the gradient of a seeded linear functional of (colour, depth, alpha).  ~10 min of CPU; test infrastructure.

    python oracle/gen_raster_golden.py [N H W]

PARITY UNPINNED vs the reference's CUDA rasteriser (source absent, see raster_oracle.py): this pins the HIP
kernels to the restatement at the size where the saturation walk-back, the 256-splat staging batches and the
per-quadrant visit lists are actually exercised.  The tile lists themselves are NOT stored: preprocess +
build_tile_lists take seconds at this size, so the GPU test recomputes them with the oracle and compares index
for index.  Stored: strided images, full-image sums, per-parameter-group gradient sums and |.|-sums, and the
gradients of a seeded sample of Gaussians.
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import raster_oracle as RO  # noqa: E402

GOLD = ROOT / "tests" / "golden"
STRIDE, OFFSET = 5, 2
NSAMPLE = 4096
BG = (0.1, 0.3, 0.7)


def loss_weights(H, W, seed=3):
    """the linear functional's weights (tests rebuild them from the seed)"""
    g = torch.Generator().manual_seed(seed)
    wc = torch.randn(3, H, W, generator=g, dtype=torch.float64)
    wd = 0.3 * torch.randn(1, H, W, generator=g, dtype=torch.float64)
    wa = torch.randn(1, H, W, generator=g, dtype=torch.float64)
    return wc, wd, wa


def sample_ids(N, seed=11):
    g = torch.Generator().manual_seed(seed)
    return torch.randperm(N, generator=g)[:min(NSAMPLE, N)].sort().values


def run(N, H, W, seed=1234, log=print):
    dt = torch.float64
    m, s, q, o, sh = RO.synthetic_gaussians(N, seed=seed, dtype=dt)
    view, proj, campos, tfx, tfy = RO.look_at_camera(H, W, dtype=dt)
    ps = [t.clone().requires_grad_(True) for t in (m, s, q, o, sh)]
    bg = torch.tensor(BG, dtype=dt)
    t0 = time.time()
    pre = RO.preprocess(ps[0], ps[1], ps[2], ps[3], ps[4], None, view, proj, campos, tfx, tfy, H, W, 3)
    keys, plist, ranges = RO.build_tile_lists(pre)
    log(f"preprocess + lists: P = {len(plist)} pairs, {time.time() - t0:.1f} s")
    wc, wd, wa = loss_weights(H, W)
    t0 = time.time()
    color, depth, alpha, n_contrib, g = RO.render_with_grads(pre, plist, ranges, bg, H, W, wc, wd, wa)
    log(f"blend forward + tile-wise gradient: {time.time() - t0:.1f} s")
    torch.autograd.backward([pre[k] for k in RO.BLEND_KEYS], [g[k] for k in RO.BLEND_KEYS])
    grads = dict(zip(("m", "s", "q", "o", "sh"), [p.grad for p in ps]))
    return dict(color=color, depth=depth, alpha=alpha, n_contrib=n_contrib, grads=grads, P=len(plist),
                screen_grad=torch.stack([g["px"], g["py"]], 1), radius=pre["radius"])


def main():
    N, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (200_000, 1080, 1920)
    torch.set_num_threads(8)
    r = run(N, H, W)
    ids = sample_ids(N)
    sl = (slice(None), slice(OFFSET, None, STRIDE), slice(OFFSET, None, STRIDE))
    out = dict(N=np.int64(N), H=np.int64(H), W=np.int64(W), P=np.int64(r["P"]), stride=np.int64(STRIDE), offset=np.int64(OFFSET),
               bg=np.array(BG), color=r["color"][sl].numpy().astype(np.float32), depth=r["depth"][sl].numpy().astype(np.float32),
               alpha=r["alpha"][sl].numpy().astype(np.float32),
               n_contrib=r["n_contrib"][OFFSET::STRIDE, OFFSET::STRIDE].astype(np.int32),
               color_sum=r["color"].sum((1, 2)).numpy(), depth_sum=r["depth"].sum().numpy(), alpha_sum=r["alpha"].sum().numpy(),
               sample_ids=ids.numpy())
    for k, gk in r["grads"].items():
        out[f"grad_{k}_sum"] = gk.sum().numpy()
        out[f"grad_{k}_abs"] = gk.abs().sum().numpy()
        out[f"grad_{k}_max"] = gk.abs().max().numpy()
        out[f"grad_{k}_sample"] = gk[ids].numpy().astype(np.float32)
    # viewspace-point gradient norm (what FSGS accumulates for densification): d L / d (pixel-space mean)
    out["screen_grad_sample"] = r["screen_grad"][ids].numpy().astype(np.float32)
    name = "raster_200k_1080p.npz" if (N, H, W) == (200_000, 1080, 1920) else f"raster_{N}_{W}x{H}.npz"
    np.savez_compressed(GOLD / name, **out)
    print("wrote", name, {k: getattr(v, "shape", ()) for k, v in out.items()})


if __name__ == "__main__":
    main()
