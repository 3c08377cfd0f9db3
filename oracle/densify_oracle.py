"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by syn3r_amd/): numpy restatement of the adaptive density
control of 3D Gaussian Splatting (Kerbl, Kopanas, Leimkuehler, Drettakis 2023, section 5.2 and the released
`GaussianModel.densify_and_prune`), which FSGS' training loop - the loop behind gsTrainer.training()/finetune(), call
sites model/diffusionGS.py:139,1640, un-vendored submodule - applies to the Gaussians.

PARITY UNPINNED (no FSGS source, no fixtures; FSGS' additional proximity-guided unpooling is not restated).  The split
positions need random draws: the standard-normal samples are an INPUT here, so the implementation under test and this
restatement consume the same numbers."""
import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def rotation_matrices(q):
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((q.shape[0], 3, 3), q.dtype)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - r * z); R[:, 0, 2] = 2 * (x * z + r * y)
    R[:, 1, 0] = 2 * (x * y + r * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - r * x)
    R[:, 2, 0] = 2 * (x * z - r * y); R[:, 2, 1] = 2 * (y * z + r * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def densify_and_prune(P: dict, grad_accum, denom, max_radii2D, normals, *, max_grad, min_opacity, extent, max_screen_size,
                      percent_dense=0.01):
    """P: {'xyz' [n,3], 'features' [n,M,3], 'opacity' [n] (logit), 'scaling' [n,3] (log), 'rotation' [n,4],
    'confidence' [n]}; normals [2*n_split,3] standard-normal draws in split order.  Returns (P_new, provenance) where
    provenance[i] is the index of the input Gaussian output i descends from."""
    P = {k: np.array(v, dtype=np.float64 if False else np.float32) for k, v in P.items()}
    n = P["xyz"].shape[0]
    with np.errstate(invalid="ignore", divide="ignore"):
        grads = (grad_accum / denom).astype(np.float32)
    grads[np.isnan(grads)] = 0.0
    gn = np.linalg.norm(grads.reshape(n, -1), axis=1)
    prov = np.arange(n)
    radii = np.array(max_radii2D, dtype=np.float32)
    smax = np.exp(P["scaling"]).max(1)
    # clone
    sel = (gn >= max_grad) & (smax <= percent_dense * extent)
    P = {k: np.concatenate([v, v[sel]]) for k, v in P.items()}
    prov = np.concatenate([prov, prov[sel]])
    radii = np.concatenate([radii, np.zeros(int(sel.sum()), np.float32)])
    # split
    m = P["xyz"].shape[0]
    padded = np.zeros(m, np.float32)
    padded[:n] = gn
    smax = np.exp(P["scaling"]).max(1)
    sel = (padded >= max_grad) & (smax > percent_dense * extent)
    k = int(sel.sum())
    assert normals.shape == (2 * k, 3), (normals.shape, k)
    sc = np.exp(P["scaling"][sel])
    stds = np.tile(sc, (2, 1))
    R = np.tile(rotation_matrices(P["rotation"][sel]), (2, 1, 1))
    new = {"xyz": np.einsum("nij,nj->ni", R, normals.astype(np.float32) * stds) + np.tile(P["xyz"][sel], (2, 1)),
           "scaling": np.log(stds / np.float32(1.6)), "rotation": np.tile(P["rotation"][sel], (2, 1)),
           "features": np.tile(P["features"][sel], (2, 1, 1)), "opacity": np.tile(P["opacity"][sel], 2),
           "confidence": np.tile(P["confidence"][sel], 2)}
    P = {kk: np.concatenate([v, new[kk].astype(np.float32)]) for kk, v in P.items()}
    prov = np.concatenate([prov, np.tile(prov[sel], 2)])
    radii = np.concatenate([radii, np.zeros(2 * k, np.float32)])
    keep = ~np.concatenate([sel, np.zeros(2 * k, bool)])
    P = {kk: v[keep] for kk, v in P.items()}
    prov, radii = prov[keep], radii[keep]
    # prune
    prune = _sigmoid(P["opacity"]) < min_opacity
    if max_screen_size:
        prune |= (radii > max_screen_size) | (np.exp(P["scaling"]).max(1) > 0.1 * extent)
    P = {kk: v[~prune] for kk, v in P.items()}
    return P, prov[~prune]


def reset_opacity(opacity_logit):
    o = np.minimum(_sigmoid(opacity_logit.astype(np.float32)), np.float32(0.01))
    return np.log(o / (1 - o))
