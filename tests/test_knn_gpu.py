"""3-nearest-neighbour mean squared distance (csrc/knn.hip, the simple-knn `distCUDA2` quantity) against the CPU
oracle: brute force in the kernel's fp32 operation order (bit-exact) and scipy's k-d tree (independent algorithm)."""
import numpy as np
import pytest
import torch

from oracle import knn_oracle as KO

pytestmark = pytest.mark.gpu


def _cloud(kind: str, n: int, seed: int) -> np.ndarray:
    g = np.random.default_rng(seed)
    if kind == "uniform":
        return g.random((n, 3), dtype=np.float32) * np.float32(4.0) - np.float32(2.0)
    if kind == "clustered":              # dense blobs + sparse background: some points need many boxes
        c = g.normal(size=(8, 3)).astype(np.float32) * 3
        p = c[g.integers(0, 8, n)] + g.normal(size=(n, 3)).astype(np.float32) * np.float32(0.05)
        p[: n // 20] = g.normal(size=(n // 20, 3)).astype(np.float32) * 20
        return p.astype(np.float32)
    if kind == "duplicates":             # repeated points (zero distances) and an integer lattice (many exact ties)
        base = g.integers(0, 12, size=(n, 3)).astype(np.float32)
        base[: n // 4] = base[n // 4: n // 2][: n // 4]
        return base
    if kind == "planar":                 # degenerate bounding box (zero extent on one axis)
        p = g.random((n, 3), dtype=np.float32)
        p[:, 2] = np.float32(0.5)
        return p
    raise ValueError(kind)


@pytest.mark.parametrize("kind,n", [("uniform", 4), ("uniform", 5), ("uniform", 1000), ("uniform", 1024), ("uniform", 1025),
                                    ("clustered", 6000), ("duplicates", 5000), ("planar", 3000), ("uniform", 20000)])
def test_knn3_matches_bruteforce_bit_exact(kind, n, gpu):
    from syn3r_amd.gs.train_ops import knn3_mean_dist2
    p = _cloud(kind, n, seed=n)
    got = knn3_mean_dist2(torch.from_numpy(p).to(gpu)).cpu().numpy()
    exp = KO.mean_dist2_bruteforce(p)
    assert np.array_equal(got, exp), (np.abs(got - exp).max(), int((got != exp).sum()))


def test_knn3_matches_kdtree_at_init_cloud_size(gpu):
    """200 k points (the benchmark's Gaussian count; a dust3r cloud is of this order): independent algorithm, float64."""
    from syn3r_amd.gs.train_ops import knn3_mean_dist2
    p = _cloud("clustered", 200_000, seed=7)
    got = knn3_mean_dist2(torch.from_numpy(p).to(gpu)).cpu().numpy().astype(np.float64)
    exp = KO.mean_dist2_kdtree(p)
    assert np.allclose(got, exp, rtol=2e-5, atol=1e-9), np.abs(got - exp).max()


def test_knn3_rejects_bad_input(gpu):
    from syn3r_amd import _lib
    from syn3r_amd.gs.train_ops import knn3_mean_dist2
    with pytest.raises(_lib.Syn3rError):
        knn3_mean_dist2(torch.zeros(3, 3, device=gpu))
    with pytest.raises(ValueError):
        knn3_mean_dist2(torch.zeros(10, 2, device=gpu))


def test_set_from_pcd_scales_follow_the_knn(gpu):
    """GaussianModel.set_from_pcd (reset_gaussians_from_pcd, diffusionGS.py:1685-1687): log-scale = log sqrt(knn mean)."""
    from syn3r_amd.gs.trainer import GaussianModel
    p = _cloud("uniform", 3000, seed=3)
    col = np.random.default_rng(1).random((3000, 3)).astype(np.float32)
    g = GaussianModel(np.zeros((1, 3)), np.zeros((1, 3)), np.array([[1., 0, 0, 0]]), np.zeros(1), np.zeros((1, 16, 3)), device=gpu)
    g.set_from_pcd(p, col, append=False)
    exp = np.log(np.sqrt(np.maximum(KO.mean_dist2_bruteforce(p), 1e-7)))
    assert g._scaling.shape == (3000, 3)
    assert np.allclose(g._scaling.detach().cpu().numpy(), np.repeat(exp[:, None], 3, 1), rtol=1e-6, atol=1e-6)
