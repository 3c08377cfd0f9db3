"""Adaptive density control of the trainer (GSTrainer.densify_and_prune / reset_opacity) against
the numpy restatement in oracle/densify_oracle.py, on the CPU: the selection rules, the split geometry from shared
normal draws, the pruning, and the optimiser moments following the Gaussians."""
import numpy as np
import torch

from oracle import densify_oracle as DO


def _model(n, seed):
    from syn3r_amd.gs.trainer import GaussianModel
    g = np.random.default_rng(seed)
    xyz = g.normal(size=(n, 3)).astype(np.float32)
    log_s = np.log(g.uniform(0.002, 0.05, size=(n, 3))).astype(np.float32)
    rot = g.normal(size=(n, 4)).astype(np.float32)
    op = g.normal(size=n).astype(np.float32) * 3
    sh = g.normal(size=(n, 16, 3)).astype(np.float32)
    gm = GaussianModel(xyz, log_s, rot, op, sh, device="cpu")
    gm.confidence = torch.from_numpy(g.uniform(0.1, 1.0, n).astype(np.float32))
    return gm, g


def _as_dict(gm):
    return {"xyz": gm._xyz.detach().numpy(), "features": gm._features.detach().numpy(), "opacity": gm._opacity.detach().numpy(),
            "scaling": gm._scaling.detach().numpy(), "rotation": gm._rotation.detach().numpy(), "confidence": gm.confidence.numpy()}


def test_densify_and_prune_matches_oracle():
    from syn3r_amd.gs.trainer import GSTrainer, OptimizationParams
    n = 600
    gm, g = _model(n, 5)
    tr = GSTrainer(gm, [], OptimizationParams())
    # optimiser moments with recognisable values: moment of Gaussian i = i + 1
    for p in gm.parameters():
        m = torch.arange(1, n + 1, dtype=torch.float32).reshape(n, *([1] * (p.dim() - 1))).expand_as(p).clone()
        tr.optimizer.state[p] = {"step": 7, "exp_avg": m.clone(), "exp_avg_sq": 2 * m}
    gm.ensure_stats()
    gm.denom[:] = torch.from_numpy(g.integers(0, 5, (n, 1)).astype(np.float32))            # some never visible (0/0 -> 0)
    gm.xyz_gradient_accum[:] = torch.from_numpy(g.uniform(0, 0.002, (n, 1)).astype(np.float32)) * gm.denom
    gm.max_radii2D[:] = torch.from_numpy(g.uniform(0, 30, n).astype(np.float32))
    before = _as_dict(gm)
    accum, denom, radii = gm.xyz_gradient_accum.numpy().copy(), gm.denom.numpy().copy(), gm.max_radii2D.numpy().copy()
    extent, max_grad, min_op, screen = 2.0, 0.0008, 0.05, 20.0
    drawn = []

    def noise(k):
        z = torch.from_numpy(np.random.default_rng(99).normal(size=(k, 3)).astype(np.float32))
        drawn.append(z.numpy())
        return z
    tr._split_noise = noise
    n_clone, n_split, n_prune = tr.densify_and_prune(max_grad, min_op, extent, screen)
    assert n_clone > 20 and n_split > 20 and n_prune > 20          # every rule is exercised
    exp, prov = DO.densify_and_prune(before, accum, denom, radii, drawn[0], max_grad=max_grad, min_opacity=min_op, extent=extent,
                                     max_screen_size=screen)
    got = _as_dict(gm)
    for k in exp:
        assert got[k].shape == exp[k].shape, k
        np.testing.assert_allclose(got[k], exp[k], rtol=2e-6, atol=2e-6, err_msg=k)
    m_new = gm._xyz.shape[0]
    assert m_new == n + n_clone + n_split - n_prune
    # optimiser: groups hold the new parameters; survivors of the input set keep their moments, new Gaussians start at 0
    for grp, p in zip(tr.optimizer.param_groups, gm.parameters()):
        assert grp["params"][0] is p
        st = tr.optimizer.state[p]
        assert st["step"] == 7 and st["exp_avg"].shape == p.shape
    # a surviving input Gaussian keeps its moment (index + 1); clones and split children start from zero
    first = tr.optimizer.state[gm._xyz]["exp_avg"][:, 0].numpy()
    kept_orig = first > 0
    assert np.array_equal(first[kept_orig], (prov[kept_orig] + 1).astype(np.float32))
    assert np.all(first[~kept_orig] == 0) and (~kept_orig).sum() > 0
    # statistics restart
    assert gm.xyz_gradient_accum.shape == (m_new, 1) and float(gm.xyz_gradient_accum.abs().sum()) == 0.0
    assert gm.max_radii2D.shape == (m_new,) and gm.confidence.shape == (m_new,)


def test_reset_opacity_matches_oracle():
    from syn3r_amd.gs.trainer import GSTrainer, OptimizationParams
    gm, _ = _model(300, 6)
    tr = GSTrainer(gm, [], OptimizationParams())
    tr.optimizer.state[gm._opacity] = {"step": 3, "exp_avg": torch.ones(300), "exp_avg_sq": torch.ones(300)}
    before = gm._opacity.detach().numpy().copy()
    tr.reset_opacity()
    np.testing.assert_allclose(gm._opacity.detach().numpy(), DO.reset_opacity(before), rtol=1e-6, atol=1e-6)
    assert float(gm.get_opacity.detach().max()) <= 0.01 + 1e-7
    st = tr.optimizer.state[gm._opacity]
    assert float(st["exp_avg"].abs().sum()) == 0.0 and tr.optimizer.param_groups[2]["params"][0] is gm._opacity


def test_density_control_schedule():
    """_density_control: statistics every iteration, densify on the interval after densify_from_iter (screen-size pruning
    only after the first opacity reset), opacity reset on its interval, nothing after densify_until_iter."""
    from syn3r_amd.gs.trainer import GSTrainer, OptimizationParams
    gm, _ = _model(50, 7)
    tr = GSTrainer(gm, [], OptimizationParams(densify_from_iter=4, densification_interval=3, opacity_reset_interval=7,
                                              densify_until_iter=12))
    calls = []
    tr.densify_and_prune = lambda *a: calls.append(("densify", tr.iteration + 1, a[3]))
    tr.reset_opacity = lambda: calls.append(("reset", tr.iteration + 1))
    for it in range(14):
        tr.iteration = it
        mp = torch.zeros(50, 3, requires_grad=True)
        mp.grad = torch.ones(50, 3)
        tr._density_control({"visibility_filter": torch.ones(50, dtype=torch.bool), "viewspace_points": mp,
                             "radii": torch.full((50,), 3)})
    assert calls == [("densify", 6, None), ("reset", 7), ("densify", 9, 20.0)]
    assert float(gm.denom[0]) == 11.0 and float(gm.max_radii2D[0]) == 3.0      # iterations 1..11 accumulate
