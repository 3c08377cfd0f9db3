"""HIP scheduler steps vs the CPU oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

from oracle import golden_inputs as GI
from oracle import scheduler_oracle as SO

pytestmark = pytest.mark.gpu


def make_scheduler():
    from syn3r_amd.schedulers.scheduling_euler_discrete import SVD_XT_SCHEDULER_CONFIG, EulerDiscreteScheduler
    s = EulerDiscreteScheduler(**SVD_XT_SCHEDULER_CONFIG)
    s.set_timesteps(100)
    return s


def run_hip(c, dev, sch, compute_grad=None, replace=False):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    v, x, cond, mask, lam = t(c["model_output"]), t(c["sample"]), t(c["temp_cond"]), t(c["mask"]), t(c["lambda_ts"])
    ts = sch.timesteps[c["step_i"]]
    if replace:
        r = sch.step_interp_prob_uncertain(v, ts, x, cond, mask, lam, step_i=c["step_i"])
    else:
        r = sch.step_interp(v, ts, x, cond, mask, lam, step_i=c["step_i"], lr=0.02, compute_grad=compute_grad)
    assert sch.step_index == c["step_i"] + 1
    return r


def check_case(c, g, s, dev):
    sch = make_scheduler()
    sig = sch.sigmas.numpy()
    lam = c["lambda_ts"][c["step_i"]]
    half = c["model_output"].dtype == np.float16
    tol = dict(atol=2e-3, rtol=2e-3) if half else dict(atol=1e-5, rtol=1e-5)
    for cg in (True, False):
        r = run_hip(c, dev, sch, compute_grad=cg)
        o = SO.step_interp(c["model_output"], c["sample"], c["temp_cond"], c["mask"], lam, sig, c["step_i"], lr=0.02,
                           compute_grad=cg)
        x0 = r.pred_original_sample.cpu().numpy()
        prev = r.prev_sample.cpu().numpy()
        assert prev.dtype == c["model_output"].dtype
        np.testing.assert_allclose(x0, o["pred_original_sample"], atol=1e-6, rtol=1e-6)
        np.testing.assert_allclose(prev.astype(np.float32), o["prev_sample"].astype(np.float32), **tol)
        if cg:
            grad = r.grad.cpu().numpy()
            bad = np.abs(grad - o["grad"]) > (1e-6 + 1e-4 * np.abs(o["grad"]))
            assert bad.mean() < 1e-4, bad.mean()
        else:
            assert r.grad is None
        if g is not None:
            tag = "g1" if cg else "g0"
            np.testing.assert_allclose(x0[..., ::s, ::s], g[f"interp_{tag}_x0"], atol=1e-5, rtol=1e-5)
            np.testing.assert_allclose(prev[..., ::s, ::s].astype(np.float32),
                                       g[f"interp_{tag}_prev"].astype(np.float32), **tol)
            if cg:
                b = g["interp_g1_grad"]
                bad = np.abs(grad[..., ::s, ::s] - b) > (1e-5 + 1e-4 * np.abs(b))
                assert bad.mean() < 1e-4, bad.mean()
    r = run_hip(c, dev, sch, replace=True)
    o = SO.step_interp_prob_uncertain(c["model_output"], c["sample"], c["temp_cond"], c["mask"], lam, sig, c["step_i"])
    x0 = r.pred_original_sample.cpu().numpy()
    prev = r.prev_sample.cpu().numpy()
    assert (np.abs(x0 - o["pred_original_sample"]) > 1e-6).mean() < 1e-4
    np.testing.assert_allclose(prev.astype(np.float32), o["prev_sample"].astype(np.float32), **tol)
    if g is not None:
        assert (np.abs(x0[..., ::s, ::s] - g["replace_x0"]) > 1e-5).mean() < 1e-4
        np.testing.assert_allclose(prev[..., ::s, ::s].astype(np.float32), g["replace_prev"].astype(np.float32), **tol)


@pytest.mark.parametrize("name", list(GI.SCHED_CASES))
def test_steps_vs_oracle_and_golden(name, gpu, golden_dir):
    c = GI.sched_case(name)
    g = np.load(golden_dir / f"sched_{name}.npz")
    check_case(c, g, c["stride"], gpu)


@pytest.mark.parametrize("F,h,w,dt", [(14, 72, 128, "float16"), (14, 9, 7, "float32"), (3, 8, 8, "float32"),
                                      (25, 48, 72, "float16")])
def test_steps_other_frame_counts_vs_oracle(F, h, w, dt, gpu):
    """F=14 (BASELINE config) and ragged sizes: the reference hard-codes F=25, so the oracle is the checker."""
    c = GI.sched_inputs(h, w, dt, 30, "uniform", "pixel", 1, 100 + F + h, F=F)
    check_case(c, None, 1, gpu)


def test_all_masked_and_all_valid_frames(gpu):
    """Edge cases: a frame with no valid pixel (n0 = h*w) and with every pixel valid (n0 = 0)."""
    c = GI.sched_inputs(10, 12, "float32", 20, "binary", "pixel", 1, 77)
    c["mask"][0, 3] = 1.0   # all invalid
    c["mask"][0, 4] = 0.0   # all valid
    check_case(c, None, 1, gpu)


def test_step_rejects_bad_input(gpu):
    from syn3r_amd import _lib
    sch = make_scheduler()
    c = GI.sched_case("tiny_f32")
    t = lambda a: torch.from_numpy(a).to(gpu)
    with pytest.raises(ValueError):
        sch.step_interp(t(c["model_output"]), 3, t(c["sample"]), step_i=1)
    with pytest.raises(_lib.Syn3rError):
        sch.step_interp(torch.from_numpy(c["model_output"]), sch.timesteps[1], t(c["sample"]), step_i=1)
    with pytest.raises(ValueError):
        sch.step_interp(t(c["model_output"]), sch.timesteps[1], t(c["sample"]), t(c["temp_cond"][:, :5]),
                        t(c["mask"]), t(c["lambda_ts"]), step_i=1, lr=0.02, compute_grad=True)
