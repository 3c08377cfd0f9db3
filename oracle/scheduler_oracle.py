"""numpy restatement of SYN3R's modified Euler steps (test oracle, not product code).

Follows thirdparty/diffusers/src/diffusers/schedulers/scheduling_euler_discrete.py
  :633-814   step_interp
  :1343-1515 step_interp_prob_uncertain
Inputs are [1,F,C,h,w] arrays (float16 or float32 model_output / sample),
temp_cond [2,F,C,h,w] float32, mask [1,F-2,C,h,w] float32, lambda_row [F] float64.
The guidance gradient uses the closed form SURVEY.md §8a S2 derives
(grad = lr * sqrt(sigma) * g / std(g), g = (x0 - cond) * top_mask).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def scalars(sigmas: np.ndarray, step_i: int):
    """fp32 scalar arithmetic as the reference's 0-dim CPU tensors (:728,:792,:800)."""
    sigma = f32(sigmas[step_i])
    denom = f32(f32(sigma * sigma) + f32(1))
    c_out = f32(-sigma / f32(np.sqrt(denom, dtype=f32)))
    sqrt_sigma = f32(np.sqrt(sigma, dtype=f32))
    dt = f32(f32(sigmas[step_i + 1]) - sigma)
    return sigma, dt, c_out, denom, sqrt_sigma


def _pred_x0(model_output, sample, c_out, denom):
    v = np.asarray(model_output)
    x = np.asarray(sample).astype(f32)
    t = v.astype(f32) * c_out
    if v.dtype == np.float16:          # torch type promotion keeps half for half * 0-dim fp32
        t = t.astype(np.float16).astype(f32)
    return (t + x / denom).astype(f32), x


def _cutoffs(x0, cond, mask, lambda_row):
    """Per interior frame: mask_t [h,w] bool, |d| [C,h,w], cutoff value (:745-771)."""
    _, F, C, h, w = x0.shape
    valid = (f32(1) - mask) > f32(0.5)                        # :742
    res = {}
    for tau in range(1, F - 1):
        mbar = valid[0, tau - 1].astype(f32).mean(axis=0, dtype=f32)   # :752 (index tau of the ones-padded mask)
        mt = mbar > f32(0.5)
        n0 = int((~mt).sum())                                 # counts pixels, not elements (:755)
        mf = mt.astype(f32)[None]
        d = np.abs(x0[0, tau] * mf - cond[tau] * mf)
        srt = np.sort(d.ravel())
        wgt = min(max(float(lambda_row[tau]), 0.4), 1.0)
        k = int(wgt * (srt.size - n0)) + n0                   # :769
        res[tau] = (mbar, mt, d, srt[k - 1])
    return res


def step_interp(model_output, sample, temp_cond, mask, lambda_row, sigmas, step_i, lr=0.02, compute_grad=False):
    sigma, dt, c_out, denom, sqrt_sigma = scalars(sigmas, step_i)
    x0, x = _pred_x0(model_output, sample, c_out, denom)
    out = {}
    if compute_grad:
        cond = np.asarray(temp_cond[1], f32)
        cuts = _cutoffs(x0, cond, np.asarray(mask, f32), lambda_row)
        _, F, C, h, w = x0.shape
        top = np.ones((F, C, h, w), bool)                      # frames 0 and F-1: ones (:777-779)
        for tau, (_, mt, d, cut) in cuts.items():
            top[tau] = (d <= cut) & mt[None]
        g = np.where(top, x0[0] - cond, f32(0)).astype(f32)
        sd = f32(np.std(g.astype(np.float64), ddof=1))
        out["grad"] = (f32(lr) * (g / sd * sqrt_sigma))[None].astype(f32)
        out["top_masks"] = top
    deriv = (x - x0) / sigma
    prev = x + deriv * dt
    out["prev_sample"] = prev.astype(np.asarray(model_output).dtype)
    out["pred_original_sample"] = x0
    return out


def step_interp_prob_uncertain(model_output, sample, temp_cond, mask, lambda_row, sigmas, step_i):
    sigma, dt, c_out, denom, _ = scalars(sigmas, step_i)
    x0, x = _pred_x0(model_output, sample, c_out, denom)
    cond = np.asarray(temp_cond[1], f32)
    cuts = _cutoffs(x0, cond, np.asarray(mask, f32), lambda_row)
    x0 = x0.copy()
    for tau, (mbar, mt, d, cut) in cuts.items():
        t = f32(1) / (f32(1) - mbar + f32(1e-6))               # :1484
        wgt = t / (f32(1) + t)
        wgt = np.where(wgt >= f32(0.51), wgt, f32(0)).astype(f32)
        wgt = (d <= cut).astype(f32) * wgt[None]
        x0[0, tau] = (f32(1) - wgt) * x0[0, tau] + wgt * cond[tau]
    x0[0, 0] = cond[0]                                         # :1496-1499
    x0[0, -1] = cond[-1]
    deriv = (x - x0) / sigma
    prev = x + deriv * dt
    return {"prev_sample": prev.astype(np.asarray(model_output).dtype), "pred_original_sample": x0}
