"""Modified Euler scheduler of SYN3R on the HIP path.

Mirror of the reference's vendored
`thirdparty/diffusers/src/diffusers/schedulers/scheduling_euler_discrete.py`
(class and method names, argument meaning, output fields).  The sigma schedule
is host arithmetic (numpy/torch, as the reference); the per-step tensor work
runs in `syn3r_step_interp` / `syn3r_step_replace` (include/syn3r_hip.h).
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Union

import numpy as np
import torch

from .. import _lib as L

# stabilityai/stable-video-diffusion-img2vid-xt scheduler_config.json values;
# `from_config` overrides them from a local weights directory.
SVD_XT_SCHEDULER_CONFIG = dict(
    num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
    prediction_type="v_prediction", interpolation_type="linear", use_karras_sigmas=True, sigma_min=0.002,
    sigma_max=700.0, timestep_spacing="leading", timestep_type="continuous", steps_offset=1,
    rescale_betas_zero_snr=False,
)


@dataclass
class EulerDiscreteSchedulerOutput:
    """Reference: scheduling_euler_discrete.py:37-53 (with the added `grad` field)."""
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None
    grad: Optional[torch.Tensor] = None


class EulerDiscreteScheduler:
    """The scheduler object the SVD pipelines drive (reference: scheduling_euler_discrete.py:137-1559, live methods
    only).  It serves ONE family of configurations — the one SVD ships: variance-preserving training betas, Karras
    noise levels, v-prediction with continuous timesteps — and rejects the rest instead of carrying unused branches.

    Host-side schedule, from the formulas (Karras et al. 2022, eq. 5; the SVD model card):
        beta_k   = lerp(sqrt(beta_start), sqrt(beta_end), k / (T-1))^2          ("scaled_linear", fp32)
        abar_k   = prod_{j<=k} (1 - beta_j)                                      (fp32 running product)
        s_k      = sqrt((1 - abar_k) / abar_k)                                   training noise levels
        sigma_i  = (smax^(1/7) + i/(n-1) * (smin^(1/7) - smax^(1/7)))^7          i = 0..n-1, float64 -> fp32
        t_i      = ln(sigma_i) / 4                                               what the UNet is conditioned on
    `smax` / `smin` default to the training levels interpolated at the first / last "leading" step
    (k = (n-1-i) * (T // n) + steps_offset) when the configuration leaves them open.  The operation order above is what
    makes `sigmas` bit-identical to tests/golden/sched_sigmas*.npz.
    """

    order = 1
    _KARRAS_RHO = 7.0

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", trained_betas=None, prediction_type: str = "epsilon",
                 interpolation_type: str = "linear", use_karras_sigmas: Optional[bool] = False,
                 sigma_min: Optional[float] = None, sigma_max: Optional[float] = None,
                 timestep_spacing: str = "linspace", timestep_type: str = "discrete", steps_offset: int = 0,
                 rescale_betas_zero_snr: bool = False):
        given = dict(locals())
        given.pop("self")
        self.config = SimpleNamespace(**given)
        required = dict(beta_schedule="scaled_linear", prediction_type="v_prediction", interpolation_type="linear",
                        use_karras_sigmas=True, timestep_spacing="leading", timestep_type="continuous",
                        trained_betas=None, rescale_betas_zero_snr=False)
        wrong = {k: given[k] for k, v in required.items() if given[k] != v}
        if wrong:
            raise NotImplementedError(f"EulerDiscreteScheduler (HIP path) serves the SVD configuration only; got {wrong}, "
                                      f"needs {({k: required[k] for k in wrong})}")
        T = int(num_train_timesteps)
        root_beta = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=torch.float32)
        self.betas = root_beta ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self._train_sigmas = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5       # ascending, fp32 [T]
        self.use_karras_sigmas = True
        self.is_scale_input_called = False
        self.num_inference_steps = None
        self._step_index = self._begin_index = None
        # before set_timesteps: the T training levels, largest first, and their conditioning values
        self._install(self._train_sigmas.flip(0))

    def _install(self, sigmas_desc: torch.Tensor, device=None):
        self.timesteps = (0.25 * sigmas_desc.log()).to(device=device)
        self.sigmas = torch.cat([sigmas_desc, torch.zeros(1)])          # host-resident: indexing it never synchronises

    @classmethod
    def from_config(cls, config: dict):
        keys = SVD_XT_SCHEDULER_CONFIG.keys() | {"trained_betas"}
        return cls(**{k: v for k, v in config.items() if k in keys})

    # ------------------------------------------------------------------ schedule (host)
    @property
    def init_noise_sigma(self):
        """Scale of the initial latents for "leading" spacing: sqrt(sigma_max^2 + 1) (:248-254)."""
        return (self.sigmas.max() ** 2 + 1) ** 0.5

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def _end_levels(self, n: int):
        """(smax, smin) when the configuration does not fix them: the training levels at the first / last leading step."""
        cfg = self.config
        k = (np.arange(n)[::-1] * (cfg.num_train_timesteps // n) + cfg.steps_offset).astype(np.float32)
        s = self._train_sigmas.numpy()
        lv = np.interp(k, np.arange(s.shape[0]), s)
        return float(lv[0]), float(lv[-1])

    def set_timesteps(self, num_inference_steps: int, device: Union[str, torch.device] = None):
        """`n` Karras noise levels + the terminal zero, and t = ln(sigma)/4 (reference :310-372 for this configuration)."""
        n = int(num_inference_steps)
        cfg = self.config
        smax, smin = cfg.sigma_max, cfg.sigma_min
        if smax is None or smin is None:
            hi, lo = self._end_levels(n)
            smax = hi if smax is None else smax
            smin = lo if smin is None else smin
        inv = 1.0 / self._KARRAS_RHO
        a, b = smax ** inv, smin ** inv
        levels = (a + np.linspace(0, 1, n) * (b - a)) ** self._KARRAS_RHO      # float64
        self.num_inference_steps = n
        self._install(torch.from_numpy(levels).to(torch.float32), device)
        self._step_index = self._begin_index = None

    def scale_model_input(self, sample: torch.Tensor, timestep, step_i) -> torch.Tensor:
        """:281-308 — `step_i` overrides the internal index (:300)."""
        self._step_index = step_i
        sigma = self.sigmas[self.step_index]
        sample = sample / ((sigma ** 2 + 1) ** 0.5)
        self.is_scale_input_called = True
        return sample

    # ------------------------------------------------------------------ steps (HIP)
    def _scalars(self):
        """0-dim fp32 CPU tensor arithmetic exactly as :728,:792,:798-800."""
        sigma = self.sigmas[self.step_index]
        c_out = -sigma / (sigma ** 2 + 1) ** 0.5
        denom = sigma ** 2 + 1
        sqrt_sigma = sigma ** 0.5
        dt = self.sigmas[self.step_index + 1] - sigma
        return float(sigma), float(dt), float(c_out), float(denom), float(sqrt_sigma)

    def _check_timestep(self, timestep):
        if isinstance(timestep, int) or isinstance(timestep, (torch.IntTensor, torch.LongTensor)):
            raise ValueError("Passing integer indices (e.g. from `enumerate(timesteps)`) as timesteps to"
                             " `EulerDiscreteScheduler.step()` is not supported. Make sure to pass"
                             " one of the `scheduler.timesteps` as a timestep.")

    def _prep(self, model_output, sample, temp_cond_latents, mask, lambda_ts):
        if self.config.prediction_type != "v_prediction":
            raise NotImplementedError("only v_prediction (the SVD configuration) is implemented on the HIP path")
        dev = L.require_gpu(model_output, sample)
        if model_output.shape != sample.shape or model_output.dim() != 5 or model_output.shape[0] != 1:
            raise ValueError(f"expected [1,F,C,h,w] tensors, got {tuple(model_output.shape)} / {tuple(sample.shape)}")
        _, F, Cc, h, w = model_output.shape
        v = model_output.detach().contiguous()
        x = sample.detach().contiguous()
        cond = msk = lam = None
        if temp_cond_latents is not None:
            b, cond_len, c2, h2, w2 = temp_cond_latents.shape
            if (cond_len, c2, h2, w2) != (F, Cc, h, w) or b < 2:
                raise ValueError(f"temp_cond_latents {tuple(temp_cond_latents.shape)} does not match the sample")
            # the conditioning latents, the mask and the lambda schedule are the same objects at every step of a pass:
            # their kernel-side forms (fp32 cond slice, expanded mask, HOST copy of lambda_ts) are prepared once per
            # object, not once per call — a device-resident lambda_ts used to cost a blocking device->host copy per call
            cond = self._prepared("cond", temp_cond_latents,
                                  lambda t: t[1].detach().to(device=dev, dtype=torch.float32).contiguous())
            msk = self._prepared("mask", mask, lambda t: t.detach().to(device=dev, dtype=torch.float32)
                                 .expand(1, F - 2, Cc, h, w).contiguous())
            lam_host = self._prepared("lambda", lambda_ts, lambda t: t.detach().to("cpu", torch.float64).numpy().copy())
            lam = L.host_f64(lam_host[self.step_index].tolist())
        return dev, v, x, cond, msk, lam, (F, Cc, h, w)

    def _prepared(self, slot: str, src: torch.Tensor, make):
        """Per-object cache (identity + in-place version counter; the entry keeps `src` alive, so its address cannot be
        handed to another tensor meanwhile).  Sixteen entries per slot: the forward and the time-flipped backward pass
        alternate two sets of objects, each with its full-frame tensors and its four guidance-tile views (10 per slot)."""
        cache = self.__dict__.setdefault("_prep_cache", {}).setdefault(slot, [])
        for ent in cache:
            if ent[0] is src and ent[1] == src._version:
                return ent[2]
        val = make(src)
        cache.insert(0, (src, src._version, val))
        del cache[16:]
        return val

    def step_interp(self, model_output, timestep, sample, temp_cond_latents=None, mask=None, lambda_ts=None,
                    step_i=None, lr=None, compute_grad=False, return_dict: bool = True):
        """Reference: scheduling_euler_discrete.py:633-814."""
        self._check_timestep(timestep)
        self._step_index = step_i
        dev, v, x, cond, msk, lam, (F, Cc, h, w) = self._prep(model_output, sample, temp_cond_latents, mask,
                                                              lambda_ts)
        if compute_grad and cond is None:
            raise ValueError("compute_grad needs temp_cond_latents / mask / lambda_ts")
        lib = L.load()
        sigma, dt, c_out, denom, sqrt_sigma = self._scalars()
        prev = torch.empty_like(v)
        x0 = torch.empty(v.shape, dtype=torch.float32, device=dev)
        grad = torch.empty(v.shape, dtype=torch.float32, device=dev) if compute_grad else None
        need = lib.syn3r_step_workspace_bytes(F, Cc, h, w)
        ws = L.workspace(dev, need, "step")
        rc = lib.syn3r_step_interp(L.ptr(v), L.dtype_tag(v), L.ptr(x), L.dtype_tag(x), L.ptr(cond), L.ptr(msk), lam,
                                   sigma, dt, c_out, denom, sqrt_sigma, float(lr) if lr is not None else 0.0,
                                   1 if compute_grad else 0, L.ptr(prev), L.ptr(x0), L.ptr(grad), F, Cc, h, w,
                                   L.ptr(ws), ws.numel(), L.stream_ptr(dev))
        L.check(rc, "syn3r_step_interp")
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return EulerDiscreteSchedulerOutput(prev_sample=prev, pred_original_sample=x0, grad=grad)

    def step_interp_prob_uncertain(self, model_output, timestep, sample, temp_cond_latents=None, mask=None,
                                   lambda_ts=None, step_i=None, weight_clamp=None, return_dict: bool = True):
        """Reference: scheduling_euler_discrete.py:1343-1515 (weight clamp hard-coded 0.4 at :1476)."""
        self._check_timestep(timestep)
        self._step_index = step_i
        if temp_cond_latents is None:
            raise ValueError("step_interp_prob_uncertain needs temp_cond_latents / mask / lambda_ts")
        dev, v, x, cond, msk, lam, (F, Cc, h, w) = self._prep(model_output, sample, temp_cond_latents, mask,
                                                              lambda_ts)
        lib = L.load()
        sigma, dt, c_out, denom, _ = self._scalars()
        prev = torch.empty_like(v)
        x0 = torch.empty(v.shape, dtype=torch.float32, device=dev)
        need = lib.syn3r_step_workspace_bytes(F, Cc, h, w)
        ws = L.workspace(dev, need, "step")
        rc = lib.syn3r_step_replace(L.ptr(v), L.dtype_tag(v), L.ptr(x), L.dtype_tag(x), L.ptr(cond), L.ptr(msk), lam,
                                    sigma, dt, c_out, denom, L.ptr(prev), L.ptr(x0), F, Cc, h, w, L.ptr(ws),
                                    ws.numel(), L.stream_ptr(dev))
        L.check(rc, "syn3r_step_replace")
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return EulerDiscreteSchedulerOutput(prev_sample=prev, pred_original_sample=x0)
