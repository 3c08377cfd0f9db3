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
    """Reference: scheduling_euler_discrete.py:137-1559 (live methods only)."""

    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", trained_betas=None, prediction_type: str = "epsilon",
                 interpolation_type: str = "linear", use_karras_sigmas: Optional[bool] = False,
                 sigma_min: Optional[float] = None, sigma_max: Optional[float] = None,
                 timestep_spacing: str = "linspace", timestep_type: str = "discrete", steps_offset: int = 0,
                 rescale_betas_zero_snr: bool = False):
        self.config = SimpleNamespace(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
            beta_schedule=beta_schedule, trained_betas=trained_betas, prediction_type=prediction_type,
            interpolation_type=interpolation_type, use_karras_sigmas=use_karras_sigmas, sigma_min=sigma_min,
            sigma_max=sigma_max, timestep_spacing=timestep_spacing, timestep_type=timestep_type,
            steps_offset=steps_offset, rescale_betas_zero_snr=rescale_betas_zero_snr)
        # :197-214
        if trained_betas is not None:
            self.betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                        dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} does is not implemented for {self.__class__}")
        if rescale_betas_zero_snr:
            raise NotImplementedError("rescale_betas_zero_snr is not used by SYN3R")
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).flip(0)
        timesteps = np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=float)[::-1].copy()
        timesteps = torch.from_numpy(timesteps).to(dtype=torch.float32)
        self.num_inference_steps = None
        if timestep_type == "continuous" and prediction_type == "v_prediction":
            self.timesteps = torch.Tensor([0.25 * sigma.log() for sigma in sigmas])
        else:
            self.timesteps = timesteps
        self.sigmas = torch.cat([sigmas, torch.zeros(1)])
        self.is_scale_input_called = False
        self.use_karras_sigmas = use_karras_sigmas
        self._step_index = None
        self._begin_index = None

    @classmethod
    def from_config(cls, config: dict):
        keys = SVD_XT_SCHEDULER_CONFIG.keys() | {"trained_betas"}
        return cls(**{k: v for k, v in config.items() if k in keys})

    # ------------------------------------------------------------------ schedule (host)
    @property
    def init_noise_sigma(self):
        """:248-254"""
        max_sigma = self.sigmas.max()
        if self.config.timestep_spacing in ["linspace", "trailing"]:
            return max_sigma
        return (max_sigma ** 2 + 1) ** 0.5

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def set_timesteps(self, num_inference_steps: int, device: Union[str, torch.device] = None):
        """:310-372"""
        cfg = self.config
        self.num_inference_steps = num_inference_steps
        if cfg.timestep_spacing == "linspace":
            timesteps = np.linspace(0, cfg.num_train_timesteps - 1, num_inference_steps, dtype=np.float32)[::-1].copy()
        elif cfg.timestep_spacing == "leading":
            step_ratio = cfg.num_train_timesteps // self.num_inference_steps
            timesteps = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.float32)
            timesteps += cfg.steps_offset
        elif cfg.timestep_spacing == "trailing":
            step_ratio = cfg.num_train_timesteps / self.num_inference_steps
            timesteps = (np.arange(cfg.num_train_timesteps, 0, -step_ratio)).round().copy().astype(np.float32)
            timesteps -= 1
        else:
            raise ValueError(f"{cfg.timestep_spacing} is not supported. Please make sure to choose one of "
                             "'linspace', 'leading' or 'trailing'.")
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        log_sigmas = np.log(sigmas)
        if cfg.interpolation_type == "linear":
            sigmas = np.interp(timesteps, np.arange(0, len(sigmas)), sigmas)
        elif cfg.interpolation_type == "log_linear":
            sigmas = torch.linspace(np.log(sigmas[-1]), np.log(sigmas[0]), num_inference_steps + 1).exp().numpy()
        else:
            raise ValueError(f"{cfg.interpolation_type} is not implemented. Please specify interpolation_type "
                             "to either 'linear' or 'log_linear'")
        if cfg.use_karras_sigmas:
            sigmas = self._convert_to_karras(in_sigmas=sigmas, num_inference_steps=self.num_inference_steps)
            timesteps = np.array([self._sigma_to_t(sigma, log_sigmas) for sigma in sigmas])
        sigmas = torch.from_numpy(sigmas).to(dtype=torch.float32)
        if cfg.timestep_type == "continuous" and cfg.prediction_type == "v_prediction":
            self.timesteps = torch.Tensor([0.25 * sigma.log() for sigma in sigmas]).to(device=device)
        else:
            self.timesteps = torch.from_numpy(timesteps.astype(np.float32)).to(device=device)
        self.sigmas = torch.cat([sigmas, torch.zeros(1)])  # kept on the CPU, as :372
        self._step_index = None
        self._begin_index = None

    def _sigma_to_t(self, sigma, log_sigmas):
        """:374-396"""
        log_sigma = np.log(np.maximum(sigma, 1e-10))
        dists = log_sigma - log_sigmas[:, np.newaxis]
        low_idx = np.cumsum((dists >= 0), axis=0).argmax(axis=0).clip(max=log_sigmas.shape[0] - 2)
        high_idx = low_idx + 1
        low = log_sigmas[low_idx]
        high = log_sigmas[high_idx]
        w = np.clip((low - log_sigma) / (low - high), 0, 1)
        t = (1 - w) * low_idx + w * high_idx
        return t.reshape(sigma.shape)

    def _convert_to_karras(self, in_sigmas, num_inference_steps):
        """:399-423 (rho = 7)"""
        sigma_min = self.config.sigma_min if self.config.sigma_min is not None else in_sigmas[-1].item()
        sigma_max = self.config.sigma_max if self.config.sigma_max is not None else in_sigmas[0].item()
        rho = 7.0
        ramp = np.linspace(0, 1, num_inference_steps)
        min_inv_rho = sigma_min ** (1 / rho)
        max_inv_rho = sigma_max ** (1 / rho)
        return (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho

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
        handed to another tensor meanwhile).  A few entries per slot: the forward and the time-flipped backward pass
        alternate two sets of objects."""
        cache = self.__dict__.setdefault("_prep_cache", {}).setdefault(slot, [])
        for ent in cache:
            if ent[0] is src and ent[1] == src._version:
                return ent[2]
        val = make(src)
        cache.insert(0, (src, src._version, val))
        del cache[4:]
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
