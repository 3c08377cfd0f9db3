"""One SVD (denoise-step, pass) unit on synthetic inputs — HOT LOOP B of bench.py.

Follows the body of the reference's denoising loop for one pass of the "Replace" variant
(model/SVD_2pass_prob_uncertain.py:661-742): duplicate the latents for classifier-free guidance,
`scale_model_input`, concatenate the conditioning-image latents on the channel axis, UNet forward,
per-frame guidance combine, `step_interp_prob_uncertain`.  Synthetic data as SURVEY.md §8d
(no checkpoint is reachable offline: seeded N(0, 0.02) weights of the SVD-XT architecture).
"""
from __future__ import annotations

import torch

from ..schedulers.scheduling_euler_discrete import SVD_XT_SCHEDULER_CONFIG, EulerDiscreteScheduler
from ..unet import ops
from ..unet.model import UNetSpatioTemporalConditionModel

MFMA_F16_PEAK_TFLOPS = 2500.0   # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)


def cfg_combine(noise_pred: torch.Tensor, guidance_scale: torch.Tensor) -> torch.Tensor:
    """`uncond + gs * (cond - uncond)` in the model dtype, as SVD_2pass_prob_uncertain.py:709-711."""
    uncond, cond = noise_pred.chunk(2)
    return uncond + guidance_scale * (cond - uncond)


class SvdStepBench:
    def __init__(self, frames: int, dev: torch.device, seed: int = 1234, h: int = 72, w: int = 128, unet=None):
        self.F, self.h, self.w, self.dev = frames, h, w, dev
        self.unet = unet if unet is not None else UNetSpatioTemporalConditionModel().init_random(dev, seed=seed)
        self.sch = EulerDiscreteScheduler(**SVD_XT_SCHEDULER_CONFIG)
        self.sch.set_timesteps(100)
        g = torch.Generator(device=dev).manual_seed(seed)
        rn = lambda *s: torch.randn(*s, generator=g, device=dev)
        F = frames
        self.latents = (rn(1, F, 4, h, w) * float(self.sch.init_noise_sigma)).half()
        self.image_latents = rn(1, F, 4, h, w).half().repeat(2, 1, 1, 1, 1)
        self.image_latents[0].zero_()                                   # CFG: unconditional half is zeros
        cond = rn(1, F, 4, h, w) / 5.6
        self.temp_cond = torch.cat([torch.zeros_like(cond), cond], 0)  # index 1 = conditioning latents
        self.ehs = torch.cat([torch.zeros(1, 1, 1024, device=dev), rn(1, 1, 1024)], 0).half()
        self.added = torch.tensor([[6.0, 127.0, 0.02]] * 2, device=dev).half()
        m = torch.rand(1, F - 2, 1, h, w, generator=g, device=dev)
        self.mask = m.expand(1, F - 2, 4, h, w).contiguous()
        self.lambda_ts = (torch.rand(100, F, generator=g, device=dev) > 0.5).double().cpu()     # host-side schedule, as the pipeline keeps it
        self.guidance = torch.linspace(1.0, 3.0, F, device=dev).half()[None, :, None, None, None]
        self.i = 0
        self.flops_per_unit = None

    def step_pass(self) -> torch.Tensor:
        i = self.i % 100
        self.i += 1
        t = self.sch.timesteps[i]
        x = torch.cat([self.latents] * 2)
        x = self.sch.scale_model_input(x, t, step_i=i).half()
        x = torch.cat([x, self.image_latents], dim=2)
        fwd = self.unet.forward_graphed if getattr(self, "use_graphs", False) else self.unet
        noise_pred = fwd(x, t, self.ehs, self.added)[0]
        noise_pred = cfg_combine(noise_pred, self.guidance)
        out = self.sch.step_interp_prob_uncertain(noise_pred, t, self.latents, self.temp_cond, self.mask,
                                                  self.lambda_ts, step_i=i)
        return out.prev_sample

    def step_pass_post(self) -> torch.Tensor:
        """One (step, pass) unit of the "Post" variant (model/SVD_2pass_prob_uncertain_post.py:700-800): the four
        guidance-tile forwards + gradient step, then the CFG forward and the plain Euler step — the variant
        `bash_scripts/batch_llff_train.sh:39` runs (at F = 25)."""
        from .svd_2pass import StableVideoDiffusionPipeline
        if getattr(self, "_post_pipe", None) is None:
            self._post_pipe = StableVideoDiffusionPipeline(None, None, self.unet, self.sch, variant="post", device=self.dev)
            self._post_pipe._guidance_scale = self.guidance
        i = self.i % 100
        self.i += 1
        return self._post_pipe._pass_post(i, self.sch.timesteps[i], self.latents, self.image_latents, self.ehs, self.added,
                                          self.temp_cond, self.mask, self.lambda_ts, True)

    def step_both(self, variant: str = "replace"):
        """BOTH passes of one denoising step the way the pipelines run them (`merge_passes`): the forward- and the
        time-flipped backward pass stacked into one UNet launch sequence, the two scheduler steps, the blend
        (SVD_2pass_prob_uncertain.py:661-742 / …post.py:679-828).  = 2 (step, pass) units."""
        from .svd_2pass import StableVideoDiffusionPipeline
        key = "_both_" + variant
        st = getattr(self, key, None)
        if st is None:
            pipe = StableVideoDiffusionPipeline(None, None, self.unet, self.sch, variant=variant, device=self.dev)
            pipe._guidance_scale = self.guidance
            g = torch.Generator(device=self.dev).manual_seed(77)
            F, h, w = self.F, self.h, self.w
            img_end = torch.randn(1, F, 4, h, w, generator=g, device=self.dev).half().repeat(2, 1, 1, 1, 1)
            img_end[0].zero_()
            ehs_end = torch.cat([torch.zeros(1, 1, 1024, device=self.dev), torch.randn(1, 1, 1024, generator=g, device=self.dev)], 0).half()
            cond_bw, mask_bw, lam_bw = self.temp_cond.flip(dims=[1]), self.mask.flip(dims=[1]), self.lambda_ts.flip(dims=[1])
            post = variant == "post"
            ops2 = ((self.temp_cond, self.mask, self.lambda_ts, pipe._tile_operands(self.temp_cond, self.mask) if post else None),
                    (cond_bw, mask_bw, lam_bw, pipe._tile_operands(cond_bw, mask_bw) if post else None))
            st = dict(pipe=pipe, img4=torch.cat([self.image_latents, img_end]).contiguous(), ehs4=torch.cat([self.ehs, ehs_end]).contiguous(),
                      added4=torch.cat([self.added, self.added]).contiguous(), ops2=ops2,
                      tile_ctx=(self.ehs[0:1].expand(4, -1, -1), self.added[0:1].expand(4, -1).contiguous(), None),
                      weight_fw=torch.linspace(1, 0, F, device=self.dev).half()[None, :, None, None, None])
            setattr(self, key, st)
        i = self.i % 100
        self.i += 1
        t = self.sch.timesteps[i]
        lat = (self.latents, self.latents.flip(dims=[1]))
        p = st["pipe"]
        if variant == "post":
            fw, bw = p._merged_post(i, t, lat, st["img4"], st["ehs4"], st["added4"], st["ops2"], True, st["tile_ctx"])
        else:
            fw, bw = p._streamed_replace(i, t, lat, st["img4"], st["ehs4"], st["added4"], st["ops2"], True)
        return st["weight_fw"] * fw + (1 - st["weight_fw"]) * bw.flip(dims=[1])

    def count_flops(self) -> dict:
        """Algorithmic FLOPs of one unit, counted from the launched contractions (2*M*N*K)."""
        ops.FLOPS.update(enabled=True, gemm=0.0, attn=0.0)
        self.step_pass()
        torch.cuda.synchronize(self.dev)
        ops.FLOPS["enabled"] = False
        self.flops_per_unit = dict(gemm=ops.FLOPS["gemm"], attn=ops.FLOPS["attn"])
        return self.flops_per_unit

    def roofline(self, kernels: dict, units: int):
        """MFMA roofline of the dominant UNet kernel from bench.py's HIP-event kernel trace."""
        if self.flops_per_unit is None:
            self.count_flops()
        # the tracer names template kernels by their source text: fold all k_gemm instantiations together
        agg = {}
        for name, (calls, ms) in kernels.items():
            key = "k_gemm" if name.startswith("k_gemm") else name
            c0, m0 = agg.get(key, (0, 0.0))
            agg[key] = (c0 + calls, m0 + ms)
        cand = {"k_gemm": "gemm", "k_attn_spatial": "attn"}
        best = None
        for name in cand:
            if name in agg and (best is None or agg[name][1] > agg[best][1]):
                best = name
        if best is None:
            return None
        calls, ms = agg[best]
        flops = self.flops_per_unit[cand[best]] * units
        ach = flops / (ms / 1e3) / 1e12
        traffic, source = _pmc_traffic(best)
        busy, _ = _pmc_traffic(best, field="mfma_busy")
        ratio, _ = _pmc_traffic(best, field="traffic_ratio")       # counted HBM bytes / algorithmic bytes: family and worst kernel
        alg_b, _ = _pmc_traffic(best, field="algorithmic_bytes_per_launch")
        return dict(bound="mfma", kernel=best, achieved=round(ach, 1), peak=MFMA_F16_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(ach / MFMA_F16_PEAK_TFLOPS, 4), traffic=traffic, traffic_source=source, mfma_busy=busy,
                    traffic_ratio=ratio, algorithmic_bytes_per_launch=alg_b,
                    avg_ms=round(ms / calls, 4), calls=calls, algorithmic_flops_per_unit=self.flops_per_unit)


def source_id() -> str:
    """sha256 (first 16 hex digits) over the kernel sources the running library was built from (csrc/*.hip, *.h and the
    C-ABI header): identifies WHICH build a committed PMC pass measured."""
    import hashlib
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    h = hashlib.sha256()
    for f in sorted(list((root / "csrc").glob("*.hip")) + list((root / "csrc").glob("*.h")) + list((root.parent / "include").glob("*.h"))):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def _pmc_traffic(kernel: str, field: str = "hbm_bytes_per_launch"):
    """(HBM bytes per launch of `kernel`, where the number comes from) out of the committed rocprofv3 PMC passes
    (profiles/*/traffic.json; bench.py cannot run the profiler on itself).  The file records the `source_id` of the
    kernel sources it profiled: if that differs from the running build the number is STALE and (None, reason) is
    returned — `traffic: null` in the bench line rather than a silently outdated figure."""
    import json
    from pathlib import Path
    root = Path(__file__).resolve().parents[2] / "profiles"
    cur = source_id()
    stale = None
    for f in sorted(root.glob("r*/traffic.json"), reverse=True):
        try:
            doc = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        rec = doc.get(kernel)
        if not rec or field not in rec:
            continue
        if doc.get("_source_id") != cur:
            stale = stale or (f"{f.relative_to(root.parent)} was measured on kernel sources {doc.get('_source_id', 'unrecorded')}, "
                              f"the running build is {cur}: not reported")
            continue
        return rec[field], (f"{f.relative_to(root.parent)} (sources {cur}): PMC FETCH_SIZE x2 (gfx950 correction) + "
                                             "WRITE_SIZE, averaged over the launches of the profiled run")
    return None, stale
