"""Measured end-to-end runs of the path at the reference's sizes (bench.py sub-benchmarks, SURVEY.md §8d composite).

  * `measure_svd_render`  one complete `StableVideoDiffusionPipeline.__call__` — 26 VAE encodes, 100 denoising steps x 2
                          passes on [*,25,8,72,128] latents, chunked 25-frame decode — as `DiffusionGS.svd_render` issues it
                          (model/diffusionGS.py:1088-1111, model/SVD_2pass_prob_uncertain_post.py:544-848);
  * `measure_schedule`    a scaled fern-like schedule THROUGH `DiffusionGS.run(1)`: initial training, ONE view pair
                          (render, perturb-and-select, warps, fusion, lambda search, svd_render), finetune on the pseudo-views.

Weights are seeded (no checkpoint is reachable offline); CLIP is out of scope (§2): a stand-in embedder.
Wall-clock is taken with the device idle before and after; the device-side kernel time of a denoising step is traced
over a few steps of a separate short call (every launch timed: ~2 % overhead, kept out of the wall-clock run) and gives
the host-gap fraction  1 - sum(kernel time) / wall  of the hot loop.
"""
from __future__ import annotations

import math
import time
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from . import _lib as L
from . import orchestrator as O


class StandInClip:
    """CLIP vision tower stand-in (out of scope): image -> deterministic `.image_embeds` [1, 1024]."""

    def __call__(self, image):
        a = image.detach().float().cpu().numpy() if isinstance(image, torch.Tensor) else np.asarray(image, dtype=np.float32)
        g = torch.Generator().manual_seed(int(abs(float(a.sum()))) % 100003)
        return SimpleNamespace(image_embeds=torch.randn(1, 1024, generator=g))


def full_size_components(dev, unet=None, seed: int = 0) -> dict:
    """`svd_components` of the reference's sizes: the 1.52 B-parameter UNet and the SVD temporal-decoder VAE
    (autoencoder_kl_temporal_decoder.py defaults of the SVD-XT checkpoint), seeded weights, on the HIP operators."""
    from .unet.model import UNetSpatioTemporalConditionModel
    from .vae import AutoencoderKLTemporalDecoder
    if unet is None:
        unet = UNetSpatioTemporalConditionModel().init_random(dev, seed=seed)
    vae = AutoencoderKLTemporalDecoder(block_out_channels=(128, 256, 512, 512), down_block_types=("DownEncoderBlock2D",) * 4,
                                       layers_per_block=2, sample_size=768).init_random(dev, seed=seed + 1)
    return dict(vae=vae, image_encoder=StandInClip(), unet=unet, dtype=torch.float16)


def _timed(fn, dev):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize(dev)
    return out, time.perf_counter() - t0


def measure_svd_render(components: dict, variant: str, dev, steps: int = 100, F: int = 25, seed: int = 0) -> dict:
    """Wall-clock of one pipeline call and its three stages (VAE encodes, denoising loop, chunked decode)."""
    from .pipeline.svd_2pass import StableVideoDiffusionPipeline
    from .schedulers.scheduling_euler_discrete import SVD_XT_SCHEDULER_CONFIG, EulerDiscreteScheduler
    c = components
    pipe = StableVideoDiffusionPipeline(c["vae"], c["image_encoder"], c["unet"], EulerDiscreteScheduler(**SVD_XT_SCHEDULER_CONFIG),
                                        variant=variant, device=dev)
    rng = np.random.default_rng(seed)
    H, W = 576, 1024
    img = lambda: torch.from_numpy(rng.random((3, H, W), dtype=np.float32)).to(dev)
    image, temp_cond = [img()], [img() for _ in range(F - 1)]
    masks = torch.from_numpy((rng.random((F - 2, H // 8, W // 8)) > 0.5).astype(np.float32))

    def call(n_steps, **kw):
        lam = O.search_hypers_v2(masks, diffusion_steps=n_steps) if F == 25 else torch.ones(n_steps, F, dtype=torch.float64)
        return pipe(image, temp_cond=temp_cond, mask=masks, lambda_ts=lam, num_frames=F, decode_chunk_size=8,
                    num_inference_steps=n_steps, output_type="np", dtype=c.get("dtype", torch.float16), **kw)

    # stage walls (encode / denoise / decode) through the pipeline's own stage methods, timed with the device drained
    stage = {}
    enc0, den0, dec0 = pipe._encode_vae_image, pipe.denoise, pipe.decode_latents

    def wrap(name, fn):
        def inner(*a, **k):
            out, dt = _timed(lambda: fn(*a, **k), dev)
            stage[name] = stage.get(name, 0.0) + dt
            return out
        return inner

    call(1)                                                   # warm-up: workspaces, caches
    pipe._encode_vae_image, pipe.denoise, pipe.decode_latents = wrap("vae_encode_s", enc0), wrap("denoise_s", den0), wrap("vae_decode_s", dec0)
    out, wall = _timed(lambda: call(steps), dev)
    pipe._encode_vae_image, pipe.denoise, pipe.decode_latents = enc0, den0, dec0
    frames = np.asarray(out.frames[0])
    assert frames.shape == (F, H, W, 3) and np.isfinite(frames).all()
    res = dict(variant=variant, frames=F, steps=steps, wall_s=round(wall, 2), **{k: round(v, 2) for k, v in stage.items()},
               denoise_ms_per_step_pass=round(1e3 * stage["denoise_s"] / (2 * steps), 2),
               peak_memory_gb=round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1))
    return res


def measure_denoise_gap(components: dict, variant: str, dev, steps: int = 3, F: int = 25, seed: int = 0) -> dict:
    """Host-gap fraction of the denoising loop: `steps` steps of `denoise` on prepared latents, once untimed-per-launch for the
    wall-clock and once with every launch timed by HIP events for the sum of kernel time."""
    from .pipeline.svd_2pass import StableVideoDiffusionPipeline
    from .schedulers.scheduling_euler_discrete import SVD_XT_SCHEDULER_CONFIG, EulerDiscreteScheduler
    c = components
    pipe = StableVideoDiffusionPipeline(None, None, c["unet"], EulerDiscreteScheduler(**SVD_XT_SCHEDULER_CONFIG), variant=variant, device=dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    h, w = 72, 128
    lat = (rn(1, F, 4, h, w) * 700.0).half()
    il = lambda: torch.cat([torch.zeros(1, F, 4, h, w, device=dev), rn(1, F, 4, h, w)]).half()
    emb = lambda: torch.cat([torch.zeros(1, 1, 1024, device=dev), rn(1, 1, 1024)]).half()
    added = torch.tensor([[6.0, 127.0, 0.02]] * 2, device=dev).half()
    cond = torch.cat([torch.zeros(1, F, 4, h, w, device=dev), rn(1, F, 4, h, w) / 5.6])
    m = torch.rand(1, F - 2, 1, h, w, generator=g, device=dev).expand(1, F - 2, 4, h, w).contiguous()
    lam = (torch.rand(steps, F, generator=g, device=dev) > 0.5).double().cpu()
    args = (lat, il(), il(), emb(), emb(), added, cond, m, lam, steps)
    # The gap is a property of ONE launch sequence (wall-clock against the sum of its kernels' durations): measured on the
    # one-stream order.  The product's default puts the independent sequences of a step on their own streams (svd_2pass.py), where
    # kernels overlap and the sum of their durations exceeds the wall-clock: its wall-clock is reported beside it.
    streams = pipe.two_streams
    pipe.two_streams = False
    pipe.denoise(*args)                                                       # warm-up
    _, wall = _timed(lambda: pipe.denoise(*args), dev)
    with L.kernel_trace() as tr:
        pipe.denoise(*args)
        torch.cuda.synchronize(dev)
    ksum = sum(v[1] for v in tr.result.values()) / 1e3
    launches = int(sum(v[0] for v in tr.result.values()))
    out = dict(variant=variant, frames=F, steps=steps, wall_ms_per_step_pass=round(1e3 * wall / (2 * steps), 2),
               kernel_ms_per_step_pass=round(1e3 * ksum / (2 * steps), 2), launches_per_step=launches // steps,
               host_gap_frac=round(max(0.0, 1.0 - ksum / wall), 4))
    pipe.two_streams = streams
    if streams:
        pipe.denoise(*args)
        _, wall2 = _timed(lambda: pipe.denoise(*args), dev)
        out["wall_ms_per_step_pass_streams"] = round(1e3 * wall2 / (2 * steps), 2)
    return out


def synthetic_scene(dev, N: int, H: int, W: int, V: int, iterations: int, model_path: str, seed: int = 0, lambda_dssim: float = 0.2):
    """A seeded Gaussian cloud seen by V cameras on a baseline: their renders are the input views, a jittered copy of the
    cloud is the model to fit (the scene of tests/ and launch.py at a chosen size)."""
    from .gs import Camera, GaussianModel, GSTrainer, OptimizationParams
    from .synthetic import synthetic_gaussians
    m, s, q, o, sh = synthetic_gaussians(N, seed=seed)
    logit = torch.log(o.clamp(1e-3, 1 - 1e-3) / (1 - o.clamp(1e-3, 1 - 1e-3)))
    truth = GaussianModel(m, torch.log(s), q, logit, sh, device=dev)
    f = W / (2 * math.tan(math.radians(30)))
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=np.float32)
    poses = []
    for v in range(V):
        p = np.eye(4, dtype=np.float32)
        p[0, 3] = -0.1 + 0.2 * v / max(V - 1, 1)
        poses.append(p)
    gt = GSTrainer(truth, [Camera.from_w2c(poses[0], K, H, W, data_device=dev)])
    views = [gt.render_view(Camera.from_w2c(p, K, H, W, data_device=dev))["render"].detach().clamp(0, 1) for p in poses]
    cams = [Camera.from_w2c(p, K, H, W, image=v, data_device=dev) for p, v in zip(poses, views)]
    g = torch.Generator().manual_seed(seed + 1)
    model = GaussianModel(m + 0.002 * torch.randn(m.shape, generator=g), torch.log(s), q, logit, sh, device=dev)
    opt = OptimizationParams(iterations=iterations, lambda_dssim=lambda_dssim, seed=seed)
    return GSTrainer(model, cams, opt, model_path=model_path, checkpoint_iterations=[iterations])


def measure_schedule(components: dict, dev, tmp_dir: str, variant: str = "post", N: int = 200_000, H: int = 1080, W: int = 1920,
                     iterations: int = 500, steps: int = 100, seed: int = 0) -> dict:
    """The scaled schedule through the orchestrator: `DiffusionGS(trainer, num_input_views=2, densify_type=
    'interpolate_loop0_gs').run(1)` = `iterations` initial training steps, ONE view pair densified (one full-size
    svd_render), `iterations` finetune steps on real + pseudo views.  Returns the wall-clock and its stages."""
    from .diffusionGS import DiffusionGS
    trainer = synthetic_scene(dev, N, H, W, 2, iterations, tmp_dir, seed=seed)
    args = SimpleNamespace(cam_confidence=0.05, pseudo_cam_sampling_rate=0.02, fps_keyframe_sampling=0,
                           densify_type="interpolate_loop0_gs", num_views_for_pcd_densification=1)
    d = DiffusionGS(trainer, num_input_views=2, save_dir=tmp_dir, diffusion_type="2PassProbUncertainPost" if variant == "post" else "2PassProbUncertain",
                    interp_type="backward_warp", input_args=args, svd_components=components, num_inference_steps=steps)
    stage = {}

    def wrap(obj, name, key):
        fn = getattr(obj, name)

        def inner(*a, **k):
            out, dt = _timed(lambda: fn(*a, **k), dev)
            stage[key] = stage.get(key, 0.0) + dt
            return out
        setattr(obj, name, inner)

    wrap(trainer, "training", "training_s")
    wrap(trainer, "finetune", "finetune_s")
    wrap(d, "svd_render", "svd_render_s")
    wrap(d, "_interpolate_between_gs_v3", "view_pair_s")
    np.random.seed(seed)
    # densification off: the timed iterations are steady ones (FSGS' clone / split / prune change the pair count run to run)
    tr0, ft0 = trainer.training, trainer.finetune
    trainer.training = lambda *a, **k: tr0(*a, **dict(k, disable_densification=True))
    trainer.finetune = lambda *a, **k: ft0(*a, **dict(k, disable_densification=True))
    _, wall = _timed(lambda: d.run(1), dev)
    its = 2 * iterations
    t_train = stage.get("training_s", 0.0) + stage.get("finetune_s", 0.0)
    orch = stage.get("view_pair_s", 0.0) - stage.get("svd_render_s", 0.0)
    return dict(variant=variant, gaussians=N, resolution=f"{W}x{H}", trainer_iterations=its, svd_renders=1, steps=steps,
                wall_s=round(wall, 2), **{k: round(v, 2) for k, v in stage.items()},
                orchestrator_other_s=round(orch, 2),                               # renders, perturb-and-select, warps, fusion, lambda search
                trainer_iters_per_s=round(its / t_train, 1) if t_train > 0 else None,
                truncated_renders=int(trainer.truncated_renders), pseudo_views=24 + 1)
