"""`StableVideoDiffusionPipeline.from_pretrained(<local directory>)` (the reference's loader call, model/diffusionGS.py:1089):
the CLIP preprocessing restatement against the reference function's fixture, the pieces of the loader that run without a GPU
on a tiny diffusers-layout checkpoint written by oracle/tiny_checkpoint.py, and - on the GPU - one denoising step through the
assembled pipeline."""
import json

import numpy as np
import pytest
import torch

from oracle import golden_inputs as GI
from oracle import tiny_checkpoint as TC


def test_clip_preprocessing_equals_reference(golden_dir):
    from syn3r_amd.pipeline.clip import clip_pixel_values
    g = np.load(golden_dir / "clip_preprocess.npz")
    for tag, (h, w) in GI.CLIP_CASES.items():
        px = clip_pixel_values(GI.clip_image(h, w)).numpy()
        assert px.shape == (1, 3, 224, 224)
        np.testing.assert_allclose(px[..., ::2, ::2], g[f"px_{tag}"], atol=2e-5)
        assert abs(float(np.abs(px).mean()) - float(g[f"mean_abs_{tag}"])) < 1e-6
        # a float HWC image in [0, 1] and a CHW tensor are the same image
        f = GI.clip_image(h, w).astype(np.float32) / 255.0
        np.testing.assert_allclose(clip_pixel_values(f).numpy(), px, atol=1e-6)
        np.testing.assert_allclose(clip_pixel_values(torch.from_numpy(f).permute(2, 0, 1)).numpy(), px, atol=1e-6)
        # the range of a float image is an ARGUMENT (a dark [0, 255] float image has nothing to guess it from)
        f255 = GI.clip_image(h, w).astype(np.float32)
        np.testing.assert_allclose(clip_pixel_values(f255, value_range=255.0).numpy(), px, atol=1e-6)
        dark = (f255 / 255.0).clip(0, 1.4)                     # all values <= 1.4: still read as [0, 255] when told so
        np.testing.assert_allclose(clip_pixel_values(dark, value_range=255.0).numpy(),
                                   clip_pixel_values(dark / 255.0).numpy(), atol=1e-6)
    import pytest
    with pytest.raises(ValueError):
        clip_pixel_values(GI.clip_image(32, 32), value_range=1.0)
    with pytest.raises(ValueError):
        clip_pixel_values(GI.clip_image(32, 32), value_range=2.0)


def test_loader_pieces_on_a_tiny_checkpoint(tmp_path):
    """Everything `from_pretrained` does up to the device boundary: the directory layout is found, the scheduler is built
    from scheduler/scheduler_config.json (sigma_max 500 there, not the hard-coded 700), CLIP comes up through transformers with
    the directory's own mean / std, the HIP modules find their config and weights - and then refuse the CPU (no fallback)."""
    from syn3r_amd import _lib
    from syn3r_amd.pipeline.clip import ClipImageEncoder
    from syn3r_amd.pipeline.svd_2pass import StableVideoDiffusionPipeline
    from syn3r_amd.schedulers.scheduling_euler_discrete import EulerDiscreteScheduler
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    d = TC.write(tmp_path / "svd", projection_dim=64)
    for sub in ("unet", "vae", "scheduler", "image_encoder", "feature_extractor"):
        assert (d / sub).is_dir()
    sch = EulerDiscreteScheduler.from_config(json.loads((d / "scheduler" / "scheduler_config.json").read_text()))
    sch.set_timesteps(10)
    assert abs(float(sch.sigmas[0]) - 500.0) < 1e-3 and abs(float(sch.init_noise_sigma) - (500.0 ** 2 + 1) ** 0.5) < 1e-2
    enc = ClipImageEncoder.from_pretrained(d / "image_encoder", d / "feature_extractor", device="cpu", dtype=torch.float32)
    assert enc.mean == tuple(TC.CLIP_MEAN) and enc.std == tuple(TC.CLIP_STD)
    e = enc(GI.clip_image(*GI.CLIP_CASES["b"])).image_embeds
    assert e.shape == (1, 64) and bool(torch.isfinite(e).all())
    with pytest.raises(FileNotFoundError):
        UNetSpatioTemporalConditionModel.from_pretrained(str(d / "unet"), "cpu", variant="bf16x")        # no such weight file ...
    (d / "unet" / "diffusion_pytorch_model.safetensors").write_bytes((d / "unet" / "diffusion_pytorch_model.fp16.safetensors").read_bytes())
    with pytest.raises(_lib.Syn3rError):                                                               # ... the plain name is the fallback
        StableVideoDiffusionPipeline.from_pretrained(d, torch_dtype=torch.float16, variant="fp16", device="cpu")
    with pytest.raises(FileNotFoundError):
        StableVideoDiffusionPipeline.from_pretrained("stabilityai/stable-video-diffusion-img2vid-xt")   # never fetched by name
    with pytest.raises(FileNotFoundError):
        StableVideoDiffusionPipeline.from_pretrained(tmp_path)                                          # not a checkpoint layout


@pytest.mark.gpu
def test_from_pretrained_runs_a_step(gpu, tmp_path):
    """The assembled pipeline (HIP UNet + HIP VAE + directory scheduler + CLIP) runs a two-step, two-pass call end to end and
    equals the same modules assembled by hand (`DiffusionGS.svd_render` hands a directory to the same loader)."""
    from syn3r_amd.pipeline.svd_2pass import StableVideoDiffusionPipeline
    from syn3r_amd.schedulers.scheduling_euler_discrete import EulerDiscreteScheduler
    d = TC.write(tmp_path / "svd")
    pipe = StableVideoDiffusionPipeline.from_pretrained(d, torch_dtype=torch.float16, variant="fp16", pipeline="replace", device=gpu)
    assert pipe.variant == "replace" and pipe.scheduler_config["sigma_max"] == 500.0
    h, w = 64, 128          # 8 x 16 latents: the VAE attention kernels want a multiple of 64 tokens
    g = torch.Generator().manual_seed(0)
    imgs = [torch.rand(3, h, w, generator=g) for _ in range(25)]
    mask = (torch.rand(23, h // 8, w // 8, generator=g) > 0.5).float()
    lam = (torch.rand(2, 25, generator=g) > 0.5).double()
    kw = dict(temp_cond=imgs[1:], mask=mask, lambda_ts=lam, height=h, width=w, num_frames=25, decode_chunk_size=8,
              num_inference_steps=2, output_type="latent", dtype=torch.float16,
              aug_noise=torch.randn(1, 3, h, w, generator=g), latents=torch.randn(1, 25, 4, h // 8, w // 8, generator=g))
    a = pipe([imgs[0]], **kw).frames.float()
    assert a.shape == (1, 25, 4, h // 8, w // 8) and bool(torch.isfinite(a).all())
    assert abs(float(pipe.scheduler.sigmas[0]) - 500.0) < 1e-3          # the call ran on the directory's schedule (set_timesteps(2))
    hand = StableVideoDiffusionPipeline(pipe.vae, pipe.image_encoder, pipe.unet, EulerDiscreteScheduler.from_config(pipe.scheduler_config),
                                        variant="replace", device=gpu)
    assert torch.equal(hand([imgs[0]], **kw).frames.float(), a)
