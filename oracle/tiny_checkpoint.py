"""Test infrastructure: write a tiny checkpoint directory in the diffusers layout of stabilityai/stable-video-diffusion-img2vid-xt
(`unet/`, `vae/`, `scheduler/`, `image_encoder/`, `feature_extractor/`, model_index.json) with seeded weights, for the loader
tests of `StableVideoDiffusionPipeline.from_pretrained` (reference call: model/diffusionGS.py:1089).  Nothing here is product
code; the weights are the name-keyed seeded tensors of oracle/unet_weights.py."""
from __future__ import annotations

import json
from pathlib import Path

import torch

from . import unet_weights as UW
from . import vae_weights as VW

SCHEDULER_CONFIG = dict(_class_name="EulerDiscreteScheduler", _diffusers_version="0.24.0.dev0", num_train_timesteps=1000,
                        beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", prediction_type="v_prediction",
                        interpolation_type="linear", use_karras_sigmas=True, sigma_min=0.002, sigma_max=500.0,   # NOT 700:
                        timestep_spacing="leading", timestep_type="continuous", steps_offset=1,                  # proves it is read
                        rescale_betas_zero_snr=False, skip_prk_steps=True, set_alpha_to_one=False, clip_sample=False)
CLIP_MEAN, CLIP_STD = [0.5, 0.45, 0.4], [0.25, 0.26, 0.27]          # not CLIP's own: proves preprocessor_config.json is read


def write(directory, variant: str = "fp16", projection_dim: int = 1024) -> Path:
    from safetensors.torch import save_file
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    from syn3r_amd.vae import AutoencoderKLTemporalDecoder
    d = Path(directory)
    suffix = f".{variant}" if variant else ""
    ucfg = dict(UW.PIPELINE_CONFIG, cross_attention_dim=projection_dim)
    (d / "unet").mkdir(parents=True, exist_ok=True)
    (d / "unet" / "config.json").write_text(json.dumps(dict(ucfg, _class_name="UNetSpatioTemporalConditionModel")))
    sd = UW.make_state_dict(UNetSpatioTemporalConditionModel(**ucfg).parameter_shapes(), seed=11)
    save_file({k: v.to(torch.float16).contiguous() for k, v in sd.items()}, str(d / "unet" / f"diffusion_pytorch_model{suffix}.safetensors"))
    vcfg = dict(VW.PIPELINE_VAE_CONFIG, scaling_factor=0.18215, force_upcast=True)
    (d / "vae").mkdir(exist_ok=True)
    (d / "vae" / "config.json").write_text(json.dumps(dict(vcfg, _class_name="AutoencoderKLTemporalDecoder")))
    sd = UW.make_state_dict(AutoencoderKLTemporalDecoder(**VW.PIPELINE_VAE_CONFIG).parameter_shapes(), seed=12)
    save_file({k: v.to(torch.float16).contiguous() for k, v in sd.items()}, str(d / "vae" / f"diffusion_pytorch_model{suffix}.safetensors"))
    (d / "scheduler").mkdir(exist_ok=True)
    (d / "scheduler" / "scheduler_config.json").write_text(json.dumps(SCHEDULER_CONFIG))
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    torch.manual_seed(13)
    clip = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                                                          image_size=224, patch_size=32, projection_dim=projection_dim))
    clip.save_pretrained(str(d / "image_encoder"), safe_serialization=True)
    (d / "feature_extractor").mkdir(exist_ok=True)
    (d / "feature_extractor" / "preprocessor_config.json").write_text(json.dumps(dict(
        image_mean=CLIP_MEAN, image_std=CLIP_STD, do_normalize=True, size=dict(shortest_edge=224), feature_extractor_type="CLIPFeatureExtractor")))
    (d / "model_index.json").write_text(json.dumps(dict(
        _class_name="StableVideoDiffusionPipeline", unet=["diffusers", "UNetSpatioTemporalConditionModel"],
        vae=["diffusers", "AutoencoderKLTemporalDecoder"], scheduler=["diffusers", "EulerDiscreteScheduler"],
        image_encoder=["transformers", "CLIPVisionModelWithProjection"], feature_extractor=["transformers", "CLIPImageProcessor"])))
    return d
