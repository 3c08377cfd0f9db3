"""Error levels of the end-to-end pipeline parity case (tests/golden/pipeline_unet.npz) (developer tool)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from oracle import pipeline_mocks as PM
from oracle import unet_weights as UW
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd.pipeline.svd_2pass import StableVideoDiffusionPipeline
from syn3r_amd.schedulers.scheduling_euler_discrete import SVD_XT_SCHEDULER_CONFIG, EulerDiscreteScheduler
from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
gpu = torch.device("cuda", 0)
gold = np.load(ROOT / "tests/golden/pipeline_unet.npz")
unet = UNetSpatioTemporalConditionModel(**UW.PIPELINE_CONFIG)
unet.load_state_dict(UW.make_state_dict(unet.parameter_shapes(), seed=3), gpu)
inp = PM.pipeline_inputs(seed=1)
for variant in ("replace", "post"):
    pipe = StableVideoDiffusionPipeline(PM.MockVAE(), PM.MockImageEncoder(), unet, EulerDiscreteScheduler(**SVD_XT_SCHEDULER_CONFIG),
                                        variant=variant, device=gpu)
    lat = pipe([im.to(gpu) for im in inp["image"]], temp_cond=[t.to(gpu) for t in inp["temp_cond"]], mask=inp["mask"].clone(),
               lambda_ts=inp["lambda_ts"], num_frames=25, decode_chunk_size=8, num_inference_steps=2, latent_num=1,
               latents=inp["latents"].clone(), output_type="latent", dtype=torch.float16, aug_noise=inp["noise"]).frames
    a, g = lat.float().cpu().numpy()[..., ::3, ::3], gold[variant]
    err, scale = np.abs(a - g), np.abs(g).max()
    print(f"{variant}: scale {scale:.3f}  mean err {err.mean() / scale:.2e}  max {err.max() / scale:.2e}  "
          f"> 2e-2: {(err > 2e-2 * scale).mean():.2e}  > 5e-3: {(err > 5e-3 * scale).mean():.2e}")

from oracle import vae_weights as VW
from syn3r_amd.vae import AutoencoderKLTemporalDecoder
gf = np.load(ROOT / "tests/golden/pipeline_unet_vae.npz")["frames"]
vae = AutoencoderKLTemporalDecoder(**VW.PIPELINE_VAE_CONFIG)
vae.load_state_dict(UW.make_state_dict(vae.parameter_shapes(), seed=11), gpu)
pipe = StableVideoDiffusionPipeline(vae, PM.MockImageEncoder(), unet, EulerDiscreteScheduler(**SVD_XT_SCHEDULER_CONFIG),
                                    variant="replace", device=gpu)
inp = PM.pipeline_inputs(seed=2)
fr = pipe([im.to(gpu) for im in inp["image"]], temp_cond=[t.to(gpu) for t in inp["temp_cond"]], mask=inp["mask"].clone(),
          lambda_ts=inp["lambda_ts"], num_frames=25, decode_chunk_size=8, num_inference_steps=2, latent_num=1,
          latents=inp["latents"].clone(), output_type="np", dtype=torch.float16, aug_noise=inp["noise"]).frames[0]
err = np.abs(np.asarray(fr, dtype=np.float32)[:, ::16, ::16] - gf)
print(f"frames (replace, with the VAE): mean err {err.mean():.2e}  max {err.max():.2e}  > 4e-2: {(err > 4e-2).mean():.2e}  > 1e-2: {(err > 1e-2).mean():.2e}")
