"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (this container only).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Imports the reference's own Python from /root/reference (SURVEY.md Appendix B
recipe: two attribute shims for pip version skew, `.to('cuda')` redirected to
the CPU for inverse_warp only), feeds it seeded synthetic inputs and stores the
inputs' recipe (seed + shapes) and the reference's outputs.  Nothing from the
reference travels to the GPU box: only these data fixtures do.  The inputs are
rebuilt in tests from `golden_inputs.py` (same seeds), so the fixtures hold
outputs only.
"""
from __future__ import annotations

import contextlib
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")
sys.dont_write_bytecode = True
REF = Path("/root/reference")
sys.path[:0] = [str(REF / "thirdparty/diffusers/src"), str(REF)]

import torch  # noqa: E402

from oracle import golden_inputs as GI  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def _import_reference():
    import huggingface_hub
    import transformers.utils as tu
    if not hasattr(huggingface_hub, "cached_download"):
        huggingface_hub.cached_download = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("offline"))
    if not hasattr(tu, "FLAX_WEIGHTS_NAME"):
        tu.FLAX_WEIGHTS_NAME = "flax_model.msgpack"
    from diffusers import EulerDiscreteScheduler
    from solver_utils.consistency import consistency_check_with_depth
    from solver_utils.forward_warp import forward_warp, inverse_warp
    return EulerDiscreteScheduler, consistency_check_with_depth, forward_warp, inverse_warp


@contextlib.contextmanager
def cuda_to_cpu():
    """inverse_warp hard-codes .to('cuda') (forward_warp.py:225-256)."""
    orig = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
        if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
            k["device"] = "cpu"
        return orig(self, *a, **k)

    orig_tensor = torch.tensor

    def tensor(*a, **k):                  # torch.tensor(x, device='cuda') at model/diffusionGS.py:1339-1344
        if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
            k["device"] = "cpu"
        return orig_tensor(*a, **k)

    torch.Tensor.to = to
    torch.tensor = tensor
    try:
        yield
    finally:
        torch.Tensor.to = orig
        torch.tensor = orig_tensor


def gen_warp(consistency, forward_warp, inverse_warp):
    for name in GI.WARP_CASES:
        c = GI.warp_case(name)
        sy, sx = c["stride"]
        out = {}
        with cuda_to_cpu():
            d = inverse_warp(torch.from_numpy(c["img"]), torch.from_numpy(c["depth"])[None],
                             torch.from_numpy(c["depth_pseudo"])[None], torch.from_numpy(c["pose1"]),
                             torch.from_numpy(c["pose2"]), torch.from_numpy(c["K"]), bg_mask=None,
                             bandwidth=c["bandwidth"])
        for k, v in d.items():
            if v is None:
                continue
            out["iw_" + k] = v.numpy()[..., ::sy, ::sx]
        err = consistency(torch.from_numpy(c["depth_pseudo"]), torch.from_numpy(c["pose2"]), torch.from_numpy(c["K"]),
                          torch.from_numpy(c["depth"]), torch.from_numpy(c["pose1"]), torch.from_numpy(c["K"]))
        out["reproj_error"] = err.numpy()[::sy, ::sx]
        frame = (c["img"].transpose(1, 2, 0) * 255.0).astype(np.float64)
        warped, mask2, flow = forward_warp(frame, None, c["depth"].astype(np.float64), c["pose1"].astype(np.float64),
                                           c["pose2"].astype(np.float64), c["K"].astype(np.float64), None)
        out["fw_warped"] = warped[::sy, ::sx]
        out["fw_mask"] = mask2[::sy, ::sx]
        out["fw_flow"] = flow[::sy, ::sx]
        np.savez_compressed(GOLD / f"warp_{name}.npz", **out)
        print("warp", name, {k: v.shape for k, v in out.items()})


def gen_sched(EulerDiscreteScheduler):
    sch = EulerDiscreteScheduler(**GI.SCHED_CONFIG)
    sch.set_timesteps(100)
    np.savez_compressed(GOLD / "sched_sigmas.npz", sigmas=sch.sigmas.numpy(), timesteps=sch.timesteps.numpy(),
                        init_noise_sigma=np.float32(sch.init_noise_sigma))
    sch25 = EulerDiscreteScheduler(**GI.SCHED_CONFIG)
    sch25.set_timesteps(25)
    np.savez_compressed(GOLD / "sched_sigmas25.npz", sigmas=sch25.sigmas.numpy(), timesteps=sch25.timesteps.numpy())
    for name in GI.SCHED_CASES:
        c = GI.sched_case(name)
        s = c["stride"]
        t = sch.timesteps[c["step_i"]]
        out = {}
        v = torch.from_numpy(c["model_output"])
        x = torch.from_numpy(c["sample"])
        cond = torch.from_numpy(c["temp_cond"])
        mask = torch.from_numpy(c["mask"])
        lam = torch.from_numpy(c["lambda_ts"])
        # guidance-gradient variant, exactly as the pipeline drives it (…post.py:727-774): sample is a leaf
        for cg in (True, False):
            sch.is_scale_input_called = True
            xs = x.clone().requires_grad_(cg)
            with torch.enable_grad() if cg else torch.no_grad():
                r = sch.step_interp(v, t, xs, cond, mask, lam, step_i=c["step_i"], lr=0.02, compute_grad=cg)
            tag = "g1" if cg else "g0"
            out[f"interp_{tag}_prev"] = r.prev_sample.detach().numpy()[..., ::s, ::s]
            out[f"interp_{tag}_x0"] = r.pred_original_sample.detach().numpy()[..., ::s, ::s]
            if cg:
                out["interp_g1_grad"] = r.grad.detach().numpy()[..., ::s, ::s]
                out["interp_g1_grad_std"] = np.float64(r.grad.detach().double().std())
        with torch.no_grad():
            r = sch.step_interp_prob_uncertain(v, t, x.clone(), cond, mask, lam, step_i=c["step_i"])
        out["replace_prev"] = r.prev_sample.numpy()[..., ::s, ::s]
        out["replace_x0"] = r.pred_original_sample.numpy()[..., ::s, ::s]
        np.savez_compressed(GOLD / f"sched_{name}.npz", **out)
        print("sched", name, {k: getattr(v_, "shape", ()) for k, v_ in out.items()})


def gen_unet():
    """Reference UNetSpatioTemporalConditionModel (small config, CPU fp32) on seeded weights/inputs."""
    from diffusers.models import UNetSpatioTemporalConditionModel
    from oracle import unet_weights as UW
    torch.manual_seed(0)
    model = UNetSpatioTemporalConditionModel(**UW.SMALL_CONFIG)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(UW.make_state_dict(shapes))
    model.eval()
    out = {"names": np.array(sorted(shapes)), "shapes": np.array([str(shapes[k]) for k in sorted(shapes)])}
    for tag, (B, F, h, w) in {"b2f5": (2, 5, 16, 24), "b1f14": (1, 14, 8, 16)}.items():
        sample, t, ehs, added = UW.make_inputs(B, F, h, w, seed=F)
        with torch.no_grad():
            y = model(sample, t, ehs, added, return_dict=False)[0]
        out[f"{tag}_out"] = y.numpy()
        print("unet", tag, y.shape, float(y.abs().mean()), float(y.std()))
    np.savez_compressed(GOLD / "unet_small.npz", **out)


UNET_FULL_CASES = {           # tag: (B, F, h, w, spatial stride of the stored sample)
    "b2f14_72x128": (2, 14, 72, 128, 3),     # bench.py's unit (BASELINE config 2)
    "b1f25_40x72": (1, 25, 40, 72, 2),       # Post variant guidance tiles (...post.py:739-779)
    "b1f25_48x72": (1, 25, 48, 72, 2),
    "b2f25_72x128": (2, 25, 72, 128, 3),     # the reference's own CFG forward (F = 25)
}


def gen_unet_full(which=None):
    """The REFERENCE UNetSpatioTemporalConditionModel() in its DEFAULT configuration (= SVD-XT: 320/640/1280/1280,
    5/10/20/20 heads, 1.52 B parameters), CPU fp32, name-keyed seeded weights, at the shapes bench.py and the
    pipelines launch.  Stores a strided sample of the output per case (tests rebuild weights and inputs from the
    seeds).  ~4-15 min of CPU per case; one file per case so they can be regenerated one at a time."""
    import time
    from diffusers.models import UNetSpatioTemporalConditionModel
    from oracle import unet_weights as UW
    torch.manual_seed(0)
    t0 = time.time()
    model = UNetSpatioTemporalConditionModel()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(UW.make_state_dict(shapes, seed=5))
    model.eval()
    print("unet_full: model ready", sum(math_prod(s) for s in shapes.values()) / 1e9, "B params", f"{time.time() - t0:.0f} s", flush=True)
    for tag, (B, F, h, w, st) in UNET_FULL_CASES.items():
        if which and tag not in which:
            continue
        sample, t, ehs, added = UW.make_inputs(B, F, h, w, seed=F, cross=1024)
        t0 = time.time()
        with torch.no_grad():
            y = model(sample, t, ehs, added, return_dict=False)[0]
        y = y.numpy()
        np.savez_compressed(GOLD / f"unet_full_{tag}.npz", out=y[..., ::st, ::st].astype(np.float32), stride=np.int64(st),
                            mean_abs=np.float64(np.abs(y).mean()), std=np.float64(y.std()),
                            checksum=np.float64(y.astype(np.float64).sum()))
        print("unet_full", tag, y.shape, float(np.abs(y).mean()), float(y.std()), f"{time.time() - t0:.0f} s", flush=True)


def math_prod(s):
    r = 1
    for v in s:
        r *= int(v)
    return r


def gen_vae():
    """Reference AutoencoderKLTemporalDecoder (reduced config, CPU fp32): encode moments and decoded frames."""
    from diffusers.models import AutoencoderKLTemporalDecoder
    from oracle import unet_weights as UW
    from oracle import vae_weights as VW
    torch.manual_seed(0)
    model = AutoencoderKLTemporalDecoder(**VW.SMALL_VAE_CONFIG)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(UW.make_state_dict(shapes, seed=7))
    model.eval()
    out = {"names": np.array(sorted(shapes)), "shapes": np.array([str(shapes[k]) for k in sorted(shapes)])}
    with torch.no_grad():
        x = VW.make_images()
        out["moments"] = model.encode(x).latent_dist.parameters.numpy()
        z = VW.make_latents()
        out["decoded_f3"] = model.decode(z, num_frames=3).sample.numpy()
        out["decoded_f1"] = model.decode(z[:2], num_frames=1).sample.numpy()
    for k in ("moments", "decoded_f3", "decoded_f1"):
        print("vae", k, out[k].shape, float(np.abs(out[k]).mean()), float(out[k].std()))
    np.savez_compressed(GOLD / "vae_small.npz", **out)


def gen_vae_full():
    """The REFERENCE AutoencoderKLTemporalDecoder in the SVD-XT configuration (97.7 M parameters), CPU fp32, at the
    pipelines' size: one encode at 576x1024 (autoencoder_kl_temporal_decoder.py:317-343) and a 2-frame decode from 72x128
    latents (:345-370); the decoded frames are stored every 8th pixel.  A few minutes of CPU."""
    import time
    from diffusers.models import AutoencoderKLTemporalDecoder
    from oracle import unet_weights as UW
    from oracle import vae_weights as VW
    model = AutoencoderKLTemporalDecoder(**VW.FULL_VAE_CONFIG)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(UW.make_state_dict(shapes, seed=13))
    model.eval()
    out = {}
    st = VW.FULL_VAE_STRIDE
    with torch.no_grad():
        t0 = time.time()
        out["moments"] = model.encode(VW.make_full_image()).latent_dist.parameters.numpy()
        print(f"vae_full encode {time.time() - t0:.0f} s", flush=True)
        t0 = time.time()
        out["decoded_f2"] = model.decode(VW.make_full_latents(), num_frames=2).sample.numpy()[..., ::st, ::st]
        print(f"vae_full decode {time.time() - t0:.0f} s", flush=True)
    for k in out:
        print("vae_full", k, out[k].shape, float(np.abs(out[k]).mean()), float(out[k].std()))
    np.savez_compressed(GOLD / "vae_full.npz", **out)


def _reference_pipe(cls, unet, vae=None):
    """An instance of the reference pipeline class `cls` with the given modules as plain attributes (DiffusionPipeline's
    register_modules / device bookkeeping skipped): mock CLIP, mock VAE unless one is passed, the reference scheduler."""
    from diffusers import EulerDiscreteScheduler
    from diffusers.image_processor import VaeImageProcessor
    from oracle import pipeline_mocks as PM

    class Pipe(cls):
        def __init__(self):
            self.vae, self.image_encoder, self.unet = (vae if vae is not None else PM.MockVAE()), PM.MockImageEncoder(), unet
            self.scheduler = EulerDiscreteScheduler(**GI.SCHED_CONFIG)
            self.feature_extractor = None
            self.vae_scale_factor = 8
            self.image_processor = VaeImageProcessor(vae_scale_factor=8)

        @property
        def _execution_device(self):
            return torch.device("cpu")

        def check_inputs(self, *a, **k):
            return None

        def maybe_free_model_hooks(self):
            return None
    return Pipe()


def gen_pipeline():
    """Run the REFERENCE pipeline classes' own __call__ (both variants) on the CPU with mock CLIP / VAE /
    UNet (oracle/pipeline_mocks.py), 3 denoise steps, output_type='latent'."""
    import diffusers.utils.torch_utils as tu_
    from diffusers import EulerDiscreteScheduler
    from diffusers.image_processor import VaeImageProcessor
    from oracle import pipeline_mocks as PM
    import model.SVD_2pass_prob_uncertain as P2
    import model.SVD_2pass_prob_uncertain_post as P1

    inp = PM.pipeline_inputs(seed=0)

    make = lambda cls: _reference_pipe(cls, PM.MockUNet())

    out = {}
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self           # `mask = mask.cuda()` (…post.py:557)
    for tag, mod in (("post", P1), ("replace", P2)):
        # both pipelines draw ONE augmentation noise with randn_tensor (…post.py:583): inject the fixture's
        orig = mod.randn_tensor
        mod.randn_tensor = lambda shape, **k: inp["noise"].clone() if tuple(shape) == tuple(inp["noise"].shape) else orig(shape, **k)
        try:
            pipe = make(mod.StableVideoDiffusionPipeline)
            with torch.no_grad():
                res = pipe(inp["image"], temp_cond=inp["temp_cond"], mask=inp["mask"].clone(), lambda_ts=inp["lambda_ts"],
                           num_frames=25, decode_chunk_size=8, num_inference_steps=3, latent_num=1,
                           latents=inp["latents"].clone(), output_type="latent")
        finally:
            mod.randn_tensor = orig
        lat = res.frames
        out[tag] = lat.float().numpy()[..., ::3, ::3]
        print("pipeline", tag, lat.shape, lat.dtype, float(lat.abs().mean()))
    torch.Tensor.cuda = orig_cuda
    np.savez_compressed(GOLD / "pipeline_mock.npz", **out)


def gen_pipeline_one_pass():
    """BASELINE config 2's "SVD_1pass": the forward-in-time pass only.  The reference has no live 1-pass pipeline
    (model/SVD_1pass.py is dead code), so the fixture is the REFERENCE two-pass classes' own
    __call__ with the forward/backward blend weight `torch.linspace(1,0,num_frames)` (...post.py:667, :828) forced
    to ones: latents = 1 * forward + 0 * backward at every step.  Mock CLIP / VAE / UNet as gen_pipeline."""
    from diffusers import EulerDiscreteScheduler
    from diffusers.image_processor import VaeImageProcessor
    from oracle import pipeline_mocks as PM
    import model.SVD_2pass_prob_uncertain as P2
    import model.SVD_2pass_prob_uncertain_post as P1

    inp = PM.pipeline_inputs(seed=0)

    make = lambda cls: _reference_pipe(cls, PM.MockUNet())

    out = {}
    orig_cuda, orig_linspace = torch.Tensor.cuda, torch.linspace

    def linspace(start, end, steps, *a, **k):
        if (start, end) == (1, 0):
            return torch.ones(steps)
        return orig_linspace(start, end, steps, *a, **k)

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.linspace = linspace
    try:
        for tag, mod in (("post", P1), ("replace", P2)):
            orig = mod.randn_tensor
            mod.randn_tensor = lambda shape, **k: inp["noise"].clone() if tuple(shape) == tuple(inp["noise"].shape) else orig(shape, **k)
            try:
                with torch.no_grad():
                    res = make(mod.StableVideoDiffusionPipeline)(
                        inp["image"], temp_cond=inp["temp_cond"], mask=inp["mask"].clone(), lambda_ts=inp["lambda_ts"],
                        num_frames=25, decode_chunk_size=8, num_inference_steps=3, latent_num=1,
                        latents=inp["latents"].clone(), output_type="latent")
            finally:
                mod.randn_tensor = orig
            out[tag] = res.frames.float().numpy()[..., ::3, ::3]
            print("pipeline one-pass", tag, res.frames.shape, float(res.frames.abs().mean()))
    finally:
        torch.Tensor.cuda, torch.linspace = orig_cuda, orig_linspace
    np.savez_compressed(GOLD / "pipeline_one_pass.npz", **out)


def gen_pipeline_real_unet():
    """The REFERENCE pipeline classes' own __call__ (both variants, CPU fp32) with the REFERENCE
    UNetSpatioTemporalConditionModel in a reduced configuration on seeded weights (mock CLIP / VAE as in
    gen_pipeline): pins the loops TOGETHER with the real UNet's numerics — CFG batch, the four guidance tiles of
    the Post variant, time flips, blend — end to end.  2 denoise steps, output_type='latent'.  ~10 min of CPU."""
    from diffusers import EulerDiscreteScheduler
    from diffusers.image_processor import VaeImageProcessor
    from diffusers.models import UNetSpatioTemporalConditionModel
    from oracle import pipeline_mocks as PM
    from oracle import unet_weights as UW
    import model.SVD_2pass_prob_uncertain as P2
    import model.SVD_2pass_prob_uncertain_post as P1

    cfg = UW.PIPELINE_CONFIG
    torch.manual_seed(0)
    unet = UNetSpatioTemporalConditionModel(**cfg)
    shapes = {k: tuple(v.shape) for k, v in unet.state_dict().items()}
    unet.load_state_dict(UW.make_state_dict(shapes, seed=3))
    unet.eval()
    inp = PM.pipeline_inputs(seed=1)

    make = lambda cls: _reference_pipe(cls, unet)

    out = {}
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    import time
    for tag, mod in (("replace", P2), ("post", P1)):
        orig = mod.randn_tensor
        mod.randn_tensor = lambda shape, **k: inp["noise"].clone() if tuple(shape) == tuple(inp["noise"].shape) else orig(shape, **k)
        t0 = time.time()
        try:
            pipe = make(mod.StableVideoDiffusionPipeline)
            with torch.no_grad():
                res = pipe(inp["image"], temp_cond=inp["temp_cond"], mask=inp["mask"].clone(), lambda_ts=inp["lambda_ts"],
                           num_frames=25, decode_chunk_size=8, num_inference_steps=2, latent_num=1,
                           latents=inp["latents"].clone(), output_type="latent")
        finally:
            mod.randn_tensor = orig
        lat = res.frames
        out[tag] = lat.float().numpy()[..., ::3, ::3]
        print("pipeline+unet", tag, lat.shape, lat.dtype, float(lat.abs().mean()), f"{time.time() - t0:.0f} s", flush=True)
    torch.Tensor.cuda = orig_cuda
    np.savez_compressed(GOLD / "pipeline_unet.npz", **out)


def gen_pipeline_full():
    """FULL SIZE: the reference `SVD_2pass_prob_uncertain.StableVideoDiffusionPipeline.__call__` ("Replace") driving the
    reference UNetSpatioTemporalConditionModel() in its default = SVD-XT configuration (1.52 B parameters, CPU fp32, the
    name-keyed seeded weights of gen_unet_full), mock CLIP / VAE, ONE denoising step (both passes: two CFG forwards at
    [2,25,8,72,128]), output_type='latent'.  ~30 min of CPU.  The Post variant is not generated at this size: the reference
    builds an autograd graph through each guidance-tile UNet forward (…post.py:727-774), which at full width needs more
    host memory than this container has (62 GB, no swap); its loop is pinned at the reduced width (gen_pipeline_real_unet)
    and its tile forwards at full width (gen_unet_full b1f25_40x72 / b1f25_48x72)."""
    import time
    from diffusers import EulerDiscreteScheduler
    from diffusers.image_processor import VaeImageProcessor
    from diffusers.models import UNetSpatioTemporalConditionModel
    from oracle import pipeline_mocks as PM
    from oracle import unet_weights as UW
    import model.SVD_2pass_prob_uncertain as P2
    unet = UNetSpatioTemporalConditionModel()
    shapes = {k: tuple(v.shape) for k, v in unet.state_dict().items()}
    unet.load_state_dict(UW.make_state_dict(shapes, seed=5))
    unet.eval()
    inp = PM.pipeline_inputs(seed=6)

    Pipe = lambda: _reference_pipe(P2.StableVideoDiffusionPipeline, unet, None)

    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    orig = P2.randn_tensor
    P2.randn_tensor = lambda shape, **k: inp["noise"].clone() if tuple(shape) == tuple(inp["noise"].shape) else orig(shape, **k)
    t0 = time.time()
    try:
        with torch.no_grad():
            res = Pipe()(inp["image"], temp_cond=inp["temp_cond"], mask=inp["mask"].clone(), lambda_ts=inp["lambda_ts"],
                         num_frames=25, decode_chunk_size=8, num_inference_steps=1, latent_num=1,
                         latents=inp["latents"].clone(), output_type="latent")
    finally:
        P2.randn_tensor = orig
        torch.Tensor.cuda = orig_cuda
    lat = res.frames.float().numpy()
    print("pipeline_full replace", lat.shape, float(np.abs(lat).mean()), f"{time.time() - t0:.0f} s", flush=True)
    np.savez_compressed(GOLD / "pipeline_unet_full.npz", replace=lat[..., ::2, ::2], mean_abs=np.float64(np.abs(lat).mean()),
                        std=np.float64(lat.std()))


class _NoGradUNet(torch.nn.Module):
    """The reference UNet run under torch.no_grad(), everything else delegated.  The Post loop calls its four guidance-tile
    forwards under torch.enable_grad() (…post.py:727-774), but the gradient it consumes is `sample.grad` of the SCHEDULER's
    loss (scheduling_euler_discrete.py:782-789) with respect to `latents1`, and the UNet's input is a detached leaf
    (…post.py:732-733): no path from the loss to `sample` runs through the UNet, so cutting the UNet's graph leaves every
    value the pipeline computes unchanged while the saved activations (tens of GB at full width) are never kept.
    `nograd_check` below proves it on the reduced-width fixture that WAS generated with the graph."""

    def __init__(self, unet):
        super().__init__()
        self.inner = unet

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("inner"), name)

    def forward(self, *a, **k):
        with torch.no_grad():
            return self.inner(*a, **k)


def _run_post_reference(unet, inp, steps):
    """The reference Post pipeline class' own __call__ on `unet` (mock CLIP / VAE), `steps` denoising steps, latents out."""
    import model.SVD_2pass_prob_uncertain_post as P1
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    orig = P1.randn_tensor
    P1.randn_tensor = lambda shape, **k: inp["noise"].clone() if tuple(shape) == tuple(inp["noise"].shape) else orig(shape, **k)
    try:
        with torch.no_grad():
            res = _reference_pipe(P1.StableVideoDiffusionPipeline, unet)(
                inp["image"], temp_cond=inp["temp_cond"], mask=inp["mask"].clone(), lambda_ts=inp["lambda_ts"], num_frames=25,
                decode_chunk_size=8, num_inference_steps=steps, latent_num=1, latents=inp["latents"].clone(), output_type="latent")
    finally:
        P1.randn_tensor = orig
        torch.Tensor.cuda = orig_cuda
    return res.frames.float().numpy()


def gen_nograd_check():
    """Equivalence proof for _NoGradUNet: the reduced-width Post run of gen_pipeline_real_unet (whose committed fixture
    `pipeline_unet.npz['post']` was produced WITH the reference's autograd graph through the tile forwards) repeated with
    the no-grad wrapper must reproduce that fixture to fp32 rounding (measured: max 2.4e-4 on a scale of 3.35 = 7e-5 relative
    after two steps - torch picks other CPU kernels / reduction splits without a graph and under another thread count; a
    gradient that DID depend on the UNet graph would differ at the scale of the guidance term itself, ~1e-1)."""
    from diffusers.models import UNetSpatioTemporalConditionModel
    from oracle import pipeline_mocks as PM
    from oracle import unet_weights as UW
    torch.manual_seed(0)
    unet = UNetSpatioTemporalConditionModel(**UW.PIPELINE_CONFIG)
    shapes = {k: tuple(v.shape) for k, v in unet.state_dict().items()}
    unet.load_state_dict(UW.make_state_dict(shapes, seed=3))
    unet.eval()
    lat = _run_post_reference(_NoGradUNet(unet), PM.pipeline_inputs(seed=1), 2)[..., ::3, ::3]
    ref = np.load(GOLD / "pipeline_unet.npz")["post"]
    d = float(np.abs(lat - ref).max())
    print("nograd_check: max |no-grad wrapper - with-graph fixture| =", d, "of scale", float(np.abs(ref).max()), flush=True)
    assert d <= 2e-4 * float(np.abs(ref).max()), "the no-grad wrapper changed the Post pipeline's result"


def gen_pipeline_full_post():
    """As gen_pipeline_full for the Post variant (`SVD_2pass_prob_uncertain_post`), the one every LLFF / DL3DV script runs:
    the reference pipeline class and scheduler untouched, the reference UNet in its default = SVD-XT configuration
    (1.52 B parameters, CPU fp32) behind _NoGradUNet (round 3 found that WITH the reference's autograd graph through the
    guidance-tile forwards the first tile needs more than this container's 62 GB), ONE denoising step = both passes, each
    four tile forwards [1,25,8,40|48,72] + one CFG forward [2,25,8,72,128].  ~25 min of CPU, peak ~35 GB.
    Writes tests/golden/pipeline_unet_full_post.npz."""
    import time
    from diffusers.models import UNetSpatioTemporalConditionModel
    from oracle import pipeline_mocks as PM
    from oracle import unet_weights as UW
    unet = UNetSpatioTemporalConditionModel()
    shapes = {k: tuple(v.shape) for k, v in unet.state_dict().items()}
    unet.load_state_dict(UW.make_state_dict(shapes, seed=5))
    unet.eval()
    inp = PM.pipeline_inputs(seed=6)
    t0 = time.time()
    lat = _run_post_reference(_NoGradUNet(unet), inp, 1)
    print("pipeline_full post", lat.shape, float(np.abs(lat).mean()), f"{time.time() - t0:.0f} s", flush=True)
    np.savez_compressed(GOLD / "pipeline_unet_full_post.npz", post=lat[..., ::2, ::2], mean_abs=np.float64(np.abs(lat).mean()),
                        std=np.float64(lat.std()))


def gen_pipeline_real_unet_vae():
    """As gen_pipeline_real_unet, 'replace' variant, with the REFERENCE AutoencoderKLTemporalDecoder too (reduced
    four-level configuration) and output_type='np': the pipelines' VAE plumbing (scaled / noised condition encodes,
    chunked temporal decode, fp32 upcast) end to end.  Frames stored at every 16th pixel."""
    from diffusers import EulerDiscreteScheduler
    from diffusers.image_processor import VaeImageProcessor
    from diffusers.models import AutoencoderKLTemporalDecoder, UNetSpatioTemporalConditionModel
    from oracle import pipeline_mocks as PM
    from oracle import unet_weights as UW
    from oracle import vae_weights as VW
    import model.SVD_2pass_prob_uncertain as P2
    import time

    torch.manual_seed(0)
    unet = UNetSpatioTemporalConditionModel(**UW.PIPELINE_CONFIG)
    unet.load_state_dict(UW.make_state_dict({k: tuple(v.shape) for k, v in unet.state_dict().items()}, seed=3))
    vae = AutoencoderKLTemporalDecoder(**VW.PIPELINE_VAE_CONFIG)
    vae.load_state_dict(UW.make_state_dict({k: tuple(v.shape) for k, v in vae.state_dict().items()}, seed=11))
    unet.eval(); vae.eval()
    inp = PM.pipeline_inputs(seed=2)

    Pipe = lambda: _reference_pipe(P2.StableVideoDiffusionPipeline, unet, vae)

    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    orig = P2.randn_tensor
    P2.randn_tensor = lambda shape, **k: inp["noise"].clone() if tuple(shape) == tuple(inp["noise"].shape) else orig(shape, **k)
    t0 = time.time()
    try:
        with torch.no_grad():
            res = Pipe()(inp["image"], temp_cond=inp["temp_cond"], mask=inp["mask"].clone(), lambda_ts=inp["lambda_ts"],
                         num_frames=25, decode_chunk_size=8, num_inference_steps=2, latent_num=1,
                         latents=inp["latents"].clone(), output_type="np")
    finally:
        P2.randn_tensor = orig
        torch.Tensor.cuda = orig_cuda
    frames = np.asarray(res.frames[0], dtype=np.float32)
    print("pipeline+unet+vae replace", frames.shape, float(frames.mean()), float(frames.std()), f"{time.time() - t0:.0f} s", flush=True)
    np.savez_compressed(GOLD / "pipeline_unet_vae.npz", frames=frames[:, ::16, ::16])


def gen_clip_preprocess():
    """The reference pipelines' CLIP preprocessing (`_encode_image`, …post.py:229-258): [-1,1], `_resize_with_antialiasing` to
    224 x 224 (the reference's own function, imported), back to [0,1], the feature extractor's mean / std normalisation written
    out as the arithmetic it is (transformers' CLIPImageProcessor with resize / crop / rescale switched off).  Two seeded
    images (golden_inputs.clip_image): 576 x 1024 (the pipelines' size) and 378 x 504 (an LLFF-like aspect); every second
    pixel of the 224 x 224 result is stored."""
    import model.SVD_2pass_prob_uncertain_post as P1
    out = {}
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1)
    for tag, (h, w) in GI.CLIP_CASES.items():
        img8 = torch.from_numpy(GI.clip_image(h, w)).float().div(255).permute(2, 0, 1)[None]      # pil_to_numpy / numpy_to_pt
        x = (P1._resize_with_antialiasing(img8 * 2.0 - 1.0, (224, 224)) + 1.0) / 2.0
        px = ((x - mean) / std).numpy()
        out[f"px_{tag}"] = px[..., ::2, ::2]
        out[f"mean_abs_{tag}"] = np.float64(np.abs(px).mean())
        print("clip", tag, px.shape, float(np.abs(px).mean()))
    np.savez_compressed(GOLD / "clip_preprocess.npz", **out)


def gen_orchestrator():
    """Pure-numpy methods of the reference's DiffusionGS (model/diffusionGS.py:1120-1296).  The module
    imports packages that are absent here (FSGS submodule, cv2, open3d, trimesh); they are not touched by
    these methods, so empty placeholder modules satisfy the import statements."""
    DiffusionGS = _reference_diffusiongs()
    out = {}
    for k, (a, b) in enumerate(GI.orch_pose_pairs()):
        poses = DiffusionGS.pose_interpolation(None, a, b)
        out[f"poses{k}"] = poses
        d, idx = DiffusionGS.compute_dists(None, poses)
        out[f"dists{k}"] = d
        out[f"idx{k}"] = np.int64(idx)
    for k, m in enumerate(GI.orch_masks()):
        out[f"lambda{k}"] = DiffusionGS.search_hypers_v2(None, torch.from_numpy(m), "/tmp").numpy()
    # perturbation candidates draw from the unseeded global np.random: seed it for the fixture
    a, b = GI.orch_pose_pairs()[0]
    anchors = DiffusionGS.pose_interpolation(None, a, b)[::4]
    np.random.seed(1234)
    cands = DiffusionGS._perturb_interp_pose_candidates(None, anchors, perturb_num=5)
    out["perturbed"] = np.array(cands)
    np.savez_compressed(GOLD / "orchestrator.npz", **out)
    print("orchestrator", {k: getattr(v, "shape", ()) for k, v in out.items()})


def gen_orch_nearby():
    """O4: the reference's own `consistency_check_from_nearby_images_bw` (model/diffusionGS.py:1300-1361) on five seeded
    576 x 1024 frames.  The method touches `self.diffusion_width / diffusion_height` and `inverse_warp` only (no cv2, no
    renderer, no FSGS), so a two-attribute stand-in for `self` and the cuda -> cpu redirect run it as it is."""
    import types
    DiffusionGS = _reference_diffusiongs()
    K, poses, images, depths = GI.orch_nearby_case()
    me = types.SimpleNamespace(diffusion_width=1024, diffusion_height=576)
    with cuda_to_cpu():
        um, im = DiffusionGS.consistency_check_from_nearby_images_bw(me, K, poses, images=images, depths=depths)
    sy, sx = GI.ORCH_NEARBY_STRIDE
    out = {"uncertainty": np.stack([u.numpy()[::sy, ::sx] for u in um]).astype(np.float32),
           "intensity_uncertainty": np.stack([u.numpy()[::sy, ::sx] for u in im]).astype(np.float32),
           "uncertainty_mean": np.array([float(u.double().mean()) for u in um]),
           "intensity_uncertainty_mean": np.array([float(u.double().mean()) for u in im])}
    np.savez_compressed(GOLD / "orch_nearby.npz", **out)
    print("orch_nearby", {k: getattr(v, "shape", ()) for k, v in out.items()}, out["uncertainty_mean"],
          out["intensity_uncertainty_mean"])


def gen_orch_fusion():
    """O5: the uncertainty fusion and condition-image selection of `_interpolate_between_gs_v3` (model/diffusionGS.py:821-867).  The
    formula is inline in a 150-line method that needs a renderer, so it is not callable as a function; its STATEMENTS are: the block
    from `def get_intensity_confidence(` to the line before `# save for debugging` is read from the reference file at generation
    time and executed as it stands (dedented) on seeded inputs bound to the names it uses (`self.diffusion_type`, `aux`,
    `pseudo_images`, the two nearby-consistency lists).  Nothing of the block is stored: the fixture holds its outputs."""
    import textwrap
    import types
    lines = (REF / "model" / "diffusionGS.py").read_text().splitlines()
    a = next(i for i, l in enumerate(lines) if i > 800 and l.strip().startswith("def get_intensity_confidence("))     # (a dead path holds a copy at :527)
    b = next(i for i, l in enumerate(lines) if i > a and l.strip() == "# save for debugging")
    assert 815 <= a + 1 <= 830 and 860 <= b + 1 <= 875, (a, b)          # the snapshot this build surveyed
    block = textwrap.dedent("\n".join(lines[a:b]))
    c = GI.orch_fusion_case()
    ns = {"np": np, "torch": torch, "self": types.SimpleNamespace(diffusion_type="2PassProbUncertainPost"),
          "aux": {"soft_masks_reproj": None, "soft_masks_reproj_ori": np.stack(c["soft_masks_reproj_ori"]), "cond_images_ori": c["cond_images_ori"]},
          "pseudo_images": c["pseudo_images"],
          "nearby_consistency_uncertainty": [torch.from_numpy(x) for x in c["nearby"]],
          "nearby_inten_consistency_uncertainty": [torch.from_numpy(x) for x in c["nearby_inten"]]}
    exec(compile(block, "diffusionGS.py:%d-%d" % (a + 1, b), "exec"), ns)
    masks, cond = ns["masks"].numpy(), np.asarray(ns["cond_image"])
    sy, sx = GI.ORCH_NEARBY_STRIDE
    gs = np.stack(c["pseudo_images"][1:-1])
    out = {"masks": masks.astype(np.float32), "cond_image": cond[:, ::sy, ::sx].astype(np.float32),
           "took_gs": np.all(cond == gs, axis=-1)[:, ::sy, ::sx], "cond_image_mean": cond.astype(np.float64).mean(axis=(1, 2, 3)),
           "geo_inten_uncertainty_debug": np.asarray(ns["geo_inten_uncertainty"])[:, ::sy, ::sx, 0].astype(np.float32)}
    np.savez_compressed(GOLD / "orch_fusion.npz", **out)
    print("orch_fusion", {k: getattr(v, "shape", ()) for k, v in out.items()}, masks.mean(axis=(1, 2)), out["took_gs"].mean(axis=(1, 2)))


def gen_train_flags():
    """The (flag, default, choices, type, action, nargs) of every `parser.add_argument` the reference's CLI declares itself
    (scripts/train.py:28-69; the FSGS parameter groups it also instantiates live in the absent submodule).  The script
    cannot be imported here (FSGS), so the calls are read off its syntax tree; only this table - data - is committed
    (tests/golden/train_flags.json), and tests/test_dist_cpu.py holds syn3r_amd/launch.py to it."""
    import ast
    import json
    tree = ast.parse((REF / "scripts" / "train.py").read_text())
    rows = []
    for node in ast.walk(tree):
        if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument"):
            continue
        if not (isinstance(node.func.value, ast.Name) and node.func.value.id == "parser"):
            continue
        flags = [ast.literal_eval(a) for a in node.args]
        kw = {}
        for k in node.keywords:
            if k.arg == "type":
                kw["type"] = k.value.id
            elif k.arg in ("default", "choices", "action", "nargs"):
                kw[k.arg] = ast.literal_eval(k.value)
        rows.append(dict(flags=flags, line=node.lineno, **kw))
    rows.sort(key=lambda r: r["line"])
    (GOLD / "train_flags.json").write_text(json.dumps(rows, indent=1) + "\n")
    print("train_flags", len(rows), [r["flags"][0] for r in rows])


def _reference_diffusiongs():
    """The reference's DiffusionGS class; the absent packages its module imports get empty placeholder modules (the
    methods called here never touch them, except the name `trimesh.Scene` that densify_views instantiates and drops)."""
    import types
    for name in ("cv2", "trimesh", "open3d", "FSGS", "FSGS.utils", "FSGS.utils.trainer", "FSGS.scene", "FSGS.scene.cameras"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["FSGS.utils.trainer"].init_GSTrainer = None
    sys.modules["FSGS.utils.trainer"].GSTrainer = object
    sys.modules["FSGS.scene.cameras"].Camera = object
    sys.modules["trimesh"].Scene = object
    from model.diffusionGS import DiffusionGS
    return DiffusionGS


class _Captured(Exception):
    pass


def gen_n2():
    """Key-frame bookkeeping of the point-cloud densification (model/diffusionGS.py:179-308) and the frame filter /
    pair graph / intrinsics of `densify_pcds` (:347-435), captured from the REFERENCE's own methods run on a stand-in
    `self`: `_interpolate_between_gs_v3`, `render_GS`, `gsTrainer.generate_corresp_mask` and `dust3r` are recorders fed
    with seeded data (the networks are absent); everything between them is the reference's code."""
    import tempfile
    import types
    DiffusionGS = _reference_diffusiongs()
    out = {}
    for name, (V, dtype, fps, nkey) in GI.N2_CASES.items():
        vposes = GI.n2_view_poses(V)
        cap = {}

        def interp(i, j, replace=True, perturb_interp_poses=False):
            poses = list(DiffusionGS.pose_interpolation(None, vposes[i], vposes[j]))
            frames = [torch.full((3, 4, 6), GI.n2_frame_id(i, k), dtype=torch.float32) for k in range(25)]
            return frames, poses, None

        def densify_pcds(frames, poses, key_frame_mask=None, input_flags=None, win_samples=-1):
            cap.update(frames=np.array([float(f[0, 0, 0]) for f in frames], np.float32), poses=np.array(poses),
                       key_frame_mask=np.array(key_frame_mask), input_flags=np.array(input_flags), win_samples=win_samples)
            raise _Captured

        with tempfile.TemporaryDirectory() as tmp:
            me = types.SimpleNamespace(num_input_views=V, save_dir=tmp, fps_keyframe_sampling=fps,
                                       _interpolate_between_gs_v3=interp, densify_pcds=densify_pcds)
            try:
                DiffusionGS.densify_views(me, 0, down_sample_rate=1, densify_type=dtype, num_views_for_pcd_densification=nkey)
            except _Captured:
                pass
        for k in ("frames", "poses", "key_frame_mask", "input_flags"):
            out[f"{name}_sel_{k}"] = cap[k]

        # ---- densify_pcds itself on what densify_views handed over
        n = len(cap["frames"])
        means = GI.n2_mask_means(n)
        rec = {}
        calls = []

        def render_GS(idx=None, pose=None, return_alpha=False):
            calls.append(np.array(pose))
            return pose, np.zeros((3, 4, 6), np.float32), np.ones((4, 6), np.float32), np.ones((4, 6), np.float32)

        def corresp(gs_renderings, svd_outputs, dist_thresh, desc_only):
            i = len(calls) - 1
            assert dist_thresh == 3 and desc_only is False
            return [[torch.full((4, 6), float(means[i]))]], None

        class Dust3r:
            def to(self, dev):
                rec.setdefault("to", []).append(dev)

            def make_pairs(self, imgs, scene_graph, global_image_inds):
                rec.update(pair_frames=np.array([float(f[0, 0, 0]) / 255.0 for f in imgs], np.float32), scene_graph=scene_graph,
                           pair_inds=np.array(global_image_inds, np.int64))
                return "pairs"

            def run(self, frames, c2w_poses, intrinsics, preset_pairs):
                assert preset_pairs == "pairs"
                rec.update(run_frames=np.array([float(f[0, 0, 0]) / 255.0 for f in frames], np.float32),
                           c2w=np.array(c2w_poses), K=np.array(intrinsics))
                return None, "trimesh_scene"

        K = np.array([[500.0, 0, 320.0], [0, 510.0, 240.0], [0, 0, 1]], np.float32)
        me = types.SimpleNamespace(render_GS=render_GS, gsTrainer=types.SimpleNamespace(generate_corresp_mask=corresp),
                                   dust3r=Dust3r(), gs_intrinsics=K, gs_width=640)
        frames = [torch.full((3, 4, 6), float(v)) for v in cap["frames"]]
        res = DiffusionGS.densify_pcds(me, frames, list(cap["poses"]), key_frame_mask=list(cap["key_frame_mask"]),
                                       input_flags=list(cap["input_flags"]), win_samples=-1)
        assert res == "trimesh_scene" and rec["scene_graph"] == "complete" and rec["to"] == ["cuda", "cpu"]
        for k in ("pair_frames", "pair_inds", "run_frames", "c2w", "K"):
            out[f"{name}_pcd_{k}"] = rec[k]
    np.savez_compressed(GOLD / "n2_bookkeeping.npz", **out)
    print("n2", {k: getattr(v, "shape", ()) for k, v in out.items()})


def main():
    GOLD.mkdir(parents=True, exist_ok=True)
    Sch, consistency, forward_warp, inverse_warp = _import_reference()
    which = sys.argv[1:] or ["warp", "sched", "unet", "vae", "pipeline", "orch", "n2"]
    if "pipeline" in which:
        gen_pipeline()
    if "pipeline_one_pass" in which:
        gen_pipeline_one_pass()
    if "pipeline_unet" in which:          # ~5 min of CPU: not part of the default set
        gen_pipeline_real_unet()
    if "pipeline_full" in which:          # ~15 min of CPU, 1.52 B parameters
        gen_pipeline_full()
    if "nograd_check" in which:           # ~5 min of CPU: the no-grad UNet wrapper reproduces the with-graph Post fixture
        gen_nograd_check()
    if "pipeline_full_post" in which:     # ~25 min of CPU: the Post variant at full size behind the no-grad wrapper
        gen_pipeline_full_post()
    if "pipeline_unet_vae" in which:      # ~10 min of CPU
        gen_pipeline_real_unet_vae()
    if "clip" in which:
        gen_clip_preprocess()
    if "orch" in which:
        gen_orchestrator()
    if "orch_fusion" in which:            # seconds: the reference's own statements on four 576 x 1024 frames
        gen_orch_fusion()
    if "train_flags" in which:
        gen_train_flags()
    if "orch_nearby" in which:            # ~1 min of CPU: eight full-size reference inverse warps
        gen_orch_nearby()
    if "n2" in which:
        gen_n2()
    if "unet" in which:
        gen_unet()
    if "unet_full" in which:              # ~30 min of CPU, 1.52 B parameters: not part of the default set
        gen_unet_full([w for w in which if w in UNET_FULL_CASES] or None)
    if "vae" in which:
        gen_vae()
    if "vae_full" in which:               # a few minutes of CPU: not part of the default set
        gen_vae_full()
    if "warp" in which:
        gen_warp(consistency, forward_warp, inverse_warp)
    if "sched" in which:
        gen_sched(Sch)


if __name__ == "__main__":
    main()
