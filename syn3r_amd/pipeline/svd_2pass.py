"""Two-pass Stable-Video-Diffusion completion pipelines on the HIP path.

Mirrors of the reference's
  model/SVD_2pass_prob_uncertain.py       ("Replace": soft replacement of x0, :661-742)
  model/SVD_2pass_prob_uncertain_post.py  ("Post": tile-wise guidance gradient + plain Euler, :671-831)
with the same class name, `__call__` signature and output object.  The denoising loop — the hot
path — runs on the HIP UNet (`syn3r_amd.unet.model`) and the fused scheduler steps
(`syn3r_amd.schedulers`).  The CLIP image encoder and the temporal VAE are OUT OF SCOPE rows
(SURVEY.md §2, N1): the pipeline takes them as objects with the reference's interface
(`image_encoder(x).image_embeds`, `vae.encode(x).latent_dist.mode()`, `vae.decode(z, num_frames=…)`)
and calls them exactly where the reference does.

Differences from the reference that do not change results:
  * the "Post" gradient pass does not run autograd through the UNet: the UNet input is detached
    (…post.py:732), so the gradient is the closed form implemented by `syn3r_step_interp`
     — saves ≈96 TFLOP per (step, pass);
  * the forward- and backward-in-time passes of a step run as ONE stack of UNet launches (`merge_passes`, default on
    with the HIP UNet): same per-sample arithmetic, weights stream once per step;
  * `one_pass=True` runs the forward-in-time pass only (BASELINE.json configs[1] "SVD_1pass");
  * `num_frames` is a parameter (the reference asserts 25, …post.py:531).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Union

import numpy as np
import torch

from .. import _lib as L
from ..tuning import FLAGS as TUNE


@dataclass
class StableVideoDiffusionPipelineOutput:
    frames: object


_SIDE_STREAMS = {}        # device index -> the two side streams of the per-pass UNet calls (see StableVideoDiffusionPipeline._fork)


def _append_dims(x, target_dims):
    return x[(...,) + (None,) * (target_dims - x.ndim)]


def preprocess_images(images, height: int, width: int) -> torch.Tensor:
    """VaeImageProcessor.preprocess for the inputs SYN3R passes (PIL / HWC uint8-or-float numpy /
    CHW float tensors in [0,1]): resize if needed, scale to [-1, 1], NCHW float32."""
    if not isinstance(images, (list, tuple)):
        images = [images]
    out = []
    for im in images:
        if isinstance(im, torch.Tensor):
            t = im.float()
            if t.dim() == 3:
                t = t[None]
        elif isinstance(im, np.ndarray):
            a = im.astype(np.float32) / (255.0 if im.dtype == np.uint8 else 1.0)
            t = torch.from_numpy(a).permute(2, 0, 1)[None]
        else:  # PIL
            if im.size != (width, height):
                from PIL import Image
                im = im.resize((width, height), resample=Image.LANCZOS)
            t = torch.from_numpy(np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0).permute(2, 0, 1)[None]
        if t.shape[-2:] != (height, width):
            t = torch.nn.functional.interpolate(t, size=(height, width), mode="bilinear", align_corners=False)
        out.append(2.0 * t - 1.0)
    dev = next((o.device for o in out if o.is_cuda), out[0].device)   # device tensors may be mixed with host images
    return torch.cat([o.to(dev) for o in out], 0)


# overlapping tiles of the 72x128 latent grid used by the Post variant (…post.py:739-758) and the
# offsets at which their gradients are stitched back (:776-778)
def post_tiles(h: int, w: int):
    """Rows [:40] | [24:], columns [:72] | [56:] on the 72x128 grid, scaled for other grids
    (5/9, 1/3 of the height; 9/16, 7/16 of the width).  Returns (tiles, overlap_y, overlap_x)."""
    if h % 9 or w % 16:
        raise ValueError(f"latent grid {h}x{w} must be divisible by 9x16 for the 4-tile guidance pass")
    th, y1 = 5 * h // 9, h // 3
    tw, x1 = 9 * w // 16, 7 * w // 16
    tiles = [(slice(0, th), slice(0, tw)), (slice(y1, h), slice(0, tw)), (slice(0, th), slice(x1, w)),
             (slice(y1, h), slice(x1, w))]
    return tiles, th - y1, tw - x1


class StableVideoDiffusionPipeline:
    """`variant="replace"` ≙ model/SVD_2pass_prob_uncertain.py, `variant="post"` ≙ …_post.py."""

    def __init__(self, vae, image_encoder, unet, scheduler, feature_extractor=None, variant: str = "post",
                 device: Union[str, torch.device] = "cuda:0"):
        if variant not in ("replace", "post"):
            raise NotImplementedError(f"unknown variant {variant}")
        self.vae, self.image_encoder, self.unet, self.scheduler = vae, image_encoder, unet, scheduler
        self.feature_extractor = feature_extractor
        self.variant = variant
        self.device = torch.device(device)
        self.vae_scale_factor = 8
        self._guidance_scale = None
        self.use_graphs = TUNE["unet_graph"]     # replay captured UNet launch sequences (hipGraph)
        # the two passes of a Replace step on TWO HIP streams (see _streamed_replace): tuning.FLAGS["two_streams"] = False turns it off
        self.two_streams = TUNE["two_streams"] and self.device.type == "cuda"
        self._side = None               # the two side streams, created on first use
        self._streams_warm = set()      # shapes whose shared caches (frame-position embeddings, scratch buffers) exist

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype: torch.dtype = torch.float16, variant: Optional[str] = "fp16",
                        *, pipeline: str = "post", device: Union[str, torch.device] = "cuda:0", **_ignored):
        """The loader call of the reference, `StableVideoDiffusionPipeline.from_pretrained("stabilityai/stable-video-diffusion-
        img2vid-xt", torch_dtype=torch.float16, variant="fp16")` (model/diffusionGS.py:1089), for a LOCAL diffusers-layout
        directory (there is no fetch by name here):
            unet/            config.json + diffusion_pytorch_model[.fp16].safetensors   -> HIP UNetSpatioTemporalConditionModel
            vae/             config.json + diffusion_pytorch_model[.fp16].safetensors   -> HIP AutoencoderKLTemporalDecoder
            scheduler/       scheduler_config.json                                      -> EulerDiscreteScheduler.from_config
            image_encoder/   CLIP vision tower (through `transformers`, SURVEY section 2: out of the HIP scope)
            feature_extractor/preprocessor_config.json  (CLIP mean / std)
        `variant` picks `*.{variant}.safetensors` first, as diffusers does; `pipeline` = "post" | "replace" selects which of the
        reference's two pipeline modules this instance mirrors.  The scheduler's configuration is kept (`scheduler_config`) so
        callers can build a fresh scheduler per call, as the reference builds a fresh pipeline per `svd_render`."""
        import json
        from pathlib import Path
        from ..schedulers.scheduling_euler_discrete import EulerDiscreteScheduler
        from ..unet.model import UNetSpatioTemporalConditionModel
        from ..vae.model import AutoencoderKLTemporalDecoder
        from .clip import ClipImageEncoder
        d = Path(pretrained_model_name_or_path)
        if not d.is_dir():
            raise FileNotFoundError(f"{d} is not a local checkpoint directory (this build never fetches "
                                    "'stabilityai/stable-video-diffusion-img2vid-xt' by name: download it and pass the path)")
        missing = [n for n in ("unet", "vae", "scheduler", "image_encoder") if not (d / n).is_dir()]
        if missing:
            raise FileNotFoundError(f"{d}: missing sub-directories {missing} (diffusers layout expected)")
        if torch_dtype != torch.float16:
            raise NotImplementedError("the HIP UNet / VAE store fp16 (torch_dtype=torch.float16, as the reference passes)")
        dev = torch.device(device)
        unet = UNetSpatioTemporalConditionModel.from_pretrained(str(d / "unet"), dev, variant=variant)
        vae = AutoencoderKLTemporalDecoder.from_pretrained(str(d / "vae"), dev, variant=variant)
        sched_cfg = json.loads((d / "scheduler" / "scheduler_config.json").read_text())
        enc = ClipImageEncoder.from_pretrained(d / "image_encoder", d / "feature_extractor", dev, torch_dtype)
        pipe = cls(vae, enc, unet, EulerDiscreteScheduler.from_config(sched_cfg), variant=pipeline, device=dev)
        pipe.scheduler_config = sched_cfg
        return pipe

    @property
    def guidance_scale(self):
        return self._guidance_scale

    # ------------------------------------------------------------------ conditioning (out-of-scope modules, called as the reference does)
    def _encode_image(self, image, do_cfg: bool) -> torch.Tensor:
        """…post.py:229-273 with a caller-supplied CLIP: `image_encoder` must map the pipeline's
        image (PIL / tensor) to `.image_embeds` [1, 1024] (the antialiased 224x224 resize and CLIP
        normalisation live with that module)."""
        emb = self.image_encoder(image).image_embeds.to(self.device)
        emb = emb.unsqueeze(1)
        if do_cfg:
            emb = torch.cat([torch.zeros_like(emb), emb])
        return emb

    def _encode_vae_image(self, image: torch.Tensor, do_cfg: bool) -> torch.Tensor:
        """…post.py:275-298"""
        lat = self.vae.encode(image.to(self.device)).latent_dist.mode()
        if do_cfg:
            lat = torch.cat([torch.zeros_like(lat), lat])
        return lat

    def _get_add_time_ids(self, fps, motion_bucket_id, noise_aug_strength, dtype, do_cfg):
        ids = torch.tensor([[fps, motion_bucket_id, noise_aug_strength]], dtype=dtype)
        if do_cfg:
            ids = torch.cat([ids, ids])
        return ids

    def decode_latents(self, latents, num_frames, decode_chunk_size=14):
        """…post.py:326-353 (VAE decode, out-of-scope module)."""
        latents = latents.flatten(0, 1)
        latents = 1 / self.vae.config.scaling_factor * latents
        frames = []
        for i in range(0, latents.shape[0], decode_chunk_size):
            n = latents[i:i + decode_chunk_size].shape[0]
            frames.append(self.vae.decode(latents[i:i + decode_chunk_size], num_frames=n).sample)
        frames = torch.cat(frames, dim=0)
        frames = frames.reshape(-1, num_frames, *frames.shape[1:]).permute(0, 2, 1, 3, 4)
        return frames.float()

    # ------------------------------------------------------------------ the hot loop
    def _unet(self, x, t, ehs, added, **kw):
        if self.use_graphs and hasattr(self.unet, "forward_graphed"):      # captured launch sequence per (shape, context)
            return self.unet.forward_graphed(x, t, ehs, added, **kw)[0]
        return self.unet(x, t, encoder_hidden_states=ehs, added_time_ids=added, return_dict=False, **kw)[0]

    def _cfg(self, noise_pred, do_cfg):
        if not do_cfg:
            return noise_pred
        u, c = noise_pred.chunk(2)
        return u + self.guidance_scale * (c - u)

    def _model_input(self, i, t, latents, image_latents, do_cfg):
        """…post.py:700-703: CFG duplication, `scale_model_input`, the conditioning-image latents on the channel axis."""
        x = torch.cat([latents] * 2) if do_cfg else latents
        x = self.scheduler.scale_model_input(x, t, step_i=i).to(latents.dtype)
        return torch.cat([x, image_latents], dim=2)

    @staticmethod
    def _tile_operands(cond, mask):
        """The guidance tiles' views of one pass's conditioning latents and mask (…post.py:739-758), cut ONCE per pass: the
        scheduler keeps the kernel-side form of these objects per object, so handing it the same tensors at every step
        is what lets that cache hit (100 steps x 4 tiles)."""
        h, w = cond.shape[-2:]
        tiles, ov_y, ov_x = post_tiles(h, w)
        ops_ = []
        for ty, tx in tiles:
            ops_.append((cond[:2, :, :, ty, tx].contiguous(), mask[0:1, :, :, ty, tx].contiguous()))
        return tiles, ov_y, ov_x, ops_

    @staticmethod
    def _stitch(grads, ov_y, ov_x):
        """…post.py:776-778"""
        g1 = torch.cat((grads[0], grads[1][:, :, :, ov_y:, :]), -2)
        g2 = torch.cat((grads[2], grads[3][:, :, :, ov_y:, :]), -2)
        return torch.cat((g1, g2[:, :, :, :, ov_x:]), -1)

    def _pass_replace(self, i, t, latents, image_latents, ehs, added, cond, mask, lam, do_cfg, tile_ops=None):
        """SVD_2pass_prob_uncertain.py:691-716"""
        x = self._model_input(i, t, latents, image_latents, do_cfg)
        noise_pred = self._cfg(self._unet(x, t, ehs, added), do_cfg)
        return self.scheduler.step_interp_prob_uncertain(noise_pred, t, latents, cond, mask, lam, step_i=i).prev_sample

    def _pass_post(self, i, t, latents, image_latents, ehs, added, cond, mask, lam, do_cfg, tile_ops=None):
        """…post.py:700-800"""
        sch = self.scheduler
        x = self._model_input(i, t, latents, image_latents, do_cfg)
        tiles, ov_y, ov_x, tops = tile_ops if tile_ops is not None else self._tile_operands(cond, mask)
        # :726-774 — four B = 1 forwards of the unconditional half in the reference.  Tiles 0/2 and 1/3 have equal
        # shapes and share the same (unconditional) context and time ids, so each pair runs as ONE batch-of-2 forward:
        # identical per-sample arithmetic (GroupNorm and attention are per sample; the reference's batch-interleaved
        # temporal context is the same vector for both), larger contractions.
        grads = [None] * 4
        for pair in ((0, 2), (1, 3)):
            xb = torch.cat([x[0:1, :, :, tiles[k][0], tiles[k][1]] for k in pair], dim=0).contiguous()
            noise = self._unet(xb, t, ehs[0:1].expand(2, -1, -1), added[0:1].expand(2, -1).contiguous())   # stride-0 context
            for n, k in enumerate(pair):
                out = sch.step_interp(noise[n:n + 1], t, latents[0:1, :, :, tiles[k][0], tiles[k][1]].contiguous(), tops[k][0],
                                      tops[k][1], lam, step_i=i, lr=0.02, compute_grad=True)
                grads[k] = out.grad
        next_latents = latents - self._stitch(grads, ov_y, ov_x).half()    # :779
        noise_pred = self._cfg(self._unet(x, t, ehs, added), do_cfg)       # :786-792 (input built from the ORIGINAL latents)
        return sch.step_interp(noise_pred, t, next_latents, cond, mask, lam, step_i=i, compute_grad=False).prev_sample

    # ---- both passes of a step in ONE UNet call -------------------------------------------------------------------
    # The forward- and backward-in-time passes of a step read the same latents (time-flipped) and do not depend on each
    # other (…post.py:679-698, SVD_2pass_prob_uncertain.py:661-742).  Stacked on the batch axis they are one launch
    # sequence of twice the rows: every weight streams once per step instead of twice and the launches whose grids
    # left CUs idle (M = 4 032 / 9 000 / 10 800 rows) fill the chip.  Per-sample arithmetic is the reference's: the only
    # batch-coupled operator, the interleaved temporal cross-attention context, is applied per pass (`ctx_group`).
    def _merged_replace(self, i, t, lat, img4, ehs4, added4, ops2, do_cfg):
        sch = self.scheduler
        g = 2 if do_cfg else 1
        x = torch.cat([self._model_input(i, t, lat[k], img4[k * g:(k + 1) * g], do_cfg) for k in range(2)])
        noise = self._unet(x, t, ehs4, added4, ctx_group=g)
        out = []
        for k in range(2):
            cond, mask, lam, _ = ops2[k]
            pred = self._cfg(noise[k * g:(k + 1) * g], do_cfg)
            out.append(sch.step_interp_prob_uncertain(pred, t, lat[k], cond, mask, lam, step_i=i).prev_sample)
        return out

    # ---- independent launch sequences of a step on their own HIP streams ----------------------------------------------
    # A contraction launch of T tiles takes ceil(T / 256) rounds of the persistent kernels: at F = 25 the stacked step loses
    # 7.9 % of its contraction time to the last, partly empty round (1 800 tiles = 7.03 rounds take 8).
    # Launch sequences that do not depend on each other fill those rounds with each other's blocks when they run on
    # different streams.  Replace: the two passes (B = 2 each).  Post: each pass's CFG forward on its own stream beside the
    # stacked guidance-tile forwards on the current one (the CFG input is built from the ORIGINAL latents, …post.py:786-792).
    # Measured at F = 25, same box, ms per (step, pass) unit (tools/two_stream_units.py, profiles/r04/two_stream_units.txt):
    # Replace 179.5 -> 174.8, Post 300.2 -> 288.9.  ONE pair of side streams per device and process (HIP multiplexes streams
    # onto ~4 hardware queues: a second pair lands on occupied queues and serialises).  Results: the per-pass arithmetic of the
    # pass-after-pass order (Replace: bit-identical to it), the same on every call.
    def _fork(self, key):
        """(side stream 1, side stream 2), both waiting for the current stream - or the current stream twice for the FIRST call of
        `key`: that call creates what the sequences share (the UNet's frame-position embeddings of this batch size, folded
        contexts, the scheduler's kernel-side operands), and a cache entry written on one side stream must not be read on the
        other before it exists.  None: tuning.FLAGS["two_streams"] = False."""
        if not self.two_streams:
            return None
        if key not in self._streams_warm:
            self._streams_warm.add(key)
            cur = torch.cuda.current_stream(self.device)
            return (cur, cur)
        if self._side is None:
            # ONE pair of side streams per device for the whole process: HIP multiplexes streams onto a few hardware queues
            # (4 by default), and two streams that land on one queue run one after the other - a second pipeline object with
            # its own pair measured 369 ms where this pair gives 351 (tools/post_parts.py)
            key = self.device.index if self.device.index is not None else torch.cuda.current_device()
            if key not in _SIDE_STREAMS:
                _SIDE_STREAMS[key] = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
            self._side = _SIDE_STREAMS[key]
        cur = torch.cuda.current_stream(self.device)
        for s_ in self._side:
            s_.wait_stream(cur)
        return self._side

    def _join(self, streams):
        cur = torch.cuda.current_stream(self.device)
        for s_ in streams:
            if s_ is not cur:
                cur.wait_stream(s_)

    def _streamed_replace(self, i, t, lat, img4, ehs4, added4, ops2, do_cfg):
        """The two passes of a Replace step (SVD_2pass_prob_uncertain.py:661-742), each on its own stream."""
        g = 2 if do_cfg else 1
        streams = self._fork(("replace", tuple(lat[0].shape)))
        if streams is None:                                  # tuning.FLAGS["two_streams"] = False: both passes stacked into one launch sequence
            return self._merged_replace(i, t, lat, img4, ehs4, added4, ops2, do_cfg)
        out = []
        for k in range(2):
            cond, mask, lam, _ = ops2[k]
            with torch.cuda.stream(streams[k]):
                out.append(self._pass_replace(i, t, lat[k], img4[k * g:(k + 1) * g], ehs4[k * g:(k + 1) * g], added4[k * g:(k + 1) * g],
                                              cond, mask, lam, do_cfg))
        self._join(streams)
        return out

    def _merged_post(self, i, t, lat, img4, ehs4, added4, ops2, do_cfg, tile_ctx):
        sch = self.scheduler
        g = 2 if do_cfg else 1
        x = torch.cat([self._model_input(i, t, lat[k], img4[k * g:(k + 1) * g], do_cfg) for k in range(2)])
        tiles, ov_y, ov_x, _ = ops2[0][3]
        grads = [[None] * 4, [None] * 4]
        ehs_t, added_t, grp = tile_ctx
        # The CFG forwards do not depend on the guidance tiles (their input is built from the ORIGINAL latents, …post.py:786-792):
        # each pass's CFG forward goes to its own stream FIRST, the tile forwards follow on the current one - three launch
        # sequences whose kernels fill each other's partly empty last rounds (see above).  tuning.FLAGS["two_streams"] = False: one stacked
        # B = 4 CFG call behind the tiles.
        streams = self._fork(("post", tuple(lat[0].shape)))
        noises = None
        if streams is not None:
            noises = []
            for k in range(2):
                with torch.cuda.stream(streams[k]):
                    noises.append(self._unet(x[k * g:(k + 1) * g], t, ehs4[k * g:(k + 1) * g], added4[k * g:(k + 1) * g]))
        for pair in ((0, 2), (1, 3)):                              # the four tile forwards of BOTH passes: two B = 4 calls
            xb = torch.cat([x[k * g:k * g + 1, :, :, tiles[q][0], tiles[q][1]] for k in range(2) for q in pair], dim=0).contiguous()
            noise = self._unet(xb, t, ehs_t, added_t, ctx_group=grp)
            for k in range(2):
                tops = ops2[k][3][3]
                for n, q in enumerate(pair):
                    o = sch.step_interp(noise[2 * k + n:2 * k + n + 1], t, lat[k][0:1, :, :, tiles[q][0], tiles[q][1]].contiguous(),
                                        tops[q][0], tops[q][1], ops2[k][2], step_i=i, lr=0.02, compute_grad=True)
                    grads[k][q] = o.grad
        if streams is None:
            noise = self._unet(x, t, ehs4, added4, ctx_group=g)    # the CFG forwards of both passes: one B = 4 call
            noises = [noise[k * g:(k + 1) * g] for k in range(2)]
        else:
            self._join(streams)
        out = []
        for k in range(2):
            cond, mask, lam, _ = ops2[k]
            nxt = lat[k] - self._stitch(grads[k], ov_y, ov_x).half()
            pred = self._cfg(noises[k], do_cfg)
            out.append(sch.step_interp(pred, t, nxt, cond, mask, lam, step_i=i, compute_grad=False).prev_sample)
        return out

    @torch.no_grad()
    def denoise(self, latents, image_latent_start, image_latent_end, emb_start, emb_end, added_time_ids,
                temp_cond_latents, mask, lambda_ts, num_inference_steps: int, min_guidance_scale=1.0,
                max_guidance_scale=3.0, one_pass: bool = False, callback: Optional[Callable] = None,
                merge_passes: Optional[bool] = None):
        """The denoising loop (…post.py:656-831 / SVD_2pass_prob_uncertain.py:649-748) on prepared
        tensors.  latents [1,F,4,h,w]; image_latent_* [B,F,4,h,w]; emb_* [B,1,D]; temp_cond_latents
        [2,F,4,h,w] fp32 (already divided by factor_s); mask [1,F-2,4,h,w]; lambda_ts [steps,F] f64.
        `merge_passes` (default: on when the UNet takes `ctx_group`, i.e. the HIP UNet; tuning.FLAGS["merge_passes"] = False turns it off):
        the two passes of a step share their UNet launches."""
        dev = self.device
        F = latents.shape[1]
        do_cfg = max_guidance_scale > 1.0
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        timesteps = self.scheduler.timesteps
        gs = torch.linspace(min_guidance_scale, max_guidance_scale, F).unsqueeze(0).to(dev, latents.dtype)
        self._guidance_scale = _append_dims(gs, latents.ndim)
        weight_fw = torch.linspace(1, 0, F)[None, :, None, None, None].to(device=dev, dtype=latents.dtype)
        post = self.variant == "post"
        step = self._pass_post if post else self._pass_replace
        mask = mask.to(dev)
        lambda_ts = lambda_ts.detach().to("cpu", torch.float64)     # read on the host, one row per step: never a device sync
        cond_bw, mask_bw, lam_bw = temp_cond_latents.flip(dims=[1]), mask.flip(dims=[1]), lambda_ts.flip(dims=[1])
        tops_fw = self._tile_operands(temp_cond_latents, mask) if post else None
        tops_bw = self._tile_operands(cond_bw, mask_bw) if post and not one_pass else None
        if merge_passes is None:
            merge_passes = TUNE["merge_passes"]
        merged = bool(merge_passes) and not one_pass and getattr(self.unet, "supports_ctx_group", False)
        if merged:
            g = 2 if do_cfg else 1
            img4 = torch.cat([image_latent_start, image_latent_end]).contiguous()
            ehs4 = torch.cat([emb_start, emb_end]).contiguous()                 # built once: the UNet keys its folded contexts on it
            added4 = torch.cat([added_time_ids, added_time_ids]).contiguous()
            ops2 = ((temp_cond_latents, mask, lambda_ts, tops_fw), (cond_bw, mask_bw, lam_bw, tops_bw))
            if post:                                                            # the tiles run the UNCONDITIONAL half (…post.py:759-760)
                if torch.equal(emb_start[0:1], emb_end[0:1]):                   # (zeros for both passes: one shared context)
                    tile_ctx = (emb_start[0:1].expand(4, -1, -1), added_time_ids[0:1].expand(4, -1).contiguous(), None)
                else:
                    tile_ctx = (torch.cat([emb_start[0:1]] * 2 + [emb_end[0:1]] * 2).contiguous(),
                                added_time_ids[0:1].expand(4, -1).contiguous(), 1)
        for i, t in enumerate(timesteps):
            if merged:
                lat = (latents, latents.flip(dims=[1]))
                fw, bw = (self._merged_post(i, t, lat, img4, ehs4, added4, ops2, do_cfg, tile_ctx) if post else
                          self._streamed_replace(i, t, lat, img4, ehs4, added4, ops2, do_cfg))
                latents = weight_fw * fw + (1 - weight_fw) * bw.flip(dims=[1])      # :828 / :736
            else:
                fw = step(i, t, latents, image_latent_start, emb_start, added_time_ids, temp_cond_latents, mask, lambda_ts,
                          do_cfg, tops_fw)
                if one_pass:
                    latents = fw
                else:
                    bw = step(i, t, latents.flip(dims=[1]), image_latent_end, emb_end, added_time_ids, cond_bw, mask_bw,
                              lam_bw, do_cfg, tops_bw)
                    latents = weight_fw * fw + (1 - weight_fw) * bw.flip(dims=[1])  # :828 / :736
            if callback is not None:
                callback(i, t, latents)
        return latents

    @torch.no_grad()
    def __call__(self, image, temp_cond, mask, lambda_ts, height: int = 576, width: int = 1024,
                 num_frames: Optional[int] = None, num_inference_steps: int = 25, min_guidance_scale: float = 1.0,
                 max_guidance_scale: float = 3.0, fps: int = 7, motion_bucket_id: int = 127,
                 noise_aug_strength: float = 0.02, decode_chunk_size: Optional[int] = None,
                 num_videos_per_prompt: Optional[int] = 1, generator=None, latents: Optional[torch.Tensor] = None,
                 output_type: Optional[str] = "pil", callback_on_step_end=None, return_dict: bool = True,
                 latent_num: int = 1, one_pass: bool = False, dtype: torch.dtype = torch.float16,
                 aug_noise: Optional[torch.Tensor] = None, merge_passes: Optional[bool] = None):
        if callback_on_step_end is not None:
            raise NotImplementedError
        if latent_num != 1 or num_videos_per_prompt != 1:
            raise NotImplementedError("latent_num / num_videos_per_prompt other than 1 are unused by SYN3R")
        dev = self.device
        num_frames = num_frames if num_frames is not None else 25
        decode_chunk_size = decode_chunk_size if decode_chunk_size is not None else num_frames
        if not isinstance(image, (list, tuple)):
            image = [image]
        if len(temp_cond) != num_frames - 1:
            raise ValueError(f"temp_cond must hold {num_frames - 1} images (warped views + the end view)")
        do_cfg = max_guidance_scale > 1.0
        emb_start = self._encode_image(image[0], do_cfg).to(dtype)            # :544
        emb_end = self._encode_image(temp_cond[-1], do_cfg).to(dtype)         # :546
        fps = fps - 1
        img = preprocess_images(image, height, width).to(dev)
        img_end = preprocess_images(temp_cond[-1:], height, width).to(dev)
        mask = mask.to(dev).unsqueeze(1).unsqueeze(0).repeat(1, 1, 4, 1, 1)   # :557-558
        tc = preprocess_images(list(temp_cond), height, width).to(dev)
        if aug_noise is not None:          # extension: caller-supplied augmentation noise (…post.py:583 draws it)
            noise = aug_noise.to(dev, img.dtype)
        elif generator is not None:
            noise = torch.randn(img.shape, generator=generator, dtype=img.dtype).to(dev)
        else:
            noise = torch.randn(img.shape, device=dev, dtype=img.dtype)

        def to_latents(x, out_dtype):                                          # :568-580
            x = x + noise_aug_strength * noise
            return self._encode_vae_image(x, do_cfg).to(out_dtype).unsqueeze(1)

        lat_start = to_latents(img[0:1], dtype)
        lat_end = to_latents(img_end[0:1], dtype)
        cond = torch.cat([to_latents(tc[k:k + 1], torch.float32) for k in range(tc.shape[0])], dim=1)
        cond = torch.cat((lat_start.float(), cond), dim=1) / 5.6                # :601-610 (factor_s)
        lat_start = lat_start.repeat(1, num_frames, 1, 1, 1)
        lat_end = lat_end.repeat(1, num_frames, 1, 1, 1)
        added = self._get_add_time_ids(fps, motion_bucket_id, noise_aug_strength, dtype, do_cfg).to(dev)
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        shape = (1, num_frames, 4, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            latents = torch.randn(shape, generator=generator, dtype=dtype).to(dev) if generator is not None else \
                torch.randn(shape, device=dev, dtype=dtype)
        latents = latents.to(dev, dtype) * self.scheduler.init_noise_sigma      # prepare_latents
        latents = self.denoise(latents, lat_start, lat_end, emb_start, emb_end, added, cond, mask, lambda_ts,
                               num_inference_steps, min_guidance_scale, max_guidance_scale, one_pass=one_pass,
                               merge_passes=merge_passes)
        if output_type == "latent":
            frames = latents
        else:
            frames = self.decode_latents(latents.float(), num_frames, decode_chunk_size)
            frames = tensor2vid(frames, output_type)
        if not return_dict:
            return frames
        return StableVideoDiffusionPipelineOutput(frames=frames)


def tensor2vid(video: torch.Tensor, output_type: str = "np"):
    """[B,C,F,H,W] in [-1,1] -> per batch item a [F,H,W,C] array in [0,1] (or PIL list)."""
    outs = []
    for b in range(video.shape[0]):
        v = ((video[b].permute(1, 2, 3, 0) / 2 + 0.5).clamp(0, 1)).cpu().numpy()
        if output_type == "pil":
            from PIL import Image
            v = [Image.fromarray((f * 255).round().astype("uint8")) for f in v]
        outs.append(v)
    return outs
