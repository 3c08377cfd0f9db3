"""Temporal-decoder VAE of Stable Video Diffusion on the HIP operators.

Mirror of `diffusers.AutoencoderKLTemporalDecoder`
(`thirdparty/diffusers/src/diffusers/models/autoencoders/autoencoder_kl_temporal_decoder.py:164-399`):
same constructor arguments, same parameter names (a diffusers checkpoint loads unchanged), same
`encode(x).latent_dist` / `decode(z, num_frames).sample` surface the SVD pipelines call
(`model/SVD_2pass_prob_uncertain_post.py:283,343`).  Graph restated from
  Encoder                      `autoencoders/vae.py:46-196`
  DownEncoderBlock2D           `unets/unet_2d_blocks.py:1395-1471`      (Downsample2D padding 0 = pad (0,1,0,1))
  UNetMidBlock2D + Attention   `unets/unet_2d_blocks.py:585-740`, `attention_processor.py:1222-1299`
  TemporalDecoder              `autoencoder_kl_temporal_decoder.py:29-161`
  MidBlock/UpBlockTemporalDecoder  `unets/unet_3d_blocks.py:1766-1900`
  SpatioTemporalResBlock, AlphaBlender (switch_spatial_to_temporal_mix)  `resnet.py:640-805`

Every activation is a channels-last fp16 token matrix [(n*h + y)*w + x, C] (as in the UNet), so the
reference's [B,C,F,H,W] permutes are index arithmetic inside the kernels.  The reference runs this model
in fp32 (`force_upcast`); here storage is fp16 with fp32 accumulation, and the parity tests state the
tolerance.  No CPU fallback.
"""
from __future__ import annotations

import json
import math
from pathlib import Path
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as Fn

from ..unet import ops
from ..unet.model import _Params

H = torch.float16


class DiagonalGaussianDistribution:
    """`autoencoders/vae.py:682-740`: moments [N, 2C, h, w] -> mean / clamped logvar."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        noise = torch.randn(self.mean.shape, generator=generator, device=self.parameters.device,
                            dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKLTemporalDecoder:
    def __init__(self, in_channels: int = 3, out_channels: int = 3,
                 down_block_types: Tuple[str, ...] = ("DownEncoderBlock2D",), block_out_channels: Tuple[int, ...] = (64,),
                 layers_per_block: int = 1, latent_channels: int = 4, sample_size: int = 32,
                 scaling_factor: float = 0.18215, force_upcast: bool = True):
        if any(t != "DownEncoderBlock2D" for t in down_block_types) or len(down_block_types) != len(block_out_channels):
            raise NotImplementedError("only DownEncoderBlock2D encoders (the SVD VAE) are supported")
        if in_channels != 3 or out_channels != 3:
            raise NotImplementedError("time_conv_out kernel is written for 3 image channels")
        if any(c % 64 for c in block_out_channels):
            raise ValueError("block_out_channels must be multiples of 64 (contraction k-tile)")
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels,
                                      down_block_types=tuple(down_block_types), block_out_channels=tuple(block_out_channels),
                                      layers_per_block=layers_per_block, latent_channels=latent_channels,
                                      sample_size=sample_size, scaling_factor=scaling_factor, force_upcast=force_upcast)
        self.dtype = torch.float32          # the pipelines feed and read fp32 (the reference upcasts the VAE)
        self.device = None
        self.p = _Params()
        self._declare()
        self.packed: Dict[str, torch.Tensor] = {}

    # ------------------------------------------------------------------ parameter table (diffusers names)
    def _decl_conv(self, pre: str, cin: int, cout: int, k: int = 3):
        self.p.declare(pre + ".weight", cout, cin, k, k)
        self.p.declare(pre + ".bias", cout)

    def _decl_resnet(self, pre: str, cin: int, cout: int):
        p = self.p
        p.norm(pre + ".norm1", cin); self._decl_conv(pre + ".conv1", cin, cout)
        p.norm(pre + ".norm2", cout); self._decl_conv(pre + ".conv2", cout, cout)
        if cin != cout:
            self._decl_conv(pre + ".conv_shortcut", cin, cout, 1)

    def _decl_st_resblock(self, pre: str, cin: int, cout: int):
        p = self.p
        self._decl_resnet(pre + ".spatial_res_block", cin, cout)
        t = pre + ".temporal_res_block"
        p.norm(t + ".norm1", cout); p.declare(t + ".conv1.weight", cout, cout, 3, 1, 1); p.declare(t + ".conv1.bias", cout)
        p.norm(t + ".norm2", cout); p.declare(t + ".conv2.weight", cout, cout, 3, 1, 1); p.declare(t + ".conv2.bias", cout)
        p.declare(pre + ".time_mixer.mix_factor", 1)

    def _decl_attn(self, pre: str, c: int):
        p = self.p
        p.norm(pre + ".group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            p.linear(pre + "." + n, c, c)

    def _declare(self):
        c, boc = self.config, self.config.block_out_channels
        self._decl_conv("encoder.conv_in", c.in_channels, boc[0])
        ch = boc[0]
        for i, co in enumerate(boc):
            for j in range(c.layers_per_block):
                self._decl_resnet(f"encoder.down_blocks.{i}.resnets.{j}", ch if j == 0 else co, co)
            ch = co
            if i != len(boc) - 1:
                self._decl_conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co)
        self._decl_resnet("encoder.mid_block.resnets.0", ch, ch)
        self._decl_attn("encoder.mid_block.attentions.0", ch)
        self._decl_resnet("encoder.mid_block.resnets.1", ch, ch)
        self.p.norm("encoder.conv_norm_out", ch)
        self._decl_conv("encoder.conv_out", ch, 2 * c.latent_channels)
        self._decl_conv("quant_conv", 2 * c.latent_channels, 2 * c.latent_channels, 1)
        # decoder
        top = boc[-1]
        self._decl_conv("decoder.conv_in", c.latent_channels, top)
        for j in range(c.layers_per_block):
            self._decl_st_resblock(f"decoder.mid_block.resnets.{j}", top, top)
        self._decl_attn("decoder.mid_block.attentions.0", top)
        rev = list(reversed(boc))
        prev = rev[0]
        for i, co in enumerate(rev):
            for j in range(c.layers_per_block + 1):
                self._decl_st_resblock(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
            if i != len(rev) - 1:
                self._decl_conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co)
            prev = co
        self.p.norm("decoder.conv_norm_out", boc[0])
        self._decl_conv("decoder.conv_out", boc[0], c.out_channels)
        self.p.declare("decoder.time_conv_out.weight", c.out_channels, c.out_channels, 3, 1, 1)
        self.p.declare("decoder.time_conv_out.bias", c.out_channels)

    def parameter_shapes(self) -> Dict[str, Tuple[int, ...]]:
        return dict(self.p.shapes)

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, torch.Tensor], device) -> "AutoencoderKLTemporalDecoder":
        missing = sorted(set(self.p.shapes) - set(sd))
        extra = sorted(set(sd) - set(self.p.shapes))
        if missing or extra:
            raise KeyError(f"state_dict mismatch: missing {missing[:4]}… ({len(missing)}), unexpected {extra[:4]}… ({len(extra)})")
        dev = torch.device(device)
        if dev.type != "cuda":
            from .. import _lib
            raise _lib.Syn3rError("the HIP VAE needs a HIP device (no CPU fallback)")
        for k, shape in self.p.shapes.items():
            if tuple(sd[k].shape) != shape:
                raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {shape}")
            keep32 = k.startswith("quant_conv") or k.startswith("decoder.time_conv_out") or k.endswith("mix_factor")
            self.p.t[k] = sd[k].detach().to(dev, torch.float32 if keep32 else H).contiguous()
        self.device = dev
        self._pack()
        return self

    def init_random(self, device, seed: int = 0) -> "AutoencoderKLTemporalDecoder":
        """Seeded fan-in-scaled weights (benchmarks / smoke: no checkpoint is reachable offline)."""
        dev = torch.device(device)
        g = torch.Generator(device=dev).manual_seed(seed)
        sd = {}
        for k, shape in self.p.shapes.items():
            if k.endswith("mix_factor"):
                sd[k] = torch.zeros(shape, device=dev)
            elif "norm" in k.split(".")[-2] and k.endswith(".weight"):
                sd[k] = torch.ones(shape, device=dev)
            elif k.endswith(".bias"):
                sd[k] = torch.zeros(shape, device=dev)
            else:
                sd[k] = torch.randn(shape, generator=g, device=dev) * math.prod(shape[1:]) ** -0.5
        return self.load_state_dict(sd, dev)

    @classmethod
    def from_pretrained(cls, path: str, device, variant: Optional[str] = "fp16") -> "AutoencoderKLTemporalDecoder":
        """`path`: a local diffusers `vae/` directory (config.json + *.safetensors)."""
        from safetensors.torch import load_file
        d = Path(path)
        cfg = json.loads((d / "config.json").read_text())
        keys = ("in_channels out_channels down_block_types block_out_channels layers_per_block latent_channels "
                "sample_size scaling_factor force_upcast").split()
        model = cls(**{k: cfg[k] for k in keys if k in cfg})
        names = ([f"diffusion_pytorch_model.{variant}.safetensors"] if variant else []) + ["diffusion_pytorch_model.safetensors"]
        for name in names:
            if (d / name).exists():
                return model.load_state_dict(load_file(str(d / name)), device)
        raise FileNotFoundError(f"no safetensors weights under {d} (looked for {names})")

    def _pack(self):
        """Kernel-side layouts: OHWI 3x3 weights (Cin padded to 64, Cout to 8), 1x1 convs as matrices,
        (3,1,1) convs as [Cout,3,Cin]."""
        pk = {}
        for k, t in self.p.t.items():
            if not k.endswith(".weight") or k.startswith("quant_conv") or k.startswith("decoder.time_conv_out"):
                continue
            if t.dim() == 4 and t.shape[-1] == 3:
                w = t
                if w.shape[1] % 64:
                    w = Fn.pad(w, (0, 0, 0, 0, 0, 64 - w.shape[1] % 64))
                if w.shape[0] % 8:
                    w = Fn.pad(w, (0, 0, 0, 0, 0, 0, 0, 8 - w.shape[0] % 8))
                    b = self.p.t[k[:-6] + "bias"]
                    pk[k[:-6] + "bias"] = Fn.pad(b, (0, 8 - b.shape[0] % 8))
                pk[k] = w.permute(0, 2, 3, 1).contiguous()
            elif t.dim() == 4 and t.shape[-1] == 1:
                pk[k] = t.reshape(t.shape[0], t.shape[1]).contiguous()
            elif t.dim() == 5:
                pk[k] = t[..., 0, 0].permute(0, 2, 1).contiguous()
        self.packed = pk
        # host-side constants, read back ONCE (a device-scalar read per block would drain the launch queue)
        self.one_minus_alpha = {k: float(torch.sigmoid(t.float())) for k, t in self.p.t.items() if k.endswith("mix_factor")}
        self.time_conv_host = (self.p.t["decoder.time_conv_out.weight"].float().cpu(),
                               self.p.t["decoder.time_conv_out.bias"].float().cpu())

    def w(self, name: str) -> torch.Tensor:
        return self.packed.get(name, self.p.t.get(name))

    # ------------------------------------------------------------------ blocks
    def _resnet(self, pre: str, x: torch.Tensor, N: int, h: int, w_: int, cin: int, cout: int, eps: float = 1e-6):
        W = self.w
        hcur = ops.groupnorm(x, W(pre + ".norm1.weight"), W(pre + ".norm1.bias"), N, eps, True)
        hcur = ops.conv3x3(hcur.view(N, h, w_, cin), W(pre + ".conv1.weight"), W(pre + ".conv1.bias")).view(-1, cout)
        hcur = ops.groupnorm(hcur, W(pre + ".norm2.weight"), W(pre + ".norm2.bias"), N, eps, True)
        skip = x
        if cin != cout:
            skip = ops.linear(x, W(pre + ".conv_shortcut.weight"), W(pre + ".conv_shortcut.bias"))
        return ops.conv3x3(hcur.view(N, h, w_, cout), W(pre + ".conv2.weight"), W(pre + ".conv2.bias"),
                           residual=skip).view(-1, cout)

    def _st_resblock(self, pre: str, x: torch.Tensor, B: int, F: int, h: int, w_: int, cin: int, cout: int):
        """SpatioTemporalResBlock(temb None, eps 1e-6 / temporal 1e-5, 'learned', switch_spatial_to_temporal_mix)."""
        W, HW = self.w, h * w_
        xs = self._resnet(pre + ".spatial_res_block", x, B * F, h, w_, cin, cout, 1e-6)
        t = pre + ".temporal_res_block"
        hcur = ops.groupnorm(xs, W(t + ".norm1.weight"), W(t + ".norm1.bias"), B, 1e-5, True)
        hcur = ops.tconv3(hcur, W(t + ".conv1.weight"), W(t + ".conv1.bias"), B, F, HW)
        hcur = ops.groupnorm(hcur, W(t + ".norm2.weight"), W(t + ".norm2.bias"), B, 1e-5, True)
        # AlphaBlender with the switch: a = 1 - sigmoid(mix); out = a*xs + (1-a)*(xs + conv2) = xs + (1-a)*conv2
        one_minus_a = self.one_minus_alpha[pre + ".time_mixer.mix_factor"]
        return ops.tconv3(hcur, W(t + ".conv2.weight"), W(t + ".conv2.bias"), B, F, HW, residual=xs,
                          s_acc=one_minus_a, s_res=1.0)

    def _attention(self, pre: str, x: torch.Tensor, N: int, S: int, C: int) -> torch.Tensor:
        """Attention(heads 1, dim_head C, group norm, bias, residual_connection) per image."""
        W = self.w
        hs = ops.groupnorm(x, W(pre + ".group_norm.weight"), W(pre + ".group_norm.bias"), N, 1e-6, False)
        q = ops.linear(hs, W(pre + ".to_q.weight"), W(pre + ".to_q.bias"))
        k = ops.linear(hs, W(pre + ".to_k.weight"), W(pre + ".to_k.bias"))
        v = ops.linear(hs, W(pre + ".to_v.weight"), W(pre + ".to_v.bias"))
        o = torch.empty_like(q)
        for n in range(N):
            sl = slice(n * S, (n + 1) * S)
            o[sl] = ops.attention_wide(q[sl], k[sl], v[sl])
        return ops.linear(o, W(pre + ".to_out.0.weight"), W(pre + ".to_out.0.bias"), residual=x)

    @staticmethod
    def _to_tokens(x: torch.Tensor) -> torch.Tensor:
        """NCHW fp32 -> channels-last fp16 [N,H,W,64] (channels zero-padded to the k-tile)."""
        n, c, h, w_ = x.shape
        t = x.permute(0, 2, 3, 1).to(H)
        return Fn.pad(t, (0, 64 - c)).contiguous()

    # ------------------------------------------------------------------ public surface
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        from .. import _lib
        _lib.require_gpu(x)
        c, boc = self.config, self.config.block_out_channels
        N, _, Hh, Ww = x.shape
        W = self.w
        hcur = ops.conv3x3(self._to_tokens(x.float()), W("encoder.conv_in.weight"), W("encoder.conv_in.bias")).view(-1, boc[0])
        h, w_, ch = Hh, Ww, boc[0]
        for i, co in enumerate(boc):
            for j in range(c.layers_per_block):
                hcur = self._resnet(f"encoder.down_blocks.{i}.resnets.{j}", hcur, N, h, w_, ch if j == 0 else co, co)
            ch = co
            if i != len(boc) - 1:
                d = f"encoder.down_blocks.{i}.downsamplers.0.conv"
                out = ops.conv3x3(hcur.view(N, h, w_, ch), W(d + ".weight"), W(d + ".bias"), stride=2, pad_lo=0)
                h, w_ = out.shape[1], out.shape[2]
                hcur = out.view(-1, ch)
        hcur = self._resnet("encoder.mid_block.resnets.0", hcur, N, h, w_, ch, ch)
        hcur = self._attention("encoder.mid_block.attentions.0", hcur, N, h * w_, ch)
        hcur = self._resnet("encoder.mid_block.resnets.1", hcur, N, h, w_, ch, ch)
        hcur = ops.groupnorm(hcur, W("encoder.conv_norm_out.weight"), W("encoder.conv_norm_out.bias"), N, 1e-6, True)
        z = ops.conv3x3(hcur.view(N, h, w_, ch), W("encoder.conv_out.weight"), W("encoder.conv_out.bias"))   # [N,h,w,2L]
        # quant_conv: 1x1 on 2L = 8 channels (64 MACs per pixel): host-side layout glue in fp32
        z = z.view(-1, z.shape[-1])[:, : 2 * c.latent_channels].float()
        qw = self.p.t["quant_conv.weight"].reshape(2 * c.latent_channels, 2 * c.latent_channels)
        moments = (z @ qw.t() + self.p.t["quant_conv.bias"]).view(N, h, w_, -1).permute(0, 3, 1, 2).contiguous()
        dist = DiagonalGaussianDistribution(moments)
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, num_frames: int, return_dict: bool = True):
        from .. import _lib
        _lib.require_gpu(z)
        c, boc = self.config, self.config.block_out_channels
        N, _, h, w_ = z.shape
        if N % num_frames:
            raise ValueError("decode: batch is not a multiple of num_frames")
        B, F = N // num_frames, num_frames
        W = self.w
        top = boc[-1]
        x = ops.conv3x3(self._to_tokens(z.float()), W("decoder.conv_in.weight"), W("decoder.conv_in.bias")).view(-1, top)
        x = self._st_resblock("decoder.mid_block.resnets.0", x, B, F, h, w_, top, top)
        if c.layers_per_block > 1:       # `zip(self.resnets[1:], self.attentions)` with ONE attention: only resnets[1] runs
            x = self._attention("decoder.mid_block.attentions.0", x, N, h * w_, top)
            x = self._st_resblock("decoder.mid_block.resnets.1", x, B, F, h, w_, top, top)
        rev = list(reversed(boc))
        prev = rev[0]
        for i, co in enumerate(rev):
            for j in range(c.layers_per_block + 1):
                x = self._st_resblock(f"decoder.up_blocks.{i}.resnets.{j}", x, B, F, h, w_, prev if j == 0 else co, co)
            if i != len(rev) - 1:
                u = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                out = ops.conv3x3(x.view(N, h, w_, co), W(u + ".weight"), W(u + ".bias"), upsample=True)
                h, w_ = out.shape[1], out.shape[2]
                x = out.view(-1, co)
            prev = co
        x = ops.groupnorm(x, W("decoder.conv_norm_out.weight"), W("decoder.conv_norm_out.bias"), N, 1e-6, True)
        y = ops.conv3x3(x.view(N, h, w_, boc[0]), W("decoder.conv_out.weight"), W("decoder.conv_out.bias"))   # [N,H,W,8]
        frames = ops.time_conv_out(y.view(-1, y.shape[-1]), self.time_conv_host[0], self.time_conv_host[1], B, F,
                                   h * w_).view(N, 3, h, w_)
        return SimpleNamespace(sample=frames) if return_dict else (frames,)
