"""UNetSpatioTemporalConditionModel on the HIP path.

Mirror of the reference's vendored
`thirdparty/diffusers/src/diffusers/models/unets/unet_spatio_temporal_condition.py` (same
constructor arguments, same `forward(sample, timestep, encoder_hidden_states, added_time_ids)`
signature and output shape, same parameter names so a diffusers `state_dict` / safetensors
checkpoint loads unchanged).  The host code only orders kernel launches and owns the buffers;
all tensor work runs in the operators of `ops.py` on channels-last fp16 token matrices.

Restructurings relative to the reference graph (each keeps the reference's result):
  * no permutes / reshapes: the layout [((b*F + f)*h + y)*w + x, C] serves the spatial blocks,
    the temporal blocks and the 3-D resnets alike (unet_3d_blocks.py, resnet.py:703-720,
    attention.py:487-489,527-529);
  * cross-attention has ONE key (encoder_hidden_states is [B*F, 1, 1024],
    unet_spatio_temporal_condition.py:425), so softmax == 1 and attn2 == to_out(to_v(ctx)): a
    per-batch vector added in the epilogue of the preceding projection; its LayerNorm is dead work;
  * time-embedding adds, residual adds and both AlphaBlender mixes are GEMM epilogues;
  * nearest-2x upsampling is folded into the following 3x3 convolution's gather.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple, Union

import torch
import torch.nn.functional as Fn

from .. import _lib as L
from ..tuning import FLAGS as TUNE
from . import ops

H = torch.float16


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """embeddings.py `Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)` -> fp32 [len(t), dim]."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


class _Params:
    """Name -> tensor store with diffusers names; tracks what a module declared."""

    def __init__(self):
        self.shapes: Dict[str, Tuple[int, ...]] = {}
        self.t: Dict[str, torch.Tensor] = {}

    def declare(self, name: str, *shape: int):
        self.shapes[name] = tuple(shape)
        return name

    def linear(self, prefix: str, cin: int, cout: int, bias: bool = True):
        self.declare(prefix + ".weight", cout, cin)
        if bias:
            self.declare(prefix + ".bias", cout)

    def norm(self, prefix: str, c: int):
        self.declare(prefix + ".weight", c)
        self.declare(prefix + ".bias", c)

    def __getitem__(self, name):
        return self.t[name]


class UNetSpatioTemporalConditionModel:
    supports_ctx_group = True          # forward(..., ctx_group=G): see the pipelines' merged passes

    def __init__(self, sample_size: Optional[int] = None, in_channels: int = 8, out_channels: int = 4,
                 down_block_types: Tuple[str, ...] = ("CrossAttnDownBlockSpatioTemporal",) * 3 + ("DownBlockSpatioTemporal",),
                 up_block_types: Tuple[str, ...] = ("UpBlockSpatioTemporal",) + ("CrossAttnUpBlockSpatioTemporal",) * 3,
                 block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280), addition_time_embed_dim: int = 256,
                 projection_class_embeddings_input_dim: int = 768, layers_per_block: Union[int, Tuple[int, ...]] = 2,
                 cross_attention_dim: Union[int, Tuple[int, ...]] = 1024,
                 transformer_layers_per_block: Union[int, Tuple[int, ...]] = 1,
                 num_attention_heads: Union[int, Tuple[int, ...]] = (5, 10, 20, 20), num_frames: int = 25):
        n = len(block_out_channels)
        if len(down_block_types) != len(up_block_types):
            raise ValueError("Must provide the same number of `down_block_types` as `up_block_types`.")
        if len(block_out_channels) != len(down_block_types):
            raise ValueError("Must provide the same number of `block_out_channels` as `down_block_types`.")
        tup = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v,) * n
        self.config = dict(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                           down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
                           block_out_channels=tuple(block_out_channels),
                           addition_time_embed_dim=addition_time_embed_dim,
                           projection_class_embeddings_input_dim=projection_class_embeddings_input_dim,
                           layers_per_block=tup(layers_per_block), cross_attention_dim=tup(cross_attention_dim),
                           transformer_layers_per_block=tup(transformer_layers_per_block),
                           num_attention_heads=tup(num_attention_heads), num_frames=num_frames)
        c = self.config
        if any(v != 1 for v in c["transformer_layers_per_block"]):
            raise NotImplementedError("transformer_layers_per_block != 1 is not used by SVD")
        for ch, hd in zip(block_out_channels, c["num_attention_heads"]):
            if ch != 64 * hd:
                raise NotImplementedError("attention head dim must be 64 (block_out_channels == 64 * num_attention_heads)")
            if ch % 64:
                raise NotImplementedError("channel counts must be multiples of 64")
        self.p = _Params()
        self._declare()
        self.packed: Dict[str, torch.Tensor] = {}
        self.device: Optional[torch.device] = None

    # ------------------------------------------------------------------ parameter declaration (diffusers names)
    def _declare_resblock(self, pre: str, cin: int, cout: int, temb: int):
        p = self.p
        s = pre + ".spatial_res_block"
        p.norm(s + ".norm1", cin)
        p.declare(s + ".conv1.weight", cout, cin, 3, 3); p.declare(s + ".conv1.bias", cout)
        p.linear(s + ".time_emb_proj", temb, cout)
        p.norm(s + ".norm2", cout)
        p.declare(s + ".conv2.weight", cout, cout, 3, 3); p.declare(s + ".conv2.bias", cout)
        if cin != cout:
            p.declare(s + ".conv_shortcut.weight", cout, cin, 1, 1); p.declare(s + ".conv_shortcut.bias", cout)
        t = pre + ".temporal_res_block"
        p.norm(t + ".norm1", cout)
        p.declare(t + ".conv1.weight", cout, cout, 3, 1, 1); p.declare(t + ".conv1.bias", cout)
        p.linear(t + ".time_emb_proj", temb, cout)
        p.norm(t + ".norm2", cout)
        p.declare(t + ".conv2.weight", cout, cout, 3, 1, 1); p.declare(t + ".conv2.bias", cout)
        p.declare(pre + ".time_mixer.mix_factor", 1)

    def _declare_attn(self, pre: str, dim: int, cross: Optional[int]):
        p = self.p
        kdim = cross if cross is not None else dim
        p.linear(pre + ".to_q", dim, dim, bias=False)
        p.linear(pre + ".to_k", kdim, dim, bias=False)
        p.linear(pre + ".to_v", kdim, dim, bias=False)
        p.linear(pre + ".to_out.0", dim, dim)

    def _declare_ff(self, pre: str, dim: int):
        self.p.linear(pre + ".net.0.proj", dim, 8 * dim)
        self.p.linear(pre + ".net.2", 4 * dim, dim)

    def _declare_transformer(self, pre: str, ch: int, cross: int):
        p = self.p
        p.norm(pre + ".norm", ch)
        p.linear(pre + ".proj_in", ch, ch)
        b = pre + ".transformer_blocks.0"
        p.norm(b + ".norm1", ch); self._declare_attn(b + ".attn1", ch, None)
        p.norm(b + ".norm2", ch); self._declare_attn(b + ".attn2", ch, cross)
        p.norm(b + ".norm3", ch); self._declare_ff(b + ".ff", ch)
        t = pre + ".temporal_transformer_blocks.0"
        p.norm(t + ".norm_in", ch); self._declare_ff(t + ".ff_in", ch)
        p.norm(t + ".norm1", ch); self._declare_attn(t + ".attn1", ch, None)
        p.norm(t + ".norm2", ch); self._declare_attn(t + ".attn2", ch, cross)
        p.norm(t + ".norm3", ch); self._declare_ff(t + ".ff", ch)
        p.linear(pre + ".time_pos_embed.linear_1", ch, 4 * ch)
        p.linear(pre + ".time_pos_embed.linear_2", 4 * ch, ch)
        p.declare(pre + ".time_mixer.mix_factor", 1)
        p.linear(pre + ".proj_out", ch, ch)

    def _declare(self):
        c, p = self.config, self.p
        boc = c["block_out_channels"]
        temb = boc[0] * 4
        p.declare("conv_in.weight", boc[0], c["in_channels"], 3, 3); p.declare("conv_in.bias", boc[0])
        p.linear("time_embedding.linear_1", boc[0], temb); p.linear("time_embedding.linear_2", temb, temb)
        p.linear("add_embedding.linear_1", c["projection_class_embeddings_input_dim"], temb)
        p.linear("add_embedding.linear_2", temb, temb)
        self.down_plan, self.up_plan = [], []
        out_ch = boc[0]
        for i, typ in enumerate(c["down_block_types"]):
            in_ch, out_ch = out_ch, boc[i]
            has_attn = typ == "CrossAttnDownBlockSpatioTemporal"
            if not has_attn and typ != "DownBlockSpatioTemporal":
                raise ValueError(f"{typ} does not exist.")
            layers = []
            for j in range(c["layers_per_block"][i]):
                pre = f"down_blocks.{i}"
                self._declare_resblock(f"{pre}.resnets.{j}", in_ch if j == 0 else out_ch, out_ch, temb)
                if has_attn:
                    self._declare_transformer(f"{pre}.attentions.{j}", out_ch, c["cross_attention_dim"][i])
                layers.append((in_ch if j == 0 else out_ch, out_ch))
            down = i != len(boc) - 1
            if down:
                p.declare(f"down_blocks.{i}.downsamplers.0.conv.weight", out_ch, out_ch, 3, 3)
                p.declare(f"down_blocks.{i}.downsamplers.0.conv.bias", out_ch)
            self.down_plan.append(dict(idx=i, attn=has_attn, layers=layers, down=down, ch=out_ch,
                                       heads=c["num_attention_heads"][i]))
        mid = boc[-1]
        self._declare_resblock("mid_block.resnets.0", mid, mid, temb)
        self._declare_transformer("mid_block.attentions.0", mid, c["cross_attention_dim"][-1])
        self._declare_resblock("mid_block.resnets.1", mid, mid, temb)
        rev = list(reversed(boc))
        rev_heads = list(reversed(c["num_attention_heads"]))
        rev_layers = list(reversed(c["layers_per_block"]))
        rev_cross = list(reversed(c["cross_attention_dim"]))
        out_ch = rev[0]
        for i, typ in enumerate(c["up_block_types"]):
            has_attn = typ == "CrossAttnUpBlockSpatioTemporal"
            if not has_attn and typ != "UpBlockSpatioTemporal":
                raise ValueError(f"{typ} does not exist.")
            prev_out, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            nl = rev_layers[i] + 1
            layers = []
            for j in range(nl):
                res_skip = in_ch if j == nl - 1 else out_ch
                res_in = prev_out if j == 0 else out_ch
                self._declare_resblock(f"up_blocks.{i}.resnets.{j}", res_in + res_skip, out_ch, temb)
                if has_attn:
                    self._declare_transformer(f"up_blocks.{i}.attentions.{j}", out_ch, rev_cross[i])
                layers.append((res_in + res_skip, out_ch))
            up = i != len(boc) - 1
            if up:
                p.declare(f"up_blocks.{i}.upsamplers.0.conv.weight", out_ch, out_ch, 3, 3)
                p.declare(f"up_blocks.{i}.upsamplers.0.conv.bias", out_ch)
            self.up_plan.append(dict(idx=i, attn=has_attn, layers=layers, up=up, ch=out_ch, heads=rev_heads[i]))
        p.norm("conv_norm_out", boc[0])
        p.declare("conv_out.weight", c["out_channels"], boc[0], 3, 3); p.declare("conv_out.bias", c["out_channels"])

    # ------------------------------------------------------------------ weights
    def parameter_shapes(self) -> Dict[str, Tuple[int, ...]]:
        return dict(self.p.shapes)

    def num_parameters(self) -> int:
        return sum(math.prod(s) for s in self.p.shapes.values())

    def load_state_dict(self, sd: Dict[str, torch.Tensor], device) -> "UNetSpatioTemporalConditionModel":
        missing = [k for k in self.p.shapes if k not in sd]
        extra = [k for k in sd if k not in self.p.shapes]
        if missing or extra:
            raise KeyError(f"state_dict mismatch: missing {missing[:5]} (+{max(0, len(missing) - 5)}), "
                           f"unexpected {extra[:5]} (+{max(0, len(extra) - 5)})")
        dev = torch.device(device)
        for k, shape in self.p.shapes.items():
            t = sd[k]
            if tuple(t.shape) != shape:
                raise ValueError(f"{k}: shape {tuple(t.shape)} != {shape}")
            self.p.t[k] = t.detach().to(device=dev, dtype=H)
        self.device = dev
        self._pack()
        return self

    def init_random(self, device, seed: int = 0, std: float = 0.02) -> "UNetSpatioTemporalConditionModel":
        """Seeded N(0, std) weights (bench / smoke: no checkpoint is reachable offline)."""
        dev = torch.device(device)
        g = torch.Generator(device=dev).manual_seed(seed)
        sd = {}
        for k, shape in self.p.shapes.items():
            if k.endswith("mix_factor"):
                sd[k] = torch.full(shape, 0.5, device=dev)
            elif ".norm" in k and k.endswith(".weight") or k == "conv_norm_out.weight":
                sd[k] = torch.ones(shape, device=dev)
            elif k.endswith(".bias"):
                sd[k] = torch.zeros(shape, device=dev)
            else:
                fan_in = math.prod(shape[1:])
                sd[k] = torch.randn(shape, generator=g, device=dev, dtype=torch.float32) * min(std, fan_in ** -0.5)
        return self.load_state_dict(sd, dev)

    @classmethod
    def from_pretrained(cls, directory: str, device, variant: Optional[str] = "fp16", **kw):
        """Load `<directory>/config.json` + `diffusion_pytorch_model[.fp16].safetensors` (local files only;
        the reference fetches by model name at model/diffusionGS.py:1089)."""
        import json
        from pathlib import Path
        from safetensors.torch import load_file
        d = Path(directory)
        cfg = json.loads((d / "config.json").read_text())
        keys = ("sample_size in_channels out_channels down_block_types up_block_types block_out_channels "
                "addition_time_embed_dim projection_class_embeddings_input_dim layers_per_block cross_attention_dim "
                "transformer_layers_per_block num_attention_heads num_frames").split()
        model = cls(**{k: cfg[k] for k in keys if k in cfg})
        names = ([f"diffusion_pytorch_model.{variant}.safetensors"] if variant else []) + ["diffusion_pytorch_model.safetensors"]
        for name in names:
            if (d / name).exists():
                return model.load_state_dict(load_file(str(d / name)), device)
        raise FileNotFoundError(f"no safetensors weights under {d} (looked for {names})")

    def _pack(self):
        """Kernel-side layouts: OHWI conv weights, fused QKV, (3,1,1) convs as [Cout,3,Cin]."""
        self._pos_cache = {}        # frame-position embeddings: functions of the weights, F and B only
        self._ctx_store = {}        # folded cross-attention vectors per context (see _context_cache)
        p, pk = self.p, {}
        for k, t in p.t.items():
            if k.endswith(".weight") and t.dim() == 4 and t.shape[-1] == 3:          # Conv2d 3x3
                w = t
                if w.shape[1] % 64:                                                    # conv_in: pad Cin to 64
                    w = Fn.pad(w, (0, 0, 0, 0, 0, 64 - w.shape[1] % 64))
                if w.shape[0] % 8:                                                     # conv_out: pad Cout to 8
                    w = Fn.pad(w, (0, 0, 0, 0, 0, 0, 0, 8 - w.shape[0] % 8))
                pk[k] = w.permute(0, 2, 3, 1).contiguous()
            elif k.endswith(".weight") and t.dim() == 5:                               # Conv3d (3,1,1)
                pk[k] = t[..., 0, 0].permute(0, 2, 1).contiguous()
            elif k.endswith("conv_shortcut.weight"):
                pk[k] = t.reshape(t.shape[0], t.shape[1]).contiguous()
        if p.t["conv_out.bias"].shape[0] % 8:
            pk["conv_out.bias"] = Fn.pad(p.t["conv_out.bias"], (0, 8 - p.t["conv_out.bias"].shape[0] % 8))
        for k in list(p.t):
            if k.endswith("attn1.to_q.weight"):
                pre = k[: -len("to_q.weight")]
                pk[pre + "qkv"] = torch.cat([p.t[pre + "to_q.weight"], p.t[pre + "to_k.weight"],
                                             p.t[pre + "to_v.weight"]], 0).contiguous()
        for k in list(p.t):
            if k.endswith(".net.0.proj.weight"):
                pre = k[: -len("weight")]
                if p.t[k].shape[1] == ops.FUSED_FF_CHANNELS and TUNE["ff_fused"]:
                    # C = 320: the single-kernel feed-forward (per-64-chunk packing); the two-kernel packing is not kept
                    wp, bp, D = ops.pack_geglu_chunked(p.t[k], p.t[pre + "bias"])
                    pk[pre + "geglu_cw"], pk[pre + "geglu_cb"] = wp, bp
                else:
                    wp, bp, D = ops.pack_geglu(p.t[k], p.t[pre + "bias"])
                    pk[pre + "geglu_w"], pk[pre + "geglu_b"] = wp, bp
                    if TUNE["ff_g256"] and D % 128 == 0:
                        # the 256 x 256 tile's packing beside it (shapes without whole tiles - level 3 at F = 14 - keep the 80-column one)
                        pk[pre + "geglu_w64"], pk[pre + "geglu_b64"], _ = ops.pack_geglu64(p.t[k], p.t[pre + "bias"])
        # every resnet's time_emb_proj (44 Linear(1280 -> Cout) on the SAME [B,1280] vector, resnet.py:352,630) as ONE
        # contraction per forward: weights / biases stacked once, each block reads its column slice
        names = [k[: -len(".weight")] for k in p.t if k.endswith("time_emb_proj.weight")]
        self._temb_slices, off = {}, 0
        for nme in names:
            n_out = p.t[nme + ".weight"].shape[0]
            self._temb_slices[nme] = (off, off + n_out)
            off += n_out
        if names:
            pk["time_emb_proj.all.weight"] = torch.cat([p.t[nme + ".weight"] for nme in names], 0).contiguous()
            pk["time_emb_proj.all.bias"] = torch.cat([p.t[nme + ".bias"] for nme in names], 0).contiguous()
        self.packed = pk
        # AlphaBlender scales as host floats, computed ONCE (reading a device scalar per block would drain the
        # launch queue 60 times per forward): alpha in fp16 as `alpha.to(x_spatial.dtype)` (resnet.py:797), and
        # 1 - alpha in fp16 arithmetic, as the reference
        self.alpha = {}
        for k, t in p.t.items():
            if k.endswith("mix_factor"):
                a = torch.sigmoid(t.float()).to(H)
                self.alpha[k] = (float(a), float((1.0 - a).to(H)))

    def _context_cache(self, ehs: torch.Tensor, shared: bool) -> dict:
        """Per-context store of the folded cross-attention vectors.  Keyed on the storage the caller's tensor views
        (address, offset, shape, strides, in-place version); the entry holds the tensor, so the address cannot be handed
        to other data while the entry lives.  A handful of contexts (start / end image, their guidance-tile views).
        INVARIANT: `_version` only sees writes made through torch.  A context buffer refilled by a raw-pointer kernel
        (this library's own operators write through `data_ptr()`) keeps its version: call `invalidate_context_cache()`
        after such a write.  The pipelines build their contexts with torch ops, once per `__call__`."""
        key = (ehs.untyped_storage().data_ptr(), ehs.storage_offset(), tuple(ehs.shape), tuple(ehs.stride()), ehs._version, shared)
        store = self.__dict__.setdefault("_ctx_store", {})
        ent = store.get(key)
        if ent is None:
            if len(store) >= 8:
                store.pop(next(iter(store)))
            ent = store[key] = (ehs, {})
        return ent[1]

    def invalidate_context_cache(self) -> None:
        """Forget every folded cross-attention vector (see `_context_cache`) and every captured graph (their launches
        read those vectors by address: a graph must not outlive the context it was captured with)."""
        self.__dict__.setdefault("_ctx_store", {}).clear()
        self.__dict__.setdefault("_graphs", {}).clear()

    def w(self, name: str) -> torch.Tensor:
        return self.packed.get(name, self.p.t.get(name))

    # ------------------------------------------------------------------ blocks
    def _blend_scales(self, name: str):
        return self.alpha[name]

    def _resblock(self, pre: str, x: torch.Tensor, st: dict, cin: int, cout: int, x2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`x2`: the block input is the channel concatenation [x | x2] (up blocks: hidden state and skip,
        unet_3d_blocks.py); norm1 and the shortcut projection read the two tensors in place."""
        B, F, h, w_ = st["B"], st["F"], st["h"], st["w"]
        HW = h * w_
        s, t = pre + ".spatial_res_block", pre + ".temporal_res_block"
        W = self.w
        # spatial ResnetBlock2D (resnet.py:325-378)
        a0, a1 = self._temb_slices[s + ".time_emb_proj"]
        tp_s = st["temb_all"][:, a0:a1]                                   # [B, cout], a column slice of the stacked projection
        hcur = ops.groupnorm(x, W(s + ".norm1.weight"), W(s + ".norm1.bias"), B * F, 1e-5, True, x2=x2)
        # (every contraction whose output a GroupNorm reads next leaves the statistics' partial sums behind: ops.groupnorm
        # then runs without its pass over the activation, TUNE["gn_epilogue"])
        hcur = ops.conv3x3(hcur.view(B * F, h, w_, cin), W(s + ".conv1.weight"), W(s + ".conv1.bias"),
                           rowvec=tp_s, rows_per_vec=F * HW, gn_stats=TUNE["gn_epilogue"])
        hcur = ops.keep_gn(hcur.view(-1, cout), hcur)
        hcur = ops.groupnorm(hcur, W(s + ".norm2.weight"), W(s + ".norm2.bias"), B * F, 1e-5, True)
        skip = x
        if x2 is not None:                 # (a concatenated input is always wider than the output: the shortcut exists)
            skip = ops.linear_cat(x, x2, W(s + ".conv_shortcut.weight"), W(s + ".conv_shortcut.bias"))
        elif cin != cout:
            skip = ops.linear(x, W(s + ".conv_shortcut.weight"), W(s + ".conv_shortcut.bias"))
        xs = ops.conv3x3(hcur.view(B * F, h, w_, cout), W(s + ".conv2.weight"), W(s + ".conv2.bias"),
                         residual=skip, gn_stats=TUNE["gn_epilogue"])
        xs = ops.keep_gn(xs.view(-1, cout), xs)
        # TemporalResnetBlock (resnet.py:613-636) + AlphaBlender (:789-802)
        a0, a1 = self._temb_slices[t + ".time_emb_proj"]
        tp_t = st["temb_all"][:, a0:a1]
        hcur = ops.groupnorm(xs, W(t + ".norm1.weight"), W(t + ".norm1.bias"), B, 1e-5, True)
        hcur = ops.tconv3(hcur, W(t + ".conv1.weight"), W(t + ".conv1.bias"), B, F, HW, rowvec=tp_t, rows_per_vec=F * HW, gn_stats=TUNE["gn_epilogue"])
        hcur = ops.groupnorm(hcur, W(t + ".norm2.weight"), W(t + ".norm2.bias"), B, 1e-5, True)
        a, om = self._blend_scales(pre + ".time_mixer.mix_factor")
        # alpha*xs + (1-alpha)*(xs + conv2)  ==  (alpha + (1-alpha))*xs + (1-alpha)*conv2
        return ops.tconv3(hcur, W(t + ".conv2.weight"), W(t + ".conv2.bias"), B, F, HW, residual=xs, s_acc=om,
                          s_res=a + om, gn_stats=TUNE["gn_epilogue"])

    def _cross_vec(self, pre: str, ehs: torch.Tensor, cache: Optional[dict] = None) -> torch.Tensor:
        """attn2 with a single key: to_out(to_v(ctx)) per batch item -> [B, C].  A function of the context alone: the
        32 vectors of a forward are kept per context (`forward` keys them on the caller's encoder_hidden_states), so the
        200 UNet calls of a denoising run compute them once instead of reading ~100 MB of to_v / to_out weights each time."""
        if cache is not None and pre in cache:
            return cache[pre]
        v = ops.linear(ehs, self.w(pre + ".to_v.weight"))
        out = ops.linear(v, self.w(pre + ".to_out.0.weight"), self.w(pre + ".to_out.0.bias"))
        if cache is not None:
            cache[pre] = out
        return out

    def _ff(self, pre: str, x: torch.Tensor, norm: Optional[str] = None, addvec: Optional[tuple] = None, **epilogue) -> torch.Tensor:
        """FeedForward of block `pre` on x; `norm` = the LayerNorm (parameter prefix) that precedes it in the reference
        (attention.py:376-392, 519-530): applied inside the fused kernel at C = 320, as its own launch otherwise.
        addvec = (vec, rows_per_vec): x + vec[row // rows_per_vec] is what `norm` normalises and what is added back as the
        residual (norm_in / ff_in, attention.py:500-517): also inside the fused kernel at C = 320."""
        D = self.p.shapes[pre + ".net.0.proj.weight"][0] // 2
        cw = self.packed.get(pre + ".net.0.proj.geglu_cw")
        fuse = TUNE["ff_ln"]
        if addvec is not None and not (cw is not None and fuse):         # the sum as a tensor, then the plain path
            x, hmix = ops.layernorm(x, self.w(norm + ".weight"), self.w(norm + ".bias"), addvec=addvec[0], rows_per_vec=addvec[1],
                                    want_sum=True)
            norm, addvec, epilogue = None, None, dict(epilogue, residual=hmix)
        if cw is not None:               # C = 320: one kernel, neither the normalised nor the hidden activation in HBM
            ln = (self.w(norm + ".weight"), self.w(norm + ".bias"), 1e-5) if norm and fuse else None
            if norm and ln is None:
                x = ops.layernorm(x, self.w(norm + ".weight"), self.w(norm + ".bias"))
            return ops.feedforward_fused(x, cw, self.packed[pre + ".net.0.proj.geglu_cb"], D, self.w(pre + ".net.2.weight"),
                                         self.w(pre + ".net.2.bias"), ln=ln, addvec=addvec, **epilogue)
        if norm:
            x = ops.layernorm(x, self.w(norm + ".weight"), self.w(norm + ".bias"))
        wp = self.w(pre + ".net.0.proj.geglu_w")
        if not TUNE["ff_tiled"]:      # tuning: row-major intermediate, two separate calls
            return ops.linear(ops.linear_geglu(x, wp, self.w(pre + ".net.0.proj.geglu_b"), D),
                              self.w(pre + ".net.2.weight"), self.w(pre + ".net.2.bias"), **epilogue)
        w64 = self.packed.get(pre + ".net.0.proj.geglu_w64") if TUNE["ff_g256"] else None
        p64 = (w64, self.packed[pre + ".net.0.proj.geglu_b64"]) if w64 is not None else None
        return ops.feedforward(x, wp, self.w(pre + ".net.0.proj.geglu_b"), D, self.w(pre + ".net.2.weight"),
                               self.w(pre + ".net.2.bias"), packed64=p64, **epilogue)

    def _norm_qkv(self, blk: str, x: torch.Tensor) -> torch.Tensor:
        """attn1's stacked q / k / v projection of norm1(x) (attention.py:340-352, 509-512): one kernel at C = 320
        (tuning.FLAGS["ln_qkv"] = False: the two launches, tuning)."""
        W = self.w
        if not TUNE["ln_qkv"]:
            return ops.linear(ops.layernorm(x, W(blk + ".norm1.weight"), W(blk + ".norm1.bias")), W(blk + ".attn1.qkv"))
        return ops.layernorm_linear(x, W(blk + ".norm1.weight"), W(blk + ".norm1.bias"), W(blk + ".attn1.qkv"))

    def _transformer(self, pre: str, x: torch.Tensor, st: dict, ch: int, heads: int) -> torch.Tensor:
        B, F, h, w_ = st["B"], st["F"], st["h"], st["w"]
        HW = h * w_
        W = self.w
        ehs = st["ehs"]
        hs = ops.groupnorm(x, W(pre + ".norm.weight"), W(pre + ".norm.bias"), B * F, 1e-6, False)
        hs = ops.linear(hs, W(pre + ".proj_in.weight"), W(pre + ".proj_in.bias"))
        # frame-position embedding (transformer_temporal.py:326-337), one row per (b, f)
        # (its input is the frame index alone, so the MLP output is a constant of the loaded weights: kept per
        # (block, F, B) instead of being recomputed by every forward as the reference does)
        ck = (pre, F, B, str(x.device))
        emb = self._pos_cache.get(ck)
        if emb is None:
            pos = timestep_embedding(torch.arange(F, device=x.device), ch).to(H)
            e = ops.linear(pos, W(pre + ".time_pos_embed.linear_1.weight"), W(pre + ".time_pos_embed.linear_1.bias"))
            e = ops.linear(Fn.silu(e), W(pre + ".time_pos_embed.linear_2.weight"), W(pre + ".time_pos_embed.linear_2.bias"))
            emb = self._pos_cache[ck] = e.repeat(B, 1).contiguous()        # [B*F, C]
        # BasicTransformerBlock (attention.py:283-403)
        b = pre + ".transformer_blocks.0"
        a1 = ops.attention(self._norm_qkv(b, hs), B * F, HW, heads)
        hs = ops.linear(a1, W(b + ".attn1.to_out.0.weight"), W(b + ".attn1.to_out.0.bias"), residual=hs,
                        rowvec=self._cross_vec(b + ".attn2", ehs, st["ctx_cache"]), rows_per_vec=(B if st["shared_ctx"] else 1) * F * HW)
        hs = self._ff(b + ".ff", hs, norm=b + ".norm3", residual=hs)
        # TemporalBasicTransformerBlock (attention.py:478-533) on hs + emb
        t = pre + ".temporal_transformer_blocks.0"
        tt = self._ff(t + ".ff_in", hs, norm=t + ".norm_in", addvec=(emb, HW))       # ff_in(norm_in(hs + emb)) + (hs + emb)
        a1 = ops.attention_temporal(self._norm_qkv(t, tt), B, F, HW, heads)
        # The reference lays the first-frame context out pixel-major / batch-minor
        # (transformer_temporal.py:310-317) while the temporal tokens are batch-major (attention.py:487-489):
        # token (b, pixel) reads the context of batch item (b*HW + pixel) mod B.  Reproduced, not fixed.
        # With ONE context shared by every batch item (B = 1, or a stride-0 expanded context) the interleave is moot.
        # `ctx_group` = G (forward(..., ctx_group=G)): the batch is a stack of B / G separate batch-of-G calls of the
        # reference (the forward- and backward-in-time passes of one denoising step), so the interleave runs inside each
        # group of G samples; G = 1 is the reference's B = 1 call per sample (every token reads its own context).
        G = st["ctx_group"]
        if st["shared_ctx"]:
            rpv, grp = B * F * HW, 0
        elif G == 1:
            rpv, grp = F * HW, 0
        else:
            if HW % G:
                raise NotImplementedError("temporal cross-attention context interleave needs h*w divisible by the batch "
                                          "size (or one context shared by the batch: pass it expanded, stride 0)")
            rpv, grp = -G, (G * F * HW if G != B else 0)
        tt = ops.linear(a1, W(t + ".attn1.to_out.0.weight"), W(t + ".attn1.to_out.0.bias"), residual=tt,
                        rowvec=self._cross_vec(t + ".attn2", ehs, st["ctx_cache"]), rows_per_vec=rpv, rv_group_rows=grp)
        a, om = self._blend_scales(pre + ".time_mixer.mix_factor")
        # alpha*hs + (1-alpha)*(ff(norm3(tt)) + tt)
        mix = self._ff(t + ".ff", tt, norm=t + ".norm3", residual=tt, aux=hs, s_acc=om, s_res=om, s_aux=a)
        return ops.linear(mix, W(pre + ".proj_out.weight"), W(pre + ".proj_out.bias"), residual=x, gn_stats=TUNE["gn_epilogue"])

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor,
                added_time_ids: torch.Tensor, return_dict: bool = False, ctx_group: Optional[int] = None):
        """Reference: unet_spatio_temporal_condition.py:356-489.  sample [B,F,Cin,h,w] fp16 -> [B,F,Cout,h,w].

        `ctx_group` (extension): the batch stacks B / ctx_group INDEPENDENT calls of the reference, each of batch size
        ctx_group, sharing timestep and time ids (the two passes of a denoising step: SVD_2pass_prob_uncertain.py:661-742).
        Every per-sample operator is unaffected by the stacking; the one batch-coupled spot — the reference's
        batch-interleaved temporal cross-attention context — is applied per group.  Default: one call (ctx_group = B)."""
        if self.device is None:
            raise L.Syn3rError("UNet weights are not loaded (load_state_dict / from_pretrained / init_random)")
        dev = L.require_gpu(sample, encoder_hidden_states, added_time_ids)
        c = self.config
        B, F, Cin, h, w_ = sample.shape
        if Cin != c["in_channels"]:
            raise ValueError(f"sample has {Cin} channels, the model expects {c['in_channels']}")
        if F > 32:
            raise NotImplementedError("temporal attention kernel supports up to 32 frames")
        W = self.w
        boc = c["block_out_channels"]
        # split-K scratch for the lowest level's convolutions when their tile grid would leave most CUs idle (F = 14: 4 032 rows =
        # 128 tiles on 256 CUs; include/syn3r_hip.h syn3r_gemm_set_splitk_workspace): set for this forward, on this thread
        down = 2 ** (len(boc) - 1)
        m_low = B * F * (h // down) * (w_ // down)
        splitk = None
        if TUNE["splitk"] and ((m_low + 255) // 256) * ((boc[-1] + 159) // 160) * 2 <= 256:
            nbytes = 4 * m_low * boc[-1] * 4                 # up to four fp32 partial tiles of the lowest level's widest output
            splitk = L.workspace(dev, nbytes, "splitk")
            # (the REQUESTED size, not the cached buffer's: which launches split must not depend on what ran before)
            L.check(L.load().syn3r_gemm_set_splitk_workspace(L.ptr(splitk), nbytes), "syn3r_gemm_set_splitk_workspace")
        try:
            return self._forward(sample, timestep, encoder_hidden_states, added_time_ids, return_dict, ctx_group)
        finally:
            if splitk is not None:
                L.load().syn3r_gemm_set_splitk_workspace(None, 0)

    def _forward(self, sample, timestep, encoder_hidden_states, added_time_ids, return_dict, ctx_group):
        dev = L.require_gpu(sample, encoder_hidden_states, added_time_ids)
        c = self.config
        B, F, Cin, h, w_ = sample.shape
        W = self.w
        boc = c["block_out_channels"]
        # 1. time (:385-418)
        ts = timestep if torch.is_tensor(timestep) else torch.tensor([timestep], dtype=torch.float64)
        ts = ts.reshape(-1).to(dev).expand(B)
        t_emb = timestep_embedding(ts, boc[0]).to(H)
        emb = ops.linear(t_emb, W("time_embedding.linear_1.weight"), W("time_embedding.linear_1.bias"))
        emb = ops.linear(Fn.silu(emb), W("time_embedding.linear_2.weight"), W("time_embedding.linear_2.bias"))
        te = timestep_embedding(added_time_ids.flatten().to(dev), c["addition_time_embed_dim"]).reshape(B, -1).to(H)
        aug = ops.linear(te, W("add_embedding.linear_1.weight"), W("add_embedding.linear_1.bias"))
        aug = ops.linear(Fn.silu(aug), W("add_embedding.linear_2.weight"), W("add_embedding.linear_2.bias"))
        emb = emb + aug
        # one context for the whole batch (a stride-0 `expand` of a single embedding, as the guidance tiles pass it):
        # the folded cross-attention vector is computed once and added to every row
        shared_ctx = B == 1 or encoder_hidden_states.stride(0) == 0
        ctx = encoder_hidden_states[:1] if shared_ctx else encoder_hidden_states
        G = B if ctx_group is None else int(ctx_group)
        if G < 1 or B % G:
            raise ValueError(f"ctx_group={ctx_group} must divide the batch size {B}")
        st = dict(B=B, F=F, h=h, w=w_, temb_act=Fn.silu(emb).contiguous(), shared_ctx=shared_ctx, ctx_group=G,
                  ehs=ctx.reshape(ctx.shape[0], -1).to(H).contiguous(), ctx_cache=self._context_cache(encoder_hidden_states, shared_ctx))
        st["temb_all"] = ops.linear(st["temb_act"], W("time_emb_proj.all.weight"), W("time_emb_proj.all.bias"))
        # 2. conv_in on NHWC with channels padded to 64 (:428)
        x = sample.to(H).flatten(0, 1).permute(0, 2, 3, 1)
        x = Fn.pad(x, (0, 64 - Cin % 64 if Cin % 64 else 0)).contiguous()
        x = ops.conv3x3(x, W("conv_in.weight"), W("conv_in.bias"), gn_stats=TUNE["gn_epilogue"])
        x = ops.keep_gn(x.view(-1, boc[0]), x)
        skips = [(x, boc[0])]
        # 3. down (:432-449)
        for blk in self.down_plan:
            i = blk["idx"]
            for j, (cin, cout) in enumerate(blk["layers"]):
                x = self._resblock(f"down_blocks.{i}.resnets.{j}", x, st, cin, cout)
                if blk["attn"]:
                    x = self._transformer(f"down_blocks.{i}.attentions.{j}", x, st, cout, blk["heads"])
                skips.append((x, cout))
            if blk["down"]:
                ch = blk["ch"]
                x = ops.conv3x3(x.view(B * F, st["h"], st["w"], ch), W(f"down_blocks.{i}.downsamplers.0.conv.weight"),
                                W(f"down_blocks.{i}.downsamplers.0.conv.bias"), stride=2, gn_stats=TUNE["gn_epilogue"])
                st["h"], st["w"] = x.shape[1], x.shape[2]
                x = ops.keep_gn(x.view(-1, ch), x)
                skips.append((x, ch))
        # 4. mid (:452-457)
        mid = boc[-1]
        x = self._resblock("mid_block.resnets.0", x, st, mid, mid)
        x = self._transformer("mid_block.attentions.0", x, st, mid, c["num_attention_heads"][-1])
        x = self._resblock("mid_block.resnets.1", x, st, mid, mid)
        # 5. up (:460-478)
        for blk in self.up_plan:
            i = blk["idx"]
            for j, (cin, cout) in enumerate(blk["layers"]):
                sk, sk_ch = skips.pop()
                if TUNE["unet_cat"]:      # tuning: materialise the concatenation
                    x = self._resblock(f"up_blocks.{i}.resnets.{j}", torch.cat([x, sk], dim=1), st, cin, cout)
                else:                                            # the skip is read in place (norm1 + shortcut take two sources)
                    x = self._resblock(f"up_blocks.{i}.resnets.{j}", x, st, cin, cout, x2=sk)
                if blk["attn"]:
                    x = self._transformer(f"up_blocks.{i}.attentions.{j}", x, st, cout, blk["heads"])
            if blk["up"]:
                ch = blk["ch"]
                x = ops.conv3x3(x.view(B * F, st["h"], st["w"], ch), W(f"up_blocks.{i}.upsamplers.0.conv.weight"),
                                W(f"up_blocks.{i}.upsamplers.0.conv.bias"), upsample=True, gn_stats=TUNE["gn_epilogue"])
                st["h"], st["w"] = x.shape[1], x.shape[2]
                x = ops.keep_gn(x.view(-1, ch), x)
        # 6. out (:481-486)
        x = ops.groupnorm(x, W("conv_norm_out.weight"), W("conv_norm_out.bias"), B * F, 1e-5, True)
        y = ops.conv3x3(x.view(B * F, st["h"], st["w"], boc[0]), W("conv_out.weight"), W("conv_out.bias"))
        y = y[..., : c["out_channels"]].permute(0, 3, 1, 2).reshape(B, F, c["out_channels"], st["h"], st["w"]).contiguous()
        if return_dict:
            from types import SimpleNamespace
            return SimpleNamespace(sample=y)
        return (y,)

    # ------------------------------------------------------------------ captured forward (hipGraph)
    # A forward is ~350 launches per batch-of-2 call at F = 14, each a Python -> ctypes call of ~5 us: with kernels as short as
    # the level-3 ones the launch queue runs dry (3 % of the unit at F = 14, 0.5 % at F = 25: bench.py `denoise_host_gap`).
    # `forward_graphed` captures the launch sequence of one (shape, context) once - torch.cuda.CUDAGraph, i.e. a hipGraph on
    # the capture stream the operators already launch on - and replays it: same kernels, same order, same buffers (the
    # graph's private pool), bit-identical results.  Inputs are copied into static buffers; the timestep is a device scalar.
    def forward_graphed(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor, added_time_ids: torch.Tensor,
                        ctx_group: Optional[int] = None):
        if L._active_trace is not None:                 # per-kernel timing brackets every launch with events: run eagerly
            return self.forward(sample, timestep, encoder_hidden_states, added_time_ids, ctx_group=ctx_group)
        dev = L.require_gpu(sample, encoder_hidden_states, added_time_ids)
        ehs = encoder_hidden_states
        # the timestep exactly as `forward` builds it (a Python float is a float64 scalar there): same embedding, bit for bit
        ts = timestep if torch.is_tensor(timestep) else torch.tensor([timestep], dtype=torch.float64)
        ts = ts.reshape(-1)[:1].to(dev)
        key = (tuple(sample.shape), ctx_group, ehs.untyped_storage().data_ptr(), ehs.storage_offset(), tuple(ehs.shape),
               tuple(ehs.stride()), ehs._version, tuple(added_time_ids.shape), ts.dtype)
        graphs = self.__dict__.setdefault("_graphs", {})
        ent = graphs.get(key)
        if ent is None:
            if len(graphs) >= 8:
                graphs.pop(next(iter(graphs)))
            s_in, t_in, a_in = sample.to(H).clone(), ts.clone(), added_time_ids.clone()
            # warm-up on the capture stream's side: workspaces, folded contexts, position embeddings, kernel attributes
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.forward(s_in, t_in, ehs, a_in, ctx_group=ctx_group)
            torch.cuda.current_stream(dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                y = self.forward(s_in, t_in, ehs, a_in, ctx_group=ctx_group)[0]
            # Everything the captured launches address that lives OUTSIDE the graph's private pool is pinned by the entry:
            # the folded cross-attention vectors (allocated by the warm-up; `_ctx_store` evicts FIFO and is cleared by
            # invalidate_context_cache) and the scratch buffers `L.workspace` handed out on the capture stream (replaced
            # when a larger request arrives on a recycled stream handle).  The stream object stays too, so its handle
            # is not given to another stream while this graph lives.
            shared_ctx = sample.shape[0] == 1 or ehs.stride(0) == 0          # as `forward` keys its context cache
            pinned = (self._context_cache(ehs, shared_ctx),
                      [buf for (di, st_, _tag), buf in L._ws_cache.items() if di == dev.index and st_ == side.cuda_stream], side)
            ent = graphs[key] = (g, s_in, t_in, a_in, y, ehs, pinned)
        g, s_in, t_in, a_in, y = ent[:5]
        s_in.copy_(sample)
        t_in.copy_(ts)
        a_in.copy_(added_time_ids)
        g.replay()
        return (y.clone(),)

    __call__ = forward
