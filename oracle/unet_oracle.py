"""CPU restatement (torch fp32) of the SVD spatio-temporal UNet forward — TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the product
(`syn3r_amd/`) never does.  It follows the reference graph in the reference's own tensor layouts (NCHW frames,
[B*F, HW, C] tokens, [B*HW, F, C] temporal tokens), each function citing the file:line it restates
(paths relative to /root/reference/thirdparty/diffusers/src/diffusers/models/):

  UNetSpatioTemporalConditionModel.forward          unets/unet_spatio_temporal_condition.py:356-489
  Down/Mid/Up blocks                                unets/unet_3d_blocks.py (DownBlockSpatioTemporal, CrossAttn*, …)
  SpatioTemporalResBlock / AlphaBlender             resnet.py:640-805 ; ResnetBlock2D :325-378 ; TemporalResnetBlock :613-636
  TransformerSpatioTemporalModel.forward            transformers/transformer_temporal.py:277-379
  BasicTransformerBlock / TemporalBasicTransformerBlock   attention.py:283-403 / :478-533
  AttnProcessor2_0, GEGLU, FeedForward              attention_processor.py:1222-1299, activations.py, attention.py:608-665

PINNED by `tests/golden/unet_small.npz` (outputs of the reference module itself, run in the build container by
`oracle/gen_golden.py unet`): `tests/test_oracle_golden.py::test_unet_oracle_matches_reference_golden`.
Weights are passed as a diffusers-named state_dict; the configuration is the reference constructor's.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as Fn


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """embeddings.py get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32) / half
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


class UNetOracle:
    def __init__(self, sd: Dict[str, torch.Tensor], config: dict):
        self.sd = {k: v.float() for k, v in sd.items()}
        c = dict(block_out_channels=(320, 640, 1280, 1280), num_attention_heads=(5, 10, 20, 20),
                 addition_time_embed_dim=256, layers_per_block=2, in_channels=8, out_channels=4,
                 down_block_types=("CrossAttnDownBlockSpatioTemporal",) * 3 + ("DownBlockSpatioTemporal",),
                 up_block_types=("UpBlockSpatioTemporal",) + ("CrossAttnUpBlockSpatioTemporal",) * 3)
        c.update(config)
        n = len(c["block_out_channels"])
        if len(c["down_block_types"]) != n:          # the reduced test configuration keeps the default block types
            c["down_block_types"] = ("CrossAttnDownBlockSpatioTemporal",) * (n - 1) + ("DownBlockSpatioTemporal",)
            c["up_block_types"] = ("UpBlockSpatioTemporal",) + ("CrossAttnUpBlockSpatioTemporal",) * (n - 1)
        if isinstance(c["layers_per_block"], int):
            c["layers_per_block"] = (c["layers_per_block"],) * n
        self.c = c

    # ---- leaf helpers
    def lin(self, pre, x, bias=True):
        return Fn.linear(x, self.sd[pre + ".weight"], self.sd[pre + ".bias"] if bias else None)

    def gn(self, pre, x, eps):
        return Fn.group_norm(x, 32, self.sd[pre + ".weight"], self.sd[pre + ".bias"], eps)

    def ln(self, pre, x):
        return Fn.layer_norm(x, (x.shape[-1],), self.sd[pre + ".weight"], self.sd[pre + ".bias"], 1e-5)

    def attn(self, pre, x, ctx, heads):
        """AttnProcessor2_0 (attention_processor.py:1222-1299): no q/k/v bias, scale 1/sqrt(d), to_out with bias."""
        ctx = x if ctx is None else ctx
        q, k, v = self.lin(pre + ".to_q", x, False), self.lin(pre + ".to_k", ctx, False), self.lin(pre + ".to_v", ctx, False)
        B, S, C = q.shape
        d = C // heads
        sp = lambda t: t.view(B, -1, heads, d).transpose(1, 2)
        o = Fn.scaled_dot_product_attention(sp(q), sp(k), sp(v))
        return self.lin(pre + ".to_out.0", o.transpose(1, 2).reshape(B, S, C))

    def ff(self, pre, x):
        """FeedForward with GEGLU (activations.py GEGLU.forward: hidden * gelu(gate), exact erf)."""
        h, g = self.lin(pre + ".net.0.proj", x).chunk(2, dim=-1)
        return self.lin(pre + ".net.2", h * Fn.gelu(g))

    # ---- blocks
    def resnet2d(self, pre, x, temb):
        """ResnetBlock2D (resnet.py:325-378), eps 1e-5, output_scale_factor 1."""
        h = Fn.conv2d(Fn.silu(self.gn(pre + ".norm1", x, 1e-5)), self.sd[pre + ".conv1.weight"], self.sd[pre + ".conv1.bias"], padding=1)
        h = h + self.lin(pre + ".time_emb_proj", Fn.silu(temb))[:, :, None, None]
        h = Fn.conv2d(Fn.silu(self.gn(pre + ".norm2", h, 1e-5)), self.sd[pre + ".conv2.weight"], self.sd[pre + ".conv2.bias"], padding=1)
        if pre + ".conv_shortcut.weight" in self.sd:
            x = Fn.conv2d(x, self.sd[pre + ".conv_shortcut.weight"], self.sd[pre + ".conv_shortcut.bias"])
        return x + h

    def temporal_resnet(self, pre, x, temb):
        """TemporalResnetBlock (resnet.py:613-636) on [B,C,F,H,W]; temb [B,F,T]."""
        h = Fn.conv3d(Fn.silu(self.gn(pre + ".norm1", x, 1e-5)), self.sd[pre + ".conv1.weight"], self.sd[pre + ".conv1.bias"], padding=(1, 0, 0))
        t = self.lin(pre + ".time_emb_proj", Fn.silu(temb)).permute(0, 2, 1)[:, :, :, None, None]
        h = h + t
        h = Fn.conv3d(Fn.silu(self.gn(pre + ".norm2", h, 1e-5)), self.sd[pre + ".conv2.weight"], self.sd[pre + ".conv2.bias"], padding=(1, 0, 0))
        return x + h

    def st_resblock(self, pre, x, temb, B, F):
        """SpatioTemporalResBlock.forward (resnet.py:691-721) + AlphaBlender (image_only_indicator all zero)."""
        x = self.resnet2d(pre + ".spatial_res_block", x, temb)
        BF, C, h, w = x.shape
        x5 = x.view(B, F, C, h, w).permute(0, 2, 1, 3, 4)
        xt = self.temporal_resnet(pre + ".temporal_res_block", x5, temb.view(B, F, -1))
        a = torch.sigmoid(self.sd[pre + ".time_mixer.mix_factor"])
        out = a * x5 + (1.0 - a) * xt
        return out.permute(0, 2, 1, 3, 4).reshape(BF, C, h, w)

    def transformer(self, pre, x, ehs, B, F, heads):
        """TransformerSpatioTemporalModel.forward (transformer_temporal.py:277-379)."""
        BF, C, h, w = x.shape
        # first-frame context, laid out pixel-major / batch-minor (:310-317) — a reference quirk kept as is
        ctx1 = ehs.view(B, F, -1, ehs.shape[-1])[:, 0]
        tctx = ctx1[None, :].broadcast_to(h * w, B, 1, ehs.shape[-1]).reshape(h * w * B, 1, ehs.shape[-1])
        res = x
        hs = self.gn(pre + ".norm", x, 1e-6).permute(0, 2, 3, 1).reshape(BF, h * w, C)
        hs = self.lin(pre + ".proj_in", hs)
        t_emb = timestep_embedding(torch.arange(F).repeat(B), C)
        emb = self.lin(pre + ".time_pos_embed.linear_2", Fn.silu(self.lin(pre + ".time_pos_embed.linear_1", t_emb)))[:, None, :]
        b = pre + ".transformer_blocks.0"                      # BasicTransformerBlock (attention.py:283-403)
        hs = hs + self.attn(b + ".attn1", self.ln(b + ".norm1", hs), None, heads)
        hs = hs + self.attn(b + ".attn2", self.ln(b + ".norm2", hs), ehs, heads)
        hs = hs + self.ff(b + ".ff", self.ln(b + ".norm3", hs))
        t = pre + ".temporal_transformer_blocks.0"             # TemporalBasicTransformerBlock (attention.py:478-533)
        S = h * w
        m = (hs + emb).view(B, F, S, C).permute(0, 2, 1, 3).reshape(B * S, F, C)
        m = m + self.ff(t + ".ff_in", self.ln(t + ".norm_in", m))
        m = m + self.attn(t + ".attn1", self.ln(t + ".norm1", m), None, heads)
        m = m + self.attn(t + ".attn2", self.ln(t + ".norm2", m), tctx, heads)
        m = m + self.ff(t + ".ff", self.ln(t + ".norm3", m))
        m = m.view(B, S, F, C).permute(0, 2, 1, 3).reshape(BF, S, C)
        a = torch.sigmoid(self.sd[pre + ".time_mixer.mix_factor"])
        hs = a * hs + (1.0 - a) * m
        hs = self.lin(pre + ".proj_out", hs)
        return hs.view(BF, h, w, C).permute(0, 3, 1, 2) + res

    # ---- forward
    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids):
        """unet_spatio_temporal_condition.py:356-489; sample [B,F,Cin,h,w] -> [B,F,Cout,h,w] (fp32)."""
        c, sd = self.c, self.sd
        boc = c["block_out_channels"]
        B, F = sample.shape[:2]
        ts = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(B)
        emb = self.lin("time_embedding.linear_2", Fn.silu(self.lin("time_embedding.linear_1", timestep_embedding(ts, boc[0]))))
        te = timestep_embedding(added_time_ids.flatten().float(), c["addition_time_embed_dim"]).reshape(B, -1)
        emb = emb + self.lin("add_embedding.linear_2", Fn.silu(self.lin("add_embedding.linear_1", te)))
        x = sample.float().flatten(0, 1)
        emb = emb.repeat_interleave(F, dim=0)
        ehs = encoder_hidden_states.float().repeat_interleave(F, dim=0)
        x = Fn.conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
        heads = c["num_attention_heads"]
        skips = [x]
        for i, typ in enumerate(c["down_block_types"]):
            for j in range(c["layers_per_block"][i]):
                x = self.st_resblock(f"down_blocks.{i}.resnets.{j}", x, emb, B, F)
                if typ.startswith("CrossAttn"):
                    x = self.transformer(f"down_blocks.{i}.attentions.{j}", x, ehs, B, F, heads[i])
                skips.append(x)
            if i != len(boc) - 1:
                d = f"down_blocks.{i}.downsamplers.0.conv"
                x = Fn.conv2d(x, sd[d + ".weight"], sd[d + ".bias"], stride=2, padding=1)
                skips.append(x)
        x = self.st_resblock("mid_block.resnets.0", x, emb, B, F)
        x = self.transformer("mid_block.attentions.0", x, ehs, B, F, heads[-1])
        x = self.st_resblock("mid_block.resnets.1", x, emb, B, F)
        rev_heads = list(reversed(heads))
        rev_layers = list(reversed(c["layers_per_block"]))
        for i, typ in enumerate(c["up_block_types"]):
            for j in range(rev_layers[i] + 1):
                x = torch.cat([x, skips.pop()], dim=1)
                x = self.st_resblock(f"up_blocks.{i}.resnets.{j}", x, emb, B, F)
                if typ.startswith("CrossAttn"):
                    x = self.transformer(f"up_blocks.{i}.attentions.{j}", x, ehs, B, F, rev_heads[i])
            if i != len(boc) - 1:
                u = f"up_blocks.{i}.upsamplers.0.conv"
                x = Fn.conv2d(Fn.interpolate(x, scale_factor=2.0, mode="nearest"), sd[u + ".weight"], sd[u + ".bias"], padding=1)
        x = Fn.conv2d(Fn.silu(self.gn("conv_norm_out", x, 1e-5)), sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
        return x.view(B, F, *x.shape[1:])
