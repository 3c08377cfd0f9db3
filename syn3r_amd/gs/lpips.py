"""LPIPS (VGG16) perceptual loss on the HIP path — the term the reference switches on for every refine
(`gsTrainer.opt.use_lpips_loss = True`, model/diffusionGS.py:1690,1697; `--lpips_weight 1` in
bash_scripts/batch_dl3dv_train.sh:84-87).

The loss itself lives in FSGS (un-vendored) and calls the `lpips` package (not in /root/reference): what is implemented is
the PUBLISHED definition — `lpips.LPIPS(net='vgg')`, version 0.1 (Zhang et al. 2018): VGG16 features after relu1_2 / 2_2 /
3_3 / 4_3 / 5_3, unit-normalised over channels, squared difference, learned 1x1 weights, spatial mean, summed — with the
parameter names of that package, so its state_dict (torchvision VGG16 `net.sliceN.K.weight/bias` + `lin{k}.model.1.weight`)
loads unchanged.  Weights are the caller's: no checkpoint is reachable offline (`init_random` for tests / timing).
PARITY UNPINNED (oracle: oracle/lpips_oracle.py, torch fp32 with autograd).

Forward and backward run on the HIP operators: the 13 convolutions (and their backward-data form: the same kernel on
transposed, flipped weights, with the ReLU mask of the layer below fused into its epilogue) on the implicit-GEMM kernel,
pooling / normalised differences / image scaling in csrc/lpips.hip.  fp16 storage, fp32 accumulation; the backward carries
a loss scale (the pixel gradients are ~1e-9, below fp16's range) that the last kernel divides out.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

from .. import _lib as L

H16 = torch.float16

# torchvision VGG16 `features` indices of the 13 convolutions, grouped in the five LPIPS slices; a max-pool precedes slices 2-5
_SLICES: Tuple[Tuple[Tuple[int, int, int], ...], ...] = (
    ((0, 3, 64), (2, 64, 64)),
    ((5, 64, 128), (7, 128, 128)),
    ((10, 128, 256), (12, 256, 256), (14, 256, 256)),
    ((17, 256, 512), (19, 512, 512), (21, 512, 512)),
    ((24, 512, 512), (26, 512, 512), (28, 512, 512)),
)
_CHNS = (64, 128, 256, 512, 512)


def _conv(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], Hh: int, Ww: int, relu: bool, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [H*W, Cin] fp16, w [Cout,3,3,Cin] -> [H*W, Cout] (syn3r_conv2d3x3_act_f16)."""
    dev = x.device
    Cin, Cout = x.shape[1], w.shape[0]
    out = torch.empty((Hh * Ww, Cout), dtype=H16, device=dev)
    rc = L.load().syn3r_conv2d3x3_act_f16(L.ptr(x), L.ptr(w), L.ptr(out), L.ptr(b), 1 if relu else 0, L.ptr(mask), 1, Hh, Ww, Cin, Cout,
                                          L.stream_ptr(dev))
    L.check(rc, "syn3r_conv2d3x3_act_f16")
    return out


class LPIPS:
    """`lpips.LPIPS(net='vgg', version='0.1')`: `loss = model(pred, target)` for [3,H,W] images in [0,1] (the package's
    `normalize=True` input convention); differentiable wrt `pred`."""

    LOSS_SCALE_PER_PIXEL = 1024.0       # backward loss scale = this x H x W (keeps the fp16 gradients of every layer in range)

    def __init__(self):
        self.shapes: Dict[str, Tuple[int, ...]] = {}
        for s, convs in enumerate(_SLICES):
            for idx, cin, cout in convs:
                self.shapes[f"net.slice{s + 1}.{idx}.weight"] = (cout, cin, 3, 3)
                self.shapes[f"net.slice{s + 1}.{idx}.bias"] = (cout,)
        for k, c in enumerate(_CHNS):
            self.shapes[f"lin{k}.model.1.weight"] = (1, c, 1, 1)
        self.device: Optional[torch.device] = None
        self._target_cache: Dict[int, tuple] = {}

    def parameter_shapes(self) -> Dict[str, Tuple[int, ...]]:
        return dict(self.shapes)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], device) -> "LPIPS":
        sd = {k: v for k, v in sd.items() if not k.startswith("scaling_layer") and ".model.0." not in k}
        missing = [k for k in self.shapes if k not in sd]
        extra = [k for k in sd if k not in self.shapes]
        if missing or extra:
            raise KeyError(f"LPIPS state_dict mismatch: missing {missing[:4]}, unexpected {extra[:4]}")
        dev = torch.device(device)
        if dev.type != "cuda":
            raise L.Syn3rError("LPIPS runs on the HIP path only (no CPU fallback)")
        self.fwd, self.bwd = [], []
        for convs in _SLICES:
            fw, bw = [], []
            for idx, cin, cout in convs:
                s = [k for k in self.shapes if k.endswith(f".{idx}.weight") and k.startswith("net.")][0]
                w = sd[s].detach().to(dev, torch.float32)
                b = sd[s[:-6] + "bias"].detach().to(dev, H16).contiguous()
                if tuple(w.shape) != self.shapes[s]:
                    raise ValueError(f"{s}: shape {tuple(w.shape)} != {self.shapes[s]}")
                cin_p = max(cin, 64)                                          # conv1_1: 3 input channels padded to 64
                wf = torch.zeros((cout, 3, 3, cin_p), dtype=torch.float32, device=dev)
                wf[..., :cin] = w.permute(0, 2, 3, 1)                         # OHWI
                # backward-data = the forward kernel on W'[ci][ky][kx][co] = W[co][2-ky][2-kx][ci]
                wb = torch.zeros((cin_p, 3, 3, cout), dtype=torch.float32, device=dev)
                wb[:cin] = w.flip(2, 3).permute(1, 2, 3, 0)
                fw.append((wf.to(H16).contiguous(), b))
                bw.append(wb.to(H16).contiguous())
            self.fwd.append(fw)
            self.bwd.append(bw)
        self.lin = [sd[f"lin{k}.model.1.weight"].detach().to(dev, torch.float32).reshape(-1).contiguous() for k in range(5)]
        self.device = dev
        self._target_cache.clear()
        return self

    def init_random(self, device, seed: int = 0) -> "LPIPS":
        """Seeded He-initialised VGG weights and positive 1x1 weights (tests / timing)."""
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for k, shape in self.shapes.items():
            if k.startswith("lin"):
                sd[k] = torch.rand(shape, generator=g) * 0.2 + 0.01
            elif k.endswith(".bias"):
                sd[k] = 0.05 * torch.randn(shape, generator=g)
            else:
                sd[k] = torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[1] * 9))
        return self.load_state_dict(sd, device)

    # ------------------------------------------------------------------ forward pieces
    def _features(self, img: torch.Tensor, keep: bool):
        """img [3,H,W] fp32 in [0,1] on the device -> the five feature maps [(H_k*W_k, C_k)] (+ every post-ReLU activation and
        the pooled inputs when `keep`, for the backward)."""
        dev = img.device
        lib = L.load()
        _, Hh, Ww = img.shape
        x = torch.empty((Hh * Ww, 64), dtype=H16, device=dev)
        L.check(lib.syn3r_lpips_image_f16(L.ptr(img), Hh, Ww, L.ptr(x), L.stream_ptr(dev)), "syn3r_lpips_image_f16")
        feats, acts, dims = [], [], []
        h, w_ = Hh, Ww
        for s, convs in enumerate(self.fwd):
            if s > 0:
                if h < 2 or w_ < 2:
                    raise ValueError("LPIPS: image too small for the five VGG stages (needs >= 16 pixels per side)")
                y = torch.empty(((h // 2) * (w_ // 2), x.shape[1]), dtype=H16, device=dev)
                L.check(lib.syn3r_maxpool2_f16(L.ptr(x), h, w_, x.shape[1], L.ptr(y), L.stream_ptr(dev)), "syn3r_maxpool2_f16")
                h, w_ = h // 2, w_ // 2
                x = y
            layer_acts = [x]                       # input of the slice (the image tensor or the pooled map)
            for wf, b in convs:
                x = _conv(x, wf, b, h, w_, relu=True)
                layer_acts.append(x)
            feats.append(x)
            dims.append((h, w_))
            if keep:
                acts.append(layer_acts)
        return feats, acts, dims

    def _target_features(self, target: torch.Tensor):
        """Feature maps of a ground-truth image, kept per tensor object (training compares many renders with few targets)."""
        key = id(target)
        hit = self._target_cache.get(key)
        if hit is not None and hit[0] is target and hit[1] == target._version:
            return hit[2]
        feats, _, _ = self._features(target.detach().to(self.device, torch.float32).contiguous(), keep=False)
        if len(self._target_cache) >= 16:
            self._target_cache.pop(next(iter(self._target_cache)))
        self._target_cache[key] = (target, target._version, feats)
        return feats

    def __call__(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if self.device is None:
            raise L.Syn3rError("LPIPS weights are not loaded (load_state_dict / init_random)")
        L.require_gpu(pred)
        if pred.dim() != 3 or pred.shape[0] != 3 or tuple(target.shape) != tuple(pred.shape):
            raise ValueError(f"LPIPS: images must both be [3,H,W], got {tuple(pred.shape)} / {tuple(target.shape)}")
        return _LpipsFn.apply(pred, target, self)


class _LpipsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, model: LPIPS):
        dev = pred.device
        lib = L.load()
        x = pred.detach().to(torch.float32).contiguous()
        feats, acts, dims = model._features(x, keep=True)
        tfeats = model._target_features(target)
        value = torch.zeros(1, dtype=torch.float32, device=dev)
        for k in range(5):
            P, C = feats[k].shape
            ws = L.workspace(dev, lib.syn3r_lpips_layer_workspace_bytes(P, C), "lpips")
            rc = lib.syn3r_lpips_layer_f16(L.ptr(feats[k]), L.ptr(tfeats[k]), L.ptr(model.lin[k]), P, C, 1 if k else 0, L.ptr(value),
                                           L.ptr(ws), ws.numel(), L.stream_ptr(dev))
            L.check(rc, "syn3r_lpips_layer_f16")
        ctx.model, ctx.acts, ctx.dims, ctx.tfeats = model, acts, dims, tfeats
        ctx.shape = tuple(pred.shape)
        return value[0]

    @staticmethod
    def backward(ctx, grad_out):
        L.join_active_trace()
        model, acts, dims, tfeats = ctx.model, ctx.acts, ctx.dims, ctx.tfeats
        lib = L.load()
        _, Hh, Ww = ctx.shape
        dev = acts[0][0].device
        scale = model.LOSS_SCALE_PER_PIXEL * Hh * Ww
        up = 1.0                                       # the upstream scalar multiplies the result at the end, on the device (no host read)
        g = None                                       # gradient wrt the current slice's OUTPUT feature map (post-ReLU), x scale
        for k in range(4, -1, -1):
            feat = acts[k][-1]
            P, C = feat.shape
            if g is None:
                g = torch.empty_like(feat)
                acc = 0
            else:
                acc = 1
            rc = lib.syn3r_lpips_layer_bwd_f16(L.ptr(feat), L.ptr(tfeats[k]), L.ptr(model.lin[k]), P, C, up * scale, acc, L.ptr(g),
                                               L.stream_ptr(dev))
            L.check(rc, "syn3r_lpips_layer_bwd_f16")
            h, w_ = dims[k]
            # back through the slice's convolutions: g is d/d(post-ReLU output of conv n) -> mask by (output > 0) -> backward-data
            # convolution -> d/d(input of conv n) = d/d(post-ReLU output of conv n-1), whose own mask is applied in the epilogue
            nconv = len(model.bwd[k])
            # (the ReLU of the slice's LAST convolution has been applied by the layer kernel: its output is the feature map)
            for n in range(nconv - 1, -1, -1):
                below = acts[k][n]                     # input of conv n (post-ReLU of conv n-1, or the slice input)
                mask = below if n > 0 else None        # the slice input is a pooled map / the image tensor: no ReLU of its own
                g = _conv(g, model.bwd[k][n], None, h, w_, relu=False, mask=mask)
            if k > 0:                                  # through the max-pool into the previous slice's output
                ph, pw = dims[k - 1]
                prev = acts[k - 1][-1]
                gp = torch.empty_like(prev)
                rc = lib.syn3r_maxpool2_bwd_f16(L.ptr(prev), L.ptr(g), ph, pw, prev.shape[1], L.ptr(gp), L.stream_ptr(dev))
                L.check(rc, "syn3r_maxpool2_bwd_f16")
                g = gp                                 # the previous slice's layer term is ADDED to it at the top of the loop
        d_img = torch.empty((3, Hh, Ww), dtype=torch.float32, device=dev)
        L.check(lib.syn3r_lpips_image_bwd(L.ptr(g), Hh, Ww, float(scale), L.ptr(d_img), L.stream_ptr(dev)), "syn3r_lpips_image_bwd")
        return d_img * grad_out.to(torch.float32), None, None
