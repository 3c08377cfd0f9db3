"""Thin torch-tensor wrappers over the UNet operators of include/syn3r_hip.h.

Every function takes contiguous fp16 HIP tensors in the channels-last token-matrix layout
(rows = ((b*F + f)*h + y)*w + x, columns = channels) and launches on torch's current stream.
No CPU fallback: CPU tensors raise `Syn3rError`.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import _lib as L

H = torch.float16

# algorithmic FLOP accounting for bench.py's roofline line (2*M*N*K per contraction)
FLOPS = {"enabled": False, "gemm": 0.0, "attn": 0.0}


def _count(kind: str, flops: float) -> None:
    if FLOPS["enabled"]:
        FLOPS[kind] += flops


def _chk(*ts):
    dev = L.require_gpu(*[t for t in ts if t is not None])
    for t in ts:
        if t is not None and (t.dtype != H or not t.is_contiguous()):
            raise L.Syn3rError(f"UNet operators need contiguous fp16 tensors, got {t.dtype} contiguous={t.is_contiguous()}")
    return dev


def _gn_request(dev, M: int, N: int, want: bool):
    """Ask the NEXT contraction of this thread for GroupNorm partial sums of its [M, N] output (include/syn3r_hip.h
    syn3r_gemm_set_gn_partials): returns the buffer, or None when the shape has no whole 32-row blocks / 80-column groups."""
    if not want:
        return None
    lib = L.load()
    nb = lib.syn3r_gn_partials_bytes(int(M), int(N))
    if not nb:
        return None
    part = torch.empty(nb // 4, dtype=torch.float32, device=dev)
    L.check(lib.syn3r_gemm_set_gn_partials(part.data_ptr(), nb), "syn3r_gemm_set_gn_partials")
    return part


def _gn_collect(out: torch.Tensor, part) -> torch.Tensor:
    """Attach the partial sums to `out` (attribute `gn_part`) if the kernel that ran wrote them."""
    if part is not None:
        lib = L.load()
        if lib.syn3r_gemm_gn_partials_written():
            out.gn_part = part
        else:
            lib.syn3r_gemm_set_gn_partials(None, 0)        # (an entry that returned before consuming the request)
    return out


def keep_gn(view: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """`view` is a reshape of `src` with the same rows x channels matrix: carry the producer's GroupNorm partial sums over."""
    part = getattr(src, "gn_part", None)
    if part is not None:
        view.gn_part = part
    return view


def _rowvec(rv: Optional[torch.Tensor]):
    """(pointer, row stride) of a per-sample row-vector operand: a [V, N] fp16 matrix or a column slice of a wider one."""
    if rv is None:
        return None, 0
    if rv.dtype != H or rv.dim() != 2 or rv.stride(1) != 1 or not rv.is_cuda:
        raise L.Syn3rError("rowvec must be a 2-D fp16 HIP tensor with unit column stride")
    return rv.data_ptr(), rv.stride(0)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
           rowvec: Optional[torch.Tensor] = None, rows_per_vec: int = 0, rv_group_rows: int = 0,
           residual: Optional[torch.Tensor] = None,
           aux: Optional[torch.Tensor] = None, s_acc: float = 1.0, s_res: float = 1.0, s_aux: float = 1.0,
           out: Optional[torch.Tensor] = None, gn_stats: bool = False) -> torch.Tensor:
    """x [M,K] @ weight[N,K]^T with the fused epilogue of syn3r_gemm_f16.  x may be a column
    slice of a wider matrix (stride(0) >= K).  rows_per_vec / rv_group_rows: see include/syn3r_hip.h.
    gn_stats: ask the kernel for GroupNorm partial sums of the output (attribute `gn_part` of the result when written)."""
    dev = L.require_gpu(x, weight)
    M, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] != K or x.stride(1) != 1:
        raise ValueError(f"linear: x {tuple(x.shape)} / weight {tuple(weight.shape)} mismatch")
    _chk(weight, bias, residual, aux)
    rv_ptr, rv_ld = _rowvec(rowvec)
    if out is None:
        out = torch.empty((M, N), dtype=H, device=dev)
    lib = L.load()
    part = _gn_request(dev, M, N, gn_stats and out.stride(0) == N)
    rc = lib.syn3r_gemm_f16(x.data_ptr(), x.stride(0), L.ptr(weight), out.data_ptr(), out.stride(0), L.ptr(bias),
                            rv_ptr, rv_ld, int(rows_per_vec), int(rv_group_rows),
                            residual.data_ptr() if residual is not None else None,
                            residual.stride(0) if residual is not None else 0,
                            aux.data_ptr() if aux is not None else None, aux.stride(0) if aux is not None else 0,
                            float(s_acc), float(s_res), float(s_aux), M, N, K, L.stream_ptr(dev))
    L.check(rc, "syn3r_gemm_f16")
    _count("gemm", 2.0 * M * N * K)
    return _gn_collect(out, part)


def pack_geglu(weight: torch.Tensor, bias: torch.Tensor):
    """Regroup the [2D, K] GEGLU projection (hidden rows, then gate rows) per 80-column output tile as
    [80 hidden | 80 gate] rows, zero-padded: the layout syn3r_gemm_geglu_f16 expects."""
    D2, K = weight.shape
    D = D2 // 2
    tiles = (D + 79) // 80
    wp = torch.zeros((tiles * 160, K), dtype=weight.dtype, device=weight.device)
    bp = torch.zeros((tiles * 160,), dtype=bias.dtype, device=bias.device)
    for t in range(tiles):
        n = min(80, D - 80 * t)
        wp[160 * t:160 * t + n] = weight[80 * t:80 * t + n]
        wp[160 * t + 80:160 * t + 80 + n] = weight[D + 80 * t:D + 80 * t + n]
        bp[160 * t:160 * t + n] = bias[80 * t:80 * t + n]
        bp[160 * t + 80:160 * t + 80 + n] = bias[D + 80 * t:D + 80 * t + n]
    return wp.contiguous(), bp.contiguous(), D


def pack_geglu64(weight: torch.Tensor, bias: torch.Tensor):
    """Regroup the [2D, K] GEGLU projection per 64-wide hidden chunk j as [hidden 64j..+63 | gate 64j..+63] rows (D a multiple
    of 64): the layout syn3r_feedforward_p64_f16 (k_gemm_g256) expects."""
    D2, K = weight.shape
    D = D2 // 2
    if D % 64:
        raise ValueError("pack_geglu64: D must be a multiple of 64")
    wp = torch.stack([weight[:D].view(D // 64, 64, K), weight[D:].view(D // 64, 64, K)], dim=1).reshape(D2, K)
    bp = torch.stack([bias[:D].view(D // 64, 64), bias[D:].view(D // 64, 64)], dim=1).reshape(D2)
    return wp.contiguous(), bp.contiguous(), D


def linear_geglu(x: torch.Tensor, wpacked: torch.Tensor, bpacked: torch.Tensor, D: int) -> torch.Tensor:
    """geglu(x @ W^T + b) in one kernel: [M,K] -> [M,D]."""
    dev = _chk(wpacked, bpacked)
    L.require_gpu(x)
    M, K = x.shape
    out = torch.empty((M, D), dtype=H, device=dev)
    rc = L.load().syn3r_gemm_geglu_f16(x.data_ptr(), x.stride(0), L.ptr(wpacked), L.ptr(bpacked), L.ptr(out), D, M, D, K,
                                       L.stream_ptr(dev))
    L.check(rc, "syn3r_gemm_geglu_f16")
    _count("gemm", 2.0 * M * 2 * D * K)
    return out


def feedforward(x: torch.Tensor, w1_packed: torch.Tensor, b1_packed: torch.Tensor, D: int, w2: torch.Tensor,
                b2: Optional[torch.Tensor] = None, *, residual: Optional[torch.Tensor] = None,
                aux: Optional[torch.Tensor] = None, s_acc: float = 1.0, s_res: float = 1.0, s_aux: float = 1.0,
                packed64: Optional[tuple] = None) -> torch.Tensor:
    """FeedForward.forward with the GEGLU activation (attention.py:608-665): geglu(x @ W1^T + b1) @ W2^T + b2 with
    the fused epilogue of `linear`; the hidden activation stays in a tiled workspace (syn3r_feedforward_f16).
    packed64 = (w1, b1) in `pack_geglu64`'s layout: net.0 runs on the 256 x 256 tile (syn3r_feedforward_p64_f16) when the
    shape has whole tiles; the result is the same bit for bit."""
    dev = _chk(w1_packed, b1_packed, w2, b2, residual, aux)
    L.require_gpu(x)
    M, K = x.shape
    N = w2.shape[0]
    if x.stride(1) != 1 or w2.shape[1] != D or w1_packed.shape[1] != K:
        raise ValueError(f"feedforward: x {tuple(x.shape)} / w1 {tuple(w1_packed.shape)} / w2 {tuple(w2.shape)} / D={D} mismatch")
    out = torch.empty((M, N), dtype=H, device=dev)
    lib = L.load()
    ws = L.workspace(dev, lib.syn3r_feedforward_workspace_bytes(M, D), "ff")
    fn, name = lib.syn3r_feedforward_f16, "syn3r_feedforward_f16"
    if packed64 is not None and x.stride(0) == K and lib.syn3r_feedforward_p64_supported(M, D, K):
        w1_packed, b1_packed = packed64
        fn, name = lib.syn3r_feedforward_p64_f16, "syn3r_feedforward_p64_f16"
    rc = fn(x.data_ptr(), x.stride(0), L.ptr(w1_packed), L.ptr(b1_packed), D, L.ptr(w2), L.ptr(b2),
            L.ptr(out), N, residual.data_ptr() if residual is not None else None, residual.stride(0) if residual is not None else 0,
            aux.data_ptr() if aux is not None else None, aux.stride(0) if aux is not None else 0,
            float(s_acc), float(s_res), float(s_aux), M, K, N, L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    L.check(rc, name)
    _count("gemm", 2.0 * M * 2 * D * K + 2.0 * M * N * D)
    return out


def pack_geglu_chunked(weight: torch.Tensor, bias: torch.Tensor):
    """Regroup the [2D, K] GEGLU projection per 64-wide hidden chunk j as 4 x [hidden 64j+16q..+15 | gate 64j+16q..+15]
    (q = 0..3): the layout syn3r_feedforward_fused_f16 expects (w1_chunked [D/64 * 128, K], b1_chunked [D/64 * 128])."""
    D2, K = weight.shape
    D = D2 // 2
    if D % 64:
        raise ValueError("pack_geglu_chunked: hidden width must be a multiple of 64")
    idx = []
    for j in range(D // 64):
        for q in (0, 16, 32, 48):
            idx.extend(range(64 * j + q, 64 * j + q + 16))
            idx.extend(range(D + 64 * j + q, D + 64 * j + q + 16))
    idx = torch.tensor(idx, device=weight.device)
    return weight.index_select(0, idx).contiguous(), bias.index_select(0, idx).contiguous(), D


FUSED_FF_CHANNELS = 320      # syn3r_feedforward_fused_f16 is built for this width (level 0 of the SVD UNet)


def feedforward_fused(x: torch.Tensor, w1_chunked: torch.Tensor, b1_chunked: torch.Tensor, D: int, w2: torch.Tensor,
                      b2: Optional[torch.Tensor] = None, *, residual: Optional[torch.Tensor] = None,
                      aux: Optional[torch.Tensor] = None, s_acc: float = 1.0, s_res: float = 1.0, s_aux: float = 1.0,
                      ln: Optional[tuple] = None, addvec: Optional[tuple] = None) -> torch.Tensor:
    """FeedForward.forward (attention.py:608-665) in ONE kernel for C = 320: geglu(x @ W1^T + b1) @ W2^T + b2 with the
    `linear` epilogue; the hidden activation never leaves the CU (syn3r_feedforward_fused_f16).
    ln = (gamma, beta, eps): x is LayerNorm'ed inside the kernel first (norm3 -> ff, attention.py:376-392; equal bit for bit
    to `layernorm` followed by this call; syn3r_feedforward_fused_ln_f16).
    addvec = (vec [rows, C], rows_per_vec), with ln: x + vec[row // rows_per_vec] (fp16 tensor add) is what gets normalised AND the
    residual (norm_in / ff_in of the temporal block, attention.py:500-517; `residual` must be left None;
    syn3r_feedforward_fused_addln_f16)."""
    dev = _chk(w1_chunked, b1_chunked, w2, b2, residual, aux)
    L.require_gpu(x)
    M, K = x.shape
    N = w2.shape[0]
    if x.stride(1) != 1 or K != N or w2.shape[1] != D or tuple(w1_chunked.shape) != (2 * D, K):
        raise ValueError(f"feedforward_fused: x {tuple(x.shape)} / w1 {tuple(w1_chunked.shape)} / w2 {tuple(w2.shape)} / D={D} mismatch")
    out = torch.empty((M, N), dtype=H, device=dev)
    tail = (L.ptr(w1_chunked), L.ptr(b1_chunked), D, L.ptr(w2), L.ptr(b2), L.ptr(out), N,
            residual.data_ptr() if residual is not None else None, residual.stride(0) if residual is not None else 0,
            aux.data_ptr() if aux is not None else None, aux.stride(0) if aux is not None else 0,
            float(s_acc), float(s_res), float(s_aux), M, K, L.stream_ptr(dev))
    if addvec is not None and (ln is None or residual is not None):
        raise ValueError("feedforward_fused: addvec needs ln and takes the place of the residual")
    if ln is not None:
        gamma, beta, eps = ln
        _chk(gamma, beta)
        if gamma.numel() != K or beta.numel() != K:
            raise ValueError(f"feedforward_fused: LayerNorm parameters of {gamma.numel()} / {beta.numel()} channels for C = {K}")
    if addvec is not None:
        vec, rpv = addvec
        _chk(vec)
        if vec.dim() != 2 or vec.shape[1] != K or not vec.is_contiguous() or rpv <= 0 or (M + rpv - 1) // rpv > vec.shape[0]:
            raise ValueError(f"feedforward_fused: add vector {tuple(vec.shape)} / rows_per_vec={rpv} for x {tuple(x.shape)}")
        rc = L.load().syn3r_feedforward_fused_addln_f16(x.data_ptr(), x.stride(0), L.ptr(vec), int(rpv), L.ptr(gamma), L.ptr(beta), float(eps),
                                                        *tail[:7], *tail[9:])
        L.check(rc, "syn3r_feedforward_fused_addln_f16")
    elif ln is not None:
        rc = L.load().syn3r_feedforward_fused_ln_f16(x.data_ptr(), x.stride(0), L.ptr(gamma), L.ptr(beta), float(eps), *tail)
        L.check(rc, "syn3r_feedforward_fused_ln_f16")
    else:
        rc = L.load().syn3r_feedforward_fused_f16(x.data_ptr(), x.stride(0), *tail)
        L.check(rc, "syn3r_feedforward_fused_f16")
    _count("gemm", 2.0 * M * 2 * D * K + 2.0 * M * N * D)
    return out


def conv3x3(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, stride: int = 1,
            upsample: bool = False, rowvec: Optional[torch.Tensor] = None, rows_per_vec: int = 0,
            residual: Optional[torch.Tensor] = None, s_acc: float = 1.0, s_res: float = 1.0,
            pad_lo: int = 1, gn_stats: bool = False) -> torch.Tensor:
    """x [NB,Hi,Wi,Cin] NHWC, weight [Cout,3,3,Cin] -> [NB,Ho,Wo,Cout].  pad_lo = 0: the (0,1,0,1) padding of the
    VAE encoder's stride-2 Downsample2D(padding=0)."""
    dev = _chk(x, weight, bias, residual)
    rv_ptr, rv_ld = _rowvec(rowvec)
    NB, Hi, Wi, Cin = x.shape
    Cout = weight.shape[0]
    if tuple(weight.shape[1:]) != (3, 3, Cin):
        raise ValueError(f"conv3x3: weight {tuple(weight.shape)} does not match Cin={Cin}")
    Hg, Wg = (2 * Hi, 2 * Wi) if upsample else (Hi, Wi)
    Ho, Wo = (Hg + pad_lo - 2) // stride + 1, (Wg + pad_lo - 2) // stride + 1
    out = torch.empty((NB, Ho, Wo, Cout), dtype=H, device=dev)
    lib = L.load()
    part = _gn_request(dev, NB * Ho * Wo, Cout, gn_stats)
    rc = lib.syn3r_conv2d3x3_f16(L.ptr(x), L.ptr(weight), L.ptr(out), Cout, L.ptr(bias), rv_ptr, rv_ld, int(rows_per_vec),
                                 L.ptr(residual), Cout if residual is not None else 0, float(s_acc), float(s_res),
                                 NB, Hi, Wi, Cin, Cout, int(stride), 1 if upsample else 0, int(pad_lo),
                                 L.stream_ptr(dev))
    L.check(rc, "syn3r_conv2d3x3_f16")
    _count("gemm", 2.0 * NB * Ho * Wo * Cout * 9 * Cin)
    return _gn_collect(out, part)


def tconv3(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], B: int, F: int, HW: int, *,
           rowvec: Optional[torch.Tensor] = None, rows_per_vec: int = 0, residual: Optional[torch.Tensor] = None,
           s_acc: float = 1.0, s_res: float = 1.0, gn_stats: bool = False) -> torch.Tensor:
    """x [B*F*HW, Cin], weight [Cout,3,Cin] -> [B*F*HW, Cout] (3-tap convolution over frames)."""
    dev = _chk(x, weight, bias, residual)
    rv_ptr, rv_ld = _rowvec(rowvec)
    M, Cin = x.shape
    Cout = weight.shape[0]
    if M != B * F * HW or tuple(weight.shape[1:]) != (3, Cin):
        raise ValueError("tconv3: shape mismatch")
    out = torch.empty((M, Cout), dtype=H, device=dev)
    lib = L.load()
    part = _gn_request(dev, M, Cout, gn_stats)
    rc = lib.syn3r_tconv3_f16(L.ptr(x), L.ptr(weight), L.ptr(out), Cout, L.ptr(bias), rv_ptr, rv_ld, int(rows_per_vec), L.ptr(residual),
                              Cout if residual is not None else 0, float(s_acc), float(s_res), B, F, HW, Cin, Cout,
                              L.stream_ptr(dev))
    L.check(rc, "syn3r_tconv3_f16")
    _count("gemm", 2.0 * M * Cout * 3 * Cin)
    return _gn_collect(out, part)


def attention(qkv: torch.Tensor, nseq: int, S: int, heads: int) -> torch.Tensor:
    """qkv [nseq*S, 3*heads*64] (q | k | v thirds) -> [nseq*S, heads*64]."""
    dev = _chk(qkv)
    C = heads * 64
    if qkv.shape != (nseq * S, 3 * C):
        raise ValueError(f"attention: qkv {tuple(qkv.shape)} != {(nseq * S, 3 * C)}")
    out = torch.empty((nseq * S, C), dtype=H, device=dev)
    base = qkv.data_ptr()
    lib = L.load()
    rc = lib.syn3r_attention_f16(base, base + 2 * C, base + 4 * C, 3 * C, L.ptr(out), C, nseq, S, heads,
                                 L.stream_ptr(dev))
    L.check(rc, "syn3r_attention_f16")
    _count("attn", 4.0 * nseq * heads * S * S * 64)
    return out


def attention_temporal(qkv: torch.Tensor, B: int, F: int, HW: int, heads: int) -> torch.Tensor:
    dev = _chk(qkv)
    C = heads * 64
    if qkv.shape != (B * F * HW, 3 * C):
        raise ValueError(f"attention_temporal: qkv {tuple(qkv.shape)} != {(B * F * HW, 3 * C)}")
    out = torch.empty((B * F * HW, C), dtype=H, device=dev)
    base = qkv.data_ptr()
    lib = L.load()
    rc = lib.syn3r_attention_temporal_f16(base, base + 2 * C, base + 4 * C, 3 * C, L.ptr(out), C, B, F, HW, heads,
                                          L.stream_ptr(dev))
    L.check(rc, "syn3r_attention_temporal_f16")
    _count("attn", 4.0 * B * HW * heads * F * F * 64)
    return out


def groupnorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, samples: int, eps: float, silu: bool,
              x2: Optional[torch.Tensor] = None, use_partials: bool = True) -> torch.Tensor:
    """x [samples*rows, C] -> same shape; 32 groups, statistics per (sample, group).  With `x2` [samples*rows, C2] the
    input is the channel concatenation [x | x2], read in place (the result has C + C2 channels).
    Inputs that carry their producer's partial sums (attribute `gn_part`, see `linear(gn_stats=True)`) skip the statistics
    pass; use_partials = False forces it."""
    dev = _chk(x, gamma, beta, x2)
    M, C = x.shape
    if M % samples:
        raise ValueError("groupnorm: rows not divisible by samples")
    lib = L.load()
    ws = L.workspace(dev, lib.syn3r_groupnorm_workspace_bytes(samples, M // samples), "gn")
    # statistics from the producers' epilogues (syn3r_groupnorm_pre_f16) when every source carries its partial sums
    p1 = getattr(x, "gn_part", None) if use_partials else None
    p2 = getattr(x2, "gn_part", None) if (use_partials and x2 is not None) else None
    Ct = C + (x2.shape[1] if x2 is not None else 0)
    if p1 is not None and (x2 is None or p2 is not None) and (M // samples) % 32 == 0 and Ct % 320 == 0 and C % 10 == 0:
        if x2 is not None and x2.shape[0] != M:
            raise ValueError("groupnorm: the two sources must have the same rows")
        y = torch.empty((M, Ct), dtype=H, device=dev)
        rc = lib.syn3r_groupnorm_pre_f16(L.ptr(x), C, L.ptr(p1), L.ptr(x2), Ct - C, L.ptr(p2), L.ptr(y), samples, M // samples,
                                         L.ptr(gamma), L.ptr(beta), float(eps), 1 if silu else 0, L.ptr(ws), ws.numel(), L.stream_ptr(dev))
        L.check(rc, "syn3r_groupnorm_pre_f16")
        return y
    if x2 is not None:
        if x2.shape[0] != M:
            raise ValueError("groupnorm: the two sources must have the same rows")
        C2 = x2.shape[1]
        y = torch.empty((M, C + C2), dtype=H, device=dev)
        rc = lib.syn3r_groupnorm_2src_f16(L.ptr(x), C, L.ptr(x2), C2, L.ptr(y), samples, M // samples, L.ptr(gamma), L.ptr(beta),
                                          float(eps), 1 if silu else 0, L.ptr(ws), ws.numel(), L.stream_ptr(dev))
        L.check(rc, "syn3r_groupnorm_2src_f16")
        return y
    y = torch.empty_like(x)
    rc = lib.syn3r_groupnorm_f16(L.ptr(x), L.ptr(y), samples, M // samples, C, L.ptr(gamma), L.ptr(beta), float(eps),
                                 1 if silu else 0, L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    L.check(rc, "syn3r_groupnorm_f16")
    return y


def linear_cat(x1: torch.Tensor, x2: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[x1 | x2] @ weight^T + bias without writing the concatenation (the shortcut projection of an up block's resnet on
    torch.cat([hidden, skip], dim=1), resnet.py:316).  Shapes the two-source kernel does not serve are concatenated."""
    dev = L.require_gpu(x1, x2, weight)
    M, K1 = x1.shape
    K2 = x2.shape[1]
    N = weight.shape[0]
    if x2.shape[0] != M or weight.shape[1] != K1 + K2 or x1.stride(1) != 1 or x2.stride(1) != 1:
        raise ValueError(f"linear_cat: x1 {tuple(x1.shape)} / x2 {tuple(x2.shape)} / weight {tuple(weight.shape)} mismatch")
    _chk(weight, bias)
    lib = L.load()
    if not lib.syn3r_gemm_2src_supported(M, N, K1, K2, x1.stride(0), x2.stride(0)):
        return linear(torch.cat([x1, x2], dim=1), weight, bias)
    out = torch.empty((M, N), dtype=H, device=dev)
    rc = lib.syn3r_gemm_2src_f16(x1.data_ptr(), x1.stride(0), K1, x2.data_ptr(), x2.stride(0), K2, L.ptr(weight), out.data_ptr(),
                                 out.stride(0), L.ptr(bias), M, N, L.stream_ptr(dev))
    L.check(rc, "syn3r_gemm_2src_f16")
    _count("gemm", 2.0 * M * N * (K1 + K2))
    return out


def layernorm_linear(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, weight: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """LayerNorm(x) @ weight^T (no bias) - norm1 followed by the stacked to_q / to_k / to_v projection (attention.py:340-352,
    509-512).  At C = 320 with an output width that is a multiple of 320: ONE kernel, the normalised activation stays on chip
    (syn3r_layernorm_linear320_f16); otherwise `layernorm` followed by `linear`."""
    dev = _chk(gamma, beta, weight)
    L.require_gpu(x)
    M, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] != K or gamma.numel() != K or beta.numel() != K or x.stride(1) != 1:
        raise ValueError(f"layernorm_linear: x {tuple(x.shape)} / weight {tuple(weight.shape)} / norm {gamma.numel()} mismatch")
    if K != FUSED_FF_CHANNELS or N % FUSED_FF_CHANNELS or x.stride(0) % 8 or x.data_ptr() % 16:
        return linear(layernorm(x.contiguous(), gamma, beta, eps), weight)
    out = torch.empty((M, N), dtype=H, device=dev)
    rc = L.load().syn3r_layernorm_linear320_f16(x.data_ptr(), x.stride(0), L.ptr(gamma), L.ptr(beta), float(eps), L.ptr(weight),
                                                out.data_ptr(), out.stride(0), M, N, K, L.stream_ptr(dev))
    L.check(rc, "syn3r_layernorm_linear320_f16")
    _count("gemm", 2.0 * M * N * K)
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, *,
              addvec: Optional[torch.Tensor] = None, rows_per_vec: int = 0, want_sum: bool = False):
    dev = _chk(x, gamma, beta, addvec)
    M, C = x.shape
    y = torch.empty_like(x)
    xsum = torch.empty_like(x) if (want_sum and addvec is not None) else None
    lib = L.load()
    rc = lib.syn3r_layernorm_f16(L.ptr(x), L.ptr(y), L.ptr(xsum), L.ptr(addvec), int(rows_per_vec), M, C,
                                 L.ptr(gamma), L.ptr(beta), float(eps), L.stream_ptr(dev))
    L.check(rc, "syn3r_layernorm_f16")
    return (y, xsum) if want_sum else y


def geglu(x: torch.Tensor) -> torch.Tensor:
    dev = _chk(x)
    M, D2 = x.shape
    y = torch.empty((M, D2 // 2), dtype=H, device=dev)
    rc = L.load().syn3r_geglu_f16(L.ptr(x), L.ptr(y), M, D2 // 2, L.stream_ptr(dev))
    L.check(rc, "syn3r_geglu_f16")
    return y


def softmax_rows(x: torch.Tensor, scale: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(scale * x, dim=1) of an fp16 [M,N] matrix in fp32 arithmetic (in place when out is x)."""
    dev = _chk(x)
    M, N = x.shape
    if out is None:
        out = torch.empty_like(x)
    rc = L.load().syn3r_softmax_rows_f16(L.ptr(x), L.ptr(out), M, N, x.stride(0), float(scale), L.stream_ptr(dev))
    L.check(rc, "syn3r_softmax_rows_f16")
    return out


def attention_wide(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Single-head attention of ONE sequence with an arbitrary head width D (the VAE mid blocks: D = 512):
    q, k, v [S,D] -> [S,D] as two contractions around a row softmax (the S x S score matrix is materialised
    in fp16: 170 MB at S = 9216)."""
    S, D = q.shape
    scores = linear(q, k, s_acc=1.0 / (D ** 0.5))                 # [S,S] = q . k^T / sqrt(D)
    softmax_rows(scores, 1.0, out=scores)
    return linear(scores, v.t().contiguous())                     # [S,D] = P . v


def time_conv_out(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, B: int, F: int, HW: int) -> torch.Tensor:
    """TemporalDecoder.time_conv_out: x [B*F*HW, >=3] fp16 channels-last, weight [3,3,3,1,1], bias [3]
    -> [B*F, 3, HW] fp32."""
    dev = _chk(x)
    if x.shape[0] != B * F * HW or x.shape[1] < 3 or weight.numel() != 27 or bias.numel() != 3:
        raise ValueError("time_conv_out: shape mismatch")
    out = torch.empty((B * F, 3, HW), dtype=torch.float32, device=dev)
    w = L.host_f32(weight.detach().float().cpu().reshape(-1).tolist())
    b = L.host_f32(bias.detach().float().cpu().reshape(-1).tolist())
    rc = L.load().syn3r_time_conv_out(L.ptr(x), x.stride(0), w, b, L.ptr(out), B, F, HW, L.stream_ptr(dev))
    L.check(rc, "syn3r_time_conv_out")
    return out
