"""Gaussian rasteriser on the HIP path, behind the Python interface FSGS's trainer binds
(`diff_gaussian_rasterization`-style `GaussianRasterizationSettings` / `GaussianRasterizer`,
returning colour, radii, depth and alpha, with a per-Gaussian `confidence`).

The reference reaches it through `gsTrainer.render_view(cam)` (model/diffusionGS.py:154,166)
and the trainer's loss.backward() (:139,1640).  Kernels: syn3r_raster_* in include/syn3r_hip.h.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch

from .. import _lib as L


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor      # world_view_transform (transposed w2c, as FSGS cameras store it)
    projmatrix: torch.Tensor      # full_proj_transform
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool = False
    debug: bool = False


_host_cache: dict = {}


def _host_floats(m: torch.Tensor):
    """Camera matrices / background live on the GPU in FSGS cameras; the ABI takes them by value.
    One device->host read per tensor OBJECT and version (the cache holds a weak reference to the tensor, so
    a recycled device address or an in-place update can never alias a stale copy)."""
    import weakref
    key = id(m)
    hit = _host_cache.get(key)
    if hit is not None and hit[0]() is m and hit[1] == m._version:
        return hit[2]
    if len(_host_cache) > 1024:
        for k in [k for k, v in _host_cache.items() if v[0]() is None]:
            del _host_cache[k]
    vals = L.host_f32(m.detach().to("cpu", torch.float32).reshape(-1).tolist())
    _host_cache[key] = (weakref.ref(m), m._version, vals)
    return vals


def _host16(m: torch.Tensor):
    return _host_floats(m)


# ---- pair-count mode ------------------------------------------------------------------------------------
# "sync"  (default): read the exact number of (Gaussian, tile) pairs back after the projection stage, as the
#                    published implementation does, and size the binning buffer exactly.
# "async" (opt-in, training loops): size the binning buffer from the previous calls of the same shape (x2),
#                    let the kernels read the live count on the device, and check the overflow flag of call k
#                    at call k+1 (or at `flush_pair_checks()`): a call that overflowed raises then — its image
#                    was built from a truncated pair list.  Removes the one host<->device round trip per render.
_HEADROOM = 2.0          # async capacity = 2 x the largest pair count seen for the shape (24 B per pair: ~125 MB at 2.6 M pairs)
_pair_mode = "sync"
_capacity: dict = {}
_pending: dict = {}


def set_pair_count_mode(mode: str) -> None:
    global _pair_mode
    if mode not in ("sync", "async"):
        raise ValueError("pair-count mode must be 'sync' or 'async'")
    _pair_mode = mode


def get_pair_count_mode() -> str:
    return _pair_mode


def _check_pending(key, wait: bool) -> int:
    """Examine finished renders of this shape (all of them if `wait`), oldest first; never blocks the host unless
    asked to or more than 8 renders are unchecked.  Returns how many of the examined renders had outgrown their binning
    buffer (their images were built from a truncated pair list); the capacity of the shape is raised on the way."""
    queue = _pending.get(key)
    truncated = 0
    while queue:
        evt, host, cap = queue[0]
        if not (wait or len(queue) > 8 or evt.query()):
            break
        evt.synchronize()
        queue.pop(0)
        P, overflow = int(host[0]), int(host[1])
        _capacity[key] = max(_capacity.get(key, 0), int(P * _HEADROOM) + 4096)
        if overflow or P > cap:
            truncated += 1
    if queue is not None and not queue:
        del _pending[key]
    return truncated


def _overflow_error(n: int) -> "L.Syn3rError":
    return L.Syn3rError(f"rasteriser (async pair-count mode): {n} previous render(s) needed more (Gaussian, tile) pairs than "
                        "the binning buffer held; those images are invalid - capacity has been raised, render again")


def flush_pair_checks() -> int:
    """Verify the overflow flag of EVERY render still unchecked, for every shape (call at the end of a training loop).
    All queues are drained before anything is raised; the error carries the total in `.truncated`."""
    total = 0
    for key in list(_pending):
        total += _check_pending(key, wait=True)
    if total:
        err = _overflow_error(total)
        err.truncated = total
        raise err
    return 0


def carry_capacity(old_key, new_key, scale: float) -> None:
    """A change of the Gaussian count (densification) changes the capacity key (device, N, H, W).  Start the new shape
    from the old one's capacity scaled by N_new / N_old instead of from the next single view's count, and forget the
    old key (its renders have been checked by then: call after `flush_pair_checks`)."""
    cap = _capacity.pop(old_key, None)
    _pending.pop(old_key, None)
    if cap is not None:
        _capacity[new_key] = max(_capacity.get(new_key, 0), int(cap * max(scale, 1.0)) + 4096)


def capacity_key(dev: torch.device, N: int, H: int, W: int):
    return (dev.index, int(N), int(H), int(W))


class RasterState:
    """What `rasterize_forward` leaves for `rasterize_backward`: the (detached, contiguous fp32) inputs, the device state of the two
    forward stages and the host-side constants.  A plain object: the autograd Function stores its tensors through
    `save_for_backward`, a caller that drives the two passes itself (`GSTrainer._explicit_step`) just keeps it."""
    __slots__ = ("settings", "host", "P", "M", "plist", "has_conf", "opacity_shape", "raw_params", "tensors")


def rasterize_forward(means3D, shs, opacities, scales, rotations, confidence, settings: GaussianRasterizationSettings,
                      raw_params: bool = False):
    """Both forward stages (`syn3r_raster_preprocess[_raw]`, `syn3r_raster_render`); returns (color, radii, depth, alpha, state).
    `raw_params`: `scales`, `rotations`, `opacities` are the trainer's PARAMETERS (log-scales, unnormalised quaternions, logits);
    the published activations run inside the projection kernel and `rasterize_backward` returns the gradients of the parameters
    (`syn3r_raster_backward_raw`)."""
    s = settings
    dev = L.require_gpu(means3D, shs, opacities, scales, rotations)
    lib = L.load()
    N = means3D.shape[0]
    H, W = int(s.image_height), int(s.image_width)
    if N == 0:
        raise ValueError("rasteriser needs at least one Gaussian")
    M = shs.shape[1]
    f32 = lambda t: t.detach().to(torch.float32).contiguous()
    m3, sh, op, sc, ro = f32(means3D), f32(shs), f32(opacities).reshape(-1), f32(scales), f32(rotations)
    cf = f32(confidence).reshape(-1) if confidence is not None else None
    if m3.shape != (N, 3) or sc.shape != (N, 3) or ro.shape != (N, 4) or op.shape != (N,) or sh.shape != (N, M, 3):
        raise ValueError("rasteriser: inconsistent Gaussian tensor shapes")
    view, proj = _host16(s.viewmatrix), _host16(s.projmatrix)
    campos = _host_floats(s.campos)
    bg = _host_floats(s.bg)
    stream = L.stream_ptr(dev)

    geom = torch.empty(lib.syn3r_raster_geom_bytes(N), dtype=torch.uint8, device=dev)
    image = torch.empty(lib.syn3r_raster_image_bytes(H, W), dtype=torch.uint8, device=dev)
    radii = torch.empty(N, dtype=torch.int32, device=dev)
    key = (dev.index, N, H, W)
    use_async = _pair_mode == "async" and not s.debug
    if use_async:
        n_trunc = _check_pending(key, wait=False)
        if n_trunc:
            err = _overflow_error(n_trunc)
            err.truncated = n_trunc
            raise err
    use_async = use_async and key in _capacity
    P = C.c_longlong(0)
    preprocess = lib.syn3r_raster_preprocess_raw if raw_params else lib.syn3r_raster_preprocess
    rc = preprocess(N, int(s.sh_degree), M, L.ptr(m3), L.ptr(sc), L.ptr(ro), L.ptr(op), L.ptr(sh),
                    L.ptr(cf), float(s.scale_modifier), view, proj, campos, float(s.tanfovx),
                    float(s.tanfovy), H, W, L.ptr(radii), L.ptr(geom), geom.numel(),
                    None if use_async else C.byref(P), stream)
    L.check(rc, "syn3r_raster_preprocess")
    if use_async:
        P = _capacity[key]                          # capacity; the kernels read the live count on the device
    else:
        P = int(P.value)
        _capacity[key] = max(_capacity.get(key, 0), int(P * _HEADROOM) + 4096)
    binning = torch.empty(lib.syn3r_raster_binning_bytes(P), dtype=torch.uint8, device=dev)
    color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    alpha = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    plist = C.c_void_p(0)
    rc = lib.syn3r_raster_render(N, H, W, bg, L.ptr(radii), L.ptr(geom), geom.numel(), L.ptr(binning),
                                 binning.numel(), L.ptr(image), image.numel(), P, L.ptr(color), L.ptr(depth),
                                 L.ptr(alpha), C.byref(plist), stream)
    L.check(rc, "syn3r_raster_render")
    if use_async:
        host = torch.empty(2, dtype=torch.int32, pin_memory=True)
        host.copy_(geom[:8].view(torch.int32), non_blocking=True)
        evt = torch.cuda.Event()
        evt.record(torch.cuda.current_stream(dev))
        _pending.setdefault(key, []).append((evt, host, P))
    if s.debug:   # expose the binning / image state (tile ranges, sorted list) to the parity tests
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        a256 = lambda x: (x + 255) & ~255
        off = plist.value - binning.data_ptr() if P > 0 else 0
        _Rasterize.debug_state = dict(
            num_rendered=P,
            point_list=binning[off:off + 4 * P].view(torch.int32).clone(),
            ranges=image[:tiles * 8].view(torch.int32).reshape(tiles, 2).clone(),
            n_contrib=image[a256(tiles * 8):a256(tiles * 8) + 4 * H * W].view(torch.int32).reshape(H, W).clone(),
            depths=geom[256:256 + 4 * N].view(torch.float32).clone(),
        )
    st = RasterState()
    st.settings, st.host, st.P, st.M, st.plist = s, (view, proj, campos, bg), P, M, plist.value
    st.has_conf, st.opacity_shape, st.raw_params = cf is not None, opacities.shape, bool(raw_params)
    st.tensors = (m3, sc, ro, op, sh, cf if cf is not None else torch.empty(0, device=dev), radii, geom, binning, image)
    return color, radii, depth, alpha, st


def rasterize_backward(st: RasterState, g_color, g_depth=None, g_alpha=None):
    """Backward of both stages (`syn3r_raster_backward[_raw]`): (d_means3D, d_means2D, d_shs, d_opacities, d_scales, d_rotations,
    d_confidence or None) - with `raw_params` the gradients of the log-scales / raw quaternions / logits."""
    m3, sc, ro, op, sh, cf, radii, geom, binning, image = st.tensors
    s = st.settings
    lib = L.load()
    dev = m3.device
    N, M = m3.shape[0], st.M
    H, W = int(s.image_height), int(s.image_width)
    view, proj, campos, bg = st.host
    gc = g_color.detach().to(torch.float32).contiguous() if g_color is not None else torch.zeros((3, H, W), device=dev)
    gd = g_depth.detach().to(torch.float32).contiguous() if g_depth is not None else None
    ga = g_alpha.detach().to(torch.float32).contiguous() if g_alpha is not None else None
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    d_m3, d_sc, d_ro, d_op, d_sh, d_m2 = new(N, 3), new(N, 3), new(N, 4), new(N), new(N, M, 3), new(N, 3)
    d_cf = new(N) if st.has_conf else None
    ws = L.workspace(dev, lib.syn3r_raster_backward_workspace_bytes(N), "raster_bwd")
    backward = lib.syn3r_raster_backward_raw if st.raw_params else lib.syn3r_raster_backward
    rc = backward(
        N, int(s.sh_degree), M, st.P, L.ptr(m3), L.ptr(sc), L.ptr(ro), L.ptr(op), L.ptr(sh),
        L.ptr(cf) if st.has_conf else None, float(s.scale_modifier), view, proj, campos, float(s.tanfovx),
        float(s.tanfovy), H, W, bg, L.ptr(radii), L.ptr(geom), geom.numel(), st.plist, L.ptr(image), image.numel(),
        L.ptr(gc), L.ptr(gd), L.ptr(ga), L.ptr(d_m3), L.ptr(d_sc), L.ptr(d_ro), L.ptr(d_op), L.ptr(d_sh),
        L.ptr(d_m2), L.ptr(d_cf), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    L.check(rc, "syn3r_raster_backward")
    return d_m3, d_m2, d_sh, d_op.reshape(st.opacity_shape), d_sc, d_ro, d_cf


class _Rasterize(torch.autograd.Function):
    """autograd around `rasterize_forward` / `rasterize_backward` (activated tensors in, as the published rasteriser takes them)."""
    debug_state = None

    @staticmethod
    def forward(ctx, means3D, means2D, shs, opacities, scales, rotations, confidence, settings):
        color, radii, depth, alpha, st = rasterize_forward(means3D, shs, opacities, scales, rotations, confidence, settings)
        ctx.save_for_backward(*st.tensors)
        st.tensors = None
        ctx.state = st
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)     # unused outputs (depth / alpha) arrive as None, not as zero tensors
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_alpha):
        L.join_active_trace()        # autograd thread: record into the caller's kernel_trace session, if one is open
        st = ctx.state
        st.tensors = ctx.saved_tensors
        try:
            d_m3, d_m2, d_sh, d_op, d_sc, d_ro, d_cf = rasterize_backward(st, g_color, g_depth, g_alpha)
        finally:
            st.tensors = None
        return d_m3, d_m2, d_sh, d_op, d_sc, d_ro, d_cf, None


class GaussianRasterizer(torch.nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, confidence: Optional[torch.Tensor] = None):
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if colors_precomp is not None or cov3D_precomp is not None:
            raise NotImplementedError("precomputed colours / covariances are not used by the SYN3R hot path")
        if scales is None or rotations is None:
            raise Exception("Please provide scales and rotations")
        return _Rasterize.apply(means3D, means2D, shs, opacities, scales, rotations, confidence, self.raster_settings)


def sort_pairs(keys: torch.Tensor, vals: torch.Tensor, nbits: int = 64):
    """Stable radix sort of (int64/uint64-bit-pattern key, int32 value) pairs on the GPU."""
    dev = L.require_gpu(keys, vals)
    lib = L.load()
    n = keys.numel()
    k = keys.contiguous().clone()
    v = vals.contiguous().clone()
    kt, vt = torch.empty_like(k), torch.empty_like(v)
    ws = L.workspace(dev, lib.syn3r_sort_pairs_workspace_bytes(n), "sort")
    flag = C.c_int(0)
    rc = lib.syn3r_sort_pairs(L.ptr(k), L.ptr(v), L.ptr(kt), L.ptr(vt), n, nbits, L.ptr(ws), ws.numel(), C.byref(flag),
                              L.stream_ptr(dev))
    L.check(rc, "syn3r_sort_pairs")
    return (kt, vt) if flag.value else (k, v)
