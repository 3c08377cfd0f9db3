"""ctypes binding of libsyn3r_hip.so (the C-ABI in include/syn3r_hip.h).

There is no CPU fallback: if the shared library is missing, or a call is made
with tensors that are not on a HIP device, this module raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import torch

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libsyn3r_hip.so"
_lib = None

c_f = C.c_float
c_i = C.c_int
c_p = C.c_void_p
c_sz = C.c_size_t
c_ll = C.c_longlong
c_d = C.c_double

F32, F16 = 0, 1

# name -> (restype, argtypes); mirrors include/syn3r_hip.h declaration by declaration
SIGNATURES = {
    "syn3r_last_error": (C.c_char_p, []),
    "syn3r_version": (c_i, []),
    "syn3r_arch": (C.c_char_p, []),
    "syn3r_trace_enable": (c_i, [c_i]),
    "syn3r_trace_filter": (c_i, [C.c_char_p]),
    "syn3r_trace_report": (c_i, [C.c_char_p, c_sz]),
    "syn3r_trace_session": (c_p, []),
    "syn3r_trace_attach": (c_i, [c_p]),
    "syn3r_inverse_warp_workspace_bytes": (c_sz, [c_i]),
    "syn3r_inverse_warp": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_i, c_i,
                                 c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "syn3r_reproj_error": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p]),
    "syn3r_warp_post": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "syn3r_fuse_uncertainty": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "syn3r_forward_warp_workspace_bytes": (c_sz, [c_i, c_i]),
    "syn3r_forward_warp": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "syn3r_step_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "syn3r_step_interp": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_i,
                                c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "syn3r_step_replace": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_f, c_f, c_f, c_f,
                                 c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "syn3r_raster_geom_bytes": (c_sz, [c_i]),
    "syn3r_raster_image_bytes": (c_sz, [c_i, c_i]),
    "syn3r_raster_binning_bytes": (c_sz, [c_ll]),
    "syn3r_raster_preprocess": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_f, c_f,
                                      c_i, c_i, c_p, c_p, c_sz, C.POINTER(c_ll), c_p]),
    "syn3r_raster_preprocess_raw": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_f, c_f,
                                          c_i, c_i, c_p, c_p, c_sz, C.POINTER(c_ll), c_p]),
    "syn3r_raster_render": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_sz, c_p, c_sz, c_p, c_sz, c_ll, c_p, c_p, c_p,
                                  C.POINTER(c_p), c_p]),
    "syn3r_raster_backward_workspace_bytes": (c_sz, [c_i]),
    "syn3r_raster_backward": (c_i, [c_i, c_i, c_i, c_ll, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_f, c_f,
                                    c_i, c_i, c_p, c_p, c_p, c_sz, c_p, c_p, c_sz, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                    c_p, c_p, c_p, c_p, c_sz, c_p]),
    "syn3r_raster_backward_raw": (c_i, [c_i, c_i, c_i, c_ll, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_f, c_f,
                                        c_i, c_i, c_p, c_p, c_p, c_sz, c_p, c_p, c_sz, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                        c_p, c_p, c_p, c_p, c_sz, c_p]),
    "syn3r_sort_pairs_workspace_bytes": (c_sz, [c_ll]),
    "syn3r_sort_pairs": (c_i, [c_p, c_p, c_p, c_p, c_ll, c_i, c_p, c_sz, C.POINTER(c_i), c_p]),
    "syn3r_gaussian_activate": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "syn3r_gaussian_activate_backward": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "syn3r_densification_stats": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "syn3r_gemm_f16": (c_i, [c_p, c_ll, c_p, c_p, c_ll, c_p, c_p, c_ll, c_i, c_i, c_p, c_ll, c_p, c_ll, c_f, c_f, c_f,
                             c_i, c_i, c_i, c_p]),
    "syn3r_gemm_set_tile": (c_i, [c_i]),
    "syn3r_gemm_geglu_f16": (c_i, [c_p, c_ll, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_p]),
    "syn3r_feedforward_workspace_bytes": (c_sz, [c_i, c_i]),
    "syn3r_feedforward_f16": (c_i, [c_p, c_ll, c_p, c_p, c_i, c_p, c_p, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_f, c_f, c_f,
                                    c_i, c_i, c_i, c_p, c_sz, c_p]),
    "syn3r_feedforward_p64_f16": (c_i, [c_p, c_ll, c_p, c_p, c_i, c_p, c_p, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_f, c_f, c_f,
                                    c_i, c_i, c_i, c_p, c_sz, c_p]),
    "syn3r_feedforward_p64_supported": (c_i, [c_i, c_i, c_i]),
    "syn3r_feedforward_fused_f16": (c_i, [c_p, c_ll, c_p, c_p, c_i, c_p, c_p, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_f, c_f, c_f,
                                          c_i, c_i, c_p]),
    "syn3r_feedforward_fused_ln_f16": (c_i, [c_p, c_ll, c_p, c_p, c_f, c_p, c_p, c_i, c_p, c_p, c_p, c_ll, c_p, c_ll, c_p, c_ll,
                                             c_f, c_f, c_f, c_i, c_i, c_p]),
    "syn3r_feedforward_fused_addln_f16": (c_i, [c_p, c_ll, c_p, c_i, c_p, c_p, c_f, c_p, c_p, c_i, c_p, c_p, c_p, c_ll, c_p, c_ll,
                                                c_f, c_f, c_f, c_i, c_i, c_p]),
    "syn3r_layernorm_linear320_f16": (c_i, [c_p, c_ll, c_p, c_p, c_f, c_p, c_p, c_ll, c_i, c_i, c_i, c_p]),
    "syn3r_conv2d3x3_f16": (c_i, [c_p, c_p, c_p, c_ll, c_p, c_p, c_ll, c_i, c_p, c_ll, c_f, c_f,
                                  c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "syn3r_tconv3_f16": (c_i, [c_p, c_p, c_p, c_ll, c_p, c_p, c_ll, c_i, c_p, c_ll, c_f, c_f,
                               c_i, c_i, c_i, c_i, c_i, c_p]),
    "syn3r_attention_f16": (c_i, [c_p, c_p, c_p, c_ll, c_p, c_ll, c_i, c_i, c_i, c_p]),
    "syn3r_attention_temporal_f16": (c_i, [c_p, c_p, c_p, c_ll, c_p, c_ll, c_i, c_i, c_i, c_i, c_p]),
    "syn3r_groupnorm_workspace_bytes": (c_sz, [c_i, c_i]),
    "syn3r_groupnorm_f16": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_f, c_i, c_p, c_sz, c_p]),
    "syn3r_groupnorm_2src_f16": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_f, c_i, c_p, c_sz, c_p]),
    "syn3r_groupnorm_pre_f16": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_f, c_i, c_p, c_sz, c_p]),
    "syn3r_gn_partials_bytes": (c_sz, [c_i, c_i]),
    "syn3r_gemm_set_gn_partials": (c_i, [c_p, c_sz]),
    "syn3r_gemm_gn_partials_written": (c_i, []),
    "syn3r_gemm_2src_supported": (c_i, [c_i, c_i, c_i, c_i, c_ll, c_ll]),
    "syn3r_gemm_2src_f16": (c_i, [c_p, c_ll, c_i, c_p, c_ll, c_i, c_p, c_p, c_ll, c_p, c_i, c_i, c_p]),
    "syn3r_layernorm_f16": (c_i, [c_p, c_p, c_p, c_p, c_i, c_ll, c_i, c_p, c_p, c_f, c_p]),
    "syn3r_geglu_f16": (c_i, [c_p, c_p, c_ll, c_i, c_p]),
    "syn3r_softmax_rows_f16": (c_i, [c_p, c_p, c_ll, c_i, c_ll, c_f, c_p]),
    "syn3r_time_conv_out": (c_i, [c_p, c_ll, c_p, c_p, c_p, c_i, c_i, c_ll, c_p]),
    "syn3r_l1_loss_workspace_bytes": (c_sz, [c_ll]),
    "syn3r_l1_loss": (c_i, [c_p, c_p, c_ll, c_f, c_p, c_p, c_sz, c_p]),
    "syn3r_l1_loss_backward": (c_i, [c_p, c_p, c_ll, c_f, c_p, c_p, c_p]),
    "syn3r_image_mse": (c_i, [c_p, c_p, c_ll, c_p, c_p, c_sz, c_p]),
    "syn3r_photo_loss_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "syn3r_photo_loss": (c_i, [c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_p, c_p, c_sz, c_p]),
    "syn3r_photo_loss_backward": (c_i, [c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_p, c_p, c_p, c_p]),
    "syn3r_photo_loss_step": (c_i, [c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "syn3r_adam_step": (c_i, [c_p, c_p, c_p, c_p, c_ll, c_f, c_f, c_f, c_f, c_i, c_p]),
    "syn3r_adam_step_multi": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p]),
    "syn3r_knn3_workspace_bytes": (c_sz, [c_i]),
    "syn3r_knn3_mean_dist2": (c_i, [c_p, c_i, c_p, c_p, c_sz, c_p]),
    "syn3r_conv2d3x3_act_f16": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "syn3r_lpips_image_f16": (c_i, [c_p, c_i, c_i, c_p, c_p]),
    "syn3r_lpips_image_bwd": (c_i, [c_p, c_i, c_i, c_f, c_p, c_p]),
    "syn3r_maxpool2_f16": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "syn3r_maxpool2_bwd_f16": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "syn3r_lpips_layer_workspace_bytes": (c_sz, [c_ll, c_i]),
    "syn3r_lpips_layer_f16": (c_i, [c_p, c_p, c_p, c_ll, c_i, c_i, c_p, c_p, c_sz, c_p]),
    "syn3r_lpips_layer_bwd_f16": (c_i, [c_p, c_p, c_p, c_ll, c_i, c_f, c_i, c_p, c_p]),
    "syn3r_pcd_outlier_workspace_bytes": (c_sz, [c_i]),
    "syn3r_pcd_statistical_outlier": (c_i, [c_p, c_i, c_i, c_d, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "syn3r_flow_cycle_mask": (c_i, [c_p, c_p, c_i, c_i, c_i, c_f, c_p, c_p, c_p]),
    "syn3r_gemm_set_splitk_workspace": (c_i, [c_p, c_sz]),
    "syn3r_unet_create": (c_i, [C.c_char_p, C.c_char_p, C.POINTER(c_p)]),
    "syn3r_unet_destroy": (c_i, [c_p]),
    "syn3r_unet_workspace_bytes": (c_sz, [c_p, c_i, c_i, c_i, c_i, c_i]),
    "syn3r_unet_forward": (c_i, [c_p, c_p, c_d, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
}


class Syn3rError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def set_library_path(path) -> None:
    """Developer A/B runs against another BUILD of the same library: an EXPLICIT call (tools/_devlib.py, the `--syn3r-lib`
    option of tests/conftest.py), before the first load().  The package itself never looks at the environment for this."""
    global _LIB_PATH
    if _lib is not None:
        raise Syn3rError("set_library_path() after the library was loaded")
    _LIB_PATH = Path(path).resolve()


def load():
    """Load the shared library (once) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise Syn3rError(
            f"{_LIB_PATH} is missing: build it with `python -m syn3r_amd.build` "
            "(there is no CPU fallback for the SYN3R hot path)")
    lib = C.CDLL(str(_LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().syn3r_last_error().decode("utf-8", "replace")
        raise Syn3rError(f"{what} failed (code {rc}): {msg}")


def require_gpu(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise Syn3rError("syn3r_amd HIP path needs tensors on a HIP device (no CPU fallback); "
                             f"got a tensor on {t.device}")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise Syn3rError(f"tensors on different devices: {dev} vs {t.device}")
    if dev is None:
        raise Syn3rError("no tensors given")
    return dev


def ptr(t: torch.Tensor | None) -> int | None:
    if t is None:
        return None
    if not t.is_contiguous():
        raise Syn3rError("non-contiguous tensor passed to the C-ABI")
    return t.data_ptr()


def stream_ptr(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def dtype_tag(t: torch.Tensor) -> int:
    if t.dtype == torch.float16:
        return F16
    if t.dtype == torch.float32:
        return F32
    raise Syn3rError(f"unsupported dtype {t.dtype}")


def host_f32(values) -> "C.Array":
    flat = [float(v) for v in values]
    return (C.c_float * len(flat))(*flat)


def host_f64(values) -> "C.Array":
    flat = [float(v) for v in values]
    return (C.c_double * len(flat))(*flat)


_ws_cache: dict = {}


def workspace(dev: torch.device, nbytes: int, tag: str = "") -> torch.Tensor:
    """A reusable per-(device, stream, tag) scratch buffer of at least nbytes."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
        _ws_cache[key] = buf
    return buf


_active_trace = None     # the session of the running `kernel_trace` block (a pointer), for threads that want to join it


def join_active_trace() -> None:
    """Called at the top of autograd `backward` functions (they run on PyTorch's autograd thread): record this thread's
    launches into the trace session a `kernel_trace` block opened on another thread - or detach when none is open.  The
    library itself keeps no process-wide trace switch (include/syn3r_hip.h)."""
    load().syn3r_trace_attach(_active_trace)


class kernel_trace:
    """Context manager: per-kernel HIP-event timing of everything launched inside.
    `.result` maps kernel name -> (calls, total_ms)."""

    def __init__(self, only: str = "", detail: bool = False):
        self.only = only
        self.detail = detail    # contraction launches are reported per shape

    def __enter__(self):
        global _active_trace
        lib = load()
        lib.syn3r_trace_filter(self.only.encode())
        lib.syn3r_trace_enable(2 if self.detail else 1)
        _active_trace = lib.syn3r_trace_session()       # other threads join with join_active_trace()
        self.result = {}
        return self

    def __exit__(self, *exc):
        global _active_trace
        lib = load()
        _active_trace = None
        lib.syn3r_trace_enable(0)
        buf = C.create_string_buffer(1 << 18)
        lib.syn3r_trace_report(buf, len(buf))
        for line in buf.value.decode().splitlines():
            name, calls, ms = line.rsplit(" ", 2)
            self.result[name] = (int(calls), float(ms))
        return False
