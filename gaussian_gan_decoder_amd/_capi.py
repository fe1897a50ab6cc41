"""ctypes binding of libggd_raster.so (C ABI: include/ggd_raster.h).

Raw device pointers + the current HIP stream handle cross the boundary; torch only provides memory and streams.
The library is required: if it is missing or fails to load, importing the rasterizer raises -- there is NO CPU
or eager fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
# GGD_LIB_PATH: load another build of the library (A/B timing of kernel variants inside one process launch)
LIB_PATH = os.environ.get("GGD_LIB_PATH") or os.path.join(_PKG, "libggd_raster.so")

EXPORTS = [
    "ggd_geom_bytes", "ggd_binning_bytes", "ggd_img_bytes", "ggd_geom_layout", "ggd_binning_layout",
    "ggd_img_layout", "ggd_sort_bits", "ggd_create", "ggd_destroy", "ggd_last_error", "ggd_version",
    "ggd_forward_geometry", "ggd_forward_render", "ggd_forward", "ggd_forward_enqueue", "ggd_forward_collect", "ggd_forward_can_speculate", "ggd_backward", "ggd_mark_visible", "ggd_debug_unsorted",
    "ggd_triplane_forward", "ggd_triplane_backward", "ggd_trigrid_forward", "ggd_trigrid_backward", "ggd_planes_gather", "ggd_planes_scatter", "ggd_surface_tmp_bytes", "ggd_surface_sample", "ggd_attrs_split", "ggd_attrs_merge", "ggd_decoder_packed_bytes", "ggd_decoder_pack", "ggd_decoder_forward", "ggd_decoder_zbuf_bytes", "ggd_decoder_packed_t_bytes",
    "ggd_decoder_forward_train", "ggd_decoder_backward", "ggd_decoder_wgrad_floats", "ggd_decoder_wgrad", "ggd_decoder_backward_wgrad", "ggd_decoder_packed_hl_bytes", "ggd_decoder_packed_t_hl_bytes", "ggd_decoder_dzbuf_hl_bytes", "ggd_decoder_pack_hl", "ggd_decoder_forward_hl", "ggd_decoder_backward_wgrad_hl", "ggd_image_loss_tmp_bytes", "ggd_image_loss", "ggd_set_option", "ggd_get_option", "ggd_blend_stats", "ggd_blend_backward_stats", "ggd_blend_timeline", "ggd_set_profiling", "ggd_stage_count", "ggd_stage_name", "ggd_stage_times",
]


class Params(C.Structure):
    _fields_ = [("P", C.c_int32), ("M", C.c_int32), ("sh_degree", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("scale_modifier", C.c_float), ("prefiltered", C.c_int32), ("debug", C.c_int32),
                ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
                ("bg", C.c_void_p), ("raw_attributes", C.c_int32), ("reserved_", C.c_int32)]


class GeomView(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("splat", "tiles_touched", "point_offsets", "clamped", "depth_keys", "rect", "header", "total")]


class BinningView(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("keys", "list", "keys_alt", "list_alt", "total")]


class ImgView(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("ranges", "final_T", "n_contrib", "total")]


OPT_EXP_MODE, OPT_BLEND_CULL, OPT_BINNING, OPT_BLEND_SPLIT, OPT_FOLD, OPT_MSD_SORT = 0, 1, 2, 3, 4, 5
STAT_FLAT_STREAK, STAT_SORT_RERUNS, STAT_MSD_FRAMES = 100, 101, 102   # read-only, through get_option
SPLAT_BYTES = 48
SPLAT_FIELDS = ("x", "y", "hA", "nB", "hC", "thr", "opacity", "r", "g", "b", "ex", "ey")

_lib = None
_lock = threading.Lock()


def load():
    """Load the shared library (after torch, so both share one HIP runtime).  Raises if it is not built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m gaussian_gan_decoder_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the rasterizer.")
        import torch  # noqa: F401  (loads torch's libamdhip64 first; our NEEDED entry then binds to the same runtime)
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
        lib.ggd_geom_bytes.restype = sz; lib.ggd_geom_bytes.argtypes = [i32]
        lib.ggd_binning_bytes.restype = sz; lib.ggd_binning_bytes.argtypes = [i64]
        lib.ggd_img_bytes.restype = sz; lib.ggd_img_bytes.argtypes = [i32, i32]
        lib.ggd_geom_layout.argtypes = [i32, C.POINTER(GeomView)]
        lib.ggd_binning_layout.argtypes = [i64, C.POINTER(BinningView)]
        lib.ggd_img_layout.argtypes = [i32, i32, C.POINTER(ImgView)]
        lib.ggd_sort_bits.argtypes = [i32, i32]
        lib.ggd_create.restype = vp; lib.ggd_create.argtypes = [C.c_int]
        lib.ggd_destroy.restype = None; lib.ggd_destroy.argtypes = [vp]
        lib.ggd_last_error.restype = C.c_char_p; lib.ggd_last_error.argtypes = [vp]
        lib.ggd_version.restype = C.c_char_p
        lib.ggd_forward_geometry.argtypes = [vp, vp, C.POINTER(Params)] + [vp] * 7 + [vp, vp, C.POINTER(i64)]
        lib.ggd_forward_render.argtypes = [vp, vp, C.POINTER(Params), vp, i64, vp, vp, vp]
        lib.ggd_forward.argtypes = [vp, vp, C.POINTER(Params)] + [vp] * 7 + [vp, vp, vp, i64, vp, vp, C.POINTER(i64)]
        lib.ggd_forward_enqueue.argtypes = [vp, vp, C.POINTER(Params)] + [vp] * 7 + [vp, vp, vp, i64, vp, vp]
        lib.ggd_forward_collect.argtypes = [vp, vp, C.POINTER(i64)]
        lib.ggd_forward_can_speculate.argtypes = [vp, C.POINTER(Params), i64]
        lib.ggd_backward.argtypes = [vp, vp, C.POINTER(Params)] + [vp] * 7 + [vp, vp, vp, vp, i64, vp] + [vp] * 8
        lib.ggd_mark_visible.argtypes = [vp, vp, i32, vp, vp, vp, vp]
        lib.ggd_debug_unsorted.argtypes = [vp, vp, vp, vp, i64]
        lib.ggd_triplane_forward.argtypes = [vp, vp, vp, i32, i32, i32, vp, i32, C.c_float, vp]
        lib.ggd_triplane_backward.argtypes = [vp, vp, i32, i32, i32, vp, i32, C.c_float, vp, vp]
        lib.ggd_trigrid_forward.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, C.c_float, vp]
        lib.ggd_trigrid_backward.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, i32, C.c_float, vp, vp]
        lib.ggd_planes_gather.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, C.c_float, vp]
        lib.ggd_planes_scatter.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, C.c_float, vp, vp, i32]
        lib.ggd_surface_tmp_bytes.restype = sz; lib.ggd_surface_tmp_bytes.argtypes = [i32]
        lib.ggd_surface_sample.argtypes = [vp, vp, vp, i32, C.c_float, i32, C.c_float, C.c_uint64, vp, vp, vp, sz]
        lib.ggd_attrs_split.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, vp]
        lib.ggd_attrs_merge.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, vp]
        lib.ggd_decoder_packed_bytes.restype = sz
        lib.ggd_decoder_pack.argtypes = [vp, vp, C.POINTER(C.c_void_p), vp, vp]
        lib.ggd_decoder_forward.argtypes = [vp, vp, vp, vp, i32, vp, vp]
        lib.ggd_decoder_zbuf_bytes.restype = sz; lib.ggd_decoder_zbuf_bytes.argtypes = [i32]
        lib.ggd_decoder_packed_t_bytes.restype = sz
        lib.ggd_decoder_forward_train.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp]
        lib.ggd_decoder_backward.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.ggd_decoder_wgrad_floats.restype = sz
        lib.ggd_decoder_wgrad.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
        lib.ggd_decoder_backward_wgrad.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.ggd_decoder_packed_hl_bytes.restype = sz
        lib.ggd_decoder_packed_t_hl_bytes.restype = sz
        lib.ggd_decoder_dzbuf_hl_bytes.restype = sz; lib.ggd_decoder_dzbuf_hl_bytes.argtypes = [i32]
        lib.ggd_decoder_pack_hl.argtypes = [vp, vp, C.POINTER(C.c_void_p), vp, vp]
        lib.ggd_decoder_forward_hl.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp]
        lib.ggd_decoder_backward_wgrad_hl.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.ggd_image_loss_tmp_bytes.restype = sz; lib.ggd_image_loss_tmp_bytes.argtypes = [i32, i32]
        lib.ggd_image_loss.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, sz]
        lib.ggd_set_option.argtypes = [vp, C.c_int, C.c_int]
        lib.ggd_get_option.argtypes = [vp, C.c_int]
        lib.ggd_blend_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_ulonglong)]
        lib.ggd_blend_backward_stats.argtypes = [vp, C.POINTER(C.c_ulonglong)]
        lib.ggd_blend_timeline.argtypes = [vp, C.POINTER(C.c_ulonglong), C.c_int]
        lib.ggd_set_profiling.argtypes = [vp, C.c_int]
        lib.ggd_stage_name.restype = C.c_char_p; lib.ggd_stage_name.argtypes = [C.c_int]
        lib.ggd_stage_times.argtypes = [vp, C.POINTER(C.c_float)]
        _lib = lib
        return lib


class RasterError(RuntimeError):
    pass


class Context:
    """One ggd_ctx (device workspace); not re-entrant, use one per (device, stream)."""
    capacity_hint: dict  # (P, W, H) -> decaying running maximum of num_rendered: sizes the binning buffer of the next
    #                      single-call forward (rasterizer.rasterize_gaussians_native)
    capacity_retries: int  # single-call forwards whose buffer was too small and that were redone in the exact two-call form

    def __init__(self, device_index: int):
        self.capacity_hint = {}
        self.capacity_retries = 0
        self.poison_outputs = False   # tests: NaN-fill the backward's output arrays before the call
        self.lib = load()
        self.device_index = int(device_index)
        self.handle = self.lib.ggd_create(self.device_index)
        if not self.handle:
            raise RasterError("ggd_create failed: " + self.lib.ggd_last_error(None).decode())

    def check(self, rc: int):
        if rc != 0:
            raise RasterError(f"ggd error {rc}: " + self.lib.ggd_last_error(self.handle).decode())

    def set_option(self, option: int, value: int):
        self.check(self.lib.ggd_set_option(self.handle, int(option), int(value)))

    def get_option(self, option: int) -> int:
        return int(self.lib.ggd_get_option(self.handle, int(option)))

    def blend_stats(self, enable):
        """Start (True / 1: work counters, 2: per-wave timeline) or stop (False) the forward-blend statistics; returns the
        counters gathered since the last start."""
        out = (C.c_ulonglong * 6)()
        self.check(self.lib.ggd_blend_stats(self.handle, int(enable), out))
        return dict(zip(("visited", "culled", "lanes", "pixels", "listed", "culled_in_loop"), [int(v) for v in out]))

    def blend_backward_stats(self):
        """Work counters of the backward blends (quarter form) since blend_stats(True); call before blend_stats(False)."""
        out = (C.c_ulonglong * 8)()
        self.check(self.lib.ggd_blend_backward_stats(self.handle, out))
        return dict(zip(("walked", "staged", "needed", "blended", "live_lanes", "atomic_spans", "rounds", "waves"), [int(v) for v in out]))

    def blend_timeline(self, waves: int):
        """[waves, 4] int64 array: start tick, end tick (100 MHz), list length, entries gathered -- of the forward blend that
        ran after blend_stats(2)."""
        import numpy as np
        out = (C.c_ulonglong * (3 * waves))()
        self.check(self.lib.ggd_blend_timeline(self.handle, out, int(waves)))
        a = np.frombuffer(out, dtype=np.uint64).reshape(waves, 3)
        return np.stack([a[:, 0], a[:, 1], a[:, 2] >> np.uint64(32), a[:, 2] & np.uint64(0xFFFFFFFF)], 1).astype(np.int64)

    def set_profiling(self, on: bool):
        self.check(self.lib.ggd_set_profiling(self.handle, int(bool(on))))

    def stage_times(self) -> dict:
        n = self.lib.ggd_stage_count()
        arr = (C.c_float * n)()
        self.check(self.lib.ggd_stage_times(self.handle, arr))
        return {self.lib.ggd_stage_name(i).decode(): float(arr[i]) for i in range(n) if arr[i] >= 0}

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ggd_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# The reference's Python API has no context argument (`rasterize_gaussians(...)`, `GaussianRasterizer(settings)(...)`): the
# library's contexts -- which hold ALL of its state: scratch, control words, options, capacity hints -- are created on
# first use per (device, HIP stream) and looked up here.  The C library itself has no global state.
_contexts: dict = {}


def context_for(device) -> Context:
    """Per-(device, stream) context cache."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, int(torch.cuda.current_stream(idx).cuda_stream))
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = _contexts[key] = Context(idx)
    return ctx


def context_and_stream(device):
    """(context of the current stream of `device`, that stream's raw handle) with one current_stream lookup."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    handle = int(torch.cuda.current_stream(idx).cuda_stream)
    key = (idx, handle)
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = _contexts[key] = Context(idx)
    return ctx, handle


def geom_view(P: int) -> GeomView:
    v = GeomView(); load().ggd_geom_layout(P, C.byref(v)); return v


def binning_view(R: int) -> BinningView:
    v = BinningView(); load().ggd_binning_layout(R, C.byref(v)); return v


def img_view(W: int, H: int) -> ImgView:
    v = ImgView(); load().ggd_img_layout(W, H, C.byref(v)); return v
