"""PyTorch-facing rasterizer API: the drop-in for the reference's `diff_gaussian_rasterization` module.

Mirrors (names, argument meaning, error behaviour) what the reference imports and calls at
gaussian_splatting/gaussian_renderer/__init__.py:14,38-53,87-95 (render) and :124-139,167-175 (render_simple):

    GaussianRasterizationSettings   NamedTuple of the per-frame parameters
    GaussianRasterizer              nn.Module: .forward(means3D, means2D, opacities, shs=None, colors_precomp=None,
                                    scales=None, rotations=None, cov3D_precomp=None) -> (color[3,H,W], radii[P]);
                                    .markVisible(positions) -> bool[P]
    rasterize_gaussians             functional form, positional arguments as upstream
    _RasterizeGaussians             the autograd.Function; backward returns gradients in INPUT order
                                    (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds, None)

and the three native entry points of upstream's `_C`, here thin wrappers over the C ABI
(include/ggd_raster.h) that pass raw device pointers and the current HIP stream:

    rasterize_gaussians_native / rasterize_gaussians_backward_native / mark_visible

All compute happens in libggd_raster.so (hand-written gfx950 kernels).  Tensors must be CUDA(HIP) fp32; there
is no CPU path -- a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import functools
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _capi


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    raw_attributes: bool = False  # extension (default = upstream behaviour): opacities / scales / rotations are the raw
    #                               decoder outputs; sigmoid / exp / normalize are fused into the kernels (fwd + bwd)


def _f32c(t: torch.Tensor, name: str, device) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device} (all rasterizer inputs must share a device)")
    if t.dtype != torch.float32:
        # upstream's extension reads every tensor as float32; a silent cast would also make the fp32 gradients of the
        # backward mismatch the leaf's dtype
        raise TypeError(f"{name} must be float32 (got {t.dtype})")
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if (t is None or t.numel() == 0) else C.c_void_p(t.data_ptr())


@functools.lru_cache(maxsize=4096)
def _sizes(kind: str, a: int, b: int) -> int:
    """ggd_geom_bytes(P) / ggd_img_bytes(W, H) / ggd_binning_bytes(R): pure functions of their arguments, memoised."""
    lib = _capi.load()
    return int(lib.ggd_geom_bytes(a) if kind == "g" else (lib.ggd_img_bytes(a, b) if kind == "i" else lib.ggd_binning_bytes(a)))


_HINT_DECAY = 0.98      # per frame: a one-off large frame is forgotten after ~100 frames (0.98^100 = 0.13)


def _capacity(hint: int) -> int:
    """Instances the binning buffer of a single-call forward is laid out for: 25 % + 64 k above the hint, rounded UP to the
    next value of a geometric grid (2^(k/4)) so that the sizes repeat and the caching allocator reuses its blocks."""
    want = int(hint * 1.25) + 65536
    k = max(0, (want - 1).bit_length() - 1)            # 2^k <= want - 1 < 2^(k+1)
    for q in (1.0, 1.189207115, 1.414213562, 1.681792831, 2.0):
        cap = int(q * (1 << k)) + 1
        if cap >= want:
            return cap
    return want


class _NoGuard:
    def __enter__(self): return None
    def __exit__(self, *a): return False


_NO_GUARD = _NoGuard()


def _device_guard(dev):
    """torch.cuda.device(dev) only when another device is current (the context manager costs ~5 us per call)."""
    idx = dev.index
    return _NO_GUARD if (idx is None or idx == torch.cuda.current_device()) else torch.cuda.device(dev)


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _params(rs: GaussianRasterizationSettings, P: int, M: int, device, keep: list) -> _capi.Params:
    view = _f32c(rs.viewmatrix, "viewmatrix", device)
    proj = _f32c(rs.projmatrix, "projmatrix", device)
    campos = _f32c(rs.campos, "campos", device)
    bg = _f32c(rs.bg, "bg", device)
    if view.numel() != 16 or proj.numel() != 16 or campos.numel() != 3 or bg.numel() != 3:
        raise ValueError("viewmatrix/projmatrix must have 16 elements, campos/bg 3")
    keep += [view, proj, campos, bg]
    return _capi.Params(P, M, int(rs.sh_degree), int(rs.image_width), int(rs.image_height), float(rs.tanfovx),
                        float(rs.tanfovy), float(rs.scale_modifier), int(bool(rs.prefiltered)),
                        int(bool(rs.debug)), view.data_ptr(), proj.data_ptr(), campos.data_ptr(), bg.data_ptr(),
                        int(bool(getattr(rs, "raw_attributes", False))), 0)


def _require_cuda(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("gaussian_gan_decoder_amd rasterizer needs HIP device tensors (got a CPU tensor); "
                           "there is no CPU fallback")


def _marshal_forward(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                     projmatrix, tanfovx, tanfovy, image_height, image_width, sh, degree, campos, prefiltered, debug,
                     raw_attributes=False):
    """Validation and marshalling of one forward call, shared by `rasterize_gaussians_native` and `FramePipeline.submit` (a
    mis-shaped input must become a ValueError on both paths, not an out-of-bounds device read).  Returns (dev, P, the seven
    contiguous fp32 input tensors or None, prm, keep): `keep` holds every tensor the C call reads, for as long as it may."""
    _require_cuda(means3D)
    dev = means3D.device
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise ValueError("means3D must have dimensions (num_points, 3)")
    P = means3D.size(0)
    means3D = _f32c(means3D, "means3D", dev)
    opacities = _f32c(opacities, "opacities", dev)
    if opacities.numel() != P:
        raise ValueError("opacities must hold one value per point")
    have = lambda t: t is not None and t.numel() > 0
    sh_c = _f32c(sh, "sh", dev) if have(sh) else None
    col_c = _f32c(colors_precomp, "colors_precomp", dev) if have(colors_precomp) else None
    sc_c = _f32c(scales, "scales", dev) if have(scales) else None
    rot_c = _f32c(rotations, "rotations", dev) if have(rotations) else None
    cov_c = _f32c(cov3D_precomp, "cov3D_precomp", dev) if have(cov3D_precomp) else None
    for t, n, what in ((col_c, 3, "colors_precomp"), (sc_c, 3, "scales"), (rot_c, 4, "rotations"), (cov_c, 6, "cov3D_precomp")):
        if t is not None and t.numel() != P * n:
            raise ValueError(f"{what} must have dimensions (num_points, {n})")
    M = 0
    if sh_c is not None:
        if sh_c.dim() != 3 or sh_c.size(0) != P or sh_c.size(2) != 3:
            raise ValueError("sh must have dimensions (num_points, num_coeffs, 3)")
        M = sh_c.size(1)
    rs = GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier,
                                       viewmatrix, projmatrix, degree, campos, prefiltered, debug, raw_attributes)
    keep: list = [means3D, opacities, sh_c, col_c, sc_c, rot_c, cov_c]
    prm = _params(rs, P, M, dev, keep)
    return dev, P, (means3D, opacities, sh_c, col_c, sc_c, rot_c, cov_c), prm, keep


def rasterize_gaussians_native(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier,
                               cov3D_precomp, viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width,
                               sh, degree, campos, prefiltered, debug, raw_attributes=False):
    """== upstream `_C.rasterize_gaussians(...)`: returns
    (num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer).  binningBuffer may be larger than
    num_rendered needs (single-call forward with a capacity hint); the sorted list sits at its offset 0 either way, so
    the backward takes num_rendered as R exactly like upstream."""
    dev, P, (means3D, opacities, sh_c, col_c, sc_c, rot_c, cov_c), prm, keep = _marshal_forward(
        bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
        tanfovx, tanfovy, image_height, image_width, sh, degree, campos, prefiltered, debug, raw_attributes)
    # (the GPU idles while this wrapper runs between two frames of a render loop: sizes are cached, the stream is looked
    # up once, and the device guard is only entered when another device is current)
    ctx, stream_handle = _capi.context_and_stream(dev)
    lib = ctx.lib
    H, W = int(image_height), int(image_width)
    u8 = dict(dtype=torch.uint8, device=dev)
    color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    geom = torch.empty((_sizes("g", P, 0),), **u8)
    img = torch.empty((_sizes("i", W, H),), **u8)
    R = C.c_int64(0)
    stream = C.c_void_p(stream_handle)
    key = (P, W, H)
    hint = ctx.capacity_hint.get(key)
    with _device_guard(dev):
        binning = None
        if hint is not None and P > 0 and not debug:   # debug: exact two-call form, buffers laid out for num_rendered
            # single-call forward: binning buffer sized from the recent frames of this shape (see _capacity); the GPU does
            # not wait for the num_rendered round trip.  Falls through to the exact two-phase path on overflow.
            cap = _capacity(hint)
            binning = torch.empty((_sizes("b", cap, 0),), **u8)
            rc = lib.ggd_forward(ctx.handle, stream, C.byref(prm), _ptr(means3D), _ptr(sh_c), _ptr(col_c),
                                 _ptr(opacities), _ptr(sc_c), _ptr(rot_c), _ptr(cov_c), _ptr(geom), _ptr(radii),
                                 _ptr(binning), cap, _ptr(img), _ptr(color), C.byref(R))
            if rc == -6:       # GGD_E_CAPACITY: R is valid, redo the render phase with an exact buffer
                binning = None
                ctx.capacity_retries += 1
            else:
                ctx.check(rc)
        else:
            ctx.check(lib.ggd_forward_geometry(ctx.handle, stream, C.byref(prm), _ptr(means3D), _ptr(sh_c),
                                               _ptr(col_c), _ptr(opacities), _ptr(sc_c), _ptr(rot_c), _ptr(cov_c),
                                               _ptr(geom), _ptr(radii), C.byref(R)))
        if binning is None:
            binning = torch.empty((lib.ggd_binning_bytes(R.value),), **u8)
            ctx.check(lib.ggd_forward_render(ctx.handle, stream, C.byref(prm), _ptr(geom), R.value, _ptr(binning),
                                             _ptr(img), _ptr(color)))
    # the hint is a DECAYING RUNNING MAXIMUM of num_rendered, not the last frame's value: consecutive scenes of a training
    # step differ several-fold (the reference draws fov ~ U[5, 17] degrees per scene, target_dataloader.py:71), and with
    # "last frame x 1.25" every upward swing took the overflow -> exact-retry path, i.e. a host sync
    ctx.capacity_hint[key] = max(int(R.value), int((hint or 0) * _HINT_DECAY))
    return int(R.value), color, radii, geom, binning, img


class FramePipeline:
    """Several forward frames in flight on one GPU: `slots` HIP streams, each with its own rasterizer context, used round-robin.
    `submit(...)` takes the arguments of `rasterize_gaussians_native`, launches the whole frame on the next slot's stream with
    `ggd_forward_enqueue` and returns WITHOUT waiting for its num_rendered -- the latency-bound front of that frame (per-Gaussian
    kernel, depth sort, binning: about one workgroup per CU) then runs under the blend of the frame before it.  What it returns
    is the result of the frame that used the slot last (None while the pipeline fills): the tuple of
    `rasterize_gaussians_native` plus a `torch.cuda.Event` recorded behind that frame -- make the consuming stream
    `wait_event` it (or synchronise) before reading the tensors.  `drain()` returns the results still pending, oldest first.

    The slot's stream first waits for the stream that is current at `submit`, so inputs produced there are safe to use; every
    input tensor is `record_stream`ed on the slot's stream (the frame's blend may still be reading `bg` when its num_rendered is
    collected and the references are dropped), so per-frame temporaries are safe to pass.  The OUTPUTS are
    allocated on the slot's stream: a consumer on another stream must `wait_event` the returned event and, if it lets go of
    the tensors before its own work on them has finished, `record_stream` them on its stream.  A
    frame for which no capacity hint exists yet (first frame of a shape), a `debug` frame, or one whose binning overflowed
    its capacity is rendered through the ordinary synchronous path on the slot's stream; results are identical either way."""

    def __init__(self, device, slots: int = 2):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("FramePipeline needs a HIP device (there is no CPU fallback)")
        self.slots = [dict(stream=torch.cuda.Stream(device=self.device), pending=None) for _ in range(max(1, int(slots)))]
        self._next = 0
        self.synchronous_frames = 0     # frames that took the ordinary path (no hint yet / overflow / unsupported)

    def _collect(self, slot):
        pend = slot["pending"]
        if pend is None:
            return None
        slot["pending"] = None
        # (the collected frame's binning / blend may still be running -- ggd_forward_collect returns with num_rendered; its inputs
        # were record_stream'ed on this slot's stream at submit, so dropping the references here is safe)
        with torch.cuda.stream(slot["stream"]):
            if "result" in pend:                      # rendered synchronously at submit time
                res = pend["result"]
            else:
                ctx, handle = _capi.context_and_stream(self.device)
                R = C.c_int64(0)
                rc = ctx.lib.ggd_forward_collect(ctx.handle, C.c_void_p(handle), C.byref(R))
                if rc == -6:                          # GGD_E_CAPACITY: this frame's outputs are not valid -- render it again
                    ctx.capacity_retries += 1
                    ctx.capacity_hint[pend["key"]] = max(int(R.value), ctx.capacity_hint.get(pend["key"]) or 0)
                    self.synchronous_frames += 1
                    res = rasterize_gaussians_native(*pend["args"])
                else:
                    ctx.check(rc)
                    hint = ctx.capacity_hint.get(pend["key"]) or 0
                    ctx.capacity_hint[pend["key"]] = max(int(R.value), int(hint * _HINT_DECAY))
                    res = (int(R.value),) + pend["outputs"]
            ev = torch.cuda.Event()
            ev.record(slot["stream"])
        return res + (ev,)

    def submit(self, bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
               projmatrix, tanfovx, tanfovy, image_height, image_width, sh, degree, campos, prefiltered, debug,
               raw_attributes=False):
        args = (bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
                tanfovx, tanfovy, image_height, image_width, sh, degree, campos, prefiltered, debug, raw_attributes)
        slot = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        prev = self._collect(slot)
        # (validation and marshalling on the CALLER's stream: a conversion to contiguous fp32 is the caller's work)
        dev, P, (means3D_c, opac_c, sh_c, col_c, sc_c, rot_c, cov_c), prm, keep = _marshal_forward(*args)
        slot["stream"].wait_stream(torch.cuda.current_stream(dev))
        for t in keep:                            # allocated on the caller's stream, read on the slot's: the caching allocator
            if t is not None:                     # must not hand the block out again before the slot's work at the time of the
                t.record_stream(slot["stream"])   # free has run (the blend reads bg long after num_rendered has been collected)
        with torch.cuda.stream(slot["stream"]):
            H, W = int(image_height), int(image_width)
            ctx, handle = _capi.context_and_stream(dev)
            key = (P, W, H)
            hint = ctx.capacity_hint.get(key)
            if hint is None or P == 0 or debug:
                self.synchronous_frames += 1
                slot["pending"] = dict(result=rasterize_gaussians_native(*args), keep=keep)
                return prev
            cap = _capacity(hint)
            lib = ctx.lib
            if not lib.ggd_forward_can_speculate(ctx.handle, C.byref(prm), cap):
                self.synchronous_frames += 1
                slot["pending"] = dict(result=rasterize_gaussians_native(*args), keep=keep)
                return prev
            u8 = dict(dtype=torch.uint8, device=dev)
            color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
            geom = torch.empty((_sizes("g", P, 0),), **u8)
            img = torch.empty((_sizes("i", W, H),), **u8)
            binning = torch.empty((_sizes("b", cap, 0),), **u8)
            with _device_guard(dev):
                ctx.check(lib.ggd_forward_enqueue(ctx.handle, C.c_void_p(handle), C.byref(prm), _ptr(means3D_c), _ptr(sh_c),
                                                  _ptr(col_c), _ptr(opac_c), _ptr(sc_c), _ptr(rot_c), _ptr(cov_c), _ptr(geom),
                                                  _ptr(radii), _ptr(binning), cap, _ptr(img), _ptr(color)))
            slot["pending"] = dict(outputs=(color, radii, geom, binning, img), key=key, args=args, keep=keep, prm=prm)
        return prev

    def drain(self):
        out = []
        n = len(self.slots)
        for k in range(n):                      # oldest first: the slot that would be used next holds the oldest frame
            res = self._collect(self.slots[(self._next + k) % n])
            if res is not None:
                out.append(res)
        return out


def rasterize_gaussians_backward_native(bg, means3D, radii, colors_precomp, scales, rotations, scale_modifier,
                                        cov3D_precomp, viewmatrix, projmatrix, tanfovx, tanfovy, dL_dout_color, sh,
                                        degree, campos, geomBuffer, R, binningBuffer, imgBuffer, debug,
                                        raw_attributes=False, opacities=None):
    """== upstream `_C.rasterize_gaussians_backward(...)`: returns
    (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)."""
    _require_cuda(means3D)
    dev = means3D.device
    P = means3D.size(0)
    H, W = int(dL_dout_color.size(-2)), int(dL_dout_color.size(-1))
    means3D = _f32c(means3D, "means3D", dev)
    have = lambda t: t is not None and t.numel() > 0
    sh_c = _f32c(sh, "sh", dev) if have(sh) else None
    col_c = _f32c(colors_precomp, "colors_precomp", dev) if have(colors_precomp) else None
    sc_c = _f32c(scales, "scales", dev) if have(scales) else None
    rot_c = _f32c(rotations, "rotations", dev) if have(rotations) else None
    cov_c = _f32c(cov3D_precomp, "cov3D_precomp", dev) if have(cov3D_precomp) else None
    g = _f32c(dL_dout_color, "dL_dout_color", dev)
    M = sh_c.size(1) if sh_c is not None else 0
    rs = GaussianRasterizationSettings(H, W, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix, degree,
                                       campos, False, debug, raw_attributes)
    op_c = _f32c(opacities, "opacities", dev) if (raw_attributes and opacities is not None) else None
    keep: list = []
    prm = _params(rs, P, M, dev, keep)
    ctx = _capi.context_for(dev)
    lib = ctx.lib
    f32 = dict(dtype=torch.float32, device=dev)
    dL_dmeans3D = torch.empty((P, 3), **f32)
    dL_dmeans2D = torch.empty((P, 3), **f32)
    dL_dcolors = torch.empty((P, 3), **f32)
    dL_dopacity = torch.empty((P, 1), **f32)
    dL_dcov3D = torch.empty((P, 6), **f32)
    dL_dsh = torch.empty((P, M, 3), **f32)
    dL_dscales = torch.empty((P, 3), **f32)
    dL_drotations = torch.empty((P, 4), **f32)
    if ctx.poison_outputs:   # tests: the library must write every element itself (it does not rely on pre-zeroed arrays)
        for t in (dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations):
            t.fill_(float("nan"))
    if P > 0:
        with torch.cuda.device(dev):
            ctx.check(lib.ggd_backward(ctx.handle, _stream(dev), C.byref(prm), _ptr(means3D), _ptr(sh_c), _ptr(col_c),
                                       _ptr(op_c), _ptr(sc_c), _ptr(rot_c), _ptr(cov_c), _ptr(radii), _ptr(geomBuffer),
                                       _ptr(binningBuffer), _ptr(imgBuffer), int(R), _ptr(g), _ptr(dL_dmeans2D),
                                       _ptr(dL_dcolors), _ptr(dL_dopacity), _ptr(dL_dmeans3D), _ptr(dL_dcov3D),
                                       _ptr(dL_dsh), _ptr(dL_dscales), _ptr(dL_drotations)))
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def mark_visible(positions, viewmatrix, projmatrix):
    """== upstream `_C.mark_visible`: bool[P], True where the point passes the z > 0.2 frustum test."""
    _require_cuda(positions)
    dev = positions.device
    pos = _f32c(positions, "positions", dev)
    view = _f32c(viewmatrix, "viewmatrix", dev)
    proj = _f32c(projmatrix, "projmatrix", dev)
    P = pos.size(0)
    out = torch.empty((P,), dtype=torch.uint8, device=dev)
    ctx = _capi.context_for(dev)
    with torch.cuda.device(dev):
        ctx.check(ctx.lib.ggd_mark_visible(ctx.handle, _stream(dev), P, _ptr(pos), _ptr(view), _ptr(proj), _ptr(out)))
    return out.bool()


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        num_rendered, color, radii, geom, binning, img = rasterize_gaussians_native(
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree,
            rs.campos, rs.prefiltered, rs.debug, getattr(rs, "raw_attributes", False))
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning,
                              img, opacities)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img, opacities = \
            ctx.saved_tensors
        raw = getattr(rs, "raw_attributes", False)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = rasterize_gaussians_backward_native(
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, rs.campos, geom,
            ctx.num_rendered, binning, img, rs.debug, raw, opacities if raw else None)
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales,
                grad_rotations, grad_cov3Ds_precomp, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = torch.empty(0, dtype=torch.float32, device=means3D.device)
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, rs)
