"""Shared helpers for the parity tests: build identical inputs for the oracle (numpy) and the HIP path (torch)."""
from __future__ import annotations

import math

import numpy as np
import torch

from gaussian_gan_decoder_amd.synthetic import make_scene, make_dL_dpix


def scene_inputs(P, size, kind="cube", seed=0, sh_degree=0, use_colors=False, use_cov=False, lsm=-6.0,
                 fov_deg=12.0, width=None, height=None, scale_modifier=1.0, h=math.pi / 2, v=math.pi / 2, sh_M=None):
    """Returns a dict of CPU torch tensors / scalars describing one rasterizer call.
    sh_M: number of SH coefficients stored per channel (>= (sh_degree+1)^2; default exactly that) -- the reference's
    container always stores (max_sh_degree+1)^2 once the active degree is > 0 (gaussian_model.py:116-120)."""
    sc = make_scene(P, size, kind, seed=seed, log_scale_mean=lsm, fov_deg=fov_deg, h=h, v=v)
    cam = sc.cam
    g = torch.Generator().manual_seed(seed + 1000)
    M = (sh_degree + 1) ** 2 if sh_M is None else int(sh_M)
    assert M >= (sh_degree + 1) ** 2
    shs = torch.cat([sc.features_dc, 0.3 * torch.randn(P, M - 1, 3, generator=g)], dim=1) if M > 1 else sc.features_dc
    W = width or size
    H = height or size
    # proper camera centre (inverse of the view matrix) when SH degree > 0 so that `dir` is meaningful
    campos = torch.inverse(cam.world_view_transform)[3, :3].contiguous() if sh_degree > 0 else cam.camera_center
    d = dict(P=P, W=W, H=H, sh_degree=sh_degree, scale_modifier=scale_modifier,
             tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
             means3D=sc.xyz, opacities=sc.opacities, viewmatrix=cam.world_view_transform.contiguous(),
             projmatrix=cam.full_proj_transform.contiguous(), campos=campos, bg=sc.bg,
             shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None)
    if use_colors:
        d["colors_precomp"] = torch.rand(P, 3, generator=g)
    else:
        d["shs"] = shs.contiguous()
    if use_cov:
        from gaussian_gan_decoder_amd.gaussian_model import build_covariance_from_scaling_rotation
        d["cov3D_precomp"] = build_covariance_from_scaling_rotation(sc.scales, scale_modifier, sc.rotations).contiguous()
    else:
        d["scales"] = sc.scales.contiguous()
        d["rotations"] = sc.rotations.contiguous()
    return d


def adversarial_inputs():
    """Hand-made collisions and degenerate members: exact duplicates (equal depth bits -> order
    by index), points behind / on the near plane, zero and full opacity, splats far larger than the image and smaller
    than a pixel, off-screen centres, a zero quaternion, an oblique camera."""
    d = scene_inputs(P=600, size=96, lsm=-4.0, seed=5, width=96, height=80, h=1.1, v=1.9)
    g = torch.Generator().manual_seed(99)
    xyz, op = d["means3D"].clone(), d["opacities"].clone()
    sc, rot = d["scales"].clone(), d["rotations"].clone()
    view = d["viewmatrix"]                      # row-vector convention: p_view = [p, 1] @ view
    cam_pos = torch.inverse(view)[3, :3]
    fwd = view[:3, 2]                           # world direction of +z_view
    xyz[0:40] = xyz[40:80]; sc[0:40] = sc[40:80]; rot[0:40] = rot[40:80]; op[0:40] = op[40:80]   # exact duplicates
    xyz[80:90] = cam_pos - 0.5 * fwd + 0.05 * torch.randn(10, 3, generator=g)        # behind the camera
    xyz[90] = cam_pos + 0.2 * fwd                                                    # z_view ~ 0.2 (near-plane test)
    xyz[91] = cam_pos + 0.2000001 * fwd
    op[100:110] = 0.0; op[110:120] = 1.0; op[120:125] = 1.0 / 255.0
    sc[130:136] = 3.0                                                                # covers the whole image many times
    sc[140:150] = 1e-7                                                               # only the 0.3 low-pass remains
    xyz[150:160] = xyz[150:160] + torch.tensor([5.0, 0.0, 0.0])                      # far off-screen
    rot[160] = 0.0                                                                   # zero quaternion (degenerate R)
    sc[170:175, 0] = 2.0; sc[170:175, 1:] = 1e-4                                     # needles
    d.update(means3D=xyz.contiguous(), opacities=op.contiguous(), scales=sc.contiguous(), rotations=rot.contiguous())
    return d


def run_oracle(d, dtype=np.float32, stop_after=None):
    from oracle import ggd_oracle as O
    np_ = lambda t: None if t is None else t.numpy()
    return O.forward(means3D=np_(d["means3D"]), opacities=np_(d["opacities"]), shs=np_(d["shs"]),
                     colors_precomp=np_(d["colors_precomp"]), scales=np_(d["scales"]), rotations=np_(d["rotations"]),
                     cov3D_precomp=np_(d["cov3D_precomp"]), viewmatrix=np_(d["viewmatrix"]),
                     projmatrix=np_(d["projmatrix"]), campos=np_(d["campos"]), bg=np_(d["bg"]), W=d["W"], H=d["H"],
                     tanfovx=d["tanfovx"], tanfovy=d["tanfovy"], sh_degree=d["sh_degree"],
                     scale_modifier=d["scale_modifier"], dtype=dtype, stop_after=stop_after)


def run_native(d, device="cuda:0", debug=True, binning=None):
    """Forward through the C ABI (via the _C-style wrapper); returns dict with torch outputs + decoded buffers.
    binning: None = library default (auto), 0 = radix-sort path, 2 = force the tile-binning path."""
    from gaussian_gan_decoder_amd import rasterizer as R, _capi
    dev = torch.device(device)
    _capi.context_for(dev).set_option(_capi.OPT_BINNING, 1 if binning is None else binning)
    t = lambda x: torch.empty(0, device=dev) if x is None else x.to(dev)
    num_rendered, color, radii, geom, binning, img = R.rasterize_gaussians_native(
        t(d["bg"]), t(d["means3D"]), t(d["colors_precomp"]), t(d["opacities"]), t(d["scales"]), t(d["rotations"]),
        d["scale_modifier"], t(d["cov3D_precomp"]), t(d["viewmatrix"]), t(d["projmatrix"]), d["tanfovx"],
        d["tanfovy"], d["H"], d["W"], t(d["shs"]), d["sh_degree"], t(d["campos"]), False, debug)
    out = dict(num_rendered=num_rendered, color=color, radii=radii, geom=geom, binning=binning, img=img)
    out.update(decode_buffers(d["P"], d["W"], d["H"], num_rendered, geom, binning, img))
    if debug and num_rendered > 0:
        ctx = _capi.context_for(dev)
        ku = torch.empty(num_rendered, dtype=torch.int64, device=dev)
        vu = torch.empty(num_rendered, dtype=torch.int32, device=dev)
        import ctypes as C
        ctx.check(ctx.lib.ggd_debug_unsorted(ctx.handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream),
                                             C.c_void_p(ku.data_ptr()), C.c_void_p(vu.data_ptr()), num_rendered))
        torch.cuda.synchronize(dev)
        out["keys_unsorted"] = ku.cpu().numpy().view(np.uint64)
        out["list_unsorted"] = vu.cpu().numpy().view(np.uint32)
    return out


def device_args(d, device="cuda:0", debug=False):
    """The positional arguments of `rasterizer.rasterize_gaussians_native` / `FramePipeline.submit` for the inputs `d`, resident
    on the device (a render loop that does not upload its scene every frame)."""
    dev = torch.device(device)
    t = lambda x: torch.empty(0, device=dev) if x is None else x.to(dev)
    return (t(d["bg"]), t(d["means3D"]), t(d["colors_precomp"]), t(d["opacities"]), t(d["scales"]), t(d["rotations"]),
            d["scale_modifier"], t(d["cov3D_precomp"]), t(d["viewmatrix"]), t(d["projmatrix"]), d["tanfovx"], d["tanfovy"],
            d["H"], d["W"], t(d["shs"]), d["sh_degree"], t(d["campos"]), False, debug)


def decode_result(d, res):
    """(num_rendered, color, radii, geom, binning, img[, event]) of a native forward -> the dict run_native returns."""
    num_rendered, color, radii, geom, binning, img = res[:6]
    out = dict(num_rendered=num_rendered, color=color, radii=radii, geom=geom, binning=binning, img=img)
    out.update(decode_buffers(d["P"], d["W"], d["H"], num_rendered, geom, binning, img))
    return out


def same_frame(a, b):
    """Two native forward results hold the same frame bit for bit: num_rendered, the sorted list, and the whole image buffer
    (ranges, final_T, n_contrib), colour and radii -- compared on the device."""
    from gaussian_gan_decoder_amd import _capi
    R = a[0]
    if not (R == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[4][:4 * R], b[4][:4 * R])):
        return False
    H, W = a[1].shape[-2:]
    iv = _capi.img_view(W, H)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    return all(torch.equal(a[5][o:o + n], b[5][o:o + n])
               for o, n in ((iv.ranges, 8 * T), (iv.final_T, 4 * W * H), (iv.n_contrib, 4 * W * H)))


def decode_buffers(P, W, H, R, geom, binning, img):
    """`keys` is only meaningful when the forward ran with debug=True (radix-sort path, buffer laid out for exactly R);
    the sorted list is at offset 0 of the binning buffer whatever capacity it was allocated for."""
    from gaussian_gan_decoder_amd import _capi
    gv, bv, iv = _capi.geom_view(P), _capi.binning_view(R), _capi.img_view(W, H)
    g = geom.cpu().numpy(); b = binning.cpu().numpy(); im = img.cpu().numpy()
    T = ((W + 15) // 16) * ((H + 15) // 16)
    splat = g[gv.splat:gv.splat + 48 * P].view(np.float32).reshape(P, 12)   # x y hA nB hC thr opacity r g b ex ey
    dk = g[gv.depth_keys:gv.depth_keys + 4 * P].view(np.uint32)
    rect = g[gv.rect:gv.rect + 8 * P].view(np.uint32).reshape(P, 2)
    out = dict(
        xy=splat[:, 0:2],
        # conic (A, B, C) = (-2 hA, -nB, -2 hC): exact rescalings, so bit-comparable with the oracle's conic
        conic_opacity=np.stack([np.float32(-2.0) * splat[:, 2], -splat[:, 3], np.float32(-2.0) * splat[:, 4], splat[:, 6]], 1),
        rgb=splat[:, 7:10], power_threshold=splat[:, 5], cull_extent=splat[:, 10:12],
        depths=np.where(dk == 0xFFFFFFFF, np.uint32(0), dk).view(np.float32),
        rect=np.stack([rect[:, 0] & 0xffff, rect[:, 1] & 0xffff, rect[:, 0] >> 16, rect[:, 1] >> 16], 1).astype(np.int32),
        tiles_touched=g[gv.tiles_touched:gv.tiles_touched + 4 * P].view(np.uint32),
        point_offsets=g[gv.point_offsets:gv.point_offsets + 4 * P].view(np.uint32),
        clamped=g[gv.clamped:gv.clamped + P],
        keys=b[bv.keys:bv.keys + 8 * R].view(np.uint64), point_list=b[bv.list:bv.list + 4 * R].view(np.uint32),
        ranges=im[iv.ranges:iv.ranges + 8 * T].view(np.uint32).reshape(T, 2),
        final_T=im[iv.final_T:iv.final_T + 4 * W * H].view(np.float32).reshape(H, W),
        n_contrib=im[iv.n_contrib:iv.n_contrib + 4 * W * H].view(np.uint32).reshape(H, W))
    return out


def run_native_backward(d, n, dL_dpix, device="cuda:0"):
    from gaussian_gan_decoder_amd import rasterizer as R
    dev = torch.device(device)
    from gaussian_gan_decoder_amd import _capi
    _capi.context_for(dev).poison_outputs = True   # NaN-fill the gradient arrays first: every element must be written by the library
    t = lambda x: torch.empty(0, device=dev) if x is None else x.to(dev)
    outs = R.rasterize_gaussians_backward_native(
        t(d["bg"]), t(d["means3D"]), n["radii"], t(d["colors_precomp"]), t(d["scales"]), t(d["rotations"]),
        d["scale_modifier"], t(d["cov3D_precomp"]), t(d["viewmatrix"]), t(d["projmatrix"]), d["tanfovx"],
        d["tanfovy"], dL_dpix.to(dev), t(d["shs"]), d["sh_degree"], t(d["campos"]), n["geom"], n["num_rendered"],
        n["binning"], n["img"], False)
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
             "dL_drots")
    torch.cuda.synchronize(dev)
    return {k: v.cpu().numpy() for k, v in zip(names, outs)}


def fragile_pixels(o, per=5000, window=1e-6):
    """bool[H, W] of the pixels whose fp32 blend holds a decision one ulp of exp() can flip (oracle/ggd_oracle.c::
    ggo_fragile_pixels: an alpha within 1e-6 of the 1/255 floor or a transmittance test within 1e-6 of the 1e-4 stop).  Both
    outcomes are correct fp32 results but differ by ~1/255 of everything behind the flipped contributor, so the parity tests
    compare colours on the other pixels and zero the upstream gradient of these (for the HIP backward and the reference
    alike).  Their number is bounded: at most one per `per` pixels (+2)."""
    from oracle import ggd_oracle as O
    m = O.fragile_pixels(o, window=window)
    assert int(m.sum()) <= 2 + (o["W"] * o["H"]) // per, f"{int(m.sum())} fragile pixels"
    return m


def assert_blend_matches(n, o, atol=1e-5, window=1e-6, per=5000, what="", color=None):
    """The blend outputs of a native forward `n` against the oracle forward `o`, excluding pixels BY CAUSE: the last
    contributor must be the oracle's everywhere outside the oracle's own fragile-pixel mask (a decision within `window` of a
    threshold -- 1e-6 covers the <= 1 ulp exp of modes 0 / 2 and the bare v_exp_f32 of the default mode 3, whose x * log2(e)
    product adds ~4e-7 relative at power = -5.5), colour and final_T within `atol` on every other pixel.  Returns (mask, max
    colour error outside the mask)."""
    frag = fragile_pixels(o, per=per, window=window)
    bad = (n["n_contrib"] != o["n_contrib"]) & ~frag
    assert not bad.any(), f"{what}: {int(bad.sum())} pixels stop at another contributor outside the oracle's fragile mask"
    c = n["color"] if color is None else color
    c = c.cpu().numpy() if torch.is_tensor(c) else c
    ok = ~frag
    err = float(np.abs(c - o["color"])[:, ok].max(initial=0.0))
    assert err <= atol, f"{what}: max |dRGB| = {err} outside the fragile mask"
    assert np.abs(n["final_T"] - o["final_T"])[ok].max(initial=0.0) <= atol, what
    return frag, err


EPS32 = 2.0 ** -24
ATOL = 1e-5      # the north_star's absolute bar
KAPPA = 0.25     # factor on the (worst-case) fp32 error budget of the reference (oracle/ggd_oracle.py::backward_ref64):
                 # measured, the HIP backward sits at <= 0.11 and the fp32 CPU oracle at <= 0.15 of the budget itself


def backward_reference(d, o, n, dL_dpix):
    """fp64 reference + per-element fp32 error budget for the backward of ONE forward: o = fp32 oracle forward of `d`,
    n = run_native(d) (its saved final_T / n_contrib / list / ranges are what the HIP backward consumes, so they are
    what the reference consumes too).  Asserts that the per-Gaussian state the backward reads is the oracle's."""
    from oracle import ggd_oracle as O
    vis = o["radii"] > 0
    np.testing.assert_array_equal(n["radii"].cpu().numpy(), o["radii"])
    np.testing.assert_array_equal(n["xy"][vis], o["xy"][vis])
    np.testing.assert_array_equal(n["conic_opacity"][vis], o["conic_opacity"][vis])
    np.testing.assert_array_equal(n["rgb"][vis], o["rgb"][vis])
    return O.backward_ref64(o, np.asarray(dL_dpix, np.float32), final_T=n["final_T"], n_contrib=n["n_contrib"],
                            point_list=n["point_list"], ranges=n["ranges"])


def check_gradients(d, got: dict, ref: dict, budget: dict, fragile, kappa=KAPPA, report=None):
    """|gpu - ref| <= ATOL + kappa * eps32 * budget for every element of every gradient array; Gaussians with a
    (pixel, Gaussian) pair on the alpha floor (`fragile`) are left out, and their number is bounded."""
    P = d["P"]
    frag = fragile > 0
    assert int(frag.sum()) <= max(4, P // 1000), f"{int(frag.sum())} Gaussians sit on the alpha floor"
    worst = 0.0
    for name, r in ref.items():
        if name == "dL_dconic" or r is None:
            continue
        if name == "dL_dsh" and d["shs"] is None:
            continue
        if name in ("dL_dscales", "dL_drots") and d["scales"] is None:
            continue
        g = got[name].reshape(r.shape).astype(np.float64)
        assert np.isfinite(g).all(), f"{name}: non-finite values (an element the library never wrote?)"
        diff = np.abs(g - r)
        tol = ATOL + kappa * EPS32 * budget[name]
        ratio = diff / tol
        ratio[frag] = 0.0
        rel = diff / np.maximum(np.abs(r), 1e-30)
        big = np.abs(r) >= 1e-3 * max(np.abs(r).max(), 1e-30)
        if report is not None:
            report.append(dict(array=name, max_abs_err=float(diff[~frag].max(initial=0.0)),
                               max_rel_err_on_large=float(rel[big & ~frag.reshape((-1,) + (1,) * (r.ndim - 1))].max(initial=0.0)),
                               max_abs_value=float(np.abs(r).max(initial=0.0)), worst_ratio=float(ratio.max(initial=0.0)),
                               median_tol=float(np.median(tol))))
        worst = max(worst, float(ratio.max(initial=0.0)))
    return worst


def msd_window(ranges, buckets=512):
    """The two-launch depth sort's key window (csrc/ggd_capi.hip::msd_fit_window) for frames whose kept depth keys (uint32 views of
    the fp32 depths) span `ranges` = [(min, max), ...]: returns (lo, shift), or None when the window is too wide for the two
    finishing passes.  A frame's bucket sizes are then np.bincount(np.minimum((keys - lo) >> shift, 1023))."""
    lo, hi = min(int(r[0]) for r in ranges), max(int(r[1]) for r in ranges)
    margin = ((hi - lo) >> 3) + 4096
    wlo, whi = max(0, lo - margin), min(0xfffffffe, hi + margin)
    span, shift = whi - wlo, 0
    while (span >> shift) >= buckets:
        shift += 1
    return (wlo, shift) if shift <= 16 else None


def depth_keys(o):
    """uint32 depth keys of the oracle forward's visible Gaussians (int64 array)"""
    return o["depths"][o["radii"] > 0].astype(np.float32).view(np.uint32).astype(np.int64)


def msd_bucket_sizes(o, window=None):
    """bucket sizes the two-launch sort would see for the oracle frame `o` over `window` (default: fitted to the frame itself)"""
    dk = depth_keys(o)
    w = window if window is not None else msd_window([(dk.min(), dk.max())])
    if w is None:
        return None
    return np.bincount(np.minimum((dk - w[0]) >> w[1], 1023), minlength=1024)
