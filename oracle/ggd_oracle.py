"""ctypes front-end of the CPU oracle (oracle/ggd_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(gaussian_gan_decoder_amd) never does.  PARITY UNPINNED: see the header of oracle/ggd_oracle_impl.inc.

Stage-by-stage, every intermediate buffer of SURVEY.md section 8a is returned so the HIP path can be compared
stage by stage (integers bit-exact, floats within the tolerance written in each test).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libggd_oracle.so")


class _Params(C.Structure):
    _fields_ = [("P", C.c_int32), ("M", C.c_int32), ("D", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
                ("prefiltered", C.c_int32), ("tanfovx", C.c_double), ("tanfovy", C.c_double),
                ("scale_modifier", C.c_double)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("ggd_oracle.c", "ggd_oracle_impl.inc", "ggd_oracle_bound.cpp",
                                          "ggd_oracle_types.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libggd_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ggo_scan.restype = C.c_int64
        _lib.ggo_fragile_pixels.restype = C.c_int64
        _lib.ggo_higher_msb.restype = C.c_uint32
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=dt))
    if shape is not None:
        a = a.reshape(shape)
    return a


def higher_msb(n: int) -> int:
    return int(lib().ggo_higher_msb(C.c_uint32(n)))


def sort_bits(W: int, H: int) -> int:
    T = ((W + 15) // 16) * ((H + 15) // 16)
    return 32 + higher_msb(T)


def forward(*, means3D, opacities, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy,
            shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            sh_degree=0, scale_modifier=1.0, prefiltered=False, dtype=np.float32, stop_after=None):
    """Run all forward stages on the CPU.  Returns a dict of numpy arrays (inputs included)."""
    L = lib()
    dt = np.dtype(dtype)
    suf = "_f32" if dt == np.float32 else "_f64"
    means3D = _c(means3D, dt, (-1, 3))
    P = means3D.shape[0]
    assert (shs is None) != (colors_precomp is None), "exactly one of shs / colors_precomp"
    assert ((scales is None) and (rotations is None)) != (cov3D_precomp is None), "exactly one of scale+rot / cov3D"
    opacities = _c(opacities, dt, (-1,))
    shs = _c(shs, dt)
    M = 0 if shs is None else (shs.shape[1] if shs.ndim == 3 else shs.reshape(max(P, 1), -1, 3).shape[1])
    if shs is not None:
        shs = shs.reshape(P, M, 3)
        assert (sh_degree + 1) ** 2 <= M
    colors_precomp = _c(colors_precomp, dt, (-1, 3))
    scales = _c(scales, dt, (-1, 3))
    rotations = _c(rotations, dt, (-1, 4))
    cov3D_precomp = _c(cov3D_precomp, dt, (-1, 6))
    view = _c(viewmatrix, dt, (16,))
    proj = _c(projmatrix, dt, (16,))
    campos = _c(campos, dt, (3,))
    bg = _c(bg, dt, (3,))
    prm = _Params(P, M, int(sh_degree), int(W), int(H), int(bool(prefiltered)),
                  float(np.float32(tanfovx)) if dt == np.float32 else float(tanfovx),
                  float(np.float32(tanfovy)) if dt == np.float32 else float(tanfovy),
                  float(np.float32(scale_modifier)) if dt == np.float32 else float(scale_modifier))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    out = dict(P=P, M=M, W=W, H=H, T=T, dtype=dt, sh_degree=int(sh_degree), prm=prm,
               means3D=means3D, opacities=opacities, shs=shs, colors_precomp=colors_precomp, scales=scales,
               rotations=rotations, cov3D_precomp=cov3D_precomp, viewmatrix=view, projmatrix=proj, campos=campos,
               bg=bg)
    depths = np.zeros(P, dt); radii = np.zeros(P, np.int32); xy = np.zeros((P, 2), dt)
    cov3D = np.zeros((P, 6), dt); conic_opacity = np.zeros((P, 4), dt); rgb = np.zeros((P, 3), dt)
    tiles_touched = np.zeros(P, np.uint32); clamped = np.zeros((P, 3), np.uint8); rect = np.zeros((P, 4), np.int32)
    rc = getattr(L, "ggo_preprocess" + suf)(C.byref(prm), _p(view), _p(proj), _p(campos), _p(means3D), _p(shs),
                                            _p(colors_precomp), _p(opacities), _p(scales), _p(rotations),
                                            _p(cov3D_precomp), _p(depths), _p(radii), _p(xy), _p(cov3D),
                                            _p(conic_opacity), _p(rgb), _p(tiles_touched), _p(clamped), _p(rect))
    if rc != 0:
        raise RuntimeError(f"oracle preprocess failed rc={rc} (prefiltered point culled)")
    out.update(depths=depths, radii=radii, xy=xy, cov3D=cov3D, conic_opacity=conic_opacity, rgb=rgb,
               tiles_touched=tiles_touched, clamped=clamped, rect=rect)
    offsets = np.zeros(P, np.uint32)
    R = int(L.ggo_scan(P, _p(tiles_touched), _p(offsets)))
    out.update(point_offsets=offsets, num_rendered=R)
    if stop_after == "scan":
        return out
    keys_u = np.zeros(R, np.uint64); vals_u = np.zeros(R, np.uint32)
    getattr(L, "ggo_duplicate" + suf)(P, W, _p(radii), _p(rect), _p(depths), _p(offsets), _p(keys_u), _p(vals_u))
    keys = np.zeros(R, np.uint64); vals = np.zeros(R, np.uint32)
    nbits = sort_bits(W, H)
    rc = L.ggo_sort_pairs(C.c_int64(R), _p(keys_u), _p(vals_u), _p(keys), _p(vals), nbits)
    assert rc == 0
    ranges = np.zeros((T, 2), np.uint32)
    L.ggo_tile_ranges(C.c_int64(R), _p(keys), T, _p(ranges))
    out.update(keys_unsorted=keys_u, list_unsorted=vals_u, keys=keys, point_list=vals, ranges=ranges,
               sort_bits=nbits)
    if stop_after == "binning":
        return out
    color = np.zeros((3, H, W), dt); final_T = np.zeros((H, W), dt); n_contrib = np.zeros((H, W), np.uint32)
    getattr(L, "ggo_render" + suf)(C.byref(prm), _p(bg), _p(ranges), _p(vals), _p(xy), _p(conic_opacity), _p(rgb),
                                   _p(color), _p(final_T), _p(n_contrib))
    out.update(color=color, final_T=final_T, n_contrib=n_contrib)
    return out


def backward(fwd: dict, dL_dpix):
    """Backward stages a10 + a11 on the CPU.  `fwd` is the dict returned by forward()."""
    L = lib()
    dt = fwd["dtype"]
    suf = "_f32" if dt == np.float32 else "_f64"
    P, M, W, H = fwd["P"], fwd["M"], fwd["W"], fwd["H"]
    prm = fwd["prm"]
    g = _c(dL_dpix, dt, (3, H, W))
    d_mean2D = np.zeros((P, 2), np.float64); d_conic = np.zeros((P, 3), np.float64)
    d_opacity = np.zeros(P, np.float64); d_colors = np.zeros((P, 3), np.float64)
    getattr(L, "ggo_render_backward" + suf)(C.byref(prm), _p(fwd["bg"]), _p(fwd["ranges"]), _p(fwd["point_list"]),
                                            _p(fwd["xy"]), _p(fwd["conic_opacity"]), _p(fwd["rgb"]),
                                            _p(fwd["final_T"]), _p(fwd["n_contrib"]), _p(g),
                                            _p(d_mean2D), _p(d_conic), _p(d_opacity), _p(d_colors))
    m2 = d_mean2D.astype(dt); co = d_conic.astype(dt); dc = d_colors.astype(dt)
    d_means3D = np.zeros((P, 3), dt); d_cov3D = np.zeros((P, 6), dt)
    d_sh = np.zeros((P, max(M, 1), 3), dt) if M > 0 else None
    d_scales = np.zeros((P, 3), dt); d_rots = np.zeros((P, 4), dt)
    have_cp = fwd["colors_precomp"] is not None
    have_c3 = fwd["cov3D_precomp"] is not None
    getattr(L, "ggo_preprocess_backward" + suf)(
        C.byref(prm), _p(fwd["viewmatrix"]), _p(fwd["projmatrix"]), _p(fwd["campos"]), _p(fwd["means3D"]),
        _p(fwd["shs"]), int(have_cp), _p(fwd["scales"]), _p(fwd["rotations"]), _p(fwd["cov3D"]), int(have_c3),
        _p(fwd["radii"]), _p(fwd["clamped"]), _p(m2), _p(co), _p(dc),
        _p(d_means3D), _p(d_cov3D), _p(d_sh), _p(d_scales), _p(d_rots))
    d_means2D = np.zeros((P, 3), dt)
    d_means2D[:, :2] = m2
    return dict(dL_dmeans2D=d_means2D, dL_dconic=co, dL_dopacity=d_opacity.astype(dt), dL_dcolors=dc,
                dL_dmeans3D=d_means3D, dL_dcov3D=d_cov3D, dL_dsh=d_sh, dL_dscales=d_scales, dL_drots=d_rots)


def backward_ref64(fwd: dict, dL_dpix, final_T=None, n_contrib=None, point_list=None, ranges=None):
    """Reference for the GPU backward tests: fp32 decisions, fp64 values (ggo_render_backward_ref64 + the fp64 twin of
    stage a11), and a per-element error budget.

    `fwd` is an fp32 forward() dict.  final_T / n_contrib / point_list / ranges default to the oracle's own; the GPU
    tests pass the buffers the HIP forward saved (what its backward consumes), so a threshold that flipped in the forward
    does not enter the comparison of the backward.

    Returns (ref, budget, fragile):
      ref[name]    fp64 gradients, same names / shapes as backward()
      budget[name] >= 0, same shapes, in units of eps32 = 2^-24: the conditioning S of the a10 sums (weighted sum of
                   |terms|, see the C source) pushed through stage a11 by running error analysis
                   (ggd_oracle_bound.cpp), so that   |fp32 result - ref| <= kappa * eps32 * budget   is the bound to test
      fragile      uint32[P]: number of (pixel, Gaussian) pairs within 1e-6 of the alpha floor (see the C source)."""
    L = lib()
    assert fwd["dtype"] == np.float32
    P, M, W, H = fwd["P"], fwd["M"], fwd["W"], fwd["H"]
    prm = fwd["prm"]
    g = _c(dL_dpix, np.float32, (3, H, W))
    fT = _c(fwd["final_T"] if final_T is None else final_T, np.float32, (H, W))
    nc = _c(fwd["n_contrib"] if n_contrib is None else n_contrib, np.uint32, (H, W))
    pl = _c(fwd["point_list"] if point_list is None else point_list, np.uint32)
    rg = _c(fwd["ranges"] if ranges is None else ranges, np.uint32, (-1, 2))
    z = lambda *sh: np.zeros(sh, np.float64)
    m2, co, dop, dc = z(P, 2), z(P, 3), z(P), z(P, 3)
    Sm2, Sco, Sop, Sdc = z(P, 2), z(P, 3), z(P), z(P, 3)
    fragile = np.zeros(P, np.uint32)
    L.ggo_render_backward_ref64(C.byref(prm), _p(fwd["bg"]), _p(rg), _p(pl), _p(fwd["xy"]), _p(fwd["conic_opacity"]),
                                _p(fwd["rgb"]), _p(fT), _p(nc), _p(g), _p(m2), _p(co), _p(dop), _p(dc),
                                _p(Sm2), _p(Sco), _p(Sop), _p(Sdc), _p(fragile))
    # stage a11 over the error-tracking number type (ggd_oracle_bound.cpp): value in double + running error bound in
    # units of eps32.  Scene data enters with e = 0 (exact fp32 numbers), the a10 sums with e = their conditioning S;
    # the discrete state (radii, clamped) is the fp32 forward's.
    er = lambda a, e=None: None if a is None else np.ascontiguousarray(
        np.stack([np.asarray(a, np.float64), np.zeros_like(a, np.float64) if e is None else e], axis=-1))
    prm64 = _Params(P, M, fwd["sh_degree"], W, H, 0, float(prm.tanfovx), float(prm.tanfovy), float(prm.scale_modifier))
    have_cp = fwd["colors_precomp"] is not None
    have_c3 = fwd["cov3D_precomp"] is not None
    view, proj, campos = er(fwd["viewmatrix"]), er(fwd["projmatrix"]), er(fwd["campos"])
    means, shs = er(fwd["means3D"]), er(fwd["shs"])
    scales, rots = er(fwd["scales"]), er(fwd["rotations"])
    if have_c3:
        cov = er(fwd["cov3D_precomp"])
    else:  # Sigma with the rounding an fp32 evaluation of R S S R^T carries
        cov = np.zeros((P, 6, 2), np.float64)
        L.ggo_cov3d_all_err(P, _p(scales), C.c_double(float(prm.scale_modifier)), _p(rots), _p(cov))
    radii, clamped = fwd["radii"], fwd["clamped"]
    ze = lambda *sh: np.zeros(sh + (2,), np.float64)
    d_means3D, d_cov3D = ze(P, 3), ze(P, 6)
    d_sh = ze(P, max(M, 1), 3) if M > 0 else None
    d_scales, d_rots = ze(P, 3), ze(P, 4)
    L.ggo_preprocess_backward_err(C.byref(prm64), _p(view), _p(proj), _p(campos), _p(means), _p(shs), int(have_cp),
                                  _p(scales), _p(rots), _p(cov), int(have_c3), _p(radii), _p(clamped),
                                  _p(er(m2, Sm2)), _p(er(co, Sco)), _p(er(dc, Sdc)), _p(d_means3D), _p(d_cov3D),
                                  _p(d_sh), _p(d_scales), _p(d_rots))
    vis = (radii > 0)
    outs = dict(dL_dmeans3D=d_means3D, dL_dcov3D=d_cov3D, dL_dsh=d_sh, dL_dscales=d_scales, dL_drots=d_rots)
    ref = {k: (None if v is None else np.ascontiguousarray(v[..., 0])) for k, v in outs.items()}
    budget = {k: (None if v is None else np.ascontiguousarray(v[..., 1])) for k, v in outs.items()}
    d_means2D = z(P, 3); d_means2D[:, :2] = m2 * vis[:, None]
    b_means2D = z(P, 3); b_means2D[:, :2] = Sm2 * vis[:, None]
    ref.update(dL_dmeans2D=d_means2D, dL_dconic=co, dL_dopacity=dop * vis, dL_dcolors=dc * vis[:, None])
    budget.update(dL_dmeans2D=b_means2D, dL_dconic=Sco, dL_dopacity=Sop * vis, dL_dcolors=Sdc * vis[:, None])
    return ref, budget, fragile


def fragile_pixels(fwd: dict, window: float = 1e-6, point_list=None, ranges=None):
    """bool[H, W]: pixels whose fp32 blend holds a decision within `window` of a threshold (alpha floor 1/255, transmittance
    stop 1e-4) -- see ggo_fragile_pixels in ggd_oracle.c.  `fwd` is an fp32 forward() dict."""
    L = lib()
    assert fwd["dtype"] == np.float32
    W, H = fwd["W"], fwd["H"]
    pl = _c(fwd["point_list"] if point_list is None else point_list, np.uint32)
    rg = _c(fwd["ranges"] if ranges is None else ranges, np.uint32, (-1, 2))
    mask = np.zeros((H, W), np.uint8)
    L.ggo_fragile_pixels(C.byref(fwd["prm"]), _p(rg), _p(pl), _p(fwd["xy"]), _p(fwd["conic_opacity"]),
                         C.c_double(window), _p(mask))
    return mask.astype(bool)


def mark_visible(means3D, viewmatrix):
    L = lib()
    m = _c(means3D, np.float32, (-1, 3)); v = _c(viewmatrix, np.float32, (16,))
    out = np.zeros(m.shape[0], np.uint8)
    L.ggo_mark_visible(m.shape[0], _p(m), _p(v), _p(out))
    return out.astype(bool)


def sh_to_rgb(deg, sh, p, campos):
    """Oracle SH evaluation of ONE Gaussian: sh [M,3], p [3], campos [3] -> (rgb[3], clamped[3])."""
    L = lib()
    sh = _c(sh, np.float32); p = _c(p, np.float32, (3,)); cp = _c(campos, np.float32, (3,))
    rgb = np.zeros(3, np.float32); cl = np.zeros(3, np.uint8)
    L.ggo_test_sh_to_rgb(int(deg), _p(sh), _p(p), _p(cp), _p(rgb), _p(cl))
    return rgb, cl


def cov3d(scale, mod, quat):
    L = lib()
    s = _c(scale, np.float32, (3,)); q = _c(quat, np.float32, (4,))
    out = np.zeros(6, np.float32)
    L.ggo_test_cov3d(_p(s), C.c_float(mod), _p(q), _p(out))
    return out
