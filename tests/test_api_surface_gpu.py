"""GPU tests of the drop-in surface the reference imports: the top-level `diff_gaussian_rasterization`
shim (gaussian_splatting/gaussian_renderer/__init__.py:14), `render()` with every `pipe` switch (:19-102),
`GaussianRasterizer.markVisible` / `mark_visible` / the C entry point `ggd_mark_visible` (row a12)."""
import ctypes as C
import math
import types

import numpy as np
import pytest
import torch

from _util import scene_inputs, run_oracle, run_native, adversarial_inputs, backward_reference, check_gradients

pytestmark = pytest.mark.gpu


def test_mark_visible_matches_oracle(native_lib):
    """a12: bool[P] = passes the view-space z > 0.2 test; through the C ABI, the functional form and the module."""
    import diff_gaussian_rasterization as dgr          # the reference's import line resolves to the gfx950 library
    from gaussian_gan_decoder_amd import _capi
    from oracle import ggd_oracle as O
    dev = torch.device("cuda:0")
    for d in (adversarial_inputs(), scene_inputs(P=100_000, size=256, kind="cube", h=0.7, v=1.2)):
        pts = d["means3D"].clone()
        # put a third of the points behind / on the near plane: along the viewing direction through the camera
        view = d["viewmatrix"]
        cam_pos, fwd = torch.inverse(view)[3, :3], view[:3, 2]
        k = pts.shape[0] // 3
        g = torch.Generator().manual_seed(3)
        pts[:k] = cam_pos + fwd * (0.2 + 0.4 * (torch.rand(k, 1, generator=g) - 0.5)) + 0.05 * torch.randn(k, 3, generator=g)
        pts[k] = cam_pos + 0.2 * fwd
        want = O.mark_visible(pts.numpy(), view.numpy())
        assert 0 < want.sum() < len(want)
        p_dev, v_dev, pr_dev = pts.to(dev), view.to(dev), d["projmatrix"].to(dev)
        got_fn = dgr.mark_visible(p_dev, v_dev, pr_dev)
        assert got_fn.dtype == torch.bool and got_fn.shape == (pts.shape[0],)
        np.testing.assert_array_equal(got_fn.cpu().numpy(), want)
        rs = dgr.GaussianRasterizationSettings(d["H"], d["W"], d["tanfovx"], d["tanfovy"], d["bg"].to(dev), 1.0, v_dev,
                                               pr_dev, 0, d["campos"].to(dev), False, False)
        got_mod = dgr.GaussianRasterizer(rs).markVisible(p_dev)
        np.testing.assert_array_equal(got_mod.cpu().numpy(), want)
        # the C entry point itself
        ctx = _capi.context_for(dev)
        out = torch.full((pts.shape[0],), 7, dtype=torch.uint8, device=dev)
        ctx.check(ctx.lib.ggd_mark_visible(ctx.handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream),
                                           pts.shape[0], C.c_void_p(p_dev.data_ptr()), C.c_void_p(v_dev.data_ptr()),
                                           C.c_void_p(pr_dev.data_ptr()), C.c_void_p(out.data_ptr())))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), want.astype(np.uint8))
    assert dgr.mark_visible(torch.empty(0, 3, device=dev), v_dev, pr_dev).numel() == 0


def _container(d, max_deg, active_deg, dev):
    """The reference's GaussianModel filled with the activated scene of `d` (raw attributes = inverse activations)."""
    from gaussian_gan_decoder_amd.gaussian_model import GaussianModel
    pc = GaussianModel(max_deg)
    pc.active_sh_degree = active_deg
    mk = lambda t: t.to(dev).clone().requires_grad_(True)
    pc._xyz = mk(d["means3D"])
    pc._scaling = mk(torch.log(d["scales"]))
    pc._rotation = mk(d["rotations"])
    op = d["opacities"].clamp(1e-6, 1 - 1e-6)
    pc._opacity = mk(torch.log(op / (1 - op)))
    pc._features_dc = mk(d["shs"][:, :1].contiguous())
    pc._features_rest = mk(d["shs"][:, 1:].contiguous())
    return pc


@pytest.mark.parametrize("cov_py", [False, True], ids=["cov3D-in-kernel", "compute_cov3D_python"])
@pytest.mark.parametrize("sh_py", [False, True], ids=["SH-in-kernel", "convert_SHs_python"])
def test_render_through_the_shim_matches_oracle(native_lib, cov_py, sh_py):
    """`render(viewpoint_camera, pc, pipe, bg)` as gaussian_splatting/train.py:86 calls it, with the rasterizer
    imported through the top-level shim, max_sh_degree 3 and active degree 1 (so shs carries 16 coefficients of which 4
    are active, the state of stock 3DGS training after the first oneupSHdegree): image, radii and every gradient that
    reaches the container against the oracle."""
    import diff_gaussian_rasterization as dgr
    from gaussian_gan_decoder_amd import rasterizer, gaussian_renderer
    from gaussian_gan_decoder_amd.synthetic import make_camera, make_dL_dpix
    assert dgr.GaussianRasterizer is rasterizer.GaussianRasterizer
    assert dgr.GaussianRasterizationSettings is rasterizer.GaussianRasterizationSettings
    from gaussian_gan_decoder_amd import _capi as _c
    _c.context_for(torch.device("cuda:0")).poison_outputs = True
    dev = torch.device("cuda:0")
    S, P = 160, 6000
    d = scene_inputs(P=P, size=S, kind="cube", seed=11, sh_degree=1, sh_M=16, lsm=-5.0, h=1.0, v=1.3)
    cam = make_camera(S, 12.0, 1.0, 1.3, device=dev)
    cam.camera_center = d["campos"].to(dev)           # what scene_inputs hands the oracle (inverse of the view matrix)
    pc = _container(d, 3, 1, dev)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=cov_py, convert_SHs_python=sh_py)
    out = gaussian_renderer.render(cam, pc, pipe, d["bg"].to(dev), scaling_modifier=1.0)
    assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii"}
    g = make_dL_dpix(S)
    (out["render"] * g.to(dev)).sum().backward()
    cpu = lambda t: t.detach().cpu()
    # the rasterizer call render() made: activations (and python SH / covariance when asked) by torch on the GPU
    dd = dict(d)
    dd["opacities"] = cpu(pc.get_opacity); dd["means3D"] = cpu(pc.get_xyz)
    if cov_py:
        dd["cov3D_precomp"] = cpu(pc.get_covariance(1.0)).contiguous(); dd["scales"] = dd["rotations"] = None
    else:
        dd["scales"], dd["rotations"] = cpu(pc.get_scaling).contiguous(), cpu(pc.get_rotation).contiguous()
    if sh_py:
        from gaussian_gan_decoder_amd.sh import eval_sh
        feats = pc.get_features
        shs_view = feats.transpose(1, 2).view(-1, 3, 16)
        dirs = pc.get_xyz - cam.camera_center.repeat(P, 1)
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        dd["colors_precomp"] = cpu(torch.clamp_min(eval_sh(1, shs_view, dirs) + 0.5, 0.0)).contiguous(); dd["shs"] = None
    else:
        dd["shs"] = cpu(pc.get_features).contiguous()
    o = run_oracle(dd)
    np.testing.assert_array_equal(cpu(out["radii"]).numpy(), o["radii"])
    np.testing.assert_array_equal(cpu(out["visibility_filter"]).numpy(), o["radii"] > 0)
    n = run_native(dd, debug=False)
    same = n["n_contrib"] == o["n_contrib"]
    assert (~same).sum() <= 1
    assert np.abs(cpu(out["render"]).numpy() - o["color"])[:, same].max() <= 1e-5
    # gradients that reach the rasterizer's inputs directly (no torch prologue in between)
    ref, bud, fragile = backward_reference(dd, o, n, g.numpy())
    got = dict(dL_dmeans2D=cpu(out["viewspace_points"].grad).numpy())
    ref2, bud2 = dict(dL_dmeans2D=ref["dL_dmeans2D"]), dict(dL_dmeans2D=bud["dL_dmeans2D"])
    if not sh_py:   # features reach the kernel unchanged: d/d(features) = dL_dsh (dc = coefficient 0, rest = 1..15)
        got["dL_dsh"] = torch.cat([cpu(pc._features_dc.grad), cpu(pc._features_rest.grad)], 1).numpy()
        ref2["dL_dsh"], bud2["dL_dsh"] = ref["dL_dsh"], bud["dL_dsh"]
        assert (got["dL_dsh"][:, 4:] == 0).all() and np.isfinite(got["dL_dsh"]).all()
    if not sh_py and not cov_py:   # xyz only feeds the rasterizer
        got["dL_dmeans3D"] = cpu(pc._xyz.grad).numpy()
        ref2["dL_dmeans3D"], bud2["dL_dmeans3D"] = ref["dL_dmeans3D"], bud["dL_dmeans3D"]
    worst = check_gradients(dd, got, ref2, bud2, fragile)
    assert worst <= 1.0, worst
    for t in (pc._xyz, pc._scaling, pc._rotation, pc._opacity, pc._features_dc, pc._features_rest):
        assert t.grad is not None and torch.isfinite(t.grad).all()
    _c.context_for(torch.device("cuda:0")).poison_outputs = False


def test_non_fp32_inputs_are_refused(native_lib):
    import diff_gaussian_rasterization as dgr
    dev = torch.device("cuda:0")
    z = lambda *s: torch.zeros(*s, device=dev)
    rs = dgr.GaussianRasterizationSettings(16, 16, 0.1, 0.1, z(3), 1.0, torch.eye(4, device=dev), torch.eye(4, device=dev),
                                           0, z(3), False, False)
    with pytest.raises(TypeError, match="float32"):
        dgr.GaussianRasterizer(rs)(z(4, 3).double(), z(4, 3), z(4, 1), shs=z(4, 1, 3), scales=z(4, 3), rotations=z(4, 4))


def test_two_streams_render_concurrently(native_lib):
    """Contexts are per (device, stream): two scenes rendered from two side streams at the same time (forward + backward,
    no synchronisation between the streams) give exactly what each gives alone on the default stream."""
    from _util import scene_inputs
    from gaussian_gan_decoder_amd import rasterizer as R
    dev = torch.device("cuda:0")

    def args_of(d, leaf):
        t = lambda x: torch.empty(0, device=dev) if x is None else x.to(dev)
        return (t(d["bg"]), leaf, t(d["colors_precomp"]), t(d["opacities"]), t(d["scales"]), t(d["rotations"]),
                d["scale_modifier"], t(d["cov3D_precomp"]), t(d["viewmatrix"]), t(d["projmatrix"]), d["tanfovx"],
                d["tanfovy"], d["H"], d["W"], t(d["shs"]), d["sh_degree"], t(d["campos"]), False, False)

    def run(d, stream):
        leaf = d["means3D"].to(dev).clone().requires_grad_(True)
        a = args_of(d, leaf)
        with torch.cuda.stream(stream):
            color, radii = R.rasterize_gaussians(a[1], torch.zeros_like(a[1]), a[14], a[2], a[3], a[4], a[5], a[7],
                                                 R.GaussianRasterizationSettings(a[12], a[13], a[10], a[11], a[0], a[6],
                                                                                 a[8], a[9], a[15], a[16], False, False))
            (color * color).sum().backward()
        return color, leaf

    scenes = [scene_inputs(P=30000, size=256, seed=1), scene_inputs(P=20000, size=192, seed=2, kind="shell", lsm=-5.5)]
    torch.cuda.synchronize(dev)
    alone = []
    for d in scenes:
        c, leaf = run(d, torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        alone.append((c.detach().cpu(), leaf.grad.detach().cpu()))
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for rep in range(3):
        outs = [run(d, s) for d, s in zip(scenes, streams)]
        torch.cuda.synchronize(dev)
        for (c, leaf), (c0, g0) in zip(outs, alone):
            assert torch.equal(c.detach().cpu(), c0)
            # the backward's float atomics are unordered: same tolerance class as two runs on one stream
            assert (leaf.grad.detach().cpu() - g0).abs().max() <= 1e-4 * max(1.0, float(g0.abs().max()))


def test_back_to_back_forwards_do_not_disturb_each_other(native_lib):
    """The single-call forward returns when num_rendered has arrived, with binning and blend possibly still running.  Three
    different scenes (same shape, so the later calls take the single-call route and share the context's scratch) are
    rendered back to back without any synchronisation in between; every result must equal the scene rendered alone."""
    from _util import scene_inputs, run_native
    dev = torch.device("cuda:0")
    scenes = [scene_inputs(P=300000, size=512, seed=s, lsm=-5.5) for s in (11, 12, 13)]
    alone = []
    for d in scenes:
        n = run_native(d, debug=False)     # first call of the shape: two-call form; later ones: single-call
        torch.cuda.synchronize(dev)
        alone.append((n["num_rendered"], n["color"].clone(), n["point_list"].copy()))
    from gaussian_gan_decoder_amd import rasterizer as R
    t = lambda x: torch.empty(0, device=dev) if x is None else x.to(dev)
    args = [(t(d["bg"]), t(d["means3D"]), t(d["colors_precomp"]), t(d["opacities"]), t(d["scales"]), t(d["rotations"]),
             d["scale_modifier"], t(d["cov3D_precomp"]), t(d["viewmatrix"]), t(d["projmatrix"]), d["tanfovx"], d["tanfovy"],
             d["H"], d["W"], t(d["shs"]), d["sh_degree"], t(d["campos"]), False, False) for d in scenes]
    torch.cuda.synchronize(dev)
    for rep in range(3):
        outs = [R.rasterize_gaussians_native(*a) for a in args]      # no sync between the calls
        torch.cuda.synchronize(dev)
        for (Rn, color, radii, geom, binning, img), (R0, c0, l0) in zip(outs, alone):
            assert Rn == R0
            assert torch.equal(color, c0)
            lst = binning.cpu().numpy()[:4 * Rn].view(np.uint32)       # the list sits at offset 0 of the binning buffer
            np.testing.assert_array_equal(lst, l0)


def test_blend_statistics_do_not_change_the_image_and_account_for_every_wave(native_lib):
    """The debug statistics of the forward blend (`ggd_blend_stats`, `ggd_blend_timeline`): the counted and the timed
    instantiation of the kernel render the same image as the production one; the counters are consistent with the
    frame (sum of list lengths = 4 quarter waves x num_rendered); the timeline has one slot per wave, every wave ends
    after it starts and gathers no more than its list holds."""
    from gaussian_gan_decoder_amd import rasterizer as R, _capi
    dev = torch.device("cuda:0")
    d = scene_inputs(P=20000, size=256, seed=5, lsm=-3.5)
    t = lambda x: torch.empty(0, device=dev) if x is None else x.to(dev)
    args = (t(d["bg"]), t(d["means3D"]), t(d["colors_precomp"]), t(d["opacities"]), t(d["scales"]), t(d["rotations"]),
            d["scale_modifier"], t(d["cov3D_precomp"]), t(d["viewmatrix"]), t(d["projmatrix"]), d["tanfovx"], d["tanfovy"],
            d["H"], d["W"], t(d["shs"]), d["sh_degree"], t(d["campos"]), False, False)
    ctx = _capi.context_for(dev)
    ref = R.rasterize_gaussians_native(*args)
    torch.cuda.synchronize(dev)
    try:
        ctx.blend_stats(1)
        counted = R.rasterize_gaussians_native(*args)
        st = ctx.blend_stats(2)
        timed = R.rasterize_gaussians_native(*args)
        waves = 4 * ((d["W"] + 15) // 16) * ((d["H"] + 15) // 16)
        tl = ctx.blend_timeline(waves)
    finally:
        ctx.blend_stats(False)
    assert torch.equal(counted[1], ref[1]) and torch.equal(timed[1], ref[1])
    assert st["listed"] == 4 * ref[0]
    assert 0 < st["visited"] <= st["listed"] and st["culled"] <= st["visited"]
    assert tl.shape == (waves, 4)
    assert (tl[:, 1] >= tl[:, 0]).all() and (tl[:, 0] > 0).all()
    assert int(tl[:, 2].sum()) == 4 * ref[0]
    assert (tl[:, 3] <= tl[:, 2]).all()
    assert int(tl[:, 3].sum()) == st["visited"]


def test_forward_calls_are_refused_while_a_frame_is_pending(native_lib):
    """Between ggd_forward_enqueue and ggd_forward_collect the frame's verification state (was the speculated sort form valid?)
    lives on the context: any other forward on that context is refused with GGD_E_INVALID instead of silently dropping it
    (ADVICE r05); after the collect the context works as before and the collected frame is the ordinary path's frame."""
    import ctypes as C
    from gaussian_gan_decoder_amd import _capi, rasterizer as R
    from _util import scene_inputs, device_args, same_frame
    dev = torch.device("cuda:0")
    d = scene_inputs(P=8000, size=128, lsm=-4.5, seed=77)
    args = device_args(d, dev)
    ref = R.rasterize_gaussians_native(*args)
    R.rasterize_gaussians_native(*args)                       # (capacity hint for the shape)
    pipe = R.FramePipeline(dev, slots=1)
    assert pipe.submit(*args) is None                          # first frame of the slot's context: rendered synchronously
    assert pipe.submit(*args) is not None                      # collected the first, enqueued the second: now pending
    slot = pipe.slots[0]
    assert slot["pending"] is not None and "result" not in slot["pending"]
    with torch.cuda.stream(slot["stream"]):
        ctx, handle = _capi.context_and_stream(dev)
        with pytest.raises(_capi.RasterError, match="pending"):
            R.rasterize_gaussians_native(*args)                # ggd_forward on the slot's context
        prm = slot["pending"]["prm"]
        Rn = C.c_int64(0)
        rc = ctx.lib.ggd_forward_geometry(ctx.handle, C.c_void_p(handle), C.byref(prm), *([None] * 7), None, None, C.byref(Rn))
        assert rc == -1 and b"pending" in ctx.lib.ggd_last_error(ctx.handle)
    (res,) = pipe.drain()
    res[-1].synchronize()
    assert same_frame(res, ref)
    with torch.cuda.stream(slot["stream"]):
        again = R.rasterize_gaussians_native(*args)
    torch.cuda.synchronize()
    assert same_frame(again, ref)
