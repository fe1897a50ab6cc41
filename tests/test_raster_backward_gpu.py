"""GPU parity of the backward pass (stages a10 + a11) against the CPU oracle.

Tolerance.  Upstream accumulates the per-(pixel, Gaussian) terms with float atomics in arbitrary order, so the
reference itself is only defined up to fp32 summation error; the oracle sums the same fp32 terms in double.  Every
element of every gradient array must satisfy
    |gpu - oracle| <= ATOL + RTOL * |oracle| + STOL * max(1, max|array|)
with ATOL = 1e-5 (the north_star's absolute bar), RTOL = 1e-4 and STOL = 2e-6: gradients reach 1e3..1e5 in
magnitude (the screen-space terms carry a 0.5*W factor, dL_dcov3D a 1/scale^2 factor), where fp32 has no 1e-5
absolute resolution, and strongly cancelling sums need the array-scale term.
"""
import numpy as np
import pytest
import torch

from _util import scene_inputs, run_oracle, run_native, run_native_backward
from gaussian_gan_decoder_amd.synthetic import make_dL_dpix

pytestmark = pytest.mark.gpu

ATOL, RTOL, STOL = 1e-5, 1e-4, 2e-6
# scale / rotation gradients are derived from dL_dcov3D (magnitude 1e5..1e6) through strongly cancelling sums: the fp32
# oracle itself is only accurate to ~1e-4 * max|array| there (measured against its fp64 twin), so they get a wider
# array-scale term; everything else keeps 2e-6.
DERIVED = ("dL_dscales", "dL_drots", "dL_dcov3D")
STOL_DERIVED = 5e-5

CASES = [
    dict(P=1, size=16, lsm=-3.0),
    dict(P=256, size=64, lsm=-4.0),
    dict(P=4096, size=100, lsm=-5.0, width=100, height=52),
    dict(P=20000, size=256, kind="shell", lsm=-5.5),
    dict(P=20000, size=256, sh_degree=3),
    dict(P=5000, size=128, sh_degree=2, lsm=-5.0),
    dict(P=5000, size=128, use_colors=True, lsm=-5.0),
    dict(P=5000, size=128, use_cov=True, lsm=-5.0, scale_modifier=1.5),
    dict(P=3000, size=64, lsm=-2.0),
    dict(P=100000, size=512),
]


def _ids(c):
    return "-".join(f"{k}{v}" for k, v in c.items())


@pytest.mark.parametrize("case", CASES + ["adversarial"], ids=lambda c: c if isinstance(c, str) else _ids(c))
def test_backward_matches_oracle(native_lib, case):
    from oracle import ggd_oracle as O
    from _util import adversarial_inputs
    d = adversarial_inputs() if case == "adversarial" else scene_inputs(**case)
    g = make_dL_dpix(max(d["W"], d["H"]))[:, :d["H"], :d["W"]].contiguous()
    o = run_oracle(d)
    n = run_native(d, debug=False)
    if not (n["n_contrib"] == o["n_contrib"]).all():
        pytest.skip("forward took a different threshold branch on some pixel (expf ulp); covered by forward test")
    ob = O.backward(o, g.numpy())
    nb = run_native_backward(d, n, g)
    # the adversarial scene holds needles / image-sized splats whose derived gradients are ill-conditioned in fp32: there
    # the fp32 oracle itself is far from its fp64 twin.  Members where it is are judged against the fp64 twin instead
    # (the HIP result must not be further from it than 10x the fp32 oracle's own error).
    ob64 = O.backward(run_oracle(d, dtype=np.float64), g.numpy().astype(np.float64)) if case == "adversarial" else None
    report = []
    worst = 0.0
    for name, ref in ob.items():
        if name == "dL_dconic" or ref is None:
            continue
        if name == "dL_dsh" and d["shs"] is None:
            continue
        if name in ("dL_dscales", "dL_drots") and d["scales"] is None:
            continue
        got = nb[name].reshape(ref.shape)
        diff = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        scale = max(1.0, float(np.abs(ref).max()))
        stol = STOL_DERIVED if name in DERIVED else STOL
        tol = ATOL + RTOL * np.abs(ref) + stol * scale
        ratio = diff / tol
        if ob64 is not None:
            ref64 = ob64[name].reshape(ref.shape)
            own = np.abs(ref.astype(np.float64) - ref64)               # error of the fp32 oracle itself
            rows = own.reshape(own.shape[0], -1)
            ill = (rows > tol.reshape(rows.shape)).any(1)              # per Gaussian: any component ill-conditioned
            own_row = np.broadcast_to(rows.max(1).reshape((-1,) + (1,) * (own.ndim - 1)), own.shape)
            ill_b = np.broadcast_to(ill.reshape((-1,) + (1,) * (own.ndim - 1)), own.shape)
            ratio = np.where(ill_b, np.abs(got.astype(np.float64) - ref64) / (10.0 * own_row + tol), ratio)
        report.append((name, float(diff.max()), scale, float(ratio.max())))
        assert np.isfinite(got).all(), name
        worst = max(worst, float(ratio.max()))
    print("\n" + "\n".join(f"  {n_:13s} max|diff|={m:.3e} scale={s:.3e} worst ratio={f:.3f}" for n_, m, s, f in report))
    assert worst <= 1.0, f"gradient outside tolerance (worst ratio {worst:.2f})"


def test_autograd_api_matches_oracle(native_lib):
    """Through GaussianRasterizer / render_simple exactly as the reference's train step does
    (main/train_pano2gaussian_decoder.py:223-232,263): activations in torch, grads on the raw attributes."""
    from gaussian_gan_decoder_amd.gaussian_model import GaussianModel
    from gaussian_gan_decoder_amd.gaussian_renderer import render_simple
    from gaussian_gan_decoder_amd.synthetic import make_scene
    from oracle import ggd_oracle as O
    import math
    dev = torch.device("cuda:0")
    sc = make_scene(8000, 128, "cube", seed=5, log_scale_mean=-5.0).to(dev)
    pc = GaussianModel(0)
    pc._xyz = sc.xyz.clone().requires_grad_(True)
    pc._scaling = sc.log_scales.clone().requires_grad_(True)
    pc._rotation = sc.rot_raw.clone().requires_grad_(True)
    pc._opacity = sc.opacity_logit.clone().requires_grad_(True)
    pc._features_dc = sc.features_dc.clone().requires_grad_(True)
    out = render_simple(sc.cam, pc, bg_color=sc.bg)
    assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii", "alpha", "depth"}
    g = make_dL_dpix(128).to(dev)
    (out["render"] * g).sum().backward()
    # oracle on the activated values; chain through the activations with torch on the CPU
    cpu = lambda t: t.detach().cpu()
    xyz = cpu(pc._xyz); ls = cpu(pc._scaling).requires_grad_(True); rr = cpu(pc._rotation).requires_grad_(True)
    ol = cpu(pc._opacity).requires_grad_(True)
    scales = torch.exp(ls); rots = torch.nn.functional.normalize(rr); opac = torch.sigmoid(ol)
    cam = sc.cam
    f = O.forward(means3D=xyz.numpy(), opacities=opac.detach().numpy(), shs=cpu(pc._features_dc).numpy(),
                  scales=scales.detach().numpy(), rotations=rots.detach().numpy(),
                  viewmatrix=cpu(cam.world_view_transform).numpy(), projmatrix=cpu(cam.full_proj_transform).numpy(),
                  campos=cpu(cam.camera_center).numpy(), bg=cpu(sc.bg).numpy(), W=128, H=128,
                  tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5))
    b = O.backward(f, cpu(g).numpy())
    np.testing.assert_array_equal(cpu(out["radii"]).numpy(), f["radii"])
    assert np.abs(cpu(out["render"]).numpy() - f["color"]).max() <= 1e-5
    torch.autograd.backward([scales, rots, opac],
                            [torch.from_numpy(b["dL_dscales"]), torch.from_numpy(b["dL_drots"]),
                             torch.from_numpy(b["dL_dopacity"]).view(-1, 1)])

    def close(a, ref, name, stol=STOL):
        a = cpu(a).numpy().astype(np.float64); ref = np.asarray(ref, np.float64).reshape(a.shape)
        diff = np.abs(a - ref)
        ratio = diff / (ATOL + RTOL * np.abs(ref) + stol * max(1.0, np.abs(ref).max()))
        assert ratio.max() <= 1.0, (name, float(diff.max()), float(ratio.max()))
    close(pc._xyz.grad, b["dL_dmeans3D"], "xyz")
    close(pc._scaling.grad, ls.grad.numpy(), "log-scale", STOL_DERIVED)
    close(pc._rotation.grad, rr.grad.numpy(), "rotation", STOL_DERIVED)
    close(pc._opacity.grad, ol.grad.numpy(), "opacity")
    close(pc._features_dc.grad, b["dL_dsh"], "features_dc")
    close(out["viewspace_points"].grad, b["dL_dmeans2D"], "viewspace_points")


def test_fused_activation_prologue_matches_unfused(native_lib):
    """render_simple(fused_activations=True) (sigmoid / exp / normalize inside the kernels) vs the
    reference-shaped path (torch getters + autograd): same image, same gradients on the RAW attributes."""
    from gaussian_gan_decoder_amd.gaussian_model import GaussianModel
    from gaussian_gan_decoder_amd.gaussian_renderer import render_simple
    from gaussian_gan_decoder_amd.synthetic import make_scene
    dev = torch.device("cuda:0")
    sc = make_scene(30000, 256, "shell", seed=9, log_scale_mean=-5.5).to(dev)
    g = make_dL_dpix(256).to(dev)
    res = []
    for fused in (False, True):
        pc = GaussianModel(0)
        pc._xyz = sc.xyz.clone().requires_grad_(True)
        pc._scaling = sc.log_scales.clone().requires_grad_(True)
        pc._rotation = (sc.rot_raw * 1.7).clone().requires_grad_(True)      # deliberately not unit length
        pc._opacity = sc.opacity_logit.clone().requires_grad_(True)
        pc._features_dc = sc.features_dc.clone().requires_grad_(True)
        out = render_simple(sc.cam, pc, bg_color=sc.bg, fused_activations=fused)
        (out["render"] * g).sum().backward()
        res.append((out, pc))
    (o0, p0), (o1, p1) = res
    # torch's exp / sigmoid / norm may differ from the kernel's by an ulp: allow a handful of radius / threshold flips
    assert (o0["radii"] != o1["radii"]).sum().item() <= 3
    err = (o0["render"] - o1["render"]).abs().amax(0)
    assert (err > 1e-5).sum().item() <= 8, float(err.max())
    for name in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc"):
        a, b = getattr(p0, name).grad, getattr(p1, name).grad
        scale = max(1.0, b.abs().max().item())
        bad = ((a - b).abs() > 1e-5 + 1e-3 * b.abs() + 5e-5 * scale).float().mean().item()
        assert bad <= 1e-3, (name, bad)
