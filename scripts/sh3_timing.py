"""Forward + backward stage times with SH degree 0 vs 3 (stock 3DGS scenes carry 16 coefficients per channel)."""
import sys, math, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene, make_dL_dpix
dev = torch.device('cuda:0')
P, S = 1_000_000, 1024
sc = make_scene(P, S, 'cube').to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
g = make_dL_dpix(S).to(dev)
for deg in (0, 3):
    M = (deg + 1) ** 2
    shs = torch.zeros(P, M, 3, device=dev); shs[:, :1] = sc.features_dc
    if M > 1: shs[:, 1:] = 0.05 * torch.randn(P, M - 1, 3, device=dev)
    shs = shs.contiguous()
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), S, S, shs, deg, cam.camera_center, False, False)
    ctx = _capi.context_for(dev)
    for _ in range(3): out = R.rasterize_gaussians_native(*args)
    bargs = (sc.bg, sc.xyz, out[2], e, sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
             cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), g, shs, deg, cam.camera_center, out[3], out[0], out[4], out[5], False)
    for _ in range(3): R.rasterize_gaussians_backward_native(*bargs)
    ctx.set_profiling(True)
    acc = {}
    for _ in range(10):
        out = R.rasterize_gaussians_native(*args); torch.cuda.synchronize()
        for k, v in ctx.stage_times().items():
            if not k.endswith('_bwd'): acc[k] = acc.get(k, 0.0) + v / 10
        R.rasterize_gaussians_backward_native(*bargs); torch.cuda.synchronize()
        for k in ("blend_bwd", "preprocess_bwd"): acc[k] = acc.get(k, 0.0) + ctx.stage_times()[k] / 10
    ctx.set_profiling(False)
    print(json.dumps(dict(sh_degree=deg, stages_us={k: round(v * 1e3, 1) for k, v in acc.items()})))
