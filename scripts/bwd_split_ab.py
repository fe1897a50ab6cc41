"""Backward blend: tile form (GGD_OPT_BLEND_SPLIT = 3) against quarter form (4) below the auto rule's 2048 tiles -- which scenes
the rule should send to the quarter form.  hipEvent stage times, 10 launches each."""
import sys, math, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene, make_dL_dpix
dev = torch.device("cuda:0")
ctx = _capi.context_for(dev)
for (P, S, kind, fov) in [(100_000, 512, "cube", 12.0), (500_000, 512, "cube", 12.0), (1_000_000, 512, "cube", 12.0), (500_000, 512, "shell", 12.0),
                          (500_000, 512, "shell", 6.0), (500_000, 512, "shell", 16.0), (100_000, 256, "cube", 12.0), (30_000, 512, "cube", 12.0),
                          (1_000_000, 704, "cube", 12.0), (200_000, 640, "shell", 12.0)]:
    sc = make_scene(P, S, kind, fov_deg=fov).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    sca, rot, shs = sc.scales.contiguous(), sc.rotations.contiguous(), sc.features_dc.contiguous()
    tx, ty = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    out = R.rasterize_gaussians_native(sc.bg, sc.xyz, e, sc.opacities.contiguous(), sca, rot, 1.0, e, cam.world_view_transform,
                                       cam.full_proj_transform, tx, ty, S, S, shs, 0, cam.camera_center, False, False)
    g = make_dL_dpix(S).to(dev)
    bargs = (sc.bg, sc.xyz, out[2], e, sca, rot, 1.0, e, cam.world_view_transform, cam.full_proj_transform, tx, ty, g, shs, 0,
             cam.camera_center, out[3], out[0], out[4], out[5], False)
    T = ((S + 15) // 16) ** 2
    res = {}
    for split in (3, 4, 3, 4):
        ctx.set_option(_capi.OPT_BLEND_SPLIT, split)
        for _ in range(2): R.rasterize_gaussians_backward_native(*bargs)
        ctx.set_profiling(True); t = 0.0
        for _ in range(10):
            R.rasterize_gaussians_backward_native(*bargs); t += ctx.stage_times()["blend_bwd"] / 10
        ctx.set_profiling(False)
        res.setdefault(split, []).append(round(t * 1e3, 1))
    ctx.set_option(_capi.OPT_BLEND_SPLIT, 1)
    print(f"{kind:5s} P={P:8d} {S}x{S} fov {fov:4.1f}: tiles {T:5d} R {out[0]:9d} R/T {out[0] / T:7.0f}  tile form {res[3]} us  quarter form {res[4]} us", flush=True)
