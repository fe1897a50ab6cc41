"""A/B of GGD_OPT_BLEND_SPLIT for the backward blend: stage time per workload."""
import sys, math, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene, make_dL_dpix
dev = torch.device('cuda:0')
for (P, S, kind) in [(1000000, 1024, 'cube'), (1000000, 1024, 'shell'), (500000, 512, 'cube')]:
    sc = make_scene(P, S, kind).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    ctx = _capi.context_for(dev)
    g = make_dL_dpix(S).to(dev)
    out = R.rasterize_gaussians_native(*args)
    bargs = (sc.bg, sc.xyz, out[2], e, sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform, cam.full_proj_transform,
             math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), g, sc.features_dc.contiguous(), 0, cam.camera_center, out[3], out[0], out[4], out[5], False)
    res = {}
    for split in [int(a) for a in (sys.argv[1:] or ["0", "2"])]:
        ctx.set_option(_capi.OPT_BLEND_SPLIT, split)
        for _ in range(3): R.rasterize_gaussians_backward_native(*bargs)
        ctx.set_profiling(True); ts = []
        for _ in range(15):
            R.rasterize_gaussians_backward_native(*bargs); ts.append(ctx.stage_times()['blend_bwd'])
        ctx.set_profiling(False)
        res[split] = round(float(np.median(ts)) * 1e3, 1)
    ctx.set_option(_capi.OPT_BLEND_SPLIT, 1)
    print(json.dumps(dict(P=P, S=S, kind=kind, bwd_us=res)))
