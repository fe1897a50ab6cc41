"""A/B of GGD_OPT_BLEND_SPLIT (waves per tile) for the forward blend: stage time per workload."""
import sys, math, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device('cuda:0')
for (P, S, kind) in [(1000000, 1024, 'cube'), (1000000, 1024, 'shell'), (100000, 512, 'cube'), (500000, 512, 'cube')]:
    sc = make_scene(P, S, kind).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    ctx = _capi.context_for(dev)
    res = {}
    for split in (0, 2, 3):
        ctx.set_option(_capi.OPT_BLEND_SPLIT, split)
        for _ in range(3): R.rasterize_gaussians_native(*args)
        ctx.set_profiling(True); ts = []
        for _ in range(20):
            R.rasterize_gaussians_native(*args); ts.append(ctx.stage_times()['blend'])
        ctx.set_profiling(False)
        ctx.blend_stats(True); R.rasterize_gaussians_native(*args); st = ctx.blend_stats(False)
        res[split] = dict(blend_us=round(float(np.median(ts)) * 1e3, 1), visited=st['visited'], culled=st['culled'], lanes=st['lanes'], pixels=st['pixels'])
    print(json.dumps(dict(P=P, S=S, kind=kind, **{f"split{k}": v for k, v in res.items()})))
