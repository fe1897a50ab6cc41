"""Stage times of the forward only up to the binning (no blend): used to A/B depth-sort variants whose results may be wrong."""
import sys, os, math, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device('cuda:0')
sc = make_scene(1000000, 1024, 'cube').to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
        cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), 1024, 1024, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
ctx = _capi.context_for(dev)
for _ in range(3): R.rasterize_gaussians_native(*args)
ctx.set_profiling(True); acc = {}
for _ in range(20):
    R.rasterize_gaussians_native(*args)
    for k, v in ctx.stage_times().items(): acc.setdefault(k, []).append(v)
ctx.set_profiling(False)
print(json.dumps({k: round(float(np.median(v)) * 1e3, 1) for k, v in acc.items()}))
