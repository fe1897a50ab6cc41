"""Does the backward blend cost the same right after a forward (training) as in a loop of backwards (bench --backward)?"""
import sys, math, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene, make_dL_dpix
dev = torch.device('cuda:0')
P, S = 1_000_000, 1024
sc = make_scene(P, S, 'cube', seed=0).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
g = make_dL_dpix(S).to(dev)
shs = sc.features_dc.contiguous()
args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
        cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), S, S, shs, 0, cam.camera_center, False, False)
ctx = _capi.context_for(dev)
for _ in range(3): out = R.rasterize_gaussians_native(*args)
def bargs(out):
    return (sc.bg, sc.xyz, out[2], e, sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), g, shs, 0, cam.camera_center, out[3], out[0], out[4], out[5], False)
for _ in range(3): R.rasterize_gaussians_backward_native(*bargs(out))
ctx.set_profiling(True)
res = {}
for mode in ("backward only", "forward then backward", "forward, sync, backward"):
    acc = 0.0
    for _ in range(10):
        if mode != "backward only":
            out = R.rasterize_gaussians_native(*args)
            if mode.endswith("sync, backward"): torch.cuda.synchronize()
        R.rasterize_gaussians_backward_native(*bargs(out)); torch.cuda.synchronize()
        acc += ctx.stage_times()["blend_bwd"] / 10
    res[mode] = round(acc * 1e3, 1)
print(json.dumps(res))
