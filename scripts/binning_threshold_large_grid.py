import sys, os, math, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device('cuda:0')
for (P, S) in [(20000, 1536), (100000, 1536), (300000, 2048), (1000000, 2048)]:
    sc = make_scene(P, S, 'cube').to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    a = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
         cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    ctx = _capi.context_for(dev); res = {}
    for mode in (0, 2, 1):
        ctx.set_option(_capi.OPT_BINNING, mode)
        for _ in range(10): out = R.rasterize_gaussians_native(*a)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(100): out = R.rasterize_gaussians_native(*a)
        torch.cuda.synchronize(); res[mode] = round((time.perf_counter() - t) / 100 * 1e6, 1)
    ctx.set_option(_capi.OPT_BINNING, 1)
    print(json.dumps(dict(P=P, S=S, R=out[0], us_per_frame=res)))
