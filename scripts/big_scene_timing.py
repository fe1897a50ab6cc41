"""Forward raster stage times for scenes of 5 - 10 M Gaussians (does anything scale worse than linearly?)."""
import sys, math, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device('cuda:0')
for (P, W, H, kind) in [(5_000_000, 1920, 1080, 'cube'), (5_000_000, 1920, 1080, 'shell'), (10_000_000, 1024, 1024, 'cube')]:
    sc = make_scene(P, max(W, H), kind).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVx*0.5) * H / W, H, W, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    ctx = _capi.context_for(dev)
    for _ in range(3): out = R.rasterize_gaussians_native(*args)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): out = R.rasterize_gaussians_native(*args)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    ctx.set_profiling(True)
    out = R.rasterize_gaussians_native(*args); torch.cuda.synchronize()
    st = {k: round(v * 1e3, 1) for k, v in ctx.stage_times().items()}
    ctx.set_profiling(False)
    print(json.dumps(dict(P=P, W=W, H=H, kind=kind, R=int(out[0]), frame_ms=round(dt * 1e3, 3), stages_us=st)))
    del sc, args, out
