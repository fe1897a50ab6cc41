"""Stage times of the forward raster on grids beyond 64 x 64 tiles (1080p, 2048^2, 4K), every binning path that applies."""
import sys, math, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device('cuda:0')
for (P, W, H) in [(1_000_000, 1920, 1080), (1_000_000, 2048, 2048), (2_000_000, 3840, 2160)]:
    sc = make_scene(P, max(W, H), 'cube').to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVx*0.5) * H / W, H, W, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    ctx = _capi.context_for(dev)
    for mode, name in ((1, "auto"), (0, "sort"), (3, "rowbin")):
        ctx.set_option(_capi.OPT_BINNING, mode)
        try:
            for _ in range(3): out = R.rasterize_gaussians_native(*args)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(30): out = R.rasterize_gaussians_native(*args)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 30
            ctx.set_profiling(True)
            out = R.rasterize_gaussians_native(*args); torch.cuda.synchronize()
            st = {k: round(v * 1e3, 1) for k, v in ctx.stage_times().items()}
            ctx.set_profiling(False)
            print(json.dumps(dict(P=P, W=W, H=H, binning=name, R=int(out[0]), frame_ms=round(dt * 1e3, 3), fps=round(1 / dt, 1), stages_us=st)))
        except Exception as ex:
            print(json.dumps(dict(P=P, W=W, H=H, binning=name, error=str(ex)[:120])))
    ctx.set_option(_capi.OPT_BINNING, 1)
