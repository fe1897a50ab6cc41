"""Scratch: time the forward/backward stages on synthetic scenes (not the bench)."""
import sys, math, time, json
sys.path.insert(0, '.')
import torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene, make_dL_dpix
dev = torch.device('cuda:0')
for (P, S, kind) in [(100000, 512, 'cube'), (1000000, 1024, 'cube'), (1000000, 1024, 'shell'), (500000, 512, 'cube')]:
    sc = make_scene(P, S, kind).to(dev)
    cam = sc.cam
    args = (sc.bg, sc.xyz, torch.empty(0, device=dev), sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0,
            torch.empty(0, device=dev), cam.world_view_transform, cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5),
            S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    ctx = _capi.context_for(dev)
    for _ in range(5): out = R.rasterize_gaussians_native(*args)
    torch.cuda.synchronize()
    t = time.perf_counter()
    N = 30
    for _ in range(N): out = R.rasterize_gaussians_native(*args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / N
    ctx.set_profiling(True)
    out = R.rasterize_gaussians_native(*args)
    torch.cuda.synchronize()
    st = ctx.stage_times()
    g = make_dL_dpix(S).to(dev)
    bargs = (sc.bg, sc.xyz, out[2], torch.empty(0, device=dev), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, torch.empty(0, device=dev),
             cam.world_view_transform, cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), g, sc.features_dc.contiguous(), 0, cam.camera_center,
             out[3], out[0], out[4], out[5], False)
    for _ in range(3): R.rasterize_gaussians_backward_native(*bargs)
    torch.cuda.synchronize()
    st2 = ctx.stage_times()
    ctx.set_profiling(False)
    t = time.perf_counter()
    for _ in range(N): R.rasterize_gaussians_backward_native(*bargs)
    torch.cuda.synchronize()
    dtb = (time.perf_counter() - t) / N
    print(json.dumps(dict(P=P, S=S, kind=kind, R=out[0], fwd_ms=dt*1e3, fps=1/dt, bwd_ms=dtb*1e3, stages={k: round(v, 4) for k, v in st2.items()})))
