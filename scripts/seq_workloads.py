import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device('cuda:0')
def run(P, S, n=50):
    sc = make_scene(P, S, 'cube').to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    for _ in range(10): R.rasterize_gaussians_native(*args)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): o = R.rasterize_gaussians_native(*args)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    ctx = _capi.context_for(dev); ctx.set_profiling(True); R.rasterize_gaussians_native(*args); st = ctx.stage_times(); ctx.set_profiling(False)
    print(P, S, f"{dt*1e3:.4f} ms", o[0], {k: round(v, 4) for k, v in st.items()}, flush=True)
order = [tuple(map(int, a.split('x'))) for a in sys.argv[1:]]
for P, S in order: run(P, S)
