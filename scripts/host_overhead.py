import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cProfile, pstats
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device('cuda:0')
def args_for(P, S):
    sc = make_scene(P, S, 'cube').to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    return (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
for (P, S) in [(2000000, 128), (1000000, 1024)]:
    a = args_for(P, S)
    for _ in range(20): R.rasterize_gaussians_native(*a)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(300): R.rasterize_gaussians_native(*a)
    torch.cuda.synchronize(); print(P, S, "us/call", (time.perf_counter() - t) / 300 * 1e6)
a = args_for(1000000, 1024)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): R.rasterize_gaussians_native(*a)
pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(14)
