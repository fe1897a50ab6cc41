import sys, math, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device('cuda:0')
P, S, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
sc = make_scene(P, S, kind).to(dev); cam = sc.cam
e = torch.empty(0, device=dev)
args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
        cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
for _ in range(5): R.rasterize_gaussians_native(*args)
torch.cuda.synchronize()
ts = []
for i in range(20):
    t0 = time.perf_counter(); out = R.rasterize_gaussians_native(*args); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((round((t1-t0)*1e3,3), round((t2-t1)*1e3,3)))
print(P, S, kind, 'R', out[0], ts)
