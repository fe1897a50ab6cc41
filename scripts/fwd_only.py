"""Scratch driver for rocprofv3: N forward (+ optional backward) rasters of one synthetic workload.
usage: python scripts/fwd_only.py [workload] [frames] [--backward]"""
import sys, math, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R
from gaussian_gan_decoder_amd.synthetic import make_scene, make_dL_dpix
W = {"1M_1024_cube": (1_000_000, 1024, "cube"), "1M_1024_shell": (1_000_000, 1024, "shell"), "500k_512_cube": (500_000, 512, "cube")}
name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "1M_1024_cube"
frames = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 10
P, S, kind = W[name]
dev = torch.device("cuda:0")
sc = make_scene(P, S, kind).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
        cam.full_proj_transform, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
g = make_dL_dpix(S).to(dev)
for _ in range(frames):
    out = R.rasterize_gaussians_native(*args)
    if "--backward" in sys.argv:
        R.rasterize_gaussians_backward_native(sc.bg, sc.xyz, out[2], e, sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
                                              cam.full_proj_transform, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), g, sc.features_dc.contiguous(), 0,
                                              cam.camera_center, out[3], out[0], out[4], out[5], False)
torch.cuda.synchronize()
print("R", out[0])
