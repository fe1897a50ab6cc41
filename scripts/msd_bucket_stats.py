"""Sizes of the two-launch depth sort's buckets (key bits 14..23 of the visible Gaussians' depths) on the bench scenes.
usage: python scripts/msd_bucket_stats.py [workload ...]"""
import sys, math, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R
from gaussian_gan_decoder_amd.synthetic import make_scene
W = {"1M_1024_cube": (1_000_000, 1024, "cube"), "1M_1024_shell": (1_000_000, 1024, "shell"), "500k_512_cube": (500_000, 512, "cube"),
     "100k_512_cube": (100_000, 512, "cube"), "2M_2160_cube": (2_000_000, 2160, "cube")}
dev = torch.device("cuda:0")
for name in (sys.argv[1:] or ["1M_1024_cube", "1M_1024_shell", "500k_512_cube", "100k_512_cube"]):
    P, S, kind = W[name]
    sc = make_scene(P, S, kind).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    out = R.rasterize_gaussians_native(sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e,
                                       cam.world_view_transform, cam.full_proj_transform, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), S, S,
                                       sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    radii = out[2]
    depth = (sc.xyz @ cam.world_view_transform[:3, 2] + cam.world_view_transform[3, 2]).float()
    keys = depth[radii > 0].contiguous().view(torch.int32).long() & 0xffffffff
    b = torch.bincount(((keys >> 14) & 1023), minlength=1024)
    top = torch.unique(keys >> 24)
    nz = b[b > 0]
    q = torch.quantile(nz.float(), torch.tensor([0.5, 0.9, 0.99], device=dev)).tolist()
    print(f"{name}: visible {keys.numel()}  top bytes {top.tolist()}  non-empty buckets {nz.numel()}  max {int(b.max())}  median {q[0]:.0f}  p90 {q[1]:.0f}  p99 {q[2]:.0f}"
          f"  > 2048: {int((b > 2048).sum())}  > 4096: {int((b > 4096).sum())}  > 6144: {int((b > 6144).sum())}")
