import sys, os, math, time
sys.path.insert(0, '/root/repo')
import torch
from gaussian_gan_decoder_amd import rasterizer as R
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device("cuda:0")
for (P, S, kind) in ((100_000, 512, "cube"), (1_000_000, 1024, "cube")):
    sc = make_scene(P, S, kind).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    for n in (1, 2, 4, 8):
        pipe = R.FramePipeline(dev, slots=n)
        for _ in range(6 * n): pipe.submit(*args)
        pipe.drain(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(400): pipe.submit(*args)
        th = time.perf_counter() - t
        pipe.drain(); torch.cuda.synchronize()
        t = time.perf_counter() - t
        print(P, S, "slots", n, "fps", round(400 / t), "host-side submit loop us/frame", round(th / 400 * 1e6, 1))
