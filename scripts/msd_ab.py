"""Frames per second of the forward raster with the depth sort
(GGD_OPT_MSD_SORT = 1) and with the separate histogram launch (0), alternating inside one process."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene
W = {"1M_1024_cube": (1_000_000, 1024, "cube"), "1M_1024_shell": (1_000_000, 1024, "shell"), "100k_512_cube": (100_000, 512, "cube"),
     "500k_512_cube": (500_000, 512, "cube")}
dev = torch.device("cuda:0")
ctx = _capi.context_for(dev)
for name in sys.argv[1:] or ["1M_1024_cube", "1M_1024_shell", "100k_512_cube", "500k_512_cube"]:
    P, S, kind = W[name]
    sc = make_scene(P, S, kind).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    res = {0: [], 1: []}
    ref = None
    for rep in range(3):
        for fold in (1, 0):
            ctx.set_option(_capi.OPT_MSD_SORT, fold)
            for _ in range(20): out = R.rasterize_gaussians_native(*args)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(200): out = R.rasterize_gaussians_native(*args)
            torch.cuda.synchronize()
            res[fold].append(200 / (time.perf_counter() - t))
            if ref is None: ref = out[1].clone()
            assert torch.equal(out[1], ref)
    print(name, "msd=1:", " ".join(f"{v:.0f}" for v in res[1]), "fps | msd=0:", " ".join(f"{v:.0f}" for v in res[0]), "fps")
ctx.set_option(_capi.OPT_MSD_SORT, 1)
