"""Soak test of the single-call forward: thousands of frames alternating between scenes of different sizes on one stream, with
the returned num_rendered and a checksum of the image checked every frame against the first rendering of that scene."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device('cuda:0')
def args_for(P, S, kind, seed, H=None):
    sc = make_scene(P, S, kind, seed=seed).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    H = H or S
    tx = math.tan(cam.FoVx*0.5)
    return (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, tx, tx * H / S, H, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
scenes = [args_for(200000, 512, 'cube', 1), args_for(50000, 256, 'shell', 2), args_for(1000000, 1024, 'cube', 0), args_for(3000, 128, 'cube', 3),
          args_for(300000, 1920, 'cube', 4, H=1080), args_for(200000, 2048, 'shell', 5)]   # the last two: grids wider than 64 tiles
# a scene whose depths span several binades (the depth keys' top byte varies: all four sort passes rank) -- after a streak of the
# others the fourth pass is not launched, and this scene's frame has to be detected and rendered again
deep = list(args_for(100000, 512, 'cube', 6))
g = torch.Generator().manual_seed(7)
deep[1] = (deep[1] * torch.exp(3.0 * torch.rand(100000, 1, generator=g)).to(dev)).contiguous()
scenes.append(tuple(deep))
ref = []
for a in scenes:
    for _ in range(2): out = R.rasterize_gaussians_native(*a)
    torch.cuda.synchronize(); ref.append((out[0], out[1].double().sum().item(), out[1].clone()))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
t = time.perf_counter(); bad = 0; pend = []
for it in range(n):
    # 40 frames cycling through the scenes, then 40 frames of one scene (a streak long enough to drop the fourth sort pass)
    k = (it * 7 + it // 5) % len(scenes) if (it // 40) % 2 == 0 else (it // 80) % len(scenes)
    out = R.rasterize_gaussians_native(*scenes[k])
    if out[0] != ref[k][0]: bad += 1
    pend.append((k, out[1]))
    if len(pend) == 16:      # image checks in batches, so that most frames run without an intervening sync
        for kk, img in pend:
            if not torch.equal(img, ref[kk][2]): bad += 1
        pend = []
torch.cuda.synchronize()
from gaussian_gan_decoder_amd import _capi
cx = _capi.context_for(dev)
print("frames", n, "mismatches", bad, "seconds", round(time.perf_counter() - t, 2), "sort reruns", cx.get_option(_capi.STAT_SORT_RERUNS),
      "capacity retries", cx.capacity_retries)
sys.exit(1 if bad else 0)
