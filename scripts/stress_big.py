"""Stress: 8 M Gaussians @ 1024^2 (R ~ 33 M) -- both tile-binning paths must agree bit for bit, timing reported."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device('cuda:0')
P, S = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000, 1024
sc = make_scene(P, S, 'cube').to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
        cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
ctx = _capi.context_for(dev)
outs = {}
for path in (2, 3):
    ctx.set_option(_capi.OPT_BINNING, path)
    for _ in range(3): out = R.rasterize_gaussians_native(*args)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): out = R.rasterize_gaussians_native(*args)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    outs[path] = out
    print(f"path {path}: R={out[0]} {dt*1e3:.3f} ms/frame")
a, b = outs[2], outs[3]
assert a[0] == b[0]
assert torch.equal(a[1], b[1]), "images differ"
lr = a[0]
bv = _capi.binning_view(lr)
la = a[4][bv.list:bv.list + 4 * a[0]]; lb = b[4][bv.list:bv.list + 4 * a[0]]
assert torch.equal(la, lb), "lists differ"
print("paths agree; image finite:", bool(torch.isfinite(a[1]).all()))
