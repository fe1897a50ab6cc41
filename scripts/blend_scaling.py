"""Blend forward time per tile as the grid grows at constant tile-list length (1 M Gaussians at 1024^2 vs 4 M at 2048^2 with
half the scale): separates the kernel's steady-state rate from its ramp / tail."""
import sys, math, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device('cuda:0')
for (P, S, mod) in [(1_000_000, 1024, 1.0), (4_000_000, 2048, 0.5), (250_000, 512, 2.0)]:
    sc = make_scene(P, S, 'cube').to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), mod, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    ctx = _capi.context_for(dev)
    for _ in range(3): out = R.rasterize_gaussians_native(*args)
    ctx.set_profiling(True)
    acc = {}
    for _ in range(20):
        out = R.rasterize_gaussians_native(*args)
        torch.cuda.synchronize()
        for k, v in ctx.stage_times().items(): acc[k] = acc.get(k, 0.0) + v / 20
    ctx.set_profiling(False)
    T = ((S + 15) // 16) ** 2
    print(json.dumps(dict(P=P, S=S, scale_modifier=mod, R=int(out[0]), list_mean=round(out[0] / T, 1), tiles=T,
                          blend_us=round(acc['blend'] * 1e3, 1), blend_ns_per_tile=round(acc['blend'] * 1e6 / T, 2),
                          stages_us={k: round(v * 1e3, 1) for k, v in acc.items()})))
