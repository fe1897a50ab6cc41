"""A/B the blend variants (exp mode x cull) in ONE process: blend stage time + parity vs the CPU oracle."""
import sys, math, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from _util import decode_buffers
from oracle import ggd_oracle as O
dev = torch.device('cuda:0')
scenes = [(1000000, 1024, 'cube'), (1000000, 1024, 'shell'), (100000, 512, 'cube')]
if len(sys.argv) > 1: scenes = scenes[:int(sys.argv[1])]
for (P, S, kind) in scenes:
    sc_cpu = make_scene(P, S, kind); sc = sc_cpu.to(dev); cam = sc.cam
    tanx, tany = math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5)
    e = torch.empty(0, device=dev)
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, tanx, tany, S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    c = sc_cpu.cam
    f = O.forward(means3D=sc_cpu.xyz.numpy(), opacities=sc_cpu.opacities.numpy(), shs=sc_cpu.features_dc.numpy(), scales=sc_cpu.scales.numpy(),
                  rotations=sc_cpu.rotations.numpy(), viewmatrix=c.world_view_transform.numpy(), projmatrix=c.full_proj_transform.numpy(),
                  campos=c.camera_center.numpy(), bg=sc_cpu.bg.numpy(), W=S, H=S, tanfovx=tanx, tanfovy=tany)
    ctx = _capi.context_for(dev)
    for cull in (0, 1):
        for em in (0, 1, 2):
            ctx.set_option(_capi.OPT_EXP_MODE, em); ctx.set_option(_capi.OPT_BLEND_CULL, cull)
            for _ in range(3): out = R.rasterize_gaussians_native(*args)
            ctx.set_profiling(True)
            ts = []
            for _ in range(20):
                out = R.rasterize_gaussians_native(*args); ts.append(ctx.stage_times()['blend'])
            ctx.set_profiling(False)
            torch.cuda.synchronize()
            d = decode_buffers(P, S, S, out[0], out[3], out[4], out[5])
            color = out[1].cpu().numpy()
            same = d['n_contrib'] == f['n_contrib']
            err = np.abs(color - f['color'])
            print(json.dumps(dict(P=P, S=S, kind=kind, cull=cull, exp_mode=em, blend_ms=round(float(np.median(ts)), 4),
                                  flips=int((~same).sum()), max_err_same=float(err[:, same].max()), max_err_all=float(err.max()),
                                  n_gt_1e5=int((err.max(0) > 1e-5).sum()))), flush=True)
    ctx.set_option(_capi.OPT_EXP_MODE, 0); ctx.set_option(_capi.OPT_BLEND_CULL, 1)
