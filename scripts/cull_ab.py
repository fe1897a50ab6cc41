"""A/B of the blend pre-cull forms (GGD_OPT_BLEND_CULL: 1 = pixel grid, 2 = rectangle, 0 = none) in ONE process: forward /
backward blend stage times, the forward's work counters, and bit-identity of the image / n_contrib across the forms."""
import sys, math, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene, make_dL_dpix
dev = torch.device('cuda:0')
scenes = [(1000000, 1024, 'cube'), (1000000, 1024, 'shell'), (100000, 512, 'cube'), (500000, 512, 'shell')]
if len(sys.argv) > 1: scenes = scenes[:int(sys.argv[1])]
modes = [int(m) for m in os.environ.get('CULL_MODES', '2,1').split(',')]
for (P, S, kind) in scenes:
    sc = make_scene(P, S, kind).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    tanx, tany = math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5)
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, tanx, tany, S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    g = make_dL_dpix(S).to(dev)
    ctx = _capi.context_for(dev)
    ref = None
    for cull in modes:
        ctx.set_option(_capi.OPT_BLEND_CULL, cull)
        for _ in range(3): out = R.rasterize_gaussians_native(*args)
        ctx.set_profiling(True)
        tf = []
        for _ in range(20):
            out = R.rasterize_gaussians_native(*args); torch.cuda.synchronize(); tf.append(ctx.stage_times()['blend'])
        bargs = (sc.bg, sc.xyz, out[2], e, sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform, cam.full_proj_transform,
                 tanx, tany, g, sc.features_dc.contiguous(), 0, cam.camera_center, out[3], out[0], out[4], out[5], False)
        tb = []
        for _ in range(12):
            gr = R.rasterize_gaussians_backward_native(*bargs); torch.cuda.synchronize(); st = ctx.stage_times()
            tb.append(st.get('blend_bwd', st.get('blend_backward', 0.0)))
        ctx.set_profiling(False)
        ctx.blend_stats(True)
        R.rasterize_gaussians_native(*args)
        bs = ctx.blend_stats(False)
        img = out[1].cpu().numpy()
        grads = [t.detach().cpu().numpy() for t in gr if isinstance(t, torch.Tensor) and t.numel() > 0]
        row = dict(P=P, S=S, kind=kind, cull=cull, fwd_blend_ms=round(float(np.median(tf)), 4), bwd_blend_ms=round(float(np.median(tb[2:])), 4),
                   visited=bs['visited'], culled=bs['culled'], culled_in_loop=bs['culled_in_loop'], updated=bs['visited'] - bs['culled'])
        if ref is None:
            ref = (img, grads)
        else:
            row['image_bit_identical'] = bool((img == ref[0]).all())
            row['grad_max_rel_diff'] = float(max(np.abs(a - b).max() / max(1e-30, np.abs(b).max()) for a, b in zip(grads, ref[1])))
        row['stage_keys'] = sorted(st.keys()) if cull == modes[0] and kind == scenes[0][2] else None
        print(json.dumps(row), flush=True)
    ctx.set_option(_capi.OPT_BLEND_CULL, 1)
