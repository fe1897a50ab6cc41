"""Work counters of the backward blend (quarter form) next to the forward's, 1 M / 1024^2 cube and shell:
what the backward walks, stages, blends and flushes -- the numbers behind "VALU-bound" and the size of what a survivor list
handed over by the forward could save.    python scripts/bwd_blend_stats.py > profiles/r06/backward_blend_counters.txt"""
import sys, math, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene, make_dL_dpix
dev = torch.device('cuda:0')
for (P, S, kind) in [(1000000, 1024, 'cube'), (1000000, 1024, 'shell')]:
    sc = make_scene(P, S, kind).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    sca, rot = sc.scales.contiguous(), sc.rotations.contiguous()
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sca, rot, 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    g = make_dL_dpix(S).to(dev)
    ctx = _capi.context_for(dev)
    out = R.rasterize_gaussians_native(*args)
    bargs = (sc.bg, sc.xyz, out[2], e, sca, rot, 1.0, e, cam.world_view_transform, cam.full_proj_transform, math.tan(cam.FoVx*0.5),
             math.tan(cam.FoVy*0.5), g, sc.features_dc.contiguous(), 0, cam.camera_center, out[3], out[0], out[4], out[5], False)
    R.rasterize_gaussians_backward_native(*bargs)
    ctx.blend_stats(True)
    out = R.rasterize_gaussians_native(*args)
    R.rasterize_gaussians_backward_native(*bargs)
    b = ctx.blend_backward_stats()
    f = ctx.blend_stats(False)
    # timings without the counters
    ctx.set_profiling(True)
    tf = tb = 0.0
    for _ in range(10):
        out = R.rasterize_gaussians_native(*args); tf += ctx.stage_times()["blend"] / 10
        R.rasterize_gaussians_backward_native(*bargs); tb += ctx.stage_times()["blend_bwd"] / 10
    ctx.set_profiling(False)
    R_ = out[0]
    fwd_used = f["visited"] - f["culled"]          # staged records at least one pixel of the quarter wave blended (any != 0)
    print(json.dumps(dict(
        scene=f"{P} Gaussians, {S}x{S}, {kind}", num_rendered=R_, quarter_records=4 * R_,
        forward=dict(**f, gathered=f["visited"], staged=f["visited"] - f["culled"] + f["culled_in_loop"], used_by_some_pixel=fwd_used,
                     kernel_us=round(tf * 1e3, 1)),
        backward=dict(**b, kernel_us=round(tb * 1e3, 1),
                      walked_frac_of_quarter_records=round(b["walked"] / (4 * R_), 3),
                      staged_frac_of_walked=round(b["staged"] / max(b["walked"], 1), 3),
                      blended_frac_of_staged=round(b["blended"] / max(b["staged"], 1), 3),
                      live_lanes_per_blended=round(b["live_lanes"] / max(b["blended"], 1), 1),
                      spans_per_blended=round(b["atomic_spans"] / max(b["blended"], 1), 3),
                      walked_per_round=round(b["walked"] / max(b["rounds"], 1), 1),
                      staged_per_round=round(b["staged"] / max(b["rounds"], 1), 1),
                      valu_wave_insts_per_blended_record="1.76e8 (PMC, profiles/r05/traffic.json) / blended = %.0f" % (1.76e8 / max(b["blended"], 1)) if kind == "cube" else None),
        survivor_list=dict(entries=fwd_used, bytes=4 * fwd_used,
                           backward_gather_rounds_now=b["rounds"], rounds_with_the_list=math.ceil(fwd_used / 64),
                           note="the backward re-runs gather + pre-cull over `walked` list entries to stage `staged` records; with the forward's "
                                "survivor list it would gather `entries` records directly"))))
