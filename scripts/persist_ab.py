"""A/B of the forward blend's dispatch forms (GGD_OPT_BLEND_PERSIST): one workgroup per (tile, quarter) against persistent
workgroups drawing tickets -- blend stage time (hipEvent pair), whole-frame time, and from the per-wave timeline
(ggd_blend_stats mode 2: start / end of every (tile, quarter) on the 100 MHz clock) the kernel's span and the average number
of quarter waves at work per SIMD over it."""
import sys, math, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device('cuda:0')
for (P, S, kind) in [(1000000, 1024, 'cube'), (1000000, 1024, 'shell'), (100000, 512, 'cube'), (500000, 512, 'cube')]:
    sc = make_scene(P, S, kind).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    ctx = _capi.context_for(dev)
    res, imgs = {}, {}
    for rep in range(2):
        for persist in (0, 1):
            ctx.set_option(_capi.OPT_BLEND_PERSIST, persist)
            for _ in range(5): out = R.rasterize_gaussians_native(*args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100): out = R.rasterize_gaussians_native(*args)
            torch.cuda.synchronize()
            frame_ms = (time.perf_counter() - t0) * 10
            imgs[persist] = out[1]
            ctx.set_profiling(True); ts = []
            for _ in range(30):
                R.rasterize_gaussians_native(*args); ts.append(ctx.stage_times()['blend'])
            ctx.set_profiling(False)
            ctx.blend_stats(2)
            R.rasterize_gaussians_native(*args)
            torch.cuda.synchronize()
            W = 4 * ((S + 15) // 16) ** 2
            tl = ctx.blend_timeline(W)
            ctx.blend_stats(False)
            t0_, t1_ = tl[:, 0], tl[:, 1]
            ok = t1_ > 0
            dur = (t1_ - t0_)[ok] / 100.0
            span = (t1_[ok].max() - t0_[ok].min()) / 100.0
            res.setdefault(persist, []).append(dict(blend_us=round(float(np.median(ts)) * 1e3, 1), frame_ms=round(frame_ms, 4),
                                                    timeline_span_us=round(float(span), 1), wave_us_mean=round(float(dur.mean()), 1),
                                                    wave_us_p99=round(float(np.percentile(dur, 99)), 1),
                                                    mean_busy_waves_per_simd=round(float(dur.sum() / span / 1024), 2)))
    ctx.set_option(_capi.OPT_BLEND_PERSIST, 0)
    assert torch.equal(imgs[0], imgs[1]), "persistent and one-shot forward differ"
    print(json.dumps(dict(P=P, S=S, kind=kind, one_shot=res[0], persistent=res[1])))
