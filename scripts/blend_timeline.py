"""Residency of the forward blend's waves (ggd_blend_stats mode 2): how long the waves live, how unequal they are, how many
are resident per SIMD over the kernel's span, and what an ideal (longest-first) schedule of the same waves would take."""
import sys, math, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device('cuda:0')
for (P, S, kind) in [(1000000, 1024, 'cube'), (1000000, 1024, 'shell'), (100000, 512, 'cube')]:
    sc = make_scene(P, S, kind).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    ctx = _capi.context_for(dev)
    for _ in range(3): R.rasterize_gaussians_native(*args)
    ctx.blend_stats(2)
    ctx.set_profiling(True)
    R.rasterize_gaussians_native(*args)
    torch.cuda.synchronize()
    blend_ms = ctx.stage_times().get('blend')
    ctx.set_profiling(False)
    W = 4 * ((S + 15) // 16) ** 2
    tl = ctx.blend_timeline(W)
    ctx.blend_stats(False)
    t0, t1 = tl[:, 0], tl[:, 1]
    ok = t1 > 0
    dur = (t1 - t0)[ok] / 100.0
    span = (t1[ok].max() - t0[ok].min()) / 100.0
    order = np.argsort(t0[ok])
    # start time (relative) of the 10 longest waves
    longest = np.argsort(-dur)[:10]
    rel_start = (t0[ok][longest] - t0[ok].min()) / 100.0
    visited_frac = tl[ok, 3] / np.maximum(tl[ok, 2], 1)
    print(json.dumps(dict(P=P, S=S, kind=kind, waves=int(ok.sum()), blend_ms_hip_events=round(blend_ms, 4), span_us=round(float(span), 1),
                          wave_us=dict(mean=round(float(dur.mean()), 1), p50=round(float(np.median(dur)), 1), p90=round(float(np.percentile(dur, 90)), 1),
                                       p99=round(float(np.percentile(dur, 99)), 1), max=round(float(dur.max()), 1)),
                          mean_resident_waves_per_simd=round(float(dur.sum() / span / 1024), 2),
                          longest_waves_start_us=[round(float(x), 1) for x in rel_start], longest_waves_us=[round(float(x), 1) for x in dur[longest]],
                          corr_duration_vs_entries_gathered=round(float(np.corrcoef(dur, tl[ok, 3])[0, 1]), 3),
                          visited_frac_mean=round(float(visited_frac.mean()), 3))))
    if kind == 'cube' and S == 1024:
        # coarse map (16 x 16 cells of 4 x 4 tiles): mean wave duration, mean start time; wave b -> tile as ggd_block_to_tile
        b = np.arange(W); q = b >> 3; tile = (q // 4) * 8 + (b & 7); gx = S // 16
        ty, tx = tile // gx, tile % gx
        durs = np.zeros(W); durs[ok] = dur
        starts = (t0 - t0[ok].min()) / 100.0
        for name, val in (("duration_us", durs), ("start_us", starts), ("gathered", tl[:, 3].astype(float)), ("listed", tl[:, 2].astype(float))):
            grid = np.zeros((16, 16)); cnt = np.zeros((16, 16))
            np.add.at(grid, (ty // 4, tx // 4), val); np.add.at(cnt, (ty // 4, tx // 4), 1)
            print(name); print(np.array2string(grid / cnt, precision=0, suppress_small=True, max_line_width=200))
    if kind == 'shell':
        # how a wave's time grows with the rounds it walks, for the waves of the first dispatch round
        first = ok & (t0 - t0[ok].min() < 500)
        rounds = (tl[first, 3] + 63) // 64
        d1 = (t1 - t0)[first] / 100.0
        A = np.stack([np.ones(first.sum()), rounds.astype(float)], 1)
        coef, *_ = np.linalg.lstsq(A, d1, rcond=None)
        heavy = rounds >= 40
        print(json.dumps(dict(first_round_waves=int(first.sum()), fit_us=dict(base=round(float(coef[0]), 1), per_round=round(float(coef[1]), 2)),
                              waves_with_40_or_more_rounds=int(heavy.sum()),
                              their_us_per_round=round(float((d1[heavy] / rounds[heavy]).mean()), 2) if heavy.any() else None,
                              rounds_histogram=np.bincount(np.minimum(rounds // 10, 9).astype(int), minlength=10).tolist())))
