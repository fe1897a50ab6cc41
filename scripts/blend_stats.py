import sys, math, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device('cuda:0')
for (P, S, kind) in [(1000000, 1024, 'cube'), (1000000, 1024, 'shell'), (100000, 512, 'cube'), (500000, 512, 'shell')]:
    sc = make_scene(P, S, kind).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    ctx = _capi.context_for(dev)
    R.rasterize_gaussians_native(*args)
    ctx.blend_stats(True)
    R.rasterize_gaussians_native(*args)
    st = ctx.blend_stats(False)
    v = max(st['visited'], 1); nc = max(v - st['culled'], 1)
    print(json.dumps(dict(P=P, S=S, kind=kind, **st, visited_frac=round(st['visited']/max(st['listed'],1), 3), culled_frac=round(st['culled']/v, 3),
                          lanes_per_live_record=round(st['lanes']/nc, 1), pixels_per_live_record=round(st['pixels']/nc, 1),
                          span_us=st['span_ticks'] / 100.0, mean_wave_us=round(st['wave_ticks_sum'] / 100.0 / (4 * ((S + 15) // 16) ** 2), 2),
                          longest_wave_us=st['wave_ticks_max'] / 100.0,
                          mean_resident_waves_per_simd=round(st['wave_ticks_sum'] / max(st['span_ticks'], 1) / 1024, 2))))
