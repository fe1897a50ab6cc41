"""Throughput of the forward raster with several frames in flight (one HIP stream + one ggd context per frame slot)."""
import sys, os, math, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import rasterizer as R
from gaussian_gan_decoder_amd.synthetic import make_scene
dev = torch.device("cuda:0")
for (P, S, kind) in [(1_000_000, 1024, "cube"), (1_000_000, 1024, "shell"), (100_000, 512, "cube")]:
    sc = make_scene(P, S, kind, seed=0).to(dev); cam = sc.cam; e = torch.empty(0, device=dev)
    args = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), S, S, sc.features_dc.contiguous(), 0, cam.camera_center, False, False)
    res = {}
    ref = None
    for nstreams in (1, 2, 3, 4):
        streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
        for s in streams: s.wait_stream(torch.cuda.current_stream(dev))
        outs = [None] * nstreams
        for i in range(4 * nstreams):
            with torch.cuda.stream(streams[i % nstreams]): outs[i % nstreams] = R.rasterize_gaussians_native(*args)
        torch.cuda.synchronize(dev)
        N = 300
        t = time.perf_counter()
        for i in range(N):
            with torch.cuda.stream(streams[i % nstreams]): outs[i % nstreams] = R.rasterize_gaussians_native(*args)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t) / N
        res[nstreams] = round(1.0 / dt, 1)
        if ref is None: ref = outs[0][1].clone()
        assert all(torch.equal(o[1], ref) for o in outs), "images differ between streams"
    print(json.dumps(dict(P=P, S=S, kind=kind, fps_by_frames_in_flight=res)))
