"""Stress of the shipped forward at full size under the reference's pose sampler: 1 M-Gaussian cube and shell scenes and a
500 k scene at 1024^2 / 512^2, each under N poses (yaw pi/2 +- 1.0, pitch pi/2 +- 0.3, fov U[5, 17] -- camera.py:6-35,
target_dataloader.py:71).  Every (scene, pose) is rendered ONCE through the exact path (debug = True: two-call form,
duplicateWithKeys + 64-bit radix sort) and reduced to checksums of num_rendered / sorted list / ranges / n_contrib / image; then
`frames` frames in random order through the single-call forward (one context: capacity hint, control blocks, key window see
every switch) and through a 3-slot FramePipeline must reproduce those checksums exactly.
usage: python scripts/stress_poses_big.py [frames] [poses per scene] [seed]"""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussian_gan_decoder_amd import rasterizer as R, _capi
from gaussian_gan_decoder_amd.synthetic import make_scene, make_camera
dev = torch.device("cuda:0")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
nposes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.RandomState(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
e = torch.empty(0, device=dev)


def checksum(res):
    Rn, color, radii, geom, binning, img = res[:6]
    H, W = color.shape[-2:]
    iv = _capi.img_view(W, H)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    lst = binning[:4 * Rn].view(torch.int32).to(torch.int64)
    w = torch.arange(1, Rn + 1, device=dev, dtype=torch.int64) % 1000003
    rg = img[iv.ranges:iv.ranges + 8 * T].view(torch.int32).to(torch.int64)
    nc = img[iv.n_contrib:iv.n_contrib + 4 * W * H].view(torch.int32).to(torch.int64)
    cb = color.view(torch.int32).to(torch.int64).flatten()
    wc = torch.arange(1, cb.numel() + 1, device=dev, dtype=torch.int64) % 999983
    return (Rn, int((lst * w).sum().item()), int((rg * (torch.arange(rg.numel(), device=dev) % 7919 + 1)).sum().item()),
            int(nc.sum().item()), int((cb * wc).sum().item()), int(radii.to(torch.int64).sum().item()))


cases = []
for (P, S, kind, seed) in [(1_000_000, 1024, "cube", 0), (1_000_000, 1024, "shell", 0), (500_000, 512, "cube", 3)]:
    sc = make_scene(P, S, kind, seed=seed).to(dev)
    fixed = (sc.bg, sc.xyz, e, sc.opacities.contiguous(), sc.scales.contiguous(), sc.rotations.contiguous(), 1.0, e)
    shs = sc.features_dc.contiguous()
    for _ in range(nposes):
        h, v, fov = float(math.pi / 2 + rng.uniform(-1, 1)), float(math.pi / 2 + rng.uniform(-0.3, 0.3)), float(rng.uniform(5, 17))
        c = make_camera(S, fov, h, v, device=dev)
        tail = (c.world_view_transform, c.full_proj_transform, math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5), S, S, shs, 0, c.camera_center, False)
        ref = checksum(R.rasterize_gaussians_native(*(fixed + tail + (True,))))
        cases.append((fixed + tail + (False,), ref, (kind, P, S, round(h, 3), round(v, 3), round(fov, 2))))
    print(f"{kind} {P} {S}: {nposes} poses, num_rendered {min(c[1][0] for c in cases[-nposes:])} .. {max(c[1][0] for c in cases[-nposes:])}", flush=True)
ctx = _capi.context_for(dev)
ctx.set_option(_capi.OPT_MSD_SORT, 1)
m0, r0, c0 = ctx.get_option(_capi.STAT_MSD_FRAMES), ctx.get_option(_capi.STAT_SORT_RERUNS), ctx.capacity_retries
t0 = time.perf_counter()
k = 0
for f in range(frames):
    r = rng.rand()
    if f == 0 or r < 0.5:
        k = (k // nposes) * nposes + int(rng.randint(nposes))          # another pose of the same scene (the train step's situation)
    elif r < 0.6:
        k = int(rng.randint(len(cases)))                               # another scene
    got = checksum(R.rasterize_gaussians_native(*cases[k][0]))
    assert got == cases[k][1], (f, cases[k][2], got, cases[k][1])
torch.cuda.synchronize()
print(f"single-call: {frames} frames all equal to the exact path; two-launch sort on {ctx.get_option(_capi.STAT_MSD_FRAMES) - m0}, "
      f"rendered again {ctx.get_option(_capi.STAT_SORT_RERUNS) - r0}, capacity retries {ctx.capacity_retries - c0}; "
      f"{time.perf_counter() - t0:.1f} s", flush=True)
pipe = R.FramePipeline(dev, slots=3)
order, got = [], []
for f in range(frames // 2):
    if rng.rand() < 0.6:
        k = (k // nposes) * nposes + int(rng.randint(nposes))
    elif rng.rand() < 0.3:
        k = int(rng.randint(len(cases)))
    order.append(k)
    res = pipe.submit(*cases[k][0])
    if res is not None:
        res[-1].synchronize(); got.append(checksum(res))
for res in pipe.drain():
    res[-1].synchronize(); got.append(checksum(res))
assert len(got) == len(order)
bad = [i for i, (g, k_) in enumerate(zip(got, order)) if g != cases[k_][1]]
assert not bad, (bad[:5], [cases[order[i]][2] for i in bad[:5]])
stats = []
for s_ in pipe.slots:
    with torch.cuda.stream(s_["stream"]):
        c_ = _capi.context_and_stream(dev)[0]
        stats.append((c_.get_option(_capi.STAT_MSD_FRAMES), c_.get_option(_capi.STAT_SORT_RERUNS), c_.capacity_retries))
print(f"FramePipeline(3): {len(order)} frames all equal; per slot (two-launch frames, rendered again, capacity retries): {stats}; "
      f"synchronous frames {pipe.synchronous_frames}")
