"""Stress of the speculative front end: a long random sequence of frames over the multi-tile sort scenes of tests/test_fuzz_gpu.py
(all / part on screen, ordered inputs, dense and oversized buckets) and small scenes, the two-launch sort on; every frame's
num_rendered / list / ranges against the oracle's.  usage: python scripts/stress_sort.py [frames] [seed]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from _util import run_oracle, run_native, scene_inputs
import test_fuzz_gpu as T
from gaussian_gan_decoder_amd import _capi
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
scenes = [T._sort_case(s)[0] for s in range(24)] + [scene_inputs(P=P, size=S, seed=90 + i, lsm=-5.0) for i, (P, S) in enumerate([(700, 96), (5000, 128), (20000, 256)])]
# round 6 (the key window): one scene under eight poses of the reference's sampler (depth ranges on either side of and across the
# binade boundary at 2.0), a scene whose depths span four binades (window too wide for some neighbours: coarse buckets), and one
# with 14 000 exact duplicates (an oversized bucket whatever the window)
import math
prng = np.random.RandomState(11)
scenes += [scene_inputs(P=20000, size=256, seed=95, lsm=-5.0, h=float(math.pi / 2 + prng.uniform(-1, 1)), v=float(math.pi / 2 + prng.uniform(-0.3, 0.3)),
                        fov_deg=float(prng.uniform(5, 17))) for _ in range(8)]
deep = scene_inputs(P=20000, size=256, seed=96, lsm=-5.0)
deep["means3D"] = (deep["means3D"] * torch.exp(3.0 * torch.rand(20000, 1, generator=torch.Generator().manual_seed(97)))).contiguous()
dup = scene_inputs(P=40000, size=256, seed=98, lsm=-5.5)
for k_ in ("means3D", "opacities", "shs", "scales", "rotations"):
    t_ = dup[k_].clone(); t_[20000:34000] = t_[123]; dup[k_] = t_.contiguous()
dup["opacities"] = (dup["opacities"] * 0.02).contiguous()
scenes += [deep, dup]
same_shape = {}
for i, d in enumerate(scenes): same_shape.setdefault((d["P"], d["W"], d["H"]), []).append(i)
oracles = [run_oracle(d) for d in scenes]
ctx = _capi.context_for(torch.device("cuda:0"))
ctx.set_option(_capi.OPT_MSD_SORT, 1)
m0, r0 = ctx.get_option(_capi.STAT_MSD_FRAMES), ctx.get_option(_capi.STAT_SORT_RERUNS)
k = 0
for f in range(frames):
    # runs of the same scene (streaks build), switches between scenes of one shape (the capacity hint's case) and of other shapes
    if f == 0 or rng.rand() < 0.15:
        k = int(rng.randint(len(scenes)))
    elif rng.rand() < 0.1:
        k = int(rng.choice(same_shape[(scenes[k]["P"], scenes[k]["W"], scenes[k]["H"])]))
    n, o = run_native(scenes[k], debug=False), oracles[k]
    assert n["num_rendered"] == o["num_rendered"], (f, k)
    assert np.array_equal(n["point_list"], o["point_list"]) and np.array_equal(n["ranges"], o["ranges"]), (f, k)
print("frames", frames, "two-launch frames", ctx.get_option(_capi.STAT_MSD_FRAMES) - m0, "re-rendered", ctx.get_option(_capi.STAT_SORT_RERUNS) - r0,
      "capacity retries", ctx.get_option(_capi.STAT_CAPACITY_RETRIES) if hasattr(_capi, "STAT_CAPACITY_RETRIES") else "n/a", "all equal")
