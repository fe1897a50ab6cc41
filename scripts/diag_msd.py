import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from _util import scene_inputs, run_native, run_oracle
from gaussian_gan_decoder_amd import _capi
ctx = _capi.context_for(torch.device("cuda:0"))
big = scene_inputs(P=70000, size=256, lsm=-5.0, seed=41)
tiny = scene_inputs(P=700, size=256, lsm=-3.5, seed=42)
slab = scene_inputs(P=60000, size=256, lsm=-5.5, seed=43)
view = slab["viewmatrix"]
fwd, cam_pos = view[:3, 2], torch.inverse(view)[3, :3]
rel = slab["means3D"] - cam_pos
slab["means3D"] = (slab["means3D"] - (rel @ fwd - 2.7)[:, None] * fwd[None, :] * (1.0 - 1e-5)).contiguous()
sc = dict(big=big, tiny=tiny, slab=slab)
o = {k: run_oracle(d) for k, d in sc.items()}
order = sys.argv[1].split(",")
for i, k in enumerate(order):
    print("->", i, k, flush=True)
    n = run_native(sc[k], debug=False)
    torch.cuda.synchronize()
    print(i, k, "streak", ctx.get_option(_capi.STAT_FLAT_STREAK), "msd", ctx.get_option(_capi.STAT_MSD_FRAMES), "reruns", ctx.get_option(_capi.STAT_SORT_RERUNS),
          "R", n["num_rendered"], o[k]["num_rendered"], "list ok", bool((n["point_list"] == o[k]["point_list"]).all()), "ranges ok", bool((n["ranges"] == o[k]["ranges"]).all()), flush=True)
