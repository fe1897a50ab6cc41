"""1920x1080 (120 x 68 tiles: beyond the 64 x 64 grid of the row binning) against the oracle: auto path selection."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from _util import scene_inputs, run_oracle, run_native
d = scene_inputs(P=120000, size=1080, lsm=-4.6, width=1920, height=1080, seed=3)
o = run_oracle(d)
print("R =", o["num_rendered"])
for path in (None, 0, 2, 3):
    n = run_native(d, debug=False, binning=path)
    assert n["num_rendered"] == o["num_rendered"]
    np.testing.assert_array_equal(n["point_list"], o["point_list"])
    np.testing.assert_array_equal(n["ranges"], o["ranges"])
    same = n["n_contrib"] == o["n_contrib"]
    err = np.abs(n["color"].cpu().numpy() - o["color"])[:, same].max()
    print("path", path, "ok; n_contrib flips", int((~same).sum()), "max|dRGB|", float(err))
