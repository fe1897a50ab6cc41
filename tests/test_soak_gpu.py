"""Soak of the single-call forward's cross-frame state (capacity hint, the two alternating control blocks, the flat-frame
streak that drops the fourth sort pass, the two-launch depth sort and its pause after a miss): 240 frames over three scenes
of ONE shape -- different sizes (the capacity hint swings), different depth ranges (constant and varying top key byte) -- in an
order that builds streaks, breaks them and alternates every frame.  Every frame's num_rendered, sorted list, tile ranges and
image must equal the two-call form's (buffers laid out for the exact num_rendered, no speculation) bit for bit.
(Promoted from scripts/soak_forward.py, which checks image checksums over thousands of frames.)"""
import numpy as np
import pytest
import torch

from _util import scene_inputs, run_native, run_oracle

pytestmark = pytest.mark.gpu


def test_240_frames_over_three_scenes_of_one_shape_equal_the_two_call_form(native_lib):
    from gaussian_gan_decoder_amd import _capi
    ctx = _capi.context_for(torch.device("cuda:0"))
    P, S = 30000, 256
    near = scene_inputs(P=P, size=S, lsm=-4.8, seed=61)
    dense = scene_inputs(P=P, size=S, lsm=-3.6, seed=62)                   # same shape, several times the instances
    deep = scene_inputs(P=P, size=S, lsm=-4.8, seed=63)
    g = torch.Generator().manual_seed(64)
    deep["means3D"] = (deep["means3D"] * torch.exp(3.0 * torch.rand(P, 1, generator=g))).contiguous()   # depths over ~4 binades
    scenes = dict(near=near, dense=dense, deep=deep)
    ref = {}
    for k, d in scenes.items():
        n = run_native(d, debug=True)                                      # the exact two-call form
        ref[k] = (n["num_rendered"], n["point_list"].copy(), n["ranges"].copy(), n["color"].clone())
    o = run_oracle(near)                                                   # (and the reference itself against the oracle, once)
    assert ref["near"][0] == o["num_rendered"]
    np.testing.assert_array_equal(ref["near"][1], o["point_list"])
    np.testing.assert_array_equal(ref["near"][2], o["ranges"])
    assert ref["dense"][0] > 3 * ref["near"][0]
    order = (["near"] * 14 + ["dense", "near"] * 10 + ["deep"] + ["dense"] * 12 + ["deep", "deep", "near"] +
             ["near", "dense", "deep"] * 8 + ["near"] * 70 + ["dense"] * 12 + ["deep"] + ["near", "dense"] * 20 + ["near"] * 26 +
             ["deep", "near", "near", "dense", "deep", "near", "dense"])
    order = order[:240] + ["near"] * max(0, 240 - len(order))
    assert len(order) == 240
    m0, r0, c0 = ctx.get_option(_capi.STAT_MSD_FRAMES), ctx.get_option(_capi.STAT_SORT_RERUNS), ctx.capacity_retries
    for i, k in enumerate(order):
        n = run_native(scenes[k], debug=False)
        R, lst, rng, img = ref[k]
        assert n["num_rendered"] == R, (i, k)
        np.testing.assert_array_equal(n["point_list"], lst, err_msg=f"frame {i} ({k})")
        np.testing.assert_array_equal(n["ranges"], rng, err_msg=f"frame {i} ({k})")
        assert torch.equal(n["color"], img), (i, k)
    msd, reruns = ctx.get_option(_capi.STAT_MSD_FRAMES) - m0, ctx.get_option(_capi.STAT_SORT_RERUNS) - r0
    print(f"\n  240 frames: {msd} sorted in two launches, {reruns} rendered again after a wrong guess, "
          f"{ctx.capacity_retries - c0} capacity retries")
    assert msd >= 40 and 1 <= reruns <= 12          # the states were actually visited
