"""Seeded random differential test of the forward + backward raster against the oracle: image shapes that are not
multiples of the tile (and strips wider / taller than 64 tiles), SH degrees 0-3 with more stored coefficients than active
ones, precomputed colours / covariances, scale modifiers, every binning path and blend form, two exp modes -- the
combinations no hand-written case lists.  120 small scenes: the whole file runs in ~10 s."""
import numpy as np
import pytest
import torch

from _util import (scene_inputs, run_oracle, run_native, run_native_backward, backward_reference, check_gradients,
                   fragile_pixels)
from gaussian_gan_decoder_amd.synthetic import make_dL_dpix

pytestmark = pytest.mark.gpu

SHAPES = [(64, 64), (100, 52), (17, 33), (256, 144), (1, 1), (16, 16), (1100, 48), (40, 1090), (333, 333), (1040, 80)]


def _case(seed):
    rng = np.random.RandomState(1000 + seed)
    W, H = SHAPES[rng.randint(len(SHAPES))]
    P = int(rng.choice([1, 7, 64, 300, 2000, 6000]))
    deg = int(rng.randint(0, 4))
    M = int(rng.choice([(deg + 1) ** 2, 16])) if deg > 0 else 1
    use_colors = bool(rng.rand() < 0.2)
    use_cov = bool(rng.rand() < 0.2)
    d = scene_inputs(P=P, size=max(W, H), kind=str(rng.choice(["cube", "shell"])), seed=seed, sh_degree=0 if use_colors else deg,
                     sh_M=None if use_colors else M, use_colors=use_colors, use_cov=use_cov, lsm=float(rng.uniform(-6.5, -3.0)),
                     fov_deg=float(rng.uniform(6.0, 20.0)), width=W, height=H, scale_modifier=float(rng.choice([1.0, 0.7, 1.6])))
    opts = dict(binning=int(rng.choice([0, 1, 2, 3])), split=int(rng.choice([1, 3, 4])), cull=int(rng.rand() < 0.8),
                exp_mode=int(rng.choice([0, 2])))
    return d, opts


@pytest.mark.parametrize("seed", range(120))
def test_random_configuration_matches_oracle(native_lib, seed):
    from gaussian_gan_decoder_amd import _capi
    d, opts = _case(seed)
    W, H, P = d["W"], d["H"], d["P"]
    o = run_oracle(d)
    frag = fragile_pixels(o)
    cx = _capi.context_for(torch.device("cuda:0"))
    saved = [cx.get_option(k) for k in (_capi.OPT_BLEND_SPLIT, _capi.OPT_BLEND_CULL, _capi.OPT_EXP_MODE)]
    try:
        cx.set_option(_capi.OPT_BLEND_SPLIT, opts["split"]); cx.set_option(_capi.OPT_BLEND_CULL, opts["cull"])
        cx.set_option(_capi.OPT_EXP_MODE, opts["exp_mode"])
        for rep in range(2):      # the second call of a shape takes the single-call (capacity hint) form where it exists
            n = run_native(d, debug=False, binning=opts["binning"])
            assert n["num_rendered"] == o["num_rendered"], (opts, W, H, P)
            np.testing.assert_array_equal(n["radii"].cpu().numpy(), o["radii"])
            np.testing.assert_array_equal(n["point_list"], o["point_list"], err_msg=str(opts))
            np.testing.assert_array_equal(n["ranges"], o["ranges"], err_msg=str(opts))
            same = (n["n_contrib"] == o["n_contrib"]) | frag
            assert same.all(), (opts, int((~same).sum()))
            if (~frag).any():
                assert np.abs(n["color"].cpu().numpy() - o["color"])[:, ~frag].max() <= 1e-5, opts
        g = make_dL_dpix(max(W, H))[:, :H, :W].contiguous()
        g[:, torch.from_numpy(frag)] = 0.0
        ref, budget, fragile = backward_reference(d, o, n, g.numpy())
        nb = run_native_backward(d, n, g)
        assert check_gradients(d, nb, ref, budget, fragile) <= 1.0, opts
    finally:
        for k, v in zip((_capi.OPT_BLEND_SPLIT, _capi.OPT_BLEND_CULL, _capi.OPT_EXP_MODE), saved):
            cx.set_option(k, v)
        cx.set_option(_capi.OPT_BINNING, 1)
