"""Seeded random differential test of the forward + backward raster against the oracle: image shapes that are not
multiples of the tile (and strips wider / taller than 64 tiles), SH degrees 0-3 with more stored coefficients than active
ones, precomputed colours / covariances, scale modifiers, every binning path and blend form, the exp modes INCLUDING the shipped
default (3: bare v_exp_f32 forward, compensated backward), the depth sort's front end folded or not, its two-launch form on or
off, and a third of the seeds through `FramePipeline` (ggd_forward_enqueue / _collect) -- the combinations no hand-written case
lists.  Pixels are excluded by cause (the oracle's fragile-pixel mask), never by outcome.  120 small scenes: ~15 s."""
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
                exp_mode=int(rng.choice([0, 2, 3, 3])), fold=int(rng.rand() < 0.6), msd=int(rng.rand() < 0.7),
                pipeline=bool(seed % 3 == 2))
    return d, opts


def _through_pipeline(d, frames, options):
    """`frames` renderings of `d` through a 2-slot FramePipeline (the first of a shape takes the ordinary path inside it), the
    slots' own contexts set to `options` {option: value} (and back to what they were afterwards: contexts are cached per
    stream handle, and handles are recycled); returns the decoded outputs of every frame."""
    from gaussian_gan_decoder_amd import rasterizer as R, _capi
    from _util import decode_buffers
    dev = torch.device("cuda:0")
    t = lambda x: torch.empty(0, device=dev) if x is None else x.to(dev)
    args = (t(d["bg"]), t(d["means3D"]), t(d["colors_precomp"]), t(d["opacities"]), t(d["scales"]), t(d["rotations"]),
            d["scale_modifier"], t(d["cov3D_precomp"]), t(d["viewmatrix"]), t(d["projmatrix"]), d["tanfovx"], d["tanfovy"],
            d["H"], d["W"], t(d["shs"]), d["sh_degree"], t(d["campos"]), False, False)
    pipe = R.FramePipeline(dev, slots=2)
    saved = []
    for s_ in pipe.slots:
        with torch.cuda.stream(s_["stream"]):
            c = _capi.context_and_stream(dev)[0]
        saved.append((c, {k: c.get_option(k) for k in options}))
        for k, v in options.items():
            c.set_option(k, v)
    try:
        res = [r for r in (pipe.submit(*args) for _ in range(frames)) if r is not None] + pipe.drain()
    finally:
        torch.cuda.synchronize(dev)
        for c, old in saved:
            for k, v in old.items():
                c.set_option(k, v)
    assert len(res) == frames
    outs = []
    for (num_rendered, color, radii, geom, binning, img, ev) in res:
        ev.synchronize()
        o = dict(num_rendered=num_rendered, color=color, radii=radii, geom=geom, binning=binning, img=img)
        o.update(decode_buffers(d["P"], d["W"], d["H"], num_rendered, geom, binning, img))
        outs.append(o)
    return outs


@pytest.mark.parametrize("seed", range(120))
def test_random_configuration_matches_oracle(native_lib, seed):
    from gaussian_gan_decoder_amd import _capi
    d, opts = _case(seed)
    W, H, P = d["W"], d["H"], d["P"]
    o = run_oracle(d)
    frag = fragile_pixels(o)
    cx = _capi.context_for(torch.device("cuda:0"))
    keys = (_capi.OPT_BLEND_SPLIT, _capi.OPT_BLEND_CULL, _capi.OPT_EXP_MODE, _capi.OPT_FOLD, _capi.OPT_MSD_SORT)
    saved = [cx.get_option(k) for k in keys]
    try:
        for k, v in zip(keys, (opts["split"], opts["cull"], opts["exp_mode"], opts["fold"], opts["msd"])):
            cx.set_option(k, v)
        if opts["pipeline"]:
            # (22 frames when the two-launch sort can arm: each of the two slots' contexts needs eight frames' key ranges first)
            frames = _through_pipeline(d, 22 if (opts["msd"] and opts["fold"] and opts["binning"] != 0) else 4, dict(zip(keys + (_capi.OPT_BINNING,), (opts["split"], opts["cull"], opts["exp_mode"],
                                                                                   opts["fold"], opts["msd"], opts["binning"]))))
        else:
            frames = None
        # the second call of a shape takes the single-call (capacity hint) form where it exists; with the folded front end and the
        # two-launch sort on, eleven frames: the window needs eight frames' key ranges (setting the options above restarted it),
        # the last two or three then sort in two launches
        many = frames is None and opts["msd"] and opts["fold"] and opts["binning"] != 0
        m_before = cx.get_option(_capi.STAT_MSD_FRAMES)
        for rep in range((11 if many else 2) if frames is None else len(frames)):
            n = run_native(d, debug=False, binning=opts["binning"]) if frames is None else frames[rep]
            assert n["num_rendered"] == o["num_rendered"], (opts, W, H, P)
            np.testing.assert_array_equal(n["radii"].cpu().numpy(), o["radii"])
            np.testing.assert_array_equal(n["point_list"], o["point_list"], err_msg=str(opts))
            np.testing.assert_array_equal(n["ranges"], o["ranges"], err_msg=str(opts))
            same = (n["n_contrib"] == o["n_contrib"]) | frag
            assert same.all(), (opts, int((~same).sum()))
            if (~frag).any():
                assert np.abs(n["color"].cpu().numpy() - o["color"])[:, ~frag].max() <= 1e-5, opts
        if many and o["num_rendered"] > 0:
            from _util import msd_bucket_sizes
            sizes = msd_bucket_sizes(o)
            if sizes is not None and sizes.max() <= 12288 and min(W, H) > 0:
                assert cx.get_option(_capi.STAT_MSD_FRAMES) >= m_before + 1, (opts, "the two-launch sort never ran")
        g = make_dL_dpix(max(W, H))[:, :H, :W].contiguous()
        g[:, torch.from_numpy(frag)] = 0.0
        ref, budget, fragile = backward_reference(d, o, n, g.numpy())
        nb = run_native_backward(d, n, g)
        assert check_gradients(d, nb, ref, budget, fragile) <= 1.0, opts
    finally:
        for k, v in zip(keys, saved):
            cx.set_option(k, v)
        cx.set_option(_capi.OPT_BINNING, 1)


def _sort_case(seed):
    """Scenes of several depth-sort tiles (4096 keys each) for the two-launch sort: all or part of the Gaussians on screen, the
    input in random / ascending / descending depth order (a sort tile then holds one depth range and most of its pieces are
    empty), depths spread over many buckets or squeezed into a few."""
    rng = np.random.RandomState(7000 + seed)
    P = int(rng.choice([9000, 12288, 20000, 33000, 70000])) if seed % 4 != 1 else int(rng.choice([33000, 70000]))
    W, H = [(256, 256), (200, 136), (320, 96)][rng.randint(3)]
    d = scene_inputs(P=P, size=max(W, H), kind=str(rng.choice(["cube", "shell"])), seed=300 + seed, lsm=float(rng.uniform(-6.0, -4.5)),
                     fov_deg=float(rng.uniform(8.0, 16.0)), width=W, height=H)
    zoom = float(rng.choice([0.25, 0.5, 1.0])) if seed % 4 != 1 else 0.25
    d["means3D"] = (zoom * d["means3D"]).contiguous()                                        # 0.25: everything on screen
    view = d["viewmatrix"]
    fwd = view[:3, 2]
    depth = d["means3D"] @ fwd + view[3, 2]
    squeeze = float(rng.choice([1.0, 1.0, 0.1, 0.02])) if seed % 4 != 1 else float(rng.choice([1.0, 0.1]))
    d["means3D"] = (d["means3D"] - ((depth - 2.7) * (1.0 - squeeze))[:, None] * fwd[None, :]).contiguous()
    order = str(rng.choice(["random", "ascending", "descending"]))
    if seed % 4 == 1:   # every fourth case: >= 8 fully kept sort tiles in ascending depth order (each tile one depth range)
        order = "ascending"
    if order != "random":
        perm = torch.argsort(d["means3D"] @ fwd + view[3, 2], descending=order == "descending")
        for k in ("means3D", "opacities", "shs", "scales", "rotations"):
            d[k] = d[k][perm].contiguous()
    return d, dict(P=P, W=W, H=H, squeeze=squeeze, order=order)


@pytest.mark.parametrize("seed", range(24))
def test_two_launch_sort_on_random_multi_tile_scenes(native_lib, seed):
    from gaussian_gan_decoder_amd import _capi
    from _util import assert_blend_matches
    d, what = _sort_case(seed)
    o = run_oracle(d)
    from _util import msd_bucket_sizes
    sizes = msd_bucket_sizes(o) if int((o["radii"] > 0).sum()) else None   # (over the window fitted to the scene's own key range)
    if sizes is None:
        pytest.skip("nothing visible, or a key range too wide for the two finishing passes: the two-launch sort does not apply")
    oversized = sizes.max() > 12288
    cx = _capi.context_for(torch.device("cuda:0"))
    saved = cx.get_option(_capi.OPT_MSD_SORT)
    cx.set_option(_capi.OPT_MSD_SORT, 1)          # (also restarts the speculation state: the window is this scene's alone)
    try:
        m0 = cx.get_option(_capi.STAT_MSD_FRAMES)
        for i in range(12):
            n = run_native(d, debug=False)
            assert n["num_rendered"] == o["num_rendered"], (what, i)
            np.testing.assert_array_equal(n["point_list"], o["point_list"], err_msg=f"{what} frame {i}")
            np.testing.assert_array_equal(n["ranges"], o["ranges"], err_msg=f"{what} frame {i}")
            if cx.get_option(_capi.STAT_MSD_FRAMES) >= m0 + 2:
                break
        assert oversized or cx.get_option(_capi.STAT_MSD_FRAMES) >= m0 + 2, what
        assert_blend_matches(n, o)
    finally:
        cx.set_option(_capi.OPT_MSD_SORT, saved)
