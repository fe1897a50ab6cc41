"""BASELINE config 1 -- "PanoHead triplane+decoder forward, 1 random z ... emit Gaussian xyz/scale/rot/SH" -- against the
reference itself: tests/golden/panohead_fixture.npz holds what the reference's own PanoHead TriPlaneGenerator
(PanoHead/training/triplane.py:18-293, random init, configuration of PanoHead/train.py:302-333, triplane_depth 3) +
SequentialDecoderReverse (main/decoder_models/sequential_decoder_reverse.py:38-86) produced on one seeded z
(tests/golden/make_panohead_golden.py).  The feature planes travel as the three windows the positions' taps touch."""
import os

import numpy as np
import pytest
import torch

from gaussian_gan_decoder_amd.decoder import SequentialDecoderReverse, sample_from_planes, triplane_mean, PLANE_AXES

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "panohead_fixture.npz")
HEADS = ("color", "opacity", "rotation", "scale", "xyz")


def _load():
    f = np.load(FIX)
    size, depth = int(f["size"]), int(f["depth"])
    win = torch.from_numpy(f["plane_windows"])
    planes = torch.zeros(3, win.shape[1], size, size)
    for p, (y0, x0) in enumerate(f["window_offsets"]):
        h, w = min(win.shape[2], size - y0), min(win.shape[3], size - x0)
        planes[p, :, y0:y0 + h, x0:x0 + w] = win[p, :, :h, :w]
    dec = SequentialDecoderReverse(plane_axes="panohead", triplane_depth=depth)
    dec.load_state_dict({k[3:]: torch.from_numpy(f[k]) for k in f.files if k.startswith("sd_")})   # the reference's keys
    return f, planes, torch.from_numpy(f["positions"]), dec.eval(), depth


def test_plane_axes_are_the_references():
    f = np.load(FIX)
    np.testing.assert_array_equal(f["plane_axes"], PLANE_AXES["panohead"].numpy())
    assert not np.array_equal(f["plane_axes"], PLANE_AXES["eg3d"].numpy())


def test_panohead_decoder_forward_matches_reference_on_cpu():
    f, planes, pos, dec, depth = _load()
    with torch.no_grad():
        out = dec(planes, pos)
    for h in HEADS:
        err = np.abs(getattr(out, h).numpy() - f[h]).max()
        assert err <= 2e-5, (h, err)
    # the generator's planes are not a blank: the samples must depend on the tri-grid depth axis and on the plane axes
    with torch.no_grad():
        a = sample_from_planes(planes, pos, 1.0, "panohead", depth).mean(0)
        b = sample_from_planes(planes, pos, 1.0, "eg3d", depth).mean(0)
    assert float(f["plane_abs_mean"]) > 0.1 and (a - b).abs().max() > 1e-2


@pytest.mark.gpu
def test_panohead_decoder_forward_matches_reference_on_gpu(native_lib):
    """Same fixture through the HIP tri-grid gather (ggd_trigrid_forward) + the fp32 module (<= 2e-5) and through the
    fused f16-MFMA decoder (<= 2e-3); the gather's backward against torch's grid_sample autograd."""
    from gaussian_gan_decoder_amd.fused_decoder import FusedDecoder
    f, planes, pos, dec, depth = _load()
    dev = torch.device("cuda:0")
    planes_d, pos_d, dec_d = planes.to(dev), pos.to(dev), dec.to(dev)
    with torch.no_grad():
        feats = triplane_mean(planes_d, pos_d, 1.0, "panohead", depth)
        ref_feats = sample_from_planes(planes, pos, 1.0, "panohead", depth).mean(0)
        assert (feats.cpu() - ref_feats).abs().max() <= 2e-6 * max(1.0, float(ref_feats.abs().max()))
        out = dec_d(planes_d, pos_d)
        fused = FusedDecoder(dec_d)(planes_d, pos_d)
    for h in HEADS:
        assert np.abs(getattr(out, h).cpu().numpy() - f[h]).max() <= 2e-5, h
        assert np.abs(getattr(fused, h).float().cpu().numpy() - f[h]).max() <= 2e-3 * max(1.0, float(np.abs(f[h]).max())), h
    # backward of the gather (scatter-add into the tri-grids) vs autograd through grid_sample, on the window region
    g = torch.Generator().manual_seed(3)
    dout = torch.randn(pos.shape[0], 32, generator=g)
    p_ref = planes.clone().requires_grad_(True)
    (sample_from_planes(p_ref, pos, 1.0, "panohead", depth).mean(0) * dout).sum().backward()
    p_hip = planes_d.clone().requires_grad_(True)
    (triplane_mean(p_hip, pos_d, 1.0, "panohead", depth) * dout.to(dev)).sum().backward()
    err = (p_hip.grad.cpu() - p_ref.grad).abs().max()
    assert err <= 1e-4 * max(1.0, float(p_ref.grad.abs().max())), float(err)
