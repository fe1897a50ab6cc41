"""Diagnostic: the 1920x1080 / 2048^2 large-splat cases against the oracle for every blend option (cull on/off, backward forms)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from _util import *
from gaussian_gan_decoder_amd import _capi
from oracle import ggd_oracle as O

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
d = scene_inputs(P=200_000, size=max(W, H), kind="cube", seed=4, lsm=-5.6, width=W, height=H)
o = run_oracle(d)
g = make_dL_dpix(max(W, H))[:, :H, :W].contiguous()
cx = _capi.context_for(torch.device("cuda:0"))
for cull in (1, 0):
    for split in (1, 3, 4):      # backward blend: auto, tile form, quarter form
        cx.set_option(_capi.OPT_BLEND_CULL, cull); cx.set_option(_capi.OPT_BLEND_SPLIT, split)
        n = run_native(d, debug=False)
        color = n["color"].cpu().numpy()
        same = n["n_contrib"] == o["n_contrib"]
        err = np.abs(color - o["color"]).max(0)
        errs = np.where(same, err, 0)
        iy, ix = np.unravel_index(errs.argmax(), errs.shape)
        print(f"cull={cull} split={split}: n_contrib flips {int((~same).sum())}, max|dRGB| {errs.max():.3e} at ({ix},{iy}) n_contrib {n['n_contrib'][iy, ix]}, "
              f"pixels > 1e-5: {int((errs > 1e-5).sum())}, > 1e-6: {int((errs > 1e-6).sum())}, mean err {errs.mean():.2e}")
        ref, budget, fragile = backward_reference(d, o, n, g.numpy())
        nb = run_native_backward(d, n, g)
        rep = []
        w = check_gradients(d, nb, ref, budget, fragile, report=rep)
        print("   bwd worst ratio", f"{w:.3f}", {r["array"]: round(r["worst_ratio"], 3) for r in rep})
        if cull == 1 and split == 1:
            # where does dL_dcolors go wrong?
            r = ref["dL_dcolors"]; gg = nb["dL_dcolors"].reshape(r.shape)
            tol = 1e-5 + 0.25 * 2.0 ** -24 * budget["dL_dcolors"]
            ratio = np.abs(gg - r) / tol
            bad = np.argsort(ratio.max(1))[::-1][:8]
            for i in bad:
                print("   gaussian", i, "ratio", ratio[i].round(2), "ref", r[i], "gpu", gg[i], "xy", o["xy"][i], "radius", o["radii"][i],
                      "conic/op", o["conic_opacity"][i], "tiles", o["tiles_touched"][i])
