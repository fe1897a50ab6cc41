import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from _util import adversarial_inputs, run_oracle, run_native, run_native_backward
from gaussian_gan_decoder_amd.synthetic import make_dL_dpix
from oracle import ggd_oracle as O
d = adversarial_inputs()
g = make_dL_dpix(max(d["W"], d["H"]))[:, :d["H"], :d["W"]].contiguous()
o32 = run_oracle(d); o64 = run_oracle(d, dtype=np.float64)
b32 = O.backward(o32, g.numpy()); b64 = O.backward(o64, g.numpy().astype(np.float64))
n = run_native(d, debug=False); nb = run_native_backward(d, n, g)
for name in ("dL_dmeans3D", "dL_dscales", "dL_drots", "dL_dcov3D"):
    ref = b64[name]; a = b32[name].astype(np.float64); c = nb[name].reshape(ref.shape).astype(np.float64)
    e_or = np.abs(a - ref).max(1); e_gpu = np.abs(c - ref).max(1); e_go = np.abs(c - a).max(1)
    idx = np.argsort(-e_go)[:8]
    print(name, "worst members", idx.tolist())
    for i in idx[:6]:
        print(f"   i={i:3d} |ref|={np.abs(ref[i]).max():.3e} oracle32-64={e_or[i]:.3e} gpu-64={e_gpu[i]:.3e} gpu-oracle32={e_go[i]:.3e}")
