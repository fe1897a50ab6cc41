"""Diagnostic: why does the fused-decoder trainer learn slower than the fp32 one at the config-3 size?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gaussian_gan_decoder_amd.train import DecoderTrainer, make_scene_batch
from gaussian_gan_decoder_amd.fused_decoder import pack_weights

dev = torch.device("cuda:0")
B, N, S = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 500_000, 512
batch = make_scene_batch(list(range(B)), N, S, dev, seed=0)
res = {}
for name, kw in (("fp32", dict()), ("fused", dict(fused_decoder=True, fused_activations=True)),
                 ("fused-noact", dict(fused_decoder=True))):
    tr = DecoderTrainer(dev, n_scenes_total=B, image_size=S, seed=11, lr=5e-3, perceptual_weight=0.05,
                        perceptual_width_div=4, backbone_params=100_000, **kw)
    p0 = [p.detach().clone() for p in tr.params]
    v0 = [p._version for p in tr.params]
    tr.flat_grad.zero_()
    loss = tr.local_loss(batch); loss.backward()
    torch.cuda.synchronize()
    g = tr.flat_grad.detach().clone()
    res[name] = g
    off = 0
    norms = []
    for p in tr.params:
        n = p.numel(); norms.append(float(g[off:off + n].norm())); off += n
    print(name, "loss", float(loss), "grad norms (first 8 tensors, planes, backbone):", [f"{x:.3e}" for x in norms[:8]], f"{norms[-2]:.3e}", f"{norms[-1]:.3e}")
    if "fused" in name:
        img0 = pack_weights(tr.decoder).clone()
    tr.allreduce_and_step()
    torch.cuda.synchronize()
    v1 = [p._version for p in tr.params]
    print("  versions bumped:", sum(int(a != b) for a, b in zip(v0, v1)), "of", len(v0),
          " max |dparam|:", max(float((a - b).abs().max()) for a, b in zip(p0, tr.params)))
    if "fused" in name:
        img1 = pack_weights(tr.decoder)
        print("  packed image changed by the step:", bool((img0 != img1).any()))   # (the images are rebuilt on every forward)
    ls = [tr.step(batch) for _ in range(6)]
    print("  next losses:", [f"{x:.5f}" for x in ls])
    del tr; torch.cuda.empty_cache()
a, b = res["fp32"], res["fused"]
print("rel L2 diff of the full flat gradient fused vs fp32:", float((a - b).norm() / a.norm()), " cos:", float(torch.dot(a, b) / (a.norm() * b.norm())))
a, b = res["fp32"], res["fused-noact"]
print("rel L2 diff fused-noact vs fp32:", float((a - b).norm() / a.norm()), " cos:", float(torch.dot(a, b) / (a.norm() * b.norm())))
