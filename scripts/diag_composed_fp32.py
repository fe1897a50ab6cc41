"""Per-tensor gradient error of the composed train step with the reference-precision fused decoder against the oracle-backed
CPU trainer (the quantities tests/test_train_step_gpu.py::test_composed_step_with_the_fused_decoder_at_fp32_precision bounds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_train_step_gpu import _make, _half_step, PLANES, N_POINTS, SMALL
from gaussian_gan_decoder_amd.train import make_scene_batch
dev = torch.device("cuda:0")
for planes in ("eg3d", "panohead"):
    cpu_tr = _make("cpu", **PLANES[planes])
    gpu_tr = _make(dev, fused_decoder=True, fused_activations=True, decoder_precision="fp32", **PLANES[planes])
    cb = make_scene_batch([0, 1], N_POINTS, SMALL["image_size"], "cpu", seed=0)
    gb = make_scene_batch([0, 1], N_POINTS, SMALL["image_size"], dev, seed=0)
    lc, lg = _half_step(cpu_tr, cb), _half_step(gpu_tr, gb)
    torch.cuda.synchronize()
    gc, gg = cpu_tr.flat_grad.clone(), gpu_tr.flat_grad.detach().cpu()
    off = 0
    names = [n for n, _ in cpu_tr.decoder.named_parameters()]
    rows = []
    for i, p in enumerate(cpu_tr.params):
        n = p.numel(); a, b = gc[off:off + n], gg[off:off + n]; off += n
        rows.append((float((a - b).abs().max()) / (1e-30 + float(a.abs().max())), float((a - b).norm() / (a.norm() + 1e-30)), tuple(p.shape)))
    rows.sort(reverse=True)
    print(planes, "loss", lc, lg)
    for r in rows[:8]:
        print("   max|err|/max|g| = %.2e   relL2 = %.2e   %s" % r)

# dynamic range of the per-point incoming gradient (what the loss scale of the fp16 dz plane has to cover)
from gaussian_gan_decoder_amd import fused_decoder as FD
orig = FD.FusedDecoderFn.backward
def spy(ctx, dattrs):
    m = dattrs.abs().amax(dim=1)
    g = float(m.max())
    q = torch.quantile((m / g).clamp_min(1e-30).log2().float().cpu(), torch.tensor([0.01, 0.1, 0.5, 0.9, 0.99]))
    cols = dattrs.abs().amax(dim=0)
    print("   dattrs: global max %.3e; log2(per-point max / global max) quantiles 1/10/50/90/99 %%: %s" % (g, [round(float(x), 1) for x in q]))
    print("   per-column max:", [f"{float(c):.1e}" for c in cols[:14]])
    return orig(ctx, dattrs)
FD.FusedDecoderFn.backward = staticmethod(spy)
gpu_tr = _make(dev, fused_decoder=True, fused_activations=True, decoder_precision="fp32")
_half_step(gpu_tr, make_scene_batch([0, 1], N_POINTS, SMALL["image_size"], dev, seed=0))
from gaussian_gan_decoder_amd.train import DecoderTrainer
tr = DecoderTrainer(dev, n_scenes_total=4, image_size=512, fused_activations=True, fused_decoder=True, decoder_precision="fp32", perceptual_weight=1.0, backbone_params=1000)
tr.step(make_scene_batch([0, 1, 2, 3], 500000, 512, dev, seed=0))
