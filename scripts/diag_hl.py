"""Diagnostic: accuracy of the split-operand decoder (precision='fp32') against float64 autograd, per tensor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussian_gan_decoder_amd.decoder import SequentialDecoderReverse, triplane_mean
from gaussian_gan_decoder_amd.fused_decoder import FusedTrainDecoder, FusedDecoderFn, device_pack, _head_tensors

dev = torch.device("cuda:0")
torch.manual_seed(3)
ref = SequentialDecoderReverse().to(dev)
for p in ref.parameters():
    if p.dim() == 2:
        p.data *= 1.5
mod = SequentialDecoderReverse().to(dev); mod.load_state_dict(ref.state_dict())
ref = ref.double()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_003
planes = torch.randn(3, 32, 64, 64, device=dev)
pos = torch.rand(N, 3, device=dev) - 0.5
g = torch.Generator().manual_seed(4)
w = {k: torch.randn(N, d, generator=g).to(dev) for k, d in (("color", 3), ("opacity", 1), ("rotation", 4), ("scale", 3), ("xyz", 3))}
loss = lambda o: sum((getattr(o, k) * w[k].to(getattr(o, k).dtype)).sum() for k in w) / N
rel = lambda a, b: ((a - b.double()).norm() / (a.norm() + 1e-30)).item()
# reference: gradient w.r.t. the per-point features (not the planes) in float64
feats64 = triplane_mean(planes, pos, 1.0).double().requires_grad_(True)
from types import SimpleNamespace
def ref_from_feats(f64):
    info = pos.double()
    d = ref
    def head(h, x): return h(x[:, :32], x[:, 32:]) if False else None
    outs = {}
    import torch.nn.functional as F
    def mlp(hd, x):
        bb = hd.backbone
        for k in (0, 2, 4): x = F.gelu(bb[k](x))
        return bb[6](x)
    color = mlp(d.color_decoder, torch.cat([f64, info], 1)); info = torch.cat([info, color], 1)
    opac = mlp(d.opacity_decoder, torch.cat([f64, info], 1)); info = torch.cat([info, opac], 1)
    rot = mlp(d.rotation_decoder, torch.cat([f64, info], 1)); info = torch.cat([info, rot], 1)
    scale = d.activate_scale(mlp(d.scale_decoder, torch.cat([f64, info], 1))); info = torch.cat([info, scale], 1)
    xyz = mlp(d.xyz_decoder, torch.cat([f64, info], 1)) * 0.01 + pos.double()
    return SimpleNamespace(color=color, opacity=opac, rotation=rot, scale=scale, xyz=xyz)
oa = ref_from_feats(feats64); loss(oa).backward()
for prec in ("bf16", "fp32"):
    for p in mod.parameters(): p.grad = None
    f32 = feats64.detach().float().requires_grad_(True)
    params = [t for h in _head_tensors(mod) for t in h]
    packed, packed_t = device_pack(mod, params, None, prec == "fp32")
    a = FusedDecoderFn.apply(f32, pos, packed, packed_t, prec == "fp32", *params)
    ob = SimpleNamespace(color=a[:, 0:3], opacity=a[:, 3:4], rotation=a[:, 4:8], scale=a[:, 8:11], xyz=a[:, 11:14])
    loss(ob).backward()
    print(prec, "outputs:", {k: f"{(getattr(oa, k) - getattr(ob, k).double()).abs().max().item():.2e}" for k in w})
    print("   dfeat rel L2:", f"{rel(feats64.grad, f32.grad):.2e}", " per-point rel (median):",
          f"{((feats64.grad - f32.grad.double()).norm(dim=1) / feats64.grad.norm(dim=1)).median().item():.2e}")
    rows = []
    for (na, pa), (nb, pb) in zip(ref.named_parameters(), mod.named_parameters()):
        rows.append((na, rel(pa.grad, pb.grad)))
    print("   params: " + "  ".join(f"{n.replace('_decoder.backbone', '')}={r:.1e}" for n, r in rows))
