"""Decoder forward (fused fast path): kernel time at 1 M points and error against the fp32 module, for the library
GGD_LIB_PATH selects (A/B of forward variants inside one gpurun call)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd.decoder import SequentialDecoderReverse
from gaussian_gan_decoder_amd.fused_decoder import FusedDecoder, FusedTrainDecoder
dev = torch.device("cuda:0")
torch.manual_seed(1)
dec = SequentialDecoderReverse().to(dev)
for p in dec.parameters():
    if p.dim() == 2:
        p.data *= 1.5
planes = torch.randn(3, 32, 64, 64, device=dev)
N = 1_000_000
pos = torch.rand(N, 3, device=dev) - 0.5
fused = FusedDecoder(dec)
with torch.no_grad():
    out = fused(planes, pos)
    ref = dec(planes, pos)
err = {}
for k in ("color", "opacity", "rotation", "scale", "xyz"):
    d = (getattr(out, k) - getattr(ref, k)).abs()
    err[k] = {"max": float(d.max()), "mean": float(d.mean()), "ref_absmax": float(getattr(ref, k).abs().max())}
print(json.dumps(err))
from torch.profiler import profile, ProfilerActivity
res = {}
for name, mod, grad in (("inference", fused, False), ("train_forward", FusedTrainDecoder(dec), True)):
    pl = planes.clone().requires_grad_(grad)
    for _ in range(3):
        o = mod(pl, pos)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            o = mod(pl, pos)
        torch.cuda.synchronize()
    for e in prof.key_averages():
        if "decoder_forward" in e.key:
            res[name] = round(e.device_time_total / e.count, 1)
print("decoder_forward kernel us:", json.dumps(res), "lib", os.environ.get("GGD_LIB_PATH", "in-tree"))
