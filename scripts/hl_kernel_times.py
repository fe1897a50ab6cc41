"""Kernel times of the decoder's training kernels (forward with z, backward, weight gradients) at 2 M points -- the reference-
precision form by default, `bf16` for the 16-bit tier -- from the torch profiler; for A/B runs with GGD_LIB_PATH."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from gaussian_gan_decoder_amd.decoder import SequentialDecoderReverse
from gaussian_gan_decoder_amd.fused_decoder import FusedDecoderFn, device_pack, _head_tensors
dev = torch.device("cuda:0")
n = 2_000_000
hl = not (len(sys.argv) > 1 and sys.argv[1] == "bf16")
torch.manual_seed(0)
dec = SequentialDecoderReverse().to(dev)
params = [t for h in _head_tensors(dec) for t in h]
feats = torch.randn(n, 32, device=dev).requires_grad_(True)
pos = torch.rand(n, 3, device=dev) - 0.5
w = torch.randn(n, 16, device=dev)
def step():
    packed, packed_t = device_pack(dec, params, None, hl)
    a = FusedDecoderFn.apply(feats, pos, packed, packed_t, hl, *params)
    (a * w).sum().backward()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(6): step()
    torch.cuda.synchronize()
res = {}
for e in prof.key_averages():
    for k in ("decoder_forward", "decoder_backward", "decoder_wgrad"):
        if k in e.key:
            res[k] = round(e.device_time_total / e.count / 1e3, 3)
print("decoder kernels ms @2M points:", json.dumps(res), "sum", round(sum(res.values()), 3), "lib", os.environ.get("GGD_LIB_PATH", "in-tree"))
