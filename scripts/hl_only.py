"""Scratch driver for rocprofv3: the reference-precision decoder kernels (forward with z, backward, weight gradients) on
2 M points (the train step's size), N iterations; `bf16` as the second argument runs the bf16 kernels instead."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd.decoder import SequentialDecoderReverse
from gaussian_gan_decoder_amd.fused_decoder import FusedDecoderFn, device_pack, _head_tensors
dev = torch.device("cuda:0")
n = 2_000_000
hl = not (len(sys.argv) > 2 and sys.argv[2] == "bf16")
torch.manual_seed(0)
dec = SequentialDecoderReverse().to(dev)
params = [t for h in _head_tensors(dec) for t in h]
feats = torch.randn(n, 32, device=dev).requires_grad_(True)
pos = torch.rand(n, 3, device=dev) - 0.5
w = torch.randn(n, 16, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    packed, packed_t = device_pack(dec, params, None, hl)
    a = FusedDecoderFn.apply(feats, pos, packed, packed_t, hl, *params)
    (a * w).sum().backward()
# inference form (no z stores), same size
from gaussian_gan_decoder_amd.fused_decoder import FusedDecoder
inf = FusedDecoder(dec, precision="fp32" if hl else "bf16")
for _ in range(3):
    inf.decode_features(feats.detach(), pos)
torch.cuda.synchronize()
print("done")
