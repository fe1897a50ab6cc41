"""Scratch driver for rocprofv3: the fused decoder MLP (inference kernel, 1 M points) x N, optionally the train kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd.decoder import SequentialDecoderReverse, triplane_mean
from gaussian_gan_decoder_amd.fused_decoder import FusedDecoder
dev = torch.device("cuda:0")
n = 1_000_000
torch.manual_seed(0)
dec = SequentialDecoderReverse().to(dev)
fused = FusedDecoder(dec)
g = torch.Generator().manual_seed(5)
planes = torch.randn(3, 32, 256, 256, generator=g).to(dev)
d = torch.randn(n, 3, generator=g)
pos = (d / d.norm(dim=1, keepdim=True) * 0.3).to(dev)
feats = triplane_mean(planes, pos, 1.0)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    fused.decode_features(feats, pos)
torch.cuda.synchronize()
print("done")
