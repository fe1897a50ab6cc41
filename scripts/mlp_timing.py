import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd.decoder import SequentialDecoderReverse, triplane_mean
from gaussian_gan_decoder_amd.fused_decoder import FusedDecoder
dev = torch.device("cuda:0"); n = 1_000_000
torch.manual_seed(0)
dec = SequentialDecoderReverse().to(dev); fused = FusedDecoder(dec)
g = torch.Generator().manual_seed(5)
planes = torch.randn(3, 32, 256, 256, generator=g).to(dev)
d = torch.randn(n, 3, generator=g); pos = (d / d.norm(dim=1, keepdim=True) * 0.3).to(dev)
feats = triplane_mean(planes, pos, 1.0)
for _ in range(5): fused.decode_features(feats, pos)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): fused.decode_features(feats, pos)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
print(os.environ.get("GGD_MLP_SKEW", "0"), "mlp ms", round(dt * 1e3, 4), "TF", round(2 * 192512 * n / dt / 1e12, 1))
