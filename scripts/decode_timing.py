import sys, os, time, math, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd.decoder import SequentialDecoderReverse, triplane_mean
from gaussian_gan_decoder_amd.fused_decoder import FusedDecoder
dev = torch.device('cuda:0')
torch.manual_seed(0)
dec = SequentialDecoderReverse().to(dev)
planes = torch.randn(3, 32, 256, 256, device=dev)
fused = FusedDecoder(dec)
for N in (100000, 500000, 1000000):
    pos = torch.rand(N, 3, device=dev) - 0.5
    feats = triplane_mean(planes, pos, 1.0)
    for _ in range(3): a = fused.decode_features(feats, pos)
    torch.cuda.synchronize(); t = time.perf_counter()
    K = 20
    for _ in range(K): a = fused.decode_features(feats, pos)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
    for _ in range(3): f2 = triplane_mean(planes, pos, 1.0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(K): f2 = triplane_mean(planes, pos, 1.0)
    torch.cuda.synchronize(); dtg = (time.perf_counter() - t) / K
    with torch.no_grad():
        for _ in range(2): r = dec(planes, pos)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): r = dec(planes, pos)
        torch.cuda.synchronize(); dtt = (time.perf_counter() - t) / 5
    flops = 2 * 192512 * N
    print(json.dumps(dict(N=N, fused_mlp_ms=round(dt*1e3, 3), TFLOPs=round(flops/dt/1e12, 1), gather_ms=round(dtg*1e3, 3), torch_decoder_ms=round(dtt*1e3, 2))))
