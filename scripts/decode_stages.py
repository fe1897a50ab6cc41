"""Stage times of the raster inside decode + render (BASELINE config 4): what the decoded scene costs per stage."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd import _capi
from gaussian_gan_decoder_amd.decoder import SequentialDecoderReverse
from gaussian_gan_decoder_amd.fused_decoder import FusedDecoder
from gaussian_gan_decoder_amd.gaussian_model import GaussianModel
from gaussian_gan_decoder_amd.gaussian_renderer import render_simple
from gaussian_gan_decoder_amd.synthetic import make_camera
dev = torch.device("cuda:0")
torch.manual_seed(0)
dec = SequentialDecoderReverse().to(dev)
fused = FusedDecoder(dec)
g = torch.Generator().manual_seed(5)
planes = torch.randn(3, 32, 256, 256, generator=g).to(dev)
d = torch.randn(1_000_000, 3, generator=g)
positions = (d / d.norm(dim=1, keepdim=True) * 0.3 * torch.clip(1 + 0.1 * torch.randn(1_000_000, 1, generator=g), 0, 1)).to(dev)
cam = make_camera(1024, 12.0, device=dev)
pc = GaussianModel(0)
bg = torch.zeros(3, device=dev)
def frame():
    with torch.no_grad():
        o = fused(planes, positions)
        pc._xyz, pc._scaling, pc._rotation, pc._opacity = o.xyz, o.scale, o.rotation, o.opacity
        pc._features_dc = o.color.unsqueeze(1)
        return render_simple(cam, pc, bg_color=bg, fused_activations=True)
for _ in range(5): out = frame()
ctx = _capi.context_for(dev)
ctx.set_profiling(True)
acc = {}
for _ in range(10):
    out = frame(); torch.cuda.synchronize()
    for k, v in ctx.stage_times().items():
        if not k.endswith("_bwd"): acc[k] = acc.get(k, 0.0) + v / 10
ctx.set_profiling(False)
print(json.dumps({k: round(v * 1e3, 1) for k, v in acc.items()}), "visible", int((out["radii"] > 0).sum()))
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(50): out = frame()
torch.cuda.synchronize()
print("decode + render ms/frame", (time.perf_counter() - t) / 50 * 1e3)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): out = frame()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=50))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=12, max_name_column_width=50))
