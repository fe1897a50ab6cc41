import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd.train import DecoderTrainer, make_scene_batch
dev = torch.device('cuda:0')
fused = '--fused' in sys.argv
# --standins: bench.py's configuration (backbone gradient payload + the perceptual slot); --eg3d: tri-planes instead of the
# PanoHead tri-grids BASELINE config 3 names
planes = dict() if '--eg3d' in sys.argv else dict(plane_axes="panohead", triplane_depth=3)
standins = dict(backbone_params=29_570_000 - 3 * 32 * (1 if '--eg3d' in sys.argv else 3) * 256 * 256, perceptual_weight=1.0) \
    if '--standins' in sys.argv else dict()
tr = DecoderTrainer(dev, n_scenes_total=4, image_size=512, fused_activations=True, fused_decoder=fused,
                    decoder_precision='fp32' if '--fp32' in sys.argv else 'bf16', **planes, **standins)
b = make_scene_batch([0,1,2,3], 500000, 512, dev, seed=0)
for _ in range(4): tr.step(b)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(b); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=60))
if "--cpu" in sys.argv: print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=60))
t=time.perf_counter()
for _ in range(10): tr.step(b)
torch.cuda.synchronize(); print('ms/iter', (time.perf_counter()-t)/10*1e3)
