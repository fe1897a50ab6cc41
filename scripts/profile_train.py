import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd.train import DecoderTrainer, make_scene_batch
dev = torch.device('cuda:0')
fused = '--fused' in sys.argv
tr = DecoderTrainer(dev, n_scenes_total=4, image_size=512, fused_activations=True, fused_decoder=fused,
                    decoder_precision='fp32' if '--fp32' in sys.argv else 'bf16')
b = make_scene_batch([0,1,2,3], 500000, 512, dev, seed=0)
for _ in range(2): tr.step(b)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(b); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=60))
if "--cpu" in sys.argv: print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=60))
t=time.perf_counter()
for _ in range(3): tr.step(b)
torch.cuda.synchronize(); print('ms/iter', (time.perf_counter()-t)/3*1e3)
