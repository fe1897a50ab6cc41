"""Which PyTorch (non-native) ops of one reference-precision train step cost GPU time, grouped by input shape.
usage: python scripts/profile_train_shapes.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd.train import DecoderTrainer, make_scene_batch
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
tr = DecoderTrainer(dev, n_scenes_total=4, image_size=512, fused_activations=True, fused_decoder=True, decoder_precision='fp32',
                    plane_axes="panohead", triplane_depth=3, backbone_params=29_570_000 - 3 * 32 * 3 * 256 * 256, perceptual_weight=1.0)
b = make_scene_batch([0, 1, 2, 3], 500000, 512, dev, seed=0)
for _ in range(4): tr.step(b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    tr.step(b); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None: t = e.self_cuda_time_total
    if t > 0: rows.append((t, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total self device time us", tot)
rows = [r for r in rows if r[2].startswith("aten::") and "convolution" not in r[2]]
print("aten (no conv) total us", sum(r[0] for r in rows))
for t, c, k, sh in rows[:60]:
    print(f"{t:9.1f} us  x{c:<3d} {k[:44]:44s} {sh}")
