import sys, os, time, json, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd.train import DecoderTrainer, make_scene_batch
dev = torch.device("cuda:0")
spg, pts = 4, int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
batches = [make_scene_batch(list(range(spg)), pts, 512, dev, seed=i) for i in range(2)]
faulthandler.dump_traceback_later(60, exit=True)
kw = dict(scene_streams=sys.argv[1] == "1")
tr = DecoderTrainer(dev, n_scenes_total=spg, image_size=512, fused_activations=True, fused_decoder=False, **kw)
t0 = time.perf_counter()
for i in range(4):
    tr.step(batches[i % 2]); torch.cuda.synchronize(); print("step", i, round(time.perf_counter() - t0, 3), flush=True)
