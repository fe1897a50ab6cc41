"""Scratch: time DecoderTrainer.step variants (stand-ins on/off, streams on/off)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd.train import DecoderTrainer, make_scene_batch
dev = torch.device("cuda:0")
spg, pts = 4, 500_000
batches = [make_scene_batch(list(range(spg)), pts, 512, dev, seed=i) for i in range(2)]
REST = 29_570_000 - 3 * 32 * 256 * 256
def run(name, iters=6, **kw):
    t0 = time.perf_counter()
    tr = DecoderTrainer(dev, n_scenes_total=spg, image_size=512, fused_activations=True, fused_decoder=True, **kw)
    for i in range(2): tr.step(batches[i % 2])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for i in range(iters): tr.step(batches[i % 2])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(json.dumps(dict(cfg=name, setup_and_warm_s=round(t1 - t0, 2), ms_per_iter=round((t2 - t1) / iters * 1e3, 2))), flush=True)
    del tr
run("r1 config, streams", scene_streams=True)
run("r1 config, no streams", scene_streams=False)
run("+backbone payload", backbone_params=REST)
run("+perceptual", perceptual_weight=1.0, iters=3)
run("+both", backbone_params=REST, perceptual_weight=1.0, iters=3)
