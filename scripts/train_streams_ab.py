"""A/B of DecoderTrainer(scene_streams=...) on the config-3 step (4 scenes x 500 k points, 512^2, PanoHead tri-grids, stand-ins on).
usage: python scripts/train_streams_ab.py [--fp32]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd.train import DecoderTrainer, make_scene_batch
dev = torch.device("cuda:0")
b = [make_scene_batch([0, 1, 2, 3], 500000, 512, dev, seed=i) for i in range(2)]
for rep in range(2):
    for ss in (False, True):
        torch.manual_seed(0)
        tr = DecoderTrainer(dev, n_scenes_total=4, image_size=512, fused_activations=True, fused_decoder=True, scene_streams=ss,
                            decoder_precision="fp32" if "--fp32" in sys.argv else "bf16", plane_axes="panohead", triplane_depth=3,
                            backbone_params=29_570_000 - 3 * 96 * 256 * 256, perceptual_weight=1.0)
        for i in range(4):
            l = tr.step(b[i % 2])
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(12):
            l = tr.step(b[i % 2])
        torch.cuda.synchronize()
        print(f"scene_streams={ss}: {(time.perf_counter() - t) / 12 * 1e3:.2f} ms/iter, last loss {l:.6f}", flush=True)
        del tr; torch.cuda.empty_cache()
