"""Time the plane gather / scatter kernels (csrc/ggd_triplane.hip) at the training step's size: 500 k head-shell points per
scene, 256 x 256 planes, 32 channels -- EG3D tri-planes and PanoHead tri-grids (depth 3).  GGD_PLANES_BWD=tile|atomic selects
the older backward forms for an A/B inside separate processes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_gan_decoder_amd.decoder import planes_channels_last, planes_gather
from gaussian_gan_decoder_amd.train import make_scene_batch

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 500_000))
K = 20


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3


batch = make_scene_batch([0, 1, 2, 3], N, 64, dev, seed=0)
mode = os.environ.get("GGD_PLANES_BWD", "sorted")
for name, depth, axes in (("tri-plane", None, "eg3d"), ("tri-grid D=3", 3, "panohead")):
    D = depth or 1
    planes = torch.randn(3, 32 * D, 256, 256, device=dev)
    pcl = planes_channels_last(planes, depth).requires_grad_(True)
    pos = batch.positions[0]
    mod = (1 + 0.25 * torch.randn(D, 32, device=dev))
    gout = torch.randn(N, 32, device=dev)
    for m in (None, mod):
        out = planes_gather(pcl, pos, 1.0, axes, depth, mod=m)
        t_f = timed(lambda: planes_gather(pcl.detach(), pos, 1.0, axes, depth, mod=m))
        t_b = timed(lambda: torch.autograd.grad(out, pcl, gout, retain_graph=True))
        print(f"{name:14s} mod={'yes' if m is not None else 'no ':3s} N={N}: gather {t_f:7.1f} us   scatter {t_b:7.1f} us  (mode {mode})")
    # 4 scenes in one node
    mods = (1 + 0.25 * torch.randn(4, D, 32, device=dev))
    out = planes_gather(pcl, batch.positions, 1.0, axes, depth, mod=mods)
    g4 = torch.randn(4 * N, 32, device=dev)
    t_f = timed(lambda: planes_gather(pcl.detach(), batch.positions, 1.0, axes, depth, mod=mods))
    t_b = timed(lambda: torch.autograd.grad(out, pcl, g4, retain_graph=True))
    print(f"{name:14s} 4 scenes: gather {t_f:7.1f} us   scatter {t_b:7.1f} us  (mode {mode})")
