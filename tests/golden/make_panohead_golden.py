"""Generate tests/golden/panohead_fixture.npz: BASELINE config 1 ("PanoHead triplane+decoder forward, 1 random z, CPU
PyTorch -- emit Gaussian xyz/scale/rot/SH") from the REFERENCE's own code, imported from /root/reference on the CPU:

  PanoHead/training/triplane.py:18-293   TriPlaneGenerator (random-initialised with the configuration of
                                         PanoHead/train.py:302-333: ffhq, triplane_size 256, triplane_depth 3, box_warp 1)
    G.mapping(z, 0, truncation_psi) -> G.synthesis(ws, c, noise_mode="const")["feature_planes"]   ([1, 3, 96, 256, 256])
  PanoHead/training/volumetric_rendering/renderer.py:47-58   sample_from_planes with triplane_depth (3-D grid_sample over
                                         the C x D tri-grid, PanoHead plane axes)
  main/decoder_models/sequential_decoder_reverse.py:38-86    SequentialDecoderReverse.forward, every line the reference's

Run in the build container:  python tests/golden/make_panohead_golden.py
Only arrays travel.  The 75 MB plane tensor is not stored: the positions are drawn inside a small cube, so the eight taps
of every sample fall inside a (2 * PAD + n)-texel window of each plane; the fixture keeps those three windows (all 96
channels) and their offsets, and the tests paste them into zero planes of the full 256 x 256 size.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "PanoHead"))

import dnnlib  # noqa: E402
from training.triplane import TriPlaneGenerator  # noqa: E402
from main.decoder_models.sequential_decoder_reverse import SequentialDecoderReverse  # noqa: E402

SIZE, DEPTH, NPTS = 256, 3, 2048
CUBE_LO, CUBE_HI = (0.02, -0.11, 0.05), (0.12, -0.01, 0.15)    # positions: a 0.1-wide cube inside the [-0.5, 0.5]^3 box


def build_generator():
    # PanoHead/train.py:302-333 (cfg = ffhq) + its G_kwargs (:236-262); superresolution as for 512 x 512 output
    rendering = dict(image_resolution=512, disparity_space_sampling=False, clamp_mode="softplus",
                     superresolution_module="training.superresolution.SuperresolutionHybrid8XDC",
                     c_gen_conditioning_zero=True, gpc_reg_prob=None, c_scale=1.0, superresolution_noise_mode="none",
                     density_reg=0.25, density_reg_p_dist=0.004, reg_type="l1", decoder_lr_mul=1.0,
                     decoder_activation="none", use_torgb_raw=True, triplane_size=SIZE, triplane_depth=DEPTH,
                     trans_reg=0.0, use_background=True, sr_antialias=True, depth_resolution=48,
                     depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1, avg_camera_radius=2.7,
                     avg_camera_pivot=[0, 0, 0.2], mask_guidance=False)
    G = TriPlaneGenerator(z_dim=512, c_dim=25, w_dim=512, img_resolution=512, img_channels=3, sr_num_fp16_res=0,
                          mapping_kwargs=dnnlib.EasyDict(num_layers=2), rendering_kwargs=rendering,
                          sr_kwargs=dnnlib.EasyDict(channel_base=32768, channel_max=512, fused_modconv_default="inference_only"),
                          channel_base=32768, channel_max=512, fused_modconv_default="inference_only", num_fp16_res=0,
                          conv_clamp=None)
    return G.eval().requires_grad_(False)


def main():
    torch.manual_seed(7)
    G = build_generator()
    print("generator parameters:", sum(p.numel() for p in G.parameters()))
    g = torch.Generator().manual_seed(8)
    z = torch.randn(1, 512, generator=g)
    cam = torch.zeros(1, 25)
    lo, hi = torch.tensor(CUBE_LO), torch.tensor(CUBE_HI)
    pos = lo + (hi - lo) * torch.rand(NPTS, 3, generator=g)
    torch.manual_seed(9)
    dec = SequentialDecoderReverse(G, hidden_dim=128, use_xyz_embedding=False, use_gen_finetune=False, device="cpu")
    dec.triplane_sr = "None"   # set by main/train_pano2gaussian_decoder.py, read at sequential_decoder_reverse.py:58
    with torch.no_grad():
        for n, p in dec.named_parameters():
            if "decoder" in n and p.dim() == 2:
                p.mul_(1.5)   # default nn.Linear init gives ~1e-2 outputs; exercise the GELUs
        out = dec(z, cam, pos, 0.7)
        ws = G.mapping(z, torch.zeros_like(cam), truncation_psi=0.7)
        planes = G.synthesis(ws, cam, noise_mode="const")["feature_planes"][0]     # [3, 96, 256, 256]
    axes = G.renderer.plane_axes.numpy()
    # the window of every plane the samples touch: projected coordinates (coords @ inv(axes), renderer.py:31-45), first two
    # components -> (W index, H index) with align_corners = False
    proj = torch.einsum("mc,pcd->pmd", 2.0 * pos, torch.linalg.inv(G.renderer.plane_axes))
    windows, offsets = [], []
    for p in range(3):
        ix = ((proj[p, :, 0] + 1) * SIZE - 1) / 2
        iy = ((proj[p, :, 1] + 1) * SIZE - 1) / 2
        x0, x1 = int(torch.floor(ix.min())) - 1, int(torch.floor(ix.max())) + 3
        y0, y1 = int(torch.floor(iy.min())) - 1, int(torch.floor(iy.max())) + 3
        assert x0 >= 0 and y0 >= 0 and x1 <= SIZE and y1 <= SIZE
        windows.append(planes[p, :, y0:y1, x0:x1].numpy().copy())
        offsets.append((y0, x0))
    hw = max(w.shape[1] for w in windows), max(w.shape[2] for w in windows)
    win = np.zeros((3, 32 * DEPTH, hw[0], hw[1]), np.float32)
    for p, w in enumerate(windows):
        win[p, :, :w.shape[1], :w.shape[2]] = w
    sd = {"sd_" + k: v.numpy() for k, v in dec.state_dict().items() if "decoder" in k and not k.startswith("G.")}
    np.savez_compressed(os.path.join(HERE, "panohead_fixture.npz"), plane_windows=win.astype(np.float16).astype(np.float32)
                        if False else win, window_offsets=np.asarray(offsets, np.int32), plane_axes=axes,
                        size=np.int32(SIZE), depth=np.int32(DEPTH), positions=pos.numpy(), z=z.numpy(),
                        color=out.color.numpy(), opacity=out.opacity.numpy(), rotation=out.rotation.numpy(),
                        scale=out.scale.numpy(), xyz=out.xyz.numpy(),
                        plane_abs_mean=np.float32(planes.abs().mean().item()), **sd)
    print("wrote panohead_fixture.npz; windows", win.shape, "offsets", offsets, "|planes| mean", float(planes.abs().mean()))


if __name__ == "__main__":
    main()
