"""Image losses of the decoder training step (main/train_pano2gaussian_decoder.py:246-261).

PyTorch re-statements (device-agnostic, same signatures and return values as the reference):
  l1_loss, l2_loss, ssim      gaussian_splatting/utils/loss_utils.py:17-63
  sobel_loss                  main/loss_utils/sobel_loss.py:19-30  (the reference builds its kernels on "cuda" at import)
and `fused_image_loss`: all four terms and d(loss)/d(image) in three HIP launches (csrc/ggd_imgloss.hip), as one
autograd node.  The perceptual / identity terms of the reference need external networks (VGG, ArcFace) and are out of
scope.
"""
from __future__ import annotations

import ctypes as C
from math import exp

import torch
import torch.nn.functional as F


def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


def _gaussian(window_size: int, sigma: float) -> torch.Tensor:
    g = torch.tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)],
                     dtype=torch.float32)
    return g / g.sum()


def create_window(window_size: int, channel: int) -> torch.Tensor:
    w1 = _gaussian(window_size, 1.5).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def ssim(img1, img2, window_size: int = 11, size_average: bool = True):
    """Returns (mean SSIM, SSIM map) like the reference (loss_utils.py:33-63)."""
    channel = img1.size(-3)
    window = create_window(window_size, channel).to(device=img1.device, dtype=img1.dtype)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    if size_average:
        return ssim_map.mean(), ssim_map
    return ssim_map.mean(1).mean(1).mean(1), ssim_map


_SOBEL_Y = [[1, 2, 1], [0, 0, 0], [-1, -2, -1]]
_SOBEL_X = [[1, 0, -1], [2, 0, -2], [1, 0, -1]]


def sobel_loss(render, target):
    """Returns (mean squared Sobel difference, its map); the 3x3 kernels sum over the three channels (sobel_loss.py:15-16)."""
    kx = torch.tensor(_SOBEL_X, dtype=torch.float32, device=render.device).unsqueeze(0).expand(1, 3, 3, 3)
    ky = torch.tensor(_SOBEL_Y, dtype=torch.float32, device=render.device).unsqueeze(0).expand(1, 3, 3, 3)
    rx = F.conv2d(render.unsqueeze(0), kx, stride=1, padding=1)
    tx = F.conv2d(target.unsqueeze(0), kx, stride=1, padding=1)
    ry = F.conv2d(render.unsqueeze(0), ky, stride=1, padding=1)
    ty = F.conv2d(target.unsqueeze(0), ky, stride=1, padding=1)
    diff = torch.square(rx - tx) + torch.square(ry - ty)
    return diff.mean(), diff


def image_loss_torch(image, target, l1_weight=0.2, l2_weight=0.1, ssim_weight=0.5, sobel_weight=0.2):
    """The reference's weighted sum (train_pano2gaussian_decoder.py:246-261, defaults :36-40) from the torch ops."""
    terms = torch.stack([l1_loss(image, target), l2_loss(image, target), 1.0 - ssim(image, target)[0],
                         sobel_loss(image, target)[0]])
    w = torch.tensor([l1_weight, l2_weight, ssim_weight, sobel_weight], dtype=terms.dtype, device=terms.device)
    return (terms * w).sum(), terms


class _FusedImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, target, weights):
        from . import _capi
        if not image.is_cuda:
            raise RuntimeError("fused_image_loss is a HIP kernel: CUDA tensors required (use image_loss_torch on CPU)")
        if image.dim() != 3 or image.shape[0] != 3 or image.shape != target.shape:
            raise ValueError("image and target must both be [3,H,W]")
        dev = image.device
        img = image.contiguous().float()
        tgt = target.contiguous().float()
        H, W = int(img.shape[1]), int(img.shape[2])
        cx = _capi.context_for(dev)
        nbytes = cx.lib.ggd_image_loss_tmp_bytes(W, H)
        tmp = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        terms = torch.empty((5,), dtype=torch.float32, device=dev)
        grad = torch.empty_like(img)
        w4 = (C.c_float * 4)(*[float(x) for x in weights])
        with torch.cuda.device(dev):
            cx.check(cx.lib.ggd_image_loss(cx.handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), W, H,
                                           C.c_void_p(img.data_ptr()), C.c_void_p(tgt.data_ptr()), w4,
                                           C.c_void_p(terms.data_ptr()), C.c_void_p(grad.data_ptr()),
                                           C.c_void_p(tmp.data_ptr()), nbytes))
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(terms)
        return terms[4].clone(), terms

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        (grad,) = ctx.saved_tensors
        return grad * g_total, None, None


def fused_image_loss(image, target, l1_weight=0.2, l2_weight=0.1, ssim_weight=0.5, sobel_weight=0.2):
    """(total, terms[5] = L1, L2, 1-SSIM, Sobel, total): same value and d/d(image) as `image_loss_torch`, three HIP
    launches instead of ~60 torch kernels.  `target` receives no gradient."""
    return _FusedImageLoss.apply(image, target, (l1_weight, l2_weight, ssim_weight, sobel_weight))
