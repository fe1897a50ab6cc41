"""TEST-ONLY PyTorch evaluation of the image losses of the decoder training step: the checker of the fused HIP loss kernel
(gaussian_gan_decoder_amd/losses.py::fused_image_loss) and the loss of the CPU (gloo) data-parallel tests.  Follows the
formulas of the reference's gaussian_splatting/utils/loss_utils.py:17-63 (l1 / l2 / ssim: the standard 11x11 sigma-1.5
Gaussian-window SSIM) and main/loss_utils/sobel_loss.py:19-30; pinned by vectors generated from the reference's own
functions (tests/golden/losses.npz).  Never imported by the product."""
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


