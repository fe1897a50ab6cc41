"""Image losses of the decoder training step (main/train_pano2gaussian_decoder.py:246-261).

`fused_image_loss`: L1, L2, 1 - SSIM and the Sobel term (gaussian_splatting/utils/loss_utils.py:17-63,
main/loss_utils/sobel_loss.py:19-30) and d(loss)/d(image) in three HIP launches (csrc/ggd_imgloss.hip), as one autograd
node; pinned by vectors from the reference's own functions (tests/golden/losses.npz).  The PyTorch evaluation of the
same terms lives under tests/ (the checker).
`PerceptualStandIn`: the slot of the reference's LPIPS term (main/loss_utils/lpips.py:6-34: VGG16 features of the image
and the target at 256 x 256, squared distance).  The pretrained VGG is an external download and out of scope; the
stand-in is a FIXED, seeded, random-initialised network of the same shape (VGG16's 13 3x3 convolutions, taps after
conv1_2 / 2_2 / 3_3 / 4_3 / 5_3, channel-normalised, per-channel weights), so that the training step carries the same
amount of convolution work and the same kind of gradient path into the rendered image.  It runs in PyTorch-ROCm (MIOpen)
like the reference's own VGG.  The identity term needs ArcFace and stays out.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn.functional as F


class _FusedImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, target, weights):
        from . import _capi
        if not image.is_cuda:
            raise RuntimeError("fused_image_loss is a HIP kernel: HIP device tensors required (there is no CPU form in the package)")
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
    """(total, terms[5] = L1, L2, 1-SSIM, Sobel, total): same value and d/d(image) as the torch evaluation under tests/, three HIP
    launches instead of ~60 torch kernels.  `target` receives no gradient."""
    return _FusedImageLoss.apply(image, target, (l1_weight, l2_weight, ssim_weight, sobel_weight))


class PerceptualStandIn(torch.nn.Module):
    """perc(target, image) of main/loss_utils/lpips.py:16-34 with a fixed random VGG16-shaped trunk (see the module
    docstring).  forward(image [3,H,W] or [B,3,H,W] in [0,1], target likewise) -> scalar (summed over the batch, as the
    reference evaluates the whole batch in one VGG call); gradients flow into `image` only."""
    CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)
    TAPS = (1, 3, 6, 9, 12)       # index of the convolution after whose ReLU a feature map is taken

    def __init__(self, seed: int = 1234, width_div: int = 1):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        convs, cin = [], 3
        for v in self.CFG:
            if v == "M":
                continue
            cout = max(4, v // width_div)
            conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
            with torch.no_grad():   # He initialisation keeps the activations O(1) through the 13 layers
                conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (9 * cin)) ** 0.5)
                conv.bias.zero_()
            convs.append(conv)
            cin = cout
        self.convs = torch.nn.ModuleList(convs)
        self.lin = torch.nn.ParameterList([torch.nn.Parameter(torch.rand(self.convs[t].out_channels, generator=g))
                                           for t in self.TAPS])
        self.requires_grad_(False)

    def features(self, img):
        x = ((img if img.dim() == 4 else img.unsqueeze(0)) * 2.0 - 1.0)
        if x.shape[2] > 256:
            x = F.interpolate(x, size=(256, 256), mode="area")     # lpips.py:23-26
        feats, ci = [], 0
        for v in self.CFG:
            if v == "M":
                x = F.max_pool2d(x, 2)
                continue
            x = F.relu(self.convs[ci](x))
            if ci in self.TAPS:
                k = self.TAPS.index(ci)
                n = x / (x.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
                feats.append((n * (self.lin[k] / (x.shape[2] * x.shape[3])).sqrt()[None, :, None, None]).flatten(1))
            ci += 1
        return torch.cat(feats, 1)

    def forward(self, image, target):
        with torch.no_grad():
            ft = self.features(target)
        return (self.features(image) - ft).square().sum()
