"""Generate tests/golden/losses.npz by IMPORTING the reference's loss functions in this container (CPU):
  gaussian_splatting/utils/loss_utils.py:17-63  l1_loss, l2_loss, ssim
  main/loss_utils/sobel_loss.py:19-30           sobel_loss
sobel_loss.py creates its kernels with device="cuda" at import time; torch.tensor's device kwarg is dropped while the
module is imported so the very same code runs on the CPU.  Only the arrays travel.
Run:  python tests/golden/make_loss_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "gaussian_splatting"))
sys.path.insert(0, os.path.join(REF, "main"))

_tensor = torch.tensor
def _tensor_cpu(*a, **k):
    k.pop("device", None)
    return _tensor(*a, **k)
torch.tensor = _tensor_cpu
from utils.loss_utils import l1_loss, l2_loss, ssim          # noqa: E402
from loss_utils.sobel_loss import sobel_loss                  # noqa: E402
torch.tensor = _tensor


def main():
    g = torch.Generator().manual_seed(41)
    out = {}
    for tag, (H, W) in (("a", (40, 56)), ("b", (33, 70))):   # not multiples of the 32-pixel tile
        tgt = torch.rand(3, H, W, generator=g)
        img = (tgt + 0.15 * torch.randn(3, H, W, generator=g)).clamp(0, 1).requires_grad_(True)
        l1, l2 = l1_loss(img, tgt), l2_loss(img, tgt)
        s, smap = ssim(img, tgt)
        sb, sbmap = sobel_loss(img, tgt)
        total = 0.2 * l1 + 0.1 * l2 + 0.5 * (1.0 - s) + 0.2 * sb   # train_pano2gaussian_decoder.py:36-40,261
        total.backward()
        out.update({f"{tag}_image": img.detach().numpy(), f"{tag}_target": tgt.numpy(),
                    f"{tag}_terms": np.array([l1.item(), l2.item(), 1.0 - s.item(), sb.item(), total.item()], np.float64),
                    f"{tag}_ssim_map": smap.detach().numpy(), f"{tag}_sobel_map": sbmap.detach().numpy(),
                    f"{tag}_grad": img.grad.numpy()})
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    print("wrote losses.npz", {k: v.shape for k, v in out.items() if k.startswith("a_")})


if __name__ == "__main__":
    main()
