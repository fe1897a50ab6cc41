"""Image losses: the test-side torch evaluation (tests/_torch_losses.py) against vectors produced by the reference's own functions
(tests/golden/losses.npz, made by tests/golden/make_loss_golden.py), and the fused HIP loss against both."""
import os

import numpy as np
import pytest
import torch

from gaussian_gan_decoder_amd import losses as L
import _torch_losses as TL

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "losses.npz"))
W4 = dict(l1_weight=0.2, l2_weight=0.1, ssim_weight=0.5, sobel_weight=0.2)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_torch_losses_match_reference_vectors(tag):
    img = torch.from_numpy(GOLD[f"{tag}_image"]).requires_grad_(True)
    tgt = torch.from_numpy(GOLD[f"{tag}_target"])
    s, smap = TL.ssim(img, tgt)
    sb, sbmap = TL.sobel_loss(img, tgt)
    np.testing.assert_allclose(smap.detach().numpy(), GOLD[f"{tag}_ssim_map"], atol=1e-6)
    np.testing.assert_allclose(sbmap.detach().numpy(), GOLD[f"{tag}_sobel_map"], atol=1e-5, rtol=1e-6)
    total, terms = TL.image_loss_torch(img, tgt, **W4)
    np.testing.assert_allclose(terms.detach().numpy(), GOLD[f"{tag}_terms"][:4], rtol=1e-6, atol=1e-7)
    total.backward()
    np.testing.assert_allclose(img.grad.numpy(), GOLD[f"{tag}_grad"], atol=1e-8, rtol=1e-5)


def test_fused_loss_requires_gpu_tensor():
    with pytest.raises(RuntimeError):
        L.fused_image_loss(torch.zeros(3, 8, 8), torch.zeros(3, 8, 8))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_fused_loss_matches_reference_vectors(tag):
    """Loss terms within 2e-6 relative, gradient within 1e-5 of its max (fp32 stencil sums in a different order)."""
    dev = torch.device("cuda:0")
    img = torch.from_numpy(GOLD[f"{tag}_image"]).to(dev).requires_grad_(True)
    tgt = torch.from_numpy(GOLD[f"{tag}_target"]).to(dev)
    total, terms = L.fused_image_loss(img, tgt, **W4)
    np.testing.assert_allclose(terms.cpu().numpy().astype(np.float64), GOLD[f"{tag}_terms"], rtol=2e-6, atol=1e-7)
    (3.0 * total).backward()
    g = GOLD[f"{tag}_grad"] * 3.0
    assert np.abs(img.grad.cpu().numpy() - g).max() <= 1e-5 * np.abs(g).max()


@pytest.mark.gpu
def test_fused_loss_matches_torch_at_training_size():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    tgt = torch.rand(3, 512, 512, generator=g).to(dev)
    img0 = (tgt.cpu() + 0.1 * torch.randn(3, 512, 512, generator=g)).to(dev)
    a = img0.clone().requires_grad_(True)
    b = img0.clone().requires_grad_(True)
    ta, terms_a = L.fused_image_loss(a, tgt, **W4)
    tb, terms_b = TL.image_loss_torch(b, tgt, **W4)
    ta.backward(); tb.backward()
    # the four terms are fp32 sums over 786 k pixels: per-workgroup partial sums combined by float atomics in the fused
    # kernel (their order varies from run to run), a tree in torch -- both carry ~sqrt(N) eps of rounding (one run in a few dozen
    # landed between 1e-5 and 5e-5 of the torch value)
    np.testing.assert_allclose(terms_a[:4].cpu().numpy(), terms_b.detach().cpu().numpy(), rtol=5e-5)
    assert (a.grad - b.grad).abs().max().item() <= 1e-5 * b.grad.abs().max().item()


def test_perceptual_stand_in_is_fixed_and_differentiable():
    """The LPIPS slot of the train step: a fixed, seeded VGG16-shaped trunk (narrow here to keep the CPU test fast)."""
    a, b = L.PerceptualStandIn(seed=5, width_div=16), L.PerceptualStandIn(seed=5, width_div=16)
    g = torch.Generator().manual_seed(1)
    tgt = torch.rand(3, 64, 64, generator=g)
    img = (tgt + 0.05 * torch.randn(3, 64, 64, generator=g)).requires_grad_(True)
    va, vb = a(img, tgt), b(img, tgt)
    assert torch.equal(va, vb) and float(va) > 0 and float(a(tgt, tgt)) == 0.0
    va.backward()
    assert img.grad is not None and torch.isfinite(img.grad).all() and img.grad.abs().max() > 0
    assert all(not p.requires_grad for p in a.parameters()) and len(a.convs) == 13
