"""Host-side logic around the fused decoder that runs without a GPU: weight-image packing (sizes the library expects,
the k-permutation shared by the forward and the transposed image) and the single-node attrs split."""
import numpy as np
import torch

from gaussian_gan_decoder_amd import _capi
from gaussian_gan_decoder_amd.decoder import SequentialDecoderReverse
from gaussian_gan_decoder_amd import fused_decoder as FD


def test_weight_images_have_the_sizes_the_library_expects():
    lib = _capi.load()
    dec = SequentialDecoderReverse()
    assert FD.pack_weights(dec).numel() == lib.ggd_decoder_packed_bytes()
    assert FD.pack_weights_t(dec).numel() == lib.ggd_decoder_packed_t_bytes()
    assert lib.ggd_decoder_zbuf_bytes(1000) == 5 * 3 * 1008 * 128 * 2   # 16-point blocks: N rounded up to a multiple of 16
    assert lib.ggd_decoder_zbuf_bytes(1024) == 5 * 3 * 1024 * 128 * 2
    assert lib.ggd_decoder_wgrad_floats() == 5 * (128 * 64 + 128 + 2 * (128 * 128 + 128) + 16 * 128 + 16)


def test_k_permutation_is_a_bijection_on_every_32_block():
    p = torch.tensor(FD._PERM32)
    assert sorted(p.tolist()) == list(range(32))
    x = torch.arange(2 * 64, dtype=torch.float32).reshape(2, 64)
    y = FD._permute_blocks(x)
    for blk in range(2):
        np.testing.assert_array_equal(np.sort(y[:, 32 * blk:32 * blk + 32].numpy(), axis=1),
                                      x[:, 32 * blk:32 * blk + 32].numpy())


def test_split_attrs_backward_assembles_one_buffer():
    torch.manual_seed(0)
    a = torch.randn(2, 7, 16, dtype=torch.float64, requires_grad=True)
    parts = FD.split_attrs(a)
    assert len(parts) == 2 and [tuple(t.shape) for t in parts[0]] == [(7, 3), (7, 3), (7, 4), (7, 1), (7, 3)]
    w = [[torch.randn_like(t) for t in scene] for scene in parts]
    loss = sum((t * wt).sum() for scene, ws in zip(parts, w) for t, wt in zip(scene, ws) if t.shape[1] != 4)
    loss.backward()                                   # the rotation outputs get no gradient: their columns must be zero
    ref = torch.zeros_like(a)
    for b in range(2):
        for (lo, hi), wt in zip(FD._SplitAttrs.COLS, w[b]):
            if hi - lo != 4:
                ref[b, :, lo:hi] = wt
    np.testing.assert_array_equal(a.grad.numpy(), ref.numpy())
