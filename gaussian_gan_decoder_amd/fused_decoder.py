"""Inference-time fused decoder: tri-plane gather + the 5 chained MLP heads as ONE bf16-MFMA kernel
(csrc/ggd_mlp.hip; SURVEY.md section 8f row 1, BASELINE config 3).

`FusedDecoder(decoder)` wraps a `SequentialDecoderReverse` (same parameters; call `.repack()` after they change) and
returns the same namespace (xyz, scale, rotation, opacity, color).  No autograd: training keeps the PyTorch modules.
Numerics: weights and activations are rounded to bf16 at every layer input, accumulation / bias / GELU in fp32
(erf-form GELU with a polynomial erf, |err| < 5.7e-5)."""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import torch

from . import _capi
from .decoder import SequentialDecoderReverse, triplane_mean

HID = 128
ROW1, ROW2 = 64 + 8, 128 + 8        # bf16 elements per padded weight row
# position (g, e) inside a 32-wide k block  <-  k = 4g+e (e < 4) or 16+4g+(e-4): the order the MFMA C/D layout
# hands a layer's outputs to the next layer's B operand (csrc/ggd_mlp.hip header)
_PERM32 = [(4 * g + e) if e < 4 else (16 + 4 * g + (e - 4)) for g in range(4) for e in range(8)]


def _permute_blocks(w: torch.Tensor) -> torch.Tensor:
    K = w.shape[1]
    idx = torch.tensor([32 * s + p for s in range(K // 32) for p in _PERM32], device=w.device)
    return w[:, idx]


def pack_weights(decoder: SequentialDecoderReverse) -> torch.Tensor:
    """-> uint8 tensor of ggd_decoder_packed_bytes(): per head [W1 128x72 | W2 128x136 | W3 128x136 | W4 16x136] bf16,
    then b1 b2 b3 [128] and b4 [16] fp32."""
    dev = next(decoder.parameters()).device
    chunks = []
    for head in (decoder.color_decoder, decoder.opacity_decoder, decoder.rotation_decoder, decoder.scale_decoder,
                 decoder.xyz_decoder):
        l1, l2, l3, l4 = head.backbone[0], head.backbone[2], head.backbone[4], head.backbone[6]
        if l1.out_features != HID or l1.in_features < 35 or l1.in_features > 32 + 16 or l4.out_features > 16:
            raise ValueError("fused decoder supports hidden_dim 128, 32 plane channels, <= 13 chained inputs")
        w1 = torch.zeros(HID, 64, device=dev)
        w1[:, :l1.in_features] = l1.weight.detach().float()        # cols 0..31 planes, 32.. = info slots
        w4 = torch.zeros(16, HID, device=dev)
        w4[:l4.out_features] = l4.weight.detach().float()
        b4 = torch.zeros(16, device=dev)
        b4[:l4.out_features] = l4.bias.detach().float()
        rows = []
        for w, row in ((w1, ROW1), (l2.weight.detach().float(), ROW2), (l3.weight.detach().float(), ROW2), (w4, ROW2)):
            wp = torch.zeros(w.shape[0], row, device=dev)
            wp[:, :w.shape[1]] = _permute_blocks(w)
            rows.append(wp.to(torch.bfloat16).contiguous().view(torch.uint8).reshape(-1))
        biases = torch.cat([l1.bias.detach().float(), l2.bias.detach().float(), l3.bias.detach().float(), b4])
        chunks += rows + [biases.contiguous().view(torch.uint8).reshape(-1)]
    packed = torch.cat(chunks).contiguous()
    expect = _capi.load().ggd_decoder_packed_bytes()
    if packed.numel() != expect:
        raise RuntimeError(f"packed decoder image is {packed.numel()} bytes, library expects {expect}")
    return packed


class FusedDecoder:
    def __init__(self, decoder: SequentialDecoderReverse):
        self.decoder = decoder
        self.box_warp = decoder.box_warp
        self.repack()

    def repack(self):
        self.packed = pack_weights(self.decoder)

    @torch.no_grad()
    def decode_features(self, feats: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        """feats [N,32], positions [N,3] (CUDA fp32) -> attrs [N,16] (layout: include/ggd_raster.h)."""
        if not feats.is_cuda:
            raise RuntimeError("the fused decoder is a HIP kernel: CUDA tensors required (use the PyTorch decoder on CPU)")
        dev = feats.device
        feats = feats.contiguous().float()
        positions = positions.contiguous().float()
        n = positions.shape[0]
        attrs = torch.empty((n, 16), dtype=torch.float32, device=dev)
        cx = _capi.context_for(dev)
        with torch.cuda.device(dev):
            cx.check(cx.lib.ggd_decoder_forward(cx.handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream),
                                                C.c_void_p(feats.data_ptr()), C.c_void_p(positions.data_ptr()), n,
                                                C.c_void_p(self.packed.data_ptr()), C.c_void_p(attrs.data_ptr())))
        return attrs

    @torch.no_grad()
    def __call__(self, feature_planes: torch.Tensor, init_position: torch.Tensor) -> SimpleNamespace:
        feats = triplane_mean(feature_planes, init_position, self.box_warp)
        a = self.decode_features(feats, init_position)
        return SimpleNamespace(color=a[:, 0:3], opacity=a[:, 3:4], rotation=a[:, 4:8], scale=a[:, 8:11], xyz=a[:, 11:14])
