"""Per-point Gaussian decoder (stays in PyTorch-ROCm per the north_star; SURVEY.md section 8f row 1 is the
optional fused MFMA kernel) -- the caller on the INPUT side of the raster hot path.

Mirrors main/decoder_models/base_decoder.py:8-27 (`Decoder`: in -> 128 -> 128 -> 128 -> out, GELU, the three planes
averaged) and main/decoder_models/sequential_decoder_reverse.py:27-36,61-86 (`SequentialDecoderReverse`: chained
colour -> opacity -> rotation -> scale -> xyz heads, scale activation -softplus(s+5)-2.5, xyz = head*0.01 + position)
with the same parameter names (`color_decoder.backbone.0.weight`, ...), so a reference state_dict loads unchanged.
The frozen/finetuned GAN that produces the feature planes (PanoHead / EG3D TriPlaneGenerator) is out of scope; the
decoder here consumes a feature-plane tensor directly: [3, C, H, W] tri-planes (EG3D,
eg3d/training/volumetric_rendering/renderer.py:23-65) or [3, C * D, H, W] tri-grids (PanoHead,
PanoHead/training/volumetric_rendering/renderer.py:47-58); tests/golden/panohead_fixture.npz pins the PanoHead form
against the reference's own generator + decoder run end to end on one seeded z.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
from torch import nn

# plane axes: eg3d/training/volumetric_rendering/renderer.py:23-38 and PanoHead/training/volumetric_rendering/renderer.py
# (generate_planes): they differ in the third plane -- EG3D projects it to (z, x), PanoHead to (y, z)
PLANE_AXES = {
    "eg3d": torch.tensor([[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                          [[1, 0, 0], [0, 0, 1], [0, 1, 0]],
                          [[0, 0, 1], [1, 0, 0], [0, 1, 0]]], dtype=torch.float32),
    "panohead": torch.tensor([[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                              [[1, 0, 0], [0, 0, 1], [0, 1, 0]],
                              [[0, 1, 0], [0, 0, 1], [1, 0, 0]]], dtype=torch.float32),
}
_PLANE_AXES = PLANE_AXES["eg3d"]


def sample_from_planes(plane_features: torch.Tensor, coordinates: torch.Tensor, box_warp: float = 1.0,
                       plane_axes: str = "eg3d", triplane_depth=None) -> torch.Tensor:
    """plane_features [3, C, H, W] (EG3D) or [3, C * D, H, W] with triplane_depth = D (PanoHead tri-grid),
    coordinates [M, 3] in [-box_warp/2, box_warp/2] -> [3, M, C]; bilinear / trilinear, zero padding,
    align_corners=False.  triplane_depth=None: the 2-D form of eg3d/.../renderer.py:40-65; an integer: the 3-D
    grid_sample of PanoHead/training/volumetric_rendering/renderer.py:47-58 over the C x D grid, all three projected
    coordinates used (also for D = 1, where the depth coordinate attenuates the sample as it does in the reference)."""
    n_planes = plane_features.shape[0]
    M = coordinates.shape[0]
    coords = (2.0 / box_warp) * coordinates
    inv = torch.linalg.inv(PLANE_AXES[plane_axes].to(coords.device, coords.dtype))   # [3,3,3]
    proj = torch.einsum("mc,pcd->pmd", coords, inv)                                    # [3, M, 3]
    if triplane_depth is None:
        out = torch.nn.functional.grid_sample(plane_features, proj[..., :2].unsqueeze(1).to(plane_features.dtype), mode="bilinear",
                                              padding_mode="zeros", align_corners=False)  # [3, C, 1, M]
        return out.permute(0, 3, 2, 1).reshape(n_planes, M, -1)
    D = int(triplane_depth)
    _, CD, H, W = plane_features.shape
    C = CD // D
    grid5 = plane_features.view(n_planes, C, D, H, W)
    out = torch.nn.functional.grid_sample(grid5, proj.unsqueeze(1).unsqueeze(2).to(plane_features.dtype), mode="bilinear",
                                          padding_mode="zeros", align_corners=False)       # [3, C, 1, 1, M]
    return out.permute(0, 4, 3, 2, 1).reshape(n_planes, M, C)


def planes_channels_last(plane_features: torch.Tensor, triplane_depth=None) -> torch.Tensor:
    """[3, C, H, W] -> [3, H, W, C] (EG3D tri-planes) or [3, C * D, H, W] -> [3, D, H, W, C] (PanoHead tri-grids, channel
    index c * D + d as in PanoHead/training/volumetric_rendering/renderer.py:52): the layout the HIP gather kernels
    read -- one texel's C channels are one contiguous line.  Plain differentiable torch ops (the permute back is
    autograd's)."""
    if triplane_depth is None:
        return plane_features.permute(0, 2, 3, 1).contiguous().float()
    n_planes, CD, H, W = plane_features.shape
    D = int(triplane_depth)
    return plane_features.view(n_planes, CD // D, D, H, W).permute(0, 2, 3, 4, 1).contiguous().float()


class _PlanesGatherFn(torch.autograd.Function):
    """mean over the 3 planes of sample_from_planes through the HIP kernels (csrc/ggd_triplane.hip: ggd_planes_gather /
    ggd_planes_scatter).  planes_cl: channel-last [3, H, W, C] (depth 0) or [3, D, H, W, C]; coordinates [B, N, 3];
    mod: None or per-scene modulations [B, max(D, 1), C] of the planes (scene b samples planes_cl * mod[b], never
    materialised).  One autograd node for all B scenes: the backward ADDS every scene's scatter into ONE gradient buffer
    (accumulate = 1) instead of B full-size gradients that autograd would sum pass by pass (a tri-grid of
    3 x 96 x 256 x 256 is 75 MB)."""

    @staticmethod
    def forward(ctx, planes_cl, coordinates, mod, box_warp, axes_mode, depth):
        import ctypes as C
        from . import _capi
        dev = planes_cl.device
        if planes_cl.dtype != torch.float32 or not planes_cl.is_contiguous():
            raise ValueError("planes_cl: contiguous float32 expected (planes_channels_last)")
        Cc, H, W = planes_cl.shape[-1], planes_cl.shape[-3], planes_cl.shape[-2]
        pos = coordinates.contiguous().float()
        B, n = pos.shape[0], pos.shape[1]
        out = torch.empty((B * n, Cc), dtype=torch.float32, device=dev)
        if mod is not None:
            mod = mod.contiguous().float()
            if mod.numel() != B * max(depth, 1) * Cc:
                raise ValueError("mod: [B, max(D, 1), C] expected")
        cx = _capi.context_for(dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            for b in range(B):
                cx.check(cx.lib.ggd_planes_gather(
                    cx.handle, stream, C.c_void_p(planes_cl.data_ptr()), Cc, depth, H, W, axes_mode,
                    C.c_void_p(mod[b].data_ptr()) if mod is not None else None, C.c_void_p(pos[b].data_ptr()), n,
                    float(box_warp), C.c_void_p(out[b * n:].data_ptr())))
        ctx.save_for_backward(pos, mod)
        ctx.meta = (Cc, depth, H, W, float(box_warp), axes_mode, tuple(planes_cl.shape))
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes as C
        from . import _capi
        pos, mod = ctx.saved_tensors
        Cc, depth, H, W, box_warp, axes_mode, shape = ctx.meta
        dev = pos.device
        B, n = pos.shape[0], pos.shape[1]
        dout = dout.contiguous().float()
        dplanes = torch.empty(shape, dtype=torch.float32, device=dev)
        cx = _capi.context_for(dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            for b in range(B):   # the first call zero-fills, the others add
                cx.check(cx.lib.ggd_planes_scatter(
                    cx.handle, stream, Cc, depth, H, W, axes_mode,
                    C.c_void_p(mod[b].data_ptr()) if mod is not None else None, C.c_void_p(pos[b].data_ptr()), n, box_warp,
                    C.c_void_p(dout[b * n:].data_ptr()), C.c_void_p(dplanes.data_ptr()), 0 if b == 0 else 1))
        return dplanes, None, None, None, None, None


def planes_gather(planes_cl: torch.Tensor, coordinates: torch.Tensor, box_warp: float = 1.0, plane_axes: str = "eg3d",
                  triplane_depth=None, mod=None) -> torch.Tensor:
    """The HIP gather on channel-last planes (planes_channels_last): mean over the three planes of the bilinear /
    trilinear sample of `planes_cl * mod`.  coordinates [M, 3] -> [M, C] (mod: [max(D, 1), C] or None), or a batch of
    scenes sharing the planes: coordinates [B, N, 3], mod [B, max(D, 1), C] -> [B * N, C]."""
    depth = 0 if triplane_depth is None else int(triplane_depth)
    if depth == 0 and plane_axes != "eg3d":
        raise ValueError("the 2-D tri-plane gather has the EG3D plane axes only")
    if not planes_cl.is_cuda:
        raise RuntimeError("planes_gather is a HIP kernel: CUDA tensors required (sample_from_planes is the torch form)")
    if coordinates.dim() == 2:
        coordinates = coordinates.unsqueeze(0)
        mod = mod.unsqueeze(0) if mod is not None else None
    return _PlanesGatherFn.apply(planes_cl, coordinates, mod, box_warp, {"eg3d": 0, "panohead": 1}[plane_axes], depth)


def triplane_mean(plane_features: torch.Tensor, coordinates: torch.Tensor, box_warp: float = 1.0,
                  plane_axes: str = "eg3d", triplane_depth=None) -> torch.Tensor:
    """== sample_from_planes(...).mean(0): [3, C(*D), H, W], [M, 3] -> [M, C].  HIP gather kernels on the GPU (C a power
    of two <= 64, 3 planes; 2-D EG3D form or the PanoHead tri-grid); the torch ops on the CPU (the decoder is host-side
    PyTorch per the north_star, so a CPU path exists for tests -- unlike the rasterizer)."""
    C = plane_features.shape[1] // (1 if triplane_depth is None else int(triplane_depth))
    ok = plane_features.is_cuda and plane_features.shape[0] == 3 and C <= 64 and (C & (C - 1)) == 0
    if ok and (triplane_depth is not None or plane_axes == "eg3d"):
        return planes_gather(planes_channels_last(plane_features, triplane_depth), coordinates, box_warp, plane_axes,
                             triplane_depth)
    return sample_from_planes(plane_features, coordinates, box_warp, plane_axes, triplane_depth).mean(0)


class _TallLinearFn(torch.autograd.Function):
    """y = x W^T + b for a TALL x ([N, in], N ~ 5e5).  Same forward as F.linear; the backward computes the weight
    gradient dW = dy^T x as a batched split-K product (the N reduction cut into chunks -> torch.bmm -> sum) instead
    of one [out, N] x [N, in] GEMM: on MI355X the single tall-skinny reduction GEMM is dispatched to a tile shape
    that leaves most CUs idle (~1 ms per layer at N = 5e5, measured), the batched form keeps all of them busy."""
    CHUNK = 2048

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ weight if ctx.needs_input_grad[0] else None
        n = x.shape[0]
        c = _TallLinearFn.CHUNK
        main = (n // c) * c
        dw = None
        if main:
            dw = torch.bmm(dy[:main].view(-1, c, dy.shape[1]).transpose(1, 2), x[:main].view(-1, c, x.shape[1])).sum(0)
        if main < n:
            tail = dy[main:].t() @ x[main:]
            dw = tail if dw is None else dw + tail
        return dx, dw, dy.sum(0)


class TallLinear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) with the split-K weight-gradient of _TallLinearFn."""

    def forward(self, x):
        if x.dim() == 2 and x.shape[0] >= 4 * _TallLinearFn.CHUNK and x.requires_grad | self.weight.requires_grad:
            return _TallLinearFn.apply(x, self.weight, self.bias)
        return super().forward(x)


class Decoder(nn.Module):
    def __init__(self, n_features, out_features=3, hidden_dim=128):
        super().__init__()
        self.backbone = nn.Sequential(
            TallLinear(n_features, hidden_dim), nn.GELU(),
            TallLinear(hidden_dim, hidden_dim), nn.GELU(),
            TallLinear(hidden_dim, hidden_dim), nn.GELU(),
            TallLinear(hidden_dim, out_features))

    def forward(self, triplane_features, gaussian_features):
        # reference signature: triplane_features [3, N, C] (averaged here); a pre-averaged [N, C] is accepted too
        feats = triplane_features.mean(0) if triplane_features.dim() == 3 else triplane_features
        return self.backbone(torch.concat([feats, gaussian_features], dim=-1))


def embed_positions(x: torch.Tensor, num_freqs: int = 10, include_input: bool = True) -> torch.Tensor:
    """Positional encoding of main/decoder_utils/pos_encoding.py (the decoders' use_xyz_embedding option): [x,
    sin(f x), cos(f x) for f in linspace(2^0, 2^(num_freqs - 1), num_freqs)] concatenated on the last axis ->
    3 + 3 * 2 * num_freqs = 63 columns for the reference's settings (frequencies spaced LINEARLY, as the reference's
    default log_sampling=False gives)."""
    freqs = torch.linspace(2.0 ** 0.0, 2.0 ** (num_freqs - 1), steps=num_freqs, dtype=x.dtype, device=x.device)
    parts = [x] if include_input else []
    for f in freqs:
        parts += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(parts, -1)


def _info_dim(position_dim: int, use_xyz_embedding: bool, num_freqs: int = 10) -> int:
    if position_dim != 3:
        raise ValueError("positions are 3-vectors")
    return 3 + 3 * 2 * num_freqs if use_xyz_embedding else 3


class SequentialDecoderReverse(nn.Module):
    """Feature planes + positions -> raw Gaussian attributes (xyz, scale, rotation, opacity, color)."""

    def __init__(self, plane_channels=32, hidden_dim=128, position_dim=3, box_warp=1.0, plane_axes="eg3d",
                 triplane_depth=None, use_xyz_embedding=False):
        """plane_axes / triplane_depth: which generator produced the planes -- "eg3d", None: EG3D tri-planes [3, 32, H, W];
        "panohead", D: PanoHead tri-grids [3, 32 * D, H, W] (G.rendering_kwargs["triplane_depth"],
        sequential_decoder_reverse.py:42-50).  use_xyz_embedding: the reference's optional positional encoding of the
        positions (sequential_decoder_reverse.py:21-22,63-66; off by default in main/train_pano2gaussian_decoder.py:46);
        PyTorch modules only -- the fused MFMA decoder takes the 3-vector form."""
        super().__init__()
        self.use_xyz_embedding = bool(use_xyz_embedding)
        f = plane_channels + _info_dim(position_dim, self.use_xyz_embedding)
        self.box_warp = box_warp
        self.plane_axes = plane_axes
        self.triplane_depth = triplane_depth
        self.color_decoder = Decoder(f, 3, hidden_dim)
        self.opacity_decoder = Decoder(f + 3, 1, hidden_dim)
        self.rotation_decoder = Decoder(f + 4, 4, hidden_dim)
        self.scale_decoder = Decoder(f + 8, 3, hidden_dim)
        self.xyz_decoder = Decoder(f + 11, 3, hidden_dim)
        self.scale_activation = nn.Softplus()

    def activate_scale(self, scale):
        return -self.scale_activation(scale + 5) - 2.5

    def forward(self, feature_planes, init_position, features=None):
        """features: the plane-mean features [N, C] when the caller has gathered them already (planes_gather on
        channel-last planes: the training step); feature_planes is not read then."""
        # the 5 heads all average the three planes' samples
        pf = features if features is not None else \
            triplane_mean(feature_planes, init_position, self.box_warp, self.plane_axes, self.triplane_depth)
        info = embed_positions(init_position) if self.use_xyz_embedding else init_position
        color = self.color_decoder(pf, info)
        info = torch.concat([info, color], dim=-1)
        opacity = self.opacity_decoder(pf, info)
        info = torch.concat([info, opacity], dim=-1)
        rotation = self.rotation_decoder(pf, info)
        info = torch.concat([info, rotation], dim=-1)
        scale = self.activate_scale(self.scale_decoder(pf, info))
        info = torch.concat([info, scale], dim=-1)
        xyz = self.xyz_decoder(pf, info) * 0.01 + init_position
        return SimpleNamespace(xyz=xyz, scale=scale, rotation=rotation, opacity=opacity, color=color)

    def get_params_custom(self):
        params = []
        for m in (self.xyz_decoder, self.scale_decoder, self.rotation_decoder, self.opacity_decoder,
                  self.color_decoder):
            params += list(m.parameters())
        return params


class SequentialDecoder(nn.Module):
    """The forward-ordered chain (main/decoder_models/sequential_decoder.py:27-84, decoder_type "sequential"):
    xyz -> scale -> rotation -> opacity -> colour, every head seeing [plane_mean, position, earlier outputs]; scale
    activation -softplus(s + 5) - 2 (NOT -2.5 as in the reversed chain).  Same parameter names as the reference."""

    def __init__(self, plane_channels=32, hidden_dim=128, position_dim=3, box_warp=1.0, plane_axes="eg3d",
                 triplane_depth=None, use_xyz_embedding=False):
        super().__init__()
        self.use_xyz_embedding = bool(use_xyz_embedding)
        f = plane_channels + _info_dim(position_dim, self.use_xyz_embedding)
        self.box_warp, self.plane_axes, self.triplane_depth = box_warp, plane_axes, triplane_depth
        self.xyz_decoder = Decoder(f, 3, hidden_dim)
        self.scale_decoder = Decoder(f + 3, 3, hidden_dim)
        self.rotation_decoder = Decoder(f + 6, 4, hidden_dim)
        self.opacity_decoder = Decoder(f + 10, 1, hidden_dim)
        self.color_decoder = Decoder(f + 11, 3, hidden_dim)
        self.scale_activation = nn.Softplus()

    def activate_scale(self, scale):
        return -self.scale_activation(scale + 5) - 2

    def forward(self, feature_planes, init_position):
        pf = triplane_mean(feature_planes, init_position, self.box_warp, self.plane_axes, self.triplane_depth)
        info = embed_positions(init_position) if self.use_xyz_embedding else init_position
        xyz = self.xyz_decoder(pf, info) * 0.01 + init_position
        info = torch.concat([info, xyz], dim=-1)
        scale = self.activate_scale(self.scale_decoder(pf, info))
        info = torch.concat([info, scale], dim=-1)
        rotation = self.rotation_decoder(pf, info)
        info = torch.concat([info, rotation], dim=-1)
        opacity = self.opacity_decoder(pf, info)
        info = torch.concat([info, opacity], dim=-1)
        color = self.color_decoder(pf, info)
        return SimpleNamespace(xyz=xyz, scale=scale, rotation=rotation, opacity=opacity, color=color)

    get_params_custom = SequentialDecoderReverse.get_params_custom


class ParallelDecoder(nn.Module):
    """Five independent heads on [plane_mean, position] (main/decoder_models/parallel_decoder.py:27-80, decoder_type
    "parallel"); scale activation -softplus(s + 5) - 2."""

    def __init__(self, plane_channels=32, hidden_dim=128, position_dim=3, box_warp=1.0, plane_axes="eg3d",
                 triplane_depth=None, use_xyz_embedding=False):
        super().__init__()
        self.use_xyz_embedding = bool(use_xyz_embedding)
        f = plane_channels + _info_dim(position_dim, self.use_xyz_embedding)
        self.box_warp, self.plane_axes, self.triplane_depth = box_warp, plane_axes, triplane_depth
        self.xyz_decoder = Decoder(f, 3, hidden_dim)
        self.scale_decoder = Decoder(f, 3, hidden_dim)
        self.rotation_decoder = Decoder(f, 4, hidden_dim)
        self.opacity_decoder = Decoder(f, 1, hidden_dim)
        self.color_decoder = Decoder(f, 3, hidden_dim)
        self.scale_activation = nn.Softplus()

    def activate_scale(self, scale):
        return -self.scale_activation(scale + 5) - 2

    def forward(self, feature_planes, init_position):
        pf = triplane_mean(feature_planes, init_position, self.box_warp, self.plane_axes, self.triplane_depth)
        pos = embed_positions(init_position) if self.use_xyz_embedding else init_position
        return SimpleNamespace(xyz=self.xyz_decoder(pf, pos) * 0.01 + init_position,
                               scale=self.activate_scale(self.scale_decoder(pf, pos)),
                               rotation=self.rotation_decoder(pf, pos), opacity=self.opacity_decoder(pf, pos),
                               color=self.color_decoder(pf, pos))

    get_params_custom = SequentialDecoderReverse.get_params_custom
