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
        out = torch.nn.functional.grid_sample(plane_features, proj[..., :2].unsqueeze(1).float(), mode="bilinear",
                                              padding_mode="zeros", align_corners=False)  # [3, C, 1, M]
        return out.permute(0, 3, 2, 1).reshape(n_planes, M, -1)
    D = int(triplane_depth)
    _, CD, H, W = plane_features.shape
    C = CD // D
    grid5 = plane_features.view(n_planes, C, D, H, W)
    out = torch.nn.functional.grid_sample(grid5, proj.unsqueeze(1).unsqueeze(2).float(), mode="bilinear",
                                          padding_mode="zeros", align_corners=False)       # [3, C, 1, 1, M]
    return out.permute(0, 4, 3, 2, 1).reshape(n_planes, M, C)


class _TrigridMeanFn(torch.autograd.Function):
    """mean over the 3 planes of the tri-grid sample_from_planes, through the HIP gather kernel (csrc/ggd_triplane.hip,
    ggd_trigrid_*): grids are handed over channel-last [3][D][H][W][C]."""

    @staticmethod
    def forward(ctx, plane_features, coordinates, box_warp, axes_mode, depth):
        import ctypes as C
        from . import _capi
        dev = plane_features.device
        n_planes, CD, H, W = plane_features.shape
        Cc = CD // depth
        grids_cl = plane_features.view(n_planes, Cc, depth, H, W).permute(0, 2, 3, 4, 1).contiguous().float()
        pos = coordinates.contiguous().float()
        out = torch.empty((pos.shape[0], Cc), dtype=torch.float32, device=dev)
        cx = _capi.context_for(dev)
        with torch.cuda.device(dev):
            cx.check(cx.lib.ggd_trigrid_forward(cx.handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream),
                                                C.c_void_p(grids_cl.data_ptr()), Cc, depth, H, W, axes_mode,
                                                C.c_void_p(pos.data_ptr()), pos.shape[0], float(box_warp),
                                                C.c_void_p(out.data_ptr())))
        ctx.save_for_backward(pos)
        ctx.meta = (Cc, depth, H, W, float(box_warp), axes_mode)
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes as C
        from . import _capi
        (pos,) = ctx.saved_tensors
        Cc, depth, H, W, box_warp, axes_mode = ctx.meta
        dev = pos.device
        dout = dout.contiguous().float()
        dgrids_cl = torch.empty((3, depth, H, W, Cc), dtype=torch.float32, device=dev)
        cx = _capi.context_for(dev)
        with torch.cuda.device(dev):
            cx.check(cx.lib.ggd_trigrid_backward(cx.handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), Cc, depth,
                                                 H, W, axes_mode, C.c_void_p(pos.data_ptr()), pos.shape[0], box_warp,
                                                 C.c_void_p(dout.data_ptr()), C.c_void_p(dgrids_cl.data_ptr())))
        return dgrids_cl.permute(0, 4, 1, 2, 3).reshape(3, Cc * depth, H, W), None, None, None, None


class _TriplaneMeanFn(torch.autograd.Function):
    """mean over the 3 planes of sample_from_planes, through the HIP gather kernel (csrc/ggd_triplane.hip)."""

    @staticmethod
    def forward(ctx, plane_features, coordinates, box_warp):
        import ctypes as C
        from . import _capi
        dev = plane_features.device
        n_planes, Cc, H, W = plane_features.shape
        planes_cl = plane_features.permute(0, 2, 3, 1).contiguous().float()
        pos = coordinates.contiguous().float()
        out = torch.empty((pos.shape[0], Cc), dtype=torch.float32, device=dev)
        cx = _capi.context_for(dev)
        with torch.cuda.device(dev):
            cx.check(cx.lib.ggd_triplane_forward(cx.handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream),
                                                 C.c_void_p(planes_cl.data_ptr()), Cc, H, W, C.c_void_p(pos.data_ptr()),
                                                 pos.shape[0], float(box_warp), C.c_void_p(out.data_ptr())))
        ctx.save_for_backward(pos)
        ctx.meta = (Cc, H, W, float(box_warp))
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes as C
        from . import _capi
        (pos,) = ctx.saved_tensors
        Cc, H, W, box_warp = ctx.meta
        dev = pos.device
        dout = dout.contiguous().float()
        dplanes_cl = torch.empty((3, H, W, Cc), dtype=torch.float32, device=dev)
        cx = _capi.context_for(dev)
        with torch.cuda.device(dev):
            cx.check(cx.lib.ggd_triplane_backward(cx.handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), Cc, H, W,
                                                  C.c_void_p(pos.data_ptr()), pos.shape[0], box_warp,
                                                  C.c_void_p(dout.data_ptr()), C.c_void_p(dplanes_cl.data_ptr())))
        return dplanes_cl.permute(0, 3, 1, 2), None, None


def triplane_mean(plane_features: torch.Tensor, coordinates: torch.Tensor, box_warp: float = 1.0,
                  plane_axes: str = "eg3d", triplane_depth=None) -> torch.Tensor:
    """== sample_from_planes(...).mean(0): [3, C(*D), H, W], [M, 3] -> [M, C].  HIP gather kernels on the GPU (C a power
    of two <= 64, 3 planes; 2-D EG3D form or the PanoHead tri-grid); the torch ops on the CPU (the decoder is host-side
    PyTorch per the north_star, so a CPU path exists for tests -- unlike the rasterizer)."""
    C = plane_features.shape[1] // (1 if triplane_depth is None else int(triplane_depth))
    ok = plane_features.is_cuda and plane_features.shape[0] == 3 and C <= 64 and (C & (C - 1)) == 0
    if ok and triplane_depth is None and plane_axes == "eg3d":
        return _TriplaneMeanFn.apply(plane_features, coordinates, box_warp)
    if ok and triplane_depth is not None:
        return _TrigridMeanFn.apply(plane_features, coordinates, box_warp, {"eg3d": 0, "panohead": 1}[plane_axes],
                                    int(triplane_depth))
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

    def forward(self, feature_planes, init_position):
        # the 5 heads all average the three planes' samples
        pf = triplane_mean(feature_planes, init_position, self.box_warp, self.plane_axes, self.triplane_depth)
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
