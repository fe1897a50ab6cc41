"""Per-point Gaussian decoder (stays in PyTorch-ROCm per the north_star; SURVEY.md section 8f row 1 is the
optional fused MFMA kernel) -- the caller on the INPUT side of the raster hot path.

Mirrors main/decoder_models/base_decoder.py:8-27 (`Decoder`: in -> 128 -> 128 -> 128 -> out, GELU, the three planes
averaged) and main/decoder_models/sequential_decoder_reverse.py:27-36,61-86 (`SequentialDecoderReverse`: chained
colour -> opacity -> rotation -> scale -> xyz heads, scale activation -softplus(s+5)-2.5, xyz = head*0.01 + position)
with the same parameter names (`color_decoder.backbone.0.weight`, ...), so a reference state_dict loads unchanged.
The frozen/finetuned GAN that produces the feature planes (PanoHead / EG3D TriPlaneGenerator) is out of scope; the
decoder here consumes a feature-plane tensor directly ([3, C, H, W], the EG3D layout of
eg3d/training/volumetric_rendering/renderer.py:23-65).
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
from torch import nn

# plane axes of eg3d/training/volumetric_rendering/renderer.py:23-38
_PLANE_AXES = torch.tensor([[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                            [[1, 0, 0], [0, 0, 1], [0, 1, 0]],
                            [[0, 0, 1], [1, 0, 0], [0, 1, 0]]], dtype=torch.float32)


def sample_from_planes(plane_features: torch.Tensor, coordinates: torch.Tensor, box_warp: float = 1.0) -> torch.Tensor:
    """plane_features [3, C, H, W], coordinates [M, 3] in [-box_warp/2, box_warp/2] -> [3, M, C]
    (bilinear, zero padding, align_corners=False; same projection as renderer.py:40-65)."""
    n_planes, C, H, W = plane_features.shape
    M = coordinates.shape[0]
    coords = (2.0 / box_warp) * coordinates
    inv = torch.linalg.inv(_PLANE_AXES.to(coords.device, coords.dtype))          # [3,3,3]
    proj = torch.einsum("mc,pcd->pmd", coords, inv)[..., :2]                        # [3, M, 2]
    out = torch.nn.functional.grid_sample(plane_features, proj.unsqueeze(1).float(), mode="bilinear",
                                          padding_mode="zeros", align_corners=False)  # [3, C, 1, M]
    return out.permute(0, 3, 2, 1).reshape(n_planes, M, C)


class Decoder(nn.Module):
    def __init__(self, n_features, out_features=3, hidden_dim=128):
        super().__init__()
        self.backbone = nn.Sequential(
            nn.Linear(n_features, hidden_dim), nn.GELU(),
            nn.Linear(hidden_dim, hidden_dim), nn.GELU(),
            nn.Linear(hidden_dim, hidden_dim), nn.GELU(),
            nn.Linear(hidden_dim, out_features))

    def forward(self, triplane_features, gaussian_features):
        x = torch.concat([triplane_features.mean(0), gaussian_features], dim=-1)
        return self.backbone(x)


class SequentialDecoderReverse(nn.Module):
    """Feature planes + positions -> raw Gaussian attributes (xyz, scale, rotation, opacity, color)."""

    def __init__(self, plane_channels=32, hidden_dim=128, position_dim=3, box_warp=1.0):
        super().__init__()
        f = plane_channels + position_dim
        self.box_warp = box_warp
        self.color_decoder = Decoder(f, 3, hidden_dim)
        self.opacity_decoder = Decoder(f + 3, 1, hidden_dim)
        self.rotation_decoder = Decoder(f + 4, 4, hidden_dim)
        self.scale_decoder = Decoder(f + 8, 3, hidden_dim)
        self.xyz_decoder = Decoder(f + 11, 3, hidden_dim)
        self.scale_activation = nn.Softplus()

    def activate_scale(self, scale):
        return -self.scale_activation(scale + 5) - 2.5

    def forward(self, feature_planes, init_position):
        pf = sample_from_planes(feature_planes, init_position, self.box_warp)
        info = init_position
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
