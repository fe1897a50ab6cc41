"""Seeded synthetic scenes for parity tests and the benchmark (definition: SURVEY.md section 8d).

  "cube"  : xyz ~ U[-0.5, 0.5]^3  -- the decoder's domain (main/decoder_utils/target_dataloader.py:100-102,
            box_warp = 1 in PanoHead/train.py:330)
  "shell" : directions uniform on S^2, radius 0.3 * clip(N(1, 0.1), 0, 1)  -- head-like
            (mirrors target_dataloader.py:115-118)
  log-scale ~ N(-6.0, 0.5) per axis (the decoder emits -softplus(s+5)-2.5, sequential_decoder_reverse.py:35-36),
  rotation ~ N(0,1)^4, opacity logit ~ N(0, 2), SH-DC ~ N(0, 1) ([P,1,3], degree 0),
  bg = (0.55717, 0.52256, 0.51045) (main/train_pano2gaussian_decoder.py:132).
  Camera: look-at geometry h = v = pi/2, radius 2.7 (main/decoder_utils/camera.py:7), fov 12 degrees
  (the reference samples U[5,17], target_dataloader.py:71), znear 0.01, zfar 10 (cameras.py:76).

Everything is generated on the CPU with torch.Generator(seed) and then moved, so a scene is identical on the
GPU box and in the CPU-only container.  Raw (pre-activation) attributes are returned next to the activated
ones because the reference's GaussianModel getters (gaussian_model.py:100-121) sit between the two.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from .cameras import CustomCam, look_at_cam2world

BG_COLOR = (0.55717, 0.52256, 0.51045)


@dataclass
class SyntheticScene:
    xyz: torch.Tensor          # [P,3]
    log_scales: torch.Tensor   # [P,3]  raw
    rot_raw: torch.Tensor      # [P,4]  raw
    opacity_logit: torch.Tensor  # [P,1] raw
    features_dc: torch.Tensor  # [P,1,3]
    bg: torch.Tensor           # [3]
    cam: CustomCam
    size: int

    @property
    def scales(self):
        return torch.exp(self.log_scales)

    @property
    def rotations(self):
        return torch.nn.functional.normalize(self.rot_raw)

    @property
    def opacities(self):
        return torch.sigmoid(self.opacity_logit)

    def gaussian_model(self, requires_grad: bool = False):
        """The scene as the reference's container class (`GaussianModel(0)` with raw, pre-activation attributes, as
        main/train_pano2gaussian_decoder.py:215-227 fills it)."""
        from .gaussian_model import GaussianModel
        pc = GaussianModel(0)
        mk = (lambda t: t.clone().requires_grad_(True)) if requires_grad else (lambda t: t)
        pc._xyz, pc._scaling, pc._rotation = mk(self.xyz), mk(self.log_scales), mk(self.rot_raw)
        pc._opacity, pc._features_dc = mk(self.opacity_logit), mk(self.features_dc)
        return pc

    def to(self, device):
        cam = make_camera(self.size, self.cam._fov_deg, self.cam._h, self.cam._v, device=device)
        return SyntheticScene(self.xyz.to(device), self.log_scales.to(device), self.rot_raw.to(device),
                              self.opacity_logit.to(device), self.features_dc.to(device), self.bg.to(device),
                              cam, self.size)


def make_camera(size: int, fov_deg: float = 12.0, h: float = math.pi / 2, v: float = math.pi / 2,
                radius: float = 2.7, device="cpu") -> CustomCam:
    cam2world = look_at_cam2world(h, v, radius)  # always built on the CPU: bit-identical everywhere
    fov = fov_deg / 360 * 2 * math.pi            # as main/train_pano2gaussian_decoder.py:230
    cam = CustomCam(size=size, fov=fov, extr=cam2world)
    for name in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
        setattr(cam, name, getattr(cam, name).to(device).contiguous())
    cam._fov_deg, cam._h, cam._v = fov_deg, h, v
    return cam


def make_scene(P: int, size: int, kind: str = "cube", seed: int = 0, fov_deg: float = 12.0,
               log_scale_mean: float = -6.0, log_scale_std: float = 0.5, device="cpu",
               h: float = math.pi / 2, v: float = math.pi / 2) -> SyntheticScene:
    g = torch.Generator().manual_seed(seed)
    if kind == "cube":
        xyz = torch.rand(P, 3, generator=g) - 0.5
    elif kind == "shell":
        d = torch.randn(P, 3, generator=g)
        d = d / d.norm(dim=1, keepdim=True)
        r = 0.3 * torch.clip(1.0 + 0.1 * torch.randn(P, 1, generator=g), 0.0, 1.0)
        xyz = d * r
    else:
        raise ValueError(f"unknown scene kind {kind!r}")
    log_scales = log_scale_mean + log_scale_std * torch.randn(P, 3, generator=g)
    rot_raw = torch.randn(P, 4, generator=g)
    opacity_logit = 2.0 * torch.randn(P, 1, generator=g)
    features_dc = torch.randn(P, 1, 3, generator=g)
    bg = torch.tensor(BG_COLOR, dtype=torch.float32)
    scene = SyntheticScene(xyz.contiguous(), log_scales, rot_raw, opacity_logit, features_dc, bg,
                           make_camera(size, fov_deg, h, v), size)
    return scene.to(device) if str(device) != "cpu" else scene


def make_dL_dpix(size: int, seed: int = 1, device="cpu") -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(3, size, size, generator=g).to(device)
