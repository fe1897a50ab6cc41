"""Camera prologue of the raster hot path.

Host-side mirror of the reference's camera classes; the matrix LAYOUT defined here is the contract the HIP
kernels consume (flat element [4*c + r] = entry (r, c) of the column-vector matrix):

  * getProjectionMatrix  <- gaussian_splatting/utils/graphics_utils.py:52-74
  * CustomCam            <- gaussian_splatting/scene/cameras.py:75-92   (EG3D cam2world -> 3DGS view/proj)
  * MiniCam              <- gaussian_splatting/scene/cameras.py:61-72
  * look_at_cam2world    <- main/camera_utils.py:63-93,125-146 (LookAtPoseSampler + create_cam2world_matrix),
                            deterministic (stddev = 0) form used by the synthetic benchmark scenes.

Unlike the reference (which hard-codes `.cuda()`), every tensor lives on the device of the `extr` argument.
"""
from __future__ import annotations

import math

import torch


def getProjectionMatrix(znear: float, zfar: float, fovX: float, fovY: float) -> torch.Tensor:
    """Symmetric-frustum perspective matrix (column-vector form, w_clip = z_view).

    Evaluated in Python doubles with the reference's expression order, then rounded once to fp32, so the result
    is bit-identical to graphics_utils.py:52-74 (pinned by tests/golden/cameras.json).  Note the reference's
    z row: P[2,2] = (zfar+znear)/(zfar-znear), P[2,3] = -zfar*znear/(zfar-znear).
    """
    half_w = math.tan(fovX / 2) * znear  # right = -left
    half_h = math.tan(fovY / 2) * znear  # top = -bottom
    rows = [
        [2.0 * znear / (half_w - -half_w), 0.0, (half_w + -half_w) / (half_w - -half_w), 0.0],
        [0.0, 2.0 * znear / (half_h - -half_h), (half_h + -half_h) / (half_h - -half_h), 0.0],
        [0.0, 0.0, 1.0 * (zfar + znear) / (zfar - znear), -(zfar * znear) / (zfar - znear)],
        [0.0, 0.0, 1.0, 0.0],
    ]
    return torch.tensor(rows, dtype=torch.float32)


class CustomCam:
    """Square-image pinhole camera built from an EG3D-style cam2world pose (reference: cameras.py:75-92).

    Note (kept from the reference, cameras.py:92): `camera_center` is row 3 of world_view_transform, not of its
    inverse; it only matters for SH degree > 0, which the decoder path never uses.
    """

    def __init__(self, size, fov, extr, znear=0.01, zfar=10.0):
        self.image_width = size
        self.image_height = size
        self.FoVy = fov
        self.FoVx = fov
        self.znear = znear
        self.zfar = zfar
        extr = torch.as_tensor(extr, dtype=torch.float32)
        self.world_view_transform = extr.T.inverse().contiguous()
        self.projection_matrix = getProjectionMatrix(znear=znear, zfar=zfar, fovX=fov, fovY=fov) \
            .transpose(0, 1).to(extr.device)
        self.full_proj_transform = (self.world_view_transform.unsqueeze(0)
                                    .bmm(self.projection_matrix.unsqueeze(0))).squeeze(0).contiguous()
        self.camera_center = self.world_view_transform[3, :3].contiguous()


class MiniCam:
    def __init__(self, width, height, fovy, fovx, znear, zfar, world_view_transform, full_proj_transform):
        self.image_width = width
        self.image_height = height
        self.FoVy = fovy
        self.FoVx = fovx
        self.znear = znear
        self.zfar = zfar
        self.world_view_transform = world_view_transform
        self.full_proj_transform = full_proj_transform
        view_inv = torch.inverse(self.world_view_transform)
        self.camera_center = view_inv[3][:3]


def _normalize(v: torch.Tensor) -> torch.Tensor:
    return v / torch.norm(v, dim=-1, keepdim=True)


def look_at_cam2world(h: float = math.pi / 2, v: float = math.pi / 2, radius: float = 2.7,
                      device="cpu") -> torch.Tensor:
    """cam2world [4,4] of a camera on a sphere of `radius` looking at the origin, y-up, no roll."""
    v = min(max(v, 1e-5), math.pi - 1e-5)
    theta = torch.tensor([[h]], dtype=torch.float32)
    phi = torch.arccos(1 - 2 * torch.tensor([[v]], dtype=torch.float32) / math.pi)
    origin = torch.zeros(1, 3)
    origin[:, 0:1] = radius * torch.sin(phi) * torch.cos(math.pi - theta)
    origin[:, 2:3] = radius * torch.sin(phi) * torch.sin(math.pi - theta)
    origin[:, 1:2] = radius * torch.cos(phi)
    forward = _normalize(-origin)
    up = torch.tensor([0, 1, 0], dtype=torch.float32).expand_as(forward)
    right = -_normalize(torch.cross(up, forward, dim=-1))
    up = _normalize(torch.cross(forward, right, dim=-1))
    rot = torch.eye(4).unsqueeze(0)
    rot[:, :3, :3] = torch.stack((right, up, forward), dim=-1)
    trans = torch.eye(4).unsqueeze(0)
    trans[:, :3, 3] = origin
    return (trans @ rot)[0].to(device)
