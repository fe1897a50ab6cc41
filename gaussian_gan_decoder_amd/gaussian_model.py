"""Minimal Gaussian attribute container for the decoder path.

Mirrors the part of gaussian_splatting/scene/gaussian_model.py the raster hot path touches: the constructor
(:47-63), the activation getters that form the rasterizer's input prologue (:100-124: exp / normalize / sigmoid /
pass-through) and get_covariance (:29-33,123-124).  The decoder overwrites `_xyz/_scaling/_rotation/_opacity/
_features_dc` every step (main/train_pano2gaussian_decoder.py:223-227), so densification, the optimizer set-up
and the simple_knn / plyfile imports of the reference class are intentionally absent.
"""
from __future__ import annotations

import torch


def build_rotation(q: torch.Tensor) -> torch.Tensor:
    """[N,4] (w,x,y,z), normalised here -> [N,3,3] (reference: utils/general_utils.py:78-99)."""
    q = q / torch.sqrt((q * q).sum(dim=1, keepdim=True))
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=1).view(-1, 3, 3)


def build_covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
    """Sigma = L L^T with L = R diag(mod*s); returns the 6 upper-triangular entries [N,6]
    (reference: gaussian_model.py:29-33, general_utils.py:64-73,101-110)."""
    L = build_rotation(rotation) * (scaling_modifier * scaling)[:, None, :]
    cov = L @ L.transpose(1, 2)
    return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=1)


class GaussianModel:
    def __init__(self, sh_degree: int):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self._xyz = torch.empty(0)
        self._features_dc = torch.empty(0)
        self._features_rest = torch.empty(0)
        self._scaling = torch.empty(0)
        self._rotation = torch.empty(0)
        self._opacity = torch.empty(0)
        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.opacity_activation = torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize
        self.covariance_activation = build_covariance_from_scaling_rotation

    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        if self.active_sh_degree == 0:
            return self._features_dc
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    def get_covariance(self, scaling_modifier=1):
        return self.covariance_activation(self.get_scaling, scaling_modifier, self._rotation)

    def save_ply(self, path):
        """Binary PLY in the reference's field order (gaussian_model.py:281-302), raw pre-activation values."""
        from .ply_io import save_ply
        save_ply(path, self)

    def load_ply(self, path, device="cpu"):
        from .ply_io import load_ply
        return load_ply(path, self, device)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1
