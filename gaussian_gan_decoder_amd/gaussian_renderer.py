"""render() / render_simple(): the callers of the rasterizer, with the reference's signatures and result dicts.

Mirrors gaussian_splatting/gaussian_renderer/__init__.py:19-102 (render, stock 3DGS) and :105-186 (render_simple,
what main/train_pano2gaussian_decoder.py:232, main/eval.py:38,84 and main/load_decoder.py:24 call).  Differences,
both deliberate: tensors are created on the device of `pc.get_xyz` instead of a hard-coded "cuda" (reference
:28,:114), and the rasterizer behind it is the gfx950 library instead of the CUDA submodule.

`viewpoint_camera` duck-type: FoVx FoVy image_height image_width world_view_transform full_proj_transform
camera_center.  `pc` duck-type: get_xyz get_opacity get_scaling get_rotation get_features active_sh_degree
max_sh_degree get_covariance().
"""
from __future__ import annotations

import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from .sh import eval_sh


def _screenspace_points(pc):
    # zero tensor whose .grad receives dL/d(screen-space mean) -- densification statistics upstream
    pts = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True) + 0
    try:
        pts.retain_grad()
    except Exception:
        pass
    return pts


def _settings(viewpoint_camera, pc, bg_color, scaling_modifier, debug, raw_attributes=False):
    return GaussianRasterizationSettings(
        raw_attributes=raw_attributes,
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=debug,
    )


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    """Stock 3DGS render (reference :19-102).  `pipe` needs .debug, .compute_cov3D_python, .convert_SHs_python."""
    screenspace_points = _screenspace_points(pc)
    rasterizer = GaussianRasterizer(_settings(viewpoint_camera, pc, bg_color, scaling_modifier, pipe.debug))
    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation
    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            feats = pc.get_features
            shs_view = feats.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.repeat(feats.shape[0], 1)
            dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color
    rendered_image, radii = rasterizer(means3D=pc.get_xyz, means2D=screenspace_points, shs=shs,
                                       colors_precomp=colors_precomp, opacities=pc.get_opacity, scales=scales,
                                       rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii}


def render_simple(viewpoint_camera, pc, bg_color: torch.Tensor, xyz_offset=None, scaling_modifier=1.0,
                  override_color=None, debug=False, fused_activations=False):
    """Decoder-path render (reference :105-186): scale/rotation always from the model, SH unless override_color.
    "alpha" and "depth" are the radii placeholders the reference returns (:184-185).
    fused_activations=True (extension): hand the RAW `_opacity/_scaling/_rotation` to the
    rasterizer, which applies sigmoid / exp / normalize (and their Jacobians in the backward) inside its kernels
    instead of three torch elementwise passes each way."""
    screenspace_points = _screenspace_points(pc)
    rasterizer = GaussianRasterizer(_settings(viewpoint_camera, pc, bg_color, scaling_modifier, debug,
                                              raw_attributes=fused_activations))
    means3D = pc.get_xyz
    if xyz_offset is not None:
        means3D = means3D + xyz_offset
    shs = colors_precomp = None
    if override_color is None:
        shs = pc.get_features
    else:
        colors_precomp = override_color
    if fused_activations:
        opacities, scales, rotations = pc._opacity, pc._scaling, pc._rotation
    else:
        opacities, scales, rotations = pc.get_opacity, pc.get_scaling, pc.get_rotation
    rendered_image, radii = rasterizer(means3D=means3D, means2D=screenspace_points, shs=shs,
                                       colors_precomp=colors_precomp, opacities=opacities,
                                       scales=scales, rotations=rotations, cov3D_precomp=None)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "alpha": radii, "depth": radii}
