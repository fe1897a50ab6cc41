"""TEST-ONLY differentiable CPU renderer built on the oracle (oracle/ggd_oracle.py).  Lets the host-side logic
(render_simple dict contract, DecoderTrainer, gloo data-parallel step) run in the CPU-only container.  Never
imported by the product."""
import math

import numpy as np
import torch

from oracle import ggd_oracle as O


class _OracleRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, opacities, scales, rotations, cam, bg, W, H):
        n = lambda t: t.detach().cpu().numpy()
        f = O.forward(means3D=n(means3D), opacities=n(opacities), shs=n(shs), scales=n(scales), rotations=n(rotations),
                      viewmatrix=n(cam.world_view_transform), projmatrix=n(cam.full_proj_transform),
                      campos=n(cam.camera_center), bg=n(bg), W=W, H=H, tanfovx=math.tan(cam.FoVx * 0.5),
                      tanfovy=math.tan(cam.FoVy * 0.5))
        ctx.f = f
        radii = torch.from_numpy(f["radii"].copy())
        ctx.mark_non_differentiable(radii)
        return torch.from_numpy(f["color"].copy()), radii

    @staticmethod
    def backward(ctx, g, _):
        b = O.backward(ctx.f, g.contiguous().numpy())
        t = lambda a, shape=None: torch.from_numpy(np.ascontiguousarray(a)).reshape(shape) if shape else torch.from_numpy(np.ascontiguousarray(a))
        P = ctx.f["P"]
        return (t(b["dL_dmeans3D"]), t(b["dL_dmeans2D"]), t(b["dL_dsh"]), t(b["dL_dopacity"], (P, 1)),
                t(b["dL_dscales"]), t(b["dL_drots"]), None, None, None, None)


def render_simple_cpu(viewpoint_camera, pc, bg_color, xyz_offset=None, scaling_modifier=1.0, override_color=None,
                      debug=False):
    screenspace_points = torch.zeros_like(pc.get_xyz, requires_grad=True) + 0
    screenspace_points.retain_grad()
    means3D = pc.get_xyz if xyz_offset is None else pc.get_xyz + xyz_offset
    W, H = int(viewpoint_camera.image_width), int(viewpoint_camera.image_height)
    img, radii = _OracleRasterize.apply(means3D, screenspace_points, pc.get_features, pc.get_opacity, pc.get_scaling,
                                        pc.get_rotation, viewpoint_camera, bg_color, W, H)
    return {"render": img, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii,
            "alpha": radii, "depth": radii}
