"""Generate tests/golden/*.json by IMPORTING the Python reference in this container (/root/reference is absent on
the GPU box; only the data written here travels).  Run:  python tests/golden/make_golden.py

What the reference can pin: it has no rasterizer source and no tests, but it does hold Python twins
of pieces of the contract, and those import here on the CPU:
  * gaussian_splatting/utils/graphics_utils.py:52-74   getProjectionMatrix
  * gaussian_splatting/scene/cameras.py:75-92          CustomCam (view / full-projection layout, camera_center)
  * main/camera_utils.py:63-93,125-146                 LookAtPoseSampler / create_cam2world_matrix
  * gaussian_splatting/utils/sh_utils.py:57-118        eval_sh (+ the clamp of gaussian_renderer/__init__.py:76-80)
  * gaussian_splatting/utils/general_utils.py:64-110   build_rotation / build_scaling_rotation / strip_symmetric
                                                       (+ gaussian_model.py:29-33: Sigma = L L^T)
The reference hard-codes device="cuda" / .cuda() in those helpers; they are neutralised below (torch.zeros device
kwarg dropped, Tensor.cuda = identity) so the very same code runs on the CPU.  An additional file holds regression
vectors produced by OUR oracle (labelled as such; it pins the oracle against accidental change, not against the
reference).
"""
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, "gaussian_splatting"))
sys.path.insert(0, os.path.join(REF, "main"))

# ---- neutralise the hard-coded CUDA placement of the reference helpers
_zeros = torch.zeros
def _zeros_cpu(*a, **k):
    k.pop("device", None)
    return _zeros(*a, **k)
torch.zeros = _zeros_cpu
torch.Tensor.cuda = lambda self, *a, **k: self

from utils.graphics_utils import getProjectionMatrix          # noqa: E402
from utils.sh_utils import eval_sh                            # noqa: E402
from utils.general_utils import build_scaling_rotation, strip_symmetric  # noqa: E402
# scene/__init__.py imports cv2 (absent here): load scene/cameras.py by path, bypassing the package __init__
import importlib.util                                         # noqa: E402
_spec = importlib.util.spec_from_file_location("ref_cameras", os.path.join(REF, "gaussian_splatting/scene/cameras.py"))
_cams = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_cams)
CustomCam = _cams.CustomCam
from camera_utils import LookAtPoseSampler                    # noqa: E402

L = lambda t: np.asarray(t, dtype=np.float64).tolist() if not isinstance(t, torch.Tensor) else t.double().tolist()


def cameras():
    out = []
    for (h, v, fov_deg, size) in [(math.pi / 2, math.pi / 2, 12.0, 512), (1.0, 1.2, 17.0, 512),
                                  (2.3, 0.9, 5.0, 1024), (math.pi / 2 + 0.4, math.pi / 2 - 0.2, 10.0, 64)]:
        c2w = LookAtPoseSampler.sample(h, v, horizontal_stddev=0, vertical_stddev=0, radius=2.7)[0]
        fov = fov_deg / 360 * 2 * np.pi  # main/train_pano2gaussian_decoder.py:230
        cam = CustomCam(size=size, fov=fov, extr=c2w)
        out.append(dict(h=h, v=v, fov_deg=fov_deg, size=size, cam2world=L(c2w), world_view_transform=L(cam.world_view_transform),
                        projection_matrix=L(cam.projection_matrix), full_proj_transform=L(cam.full_proj_transform),
                        camera_center=L(cam.camera_center)))
    proj = []
    for (zn, zf, fx, fy) in [(0.01, 10.0, 0.2, 0.2), (0.01, 100.0, 1.0, 0.7), (0.1, 50.0, 0.5, 1.3)]:
        proj.append(dict(znear=zn, zfar=zf, fovX=fx, fovY=fy, P=L(getProjectionMatrix(zn, zf, fx, fy))))
    return dict(cameras=out, projection=proj)


def sh():
    g = torch.Generator().manual_seed(11)
    N = 24
    shc = torch.randn(N, 16, 3, generator=g)                       # our layout [N, M, 3]
    p = torch.randn(N, 3, generator=g)
    campos = torch.randn(3, generator=g) * 0.1
    d = p - campos
    d = d / d.norm(dim=1, keepdim=True)
    res = {}
    for deg in range(4):
        M = (deg + 1) ** 2
        rgb = eval_sh(deg, shc[:, :M, :].transpose(1, 2), d)        # reference layout [N, 3, M]
        res[str(deg)] = L(torch.clamp_min(rgb + 0.5, 0.0))         # gaussian_renderer/__init__.py:76-80
        res[str(deg) + "_raw"] = L(rgb + 0.5)
    return dict(sh=L(shc), p=L(p), campos=L(campos), rgb=res)


def cov3d():
    g = torch.Generator().manual_seed(12)
    N = 32
    s = torch.exp(torch.randn(N, 3, generator=g) * 0.7 - 4.0)
    q = torch.nn.functional.normalize(torch.randn(N, 4, generator=g))
    out = []
    for mod in (1.0, 1.7):
        Lm = build_scaling_rotation(mod * s, q)                     # gaussian_model.py:29-33
        out.append(dict(mod=mod, cov6=L(strip_symmetric(Lm @ Lm.transpose(1, 2)))))
    return dict(scales=L(s), rotations=L(q), cases=out)


def decoder_fixture():
    """Config-1 plumbing (BASELINE.json configs[0]): the reference's 2-D tri-plane sampler
    (eg3d/training/volumetric_rendering/renderer.py:23-65) and per-point Decoder MLP
    (main/decoder_models/base_decoder.py:8-27) on seeded inputs."""
    sys.path.insert(0, os.path.join(REF, "eg3d"))
    from training.volumetric_rendering.renderer import sample_from_planes, generate_planes
    from decoder_models.base_decoder import Decoder
    g = torch.Generator().manual_seed(21)
    planes = torch.randn(3, 8, 16, 16, generator=g)
    pos = torch.rand(40, 3, generator=g) - 0.5
    pf = sample_from_planes(generate_planes(), planes.unsqueeze(0), pos.unsqueeze(0), padding_mode="zeros", box_warp=1)[0]
    torch.manual_seed(22)
    dec = Decoder(n_features=8 + 3, out_features=4, hidden_dim=16)
    out = dec(pf, pos)
    sd = {"sd_" + k: v.numpy() for k, v in dec.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "decoder_fixture.npz"), planes=planes.numpy(), positions=pos.numpy(),
                        plane_features=pf.numpy(), decoder_out=out.detach().numpy(), n_features=11, out_features=4,
                        hidden_dim=16, **sd)


def oracle_regression():
    from oracle import ggd_oracle as O
    from gaussian_gan_decoder_amd.synthetic import make_scene, make_dL_dpix
    torch.zeros = _zeros
    sc = make_scene(300, 48, "cube", seed=4, log_scale_mean=-4.0)
    cam = sc.cam
    f = O.forward(means3D=sc.xyz.numpy(), opacities=sc.opacities.numpy(), shs=sc.features_dc.numpy(),
                  scales=sc.scales.numpy(), rotations=sc.rotations.numpy(), viewmatrix=cam.world_view_transform.numpy(),
                  projmatrix=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(), bg=sc.bg.numpy(),
                  W=48, H=48, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5))
    b = O.backward(f, make_dL_dpix(48).numpy())
    np.savez_compressed(os.path.join(HERE, "oracle_regression.npz"), radii=f["radii"], tiles_touched=f["tiles_touched"],
                        keys=f["keys"], point_list=f["point_list"], ranges=f["ranges"], n_contrib=f["n_contrib"],
                        color=f["color"], dL_dmeans3D=b["dL_dmeans3D"], dL_dscales=b["dL_dscales"],
                        dL_drots=b["dL_drots"], dL_dopacity=b["dL_dopacity"], dL_dsh=b["dL_dsh"])


if __name__ == "__main__":
    for name, fn in (("cameras", cameras), ("sh", sh), ("cov3d", cov3d)):
        with open(os.path.join(HERE, name + ".json"), "w") as fh:
            json.dump(fn(), fh)
    decoder_fixture()
    oracle_regression()
    print("wrote", sorted(os.listdir(HERE)))
