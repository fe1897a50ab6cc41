"""CPU tests of the PLY writer/reader: header/field order of the reference (gaussian_model.py:266-302) and round trip."""
import numpy as np
import torch

from gaussian_gan_decoder_amd.gaussian_model import GaussianModel
from gaussian_gan_decoder_amd.ply_io import load_ply, read_ply, save_ply


def _model(P, deg):
    g = torch.Generator().manual_seed(P + deg)
    pc = GaussianModel(deg)
    pc._xyz = torch.randn(P, 3, generator=g); pc._scaling = torch.randn(P, 3, generator=g)
    pc._rotation = torch.randn(P, 4, generator=g); pc._opacity = torch.randn(P, 1, generator=g)
    pc._features_dc = torch.randn(P, 1, 3, generator=g)
    if deg > 0:
        pc._features_rest = torch.randn(P, (deg + 1) ** 2 - 1, 3, generator=g)
    return pc


def test_ply_header_matches_reference_layout(tmp_path):
    pc = _model(17, 0)
    path = str(tmp_path / "sub" / "pc.ply")
    save_ply(path, pc)
    raw = open(path, "rb").read()
    head = raw[:raw.index(b"end_header\n")].decode()
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 17\n")
    props = [l.split()[2] for l in head.splitlines() if l.startswith("property")]
    assert props == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1",
                     "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert all(l.split()[1] == "float" for l in head.splitlines() if l.startswith("property"))
    names, data = read_ply(path)
    assert data.shape == (17, 17)
    np.testing.assert_array_equal(data[:, 0:3], pc._xyz.numpy())
    np.testing.assert_array_equal(data[:, 3:6], 0)                     # normals are zeros
    np.testing.assert_array_equal(data[:, 9], pc._opacity.numpy()[:, 0])


def test_ply_round_trip_deg0_and_deg3(tmp_path):
    for deg in (0, 3):
        pc = _model(33, deg)
        path = str(tmp_path / f"d{deg}.ply")
        save_ply(path, pc)
        q = load_ply(path, GaussianModel(deg))
        for name in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc"):
            assert torch.equal(getattr(q, name), getattr(pc, name)), name
        if deg:
            assert torch.equal(q._features_rest, pc._features_rest)
            names, _ = read_ply(path)
            assert names.count("f_rest_44") == 1 and len([n for n in names if n.startswith("f_rest_")]) == 45
        assert q.active_sh_degree == deg
        # channel-major storage of SH like the reference: f_dc_k = features_dc[:, 0, k]
        names, data = read_ply(path)
        np.testing.assert_array_equal(data[:, names.index("f_dc_1")], pc._features_dc[:, 0, 1].numpy())


def test_ply_bytes_are_what_the_reference_writer_emits(tmp_path):
    """Byte-for-byte: the reference writes through plyfile (`PlyData([PlyElement.describe(elements, 'vertex')]).write`,
    gaussian_model.py:281-302) with every field 'f4': an ASCII header `ply / format binary_little_endian 1.0 /
    element vertex N / property float <name> ... / end_header` followed by N packed little-endian records in the order
    x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_* -- f_dc / f_rest channel-major (`transpose(1, 2).flatten`)."""
    import struct
    pc = _model(3, 1)
    path = str(tmp_path / "ref.ply")
    save_ply(path, pc)
    names = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(9)]
             + ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])
    header = "ply\nformat binary_little_endian 1.0\nelement vertex 3\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    body = b""
    for i in range(3):
        rec = (pc._xyz[i].tolist() + [0.0, 0.0, 0.0] + pc._features_dc[i].t().reshape(-1).tolist()
               + pc._features_rest[i].t().reshape(-1).tolist() + pc._opacity[i].tolist() + pc._scaling[i].tolist()
               + pc._rotation[i].tolist())
        body += struct.pack("<" + "f" * len(rec), *rec)
    assert open(path, "rb").read() == header.encode("ascii") + body


import pytest


@pytest.mark.gpu
def test_ply_round_trip_on_the_gpu_renders_the_same_image(native_lib, tmp_path):
    """GaussianModel on the device -> save_ply -> load_ply(device) -> render_simple: identical image and radii."""
    from gaussian_gan_decoder_amd.gaussian_renderer import render_simple
    from gaussian_gan_decoder_amd.synthetic import make_scene
    dev = torch.device("cuda:0")
    sc = make_scene(20000, 128, "shell", seed=4, log_scale_mean=-5.0).to(dev)
    pc = sc.gaussian_model()
    path = str(tmp_path / "gpu.ply")
    pc.save_ply(path)
    q = GaussianModel(0)
    q.load_ply(path, device=dev)
    for name in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc"):
        assert getattr(q, name).is_cuda and torch.equal(getattr(q, name), getattr(pc, name)), name
    a = render_simple(sc.cam, pc, bg_color=sc.bg)
    b = render_simple(sc.cam, q, bg_color=sc.bg)
    assert torch.equal(a["render"], b["render"]) and torch.equal(a["radii"], b["radii"])
