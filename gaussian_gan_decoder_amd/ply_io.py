"""Dependency-free binary-PLY writer / reader for Gaussian sets.

On-disk format == the reference's `GaussianModel.save_ply / load_ply`
(gaussian_splatting/scene/gaussian_model.py:266-302, 309-350; what splatviz and every 3DGS viewer consume,
README.md:70): one `vertex` element, little-endian float32 properties in this order
    x y z  nx ny nz  f_dc_0..2  [f_rest_0..3*((D+1)^2-1)-1]  opacity  scale_0..2  rot_0..3
holding the RAW (pre-activation) attributes; f_dc / f_rest are stored channel-major (features.transpose(1,2).flatten).
The reference uses the `plyfile` package; this module needs only numpy.
"""
from __future__ import annotations

import os

import numpy as np
import torch


def attribute_names(n_dc: int, n_rest: int, with_rest: bool):
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(n_dc)]
    if with_rest:
        names += [f"f_rest_{i}" for i in range(n_rest)]
    return names + ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]


def save_ply(path: str, pc) -> None:
    """`pc` duck-type: _xyz [P,3], _features_dc [P,1,3], _features_rest [P,M-1,3] (optional), _opacity [P,1],
    _scaling [P,3], _rotation [P,4], max_sh_degree."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    c = lambda t: t.detach().float().cpu()
    xyz = c(pc._xyz).numpy()
    P = xyz.shape[0]
    f_dc = c(pc._features_dc).transpose(1, 2).flatten(start_dim=1).contiguous().numpy()
    with_rest = getattr(pc, "max_sh_degree", 0) > 0
    cols = [xyz, np.zeros_like(xyz), f_dc]
    n_rest = 0
    if with_rest:
        f_rest = c(pc._features_rest).transpose(1, 2).flatten(start_dim=1).contiguous().numpy()
        n_rest = f_rest.shape[1]
        cols.append(f_rest)
    cols += [c(pc._opacity).reshape(P, 1).numpy(), c(pc._scaling).numpy(), c(pc._rotation).numpy()]
    data = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
    names = attribute_names(f_dc.shape[1], n_rest, with_rest)
    assert data.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as fh:
        fh.write(header.encode("ascii"))
        fh.write(data.tobytes())


def read_ply(path: str):
    """-> (names, float32 array [P, len(names)]) of the first element of a binary little-endian float PLY."""
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        count, names, fmt = None, [], None
        while True:
            line = fh.readline()
            if not line:
                raise ValueError("unterminated PLY header")
            tok = line.decode("ascii").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if count is not None:
                    break  # only the first element is read
                count = int(tok[2])
            elif tok[0] == "property":
                if tok[1] not in ("float", "float32"):
                    raise ValueError(f"unsupported property type {tok[1]}")
                names.append(tok[2])
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian":
            raise ValueError("only binary_little_endian PLY is supported")
        data = np.frombuffer(fh.read(count * len(names) * 4), dtype="<f4").reshape(count, len(names))
    return names, data


def load_ply(path: str, pc, device="cpu"):
    """Fill `pc` (a GaussianModel) from a PLY written by save_ply / by the reference; sets active_sh_degree."""
    names, data = read_ply(path)
    col = {n: i for i, n in enumerate(names)}
    pick = lambda prefix: sorted((n for n in names if n.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
    P = data.shape[0]
    pc._xyz = t(data[:, [col["x"], col["y"], col["z"]]])
    pc._opacity = t(data[:, [col["opacity"]]])
    dc = data[:, [col[n] for n in pick("f_dc_")]].reshape(P, 3, -1)
    pc._features_dc = t(dc).transpose(1, 2).contiguous()
    rest_names = pick("f_rest_")
    n_rest = len(rest_names)
    if n_rest:
        if n_rest != 3 * (pc.max_sh_degree + 1) ** 2 - 3:
            raise ValueError("f_rest count does not match max_sh_degree")
        rest = data[:, [col[n] for n in rest_names]].reshape(P, 3, n_rest // 3)
        pc._features_rest = t(rest).transpose(1, 2).contiguous()
    pc._scaling = t(data[:, [col[n] for n in pick("scale_")]])
    pc._rotation = t(data[:, [col[n] for n in pick("rot_")]])
    pc.active_sh_degree = pc.max_sh_degree if n_rest else 0
    return pc
