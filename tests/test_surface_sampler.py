"""GPU iso-surface point sampler (target_dataloader.py:96-118) against its numpy restatement and
against analytic surfaces.  The reference's sampler is skimage marching cubes + trimesh on the CPU (neither is in this
image), so there are no reference vectors: the restatement pins the kernel bit for bit (face numbering, triangle
vertices, random weights), analytic fields (sphere, torus) pin the geometry."""
import numpy as np
import pytest
import torch

import _surface_ref as SR


def _field(kind, n):
    ax = (np.arange(n, dtype=np.float32) / np.float32(n)) - np.float32(0.5)      # the reference's units: index / n - 0.5
    x, y, z = np.meshgrid(ax, ax, ax, indexing="ij")
    if kind == "sphere":
        d = np.float32(0.3) - np.sqrt(x * x + y * y + z * z)                     # > 0 inside, signed distance
    else:   # torus around the z axis, R = 0.28, r = 0.09
        d = np.float32(0.09) - np.sqrt((np.sqrt(x * x + y * y) - np.float32(0.28)) ** 2 + z * z)
    return (np.float32(10.0) + np.float32(400.0) * d).astype(np.float32)          # level 10 <=> d = 0


def _distance(kind, p):
    if kind == "sphere":
        return np.abs(np.linalg.norm(p, axis=1) - 0.3)
    return np.abs(np.sqrt((np.sqrt(p[:, 0] ** 2 + p[:, 1] ** 2) - 0.28) ** 2 + p[:, 2] ** 2) - 0.09)


@pytest.mark.parametrize("kind", ["sphere", "torus"])
def test_restatement_on_analytic_fields(kind):
    n = 48
    sig = _field(kind, n)
    cnt = SR.cell_face_counts(sig, 10.0)
    off = np.cumsum(cnt)
    F = int(off[-1])
    area = 4 * np.pi * 0.3 ** 2 if kind == "sphere" else 4 * np.pi ** 2 * 0.28 * 0.09
    assert 2 * area * n * n <= F <= 12 * area * n * n         # marching tetrahedra: ~9 triangles per voxel-face of area
    rng = np.random.default_rng(0)
    for face in rng.integers(0, F, 200):
        tri = SR.face_triangle(sig, 10.0, off, int(face)) / n - 0.5
        assert _distance(kind, tri).max() <= 0.5 / n            # vertices within half a voxel of the true surface


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n", [("sphere", 64), ("torus", 128)])
def test_gpu_sampler_matches_restatement_and_surface(native_lib, kind, n):
    from gaussian_gan_decoder_amd.target_sampler import sample_surface_points
    dev = torch.device("cuda:0")
    sig = _field(kind, n)
    N = 500_000
    pos, nf = sample_surface_points(torch.from_numpy(sig).to(dev), level=10.0, num_points=N, surface_thickness=0.0, seed=77)
    cnt = SR.cell_face_counts(sig, 10.0)
    off = np.cumsum(cnt)
    F = int(off[-1])
    assert int(nf.item()) == F and 0 < F < N
    p = pos.cpu().numpy()
    assert np.isfinite(p).all()
    # geometry: every point within half a voxel of the analytic surface
    assert _distance(kind, p).max() <= 0.5 / n
    # bit-level pin on a sample of output indices: same face, same triangle, same random weights
    rng = np.random.default_rng(1)
    idx = np.unique(np.concatenate([rng.integers(0, N, 300), [0, F - 1, F, N - 1]]))
    w0, w1, w2, _ = SR.point_random_numbers(77, idx)
    for k, i in enumerate(idx):
        tri = SR.face_triangle(sig, 10.0, off, int(i % F))
        want = ((w0[k] * tri[0] + w1[k] * tri[1]) + w2[k] * tri[2]) / np.float32(n) - np.float32(0.5)
        assert np.abs(p[i] - want).max() <= 2e-6, (i, p[i], want)
    # one point per face and pass: every face of the first pass is hit exactly once (weights differ between passes)
    first, second = p[:F], p[F:2 * F]
    assert np.abs(first - second).max() > 1e-4
    # thickness: positions scale by clip(1 + t * N(0,1), 0, 1) of an independent draw
    pos_t, _ = sample_surface_points(torch.from_numpy(sig).to(dev), level=10.0, num_points=N, surface_thickness=0.1, seed=77)
    ratio = (np.linalg.norm(pos_t.cpu().numpy(), axis=1) / np.maximum(np.linalg.norm(p, axis=1), 1e-9))
    g = SR.point_random_numbers(77, np.arange(N))[3]
    want_ratio = np.clip(1.0 + 0.1 * g, 0.0, 1.0)
    assert np.abs(ratio - want_ratio).max() <= 1e-4
    assert abs(float(g.mean())) < 0.01 and abs(float(g.std()) - 1.0) < 0.01     # the Gaussian draw is standard normal


@pytest.mark.gpu
def test_gpu_sampler_edge_cases(native_lib):
    from gaussian_gan_decoder_amd.target_sampler import sample_surface_points
    dev = torch.device("cuda:0")
    empty = torch.zeros(32, 32, 32, device=dev)                     # nothing above the level: no faces, zero-filled output
    pos, nf = sample_surface_points(empty, num_points=1000)
    assert int(nf.item()) == 0 and float(pos.abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        sample_surface_points(torch.zeros(8, 8, 8))
    with pytest.raises(ValueError):
        sample_surface_points(torch.zeros(8, 8, 4, device=dev))
