"""TEST-ONLY numpy restatement of the GPU iso-surface point sampler (gaussian_gan_decoder_amd/csrc/ggd_surface.hip):
marching tetrahedra over the 6 tetrahedra around every cell's 0-7 diagonal, faces numbered cell by cell (cells in
[x][y][z] order), tetrahedron by tetrahedron; point i sits on face i mod F with weights from the same counter-based
generator.  The mesh family and the sampling rule follow main/decoder_utils/target_dataloader.py:96-118."""
import numpy as np

TET = np.array([[0, 1, 3, 7], [0, 1, 5, 7], [0, 2, 3, 7], [0, 2, 6, 7], [0, 4, 5, 7], [0, 4, 6, 7]])
CORNER = np.array([[(c >> 2) & 1, (c >> 1) & 1, c & 1] for c in range(8)])
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def corner_values(sigma, level):
    n = sigma.shape[0]
    m = n - 1
    f = np.empty((m, m, m, 8), np.float32)
    for c in range(8):
        dx, dy, dz = CORNER[c]
        f[..., c] = sigma[dx:dx + m, dy:dy + m, dz:dz + m] - np.float32(level)
    return f.reshape(-1, 8)


def cell_face_counts(sigma, level):
    f = corner_values(sigma, level)
    cnt = np.zeros(f.shape[0], np.int64)
    for t in range(6):
        inside = (f[:, TET[t]] > 0).sum(1)
        cnt += np.where((inside == 0) | (inside == 4), 0, np.where(inside == 2, 2, 1))
    return cnt


def _edge(fc, a, b):
    t = fc[a] / (fc[a] - fc[b])
    return CORNER[a].astype(np.float32) + np.float32(t) * (CORNER[b] - CORNER[a]).astype(np.float32)


def face_triangle(sigma, level, offsets, face):
    """Vertices (index coordinates, float32 [3,3]) of face number `face`; offsets = inclusive cumsum of the cell counts."""
    n = sigma.shape[0]; m = n - 1
    cell = int(np.searchsorted(offsets, face, side="right"))
    local = int(face - (offsets[cell - 1] if cell else 0))
    cz, cy, cx = cell % m, (cell // m) % m, cell // (m * m)
    fc = np.array([sigma[cx + CORNER[c][0], cy + CORNER[c][1], cz + CORNER[c][2]] for c in range(8)], np.float32) - np.float32(level)
    for t in range(6):
        ins = [c for c in TET[t] if fc[c] > 0]
        outs = [c for c in TET[t] if not fc[c] > 0]
        nt = 0 if len(ins) in (0, 4) else (2 if len(ins) == 2 else 1)
        if local >= nt:
            local -= nt
            continue
        if len(ins) in (1, 3):
            s = ins[0] if len(ins) == 1 else outs[0]
            o = outs if len(ins) == 1 else ins
            v = [_edge(fc, s, o[0]), _edge(fc, s, o[1]), _edge(fc, s, o[2])]
        else:
            p, q, r, s2 = ins[0], ins[1], outs[0], outs[1]
            v = [_edge(fc, p, r)] + ([_edge(fc, p, s2), _edge(fc, q, s2)] if local == 0 else [_edge(fc, q, s2), _edge(fc, q, r)])
        return np.stack(v) + np.array([cx, cy, cz], np.float32)
    raise AssertionError("face index beyond the cell's triangles")


def _mix64(z):
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
        return z ^ (z >> np.uint64(31))


def _u01(bits):
    return ((bits & np.uint64(0xFFFFFF)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def point_random_numbers(seed, idx):
    """(w0, w1, w2, thickness gaussian) of output points idx (array of ints), as the kernel draws them."""
    i = np.asarray(idx, np.uint64)
    with np.errstate(over="ignore"):
        h0 = _mix64(np.uint64(seed) ^ ((i * np.uint64(0xD1342543DE82EF95)) & M64))
    h1 = _mix64(h0)
    r0, r1, r2 = _u01(h0), _u01(h0 >> np.uint64(24)), _u01(h1)
    rs = (r0 + r1) + r2
    g1, g2 = _u01(h1 >> np.uint64(24)), _u01(_mix64(h1))
    gauss = np.sqrt(np.float32(-2.0) * np.log(g1)) * np.cos(np.float32(6.283185307179586) * g2)
    return r0 / rs, r1 / rs, r2 / rs, gauss.astype(np.float32)


def all_triangles(sigma, level):
    """Every triangle of the marching-tetrahedra surface, vectorised (the SET the kernel samples from, not its face numbering):
    float64 [F, 3, 3] vertices in index coordinates.  len == cell_face_counts(...).sum()."""
    n = sigma.shape[0]; m = n - 1
    f = corner_values(sigma, level).astype(np.float64)                      # [cells, 8]
    cell = np.arange(f.shape[0])
    origin = np.stack([cell // (m * m), (cell // m) % m, cell % m], 1).astype(np.float64)
    corner = CORNER.astype(np.float64)
    out = []

    def edge(fc, rows, a, b):                                              # a, b: per-row corner ids
        fa, fb = fc[rows, a], fc[rows, b]
        t = (fa / (fa - fb))[:, None]
        return corner[a] + t * (corner[b] - corner[a])
    for t in range(6):
        ids = TET[t]
        inside = f[:, ids] > 0
        k = inside.sum(1)
        for kk in (1, 2, 3):
            rows = np.nonzero(k == kk)[0]
            if rows.size == 0:
                continue
            ins = inside[rows]
            # stable order: the `kk` inside vertices first (kk = 1, 2) / the outside vertex first (kk = 3), each group in TET order
            key = ~ins if kk in (1, 2) else ins
            order = np.argsort(key, axis=1, kind="stable")
            v = ids[order]                                                 # [rows, 4] corner ids
            if kk in (1, 3):
                tri = np.stack([edge(f, rows, v[:, 0], v[:, j]) for j in (1, 2, 3)], 1)
                out.append(tri + origin[rows][:, None, :])
            else:
                p, q, r, s2 = v[:, 0], v[:, 1], v[:, 2], v[:, 3]
                e_pr, e_ps, e_qs, e_qr = edge(f, rows, p, r), edge(f, rows, p, s2), edge(f, rows, q, s2), edge(f, rows, q, r)
                out.append(np.stack([e_pr, e_ps, e_qs], 1) + origin[rows][:, None, :])
                out.append(np.stack([e_pr, e_qs, e_qr], 1) + origin[rows][:, None, :])
    return np.concatenate(out, 0) if out else np.zeros((0, 3, 3))


def triangle_areas(tri):
    return 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
