"""Pure-PyTorch CPU rasterizer (forward).  TEST INFRASTRUCTURE + bench.py's `cpu_baseline` leg ONLY.

BASELINE.json's north_star asks for "a pure-PyTorch CPU raster timed on the box's own host cores (core count stated)"
next to every GPU number; this is it (SURVEY.md 8d "CPU baseline").  It is a SECOND, independent restatement of the
published 3DGS forward (call sites gaussian_splatting/gaussian_renderer/__init__.py:87-95,
167-175) -- whole-array torch ops instead of the C oracle's scalar loops -- and tests/test_torch_raster.py checks the
two against each other (integer stages exact, RGB <= 1e-5).  PARITY UNPINNED like the rest of oracle/: the
reference's native rasterizer is an empty, unpinned submodule (/root/reference/.gitmodules:4-6).
Only tests/ and bench.py may import this module; the product (gaussian_gan_decoder_amd) never does.

Stages: per-Gaussian preprocess vectorised over P -> cumsum -> duplicateWithKeys (repeat_interleave) -> stable sort
of the 64-bit (tile | depth bits) keys -> tile ranges (searchsorted) -> blend: tiles are grouped by list length and
blended `batch` tiles at a time as [tiles, list position, 256 pixels] arrays; the front-to-back recursion
T <- T (1 - alpha) with its two skip rules and the T < 1e-4 stop is an exclusive cumulative product along the list axis.
The cumulative product multiplies in the same order as the sequential loop, so transmittances match the C oracle to
the last bit or two; the colour sum uses torch's reduction order (<= 1e-6 from the sequential sum).
"""
from __future__ import annotations

import math

import torch

SH_C0 = 0.28209479177387814
TILE = 16


def _tile_rect(xy, radius, gx, gy):
    r = radius.to(torch.float32)
    f = lambda v, hi: torch.clamp((v / 16.0).to(torch.int32), 0, hi)      # C-style truncation, clamp to the grid
    x0, y0 = f(xy[:, 0] - r, gx), f(xy[:, 1] - r, gy)
    x1, y1 = f(xy[:, 0] + r + 15.0, gx), f(xy[:, 1] + r + 15.0, gy)
    return x0, y0, x1, y1


def preprocess(means3D, opacities, shs, scales, rotations, viewmatrix, projmatrix, W, H, tanfovx, tanfovy,
               scale_modifier=1.0):
    """SURVEY.md 9.2 at SH degree 0 (the decoder path: GaussianModel(0), main/train_pano2gaussian_decoder.py:215)."""
    P = means3D.shape[0]
    V, PV = viewmatrix.reshape(4, 4), projmatrix.reshape(4, 4)            # row-vector convention: [p, 1] @ V
    hom = torch.cat([means3D, torch.ones(P, 1)], 1)
    t = hom @ V
    h = hom @ PV
    tz = t[:, 2]
    in_front = tz > 0.2
    pw = 1.0 / (h[:, 3] + 0.0000001)
    ndc = h[:, :2] * pw[:, None]
    # Sigma = R S S R^T (gaussian_model.py:29-33, general_utils.py:78-110); quaternion (w, x, y, z) as given
    r, x, y, z = rotations.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(P, 3, 3)
    L = R * (scale_modifier * scales)[:, None, :]
    Sigma = L @ L.transpose(1, 2)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tzs = torch.where(in_front, tz, torch.ones_like(tz))
    tx = torch.clamp(t[:, 0] / tzs, -limx, limx) * tzs
    ty = torch.clamp(t[:, 1] / tzs, -limy, limy) * tzs
    J = torch.zeros(P, 2, 3)
    J[:, 0, 0] = fx / tzs; J[:, 0, 2] = -(fx * tx) / (tzs * tzs)
    J[:, 1, 1] = fy / tzs; J[:, 1, 2] = -(fy * ty) / (tzs * tzs)
    Wm = V[:3, :3].t()                                                    # W[r][c] = view[4c + r]
    T = J @ Wm
    cov2 = T @ Sigma @ T.transpose(1, 2)
    a, b, c = cov2[:, 0, 0] + 0.3, cov2[:, 0, 1], cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    ok = in_front & (det != 0)
    dinv = 1.0 / torch.where(det != 0, det, torch.ones_like(det))
    conic = torch.stack([c * dinv, -b * dinv, a * dinv], 1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).to(torch.int32)
    xy = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], 1)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    x0, y0, x1, y1 = _tile_rect(xy, radius, gx, gy)
    tiles = ((x1 - x0) * (y1 - y0)).to(torch.int64)
    ok = ok & (tiles > 0)
    tiles = torch.where(ok, tiles, torch.zeros_like(tiles))
    radius = torch.where(ok, radius, torch.zeros_like(radius))
    rgb = torch.clamp_min(SH_C0 * shs[:, 0, :] + 0.5, 0.0)
    return dict(depth=tz, radii=radius, xy=xy, conic=conic, opacity=opacities.reshape(-1), rgb=rgb,
                tiles_touched=tiles, rect=(x0, y0, x1, y1), gx=gx, gy=gy)


def bin_and_sort(g, W, H):
    """duplicateWithKeys + stable sort by (tile, depth bits) + tile ranges."""
    x0, y0, x1, y1 = g["rect"]
    gx, gy = g["gx"], g["gy"]
    tiles = g["tiles_touched"]
    offsets = torch.cumsum(tiles, 0)
    R = int(offsets[-1]) if tiles.numel() else 0
    ids = torch.repeat_interleave(torch.arange(tiles.numel()), tiles)                  # emission order: by Gaussian
    first = (offsets - tiles)[ids]
    k = torch.arange(R) - first                                                        # index inside the rect
    w = (x1 - x0).to(torch.int64)[ids]
    ty = y0.to(torch.int64)[ids] + k // torch.clamp(w, min=1)
    tx = x0.to(torch.int64)[ids] + k % torch.clamp(w, min=1)
    depth_bits = g["depth"].contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    keys = ((ty * gx + tx) << 32) | depth_bits[ids]
    keys_sorted, order = torch.sort(keys, stable=True)
    point_list = ids[order]
    tile_of = keys_sorted >> 32
    T = gx * gy
    bounds = torch.searchsorted(tile_of, torch.arange(T + 1))
    lo, hi = bounds[:-1], bounds[1:]
    ranges = torch.stack([torch.where(hi > lo, lo, torch.zeros_like(lo)), torch.where(hi > lo, hi, torch.zeros_like(hi))], 1)
    return dict(num_rendered=R, point_offsets=offsets, keys=keys_sorted, point_list=point_list, ranges=ranges)


def blend(g, b, bg, W, H, batch_elems=1 << 25, tile_subset=None):
    """SURVEY.md 9.4, all pixels of a batch of tiles at once.  tile_subset: optional LongTensor of tile ids -- only those
    tiles are blended (bench.py's bounded CPU sample); the other tiles keep the background."""
    gx, gy = g["gx"], g["gy"]
    T = gx * gy
    lens = (b["ranges"][:, 1] - b["ranges"][:, 0])
    color = torch.empty(3, gy * TILE, gx * TILE)
    final_T = torch.ones(gy * TILE, gx * TILE)
    n_contrib = torch.zeros(gy * TILE, gx * TILE, dtype=torch.int32)
    color[:] = bg.view(3, 1, 1)
    py, px = torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij")
    px, py = px.reshape(-1).float(), py.reshape(-1).float()                             # 256 pixels of a tile
    order = torch.argsort(lens, descending=True)
    order = order[lens[order] > 0]
    if tile_subset is not None:
        keep = torch.zeros(T, dtype=torch.bool); keep[tile_subset] = True
        order = order[keep[order]]
    rec = torch.cat([g["xy"], g["conic"], g["opacity"][:, None], g["rgb"]], 1)         # [P, 9]
    i = 0
    while i < order.numel():
        n = int(lens[order[i]])
        nb = max(1, min(order.numel() - i, batch_elems // (n * 256)))
        tl = order[i:i + nb]
        i += nb
        ln = lens[tl]
        pos = torch.arange(n)[None, :].expand(nb, n)
        valid = pos < ln[:, None]
        idx = torch.where(valid, b["ranges"][tl, 0][:, None] + pos, torch.zeros_like(pos))
        r = rec[b["point_list"][idx]]                                                    # [nb, n, 9]
        ox = ((tl % gx) * TILE).float()[:, None, None] + px[None, None, :]
        oy = ((tl // gx) * TILE).float()[:, None, None] + py[None, None, :]
        dx = r[:, :, 0:1] - ox
        dy = r[:, :, 1:2] - oy
        A, B, C = r[:, :, 2:3], r[:, :, 3:4], r[:, :, 4:5]
        power = torch.addcmul(((-0.5 * C) * dy) * dy, torch.addcmul((-B) * dy, -0.5 * A, dx), dx)
        alpha = torch.clamp_max(r[:, :, 5:6] * torch.exp(power), 0.99)
        alpha = torch.where((power > 0) | (alpha < 1.0 / 255.0) | ~valid[:, :, None], torch.zeros_like(alpha), alpha)
        Tinc = torch.cumprod(1.0 - alpha, dim=1)                                         # T after each record
        Texc = torch.cat([torch.ones(nb, 1, 256), Tinc[:, :-1]], 1)
        stop = (Tinc < 0.0001) & (alpha > 0)
        live = torch.cumsum(stop.to(torch.int32), 1) == 0                                # records before the stop
        wgt = torch.where(live, alpha * Texc, torch.zeros_like(alpha))
        col = torch.einsum("tnp,tnc->tcp", wgt, r[:, :, 6:9])
        contrib = live & (alpha > 0)
        last = torch.where(contrib, (pos + 1)[:, :, None].expand_as(contrib), torch.zeros(1, dtype=torch.int64)).amax(1)
        nlive = live.sum(1)                                                              # [nb, 256]
        Tfin = torch.where(nlive > 0, torch.gather(Tinc, 1, torch.clamp(nlive - 1, min=0)[:, None, :]).squeeze(1),
                           torch.ones(nb, 256))
        col = col + Tfin[:, None, :] * bg.view(1, 3, 1)
        for j in range(nb):       # scatter the tile back (cheap: nb tiles of 256 pixels)
            t = int(tl[j]); y0, x0 = (t // gx) * TILE, (t % gx) * TILE
            color[:, y0:y0 + TILE, x0:x0 + TILE] = col[j].view(3, TILE, TILE)
            final_T[y0:y0 + TILE, x0:x0 + TILE] = Tfin[j].view(TILE, TILE)
            n_contrib[y0:y0 + TILE, x0:x0 + TILE] = last[j].view(TILE, TILE).to(torch.int32)
    return color[:, :H, :W].contiguous(), final_T[:H, :W].contiguous(), n_contrib[:H, :W].contiguous()


@torch.no_grad()
def forward(*, means3D, opacities, shs, scales, rotations, viewmatrix, projmatrix, bg, W, H, tanfovx, tanfovy,
            scale_modifier=1.0):
    """One forward raster on the CPU with torch's intra-op thread pool.  Inputs: CPU float32 tensors."""
    g = preprocess(means3D.float(), opacities.float(), shs.float(), scales.float(), rotations.float(),
                   viewmatrix.float(), projmatrix.float(), W, H, float(tanfovx), float(tanfovy), scale_modifier)
    b = bin_and_sort(g, W, H)
    color, final_T, n_contrib = blend(g, b, bg.float(), W, H)
    out = dict(g); out.update(b)
    out.update(color=color, final_T=final_T, n_contrib=n_contrib)
    return out
