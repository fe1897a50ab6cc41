"""GPU iso-surface point sampler: the `vertices` (init positions) of the decoder training step without leaving the device.

Mirrors the marching-cubes branch of TargetDataloader.get_data (main/decoder_utils/target_dataloader.py:96-118,172-176):
density grid [n, n, n] -> iso-surface at level 10 -> 500 000 random surface points, one per face and pass with
barycentric weights rand(3) / sum -> scaled by clip(1 + surface_thickness * N(0,1), 0, 1).  The reference does this with
skimage + trimesh on the CPU (a D2H copy of the grid and an H2D copy of the mesh every step); here it is three HIP launches
(csrc/ggd_surface.hip: marching tetrahedra, see there) and no host sync.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _capi


def sample_surface_points(sigmas: torch.Tensor, level: float = 10.0, num_points: int = 500_000,
                          surface_thickness: float = 0.1, seed: int = 0):
    """sigmas: CUDA float32 [n, n, n] ([x][y][z]).  Returns (positions [num_points, 3] float32, num_faces: CUDA uint32
    scalar tensor -- read it only if you can afford the sync).  No CPU fallback."""
    if not sigmas.is_cuda:
        raise RuntimeError("sample_surface_points needs a HIP device tensor (there is no CPU fallback)")
    if sigmas.dim() != 3 or sigmas.shape[0] != sigmas.shape[1] or sigmas.shape[1] != sigmas.shape[2]:
        raise ValueError("sigmas must be a cube [n, n, n]")
    if sigmas.dtype != torch.float32:
        raise TypeError("sigmas must be float32")
    dev = sigmas.device
    sig = sigmas.contiguous()
    n = int(sig.shape[0])
    cx = _capi.context_for(dev)
    nbytes = cx.lib.ggd_surface_tmp_bytes(n)
    tmp = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    pos = torch.empty((int(num_points), 3), dtype=torch.float32, device=dev)
    nf = torch.zeros((1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        cx.check(cx.lib.ggd_surface_sample(cx.handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream),
                                           C.c_void_p(sig.data_ptr()), n, float(level), int(num_points),
                                           float(surface_thickness), C.c_uint64(int(seed) & (2 ** 64 - 1)),
                                           C.c_void_p(pos.data_ptr()) if num_points else None, C.c_void_p(nf.data_ptr()),
                                           C.c_void_p(tmp.data_ptr()), nbytes))
    return pos, nf
