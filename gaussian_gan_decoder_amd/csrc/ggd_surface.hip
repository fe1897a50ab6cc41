// ggd_surface.hip -- GPU iso-surface point sampler: the target / position generator of the decoder training step
// (SURVEY.md section 8f row 4).
//
// Replaces, on the device and without any host round trip, what main/decoder_utils/target_dataloader.py:96-118 does on
// the CPU every step: 128^3 density grid -> skimage.measure.marching_cubes(level = 10) + trimesh (:172-176) -> one
// random point per face, pass after pass, until 500 000 points (:104-110, barycentric weights rand(3) / sum as there)
// -> positions scaled by clip(1 + surface_thickness * N(0,1), 0, 1) (:113-116); vertices in the reference's units
// (index / grid size - 0.5, :99-101).  The density grid is [x][y][z], z fastest (main/marching_cube/sample.py:15-17).
//
// The iso-surface is extracted by MARCHING TETRAHEDRA (every cell cut into the 6 tetrahedra around its 0-7 diagonal;
// 16 cases, at most 2 triangles per tetrahedron, vertices by linear interpolation along the cut edges): the same
// piecewise-linear surface family as marching cubes, crack-free across cells, with a case table small enough to live in
// registers -- the reference's exact triangulation (Lewiner tables inside skimage) is a third-party detail that does not
// enter the training objective.  Three launches: per-cell triangle counts -> inclusive scan (shared with the rasterizer)
// -> one lane per OUTPUT point: point i takes face i mod F of pass i div F (F stays on the device), finds its cell by
// bisection in the scanned counts, rebuilds the triangle and draws its weights from a counter-based generator keyed by
// (seed, i), so the result is a pure function of (grid, seed): reproducible, and restated in numpy by the tests.
#include "ggd_common.h"

namespace {

// the 6 tetrahedra of a cell: corners as bit triples (x = bit 2 ... z = bit 0 of the corner index -> offset (dx, dy, dz))
__device__ __constant__ const unsigned char kTet[6][4] = {{0, 1, 3, 7}, {0, 1, 5, 7}, {0, 2, 3, 7},
                                                          {0, 2, 6, 7}, {0, 4, 5, 7}, {0, 4, 6, 7}};

__device__ __forceinline__ void cell_corners(const float* __restrict__ sigma, int n, int cx, int cy, int cz, float level,
                                             float (&f)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int x = cx + ((c >> 2) & 1), y = cy + ((c >> 1) & 1), z = cz + (c & 1);
    f[c] = sigma[((size_t)x * n + y) * n + z] - level;
  }
}

__device__ __forceinline__ int tet_triangles(const float (&f)[8], int t) {
  int inside = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) inside += f[kTet[t][k]] > 0.0f ? 1 : 0;
  return inside == 0 || inside == 4 ? 0 : (inside == 2 ? 2 : 1);
}

__global__ __launch_bounds__(256) void surface_count_kernel(const float* __restrict__ sigma, int n, float level,
                                                            uint32_t* __restrict__ counts) {
  const int m = n - 1;
  const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (cell >= (int64_t)m * m * m) return;
  const int cz = (int)(cell % m), cy = (int)((cell / m) % m), cx = (int)(cell / ((int64_t)m * m));
  float f[8];
  cell_corners(sigma, n, cx, cy, cz, level, f);
  int cnt = 0;
#pragma unroll
  for (int t = 0; t < 6; ++t) cnt += tet_triangles(f, t);
  counts[cell] = (uint32_t)cnt;
}

// counter-based generator: splitmix64 of (seed, index, stream) -> 24-bit uniforms in (0, 1)
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ float u01(uint64_t bits24) { return ((float)(bits24 & 0xFFFFFFull) + 0.5f) * (1.0f / 16777216.0f); }

__device__ __forceinline__ void edge_point(const float (&f)[8], int a, int b, float (&p)[3]) {
  const float t = f[a] / (f[a] - f[b]);   // f changes sign along the edge: the denominator is not zero
  const float ax = (float)((a >> 2) & 1), ay = (float)((a >> 1) & 1), az = (float)(a & 1);
  const float bx = (float)((b >> 2) & 1), by = (float)((b >> 1) & 1), bz = (float)(b & 1);
  p[0] = ax + t * (bx - ax); p[1] = ay + t * (by - ay); p[2] = az + t * (bz - az);
}

__global__ __launch_bounds__(256) void surface_sample_kernel(const float* __restrict__ sigma, int n, float level,
                                                             const uint32_t* __restrict__ offsets /* inclusive */,
                                                             const uint32_t* __restrict__ n_faces_ptr, int num_points,
                                                             float thickness, uint64_t seed, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= num_points) return;
  const uint32_t F = *n_faces_ptr;
  if (F == 0u) { out[3 * i] = 0.0f; out[3 * i + 1] = 0.0f; out[3 * i + 2] = 0.0f; return; }
  const uint32_t face = (uint32_t)i % F;
  const int m = n - 1;
  const int64_t cells = (int64_t)m * m * m;
  // first cell whose inclusive count exceeds `face`
  int64_t lo = 0, hi = cells - 1;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (offsets[mid] > face) hi = mid; else lo = mid + 1;
  }
  const int64_t cell = lo;
  int local = (int)(face - (cell ? offsets[cell - 1] : 0u));
  const int cz = (int)(cell % m), cy = (int)((cell / m) % m), cx = (int)(cell / ((int64_t)m * m));
  float f[8];
  cell_corners(sigma, n, cx, cy, cz, level, f);
  float v[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int t = 0; t < 6; ++t) {
    const int nt = tet_triangles(f, t);
    if (local >= nt) { local -= nt; continue; }
    int in[4], out_[4], ni = 0, no = 0;
    for (int k = 0; k < 4; ++k) {
      const int c = kTet[t][k];
      if (f[c] > 0.0f) in[ni++] = c; else out_[no++] = c;
    }
    if (ni == 1 || ni == 3) {
      // the lone corner s and the three others: triangle on the edges (s, a), (s, b), (s, c)
      const int s = ni == 1 ? in[0] : out_[0];
      const int* o = ni == 1 ? out_ : in;
      edge_point(f, s, o[0], v[0]); edge_point(f, s, o[1], v[1]); edge_point(f, s, o[2], v[2]);
    } else {
      // 2 - 2: the quad (p r)(p s)(q s)(q r), p q inside, r s outside, as triangles (pr, ps, qs) and (pr, qs, qr)
      const int p = in[0], q = in[1], r = out_[0], s2 = out_[1];
      edge_point(f, p, r, v[0]);
      if (local == 0) { edge_point(f, p, s2, v[1]); edge_point(f, q, s2, v[2]); }
      else { edge_point(f, q, s2, v[1]); edge_point(f, q, r, v[2]); }
    }
    break;
  }
  // barycentric weights rand(3) / sum (target_dataloader.py:106-108), thickness factor clip(1 + thickness * N(0,1), 0, 1)
  const uint64_t h0 = mix64(seed ^ ((uint64_t)(uint32_t)i * 0xD1342543DE82EF95ull));
  const uint64_t h1 = mix64(h0);
  const float r0 = u01(h0), r1 = u01(h0 >> 24), r2 = u01(h1);
  const float rs = (r0 + r1) + r2;
  const float w0 = r0 / rs, w1 = r1 / rs, w2 = r2 / rs;
  const float g1 = u01(h1 >> 24), g2 = u01(mix64(h1));
  const float gauss = sqrtf(-2.0f * logf(g1)) * cosf(6.283185307179586f * g2);
  const float sc = fminf(1.0f, fmaxf(0.0f, 1.0f + thickness * gauss));
  const float base[3] = {(float)cx, (float)cy, (float)cz};
  const float inv_n = 1.0f / (float)n;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float idx = base[d] + ((w0 * v[0][d] + w1 * v[1][d]) + w2 * v[2][d]);
    out[3 * i + d] = (idx * inv_n - 0.5f) * sc;    // vertices /= sigmas.shape[0]; vertices -= 0.5  (:99-101)
  }
}

}  // namespace

extern "C" size_t ggd_surface_tmp_bytes(int32_t n) {
  if (n < 2) return 0;
  const size_t cells = (size_t)(n - 1) * (n - 1) * (n - 1);
  return 2 * ggd_align(cells * sizeof(uint32_t)) + ggd_align(ggd_scan_tmp_bytes((int64_t)cells)) + 256;
}

extern "C" int ggd_surface_sample(ggd_ctx* ctx, void* stream, const float* sigma, int32_t n, float level,
                                  int32_t num_points, float thickness, uint64_t seed, float* positions,
                                  uint32_t* num_faces, void* tmp, size_t tmp_bytes) {
  if (!ctx) return GGD_E_INVALID;
  if (n < 2 || n > 1024 || num_points < 0) return ggd_fail(ctx, GGD_E_INVALID, "ggd_surface_sample: bad grid size / point count");
  if (!sigma || !num_faces || !tmp || (num_points > 0 && !positions)) return ggd_fail(ctx, GGD_E_INVALID, "ggd_surface_sample: NULL pointer");
  if (tmp_bytes < ggd_surface_tmp_bytes(n)) return ggd_fail(ctx, GGD_E_INVALID, "ggd_surface_sample: tmp too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t cells = (int64_t)(n - 1) * (n - 1) * (n - 1);
  char* p = static_cast<char*>(tmp);
  uint32_t* counts = reinterpret_cast<uint32_t*>(p); p += ggd_align((size_t)cells * 4);
  uint32_t* offsets = reinterpret_cast<uint32_t*>(p); p += ggd_align((size_t)cells * 4);
  void* scan_tmp = p;
  hipLaunchKernelGGL(surface_count_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, sigma, n, level, counts);
  int rc = ggd_launch_inclusive_scan(ctx, s, counts, offsets, cells, num_faces, scan_tmp, ggd_scan_tmp_bytes(cells));
  if (rc != GGD_OK) return rc;
  if (num_points > 0)
    hipLaunchKernelGGL(surface_sample_kernel, dim3((unsigned)((num_points + 255) / 256)), dim3(256), 0, s, sigma, n, level,
                       offsets, num_faces, num_points, thickness, seed, positions);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}
