// ggd_tilebin.hip -- per-tile sorted lists WITHOUT sorting the instance list (GGD_OPT_BINNING = 1).
//
// The contract of stages a6-a8 is the RESULT: per tile, the Gaussians whose rect covers it, ordered
// by (fp32 depth bits, then emission order = Gaussian index), plus ranges.  (tile, depth bits, index) is a total
// order, so the same lists come out of
//   1. ONE stable sort of the P Gaussians by depth bits (32-bit keys, ggd_launch_sort32_iota) and
//   2. ONE stable binning pass that walks the Gaussians in that order and appends each to the lists of the tiles of
//      its rect,
// instead of materialising R = sum(tiles_touched) 12-byte (key, value) pairs and radix-sorting them 6 times
// (24*R bytes per pass).  R-proportional HBM traffic drops from ~150*R to the 4*R bytes of the final list.
//
// Binning = a counting sort with T (= #tiles) bins over instances that are generated on the fly:
//   count   : block b takes 1024 consecutive depth-ordered Gaussians, pools their instances in LDS (exclusive
//             offsets + rect, as the duplicate kernel does) and histograms the tile ids -> counts[b][t];
//   scan    : per tile, exclusive prefix over blocks (in place) + tile totals;
//   scatter : block b regenerates its instances in order; a wave ranks 64 consecutive instances with ballot
//             matching on the tile id (equal ids in one round come from different Gaussians in depth order, so
//             "number of lower peer lanes" is the stable rank); position = start[t] + prefix[b][t] + wave base + rank.
#include "ggd_common.h"

namespace {

constexpr int TB_G = 1024;  // depth-ordered Gaussians per block
constexpr int TB_THREADS = 256;
constexpr int TB_MAX_TILES = 8192;  // LDS budget of the scatter kernel: 16 KiB + 12 B per tile <= 160 KiB

struct TbPool {  // fixed part of the dynamic LDS: the block's pooled instance list
  uint32_t excl[TB_G + 1];
  uint32_t origin[TB_G];  // minx | miny << 16
  uint32_t width[TB_G];
  uint32_t id[TB_G];
  uint32_t scan[4];
  uint32_t mark[4][64];  // per-wave scratch of the slot -> owner expansion
};

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// Fill the pool for block `blk`; returns the number of instances of the block.
__device__ __forceinline__ uint32_t tb_load_pool(TbPool& sh, int blk, uint32_t n_vis, int gx, int gy,
                                                 const uint2* __restrict__ rect,
                                                 const uint32_t* __restrict__ tiles_touched,
                                                 const uint32_t* __restrict__ order) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t first = (uint32_t)blk * TB_G + 4u * tid;  // this thread's 4 consecutive ranks
  uint32_t nt[4], sum = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t rnk = first + q;
    uint32_t id = 0, n = 0, origin = 0, width = 1;
    if (rnk < n_vis) {
      id = order[rnk];
      n = tiles_touched[id];
      if (n > 0) {
        const uint2 rc = rect[id];   // {minx | maxx << 16, miny | maxy << 16}
        origin = (rc.x & 0xffffu) | ((rc.y & 0xffffu) << 16);
        width = (rc.x >> 16) - (rc.x & 0xffffu);
      }
    }
    nt[q] = n; sum += n;
    sh.origin[4 * tid + q] = origin; sh.width[4 * tid + q] = width; sh.id[4 * tid + q] = id;
  }
  const uint32_t inc = wave_incl_scan(sum);
  if (lane == 63) sh.scan[wv] = inc;
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) { const uint32_t s = sh.scan[w]; if (w < wv) base += s; total += s; }
  uint32_t run = base + inc - sum;
#pragma unroll
  for (int q = 0; q < 4; ++q) { sh.excl[4 * tid + q] = run; run += nt[q]; }
  if (tid == TB_THREADS - 1) sh.excl[TB_G] = total;
  __syncthreads();
  return total;
}

// Index of the Gaussian that owns pooled instance j: largest g with excl[g] <= j (binary search; once per wave).
__device__ __forceinline__ int tb_owner(const TbPool& sh, uint32_t j) {
  int lo = 0;
#pragma unroll
  for (int step = TB_G / 2; step >= 1; step >>= 1) {
    const int probe = lo + step;
    if (sh.excl[probe] <= j) lo = probe;
  }
  return lo;
}

// Walk the pooled instances [wbeg, wend) of one wave in rounds of 64 consecutive instances (lane = instance) and
// call f(ok, tile, id).  The slot -> Gaussian mapping of a round is expanded without a per-lane search: lane l loads
// the offset of Gaussian gcur+l, marks the slot where that Gaussian starts, and a wave max-scan over the marks gives
// every slot its owner (3 LDS round trips per round instead of a 10-deep dependent binary search).
template <typename F>
__device__ __forceinline__ void tb_for_rounds(TbPool& sh, uint32_t wbeg, uint32_t wend, int gx, F&& f) {
  if (wbeg >= wend) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t* mark = sh.mark[wv];
  int gcur = tb_owner(sh, wbeg);  // wave-uniform
  for (uint32_t j0 = wbeg; j0 < wend; j0 += 64) {
    const int g = min(gcur + lane, TB_G - 1);
    const uint32_t e0 = sh.excl[g], e1 = sh.excl[g + 1];
    mark[lane] = 0;
    // Gaussian gcur+lane covers slots [e0 - j0, e1 - j0) (clipped to [0, 64)); visible Gaussians have >= 1 tile,
    // so the 64 Gaussians from gcur on cover all 64 slots.
    const int s0 = (int)(e0 - j0), s1 = (int)(e1 - j0);
    if (gcur + lane < TB_G && e1 > e0 && s1 > 0 && s0 < 64) mark[max(s0, 0)] = (uint32_t)lane + 1u;
    __builtin_amdgcn_wave_barrier();
    uint32_t own = mark[lane];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)own, d, 64);
      if (lane >= d) own = max(own, o);
    }
    const uint32_t j = j0 + lane;
    const bool ok = j < wend;
    const int gi = gcur + (int)own - 1;
    uint32_t tile = 0, id = 0;
    if (ok) {
      const uint32_t k = j - sh.excl[gi];
      const uint32_t w = sh.width[gi];
      const uint32_t ry = k / w, rx = k - ry * w;
      const uint32_t org = sh.origin[gi];
      tile = ((org >> 16) + ry) * (uint32_t)gx + (org & 0xffffu) + rx;
      id = sh.id[gi];
    }
    f(ok, tile, id);
    // next round starts at the owner of slot 63, or at the one after it if that Gaussian ends exactly here
    const int g63 = __builtin_amdgcn_readlane(gi, 63);
    const uint32_t end63 = sh.excl[min(g63 + 1, TB_G)];
    gcur = g63 + ((end63 <= j0 + 64u) ? 1 : 0);
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(TB_THREADS) void tilebin_count_kernel(int W, int H, const uint2* __restrict__ rect,
                                                                   const uint32_t* __restrict__ tiles_touched,
                                                                   const uint32_t* __restrict__ order,
                                                                   const uint32_t* __restrict__ n_vis_ptr, int P,
                                                                   uint32_t* __restrict__ counts /*[nb][T]*/, int T) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TbPool& sh = *reinterpret_cast<TbPool*>(smem);
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem + sizeof(TbPool));
  const uint32_t n_vis = min((uint32_t)P, *n_vis_ptr);
  if ((uint32_t)blockIdx.x * TB_G >= n_vis) return;
  const int gx = (W + 15) / 16, gy = (H + 15) / 16;
  for (int t = threadIdx.x; t < T; t += TB_THREADS) hist[t] = 0;
  const uint32_t total = tb_load_pool(sh, blockIdx.x, n_vis, gx, gy, rect, tiles_touched, order);
  {
    const int wv = threadIdx.x >> 6;
    const uint32_t quarter = ((total + 3u) / 4u + 63u) & ~63u;
    const uint32_t wbeg = min(total, (uint32_t)wv * quarter), wend = min(total, wbeg + quarter);
    tb_for_rounds(sh, wbeg, wend, gx, [&](bool ok, uint32_t tile, uint32_t) { if (ok) atomicAdd(&hist[tile], 1u); });
  }
  __syncthreads();
  uint32_t* row = counts + (size_t)blockIdx.x * T;
  for (int t = threadIdx.x; t < T; t += TB_THREADS) row[t] = hist[t];
}

// Block = 64 tiles (lane = tile) x 16 waves over the block rows: exclusive prefix over blocks, in place; tile totals.
constexpr int TS_WAVES = 16;
__global__ __launch_bounds__(64 * TS_WAVES) void tilebin_scan_kernel(uint32_t* __restrict__ counts, int T,
                                                                     const uint32_t* __restrict__ n_vis_ptr, int P,
                                                                     uint32_t* __restrict__ totals) {
  __shared__ uint32_t part[TS_WAVES][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const uint32_t n_vis = min((uint32_t)P, *n_vis_ptr);
  const int nb = (int)((n_vis + TB_G - 1) / TB_G);
  const int per = (nb + TS_WAVES - 1) / TS_WAVES;
  const int r0 = min(nb, wv * per), r1 = min(nb, r0 + per);
  uint32_t s = 0;
  if (col < T) {
#pragma unroll 8
    for (int r = r0; r < r1; ++r) s += counts[(size_t)r * T + col];
  }
  part[wv][lane] = s;
  __syncthreads();
  uint32_t run = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < TS_WAVES; ++w) { const uint32_t p = part[w][lane]; if (w < wv) run += p; tot += p; }
  if (col < T) {
#pragma unroll 8
    for (int r = r0; r < r1; ++r) {
      const size_t idx = (size_t)r * T + col;
      const uint32_t c = counts[idx];
      counts[idx] = run;
      run += c;
    }
    if (wv == 0) totals[col] = tot;
  }
}

__global__ __launch_bounds__(TB_THREADS) void tilebin_scatter_kernel(
    int W, int H, const uint2* __restrict__ rect, const uint32_t* __restrict__ tiles_touched,
    const uint32_t* __restrict__ order, const uint32_t* __restrict__ n_vis_ptr, int P,
    const uint32_t* __restrict__ prefix /*[nb][T]*/, const uint32_t* __restrict__ totals, int T, int tbits,
    uint32_t* __restrict__ list, uint32_t* __restrict__ ranges, uint32_t capacity) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TbPool& sh = *reinterpret_cast<TbPool*>(smem);
  uint32_t* base = reinterpret_cast<uint32_t*>(smem + sizeof(TbPool));           // [T]
  // per-wave 16-bit counters [4][Ts]: the row stride is rounded to an even number of elements so that every row
  // starts on a 32-bit boundary (sweep 1 adds into the 32-bit word holding a counter; an odd T would misalign it)
  const int Ts = (T + 1) & ~1;
  uint16_t* cnt = reinterpret_cast<uint16_t*>(smem + sizeof(TbPool) + (size_t)T * 4);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t n_vis = min((uint32_t)P, *n_vis_ptr);
  const bool active = (uint32_t)blockIdx.x * TB_G < n_vis;
  if (!active && blockIdx.x != 0) return;
  const int gx = (W + 15) / 16, gy = (H + 15) / 16;

  // start[t] = exclusive scan of the tile totals (every block needs it; block 0 also writes the ranges)
  {
    uint32_t carry = 0;
    for (int t0 = 0; t0 < T; t0 += TB_THREADS) {
      const int t = t0 + tid;
      const uint32_t v = t < T ? totals[t] : 0u;
      const uint32_t inc = wave_incl_scan(v);
      if (lane == 63) sh.scan[wv] = inc;
      __syncthreads();
      uint32_t b = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) { const uint32_t s = sh.scan[w]; if (w < wv) b += s; tot += s; }
      const uint32_t start = carry + b + inc - v;
      if (t < T) {
        base[t] = start;
        if (blockIdx.x == 0) {
          ranges[2 * t] = v ? start : 0u;
          ranges[2 * t + 1] = v ? start + v : 0u;
        }
      }
      carry += tot;
      __syncthreads();
    }
  }
  if (!active) return;
  for (int t = tid; t < 4 * Ts; t += TB_THREADS) cnt[t] = 0;
  const uint32_t total = tb_load_pool(sh, blockIdx.x, n_vis, gx, gy, rect, tiles_touched, order);
  // each wave owns a contiguous quarter of the pooled list, rounded to whole rounds of 64
  const uint32_t quarter = ((total + 3u) / 4u + 63u) & ~63u;
  const uint32_t wbeg = min(total, (uint32_t)wv * quarter), wend = min(total, wbeg + quarter);
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  uint16_t* mycnt = cnt + (size_t)wv * Ts;

  // sweep 1: per-wave tile counts -- order does not matter for counting, so one LDS atomic per instance on the
  // 32-bit word that holds the tile's 16-bit counter (counts <= 1024 never carry into the neighbour)
  {
    uint32_t* mycnt32 = reinterpret_cast<uint32_t*>(mycnt);
    tb_for_rounds(sh, wbeg, wend, gx, [&](bool ok, uint32_t tile, uint32_t) {
      if (ok) atomicAdd(&mycnt32[tile >> 1], 1u << (16u * (tile & 1u)));
    });
  }
  __syncthreads();
  {
    const uint32_t* row = prefix + (size_t)blockIdx.x * T;
    for (int t = tid; t < T; t += TB_THREADS) {
      const uint32_t c0 = cnt[t], c1 = cnt[Ts + t], c2 = cnt[2 * Ts + t];
      base[t] += row[t];
      cnt[t] = 0; cnt[Ts + t] = (uint16_t)c0; cnt[2 * Ts + t] = (uint16_t)(c0 + c1);
      cnt[3 * Ts + t] = (uint16_t)(c0 + c1 + c2);
    }
  }
  __syncthreads();
  // sweep 2: stable rank + scatter
  tb_for_rounds(sh, wbeg, wend, gx, [&](bool ok, uint32_t tile, uint32_t id) {
    uint64_t peers = __ballot(ok);
    for (int b = 0; b < tbits; ++b) {
      const uint64_t m = __ballot((tile >> b) & 1u);
      peers &= ((tile >> b) & 1u) ? m : ~m;
    }
    const uint32_t below = (uint32_t)__popcll(peers & lt_mask);
    uint32_t before = 0;
    if (ok) before = mycnt[tile];
    __builtin_amdgcn_wave_barrier();
    if (ok) {
      const uint32_t dst = base[tile] + before + below;
      if (dst < capacity) list[dst] = id;  // capacity < R only in the speculative single-call forward (then retried)
      if (below == 0) mycnt[tile] = (uint16_t)(before + (uint32_t)__popcll(peers));
    }
  });
}

}  // namespace

bool ggd_tilebin_supported(int T) { return T > 0 && T <= TB_MAX_TILES; }

static inline int tb_blocks(int P) { return (P + TB_G - 1) / TB_G; }

size_t ggd_tilebin_tmp_bytes(int P, int T) {
  return ggd_align((size_t)tb_blocks(P) * T * sizeof(uint32_t)) + ggd_align((size_t)T * sizeof(uint32_t));
}

int ggd_launch_tilebin(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const uint2* rect,
                       const uint32_t* tiles_touched, const uint32_t* order, const uint32_t* n_vis_ptr,
                       uint32_t* list, uint32_t* ranges, uint32_t capacity, void* tmp, size_t tmp_bytes) {
  const int T = ((prm.width + 15) / 16) * ((prm.height + 15) / 16);
  if (!ggd_tilebin_supported(T)) return ggd_fail(ctx, GGD_E_INVALID, "tile grid too large for the binning path");
  if (tmp_bytes < ggd_tilebin_tmp_bytes(prm.P, T)) return ggd_fail(ctx, GGD_E_INVALID, "tilebin tmp too small");
  const int nb = tb_blocks(prm.P);
  uint32_t* counts = static_cast<uint32_t*>(tmp);
  uint32_t* totals = reinterpret_cast<uint32_t*>(static_cast<char*>(tmp) + ggd_align((size_t)nb * T * sizeof(uint32_t)));
  int tbits = 0;
  while ((1 << tbits) < T) ++tbits;
  const size_t lds_count = sizeof(TbPool) + (size_t)T * 4;
  const size_t lds_scatter = sizeof(TbPool) + (size_t)T * 4 + (size_t)((T + 1) & ~1) * 8;
  if (!(ctx->attr_mask & GGD_ATTR_TILEBIN)) {
    GGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tilebin_count_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    GGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tilebin_scatter_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    ctx->attr_mask |= GGD_ATTR_TILEBIN;
  }
  hipLaunchKernelGGL(tilebin_count_kernel, dim3(nb), dim3(TB_THREADS), lds_count, s, prm.width, prm.height, rect,
                     tiles_touched, order, n_vis_ptr, prm.P, counts, T);
  hipLaunchKernelGGL(tilebin_scan_kernel, dim3((T + 63) / 64), dim3(64 * TS_WAVES), 0, s, counts, T, n_vis_ptr, prm.P, totals);
  hipLaunchKernelGGL(tilebin_scatter_kernel, dim3(nb), dim3(TB_THREADS), lds_scatter, s, prm.width, prm.height, rect,
                     tiles_touched, order, n_vis_ptr, prm.P, counts, totals, T, tbits, list, ranges, capacity);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}
