// ggd_rowbin.hip -- two-level stable tile binning (GGD_OPT_BINNING = 3 / auto): the kernels below for tile grids up to 64 x 64,
// ggd_rowbin_wide.inc (same algorithm, bins spread over lane groups) for grids up to 255 x 255.
//
// Contract (stages a6-a8 as a RESULT: per tile, the Gaussians whose rect covers it in
// (depth bits, index) order, plus ranges), from the depth-ordered Gaussians.  The single-level pass there pays O(T) LDS
// work per 1024 Gaussians and writes one lone 4-byte store per (block, tile).  A tile rect is a product of two
// intervals, so the expansion is split into two 1-D counting sorts with <= 64 bins each -- one bin per LANE:
//   level 1 (rows):    Gaussian [y0, y1)  -> one entry {id, x0, x1} in the list of every tile ROW it touches
//   level 2 (columns): row entry [x0, x1) -> the Gaussian id in the list of every TILE of that row it touches
// Both levels are stable (items are consumed in order, chunk by chunk), so depth order survives and the final lists
// equal the radix-sort path's bit for bit.  Per level: count (LDS difference array: +1 at lo, -1 at hi, prefix over
// lanes), scan (one workgroup: exclusive prefix over chunks per bin, bin starts), scatter.  The scatter is
// item-serial / bin-parallel: a wave keeps one running destination per lane (= bin); for each item the lanes in
// [lo, hi) store and advance.  No per-tile LDS tables, no O(T) work per block, entries of one (chunk, bin) are
// consecutive in memory.
#include "ggd_common.h"

namespace {

#include "ggd_scan.inc"

constexpr int RB_THREADS = 256;
constexpr int RB_WAVES = RB_THREADS / 64;
constexpr int RB_STAGE = 4096;                     // instances a level-2 workgroup can stage in LDS (16 KB; measured: 4096 / 6144 /
                                                   // 8192 / 12288 -> 3656 / 3640 / 3594 / 3589 frames/s: occupancy beats coverage)
constexpr int RB_STAGE1 = 4096;                    // row entries a level-1 workgroup can stage (32 KB)
constexpr int RB_IPL = 4;                          // items per lane (2 and 8 measured slower)
constexpr int RB_CHUNK = RB_THREADS * RB_IPL;      // 1024 items per workgroup
constexpr int RB_WCHUNK = 64 * RB_IPL;             // 256 items per wave

// tables (uint32): [0..64] row starts (65 entries, [64] = total entries), [65..129] first level-2 block of each row
// (+ total), [130 .. 130 + 64*64) start of every (row, column) tile list
constexpr int RB_TAB_ROWSTART = 0, RB_TAB_ROWBLK = 65, RB_TAB_TILESTART = 130, RB_TAB_ROWINST = 130 + 64 * 64,
              RB_TAB_FLAG = RB_TAB_ROWINST + 64, RB_TAB_WORDS = RB_TAB_FLAG + 64;
// [ROWINST + r] instances of row r, [FLAG + r] != 0 once it is published (level-2 scan, one workgroup per row)

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// ---- level 1 count: rect of every depth-ordered Gaussian (kept, packed, for the scatter) + per-row counts of the chunk
__global__ __launch_bounds__(RB_THREADS) void rb_count1_kernel(int W, int H, const uint2* __restrict__ rect,
                                                               const uint32_t* __restrict__ order,
                                                               const uint32_t* __restrict__ n_vis_ptr, int P,
                                                               uint2* __restrict__ packed, uint32_t* __restrict__ counts1,
                                                               const uint32_t* __restrict__ order_alt,
                                                               const uint32_t* __restrict__ use_alt) {
  __shared__ int diff[65];
  if (use_alt && *use_alt != 0u) order = order_alt;   // the depth sort's last pass was the identity and copied nothing
  const uint32_t n_vis = min((uint32_t)P, *n_vis_ptr);
  const uint32_t base = (uint32_t)blockIdx.x * RB_CHUNK;
  if (base >= n_vis) return;
  const int tid = threadIdx.x;
  if (tid < 65) diff[tid] = 0;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < RB_IPL; ++q) {
    const uint32_t rnk = base + (uint32_t)q * RB_THREADS + tid;
    if (rnk < n_vis) {
      const uint32_t id = order[rnk];
      const uint2 rc = rect[id];   // {minx | maxx << 16, miny | maxy << 16} (grids up to 64 x 64 here)
      int x0 = (int)(rc.x & 0xffffu), x1 = (int)(rc.x >> 16), y0 = (int)(rc.y & 0xffffu), y1 = (int)(rc.y >> 16);
      const int n = (x1 - x0) * (y1 - y0);
      if (n <= 0) { x0 = x1 = y0 = y1 = 0; }
      packed[rnk] = make_uint2(id, (uint32_t)x0 | ((uint32_t)x1 << 8) | ((uint32_t)y0 << 16) | ((uint32_t)y1 << 24));
      if (n > 0) { atomicAdd(&diff[y0], 1); atomicAdd(&diff[y1], -1); }
    }
  }
  __syncthreads();
  if (tid < 64) counts1[(size_t)blockIdx.x * 64 + tid] = (uint32_t)wave_incl_scan_i32(diff[tid]);
}

// ---- scan over chunks, one workgroup, lane = bin: counts[c][bin] -> exclusive prefix within the bin; totals per bin
//      level 1: rows of one segment (all chunks);   level 2 is handled per row in rb_scan2_kernel.
__global__ __launch_bounds__(1024) void rb_scan1_kernel(uint32_t* __restrict__ counts1,
                                                        const uint32_t* __restrict__ n_vis_ptr, int P,
                                                        uint32_t* __restrict__ tab, uint32_t ent_cap) {
  __shared__ uint32_t part[16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t n_vis = min((uint32_t)P, *n_vis_ptr);
  const int nb = (int)((n_vis + RB_CHUNK - 1) / RB_CHUNK);
  const int per = (nb + 15) / 16;
  const int r0 = min(nb, wv * per), r1 = min(nb, r0 + per);
  uint32_t s = 0;
#pragma unroll 8
  for (int r = r0; r < r1; ++r) s += counts1[(size_t)r * 64 + lane];
  part[wv][lane] = s;
  __syncthreads();
  uint32_t run = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) { const uint32_t p = part[w][lane]; if (w < wv) run += p; tot += p; }
#pragma unroll 8
  for (int r = r0; r < r1; ++r) {
    const size_t idx = (size_t)r * 64 + lane;
    const uint32_t c = counts1[idx];
    counts1[idx] = run;
    run += c;
  }
  if (wv == 0) {
    const uint32_t inc = wave_incl_scan_u32(tot);
    tab[RB_TAB_ROWSTART + lane] = inc - tot;
    if (lane == 63) tab[RB_TAB_ROWSTART + 64] = inc;
    // level-2 blocks cover only the entries that exist in `ent` (ent_cap of them): when a speculative forward
    // under-estimated the capacity the rows are clamped, so that counts2 (sized for ent_cap) is never overrun
    const uint32_t rs = inc - tot;
    const uint32_t have = rs < ent_cap ? min(tot, ent_cap - rs) : 0u;
    const uint32_t nblk = (have + RB_CHUNK - 1) / RB_CHUNK;
    const uint32_t binc = wave_incl_scan_u32(nblk);
    tab[RB_TAB_ROWBLK + lane] = binc - nblk;
    if (lane == 63) tab[RB_TAB_ROWBLK + 64] = binc;
    tab[RB_TAB_FLAG + lane] = 0u;   // consumed by rb_scan2_kernel (next launch on the stream)
  }
}

// 64 x 64 bit-matrix transpose across the wave: lane i passes row i (bit b = M[i][b]) and gets column `lane`
// (bit i = M[i][lane]).  Six butterfly stages (block sizes 32 .. 1): in every lane pair (l, l ^ d) the off-diagonal
// d x d blocks are exchanged; one cross-lane move per 32-bit word and stage.
template <int D, uint32_t M>
__device__ __forceinline__ void tr_stage(uint32_t& lo, uint32_t& hi, uint32_t lane) {
  const bool up = (lane & (uint32_t)D) != 0;
  const uint32_t keep = up ? ~M : M;
  const uint32_t slo = up ? ((lo & M) << D) : ((lo & ~M) >> D);
  const uint32_t shi = up ? ((hi & M) << D) : ((hi & ~M) >> D);
  lo = (lo & keep) | (uint32_t)__shfl_xor((int)slo, D, 64);
  hi = (hi & keep) | (uint32_t)__shfl_xor((int)shi, D, 64);
}

__device__ __forceinline__ uint64_t wave_transpose64(uint64_t row) {
  const uint32_t lane = threadIdx.x & 63;
  uint32_t lo = (uint32_t)row, hi = (uint32_t)(row >> 32);
  {
    const bool up = (lane & 32u) != 0;
    const uint32_t recv = (uint32_t)__shfl_xor((int)(up ? lo : hi), 32, 64);
    if (up) lo = recv; else hi = recv;
  }
  // stages 16 .. 1 act on the two 32-bit words independently (compile-time masks and lane distances)
  tr_stage<16, 0x0000FFFFu>(lo, hi, lane);
  tr_stage<8, 0x00FF00FFu>(lo, hi, lane);
  tr_stage<4, 0x0F0F0F0Fu>(lo, hi, lane);
  tr_stage<2, 0x33333333u>(lo, hi, lane);
  tr_stage<1, 0x55555555u>(lo, hi, lane);
  return (uint64_t)lo | ((uint64_t)hi << 32);
}

// A wave's items (RB_IPL rounds of 64, lane = item, interval [lo, hi) of bins) against its 64 bins (lane = bin), round 5 form:
//   wave_cols   every round's 64 x 64 item / bin bit matrix, transposed: cols[q] of lane b = the items of round q that cover bin
//               b, in item order.  Returns the lane's (= bin's) number of entries: popcount of its masks -- the counts cost
//               nothing beyond the transposes the emission needs anyway (they used to be an LDS difference array filled with
//               two atomics per item, 25 of rb_scatter2_kernel's 59 us on the head-like scene: same-address LDS atomics
//               serialise).
//   wave_emit   ITEM-parallel: the lane keeps its item and walks the bins b = lo .. hi - 1 it covers; for each it reads
//               {mask of bin b, running destination of bin b} (one 16-byte LDS read from the wave's slab, written by the bin
//               lanes) and stores its payload at destination + (items below it in the mask).  Steps per round = the widest
//               item (a handful), whatever the data; the bin-parallel drain it replaces took as many steps as the fullest bin
//               holds items -- up to 64 where a depth slice of the scene projects onto a few rows (the caps of a shell).
__device__ __forceinline__ uint32_t wave_cols(const uint32_t (&iv)[RB_IPL], int n_items, uint64_t (&cols)[RB_IPL]) {
  uint32_t cnt = 0;
#pragma unroll
  for (int q = 0; q < RB_IPL; ++q) {
    cols[q] = 0ull;
    if (q * 64 >= n_items) continue;   // (no `break`: keeps the loop fully unrollable)
    const uint32_t l = iv[q] & 0xffu, h = iv[q] >> 8;
    const uint64_t below_h = h >= 64u ? ~0ull : ((1ull << h) - 1ull);
    const uint64_t row = h > l ? (below_h & ~((1ull << l) - 1ull)) : 0ull;
    cols[q] = wave_transpose64(row);
    cnt += (uint32_t)__popcll(cols[q]);
  }
  return cnt;
}

// Which walk is shorter for this round?  Item-parallel takes as many steps as the widest item has bins, the bin-parallel drain
// as many as the fullest bin has items; both maxima are bracketed with three ballots each (<= 4, 8, 16, more) -- small splats on
// a 64 x 64-tile grid favour the items (and decisively so where a depth slice piles onto a few rows), the wide items of a 4K
// frame the drain (measured with the items only: 4K binning 515 -> 610 us; 1080p 125 -> 117; the 1 M shell 125 -> 88).
__device__ __forceinline__ bool rb_walk_items(uint32_t width, uint32_t fill) {
  const int wi = __ballot(width > 16u) ? 3 : (__ballot(width > 8u) ? 2 : (__ballot(width > 4u) ? 1 : 0));
  const int wf = __ballot(fill > 16u) ? 3 : (__ballot(fill > 8u) ? 2 : (__ballot(fill > 4u) ? 1 : 0));
  return wi <= wf + 1;   // (one LDS read per item step against two cross-lane moves per drain step)
}

template <bool TWO, typename Emit>
__device__ __forceinline__ void wave_emit(const uint32_t (&iv)[RB_IPL], const uint32_t (&pa)[RB_IPL],
                                          const uint32_t (&pb)[RB_IPL], const uint64_t (&cols)[RB_IPL], int n_items,
                                          uint32_t& dst, uint4* slab /* [64], this wave's */, Emit&& emit) {
  const int lane = threadIdx.x & 63;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int q = 0; q < RB_IPL; ++q) {
    if (q * 64 >= n_items) continue;
    __builtin_amdgcn_wave_barrier();                      // the previous round's reads of the slab are done (LDS is in order per wave)
    slab[lane] = make_uint4((uint32_t)cols[q], (uint32_t)(cols[q] >> 32), dst, 0u);
    __builtin_amdgcn_wave_barrier();
    const uint32_t l = iv[q] & 0xffu, h = iv[q] >> 8;
    if (rb_walk_items(h > l ? h - l : 0u, (uint32_t)__popcll(cols[q]))) {
      for (uint32_t b = l; __ballot(b < h) != 0ull; ++b) {
        const bool act = b < h;
        const uint4 e = slab[act ? b : 0u];
        const uint64_t m = (uint64_t)e.x | ((uint64_t)e.y << 32);
        if (act) emit(e.z + (uint32_t)__popcll(m & lt_mask), pa[q], TWO ? pb[q] : 0u);
      }
      dst += (uint32_t)__popcll(cols[q]);
    } else {   // bin-parallel drain: the lane (= bin) takes its items lowest first, the payload comes over with ds_bpermute
      uint64_t col = cols[q];
      while (__ballot(col != 0ull) != 0ull) {
        const bool act = col != 0ull;
        const int k = act ? (__ffsll((unsigned long long)col) - 1) : 0;
        col &= col - 1ull;
        const uint32_t a = (uint32_t)__shfl((int)pa[q], k, 64);
        const uint32_t bb = TWO ? (uint32_t)__shfl((int)pb[q], k, 64) : 0u;
        if (act) { emit(dst, a, bb); dst += 1; }
      }
    }
  }
}

// ---- level 1 scatter: {id, x0 | x1 << 8} into the list of every row in [y0, y1)
__global__ __launch_bounds__(RB_THREADS) void rb_scatter1_kernel(const uint2* __restrict__ packed,
                                                                 const uint32_t* __restrict__ n_vis_ptr, int P,
                                                                 const uint32_t* __restrict__ prefix1,
                                                                 const uint32_t* __restrict__ tab,
                                                                 uint2* __restrict__ ent, uint32_t ent_cap) {
  __shared__ uint4 slab[RB_WAVES][64];
  __shared__ uint32_t wcnt[RB_WAVES][64];
  const uint32_t n_vis = min((uint32_t)P, *n_vis_ptr);
  const uint32_t base = (uint32_t)blockIdx.x * RB_CHUNK;
  if (base >= n_vis) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t wbeg = base + (uint32_t)wv * RB_WCHUNK;
  const int n_items = (int)min((uint32_t)RB_WCHUNK, n_vis > wbeg ? n_vis - wbeg : 0u);
  uint32_t iv[RB_IPL], pa[RB_IPL], pb[RB_IPL];
#pragma unroll
  for (int q = 0; q < RB_IPL; ++q) {
    const int i = q * 64 + lane;
    uint2 it = make_uint2(0, 0);
    if (i < n_items) it = packed[wbeg + i];
    pa[q] = it.x; pb[q] = it.y & 0xffffu;
    iv[q] = it.y >> 16;   // y0 | y1 << 8
  }
  uint64_t cols[RB_IPL];
  const uint32_t mine = wave_cols(iv, n_items, cols);
  wcnt[wv][lane] = mine;
  __syncthreads();
  uint32_t dst = tab[RB_TAB_ROWSTART + lane] + prefix1[(size_t)blockIdx.x * 64 + lane];
#pragma unroll
  for (int w = 0; w < RB_WAVES; ++w) if (w < wv) dst += wcnt[w][lane];
  wave_emit<true>(iv, pa, pb, cols, n_items, dst, slab[wv], [&](uint32_t d, uint32_t id, uint32_t xx) {
    if (d < ent_cap) ent[d] = make_uint2(id, xx);
  });
}

// ---- level 1 in ONE launch (folded front end: the entries per row were counted by the preprocess kernel, GGD_FOLD_ROWTOT):
//      count, decoupled look-back over the earlier chunks (lane = row; status words [chunk][64] then [group][64], zero at
//      launch, bit 30 = published -- the scheme of the depth sort's passes), scatter.  Replaces rb_count1 / rb_scan1 /
//      rb_scatter1 and the `packed` round trip between them.  Every workgroup derives the row starts from the 64 x REPS row
//      totals itself; workgroup 0 also writes the tables level 2 reads.  Workgroups wait only for lower-numbered ones, which
//      the dispatcher started earlier.
constexpr uint32_t RB_PUB = 1u << 30;
constexpr int RB_LB_BATCH = 16;
__device__ __forceinline__ uint32_t rb_ld(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rb_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// sum of `cnt` published words p[0], p[64], ...: RB_LB_BATCH requests in flight, unpublished ones are polled
__device__ __forceinline__ uint32_t rb_sum_words(uint32_t* p, int cnt) {
  uint32_t acc = 0;
  for (int b0 = 0; b0 < cnt; b0 += RB_LB_BATCH) {
    uint32_t v[RB_LB_BATCH];
#pragma unroll
    for (int i = 0; i < RB_LB_BATCH; ++i) v[i] = (b0 + i < cnt) ? rb_ld(p + (size_t)(b0 + i) * 64) : RB_PUB;
#pragma unroll
    for (int i = 0; i < RB_LB_BATCH; ++i) {
      uint32_t x = v[i];
      if ((x >> 30) == 0u) {
        uint32_t* q = p + (size_t)(b0 + i) * 64;
        do { __builtin_amdgcn_s_sleep(1); x = rb_ld(q); } while ((x >> 30) == 0u);
      }
      acc += x & (RB_PUB - 1u);
    }
  }
  return acc;
}

__global__ __launch_bounds__(RB_THREADS) void rb_level1_kernel(const uint2* __restrict__ rect, const uint32_t* __restrict__ order,
                                                               const uint32_t* __restrict__ n_vis_ptr, int P,
                                                               const uint32_t* __restrict__ rowtot, uint32_t* status,
                                                               int chunks_max, int gshift_max, uint32_t* __restrict__ tab,
                                                               uint2* __restrict__ ent, uint32_t ent_cap,
                                                               const uint32_t* __restrict__ order_alt,
                                                               const uint32_t* __restrict__ use_alt) {
  __shared__ uint4 slab[RB_WAVES][64];
  __shared__ uint32_t wcnt[RB_WAVES][64];
  __shared__ uint32_t s_base[64];
  __shared__ uint2 stage[RB_STAGE1];
  if (use_alt && *use_alt != 0u) order = order_alt;   // the depth sort's last pass was the identity and copied nothing
  const uint32_t n_vis = min((uint32_t)P, *n_vis_ptr);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint32_t rowstart = 0;
  if (wv == 0) {
    uint32_t tot = 0;
#pragma unroll
    for (int r = 0; r < GGD_FOLD_REPS; ++r) tot += rowtot[r * 64 + lane];
    const uint32_t inc = wave_incl_scan_u32(tot);
    rowstart = inc - tot;
    if (blockIdx.x == 0) {   // the tables of level 2 (as rb_scan1_kernel writes them)
      tab[RB_TAB_ROWSTART + lane] = rowstart;
      if (lane == 63) tab[RB_TAB_ROWSTART + 64] = inc;
      const uint32_t have = rowstart < ent_cap ? min(tot, ent_cap - rowstart) : 0u;
      const uint32_t nblk = (have + RB_CHUNK - 1) / RB_CHUNK;
      const uint32_t binc = wave_incl_scan_u32(nblk);
      tab[RB_TAB_ROWBLK + lane] = binc - nblk;
      if (lane == 63) tab[RB_TAB_ROWBLK + 64] = binc;
      tab[RB_TAB_FLAG + lane] = 0u;   // consumed by rb_scan2_kernel
    }
  }
  const uint32_t base = (uint32_t)blockIdx.x * RB_CHUNK;
  if (base >= n_vis) return;
  const uint32_t wbeg = base + (uint32_t)wv * RB_WCHUNK;
  const int n_items = (int)min((uint32_t)RB_WCHUNK, n_vis > wbeg ? n_vis - wbeg : 0u);
  uint32_t iv[RB_IPL], pa[RB_IPL], pb[RB_IPL];
#pragma unroll
  for (int q = 0; q < RB_IPL; ++q) {
    const int i = q * 64 + lane;
    pa[q] = 0; pb[q] = 0; iv[q] = 0;
    if (i < n_items) {
      const uint32_t id = order[wbeg + i];
      const uint2 rc = rect[id];   // {minx | maxx << 16, miny | maxy << 16}
      const uint32_t x0 = rc.x & 0xffffu, x1 = rc.x >> 16, y0 = rc.y & 0xffffu, y1 = rc.y >> 16;
      pa[q] = id;
      if ((int)(x1 - x0) * (int)(y1 - y0) > 0) { pb[q] = x0 | (x1 << 8); iv[q] = y0 | (y1 << 8); }
    }
  }
  uint64_t cols[RB_IPL];
  const uint32_t mine = wave_cols(iv, n_items, cols);
  wcnt[wv][lane] = mine;
  __syncthreads();
  if (wv == 0) {
    const uint32_t local = wcnt[0][lane] + wcnt[1][lane] + wcnt[2][lane] + wcnt[3][lane];
    const int chunk = (int)blockIdx.x;
    const int live = (int)((n_vis + RB_CHUNK - 1) / RB_CHUNK);   // group size ~ sqrt(live chunks), never above the launch's
    int gs = 2;
    while ((1 << (2 * gs)) < live) ++gs;
    gs = min(gs, gshift_max);
    const int grp = chunk >> gs, mem = chunk & ((1 << gs) - 1);
    uint32_t* cw = status + lane;                              // [chunk][64]
    uint32_t* gw = status + (size_t)chunks_max * 64 + lane;    // [group][64]
    rb_st(cw + (size_t)chunk * 64, RB_PUB | local);
    const uint32_t in_group = rb_sum_words(cw + ((size_t)grp << gs) * 64, mem);
    if (mem == (1 << gs) - 1) rb_st(gw + (size_t)grp * 64, RB_PUB | (in_group + local));
    s_base[lane] = rowstart + in_group + rb_sum_words(gw, grp);
  }
  __syncthreads();
  uint32_t before = 0, rowtot_c = 0;   // lane = row: this chunk's entries of the earlier waves / of the whole chunk
#pragma unroll
  for (int w = 0; w < RB_WAVES; ++w) { const uint32_t c = wcnt[w][lane]; if (w < wv) before += c; rowtot_c += c; }
  // As in rb_scatter2_kernel: the chunk's entries are staged in LDS in (row, item) order and every row's run leaves as
  // lane-consecutive 8-byte stores -- emitted directly every entry was a lone store into one of 64 row lists (the head-like
  // scene writes 3.3 M of them).  Chunks with more entries than the buffer holds keep the direct form.
  const uint32_t rowend = wave_incl_scan_u32(rowtot_c);          // same in every wave
  const uint32_t total = (uint32_t)__shfl((int)rowend, 63, 64);
  const uint32_t rowbeg = rowend - rowtot_c;
  const uint32_t gbase = s_base[lane];
  if (total <= (uint32_t)RB_STAGE1) {
    uint32_t ldst = rowbeg + before;
    wave_emit<true>(iv, pa, pb, cols, n_items, ldst, slab[wv], [&](uint32_t d, uint32_t id, uint32_t xx) { stage[d] = make_uint2(id, xx); });
    __syncthreads();
#pragma unroll 4
    for (int c = wv; c < 64; c += RB_WAVES) {
      const uint32_t n = (uint32_t)__shfl((int)rowtot_c, c, 64);
      if (n == 0) continue;
      const uint32_t b0 = (uint32_t)__shfl((int)rowbeg, c, 64), g = (uint32_t)__shfl((int)gbase, c, 64);
      for (uint32_t k = lane; k < n; k += 64) {
        const uint32_t d = g + k;
        if (d < ent_cap) ent[d] = stage[b0 + k];
      }
    }
    return;
  }
  uint32_t dst = gbase + before;
  wave_emit<true>(iv, pa, pb, cols, n_items, dst, slab[wv], [&](uint32_t d, uint32_t id, uint32_t xx) {
    if (d < ent_cap) ent[d] = make_uint2(id, xx);
  });
}

// level-2 block -> (row, chunk within the row)
// (one load per lane + a ballot: a binary search over the table is six DEPENDENT global loads, which was most of a
// level-2 workgroup's lifetime)
__device__ __forceinline__ bool rb_block_row(const uint32_t* __restrict__ tab, uint32_t blk, int& row, uint32_t& chunk) {
  const uint32_t first = tab[RB_TAB_ROWBLK + (threadIdx.x & 63)];   // first block of row `lane` (non-decreasing)
  const uint32_t total = tab[RB_TAB_ROWBLK + 64];
  if (blk >= total) return false;
  const int r = __popcll(__ballot(first <= blk)) - 1;               // last row that starts at or before blk
  row = r; chunk = blk - (uint32_t)__shfl((int)first, r, 64);
  return true;
}
// ... and the row's entry range with it: the block table and the row starts are requested TOGETHER (a level-2 workgroup is a
// chain of dependent memory round trips at full occupancy -- block table -> row start -> entries -> tile starts was four of
// them; this form makes it two: {block table, row starts}, then {entries, tile starts, chunk prefixes})
__device__ __forceinline__ bool rb_block_row_range(const uint32_t* __restrict__ tab, uint32_t blk, uint32_t ent_cap, int& row,
                                                   uint32_t& chunk, uint32_t& rbeg, uint32_t& rend) {
  const int lane = threadIdx.x & 63;
  const uint32_t first = tab[RB_TAB_ROWBLK + lane];
  const uint32_t total = tab[RB_TAB_ROWBLK + 64];
  const uint32_t start = tab[RB_TAB_ROWSTART + lane];
  const uint32_t all = tab[RB_TAB_ROWSTART + 64];
  if (blk >= total) return false;
  const int r = __popcll(__ballot(first <= blk)) - 1;
  row = r; chunk = blk - (uint32_t)__shfl((int)first, r, 64);
  rbeg = (uint32_t)__shfl((int)start, r, 64);
  rend = min(r == 63 ? all : (uint32_t)__shfl((int)start, (r + 1) & 63, 64), ent_cap);
  return true;
}

// (counts by an LDS difference array here: the kernel does nothing else, and four bit-matrix transposes per wave -- the form the
// scatter kernels take their counts from, where the emission needs the transposes anyway -- cost more than the atomics:
// measured 7.7 vs 4.8 us on the 1 M cube scene, 14.7 vs 9.0 us on the shell)
__global__ __launch_bounds__(RB_THREADS) void rb_count2_kernel(const uint2* __restrict__ ent, uint32_t ent_cap,
                                                               const uint32_t* __restrict__ tab,
                                                               uint32_t* __restrict__ counts2) {
  __shared__ int diff[65];
  int row; uint32_t chunk, rbeg, rend;
  if (!rb_block_row_range(tab, blockIdx.x, ent_cap, row, chunk, rbeg, rend)) return;
  const int tid = threadIdx.x;
  const uint32_t base = rbeg + chunk * RB_CHUNK;
  uint32_t xs[RB_IPL];
#pragma unroll
  for (int q = 0; q < RB_IPL; ++q) {       // (requested before the barrier below)
    const uint32_t i = base + (uint32_t)q * RB_THREADS + tid;
    xs[q] = i < rend ? ent[i].y : 0u;
  }
  if (tid < 65) diff[tid] = 0;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < RB_IPL; ++q) {
    const uint32_t x0 = xs[q] & 0xffu, x1 = (xs[q] >> 8) & 0xffu;
    if (x1 > x0) { atomicAdd(&diff[x0], 1); atomicAdd(&diff[x1], -1); }
  }
  __syncthreads();
  if (tid < 64) counts2[(size_t)blockIdx.x * 64 + tid] = (uint32_t)wave_incl_scan_i32(diff[tid]);
}

// ---- level-2 scan, one workgroup per tile row: exclusive prefix over the row's chunks (lane = column, 16 waves split
//      the chunks), tile totals, then the start of every tile list.  The only cross-row quantity is the number of
//      instances in the rows above: every workgroup publishes its row total (release) and sums the totals of the lower
//      rows (relaxed poll + acquire).  Workgroups only ever wait for lower-numbered ones.
__global__ __launch_bounds__(1024) void rb_scan2_kernel(uint32_t* __restrict__ counts2, uint32_t* tab, int gx, int gy,
                                                        uint32_t* __restrict__ ranges) {
  __shared__ uint32_t part[16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = blockIdx.x;
  const uint32_t b0 = tab[RB_TAB_ROWBLK + r], b1 = tab[RB_TAB_ROWBLK + r + 1];
  const uint32_t nb = b1 - b0, per = (nb + 15u) / 16u;
  const uint32_t c0 = b0 + min(nb, (uint32_t)wv * per), c1 = min(b1, c0 + per);
  uint32_t s = 0;
#pragma unroll 4
  for (uint32_t b = c0; b < c1; ++b) s += counts2[(size_t)b * 64 + lane];
  part[wv][lane] = s;
  __syncthreads();
  uint32_t run = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) { const uint32_t p = part[w][lane]; if (w < wv) run += p; tot += p; }
#pragma unroll 4
  for (uint32_t b = c0; b < c1; ++b) {
    const size_t idx = (size_t)b * 64 + lane;
    const uint32_t c = counts2[idx];
    counts2[idx] = run;
    run += c;
  }
  if (wv == 0) {
    const uint32_t inc = wave_incl_scan_u32(tot);
    if (lane == 63) {
      tab[RB_TAB_ROWINST + r] = inc;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(&tab[RB_TAB_FLAG + r], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t below = 0;
    if (lane < r) {
      while (__hip_atomic_load(&tab[RB_TAB_FLAG + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {}
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (lane < r) below = __hip_atomic_load(&tab[RB_TAB_ROWINST + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t binc = wave_incl_scan_u32(below);
    const uint32_t rowbase = (uint32_t)__builtin_amdgcn_readlane((int)binc, 63);
    if (lane < gx) {
      const uint32_t start = rowbase + inc - tot;
      tab[RB_TAB_TILESTART + r * 64 + lane] = start;
      const int t = r * gx + lane;
      ranges[2 * t] = tot ? start : 0u;
      ranges[2 * t + 1] = tot ? start + tot : 0u;
    }
  }
}

__global__ __launch_bounds__(RB_THREADS) void rb_scatter2_kernel(const uint2* __restrict__ ent, uint32_t ent_cap,
                                                                 const uint32_t* __restrict__ tab,
                                                                 const uint32_t* __restrict__ prefix2,
                                                                 uint32_t* __restrict__ list, uint32_t capacity,
                                                                 uint32_t main_blocks, ggd_scan_piggy pg) {
  __shared__ uint4 slab[RB_WAVES][64];
  __shared__ uint32_t wcnt[RB_WAVES][64];
  __shared__ uint32_t stage[RB_STAGE];
  if (blockIdx.x >= main_blocks) {   // appended workgroups: last step of the offsets scan (see ggd_scan_piggy)
    scan_apply_block<false>(pg.in, pg.out, pg.n, pg.block_sums, (int)(blockIdx.x - main_blocks), stage, pg.sum_stride);
    return;
  }
  int row; uint32_t chunk, rbeg, rend;
  if (!rb_block_row_range(tab, blockIdx.x, ent_cap, row, chunk, rbeg, rend)) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t wbeg = rbeg + chunk * RB_CHUNK + (uint32_t)wv * RB_WCHUNK;
  const int n_items = (int)min((uint32_t)RB_WCHUNK, rend > wbeg ? rend - wbeg : 0u);
  uint32_t iv[RB_IPL], pa[RB_IPL], pb[RB_IPL];
  // second (and last) round trip: the entries, the tile starts and the chunk's prefixes, all requested before anything waits
#pragma unroll
  for (int q = 0; q < RB_IPL; ++q) {
    const int i = q * 64 + lane;
    uint2 it = make_uint2(0, 0);
    if (i < n_items) it = ent[wbeg + i];
    pa[q] = it.x; pb[q] = 0;
    iv[q] = it.y & 0xffffu;   // x0 | x1 << 8
  }
  const uint32_t gbase = tab[RB_TAB_TILESTART + row * 64 + lane] + prefix2[(size_t)blockIdx.x * 64 + lane];
  uint64_t cols[RB_IPL];
  const uint32_t mine = wave_cols(iv, n_items, cols);
  wcnt[wv][lane] = mine;
  __syncthreads();
  uint32_t before = 0, coltot = 0;   // lane = column: instances of the earlier waves / of the whole chunk
#pragma unroll
  for (int w = 0; w < RB_WAVES; ++w) { const uint32_t c = wcnt[w][lane]; if (w < wv) before += c; coltot += c; }
  // The chunk's instances are staged in LDS in (column, item) order and copied out with every column's run as
  // lane-consecutive stores: emitted directly, the list goes out as lone 4-byte words (2.2x write amplification; those
  // stores were 15 of the kernel's 27 us).  Chunks with more instances than the buffer holds keep the direct form.
  const uint32_t colend = wave_incl_scan_u32(coltot);          // same in every wave
  const uint32_t total = (uint32_t)__shfl((int)colend, 63, 64);
  const uint32_t colbeg = colend - coltot;
  if (total <= (uint32_t)RB_STAGE) {
    uint32_t ldst = colbeg + before;
    wave_emit<false>(iv, pa, pb, cols, n_items, ldst, slab[wv], [&](uint32_t d, uint32_t id, uint32_t) { stage[d] = id; });
    __syncthreads();
    // wave w copies columns w, w + 4, ...; the column's (base, count, destination) come from the lane that owns it
#pragma unroll 4
    for (int c = wv; c < 64; c += RB_WAVES) {
      const uint32_t n = (uint32_t)__shfl((int)coltot, c, 64);
      if (n == 0) continue;
      const uint32_t b0 = (uint32_t)__shfl((int)colbeg, c, 64), g = (uint32_t)__shfl((int)gbase, c, 64);
      for (uint32_t k = lane; k < n; k += 64) {
        const uint32_t d = g + k;
        if (d < capacity) list[d] = stage[b0 + k];
      }
    }
    return;
  }
  uint32_t dst = gbase + before;
  wave_emit<false>(iv, pa, pb, cols, n_items, dst, slab[wv], [&](uint32_t d, uint32_t id, uint32_t) {
    if (d < capacity) list[d] = id;
  });
}

static inline int rb_blocks1(int P) { return (P + RB_CHUNK - 1) / RB_CHUNK; }
static inline uint32_t rb_blocks2(uint32_t cap) { return (cap + RB_CHUNK - 1) / RB_CHUNK + 64; }

#include "ggd_rowbin_wide.inc"

static inline uint32_t rbw_blocks2(uint32_t cap) { return (cap + RB_CHUNK - 1) / RB_CHUNK + RBW_BINS; }
static inline bool rb_is_wide(int W, int H) { return (W + 15) / 16 > 64 || (H + 15) / 16 > 64; }

}  // namespace

// grids up to 64 x 64 tiles: the lane-per-bin kernels above; up to 255 x 255: ggd_rowbin_wide.inc
bool ggd_rowbin_supported(int W, int H) { return W > 0 && H > 0 && (W + 15) / 16 <= 255 && (H + 15) / 16 <= 255; }

size_t ggd_rowbin_tmp_bytes(int P, uint32_t capacity, int W, int H) {
  if (rb_is_wide(W, H)) {
    const int nby = 64 * (((H + 15) / 16 + 63) / 64), nbx = 64 * (((W + 15) / 16 + 63) / 64);
    return ggd_align((size_t)P * sizeof(uint2)) + ggd_align((size_t)rb_blocks1(P) * nby * 4) +
           ggd_align((size_t)RBW_TAB_WORDS * 4) + ggd_align((size_t)capacity * sizeof(uint2)) +
           ggd_align((size_t)rbw_blocks2(capacity) * nbx * 4);
  }
  return ggd_align((size_t)P * sizeof(uint2)) + ggd_align((size_t)rb_blocks1(P) * 64 * 4) +
         ggd_align((size_t)RB_TAB_WORDS * 4) + ggd_align((size_t)capacity * sizeof(uint2)) +
         ggd_align((size_t)rb_blocks2(capacity) * 64 * 4);
}

// capacity: upper bound on num_rendered (the level-1 entry count is <= num_rendered); order = depth-sorted ids.
int ggd_launch_rowbin(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const uint2* rect, const uint32_t* order,
                      const uint32_t* n_vis_ptr, uint32_t* list, uint32_t* ranges, uint32_t capacity, void* tmp,
                      size_t tmp_bytes, const uint32_t* order_alt, const uint32_t* use_alt, const ggd_scan_piggy* apply,
                      const uint32_t* fold_rowtot, uint32_t* fold_status1) {
  if (!ggd_rowbin_supported(prm.width, prm.height)) return ggd_fail(ctx, GGD_E_INVALID, "tile grid too large for row binning");
  if (tmp_bytes < ggd_rowbin_tmp_bytes(prm.P, capacity, prm.width, prm.height)) return ggd_fail(ctx, GGD_E_INVALID, "rowbin tmp too small");
  const int gx = (prm.width + 15) / 16, gy = (prm.height + 15) / 16;
  char* p = static_cast<char*>(tmp);
  if (rb_is_wide(prm.width, prm.height)) {
    const int ngy = (gy + 63) / 64, ngx = (gx + 63) / 64, nby = 64 * ngy, nbx = 64 * ngx;
    uint2* packed = reinterpret_cast<uint2*>(p); p += ggd_align((size_t)prm.P * sizeof(uint2));
    uint32_t* counts1 = reinterpret_cast<uint32_t*>(p); p += ggd_align((size_t)rb_blocks1(prm.P) * nby * 4);
    uint32_t* tab = reinterpret_cast<uint32_t*>(p); p += ggd_align((size_t)RBW_TAB_WORDS * 4);
    uint2* ent = reinterpret_cast<uint2*>(p); p += ggd_align((size_t)capacity * sizeof(uint2));
    uint32_t* counts2 = reinterpret_cast<uint32_t*>(p);
    const int nb1 = rb_blocks1(prm.P);
    const uint32_t nb2 = rbw_blocks2(capacity);
    hipLaunchKernelGGL(rbw_count1_kernel, dim3(nb1), dim3(RB_THREADS), 0, s, rect, order, n_vis_ptr, prm.P, nby, packed,
                       counts1, order_alt, use_alt);
    hipLaunchKernelGGL(rbw_scan1_kernel, dim3(1), dim3(1024), 0, s, counts1, n_vis_ptr, prm.P, ngy, tab, capacity);
    hipLaunchKernelGGL(rbw_scatter1_kernel, dim3(nb1), dim3(RB_THREADS), 0, s, packed, n_vis_ptr, prm.P, ngy, counts1, tab,
                       ent, capacity);
    hipLaunchKernelGGL(rbw_count2_kernel, dim3(nb2), dim3(RB_THREADS), 0, s, ent, capacity, tab, nbx, counts2);
    hipLaunchKernelGGL(rbw_scan2_kernel, dim3(gy), dim3(1024), 0, s, counts2, tab, gx, gy, ngx, ranges);
    hipLaunchKernelGGL(rbw_scatter2_kernel, dim3(nb2 + (apply ? (uint32_t)apply->nb : 0u)), dim3(RB_THREADS), 0, s, ent,
                       capacity, tab, ngx, counts2, list, capacity, nb2, apply ? *apply : ggd_scan_piggy{});
    GGD_HIP(hipGetLastError());
    return GGD_OK;
  }
  uint2* packed = reinterpret_cast<uint2*>(p); p += ggd_align((size_t)prm.P * sizeof(uint2));
  uint32_t* counts1 = reinterpret_cast<uint32_t*>(p); p += ggd_align((size_t)rb_blocks1(prm.P) * 64 * 4);
  uint32_t* tab = reinterpret_cast<uint32_t*>(p); p += ggd_align((size_t)RB_TAB_WORDS * 4);
  uint2* ent = reinterpret_cast<uint2*>(p); p += ggd_align((size_t)capacity * sizeof(uint2));
  uint32_t* counts2 = reinterpret_cast<uint32_t*>(p);
  const int nb1 = rb_blocks1(prm.P);
  const uint32_t nb2 = rb_blocks2(capacity);
  if (fold_rowtot && fold_status1) {
    int gsm = 2;
    while ((1 << (2 * gsm)) < nb1) ++gsm;   // (= rs_gshift(nb1): the group words were laid out for it, ggd_fold_ctl_words)
    hipLaunchKernelGGL(rb_level1_kernel, dim3(nb1), dim3(RB_THREADS), 0, s, rect, order, n_vis_ptr, prm.P, fold_rowtot,
                       fold_status1, nb1, gsm, tab, ent, capacity, order_alt, use_alt);
  } else {
    hipLaunchKernelGGL(rb_count1_kernel, dim3(nb1), dim3(RB_THREADS), 0, s, prm.width, prm.height, rect, order,
                       n_vis_ptr, prm.P, packed, counts1, order_alt, use_alt);
    hipLaunchKernelGGL(rb_scan1_kernel, dim3(1), dim3(1024), 0, s, counts1, n_vis_ptr, prm.P, tab, capacity);
    hipLaunchKernelGGL(rb_scatter1_kernel, dim3(nb1), dim3(RB_THREADS), 0, s, packed, n_vis_ptr, prm.P, counts1, tab,
                       ent, capacity);
  }
  hipLaunchKernelGGL(rb_count2_kernel, dim3(nb2), dim3(RB_THREADS), 0, s, ent, capacity, tab, counts2);
  hipLaunchKernelGGL(rb_scan2_kernel, dim3(gy), dim3(1024), 0, s, counts2, tab, gx, gy, ranges);
  static_assert(RB_THREADS == SCAN_THREADS, "the appended scan workgroups share the launch's block size");
  hipLaunchKernelGGL(rb_scatter2_kernel, dim3(nb2 + (apply ? (uint32_t)apply->nb : 0u)), dim3(RB_THREADS), 0, s, ent, capacity,
                     tab, counts2, list, capacity, nb2, apply ? *apply : ggd_scan_piggy{});
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}
