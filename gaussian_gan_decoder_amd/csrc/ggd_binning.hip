// ggd_binning.hip -- stages a5 (inclusive scan), a6 (duplicateWithKeys), a7 (stable radix sort), a8 (tile ranges).
//
// Integer-only, HBM-bound work; every result here is bit-exact by contract.  The reference
// reaches these stages inside `_C.rasterize_gaussians` (gaussian_renderer/__init__.py:167-175); upstream uses
// cub::DeviceScan / cub::DeviceRadixSort, here they are hand-written for wave64:
//   * scan: two-level (per-block reduce -> one block scans the block sums -> per-block scan+offset);
//   * duplicate: the 256 Gaussians of a block pool their instances and emit them with consecutive lanes writing
//     consecutive instances (coalesced 8 B + 4 B stores) instead of one divergent per-Gaussian loop each;
//   * sort: onesweep LSD radix, 8-bit digits, stable by construction: ONE global histogram kernel for all passes, then
//     one kernel per pass -- per-tile ranking with wave64 ballot matching (rank inside a wave = popcount of lower
//     peer lanes), decoupled look-back over the earlier tiles' digit counts, scatter.  Two instantiations: 64-bit
//     (tile | depth bits) keys over [0, 32+msb(T)) for the duplicateWithKeys path, and 32-bit depth keys with
//     identity values for the tile-binning paths (ggd_rowbin.hip), where culled Gaussians are
//     dropped in pass 0 and constant-digit passes degrade to copies.  Keys embed raw fp32 depth bits, so the sorted
//     order (ties included) is identical to a stable sort of the emission order.
#include "ggd_common.h"

namespace {

#include "ggd_scan.inc"

__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const uint32_t* __restrict__ in, int64_t n,
                                                                   uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t lds4[4];
  scan_reduce_block(in, n, block_sums, (int)blockIdx.x, lds4);
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_blocksums_kernel(uint32_t* __restrict__ block_sums, int nb,
                                                                      uint32_t* __restrict__ d_total,
                                                                      uint32_t* h_total = nullptr) {
  __shared__ uint32_t lds4[4];
  scan_blocksums_block(block_sums, nb, d_total, h_total, lds4);
}

template <bool EXCLUSIVE>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(const uint32_t* __restrict__ in,
                                                                  uint32_t* __restrict__ out, int64_t n,
                                                                  const uint32_t* __restrict__ block_prefix) {
  __shared__ uint32_t lds4[4];
  scan_apply_block<EXCLUSIVE>(in, out, n, block_prefix, (int)blockIdx.x, lds4);
}

template <bool EXCLUSIVE>
int launch_scan(ggd_ctx* ctx, hipStream_t s, const uint32_t* in, uint32_t* out, int64_t n, uint32_t* d_total,
                void* tmp, size_t tmp_bytes, uint32_t* h_total = nullptr) {
  if (n <= 0) {
    if (d_total) GGD_HIP(hipMemsetAsync(d_total, 0, sizeof(uint32_t), s));
    if (h_total) *h_total = 0u;
    return GGD_OK;
  }
  const int nb = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
  if (tmp_bytes < (size_t)nb * sizeof(uint32_t)) return ggd_fail(ctx, GGD_E_INVALID, "scan tmp too small");
  uint32_t* block_sums = static_cast<uint32_t*>(tmp);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(SCAN_THREADS), 0, s, in, n, block_sums);
  hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, block_sums, nb, d_total, h_total);
  hipLaunchKernelGGL(scan_apply_kernel<EXCLUSIVE>, dim3(nb), dim3(SCAN_THREADS), 0, s, in, out, n, block_sums);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

// ------------------------------------------------------------------------------------------------- duplicate --
// 256 Gaussians per block.  LDS holds, per Gaussian of the block, the block-local exclusive instance offset,
// the tile rect origin/width and the depth bits; lanes then walk the block's pooled instance list.
__global__ __launch_bounds__(256) void duplicate_kernel(int P, int W, int H, const uint2* __restrict__ rect,
                                                        const uint32_t* __restrict__ depth_keys,
                                                        const uint32_t* __restrict__ tiles_touched,
                                                        const uint32_t* __restrict__ offsets,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  __shared__ uint32_t s_excl[257];
  __shared__ uint32_t s_origin[256];  // minx | miny << 16
  __shared__ uint32_t s_width[256];
  __shared__ uint32_t s_depth[256];
  __shared__ uint32_t s_base;
  const int tid = threadIdx.x;
  const int first = blockIdx.x * 256;
  const int i = first + tid;
  uint32_t nt = 0, inc = 0;
  if (i < P) { nt = tiles_touched[i]; inc = offsets[i]; }
  const uint32_t excl = inc - nt;
  if (tid == 0) s_base = excl;
  const int gx = (W + 15) / 16;
  uint32_t origin = 0, width = 1, dbits = 0;
  if (nt > 0) {
    const uint2 rc = rect[i];   // {minx | maxx << 16, miny | maxy << 16}, written by the per-Gaussian stage
    origin = (rc.x & 0xffffu) | ((rc.y & 0xffffu) << 16);
    width = (rc.x >> 16) - (rc.x & 0xffffu);
    dbits = depth_keys[i];
  }
  __syncthreads();
  const uint32_t base = s_base;
  s_excl[tid] = excl - base;
  s_origin[tid] = origin; s_width[tid] = width; s_depth[tid] = dbits;
  // block total = inclusive offset of the last valid Gaussian of the block - base
  const int last = min(P, first + 256) - 1 - first;
  if (tid == last) s_excl[256] = inc - base;
  __syncthreads();
  const uint32_t total = s_excl[256];
  if (tid > last) s_excl[tid] = total;  // padding entries (only when the block is ragged) never match
  __syncthreads();
  for (uint32_t j = tid; j < total; j += 256) {
    // largest g in [0,256) with s_excl[g] <= j  (entries are non-decreasing)
    int lo = 0;
#pragma unroll
    for (int step = 128; step >= 1; step >>= 1) {
      const int probe = lo + step;
      if (probe < 256 && s_excl[probe] <= j) lo = probe;
    }
    // s_excl may hold runs of equal values (Gaussians with zero tiles): the LAST of a run owns instance j only
    // if it has tiles; the search above already returns the last index with excl <= j, which is the owner.
    const uint32_t k = j - s_excl[lo];
    const uint32_t w = s_width[lo];
    const uint32_t ry = k / w, rx = k - ry * w;
    const uint32_t org = s_origin[lo];
    const uint32_t tile = ((org >> 16) + ry) * (uint32_t)gx + (org & 0xffffu) + rx;
    const size_t dst = (size_t)base + j;
    keys[dst] = ((uint64_t)tile << 32) | (uint64_t)s_depth[lo];
    vals[dst] = (uint32_t)(first + lo);
  }
}

// ----------------------------------------------------------------------------------------------- radix sort ---
// "Onesweep" LSD radix sort: ONE kernel per 8-bit digit pass (+ one up-front kernel that histograms every digit).
//   * global digit histograms are permutation-invariant, so all passes' bin bases come from one read of the keys;
//   * inside a pass every 4096-pair tile computes its digit counts, publishes them, and obtains the sum over all
//     EARLIER tiles from a fixed two-level tree: tiles form groups of G = 2^gshift (G ~ sqrt(#tiles)); the last tile
//     of a group sums the group's counts and publishes the group aggregate; a tile then adds the aggregates of the
//     earlier groups and the counts of the earlier tiles of its own group -- at most 2G independent loads, three
//     dependent memory round trips per pass.  (The classic decoupled look-back chain needs ~#tiles / 16 dependent
//     round trips here: all tiles of a pass start together, so nobody finds an inclusive prefix nearby -- 15 round
//     trips and 7.5 M uncached status loads per pass at 245 tiles, the 14 us floor the passes used to have.)
//     A status word per (tile | group, digit) = flag bit | 30-bit count, written / read as single agent-scope relaxed
//     atomics (the value IS the flag, so no fence is needed and the protocol is placement independent: per-XCD L2s
//     are not coherent, agent-scope atomics bypass them); tiles take their index from an atomic ticket and only ever
//     wait for lower-numbered tiles, i.e. for tiles that are already running;
//   * stability: element order inside a tile = (wave, round, lane); rank inside a wave by ballot matching.
constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 16;                      // keys per lane
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;    // 4096 pairs per tile
constexpr int RS_BINS = 256;
constexpr int RS_MAX_PASSES = 8;
constexpr int RS32_ITEMS = 16;                   // tile size of the 32-bit (depth) sort (4 was slower: longer look-back)
constexpr int RS32_TILE = RS_THREADS * RS32_ITEMS;
constexpr int RS_HWORDS = RS_MAX_PASSES * RS_BINS;
constexpr int RS_RESIDENT = 1024;   // tiles (256-thread workgroups, 5 KB LDS) that are certainly resident together on 256 CUs
constexpr int RS_BATCH = 16;     // status words requested together while summing predecessors (batches are dependent round trips)
constexpr uint32_t RS_FLAG = 1u << 30, RS_COUNT_MASK = (1u << 30) - 1u;
// tiles per group of the two-level prefix: the power of two nearest to sqrt(ntiles) from above
static inline int rs_gshift(int64_t ntiles) { int g = 2; while (((int64_t)1 << (2 * g)) < ntiles) ++g; return g; }
// per pass: tile words, then group words.  A pass picks its group size from the number of tiles that hold elements (a
// device-side count in the compacting sort): g <= rs_gshift(ntiles) with up to 2^g groups in use -- more than
// ntiles >> rs_gshift(ntiles) when the count falls just below a power of four (257 launched tiles, 256 live ones: 16 groups
// against 9 rows; the overflow landed in the next pass's tile words).  2^rs_gshift + 1 rows cover every choice.
static inline int64_t rs_status_words(int64_t ntiles) {
  return (ntiles + ((int64_t)1 << rs_gshift(ntiles)) + 1) * RS_BINS;
}

__device__ __forceinline__ uint32_t rs_load(uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rs_store(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ghist[pass * 256 + digit] += occurrences, for every pass at once.  High key bytes are heavily skewed (tile bits of
// consecutive instances, exponent byte of the depth): when a whole wave shares the digit one lane adds the count
// instead of 64 conflicting LDS atomics.
// SKIP (depth sort): keys equal to ~0 (culled Gaussians) are not counted; *n_valid += number of counted keys.
template <typename KeyT, int ITEMS, bool SKIP = false>
__global__ __launch_bounds__(RS_THREADS) void sort_global_hist_kernel(const KeyT* __restrict__ keys, int64_t n,
                                                                      int passes, uint32_t* __restrict__ ghist,
                                                                      uint32_t* __restrict__ n_valid = nullptr,
                                                                      uint32_t* __restrict__ zero_ptr = nullptr,
                                                                      size_t zero_words = 0, int main_blocks = 0x7fffffff,
                                                                      ggd_scan_piggy pg = ggd_scan_piggy{}) {
  __shared__ uint32_t s_hist[RS_MAX_PASSES * RS_BINS];
  if ((int)blockIdx.x >= main_blocks) {   // appended workgroups: step 1 of the offsets scan (independent of the histogram)
    scan_reduce_block(pg.in, pg.n, pg.block_sums, (int)blockIdx.x - main_blocks, s_hist);
    return;
  }
  const size_t nmain = (size_t)min((int)gridDim.x, main_blocks);
  // the passes' status words are cleared here (every workgroup a slice) when the caller skipped the memset
  for (size_t i = (size_t)blockIdx.x * RS_THREADS + threadIdx.x; i < zero_words; i += nmain * RS_THREADS)
    zero_ptr[i] = 0u;
  uint32_t my_valid = 0;
  for (int b = threadIdx.x; b < passes * RS_BINS; b += RS_THREADS) s_hist[b] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (RS_THREADS * ITEMS);
  const int lane = threadIdx.x & 63;
  // all ITEMS loads of the lane are issued before the first key is consumed (one workgroup per CU: the kernel is bound
  // by the latency of its loads, not by their bandwidth)
  KeyT kreg[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t idx = base + (int64_t)k * RS_THREADS + threadIdx.x;
    kreg[k] = idx < n ? keys[idx] : (KeyT)0;
  }
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t idx = base + (int64_t)k * RS_THREADS + threadIdx.x;
    const KeyT key = kreg[k];
    const bool ok = idx < n && !(SKIP && key == (KeyT)~(KeyT)0);
    const uint64_t act = __ballot(ok);
    if (act == 0ull) continue;
    if (SKIP && (threadIdx.x & 63) == 0) my_valid += (uint32_t)__popcll(act);
    const int leader = __builtin_ctzll(act);
    for (int p = 0; p < passes; ++p) {
      const uint32_t d = (uint32_t)(key >> (8 * p)) & 0xffu;
      if (p < 2) {   // the two low bytes (mantissa bits) are never shared by a wave: no point in looking
        if (ok) atomicAdd(&s_hist[p * RS_BINS + d], 1u);
        continue;
      }
      const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, leader);
      if (__ballot(ok && d != d0) == 0ull) {
        if (lane == leader) atomicAdd(&s_hist[p * RS_BINS + d0], (uint32_t)__popcll(act));
      } else if (ok) {
        atomicAdd(&s_hist[p * RS_BINS + d], 1u);
      }
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < passes * RS_BINS; b += RS_THREADS) {
    const uint32_t c = s_hist[b];
    if (c) atomicAdd(&ghist[b], c);
  }
  if (SKIP && n_valid && my_valid) atomicAdd(n_valid, my_valid);
}

// The kept keys' range of this frame (reduced from the preprocess workgroups' records by scan_info_block), stored -- BEFORE the
// tagged word's release store, by the same thread -- into d_total[3..5] and the pinned words 4..6: {min key, max key, flags (bit 0:
// top byte constant, 1: no key outside the two-launch sort's window, 2: no bucket above its capacity)}.  The host fits the next
// frames' window to these ranges.
__device__ __forceinline__ void fold_publish_range(const ggd_scan_piggy& pg, uint32_t flags, uint32_t nmin, uint32_t kmax) {
  if (threadIdx.x != 0) return;
  if (pg.d_total) { pg.d_total[3] = ~nmin; pg.d_total[4] = kmax; pg.d_total[5] = flags; }
  if (pg.h_total) { pg.h_total[4] = ~nmin; pg.h_total[5] = kmax; pg.h_total[6] = flags; }
}

// COMPACT (depth sort): pass 0 (IOTA) drops keys equal to ~0 -- they are neither ranked nor written -- and every later
// pass takes its element count from *n_dev (= number of kept keys, written by the histogram kernel).  A later pass
// whose digit is the same for all elements (ghist[d] == n: e.g. the exponent byte of a scene's depth range) degrades to
// a tile copy: no ranking, no look-back.
// STAGE (large inputs): the tile's pairs are first put in digit order in LDS and copied out from there, so that the 16 or so
// pairs a tile holds per digit leave as one run of lane-consecutive stores -- scattered straight from the registers every
// pair is a lone 4-byte store, and from ~1 M kept keys on those partial-line writes, not the look-back, set a pass's time.
template <typename KeyT, bool IOTA, int ITEMS, bool COMPACT = false, bool STAGE = false>
__global__ __launch_bounds__(RS_THREADS) void sort_onesweep_kernel(
    const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, KeyT* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, int64_t n_host, int shift, const uint32_t* __restrict__ ghist /*[256] of this pass*/,
    uint32_t* status /*[ntiles + groups][256]*/, int ntiles, int gshift, uint32_t* ticket,
    const uint32_t* __restrict__ n_dev = nullptr, int piggy_role = 0, ggd_scan_piggy pg = ggd_scan_piggy{},
    uint32_t* __restrict__ flat_flag = nullptr, int hist_reps = 1) {
  __shared__ uint32_t s_cnt[4][RS_BINS];   // per-wave running digit counts -> per-wave scatter bases
  if ((int)blockIdx.x >= ntiles) {   // appended workgroups: steps 2 / 3 of the offsets scan (see ggd_scan_piggy)
    if (piggy_role == 2 && pg.wg_info) {
      const uint4 tv = scan_info_block(pg.wg_info, pg.n_info, pg.block_sums, pg.n_valid, pg.d_total, pg.h_total, &s_cnt[0][0]);
      int flat = 0;
      if (pg.fold_hist) {   // passes 1 .. 3 then read ONE histogram (pass 0's tiles, running beside us, read their own 256 bins
                            // of every replica: reading all replicas cost each pass ~2 us)
        uint32_t acc[3] = {0u, 0u, 0u};
#pragma unroll 8
        for (int r = 1; r < GGD_FOLD_REPS; ++r) {
#pragma unroll
          for (int p = 1; p < 4; ++p) acc[p - 1] += pg.fold_hist[r * GGD_FOLD_REP_STRIDE + p * RS_BINS + threadIdx.x];
        }
#pragma unroll
        for (int p = 1; p < 4; ++p) {
          acc[p - 1] += pg.fold_hist[p * RS_BINS + threadIdx.x];
          pg.fold_hist[p * RS_BINS + threadIdx.x] = acc[p - 1];
        }
        flat = __syncthreads_or(acc[2] == tv.y);   // the top digit of every kept key is the same (also: nothing kept)
        fold_publish_range(pg, flat ? 1u : 0u, tv.z, tv.w);
      }
      if (threadIdx.x == 0) {
        if (pg.spec_flat && pg.flat_flag) *pg.flat_flag = 1u;
        if (pg.d_total) pg.d_total[2] = (uint32_t)flat;
        if (pg.h_tagged)   // one 64-bit store: the host sees tag, flag and total together
          __hip_atomic_store(pg.h_tagged, ((unsigned long long)flat << 63) | ((unsigned long long)(pg.tag & 0x3fffffffu) << 32) | tv.x,
                             __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    else if (piggy_role == 2)
      scan_blocksums_block(pg.block_sums, pg.nb, pg.d_total, pg.h_total, &s_cnt[0][0], pg.h_tagged, pg.tag);
    else
      scan_apply_block<false>(pg.in, pg.out, pg.n, pg.block_sums, (int)blockIdx.x - ntiles, &s_cnt[0][0], pg.sum_stride);
    return;
  }
  __shared__ uint32_t s_scan[4];
  __shared__ uint32_t s_tile;
  __shared__ uint32_t s_flat;
  __shared__ KeyT s_keys[STAGE ? RS_THREADS * ITEMS : 1];
  __shared__ uint32_t s_vals[STAGE ? RS_THREADS * ITEMS : 1];
  __shared__ uint32_t s_gdst[STAGE ? RS_BINS : 1];
  __shared__ uint32_t s_ntile;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // A pass is a chain of dependent memory round trips (1-2 us each, one workgroup per CU, nothing to overlap them with):
  // everything that does not depend on anything else is requested up front -- the element count, this thread's bin
  // of the global histogram, and the tile's keys / values (bounded by the host's n; masked by the device's n later).
  // With all tiles resident at once (ntiles <= RS_RESIDENT) the tile index is blockIdx.x -- workgroups are started in
  // index order and nobody is ever descheduled, so waiting on lower indices cannot deadlock; beyond that the ticket.
  const bool use_ticket = ntiles > RS_RESIDENT;
  if (use_ticket && threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  const uint32_t n_dev_v = (COMPACT && !IOTA) ? *n_dev : 0u;
  uint32_t my_bin = ghist[threadIdx.x];
  if (hist_reps > 1) {   // folded front end: GGD_FOLD_REPS replicas, all requested at once (a loop on the runtime count made
                         // every replica its own round trip: + 7 us per pass)
    uint32_t rep[GGD_FOLD_REPS - 1];
#pragma unroll
    for (int r = 1; r < GGD_FOLD_REPS; ++r) rep[r - 1] = ghist[r * GGD_FOLD_REP_STRIDE + threadIdx.x];
#pragma unroll
    for (int r = 1; r < GGD_FOLD_REPS; ++r) my_bin += rep[r - 1];
  }
  for (int b = threadIdx.x; b < 4 * RS_BINS; b += RS_THREADS) (&s_cnt[0][0])[b] = 0;
  if (threadIdx.x == 0) s_flat = 0u;
  uint32_t tile = blockIdx.x;
  if (use_ticket) { __syncthreads(); tile = s_tile; }

  const int64_t wbase = (int64_t)tile * (RS_THREADS * ITEMS) + (int64_t)wv * (64 * ITEMS);
  KeyT key[ITEMS];
  uint32_t val[ITEMS];
  uint32_t rank[ITEMS];
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int64_t idx = wbase + r * 64 + lane;
    const bool ok = idx < n_host;
    key[r] = ok ? keys_in[idx] : (KeyT)~(KeyT)0;
    val[r] = IOTA ? (uint32_t)idx : (ok ? vals_in[idx] : 0u);
  }
  const int64_t n = (COMPACT && !IOTA) ? (int64_t)n_dev_v : n_host;
  if (COMPACT && (int64_t)tile * (RS_THREADS * ITEMS) >= n) return;   // beyond the kept elements (uniform per block)
  if (!use_ticket) __syncthreads();                                    // s_cnt / s_flat are zeroed
  if (COMPACT && !IOTA) {
    // constant digit (one bin of the pass's histogram holds every element): the pass is the identity permutation
    if ((int64_t)my_bin == n) s_flat = 1u;
    __syncthreads();
    if (s_flat != 0u) {
      if (flat_flag) {   // last pass: the consumer reads the input buffer instead (no copy at all)
        if (tile == 0u && threadIdx.x == 0) *flat_flag = 1u;
        return;
      }
#pragma unroll
      for (int r = 0; r < ITEMS; ++r) {
        const int64_t idx = wbase + r * 64 + lane;
        if (idx < n) { keys_out[idx] = key[r]; vals_out[idx] = val[r]; }
      }
      return;
    }
  }
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int64_t idx = wbase + r * 64 + lane;
    const bool ok = idx < n && !(COMPACT && IOTA && key[r] == (KeyT)~(KeyT)0);
    const uint32_t d = (uint32_t)(key[r] >> shift) & 0xffu;
    uint64_t peers = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const uint32_t before = s_cnt[wv][d];  // all peers read the same word (LDS broadcast)
    const uint32_t below = (uint32_t)__popcll(peers & lt_mask);
    rank[r] = before + below;
    __builtin_amdgcn_wave_barrier();       // every peer has read `before` before the leader updates it
    if (ok && below == 0) s_cnt[wv][d] = before + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {
    // thread d owns digit d: tile-local count, look-back over earlier tiles, global base
    const int d = threadIdx.x;
    const uint32_t c0 = s_cnt[0][d], c1 = s_cnt[1][d], c2 = s_cnt[2][d], c3 = s_cnt[3][d];
    const uint32_t local = c0 + c1 + c2 + c3;
    // sum of `cnt` published words p[0], p[stride], ...: RS_BATCH requests in flight, unpublished ones are polled
    auto sum_words = [&](uint32_t* p, int cnt) {
      uint32_t acc = 0;
      for (int b0 = 0; b0 < cnt; b0 += RS_BATCH) {
        uint32_t v[RS_BATCH];
#pragma unroll
        for (int i = 0; i < RS_BATCH; ++i) v[i] = (b0 + i < cnt) ? rs_load(p + (size_t)(b0 + i) * RS_BINS) : RS_FLAG;
#pragma unroll
        for (int i = 0; i < RS_BATCH; ++i) {
          uint32_t x = v[i];
          if ((x >> 30) == 0u) {
            uint32_t* q = p + (size_t)(b0 + i) * RS_BINS;
            do { __builtin_amdgcn_s_sleep(1); x = rs_load(q); } while ((x >> 30) == 0u);
          }
          acc += x & RS_COUNT_MASK;
        }
      }
      return acc;
    };
    // group size from the number of tiles that hold elements (known on the device only in the compacting sort):
    // G = 2^gshift ~ sqrt(live tiles), never above the launch's (the group words were laid out for that)
    const int64_t live_tiles = (n + RS_THREADS * ITEMS - 1) / (RS_THREADS * ITEMS);
    int gs = 2;
    while (((int64_t)1 << (2 * gs)) < live_tiles) ++gs;
    gshift = min(gshift, gs);
    const int grp = (int)(tile >> gshift), mem = (int)(tile & ((1u << gshift) - 1u));
    uint32_t* tile_words = status + d;                               // [tile][256]
    uint32_t* group_words = status + (size_t)ntiles * RS_BINS + d;   // [group][256]
    rs_store(tile_words + (size_t)tile * RS_BINS, RS_FLAG | local);
    const uint32_t in_group = sum_words(tile_words + ((size_t)grp << gshift) * RS_BINS, mem);   // earlier tiles of my group
    if (mem == (1 << gshift) - 1) rs_store(group_words + (size_t)grp * RS_BINS, RS_FLAG | (in_group + local));
    const uint32_t excl = in_group + sum_words(group_words, grp);                              // earlier groups
    uint32_t tot;
    const uint32_t gbase = block_exclusive_scan_256(my_bin, &tot, s_scan);  // contains __syncthreads
    const uint32_t base = gbase + excl;
    if constexpr (STAGE) {
      uint32_t ltot;
      const uint32_t lexcl = block_exclusive_scan_256(local, &ltot, s_scan);   // digit d's first slot inside the tile
      s_cnt[0][d] = lexcl; s_cnt[1][d] = lexcl + c0; s_cnt[2][d] = lexcl + c0 + c1; s_cnt[3][d] = lexcl + c0 + c1 + c2;
      s_gdst[d] = base - lexcl;          // slot p of the tile (digit d) goes to s_gdst[d] + p (mod 2^32)
      if (d == 0) s_ntile = ltot;
    } else {
      s_cnt[0][d] = base; s_cnt[1][d] = base + c0; s_cnt[2][d] = base + c0 + c1; s_cnt[3][d] = base + c0 + c1 + c2;
    }
  }
  __syncthreads();
  if constexpr (STAGE) {
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      const int64_t idx = wbase + r * 64 + lane;
      if (idx < n && !(COMPACT && IOTA && key[r] == (KeyT)~(KeyT)0)) {
        const uint32_t d = (uint32_t)(key[r] >> shift) & 0xffu;
        const uint32_t p = s_cnt[wv][d] + rank[r];
        s_keys[p] = key[r];
        s_vals[p] = val[r];
      }
    }
    __syncthreads();
    const uint32_t nt = s_ntile;
    for (uint32_t p = threadIdx.x; p < nt; p += RS_THREADS) {
      const KeyT k = s_keys[p];
      const uint32_t d = (uint32_t)(k >> shift) & 0xffu;
      const size_t dst = (size_t)(uint32_t)(s_gdst[d] + p);
      keys_out[dst] = k;
      vals_out[dst] = s_vals[p];
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int64_t idx = wbase + r * 64 + lane;
    if (idx < n && !(COMPACT && IOTA && key[r] == (KeyT)~(KeyT)0)) {
      const uint32_t d = (uint32_t)(key[r] >> shift) & 0xffu;
      const size_t dst = (size_t)s_cnt[wv][d] + rank[r];
      keys_out[dst] = key[r];
      vals_out[dst] = val[r];
    }
  }
}


// ------------------------------------------------------------------------------- two-launch depth sort (msd) ---
#include "ggd_msd_finish.inc"
constexpr int MSD_DIGIT_BITS = 10;
static_assert((1 << MSD_DIGIT_BITS) == GGD_MSD_BINS && GGD_MSD_BINS % (4 * RS_THREADS) == 0, "1024 buckets");
// See ggd_common.h (GGD_MSD_*).  Launch 1: a tile partitions its own 4096 keys by their bucket in the key window -- the ranking of a onesweep
// pass with 1024 digits, but nothing is published and nobody is waited for: the tile's keys go, in digit order, to the tile's
// own region of (keys_out, vals_out), and table[tile][digit] = (first slot inside the tile << 16 | count).

// the workgroup appended to launch 1: step 2 of the offsets scan (as in the onesweep form), the sum of the histogram replicas
// (nobody else reads them during this launch: the totals go to replica 0, where launch 2 reads its bucket bases), and the
// verdict on the speculation: no kept key outside the window and no bucket above GGD_MSD_CAP (bit 62 of the tagged word; bit 63
// keeps saying "top byte constant", for the three-pass form's streak)
__device__ __forceinline__ void msd_piggy_block(const ggd_scan_piggy& pg, uint32_t* lds) {
  constexpr int NS = GGD_FOLD_REP_STRIDE / 256;   // 256-word slabs per replica: the bucket histogram, then the top byte's
  // (the replicas' words and the outside-the-window count are requested before the scan, all at once: nothing here depends on it)
  uint32_t rep[GGD_FOLD_REPS][NS];
#pragma unroll
  for (int r = 0; r < GGD_FOLD_REPS; ++r)
#pragma unroll
    for (int q = 0; q < NS; ++q) rep[r][q] = pg.fold_hist[r * GGD_FOLD_REP_STRIDE + q * 256 + threadIdx.x];
  const uint32_t outside = pg.fold_hist[GGD_FOLD_OUTSIDE];
  const uint4 tv = scan_info_block(pg.wg_info, pg.n_info, pg.block_sums, pg.n_valid, pg.d_total, pg.h_total, lds);
  uint32_t acc[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) acc[q] = 0u;
#pragma unroll
  for (int r = 0; r < GGD_FOLD_REPS; ++r) {
#pragma unroll
    for (int q = 0; q < NS; ++q) acc[q] += rep[r][q];
  }
  bool over = false;
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    pg.fold_hist[q * 256 + threadIdx.x] = acc[q];
    if (q < NS - 1) over = over || acc[q] > (uint32_t)GGD_MSD_CAP;
  }
  const int flat = __syncthreads_or(acc[NS - 1] == tv.y);   // (also: nothing kept)
  const int big = __syncthreads_or(over);
  const bool inside = outside == 0u;
  const unsigned long long ok = (inside && !big) ? 1ull : 0ull;
  fold_publish_range(pg, (flat ? 1u : 0u) | (inside ? 2u : 0u) | (big ? 0u : 4u), tv.z, tv.w);
  if (threadIdx.x == 0) {
    if (pg.d_total) pg.d_total[2] = (uint32_t)(flat ? 1u : 0u) | ((uint32_t)ok << 1);
    if (pg.h_tagged)
      __hip_atomic_store(pg.h_tagged, ((unsigned long long)(flat ? 1 : 0) << 63) | (ok << 62) |
                                      ((unsigned long long)(pg.tag & 0x3fffffffu) << 32) | tv.x,
                         __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(RS_THREADS) void sort_msd_partition_kernel(const uint32_t* __restrict__ keys_in,
                                                                        uint32_t* __restrict__ keys_out,
                                                                        uint32_t* __restrict__ vals_out, int64_t n,
                                                                        uint32_t* __restrict__ table, int ntiles,
                                                                        uint32_t lo, int shift, ggd_scan_piggy pg) {
  __shared__ uint32_t s_cnt[4][GGD_MSD_BINS];
  __shared__ uint32_t s_keys[MSD_TILE];
  __shared__ uint32_t s_vals[MSD_TILE];
  __shared__ uint32_t s_scan[4];
  if ((int)blockIdx.x >= ntiles) { msd_piggy_block(pg, &s_cnt[0][0]); return; }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t tile = blockIdx.x;
  const int64_t wbase = (int64_t)tile * MSD_TILE + (int64_t)wv * (64 * MSD_ITEMS);
  uint32_t key[MSD_ITEMS], rank[MSD_ITEMS];
#pragma unroll
  for (int r = 0; r < MSD_ITEMS; ++r) {
    const int64_t idx = wbase + r * 64 + lane;
    key[r] = idx < n ? keys_in[idx] : 0xffffffffu;    // (0xFFFFFFFF = culled: dropped here, as in the onesweep form's pass 0)
  }
  // from here on a kept key is its offset inside the window (what the finish kernel orders by its low `shift` bits), and its
  // bucket the offset's high part, clamped (a key outside the window has already failed the frame: it only has to stay a member
  // of a valid permutation, and the preprocess kernel's histogram clamps the same way)
  uint32_t dig[MSD_ITEMS];
#pragma unroll
  for (int r = 0; r < MSD_ITEMS; ++r) {
    const bool ok = key[r] != 0xffffffffu;
    const uint32_t rel = key[r] - lo;
    dig[r] = min(rel >> shift, (uint32_t)(GGD_MSD_BINS - 1));
    key[r] = ok ? rel : 0xffffffffu;
    if (!ok) dig[r] = 0xffffffffu;
  }
  for (int b = threadIdx.x; b < 4 * GGD_MSD_BINS; b += RS_THREADS) (&s_cnt[0][0])[b] = 0;
  __syncthreads();
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < MSD_ITEMS; ++r) {
    const bool ok = dig[r] != 0xffffffffu;
    const uint32_t d = dig[r] & (GGD_MSD_BINS - 1);
    uint64_t peers = __ballot(ok);
#pragma unroll
    for (int b = 0; b < MSD_DIGIT_BITS; ++b) {
      const uint64_t m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const uint32_t before = s_cnt[wv][d];
    const uint32_t below = (uint32_t)__popcll(peers & lt_mask);
    rank[r] = before + below;
    __builtin_amdgcn_wave_barrier();
    if (ok && below == 0) s_cnt[wv][d] = before + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  uint32_t ltot;
  {
    // thread t owns digits DPT t .. DPT t + DPT - 1 (in groups of four)
    constexpr int DPT = GGD_MSD_BINS / RS_THREADS, G = DPT / 4;
    uint4 c[G][4];
    uint32_t mine = 0;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        c[g][w] = *reinterpret_cast<const uint4*>(&s_cnt[w][DPT * threadIdx.x + 4 * g]);
        mine += c[g][w].x + c[g][w].y + c[g][w].z + c[g][w].w;
      }
    uint32_t e = block_exclusive_scan_256(mine, &ltot, s_scan);   // contains __syncthreads
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const uint32_t l0 = c[g][0].x + c[g][1].x + c[g][2].x + c[g][3].x, l1 = c[g][0].y + c[g][1].y + c[g][2].y + c[g][3].y;
      const uint32_t l2 = c[g][0].z + c[g][1].z + c[g][2].z + c[g][3].z, l3 = c[g][0].w + c[g][1].w + c[g][2].w + c[g][3].w;
      const uint32_t e0 = e, e1 = e0 + l0, e2 = e1 + l1, e3 = e2 + l2;
      e = e3 + l3;
      uint4 run = make_uint4(e0, e1, e2, e3);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        *reinterpret_cast<uint4*>(&s_cnt[w][DPT * threadIdx.x + 4 * g]) = run;
        run.x += c[g][w].x; run.y += c[g][w].y; run.z += c[g][w].z; run.w += c[g][w].w;
      }
      *reinterpret_cast<uint4*>(table + (size_t)tile * GGD_MSD_BINS + DPT * threadIdx.x + 4 * g) =
          make_uint4((e0 << 16) | l0, (e1 << 16) | l1, (e2 << 16) | l2, (e3 << 16) | l3);
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < MSD_ITEMS; ++r) {
    if (dig[r] != 0xffffffffu) {
      const uint32_t d = dig[r];
      const uint32_t p = s_cnt[wv][d] + rank[r];
      s_keys[p] = key[r];
      s_vals[p] = (uint32_t)(wbase + r * 64 + lane);
    }
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < ltot; p += RS_THREADS) {
    keys_out[(size_t)tile * MSD_TILE + p] = s_keys[p];
    vals_out[(size_t)tile * MSD_TILE + p] = s_vals[p];
  }
}

// Launch 2 (see ggd_msd_finish.inc): the bucket's sorted run written to its final place (bucket bases = prefix of the histogram).
// (keys_out: only the exchange area of a bucket above MSD_XCAP -- nobody behind the sort reads sorted keys, the binning takes
// the order of the indices)
__global__ __launch_bounds__(MSDF_THREADS, 8) void sort_msd_finish_kernel(const uint32_t* __restrict__ keys_in,
                                                                       const uint32_t* __restrict__ vals_in,
                                                                       uint32_t* __restrict__ keys_out,
                                                                       uint32_t* __restrict__ vals_out,
                                                                       const uint32_t* __restrict__ hist /* [GGD_MSD_BINS] totals */,
                                                                       const uint32_t* __restrict__ table, int ntiles, int kshift) {
  __shared__ msd_lds L;
  const int tid = threadIdx.x;
  const uint32_t b = blockIdx.x;
  // (everything the workgroup needs before it can gather -- its bucket's size, the histogram below it, its column of the table --
  // is requested at once: as three dependent trips they were a third of a small bucket's lifetime)
  const uint32_t nb = hist[b];
  constexpr int HPT = GGD_MSD_BINS / MSDF_THREADS;
  uint32_t below = 0;
#pragma unroll
  for (int q = 0; q < HPT; ++q) { const uint32_t i = (uint32_t)(HPT * tid + q); const uint32_t h = hist[i]; if (i < b) below += h; }
  const uint32_t e0 = tid < ntiles ? table[(size_t)tid * GGD_MSD_BINS + b] : 0u;
  if (nb == 0u) return;
  uint32_t base;
  block_exclusive_scan_1024(below, &base, L.part);
  const bool oversized = nb > (uint32_t)GGD_MSD_CAP;
  msd_bucket_pieces(L, table, ntiles, b, e0, oversized);
  if (oversized) {
    // more keys than the workgroup holds: the frame's histogram check has already failed and the host renders the frame again --
    // but the kernels behind this one still run, so they must find a valid permutation: the pieces, gathered in tile order
    for (uint32_t p = tid; p < nb; p += MSDF_THREADS) {
      const size_t src = msd_piece_src_big(L, table, b, ntiles, p);
      vals_out[(size_t)base + p] = vals_in[src];
    }
    return;
  }
  if (nb > (uint32_t)MSD_XCAP) {   // exchange through the bucket's slice of the output; pass 2 leaves the final order there
    msd_bucket_sort<true, MSDF_ITEMS>(L, keys_in, vals_in, ntiles, nb, kshift, keys_out + base, vals_out + base);
    return;
  }
  msd_bucket_sort<false, MSDX_ITEMS>(L, keys_in, vals_in, ntiles, nb, kshift, nullptr, nullptr);
  for (uint32_t p = tid; p < nb; p += MSDF_THREADS) vals_out[(size_t)base + p] = L.kv[MSD_XCAP + p];
}

// ------------------------------------------------------------------------------------------- tile ranges -----
__global__ __launch_bounds__(256) void ranges_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                     uint32_t* __restrict__ ranges) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t tile = (uint32_t)(keys[i] >> 32);
  if (i == 0) {
    ranges[2 * tile] = 0;
  } else {
    const uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
    if (prev != tile) {
      ranges[2 * prev + 1] = (uint32_t)i;
      ranges[2 * tile] = (uint32_t)i;
    }
  }
  if (i == n - 1) ranges[2 * tile + 1] = (uint32_t)n;
}

}  // namespace

size_t ggd_scan_tmp_bytes(int64_t n) {
  const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  return ggd_align((size_t)(nb > 0 ? nb : 1) * sizeof(uint32_t));
}

int ggd_launch_inclusive_scan(ggd_ctx* ctx, hipStream_t s, const uint32_t* in, uint32_t* out, int64_t n,
                              uint32_t* d_total, void* tmp, size_t tmp_bytes) {
  return launch_scan<false>(ctx, s, in, out, n, d_total, tmp, tmp_bytes);
}

int ggd_launch_inclusive_scan_ex(ggd_ctx* ctx, hipStream_t s, const uint32_t* in, uint32_t* out, int64_t n,
                                 uint32_t* d_total, void* tmp, size_t tmp_bytes, uint32_t* h_total) {
  return launch_scan<false>(ctx, s, in, out, n, d_total, tmp, tmp_bytes, h_total);
}
int ggd_scan_blocks(int64_t n) { return n > 0 ? (int)((n + SCAN_TILE - 1) / SCAN_TILE) : 0; }

int ggd_launch_duplicate(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const uint2* rect, const uint32_t* depth_keys,
                         const uint32_t* tiles_touched, const uint32_t* offsets, uint64_t* keys, uint32_t* vals) {
  if (prm.P == 0) return GGD_OK;
  hipLaunchKernelGGL(duplicate_kernel, dim3((prm.P + 255) / 256), dim3(256), 0, s, prm.P, prm.width, prm.height,
                     rect, depth_keys, tiles_touched, offsets, keys, vals);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

static inline int sort_passes(int nbits) { return (nbits + 7) / 8; }
int ggd_sort_input_is_alt(int nbits) { return sort_passes(nbits) & 1; }

// tmp layout: [ghist: MAX_PASSES*256 u32][tickets: MAX_PASSES u32 (padded)][status: passes * ntiles * 256 u32]
static inline size_t sort_ctrl_bytes() { return ggd_align((size_t)(RS_HWORDS + 64) * sizeof(uint32_t)); }
size_t ggd_sort_tmp_bytes(int64_t n) {
  const int64_t ntiles = (n + RS_TILE - 1) / RS_TILE;
  return sort_ctrl_bytes() + ggd_align((size_t)RS_MAX_PASSES * (size_t)rs_status_words(ntiles > 0 ? ntiles : 1) * sizeof(uint32_t));
}

size_t ggd_sort32_tmp_bytes(int64_t n) {
  const int64_t ntiles = (n + RS32_TILE - 1) / RS32_TILE;
  return sort_ctrl_bytes() + ggd_align((size_t)4 * (size_t)rs_status_words(ntiles > 0 ? ntiles : 1) * sizeof(uint32_t));
}

template <typename KeyT>
static int launch_sort_t(ggd_ctx* ctx, hipStream_t s, KeyT* keys_a, uint32_t* vals_a, KeyT* keys_b, uint32_t* vals_b,
                         int64_t n, int nbits, void* tmp, size_t tmp_bytes) {
  if (n <= 0) return GGD_OK;
  const int passes = sort_passes(nbits);
  if (passes > RS_MAX_PASSES) return ggd_fail(ctx, GGD_E_INVALID, "sort: too many key bits");
  if (tmp_bytes < ggd_sort_tmp_bytes(n)) return ggd_fail(ctx, GGD_E_INVALID, "sort tmp too small");
  const int ntiles = (int)((n + RS_TILE - 1) / RS_TILE);
  uint32_t* ghist = static_cast<uint32_t*>(tmp);
  uint32_t* tickets = ghist + RS_HWORDS;
  uint32_t* status = reinterpret_cast<uint32_t*>(static_cast<char*>(tmp) + sort_ctrl_bytes());
  const size_t pass_words = (size_t)rs_status_words(ntiles);
  const int gshift = rs_gshift(ntiles);
  const size_t status_bytes = (size_t)passes * pass_words * sizeof(uint32_t);
  GGD_HIP(hipMemsetAsync(tmp, 0, sort_ctrl_bytes() + status_bytes, s));   // histograms, tickets, status words
  KeyT* kin = (passes & 1) ? keys_b : keys_a;
  uint32_t* vin = (passes & 1) ? vals_b : vals_a;
  KeyT* kout = (passes & 1) ? keys_a : keys_b;
  uint32_t* vout = (passes & 1) ? vals_a : vals_b;
  hipLaunchKernelGGL((sort_global_hist_kernel<KeyT, RS_ITEMS>), dim3(ntiles), dim3(RS_THREADS), 0, s, kin, n, passes,
                     ghist);
  for (int p = 0; p < passes; ++p) {
    hipLaunchKernelGGL((sort_onesweep_kernel<KeyT, false, RS_ITEMS>), dim3(ntiles), dim3(RS_THREADS), 0, s, kin, vin, kout, vout, n,
                       8 * p, ghist + p * RS_BINS, status + (size_t)p * pass_words, ntiles, gshift, tickets + p);
    KeyT* tk = kin; kin = kout; kout = tk;
    uint32_t* tv = vin; vin = vout; vout = tv;
  }
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

const uint32_t* ggd_sort32_nvalid_ptr(const void* tmp) {
  return static_cast<const uint32_t*>(tmp) + RS_HWORDS + RS_MAX_PASSES;
}

const uint32_t* ggd_sort32_flat_ptr(const void* ctl) {
  return static_cast<const uint32_t*>(ctl) + RS_HWORDS + RS_MAX_PASSES + 1;
}

size_t ggd_sort_ctrl_words() { return sort_ctrl_bytes() / sizeof(uint32_t); }

const uint32_t* ggd_fold_nvalid_ptr(const uint32_t* fold_ctl) { return fold_ctl + GGD_FOLD_REPS * GGD_FOLD_REP_STRIDE + RS_MAX_PASSES; }
const uint32_t* ggd_fold_flat_ptr(const uint32_t* fold_ctl) { return fold_ctl + GGD_FOLD_REPS * GGD_FOLD_REP_STRIDE + RS_MAX_PASSES + 1; }
size_t ggd_fold_l1_offset(int64_t P) {
  const int64_t ntiles = (P + RS32_TILE - 1) / RS32_TILE;
  return (size_t)GGD_FOLD_HEAD + (size_t)4 * (size_t)rs_status_words(ntiles > 0 ? ntiles : 1);
}
size_t ggd_fold_ctl_words(int64_t P) {   // + level-1 binning: one 64-word row per 1024-Gaussian chunk and per group of chunks
  const int64_t chunks = (P + 1023) / 1024;
  return ggd_fold_l1_offset(P) + (size_t)(chunks + ((int64_t)1 << rs_gshift(chunks > 0 ? chunks : 1)) + 2) * 64;
}

int ggd_launch_sort32_iota(ggd_ctx* ctx, hipStream_t s, const uint32_t* keys_src, uint32_t* keys_a, uint32_t* vals_a,
                           uint32_t* keys_b, uint32_t* vals_b, int64_t n, int nbits, void* tmp, size_t tmp_bytes,
                           uint32_t* clean_ctl, const ggd_scan_piggy* piggy, bool flag_flat_last, bool apply_here,
                           const ggd_fold* fold, bool skip_last) {
  if (n <= 0) return GGD_OK;
  const int passes = sort_passes(nbits);
  if (passes > RS_MAX_PASSES || (passes & 1)) return ggd_fail(ctx, GGD_E_INVALID, "sort32: need an even pass count");
  if (fold && (passes != 4 || !piggy || !piggy->wg_info)) return ggd_fail(ctx, GGD_E_INVALID, "sort32: folded front end needs 4 passes and the workgroup sums");
  if (!fold && tmp_bytes < ggd_sort32_tmp_bytes(n)) return ggd_fail(ctx, GGD_E_INVALID, "sort tmp too small");
  const int ntiles = (int)((n + RS32_TILE - 1) / RS32_TILE);
  // control block (histograms, tickets, n_valid): `clean_ctl` = a block an earlier kernel of this frame has already
  // cleared (then the status words are cleared by the histogram kernel and the sort needs no memset launch), else the
  // head of tmp.  fold: a whole block (status words included) the previous frame's preprocess cleared, histograms (in
  // GGD_FOLD_REPS replicas) filled by this frame's.
  uint32_t* ghist = fold ? fold->ctl : (clean_ctl ? clean_ctl : static_cast<uint32_t*>(tmp));
  uint32_t* tickets = ghist + (fold ? GGD_FOLD_REPS * GGD_FOLD_REP_STRIDE : RS_HWORDS);
  uint32_t* status = fold ? fold->ctl + GGD_FOLD_HEAD : reinterpret_cast<uint32_t*>(static_cast<char*>(tmp) + sort_ctrl_bytes());
  const int reps = fold ? GGD_FOLD_REPS : 1;
  const size_t pass_words = (size_t)rs_status_words(ntiles);
  const int gshift = rs_gshift(ntiles);
  const size_t status_bytes = (size_t)passes * pass_words * sizeof(uint32_t);
  if (!clean_ctl && !fold) GGD_HIP(hipMemsetAsync(tmp, 0, sort_ctrl_bytes() + status_bytes, s));
  // keys equal to 0xFFFFFFFF (culled Gaussians) are dropped by pass 0; n_valid (device) = number of kept keys, the
  // element count of every later pass and of the binning that consumes the order (word RS_MAX_PASSES of the tickets)
  uint32_t* n_valid = tickets + RS_MAX_PASSES;
  // the offsets scan rides on the first three launches as appended workgroups (reduce | block sums | apply)
  ggd_scan_piggy pg = piggy ? *piggy : ggd_scan_piggy{};
  const int pnb = piggy ? pg.nb : 0;
  if (skip_last && !(fold && flag_flat_last)) return ggd_fail(ctx, GGD_E_INVALID, "sort32: skip_last needs the folded front end");
  if (fold) { pg.n_valid = n_valid; pg.fold_hist = ghist; pg.flat_flag = tickets + RS_MAX_PASSES + 1; pg.spec_flat = skip_last ? 1 : 0; }   // (pass 0 does not read n_valid: its appended workgroup writes it
                                                              // -- and sums the replicas -- for the later passes)
  if (!fold) {
    // (tile size of this launch, measured at 1 M keys: 4 / 16 / 32 / 64 keys per thread = 977 / 245 / 123 / 62 workgroups flushing
    // into the same 1024 words: 50.4 / 19.2 / 19.3 / 29.1 us -- atomic instructions on one 64-byte line serialise, ~43 ns per 16 lanes)
    const int htiles = ntiles;
    hipLaunchKernelGGL((sort_global_hist_kernel<uint32_t, RS32_ITEMS, true>), dim3(htiles + pnb), dim3(RS_THREADS), 0, s,
                       keys_src, n, passes, ghist, n_valid, clean_ctl ? status : nullptr,
                       clean_ctl ? status_bytes / sizeof(uint32_t) : (size_t)0, htiles, pg);
  }
  // every pass copies its pairs out of LDS in digit order (STAGE; measured, sort stage with / without: 1 M Gaussians 69.9 /
  // 71.1 us, 5 M 201 / 235, 5 M shell 252 / 342, 10 M 340 / 390)
  // pass 0: keys_src (read-only, caller's buffer) -> B with identity values; then B -> A -> B -> A ...
  const uint32_t* kin = keys_src;
  const uint32_t* vin = nullptr;
  for (int p = 0; p < passes - (skip_last ? 1 : 0); ++p) {
    uint32_t* kout = (p & 1) ? keys_a : keys_b;
    uint32_t* vout = (p & 1) ? vals_a : vals_b;
    if (p == 0)
      hipLaunchKernelGGL((sort_onesweep_kernel<uint32_t, true, RS32_ITEMS, true, true>), dim3(ntiles + (pnb ? 1 : 0)),
                         dim3(RS_THREADS), 0, s, kin, vin, kout, vout, n, 0, ghist, status, ntiles, gshift, tickets, n_valid,
                         2, pg, (uint32_t*)nullptr, reps);
    if (p != 0)
      hipLaunchKernelGGL((sort_onesweep_kernel<uint32_t, false, RS32_ITEMS, true, true>), dim3(ntiles + ((p == 1 && apply_here) ? pnb : 0)),
                         dim3(RS_THREADS), 0, s, kin, vin, kout, vout, n, 8 * p, ghist + p * RS_BINS,
                         status + (size_t)p * pass_words, ntiles, gshift, tickets + p, n_valid, 3, pg,
                         (flag_flat_last && p == passes - 1) ? tickets + RS_MAX_PASSES + 1 : nullptr, 1);
    kin = kout; vin = vout;
  }
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}


size_t ggd_sort32_msd_table_bytes(int64_t n) {
  const int64_t ntiles = (n + MSD_TILE - 1) / MSD_TILE;
  return ggd_align((size_t)(ntiles > 0 ? ntiles : 1) * GGD_MSD_BINS * sizeof(uint32_t));
}
bool ggd_sort32_msd_supported(int64_t n) { return n > 0 && (n + MSD_TILE - 1) / MSD_TILE <= GGD_MSD_MAX_TILES; }

// The two-launch form of ggd_launch_sort32_iota (folded front end built with fold.msd; the scan's step 2 rides on launch 1 as
// before, its step 3 is left to the caller's binning launch): result in (keys_a, vals_a); the consumers' "flat" word stays 0.
int ggd_launch_sort32_msd(ggd_ctx* ctx, hipStream_t s, const uint32_t* keys_src, uint32_t* keys_a, uint32_t* vals_a,
                          uint32_t* keys_b, uint32_t* vals_b, int64_t n, uint32_t* table, const ggd_scan_piggy* piggy,
                          const ggd_fold* fold) {
  if (n <= 0) return GGD_OK;
  if (!fold || !fold->msd || !piggy || !piggy->wg_info || !ggd_sort32_msd_supported(n))
    return ggd_fail(ctx, GGD_E_INVALID, "sort32 (two launches): needs the folded front end in its msd form");
  const int ntiles = (int)((n + MSD_TILE - 1) / MSD_TILE);
  uint32_t* tickets = fold->ctl + GGD_FOLD_REPS * GGD_FOLD_REP_STRIDE;
  if (fold->msd_shift < 0 || fold->msd_shift > GGD_MSD_MAX_SHIFT) return ggd_fail(ctx, GGD_E_INVALID, "sort32 (two launches): bucket shift out of range");
  ggd_scan_piggy pg = *piggy;
  pg.n_valid = tickets + RS_MAX_PASSES; pg.fold_hist = fold->ctl; pg.msd = 1; pg.msd_lo = fold->msd_lo; pg.msd_shift = fold->msd_shift;
  hipLaunchKernelGGL(sort_msd_partition_kernel, dim3(ntiles + 1), dim3(RS_THREADS), 0, s, keys_src, keys_b, vals_b, n, table,
                     ntiles, fold->msd_lo, fold->msd_shift, pg);
  hipLaunchKernelGGL(sort_msd_finish_kernel, dim3(GGD_MSD_BINS), dim3(MSDF_THREADS), 0, s, keys_b, vals_b, keys_a, vals_a,
                     fold->ctl, table, ntiles, fold->msd_shift);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

int ggd_launch_sort(ggd_ctx* ctx, hipStream_t s, uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b,
                    uint32_t* vals_b, int64_t n, int nbits, void* tmp, size_t tmp_bytes) {
  return launch_sort_t<uint64_t>(ctx, s, keys_a, vals_a, keys_b, vals_b, n, nbits, tmp, tmp_bytes);
}

int ggd_launch_ranges(ggd_ctx* ctx, hipStream_t s, const uint64_t* keys, int64_t n, uint32_t* ranges, int T) {
  GGD_HIP(hipMemsetAsync(ranges, 0, (size_t)T * 2 * sizeof(uint32_t), s));
  if (n <= 0) return GGD_OK;
  hipLaunchKernelGGL(ranges_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keys, n, ranges);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}
