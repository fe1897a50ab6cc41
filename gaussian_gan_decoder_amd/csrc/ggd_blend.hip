// ggd_blend.hip -- stages a9 (forward alpha blend) and a10 (backward blend) for gfx950.
//
// Replaces renderCUDA fwd/bwd inside `_C.rasterize_gaussians{,_backward}` (reached from
// gaussian_splatting/gaussian_renderer/__init__.py:167-175; algorithm restated in SURVEY.md 9.4 / 9.5).
//
// wave64 design (not the 16x16-thread block of the CUDA original):
//   * The unit of work is an 8x8 QUARTER of a 16x16 tile, one pixel per lane: forward = one single-wave workgroup per
//     quarter (the four quarters of a tile in consecutive dispatch slots of one XCD, so that they share its L2); backward =
//     four independent quarter waves (large grids) or four quarter waves in one workgroup that combine their per-record sums
//     in LDS (small grids).  Wider blocks (16x8 with 2 pixels per lane, 16x16 with 4) are kept as options: the VALU cost of a
//     blended record is proportional to the pixels a wave holds (packed fp32 brings no throughput on gfx950), while a
//     record typically reaches less than half of a block, so the smallest block culls finest and idles fewest lanes.
//   * A tile's depth-sorted list is consumed 64 records per round: lane j gathers record j (one 48-byte ggd_splat, three
//     16-byte loads).  The gather is a two-stage software pipeline -- the list entries of round k + 2 and the records of
//     round k + 1 are in flight while round k is blended.  The gathering lane tests its record against the wave's pixel
//     rectangle (axis-aligned box of {alpha >= 1/255}, then the exact maximum of `power` over the rectangle); only the
//     survivors are staged, compacted, in LDS, as loaded plus two in-place words.  The rectangle shrinks round by round
//     to the pixels that are still live (forward) / that can see the round (backward).
//   * Staged records are read back as LDS broadcasts in straight-line groups of 8.  Per record: a wave-level cull in the
//     power domain before any exp (one scalar branch), then a branch-free update in which every per-pixel condition is a
//     wave-uniform 64-bit lane mask in an SGPR pair (v_cmp -> SGPR, combined on the scalar unit) and every state change a VOP3
//     select on such a mask with a tied destination (inline asm: the compiler's VCC-form select costs 8.7 issue slots).
//   * No workgroup barrier in the forward or the quarter backward: "wave finished" is one wave-uniform ballot.
//   * Backward: the 9 per-Gaussian partial gradients are reduced over the wave with a transposed DPP butterfly folded by the
//     gfx950 lane-block swaps (no LDS round trip), parked in LDS per touched record, and flushed as (record, component) pairs
//     over the lanes -- component fastest, so a record's nine float atomics form one 36-byte span -- instead of one atomic
//     per (pixel, Gaussian, component).
#include "ggd_common.h"

namespace {

constexpr float ALPHA_FLOOR = 1.0f / 255.0f;

typedef float f2 __attribute__((ext_vector_type(2)));  // -> v_pk_{add,mul,fma}_f32: two pixels per VALU issue

// power = -1/2 (A dx^2 + C dy^2) - B dx dy for a pixel PAIR, in the project's fixed operation order (same as the
// oracle's gauss_power):  fma( fma(-A/2, dx, -B*dy), dx, ((-C/2)*dy)*dy ).  hA = -A/2; nBdy, hCdy2 are per-lane.
__device__ __forceinline__ f2 gauss_power2(float hA, float nBdy, float hCdy2, f2 dx) {
  const f2 inner = __builtin_elementwise_fma((f2){hA, hA}, dx, (f2){nBdy, nBdy});
  return __builtin_elementwise_fma(inner, dx, (f2){hCdy2, hCdy2});
}

// exp() variants for the blend (GGD_OPT_EXP_MODE).
// The bare hardware exponential, written out: ONE expression shared by the forward (mode 1, and the forward half of the default
// mode 3) and by the backward's contribution decision (bwd_update<3>), so that the two passes cannot drift apart with the way a
// ROCm release lowers __expf: v_exp_f32(fl(x * log2e)), ~3 ulp on [-6, 0].
constexpr float BLEND_LOG2E = 1.44269502162933349609375f;
__device__ __forceinline__ float blend_exp_bare(float x) { return __builtin_amdgcn_exp2f(x * BLEND_LOG2E); }

template <int MODE>
__device__ __forceinline__ float blend_exp(float x) {
  if (MODE == 1) return blend_exp_bare(x);
  if (MODE == 2) {                  // 2^(hi) * (1 + lo*ln2): hi = fl(x*log2e), lo = exact product residual + low bits
    const float t = x * 1.44269502162933349609375f;
    float lo = __builtin_fmaf(x, 1.44269502162933349609375f, -t);
    lo = __builtin_fmaf(x, 1.925963033500011e-08f, lo);
    const float e = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(e, lo * 0.693147182464599609375f, e);
  }
  return expf(x);                   // ocml, <= 1 ulp
}

// Two exps per call; MODE 2 is the same arithmetic as blend_exp<2> per component, with the multiplies / FMAs packed.
template <int MODE>
__device__ __forceinline__ f2 blend_exp2v(f2 x) {
  if (MODE == 2) {
    const f2 L2E = {1.44269502162933349609375f, 1.44269502162933349609375f};
    const f2 t = x * L2E;
    f2 lo = __builtin_elementwise_fma(x, L2E, -t);
    lo = __builtin_elementwise_fma(x, (f2){1.925963033500011e-08f, 1.925963033500011e-08f}, lo);
    const f2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    return __builtin_elementwise_fma(e, lo * 0.693147182464599609375f, e);
  }
  return (f2){blend_exp<MODE>(x.x), blend_exp<MODE>(x.y)};
}

// Record-level pre-cull, done ONCE per gathered record by the lane that gathers it (the per-pixel cull costs every lane
// of the wave ~13 issue slots per record): ggd_splat carries the half extents (ex, ey) of the axis-aligned box outside
// of which power < thr, i.e. alpha < 1/255 (formed once per Gaussian by the preprocess kernel, with the fp32 safety
// margins described there).  If that box misses the wave's pixel rectangle no pixel can pass the `power >= thr` test, so
// the record is never written to LDS.  Comparisons with +inf (indefinite conic: always kept) / -inf (opacity < 1/255:
// never kept) extents do the right thing.
__device__ __forceinline__ bool record_box_hits(float x, float y, float ex, float ey, float wx0, float wx1, float wy0,
                                                float wy1) {
  return !(x + ex < wx0 || x - ex > wx1 || y + ey < wy0 || y - ey > wy1);
}

// Second, exact stage of the pre-cull (same lane, only for records whose box hit): the largest `power` any point of the
// wave's pixel rectangle can reach.  power(d) = hA dx^2 + nB dx dy + hC dy^2 (d = centre - pixel) is concave for a
// positive-definite conic, so its maximum over the rectangle [dx0, dx1] x [dy0, dy1] is 0 if the rectangle contains the
// centre and otherwise sits on an edge facing the centre, where it is a 1-D parabola maximised at the clamped vertex.  An
// elongated, tilted splat whose bounding box clips a block's corner is dropped here instead of costing every lane of
// the wave a `power` evaluation.  Conservative: the test allows for the fp32 rounding of both evaluations (margin
// proportional to the magnitude of the three terms); non-positive-definite conics (ex = +inf) are always kept.
__device__ __forceinline__ bool record_reaches_block(float x, float y, float hA, float nB, float hC, float thr, float ex,
                                                     float wx0, float wx1, float wy0, float wy1) {
  if (!(ex < __builtin_huge_valf())) return true;
  const float dx0 = x - wx1, dx1 = x - wx0, dy0 = y - wy1, dy1 = y - wy0;
  if (dx0 <= 0.0f && dx1 >= 0.0f && dy0 <= 0.0f && dy1 >= 0.0f) return true;
  // vertex of the edge parabolas: dy* = -nB X / (2 hC), dx* = -nB Y / (2 hA).  (v_rcp_f32, 1 ulp: the correctly rounded
  // divisions the build flags ask for cost two dozen instructions per record here, and the margin below dwarfs an ulp)
  const float ihC = -0.5f * __builtin_amdgcn_rcpf(hC), ihA = -0.5f * __builtin_amdgcn_rcpf(hA);
  // A concave function grows along every segment towards its maximiser (the centre, outside the rectangle here), so its
  // maximum over the rectangle sits on an edge that FACES the centre: at most one edge per axis, the one nearer to it.
  const bool in_x = dx0 <= 0.0f && dx1 >= 0.0f, in_y = dy0 <= 0.0f && dy1 >= 0.0f;
  const float X = dx0 > 0.0f ? dx0 : dx1, Y = dy0 > 0.0f ? dy0 : dy1;
  const float dyv = fminf(fmaxf((nB * X) * ihC, dy0), dy1);
  const float t1 = hA * X * X, t2 = nB * X * dyv, t3 = hC * dyv * dyv;
  const float px_ = (t1 + t2) + t3, mx_ = (fabsf(t1) + fabsf(t2)) + fabsf(t3);
  const float dxv = fminf(fmaxf((nB * Y) * ihA, dx0), dx1);
  const float u1 = hA * dxv * dxv, u2 = nB * dxv * Y, u3 = hC * Y * Y;
  const float py_ = (u1 + u2) + u3, my_ = (fabsf(u1) + fabsf(u2)) + fabsf(u3);
  const float pmax = in_x ? py_ : (in_y ? px_ : fmaxf(px_, py_));
  const float mag = in_x ? my_ : (in_y ? mx_ : fmaxf(mx_, my_));
  return !(pmax < thr - (1e-3f + 1e-5f * mag));
}

// Workgroup -> tile placement of the blend kernels (speed only; any placement is correct).  Workgroup b runs on XCD
// b % 8 and every XCD has its own 4 MB L2: the nsub sub-blocks of one tile take consecutive slots of ONE XCD's dispatch
// stream (b = 8 q + x: slot q of XCD x), so they are resident together and all but the first gather of a record hit
// that L2 (measured, 4 waves per tile: FETCH_SIZE 178 -> 46 MB per frame, 156 -> 149 us; with the sub-blocks T
// workgroups apart the later ones re-fetched every record from the Infinity Cache).  Walking 8x8-tile patches per XCD
// on top of that halves the fetches again (23 MB) but costs time (151 us; shell 335 -> 354 us): neighbouring tiles
// finish together and leave the XCDs unevenly loaded -- not used.  Tile counts that are not multiples of 8 keep
// tile = b % T.
__device__ __forceinline__ void ggd_block_to_tile(int b, int nsub, int gx, int gy, int T, int& tile, int& sub) {
  (void)gx; (void)gy;
  if ((T & 7) == 0) { const int q = b >> 3; sub = q % nsub; tile = (q / nsub) * 8 + (b & 7); }
  else { tile = b % T; sub = b / T; }
}

// Wave <-> pixel mapping of the blend kernels: a wave owns a BW x BH pixel block of a 16x16 tile, PXL horizontally
// adjacent pixels per lane (BW / PXL lanes per row, BH = 64 * PXL / BW rows):
//   PXL = 4, BW = 16: one wave per tile;      PXL = 2, BW = 16: two waves per tile (16x8 halves);
//   PXL = 1, BW = 8 : four waves per tile (8x8 quarters).
// The VALU cost of a blended record is proportional to the pixels a wave holds (packed fp32 brings no throughput on
// gfx950), while a record typically reaches ~40 % of a 16x8 block: smaller blocks cull finer and leave fewer idle
// lanes, at the price of more list scans.
template <int PXL, int BW = 16>
struct WaveGeom {
  static constexpr int LPR = BW / PXL, BH = 64 / LPR, NSX = 16 / BW, NSUB = NSX * (16 / BH);
  int px0, py;
  uint32_t lo, hi;
  __device__ __forceinline__ WaveGeom(int b, int gx, int T, const uint32_t* __restrict__ ranges, uint32_t capacity = 0xffffffffu) {
    int tile, sub;
    ggd_block_to_tile(b, NSUB, gx, T / gx, T, tile, sub);
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    px0 = tx * 16 + (sub % NSX) * BW + (lane % LPR) * PXL;
    py = ty * 16 + (sub / NSX) * BH + lane / LPR;
    const uint2 r = reinterpret_cast<const uint2*>(ranges)[tile];
    lo = min(r.x, capacity); hi = min(r.y, capacity);  // capacity < R only in a speculative forward that is retried
  }
};

// Per-lane selects on WAVE-UNIFORM 64-bit lane masks held in SGPR pairs.  Measured on gfx950
// (scripts/probes/valu_rate_probe.hip): v_cndmask_b32 in its VOP2 form (mask in VCC) issues at 1/8.7 of the v_fma_f32
// rate, the VOP3 form with an SGPR-pair mask at 1/1.6 -- and hipcc picks the VOP2 form whenever the mask happens to
// live in VCC.  Going through these helpers pins the VOP3 form and ties destination = old value (no copies where the
// culled and the updated path of the record loop join).
__device__ __forceinline__ void sel_into(float& dst, float src, uint64_t mask) {
  asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(dst) : "v"(src), "s"(mask));
}
__device__ __forceinline__ void sel_into(uint32_t& dst, uint32_t src, uint64_t mask) {
  asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(dst) : "v"(src), "s"(mask));
}
__device__ __forceinline__ float sel_or_zero(float src, uint64_t mask) {
  float d;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(d) : "v"(src), "s"(mask));
  return d;
}
// dst = mask ? src : dst, ordered after the computation of `after` (a value that still needs the OLD dst: keeps the
// scheduler from sinking that use below the select, which would force a copy of the old value)
__device__ __forceinline__ void sel_into_after(float& dst, float src, uint64_t mask, float after) {
  asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(dst) : "v"(src), "s"(mask), "v"(after));
}
// acc += a * b in place (v_fmac_f32; through asm so that the SLP vectoriser does not pair the three colour channels
// into v_pk_fma_f32 with register shuffles -- packed fp32 has no throughput advantage on gfx950)
__device__ __forceinline__ void fma_into(float& acc, float a, float b) {
  asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
}

// Forward blend.  Per staged record:
//   (1) the `power` values of the lane's pixels (operation order == the published form),
//   (2) CULL: if no pixel of the wave can reach alpha >= 1/255 -- tested in the power domain against a per-record
//       threshold ln(1/(255*opacity)) lowered by a safety margin, so the decision is exact w.r.t. the float alpha
//       test that follows -- the whole wave skips the record before any exp,
//   (3) the exact per-pixel tests and the blend update, branch-free: every condition is a wave-uniform 64-bit lane mask
//       (v_cmp -> SGPR pair, combined on the scalar unit), every state update one select on such a mask.
// A finished pixel gets x = +inf: its power becomes -inf/NaN and it drops out in (2) with no extra instructions.
// Cost model (issue slots in units of one v_fma_f32 = ~2.5 cycles of a SIMD with >= 3 issuing waves, measured by the probes
// under scripts/probes): plain VALU 1, packed fp32 1.85 (no throughput gain over two plain ops on gfx950), v_cmp / VOP3
// select / v_min 1.6, v_exp_f32 3.1.  The forward executes 8 VALU instructions for a staged record no pixel sees and 28
// for one some pixel sees; the kernel runs at ~0.85 of the VALU rate in its steady state and spends a third of its time
// in ramp and tail.
// LDS reads through an explicit address-space-3 pointer (plain vector types: HIP's float4 class does not bind there)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const v4f lds_cf4;
typedef __attribute__((address_space(3))) const v2f lds_cf2;
__device__ __forceinline__ float4 lds_read4(lds_cf4* p) { const v4f v = *p; return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ float2 lds_read2(lds_cf4* p, int word) {
  const v2f v = *(lds_cf2*)((__attribute__((address_space(3))) const float*)p + word);
  return make_float2(v[0], v[1]);
}

template <int EXP_MODE, bool CULL, int PXL, int BW, bool STATS>
__device__ __forceinline__ void blend_forward_block(int blk, float4* s_rec, int W, int H, int gx, int T,
                                                    const ggd_splat* __restrict__ splat,
                                                    const uint32_t* __restrict__ list,
                                                    const uint32_t* __restrict__ ranges, uint32_t capacity,
                                                    const float* __restrict__ bg,
                                                    float* __restrict__ out_color,
                                                    float* __restrict__ final_T,
                                                    uint32_t* __restrict__ n_contrib,
                                                    unsigned long long* __restrict__ stats) {
#ifndef GGD_FWD_GRP
#define GGD_FWD_GRP 8
#endif
  constexpr int GRP = GGD_FWD_GRP;   // staged records per straight-line group
  const int lane = threadIdx.x;
  using Geom = WaveGeom<PXL, BW>;
  const Geom g(blk, gx, T, ranges, capacity);
  const bool row_in = g.py < H;
  float INF = __builtin_huge_valf();
  asm volatile("" : "+v"(INF));   // keep it in a VGPR (VOP3 selects take no 32-bit literal)
  uint32_t st_visited = 0, st_culled = 0, st_lanes = 0, st_pixels = 0, st_inloop = 0;  // wave-uniform debug counters (GGD stats)
  const uint64_t st_t0 = STATS ? wall_clock64() : 0ull;                // 100 MHz constant clock: the wave's residency
  float Tr[PXL], C[PXL][3];   // per pixel: transmittance, accumulated colour
  uint32_t last[PXL];
  float px[PXL];              // pixel x coordinates; +inf once the pixel is finished / outside the image
  const float pyf = (float)g.py;
#pragma unroll
  for (int k = 0; k < PXL; ++k) {
    Tr[k] = 1.0f; C[k][0] = C[k][1] = C[k][2] = 0.0f;
    last[k] = 0;
    px[k] = (row_in && (g.px0 + k) < W) ? (float)(g.px0 + k) : INF;
  }
  auto wave_alive = [&]() {
    uint64_t m = 0ull;
#pragma unroll
    for (int k = 0; k < PXL; ++k) m |= __ballot(px[k] < INF);
    return m != 0ull;
  };

  // the wave's pixel rectangle (pixel centres), for the record-level pre-cull
  const float wx0 = (float)(g.px0 - (lane % Geom::LPR) * PXL), wx1 = wx0 + (float)(BW - 1);
  const float wy0 = (float)(g.py - lane / Geom::LPR), wy1 = wy0 + (float)(Geom::BH - 1);
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  // ... shrunk, round by round, to the bounding rectangle of the pixels that are still LIVE (8x8 one-pixel-per-lane form):
  // a wave runs until its last pixel is finished, and towards the end most records reach the block but none of the few
  // pixels left -- each such record cost a whole-wave test in the blend loop instead of one lane of the pre-cull.  Exact:
  // a finished pixel ignores every record.
  float lx0 = wx0, lx1 = wx1, ly0 = wy0, ly1 = wy1;
  auto shrink_rect = [&]() {
    if constexpr (PXL == 1 && BW == 8) {
      const uint64_t live = __ballot(px[0] < INF);
      if (live != 0ull) {
        const int rmin = __builtin_ctzll(live) >> 3, rmax = (63 - __builtin_clzll(live)) >> 3;
        uint32_t m = (uint32_t)live | (uint32_t)(live >> 32);
        m |= m >> 16; m |= m >> 8; m &= 0xffu;
        const int cmin = __builtin_ctz(m), cmax = 31 - __builtin_clz(m);
        lx0 = wx0 + (float)cmin; lx1 = wx0 + (float)cmax;
        ly0 = wy0 + (float)rmin; ly1 = wy0 + (float)rmax;
      }
    }
  };

  // staged record = the 48-byte ggd_splat as loaded, two words replaced in place (three 16-byte LDS words, read back as
  // broadcasts; the cull stage needs the first two only):
  //   q0 = {x, y, -A/2, -B}   q1 = {-C/2, power threshold, opacity, r}   q2 = {g, b, contributor index (1-based list
  //   position; replaces the cull extent ex), -}
  // No component moves between the three words: forming a new 4-register tuple from parts of two loaded ones made the
  // register allocator copy those parts right behind the loads, i.e. wait for them there.
  bool keep = false;
  // The gather is a two-stage software pipeline, each stage one round (64 list entries) ahead of its consumer: the list
  // entries of round k + 2 and the 48-byte records of round k + 1 are in flight while round k is blended, and nothing is
  // waited for until the values are needed at the top of the next round.  (With the cull arithmetic inside the fetch the
  // compiler had to wait for both dependent loads -- list entry, then record -- right where they were issued: two exposed
  // memory round trips per round.)
  uint32_t id_nxt = 0;
  float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;   // x y hA nB | hC thr opacity r | g b ex ey
  // (unconditional loads from clamped positions: a lane past the end of the list re-reads the last entry -- one cache
  // line for the whole wave -- and its record is never consumed; exec-masked loads made r0..r2 loop-carried merges that the
  // register allocator split with copies right behind the loads, i.e. waits)
  const uint32_t last_pos = g.hi - 1u;   // only used when g.hi > g.lo
  auto load_id = [&](uint32_t pos) { id_nxt = list[min(pos, last_pos)]; };
  auto load_rec = [&](uint32_t) {
    const float4* p = reinterpret_cast<const float4*>(splat + id_nxt);
    r0 = p[0]; r1 = p[1]; r2 = p[2];
  };
  auto consume = [&](uint32_t pos) {   // cull decision of the record in r0..r2, then its two in-place edits
    keep = false;
    if (pos < g.hi) {
      keep = CULL ? (record_box_hits(r0.x, r0.y, r2.z, r2.w, lx0, lx1, ly0, ly1) &&
                     record_reaches_block(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r2.z, lx0, lx1, ly0, ly1)) : true;
      if (!CULL) r1.y = -__builtin_huge_valf();
      r2.z = __uint_as_float(pos - g.lo + 1u);
    }
  };
  if (g.hi <= g.lo || !wave_alive()) goto all_done;
  load_id(g.lo + lane);
  load_rec(g.lo + lane);
  load_id(g.lo + 64 + lane);
  for (uint32_t base = g.lo; base < g.hi; base += 64) {
    if (CULL) shrink_rect();
    consume(base + lane);          // the records requested one round ago
    __syncthreads();  // single-wave block: orders the previous round's LDS reads before this round's writes
    const uint64_t kept = __ballot(keep);
    if (keep) {  // compacted: only records whose box reaches this wave's pixels are staged
      const int slot = __popcll(kept & lt_mask);
      s_rec[slot * 3 + 0] = r0; s_rec[slot * 3 + 1] = r1; s_rec[slot * 3 + 2] = r2;
    }
    // the staged list is walked in groups of 8 of straight-line code: LDS addresses are immediates, there is no loop
    // counter, a record that no pixel of the wave sees costs one scalar branch, and so does the end of the list inside the
    // last group (padding that group with records nobody can see cost their cull tests: 1 % of the kernel).  The blend
    // state is updated in place by the tied-operand selects below, inside a wave-uniform `if` -- nothing is copied where
    // the culled and the updated path join.
    const int n = __popcll(kept), n8 = (n + GRP - 1) & ~(GRP - 1);
    if (STATS) {
      st_visited += min(64u, g.hi - base);
      st_culled += min(64u, g.hi - base) - (uint32_t)__popcll(kept);
    }
    load_rec(base + 64 + lane);    // next round's records (their list entries were requested one round ago)
    load_id(base + 128 + lane);    // and the list entries of the round after it
    __syncthreads();
    for (int j0 = 0; j0 < n8; j0 += GRP) {
      if (!wave_alive()) goto all_done;
      // the group's LDS address, held in a VGPR the compiler cannot rematerialise: as a wave-uniform value it lives in an
      // SGPR and every ds_read whose destination overlapped the address register re-copied it (two v_mov per record:
      // 6 % of the issue slots of an update, 11 % of an in-loop cull)
      uint32_t ga = (uint32_t)(uintptr_t)(lds_cf4*)(s_rec + j0 * 3);
      asm volatile("" : "+v"(ga));
      lds_cf4* grp = (lds_cf4*)(uintptr_t)ga;
      // ALL twelve words of a record are requested one record ahead (round 5; round 4 prefetched the six words of the cull
      // test and fetched the other six after the test had passed, 3 instructions before their first use -- an exposed LDS
      // round trip per update, and since the exact rectangle pre-cull 95 % of the staged records are updates).  Inside a
      // group only: carried across the group boundary the prefetched record is live on the loop's nine exit edges in
      // alternating register sets, which the compiler reconciles with ~100 v_mov per group.
      float4 a_nx = lds_read4(grp), b_nx = lds_read4(grp + 1), c_nx = lds_read4(grp + 2);
#pragma unroll
      for (int jj = 0; jj < GRP; ++jj) {
        if (j0 + jj >= n) break;   // (scalar compare + branch: the last group of a round is usually not full)
        const float4 a = a_nx, b4 = b_nx, c = c_nx;
        // (the record's twelfth word is not used: left dead, the register allocator hands its VGPR out as a temporary while the
        // load is still in flight, and the write-after-write hazard puts a wait for the whole prefetch three instructions
        // behind its issue)
        asm volatile("" : : "v"(c.w));
        const float2 b01 = make_float2(b4.x, b4.y);
        if (jj < GRP - 1) {
          a_nx = lds_read4(grp + (jj + 1) * 3);
          b_nx = lds_read4(grp + (jj + 1) * 3 + 1);
          c_nx = lds_read4(grp + (jj + 1) * 3 + 2);
        }
        const float dy = a.y - pyf;
        const float nBdy = a.w * dy, hCdy2 = (b01.x * dy) * dy;
        float pw[PXL];
        uint64_t need[PXL], any = 0ull;
#pragma unroll
        for (int k = 0; k < PXL; ++k) {
          const float dx = a.x - px[k];
          pw[k] = __builtin_fmaf(__builtin_fmaf(a.z, dx, nBdy), dx, hCdy2);
          need[k] = __ballot(pw[k] >= b01.y);
          any |= need[k];
        }
        if (any == 0ull) { if (STATS && j0 + jj < n) { st_culled += 1; st_inloop += 1; } continue; }
        if (STATS) {
          uint64_t lanes = 0ull;
#pragma unroll
          for (int k = 0; k < PXL; ++k) { lanes |= need[k]; st_pixels += (uint32_t)__popcll(need[k]); }
          st_lanes += (uint32_t)__popcll(lanes);
        }
        const float2 b23 = make_float2(b4.z, b4.w);                                                                   // opacity, r
        const uint32_t contributor = __float_as_uint(c.z);
#pragma unroll
        for (int k = 0; k < PXL; ++k) {
          const float G = blend_exp<EXP_MODE>(pw[k]);
          const float alpha = fminf(0.99f, G * b23.x);
          const uint64_t live = need[k] & ~__ballot(pw[k] > 0.0f) & ~__ballot(alpha < ALPHA_FLOOR);
          const float test_T = Tr[k] * (1.0f - alpha);
          const uint64_t low = __ballot(test_T < 0.0001f);
          const uint64_t upd = live & ~low, stop = live & low;
          const float w = sel_or_zero(alpha * Tr[k], upd);
          fma_into(C[k][0], b23.y, w);
          fma_into(C[k][1], c.x, w);
          fma_into(C[k][2], c.y, w);
          sel_into_after(Tr[k], test_T, upd, w);
          sel_into(last[k], contributor, upd);
          if (stop) sel_into(px[k], INF, stop);   // a pixel stops once: a scalar branch (SCC of the s_and above), not a select per update
        }
      }
    }
  }
all_done:

  if (STATS && lane == 0 && stats[GGD_STATS_MODE] != 0ull) {
    // timeline mode: this wave's [start, end] on the 100 MHz constant clock and the list entries it gathered -- plain
    // stores to its own slot (the counters' same-address atomics stretch the kernel tenfold); frames with more quarter
    // waves than the buffer has slots (beyond 32 768 tiles) record the first GGD_STATS_MAX_WAVES only
    if (blk < GGD_STATS_MAX_WAVES) {
      unsigned long long* slot = stats + GGD_STATS_HEAD + 3ull * (unsigned)blk;
      slot[0] = st_t0; slot[1] = wall_clock64(); slot[2] = ((unsigned long long)(g.hi - g.lo) << 32) | st_visited;
    }
  } else if (STATS && lane == 0) {
    atomicAdd(stats + 0, (unsigned long long)st_visited);
    atomicAdd(stats + 1, (unsigned long long)st_culled);
    atomicAdd(stats + 2, (unsigned long long)st_lanes);
    atomicAdd(stats + 3, (unsigned long long)st_pixels);
    atomicAdd(stats + 4, (unsigned long long)(g.hi - g.lo));
    atomicAdd(stats + 5, (unsigned long long)st_inloop);
  }
  if (!row_in) return;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t HW = (size_t)H * W;
  const size_t pix0 = (size_t)g.py * W + g.px0;
  if (g.px0 + PXL - 1 < W && (W & 3) == 0) {
    if constexpr (PXL == 4) {
      *reinterpret_cast<float4*>(final_T + pix0) = make_float4(Tr[0], Tr[1], Tr[2], Tr[3]);
      *reinterpret_cast<uint4*>(n_contrib + pix0) = make_uint4(last[0], last[1], last[2], last[3]);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float bgc = ch == 0 ? bg0 : (ch == 1 ? bg1 : bg2);
        *reinterpret_cast<float4*>(out_color + ch * HW + pix0) = make_float4(
            C[0][ch] + Tr[0] * bgc, C[1][ch] + Tr[1] * bgc, C[2][ch] + Tr[2] * bgc, C[3][ch] + Tr[3] * bgc);
      }
    } else if constexpr (PXL == 2) {
      *reinterpret_cast<float2*>(final_T + pix0) = make_float2(Tr[0], Tr[1]);
      *reinterpret_cast<uint2*>(n_contrib + pix0) = make_uint2(last[0], last[1]);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float bgc = ch == 0 ? bg0 : (ch == 1 ? bg1 : bg2);
        *reinterpret_cast<float2*>(out_color + ch * HW + pix0) =
            make_float2(C[0][ch] + Tr[0] * bgc, C[1][ch] + Tr[1] * bgc);
      }
    } else {
      final_T[pix0] = Tr[0];
      n_contrib[pix0] = last[0];
      out_color[pix0] = C[0][0] + Tr[0] * bg0;
      out_color[HW + pix0] = C[0][1] + Tr[0] * bg1;
      out_color[2 * HW + pix0] = C[0][2] + Tr[0] * bg2;
    }
  } else {
#pragma unroll
    for (int k = 0; k < PXL; ++k) {
      if (g.px0 + k < W) {
        final_T[pix0 + k] = Tr[k];
        n_contrib[pix0 + k] = last[k];
        out_color[pix0 + k] = C[k][0] + Tr[k] * bg0;
        out_color[HW + pix0 + k] = C[k][1] + Tr[k] * bg1;
        out_color[2 * HW + pix0 + k] = C[k][2] + Tr[k] * bg2;
      }
    }
  }
}

// One 8x8 quarter per single-wave workgroup (the launch has 4 T workgroups; workgroup b -> (XCD, tile, quarter), see
// ggd_block_to_tile).  A PERSISTENT form -- as many workgroups as the device holds, each drawing (tile, quarter) tickets
// from atomic counters, next ticket requested before the current tile is started -- was built and measured in round 4
// (commit 1e08ef1, profiles/r04/persist_ab_*.txt): 413 us with one counter per XCD, 184 us with 64 counters per XCD, against
// 104 us for this one-shot launch at 1 M / 1024^2 (shell 260 vs 194, 100 k / 512^2 79 vs 25): the same waves take the same
// time per tile, but 3.8 instead of 5.4 of them are at work per SIMD -- a ticket is a returning atomic's round trip through
// memory, the hardware dispatcher back-fills a finished wave's slot without one.  Removed again.
template <int EXP_MODE, bool CULL, int PXL, int BW, bool STATS>
__global__ __launch_bounds__(64) void blend_forward_kernel(int W, int H, int gx, int T,
                                                           const ggd_splat* __restrict__ splat,
                                                           const uint32_t* __restrict__ list,
                                                           const uint32_t* __restrict__ ranges, uint32_t capacity,
                                                           const float* __restrict__ bg,
                                                           float* __restrict__ out_color,
                                                           float* __restrict__ final_T,
                                                           uint32_t* __restrict__ n_contrib,
                                                           unsigned long long* __restrict__ stats) {
  __shared__ float4 s_rec[64 * 3];
  blend_forward_block<EXP_MODE, CULL, PXL, BW, STATS>((int)blockIdx.x, s_rec, W, H, gx, T, splat, list, ranges, capacity, bg,
                                                      out_color, final_T, n_contrib, stats);
}

// Backward blend: the per-record update of ONE pixel, shared by the two kernel forms below.  Records are visited back to
// front; the body is SELECT-FREE: a pixel that does not see the record (culled, beyond its n_contrib, alpha < 1/255) gets
// alpha = G = 0, which makes every state update an exact no-op (T / (1 - 0) = T, acc + 0 * (c - acc) = acc), so no per-pixel
// branches or selects on the recurrence state are needed.  1/(1 - alpha) is one v_rcp + one Newton step.
//   * The colour behind the record is folded in RIGHT AFTER its use: the published form keeps (last_alpha, last_color) and
//     folds one record late, accum = la * lc + (1 - la) * accum -- 17 instructions per update; with d = c - accum (needed for
//     dL/dalpha anyway) the same value is accum + alpha * d: 9 instructions and two words less state per pixel.
//   * The mean gradient leaves the pixel loop as the two sums  sum(dL/dalpha * G * dx), sum(dL/dalpha * G * dy); the conic's
//     2 x 2 matrix, the opacity, the -1/2 and the 0.5 W / 0.5 H of the pixel-to-NDC map are applied once per (record,
//     component) by the flush, not per pixel.
struct BwdPixel { float T, nTfin, bgdot, acc[3], gpx[3]; };

template <int EXP_MODE>
__device__ __forceinline__ uint64_t bwd_update(BwdPixel& st, float pw, float dx, float dy, uint64_t need, float opacity,
                                               const float (&col)[3], float (&s)[8], float& sop) {
  float g0, adec;
  if constexpr (EXP_MODE == 3) {
    // the default pairing (bare v_exp_f32 in the forward, compensated 2^x here): the forward's G is exactly the `e` this form
    // starts from, so the CONTRIBUTION DECISION alpha >= 1/255 is taken on the forward's own number -- a record within a few
    // ulp of the floor is then in or out in both passes, and the transmittance replayed by T / (1 - alpha) stays consistent
    // with the saved final_T (decided on the accurate value, the two passes could disagree on such a record, 0.4 %
    // on everything in front of it in that pixel) -- while the VALUES use the compensated exponential.
    const float t = pw * BLEND_LOG2E;
    float lo = __builtin_fmaf(pw, BLEND_LOG2E, -t);
    lo = __builtin_fmaf(pw, 1.925963033500011e-08f, lo);
    const float e = blend_exp_bare(pw);          // == the forward's G, by construction (the compiler shares the product t)
    adec = fminf(0.99f, e * opacity);
    g0 = __builtin_fmaf(e, lo * 0.693147182464599609375f, e);
  } else {
    g0 = blend_exp<EXP_MODE>(pw);
  }
  const float a0 = fminf(0.99f, opacity * g0);
  if constexpr (EXP_MODE != 3) adec = a0;
  const uint64_t live = need & ~__ballot(pw > 0.0f) & ~__ballot(adec < ALPHA_FLOOR);
  const float G = sel_or_zero(g0, live), alpha = sel_or_zero(a0, live);
  const float om = 1.0f - alpha;
  float inv = __builtin_amdgcn_rcpf(om);
  inv = __builtin_fmaf(inv, __builtin_fmaf(-om, inv, 1.0f), inv);
  st.T = st.T * inv;
  const float dchannel_dcolor = alpha * st.T;
  float dL_dalpha = 0.0f;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float d = col[ch] - st.acc[ch];
    dL_dalpha = __builtin_fmaf(d, st.gpx[ch], dL_dalpha);
    fma_into(st.acc[ch], alpha, d);                       // in place: the culled path carries no copy
    s[ch] = dchannel_dcolor * st.gpx[ch];
  }
  dL_dalpha = __builtin_fmaf(st.nTfin * inv, st.bgdot, dL_dalpha * st.T);
  const float wx = (G * dx) * dL_dalpha, wy = (G * dy) * dL_dalpha;
  s[3] = wx * dx;
  s[4] = wx * dy;
  s[5] = wy * dy;
  s[6] = wx;
  s[7] = wy;
  sop = G * dL_dalpha;
  return live;
}

// one (record, component) of the flush: v = the component's reduced sum (for the two mean components the pair of sums),
// comp in accumulator-record order (conic A B C | opacity | mean x y | colour r g b)
__device__ __forceinline__ float bwd_scale(int comp, float v, float swx, float swy, float hA, float nB, float hC, float op,
                                           float ddelx_dx, float ddely_dy) {
  // d alpha / d G = opacity; d G / d conic = -1/2 G d d^T; d G / d mean = -G * conic * d, conic = (-2 hA, -nB, -2 hC)
  if (comp < 3) return (-0.5f * op) * v;
  if (comp == 4) return (op * ddelx_dx) * __builtin_fmaf(2.0f * hA, swx, nB * swy);
  if (comp == 5) return (op * ddely_dy) * __builtin_fmaf(2.0f * hC, swy, nB * swx);
  return v;
}

// Nine wave sums with no LDS round trip: the eight of wave_reduce8_transposed plus a ninth that takes four all-lane DPP
// steps inside its row, then rides in the redundant lanes of the first register (after the quad steps the four lanes of
// a bank hold the same value; lane 4b + 1 of every bank is replaced by the ninth value's row sum).  The four rows are
// folded with the gfx950 lane-block swaps: v_permlane32_swap(x0, x1) leaves {x0 rows 0,1 | x1 rows 0,1} and
// {x0 rows 2,3 | x1 rows 2,3}, whose sum holds x0's values in lanes 0-31 and x1's in lanes 32-63 (two rows each);
// v_permlane16_swap of that sum with a copy of itself pairs the remaining two rows.  Result per lane l:
//   (l & 3) != 1:  value 4 * (l >> 5) + {0, 2, 1, 3}[(l >> 2) & 3]     l < 32 and (l & 3) == 1:  the ninth value
__device__ __forceinline__ float wave_reduce9_swap(float (&v)[8], float ninth, uint64_t ninth_lanes /* 0x2222... */) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %2, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %4, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %6, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %4, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %8, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %4, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e64 %0, %0, %8, %9\n\t"
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %4\n\t"
      "s_nop 1\n\t"
      "v_add_f32_e32 %0, %0, %4\n\t"
      "v_mov_b32_e32 %4, %0\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %0, %4\n\t"
      "s_nop 1\n\t"
      "v_add_f32_e32 %0, %0, %4"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(ninth)
      : "s"(ninth_lanes));
  return v[0];
}

// ---- backward blend, tile form: the four 8x8 quarter waves of a tile in ONE workgroup --------------------------------
// (~60 VGPRs -> 8 waves per SIMD.)  Per round of 64 list entries every wave gathers the records itself (the other waves'
// copies hit L2), pre-culls them against its own rectangle and stages the survivors; the NEXT round's gather is issued
// before the current round is blended.  Staged records are walked back to front in groups of 8 as straight-line code (list
// padded with never-visible records, see the forward).  Per-record sums: transposed butterfly, parked in LDS per wave,
// combined over the waves by the flush -> one float atomic per (tile, Gaussian, component).
template <int EXP_MODE, bool CULL>
__global__ __launch_bounds__(256) void blend_backward_tile_kernel(
    int W, int H, int gx, int gy, const ggd_splat* __restrict__ splat, const uint32_t* __restrict__ list,
    const uint32_t* __restrict__ ranges, const float* __restrict__ bg, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix, float* __restrict__ grad_acc) {
  constexpr int NW = 4;
  __shared__ float4 s_rec[NW][64 * 3];
  // per-round results are double-buffered: a round's flush reads buffer `par` while early waves already fill the other
  // one, so a round costs ONE workgroup barrier (everybody finished the round), not two
  __shared__ float s_sum[2][NW][64][9];
  __shared__ uint32_t s_id[2][64];
  __shared__ float4 s_cop[2][64];      // hA, nB, hC, opacity of the round's records (the flush applies them)
  __shared__ uint32_t s_touch[2][NW][2];
  __shared__ uint32_t s_maxn[NW];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int tile, sub_unused;
  ggd_block_to_tile((int)blockIdx.x, 1, gx, gy, gx * gy, tile, sub_unused);
  const int tx = tile % gx, ty = tile / gx;
  const int qx = wv & 1, qy = wv >> 1;
  const int px0 = tx * 16 + qx * 8 + (lane & 7), py = ty * 16 + qy * 8 + (lane >> 3);
  const uint2 rg = reinterpret_cast<const uint2*>(ranges)[tile];
  const bool in = py < H && px0 < W;
  const size_t HW = (size_t)H * W;
  const size_t pix0 = (size_t)py * W + px0;

  BwdPixel st;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const float pxf = (float)px0, pyf = (float)py;
  const float tf = in ? final_T[pix0] : 0.0f;
  const uint32_t lastn = in ? n_contrib[pix0] : 0u;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) { st.gpx[ch] = in ? dL_dpix[ch * HW + pix0] : 0.0f; st.acc[ch] = 0.0f; }
  st.bgdot = (bg0 * st.gpx[0] + bg1 * st.gpx[1]) + bg2 * st.gpx[2];
  st.T = tf; st.nTfin = -tf;
  uint32_t maxn = lastn;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) maxn = max(maxn, (uint32_t)__shfl_xor((int)maxn, d, 64));
  if (lane == 0) s_maxn[wv] = maxn;
  __syncthreads();
  maxn = s_maxn[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) maxn = max(maxn, s_maxn[w]);   // workgroup-uniform: positions anyone in the tile contributed to
  if (maxn == 0) return;
  const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
  const float wx0 = (float)(tx * 16 + qx * 8), wy0 = (float)(ty * 16 + qy * 8);   // this wave's pixel rectangle
  const float wx1 = wx0 + 7.0f, wy1 = wy0 + 7.0f;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  float4* rec = s_rec[wv];
  const bool is_writer = (lane & 19) == 0 || lane == 1;
  // reduction output index (colour r g b | conic A B C | mean sums x y | opacity) -> slot in the accumulator record's order
  const int writer_val = lane == 1 ? 8 : 4 * (lane >> 5) + (((lane >> 2) & 1) << 1) + ((lane >> 3) & 1);
  const int writer_comp = writer_val < 3 ? GGD_ACC_COLOR + writer_val
                        : (writer_val < 6 ? GGD_ACC_CONIC + (writer_val - 3)
                        : (writer_val < 8 ? GGD_ACC_MEAN2D + (writer_val - 6) : GGD_ACC_OPACITY));

  // staged = the record as loaded with three words replaced in place (no component changes its 16-byte word, see the forward):
  //   {x, y, hA, nB} {hC, power threshold, opacity, 0-based list position} {g, b, r, index inside the round}
  bool keep = false;
  // Two-stage software pipeline of the gather, as in the forward: while round k is blended the records of round k + 1
  // (the round in FRONT of it: the list is walked back to front) and the list entries of round k + 2 are in flight.
  uint32_t id_cur = 0, id_nxt = 0;     // Gaussian ids of the round whose records are in r0..r2 / of the round after it
  float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;    // x y hA nB | hC thr opacity r | g b ex ey
  auto round_start = [&](uint32_t ce) { return (ce - rg.x > 64u) ? ce - 64u : rg.x; };   // ce > rg.x
  // (unconditional loads from clamped positions, see the forward; maxn > 0 here, so the list is not empty)
  auto load_id = [&](uint32_t ce) {    // the round that ends at list position ce (exclusive); ce <= rg.x: a dummy re-read
    const uint32_t cs = ce > rg.x ? round_start(ce) : rg.x;
    id_nxt = list[min(cs + (uint32_t)lane, rg.x + maxn - 1u)];
  };
  auto load_rec = [&](uint32_t) {
    id_cur = id_nxt;
    const float4* p = reinterpret_cast<const float4*>(splat + id_nxt);
    r0 = p[0]; r1 = p[1]; r2 = p[2];
  };
  auto consume = [&](uint32_t ce) {
    keep = false;
    const uint32_t cs = round_start(ce);
    if ((uint32_t)lane < ce - cs) {
      keep = CULL ? (record_box_hits(r0.x, r0.y, r2.z, r2.w, wx0, wx1, wy0, wy1) &&
                     record_reaches_block(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r2.z, wx0, wx1, wy0, wy1)) : true;
      if (!CULL) r1.y = -__builtin_huge_valf();
      r2.z = r1.w;                                                   // r joins g, b
      r1.w = __uint_as_float((cs - rg.x) + (uint32_t)lane);          // 0-based list position
      r2.w = __uint_as_float((uint32_t)lane);                        // index inside the round
    }
  };

  uint32_t cend = rg.x + maxn;  // one past the last position that matters
  load_id(cend);
  load_rec(cend);
  load_id(round_start(cend));
  int par = 0;
  while (cend > rg.x) {
    const uint32_t cstart = round_start(cend);
    const int n = (int)(cend - cstart);
    consume(cend);                                       // the records requested one round ago
    const uint64_t kept = __ballot(keep);
    const int nk = __popcll(kept), n8 = (nk + 7) & ~7;
    if (keep) {  // compacted, order preserved
      const int slot = __popcll(kept & lt_mask);
      rec[slot * 3 + 0] = r0; rec[slot * 3 + 1] = r1; rec[slot * 3 + 2] = r2;
    }
    if (lane >= nk && lane < n8) {   // padding: a record nobody sees
      rec[lane * 3 + 0] = make_float4(0, 0, 0, 0);
      rec[lane * 3 + 1] = make_float4(0, __builtin_huge_valf(), 0, 0);
    }
    if (wv == 0 && lane < n) { s_id[par][lane] = id_cur; s_cop[par][lane] = make_float4(r0.z, r0.w, r1.x, r1.z); }
    load_rec(cstart);                                    // next round's records
    load_id(cstart > rg.x ? round_start(cstart) : rg.x); // and the list entries of the round after it
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    uint64_t touched = 0ull;
    for (int j0 = n8 - 8; j0 >= 0; j0 -= 8) {
      const float4* grp = rec + j0 * 3;
#pragma unroll
      for (int jj = 7; jj >= 0; --jj) {
        const float4 a = grp[jj * 3 + 0], b = grp[jj * 3 + 1];
        const float dy = a.y - pyf;
        const float nBdy = a.w * dy, hCdy2 = (b.x * dy) * dy;
        const uint32_t pos0 = __float_as_uint(b.w);
        const float dx = a.x - pxf;
        const float pw = __builtin_fmaf(__builtin_fmaf(a.z, dx, nBdy), dx, hCdy2);
        const uint64_t need = __ballot(pos0 < lastn) & __ballot(pw >= b.y);   // two compares into SGPR pairs + s_and
        if (need == 0ull) continue;
        const float4 c = grp[jj * 3 + 2];                            // g, b, r, index inside the round
        const float col[3] = {c.z, c.x, c.y};
        float s[8], sop;                                             // colour r g b | conic A B C | mean sums x y ; opacity
        const uint64_t live = bwd_update<EXP_MODE>(st, pw, dx, dy, need, b.z, col, s, sop);
        if (live != 0ull) {   // wave-uniform: somebody in this wave saw the Gaussian
          const uint32_t ridx = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(c.w));
          touched |= 1ull << ridx;
          const float tot = wave_reduce9_swap(s, sop, 0x2222222222222222ull);
          // writers: lanes 0,4,8,12 | 32,36,40,44 (component from the table above), lane 1 the opacity sum
          if (is_writer) s_sum[par][wv][ridx][writer_comp] = tot;
        }
      }
    }
    if (lane == 0) { s_touch[par][wv][0] = (uint32_t)touched; s_touch[par][wv][1] = (uint32_t)(touched >> 32); }
    __syncthreads();
    {
      // flush: the round's (record, component) sums over the workgroup's threads, component fastest -- s_sum keeps the
      // accumulator record's order (conic A B C | opacity | mean x y | colour r g b), so consecutive lanes add to
      // consecutive floats of one 48-byte record and the atomics of a record reach L2 as one or two requests
      // instead of nine (a lane per record and one component per instruction was bound by the L2 atomic units).
      // All waves' sums are combined and the per-record constants applied here (bwd_scale).
      uint64_t tw[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) tw[w] = (uint64_t)s_touch[par][w][0] | ((uint64_t)s_touch[par][w][1] << 32);
      for (int t = threadIdx.x; t < 64 * 9; t += 64 * NW) {
        const int r = t / 9, slot = t - 9 * r;
        if (r >= n) break;
        float v = 0.0f, swx = 0.0f, swy = 0.0f;
        bool hany = false;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const bool h = (tw[w] >> r) & 1ull;
          hany = hany || h;
          v += h ? s_sum[par][w][r][slot] : 0.0f;
          swx += h ? s_sum[par][w][r][GGD_ACC_MEAN2D] : 0.0f;
          swy += h ? s_sum[par][w][r][GGD_ACC_MEAN2D + 1] : 0.0f;
        }
        if (!hany) continue;
        const float4 co = s_cop[par][r];
        atomicAdd(grad_acc + GGD_ACC_FLOATS * (size_t)s_id[par][r] + slot,
                  bwd_scale(slot, v, swx, swy, co.x, co.y, co.z, co.w, ddelx_dx, ddely_dy));
      }
    }
    cend = cstart;
    par ^= 1;
  }
}


// ---- backward blend, quarter form: every 8x8 quarter of a tile is an INDEPENDENT single-wave workgroup --------------------
// No workgroup barrier and no cross-wave combine: a wave walks the tile's list back to front from ITS OWN last contributor,
// parks the per-record sums of a round in LDS in processing order (row = 9 sums + the record's slot in the staging area,
// where its id, opacity and conic still are), and flushes them itself -- (record, component) pairs over the lanes, component
// fastest, so a record's nine atomics form one 36-byte span -- at the top of the NEXT round, before that round's records
// overwrite the staging area and before its prefetch loads are issued: a wait for the loads then never waits for younger
// atomics (loads and atomics share one in-order counter).  Compared with the tile form a Gaussian that touches k quarters of
// a tile costs k coalesced atomic spans instead of one; in exchange the four quarters never wait for each other (the tile
// form spent 13 % / 28 % of its time in the per-round lock step: cube / shell), and the four quarters of a tile are placed in
// consecutive dispatch slots of one XCD like the forward's.
// STATS (ggd_blend_stats, debug): per-wave work counters added to stats[GGD_STATS_BWD ..] at the wave's end -- list entries the
// wave walks (its share of the tile's list up to its last contributor), records staged after the pre-cull, staged records some
// pixel still needed (`need`), records at least one pixel actually blended (`live`: a 9-sum reduction + a parked row each),
// the live lanes of those, rows flushed (= 36-byte atomic spans), gather rounds.
template <int EXP_MODE, bool CULL, bool STATS = false>
__global__ __launch_bounds__(64) void blend_backward_quarter_kernel(
    int W, int H, int gx, int gy, const ggd_splat* __restrict__ splat, const uint32_t* __restrict__ list,
    const uint32_t* __restrict__ ranges, const float* __restrict__ bg, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix, float* __restrict__ grad_acc,
    unsigned long long* __restrict__ stats = nullptr) {
  uint32_t st_staged = 0, st_need = 0, st_live = 0, st_lanes = 0, st_spans = 0, st_rounds = 0;
  __shared__ float4 s_rec[64 * 3];
  // [touched record, in processing order][9 sums | staging slot] (5632 B of LDS per wave with s_rec); ONE buffer: a round's
  // rows are flushed at the top of the next round, before that round's first row is written (LDS operations of a wave
  // execute in order)
  __shared__ float s_sum[64][10];
  const int lane = threadIdx.x;
  int tile, sub;
  ggd_block_to_tile((int)blockIdx.x, 4, gx, gy, gx * gy, tile, sub);
  const int tx = tile % gx, ty = tile / gx;
  const int qx = sub & 1, qy = sub >> 1;
  const int px0 = tx * 16 + qx * 8 + (lane & 7), py = ty * 16 + qy * 8 + (lane >> 3);
  const uint2 rg = reinterpret_cast<const uint2*>(ranges)[tile];
  const bool in = py < H && px0 < W;
  const size_t HW = (size_t)H * W;
  const size_t pix0 = (size_t)py * W + px0;

  BwdPixel st;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const float pxf = (float)px0, pyf = (float)py;
  const float tf = in ? final_T[pix0] : 0.0f;
  const uint32_t lastn = in ? n_contrib[pix0] : 0u;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) { st.gpx[ch] = in ? dL_dpix[ch * HW + pix0] : 0.0f; st.acc[ch] = 0.0f; }
  st.bgdot = (bg0 * st.gpx[0] + bg1 * st.gpx[1]) + bg2 * st.gpx[2];
  st.T = tf; st.nTfin = -tf;
  uint32_t maxn = lastn;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) maxn = max(maxn, (uint32_t)__shfl_xor((int)maxn, d, 64));
  maxn = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxn);
  if (maxn == 0) return;
  const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
  const float wx0 = (float)(tx * 16 + qx * 8), wy0 = (float)(ty * 16 + qy * 8);   // this wave's pixel rectangle
  const float wx1 = wx0 + 7.0f, wy1 = wy0 + 7.0f;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  // lanes that hold a result of wave_reduce9_swap, and where it goes in the row (accumulator-record order: conic A B C |
  // opacity | mean sums x y | colour r g b); lane 2 adds the record's staging slot to the same LDS store
  const bool is_writer = (lane & 19) == 0 || lane == 1;
  const int writer_val = lane == 1 ? 8 : 4 * (lane >> 5) + (((lane >> 2) & 1) << 1) + ((lane >> 3) & 1);
  const int writer_comp = writer_val < 3 ? GGD_ACC_COLOR + writer_val
                        : (writer_val < 6 ? GGD_ACC_CONIC + (writer_val - 3)
                        : (writer_val < 8 ? GGD_ACC_MEAN2D + (writer_val - 6) : GGD_ACC_OPACITY));
  const bool stores = is_writer || lane == 2;
  const int store_col = is_writer ? writer_comp : 9;

  // staged = the record as loaded with three words replaced in place:
  //   {x, y, hA, nB} {hC, power threshold, opacity, 0-based list position} {g, b, r, Gaussian id}
  bool keep = false;
  uint32_t id_cur = 0, id_nxt = 0;
  float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;    // x y hA nB | hC thr opacity r | g b ex ey
  const uint32_t last_pos = rg.x + maxn - 1u;
  auto round_start = [&](uint32_t ce) { return (ce - rg.x > 64u) ? ce - 64u : rg.x; };   // ce > rg.x
  auto load_id = [&](uint32_t ce) {
    const uint32_t cs = ce > rg.x ? round_start(ce) : rg.x;
    id_nxt = list[min(cs + (uint32_t)lane, last_pos)];
  };
  auto load_rec = [&]() {
    id_cur = id_nxt;
    const float4* p = reinterpret_cast<const float4*>(splat + id_nxt);
    r0 = p[0]; r1 = p[1]; r2 = p[2];
  };
  // the pre-cull rectangle of a round = the bounding rectangle of the pixels that can see ANY record of the round (those
  // whose last contributor lies at or behind the round's first position): walking back to front a wave starts at its
  // deepest pixel, and until the others join, most records only reach pixels that are not live yet
  float lx0 = wx0, lx1 = wx1, ly0 = wy0, ly1 = wy1;
  auto shrink_rect = [&](uint32_t first_pos) {   // 0-based list position of the round's first record
    const uint64_t live = __ballot(lastn > first_pos);
    if (live != 0ull) {
      const int rmin = __builtin_ctzll(live) >> 3, rmax = (63 - __builtin_clzll(live)) >> 3;
      uint32_t m = (uint32_t)live | (uint32_t)(live >> 32);
      m |= m >> 16; m |= m >> 8; m &= 0xffu;
      const int cmin = __builtin_ctz(m), cmax = 31 - __builtin_clz(m);
      lx0 = wx0 + (float)cmin; lx1 = wx0 + (float)cmax;
      ly0 = wy0 + (float)rmin; ly1 = wy0 + (float)rmax;
    }
  };
  auto consume = [&](uint32_t ce) {
    keep = false;
    const uint32_t cs = round_start(ce);
    if (CULL) shrink_rect(cs - rg.x);
    if ((uint32_t)lane < ce - cs) {
      keep = CULL ? (record_box_hits(r0.x, r0.y, r2.z, r2.w, lx0, lx1, ly0, ly1) &&
                     record_reaches_block(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r2.z, lx0, lx1, ly0, ly1)) : true;
      if (!CULL) r1.y = -__builtin_huge_valf();
      r2.z = r1.w;
      r1.w = __uint_as_float((cs - rg.x) + (uint32_t)lane);
      r2.w = __uint_as_float(id_cur);
    }
  };
  // the flush of one round's parked sums (cnt rows), reading the round's records where they were staged
  auto flush = [&](int cnt) {
    const float* rows = &s_sum[0][0];
    for (int p = lane; p < cnt * 9; p += 64) {
      const int r = p / 9, comp = p - 9 * r;
      const float v = rows[r * 10 + comp];
      const float swx = rows[r * 10 + GGD_ACC_MEAN2D], swy = rows[r * 10 + GGD_ACC_MEAN2D + 1];
      const int slot = (int)__float_as_uint(rows[r * 10 + 9]);
      const float4 a = s_rec[slot * 3 + 0], b = s_rec[slot * 3 + 1];
      const uint32_t id = __float_as_uint(s_rec[slot * 3 + 2].w);
      atomicAdd(grad_acc + GGD_ACC_FLOATS * (size_t)id + comp,
                bwd_scale(comp, v, swx, swy, a.z, a.w, b.x, b.z, ddelx_dx, ddely_dy));
    }
  };

  uint32_t cend = rg.x + maxn;  // one past the last position this quarter needs
  load_id(cend);
  load_rec();
  load_id(round_start(cend));
  int prev_cnt = 0;
  while (cend > rg.x) {
    const uint32_t cstart = round_start(cend);
    consume(cend);                                       // the records requested one round ago
    __builtin_amdgcn_wave_barrier();                     // (the previous round's LDS reads are done: in-order per wave)
    flush(prev_cnt);                                     // the previous round's sums: BEFORE its records are overwritten and
    __builtin_amdgcn_wave_barrier();                     // before the new loads are issued
    const uint64_t kept = __ballot(keep);
    const int nk = __popcll(kept), n8 = (nk + 7) & ~7;
    if (STATS) { st_staged += (uint32_t)nk; st_rounds += 1; st_spans += (uint32_t)prev_cnt; }
    if (keep) {  // compacted, order preserved
      const int slot = __popcll(kept & lt_mask);
      s_rec[slot * 3 + 0] = r0; s_rec[slot * 3 + 1] = r1; s_rec[slot * 3 + 2] = r2;
    }
    if (lane >= nk && lane < n8) {   // padding: a record nobody sees
      s_rec[lane * 3 + 0] = make_float4(0, 0, 0, 0);
      s_rec[lane * 3 + 1] = make_float4(0, __builtin_huge_valf(), 0, 0);
    }
    load_rec();                                          // next round's records
    load_id(cstart > rg.x ? round_start(cstart) : rg.x); // and the list entries of the round after it
    __builtin_amdgcn_wave_barrier();
    int cnt = 0;
    float* rows = &s_sum[0][0];
    for (int j0 = n8 - 8; j0 >= 0; j0 -= 8) {
      uint32_t ga = (uint32_t)(uintptr_t)(lds_cf4*)(s_rec + j0 * 3);   // see blend_forward_kernel
      asm volatile("" : "+v"(ga));
      lds_cf4* grp = (lds_cf4*)(uintptr_t)ga;
      // (all twelve words one record ahead, inside the group: see the forward)
      float4 a_nx = lds_read4(grp + 7 * 3), b_nx = lds_read4(grp + 7 * 3 + 1), c_nx = lds_read4(grp + 7 * 3 + 2);
#pragma unroll
      for (int jj = 7; jj >= 0; --jj) {
        const float4 a = a_nx, b = b_nx, c4 = c_nx;
        asm volatile("" : : "v"(c4.w));   // (keeps the unused twelfth word's VGPR from being handed out while the load is in flight)
        if (jj > 0) {
          a_nx = lds_read4(grp + (jj - 1) * 3); b_nx = lds_read4(grp + (jj - 1) * 3 + 1); c_nx = lds_read4(grp + (jj - 1) * 3 + 2);
        }
        const float dy = a.y - pyf;
        const float nBdy = a.w * dy, hCdy2 = (b.x * dy) * dy;
        const uint32_t pos0 = __float_as_uint(b.w);
        const float dx = a.x - pxf;
        const float pw = __builtin_fmaf(__builtin_fmaf(a.z, dx, nBdy), dx, hCdy2);
        const uint64_t need = __ballot(pos0 < lastn) & __ballot(pw >= b.y);
        if (need == 0ull) continue;
        if (STATS) st_need += 1;
        const float col[3] = {c4.z, c4.x, c4.y};                     // r | g, b
        float s[8], sop;                                             // colour r g b | conic A B C | mean sums x y ; opacity
        const uint64_t live = bwd_update<EXP_MODE>(st, pw, dx, dy, need, b.z, col, s, sop);
        if (live != 0ull) {   // wave-uniform: somebody in this wave saw the Gaussian
          if (STATS) { st_live += 1; st_lanes += (uint32_t)__popcll(live); }
          const float tot = wave_reduce9_swap(s, sop, 0x2222222222222222ull);
          // writers: lanes 0,4,8,12 | 32,36,40,44 (component from the table above), lane 1 the opacity sum; lane 2 the
          // record's slot in the staging area
          const float v = is_writer ? tot : __uint_as_float((uint32_t)(j0 + jj));
          if (stores) rows[cnt * 10 + store_col] = v;
          ++cnt;
        }
      }
    }
    prev_cnt = cnt;
    cend = cstart;
  }
  __builtin_amdgcn_wave_barrier();
  flush(prev_cnt);
  if (STATS && lane == 0 && stats) {
    unsigned long long* o = stats + GGD_STATS_BWD;
    atomicAdd(o + 0, (unsigned long long)maxn);
    atomicAdd(o + 1, (unsigned long long)st_staged);
    atomicAdd(o + 2, (unsigned long long)st_need);
    atomicAdd(o + 3, (unsigned long long)st_live);
    atomicAdd(o + 4, (unsigned long long)st_lanes);
    atomicAdd(o + 5, (unsigned long long)(st_spans + (uint32_t)prev_cnt));
    atomicAdd(o + 6, (unsigned long long)st_rounds);
    atomicAdd(o + 7, 1ull);
  }
}

}  // namespace

int ggd_launch_blend(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const ggd_splat* splat,
                     const uint32_t* list, const uint32_t* ranges, uint32_t capacity, float* out_color, float* final_T,
                     uint32_t* n_contrib) {
  const int gx = (prm.width + 15) / 16, gy = (prm.height + 15) / 16;
  if (gx * gy == 0) return GGD_OK;
  const int em = ctx->opt[GGD_OPT_EXP_MODE] == 3 ? 1 : ctx->opt[GGD_OPT_EXP_MODE];   // 3 (default): bare v_exp_f32 in the forward
  const bool cull = ctx->opt[GGD_OPT_BLEND_CULL] != 0;
  const int T = gx * gy;
  static const int lds_pad = getenv("GGD_BLEND_LDS_PAD") ? atoi(getenv("GGD_BLEND_LDS_PAD")) : 0;   // experiment: caps the waves per CU
#define GGD_LAUNCH_FWD2(EM, CU, ST)                                                                                     \
  hipLaunchKernelGGL((blend_forward_kernel<EM, CU, 1, 8, ST>), dim3(4 * T), dim3(64), lds_pad, s, prm.width, prm.height, \
                     gx, T, splat, list, ranges, capacity, prm.bg, out_color, final_T, n_contrib, ctx->blend_stats)
#define GGD_LAUNCH_FWD(EM, CU)                                                                                          \
  do {                                                                                                                  \
    if (ctx->blend_stats) GGD_LAUNCH_FWD2(EM, CU, true); else GGD_LAUNCH_FWD2(EM, CU, false);                           \
  } while (0)
  if (cull) {
    if (em == 0) GGD_LAUNCH_FWD(0, true); else if (em == 1) GGD_LAUNCH_FWD(1, true); else GGD_LAUNCH_FWD(2, true);
  } else {
    if (em == 0) GGD_LAUNCH_FWD(0, false); else if (em == 1) GGD_LAUNCH_FWD(1, false); else GGD_LAUNCH_FWD(2, false);
  }
#undef GGD_LAUNCH_FWD
#undef GGD_LAUNCH_FWD2
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

int ggd_launch_blend_backward(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const ggd_splat* splat,
                              const uint32_t* list, const uint32_t* ranges, const float* final_T,
                              const uint32_t* n_contrib, const float* dL_dpix, float* grad_acc) {
  const int gx = (prm.width + 15) / 16, gy = (prm.height + 15) / 16;
  if (gx * gy == 0) return GGD_OK;
  const int em = ctx->opt[GGD_OPT_EXP_MODE];   // 3 (default): compensated 2^x, contribution decision on the forward's bare v_exp_f32 value
  const bool cull = ctx->opt[GGD_OPT_BLEND_CULL] != 0;
  const int T = gx * gy;
  int split = ctx->opt[GGD_OPT_BLEND_SPLIT];
  // 4: four INDEPENDENT 8x8 quarter waves per tile (blend_backward_quarter_kernel); any other explicit value: the four
  // quarter waves of a tile in one workgroup, per-record sums combined in LDS (blend_backward_tile_kernel).  1 (auto): the
  // quarter form.  Until round 6 the rule sent grids below 2048 tiles to the tile form (round 3: 500 k / 512^2 184 vs 206 us);
  // with the quarter kernel's later changes (whole-record LDS prefetch, live-rectangle pre-cull, flush before the loads) it
  // is equal or faster on every grid measured -- 100 k / 512^2 54.9 vs 58.5 us, 30 k / 512^2 23.6 vs 24.9, 100 k / 256^2 55 vs 65,
  // 500 k / 512^2 cube 175 vs 173 (a tie), head-like 500 k / 512^2 (the train step's scenes) 281 vs 394 and 311 vs 484,
  // 1 M / 704^2 225 vs 250 (profiles/r06/backward_blend_form_ab.txt)
  if (split == 1) split = 4;
  static const int lds_pad = getenv("GGD_BLEND_BWD_LDS_PAD") ? atoi(getenv("GGD_BLEND_BWD_LDS_PAD")) : 0;   // experiment
  if (split == 4) {
#define GGD_LAUNCH_BQ(EM, CU)                                                                                             \
    do {                                                                                                                  \
      if (ctx->blend_stats)                                                                                               \
        hipLaunchKernelGGL((blend_backward_quarter_kernel<EM, CU, true>), dim3(4 * T), dim3(64), lds_pad, s, prm.width,   \
                           prm.height, gx, gy, splat, list, ranges, prm.bg, final_T, n_contrib, dL_dpix, grad_acc,        \
                           ctx->blend_stats);                                                                             \
      else                                                                                                                \
        hipLaunchKernelGGL((blend_backward_quarter_kernel<EM, CU>), dim3(4 * T), dim3(64), lds_pad, s, prm.width,         \
                           prm.height, gx, gy, splat, list, ranges, prm.bg, final_T, n_contrib, dL_dpix, grad_acc,        \
                           (unsigned long long*)nullptr);                                                                 \
    } while (0)
    if (cull) {
      if (em == 0) GGD_LAUNCH_BQ(0, true); else if (em == 1) GGD_LAUNCH_BQ(1, true); else if (em == 2) GGD_LAUNCH_BQ(2, true); else GGD_LAUNCH_BQ(3, true);
    } else {
      if (em == 0) GGD_LAUNCH_BQ(0, false); else if (em == 1) GGD_LAUNCH_BQ(1, false); else if (em == 2) GGD_LAUNCH_BQ(2, false); else GGD_LAUNCH_BQ(3, false);
    }
#undef GGD_LAUNCH_BQ
    GGD_HIP(hipGetLastError());
    return GGD_OK;
  }
#define GGD_LAUNCH_BT(EM, CU)                                                                                             \
  hipLaunchKernelGGL((blend_backward_tile_kernel<EM, CU>), dim3(T), dim3(256), 0, s, prm.width, prm.height, gx,           \
                     gy, splat, list, ranges, prm.bg, final_T, n_contrib, dL_dpix, grad_acc)
  if (cull) {
    if (em == 0) GGD_LAUNCH_BT(0, true); else if (em == 1) GGD_LAUNCH_BT(1, true); else if (em == 2) GGD_LAUNCH_BT(2, true); else GGD_LAUNCH_BT(3, true);
  } else {
    if (em == 0) GGD_LAUNCH_BT(0, false); else if (em == 1) GGD_LAUNCH_BT(1, false); else if (em == 2) GGD_LAUNCH_BT(2, false); else GGD_LAUNCH_BT(3, false);
  }
#undef GGD_LAUNCH_BT
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}
