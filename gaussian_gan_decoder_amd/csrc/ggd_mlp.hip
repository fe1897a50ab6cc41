// ggd_mlp.hip -- fused per-point Gaussian decoder (5 chained MLP heads) on the gfx950 matrix cores.
//
// SURVEY.md section 8f row 1 / BASELINE config 3: the only genuinely dense contraction next to the raster path.
// Replaces, for inference, the 5 x `Decoder` calls of main/decoder_models/sequential_decoder_reverse.py:68-85
// (each: concat -> Linear+GELU x3 -> Linear, main/decoder_models/base_decoder.py:8-27):
//   color(3) -> opacity(1) -> rotation(4) -> scale(3, -softplus(s+5)-2.5) -> xyz(3, *0.01 + position),
// every head seeing [plane_mean(32), position(3), outputs of the earlier heads].
//
// Formulation: TRANSPOSED GEMMs, Y^T[feature][point] = W[feature][k] . X^T[k][point], with
// v_mfma_f32_16x16x32_f16 (fp32 accumulate; f16 operands in the forward since round 4 -- 11 significant bits instead of
// bf16's 8 at the same MFMA rate, and a GELU that runs in packed f16; the backward kernels keep bf16 for dz's range).  The C/D layout of one layer (lane = point column, 4 consecutive
// feature rows per lane group) IS a valid B-operand layout of the next layer, so activations never leave registers
// between layers: no LDS round trip, no transposes.  (The MFMA pairs element e of lane (i,g) in A with element e
// of lane (j,g) in B, so any assignment of k to (g,e) is legal as long as A and B agree; the weight rows are
// pre-permuted on the host to the order the D layout produces.)  Weights of ONE head (86 KiB of 16-bit values, 16-byte slots
// XOR-swizzled by the row so that the ds_read_b128 lane groups are bank-conflict free) are resident in LDS; a 512-thread workgroup
// (8 waves: two per SIMD, one in its MFMA phase while the other does its GELUs) walks 32-point slabs.
// GELU: transcendental-free polynomials -- forward in packed f16 (gelu_h2x4), weight-gradient recompute in packed fp32 (gelu2x4).
#include "ggd_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int HID = 128;
constexpr int NHEAD = 5;
// LDS image of one head (bytes); rows are K bf16 + 8 bf16 of padding
// Weight rows are K bf16 with NO padding; the 16-byte slots of a row are XOR-swizzled by the row index so that a
// ds_read_b128 of one k-slot of 16 consecutive rows by the four lane groups of a wave (rows i = lane & 15, slot g + 4 s) is
// bank-conflict free: the LDS services a b128 read in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... over a
// 256-byte bank row; with padded rows (stride 272 B) two lanes of every group met on one 16-byte slot (47 % of the LDS
// cycles of these kernels were bank conflicts).  wslot() is the one place that knows the mapping; the pack kernels
// (ggd_mlp_pack.inc, ggd_mlp_hl.inc) and fused_decoder.pack_weights write through the same formula.
constexpr int ROW1 = 64 * 2;     // layer 1: K = 64 (32 plane features + 32 info slots): two rows per 256-byte bank row
constexpr int ROW2 = 128 * 2;    // layers 2..4: K = 128: one row per bank row
template <int ROW>
__device__ __forceinline__ int wslot(int row, int q) {   // byte offset of logical 16-byte slot q of `row`
  if (ROW == ROW2) return row * ROW2 + ((q ^ (row & 15)) << 4);
  if (ROW == ROW1) return row * ROW1 + ((q ^ ((row >> 1) & 7)) << 4);
  return row * ROW + (q << 4);                           // padded rows (W4^T): no swizzle
}
constexpr int OFF_W1 = 0;
constexpr int OFF_W2 = OFF_W1 + HID * ROW1;
constexpr int OFF_W3 = OFF_W2 + HID * ROW2;
constexpr int OFF_W4 = OFF_W3 + HID * ROW2;
constexpr int OFF_B = OFF_W4 + 16 * ROW2;          // fp32 biases: b1[128] b2[128] b3[128] b4[16]
constexpr int HEAD_BYTES = OFF_B + (3 * HID + 16) * 4;  // 87,616 B
#ifndef GGD_MLP_THREADS   // (timing experiments: 256 = one wave per SIMD in the 16-bit backward kernel, profiles/REJECTED.md round 6)
#define GGD_MLP_THREADS 512
#endif
constexpr int MLP_THREADS = GGD_MLP_THREADS;
constexpr int MLP_WAVES = MLP_THREADS / 64;
// forward kernel: the waves that share one head's weights in LDS (95 KB: one workgroup per CU).  Measured at 1 M points:
// 512 threads 0.796 ms, 768 (3 waves per SIMD, 150 VGPRs still fit) 0.787 ms, 1024 (128-VGPR cap: spills) 0.827 ms --
// the kernel is not occupancy-bound.
constexpr int FWD_THREADS = 512;
constexpr int FWD_WAVES = FWD_THREADS / 64;
constexpr int SLAB = 32;  // points per wave iteration (two 16-point MFMA column tiles share every weight read)

typedef float f2v __attribute__((ext_vector_type(2)));

#include "ggd_mlp_gelu.inc"

// fp32 form (the weight-gradient kernel recomputes a = gelu(z) with it; the forward used it until round 4).
// Two GELUs per call, transcendental-free and packed (v_pk_fma_f32): x * Phi(x) with
// Phi(x) ~= 0.5 + xc * P7(xc^2), xc = clamp(x, -4, 4) (least-squares fit on Chebyshev nodes; max |Phi error| 4.9e-5,
// max |GELU error| 2e-4 on [-4, 4] and 4.9e-5 * |x| beyond -- an order below the bf16 rounding the activation gets
// next).  12 VALU ops per pair.  (History: exact-erf Abramowitz-Stegun cost a v_rcp and a v_exp per value -- 14k of
// 17k cycles per slab; an erf polynomial of degree 17 cost 17 ops per pair.)
__device__ __forceinline__ f2v gelu2(f2v x) {
  const f2v xc = {__builtin_amdgcn_fmed3f(x.x, -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x.y, -4.0f, 4.0f)};
  const f2v s2 = xc * xc;
  f2v p = {-1.520480094e-09f, -1.520480094e-09f};
  p = __builtin_elementwise_fma(p, s2, (f2v){1.180964698e-07f, 1.180964698e-07f});
  p = __builtin_elementwise_fma(p, s2, (f2v){-4.014221549e-06f, -4.014221549e-06f});
  p = __builtin_elementwise_fma(p, s2, (f2v){7.960997496e-05f, 7.960997496e-05f});
  p = __builtin_elementwise_fma(p, s2, (f2v){-1.041295812e-03f, -1.041295812e-03f});
  p = __builtin_elementwise_fma(p, s2, (f2v){9.641715482e-03f, 9.641715482e-03f});
  p = __builtin_elementwise_fma(p, s2, (f2v){-6.614117560e-02f, -6.614117560e-02f});
  p = __builtin_elementwise_fma(p, s2, (f2v){3.988329117e-01f, 3.988329117e-01f});
  const f2v phi = __builtin_elementwise_fma(xc, p, (f2v){0.5f, 0.5f});
  return x * phi;
}

// Four GELU pairs in lock step: the Horner recurrences of a packed polynomial are serial, and a v_pk_* op that reads the
// result of the previous packed op needs a wait state (the compiler pads every step with s_nop 0: 800 of them per
// slab and head) -- with four independent chains advanced together every dependent pair is four instructions apart
// (904 -> 131 s_nop, 0.834 -> 0.806 ms at 1 M points).  The same polynomial as eight independent plain-fp32 chains (no
// v_pk_* at all) was measured too: 0.795 - 0.805 ms against 0.774 - 0.776 ms packed -- the packed form stays.
__device__ __forceinline__ void gelu2x4(f2v (&x)[4]) {
  f2v xc[4], s2[4], p[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    xc[i] = (f2v){__builtin_amdgcn_fmed3f(x[i].x, -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x[i].y, -4.0f, 4.0f)};
    s2[i] = xc[i] * xc[i];
    p[i] = (f2v){-1.520480094e-09f, -1.520480094e-09f};
  }
  constexpr float C[7] = {1.180964698e-07f, -4.014221549e-06f, 7.960997496e-05f, -1.041295812e-03f,
                          9.641715482e-03f, -6.614117560e-02f, 3.988329117e-01f};
#pragma unroll
  for (int k = 0; k < 7; ++k) {
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], s2[i], (f2v){C[k], C[k]});
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = x[i] * __builtin_elementwise_fma(xc[i], p[i], (f2v){0.5f, 0.5f});
}

__device__ __forceinline__ bf16x8 pack8(const f4& lo, const f4& hi) {
  bf16x8 r;
  r[0] = (__bf16)lo[0]; r[1] = (__bf16)lo[1]; r[2] = (__bf16)lo[2]; r[3] = (__bf16)lo[3];
  r[4] = (__bf16)hi[0]; r[5] = (__bf16)hi[1]; r[6] = (__bf16)hi[2]; r[7] = (__bf16)hi[3];
  return r;
}

// ---- forward operands: f16 (weights and activations), v_mfma_f32_16x16x32_f16
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));

// Four GELU pairs in packed f16, in lock step.  The hidden layers' weights and biases are HALVED in the forward image (an exact
// scaling), so the accumulators hold y = z / 2 and
//   gelu(z) = z / 2 + |z| (1/2 - Q(|z|)) = y + a s(v),   a = |y|,  v = min(a, 2) - 1,   s(v) ~ 1 - 2 Q(2 (v + 1)),  Q = 1 - Phi
// s = degree-6 polynomial in the monomial basis of v in [-1, 1] (all partial sums O(1): a Horner chain in f16 loses nothing to
// cancellation, whereas Phi = 0.5 + x P(x^2) has alternating coefficients up to 13 and is off by 1.5e-2 in f16), minimax fit
// under s(1) = 1: beyond |z| = 4 the result is z or 0 whatever |z| (scripts/gelu_f16_fit.py: fit, and the error of THIS
// instruction sequence with its roundings over every f16 value: <= 1.54e-3 for 2 <= z < 4 -- 9.8e-4 of it is the f16 rounding
// of the result; a bf16 result is off by 7.8e-3 there -- <= 8.8e-4 elsewhere, mean 7e-5).  NOT SATURATING (this tier does not
// spend two packed instructions per pair on a clamp; the reference-precision kernels do clamp): a pre-activation z > 65504
// gives +inf (y = z / 2 is finite up to 131008, but y + |y| is not), and z < -131008 gives NaN (y = -inf: inf - inf) where the
// true value is 0 -- one exploding pre-activation poisons THAT POINT's attributes, no other point's
// (tests/test_decoder_gpu.py::test_exploding_preactivation_stays_inside_its_point).
// 11 instructions per pair, conversion included: cvt, and, min, add, 6 fma, fma.
// Measured at 1 M points (profiles/r04/decoder_forward_f16.txt), against the packed-fp32 Phi polynomial on bf16 operands of
// rounds 1-3: kernel 905 -> 798 us with the z-form of this polynomial (12 instructions), error of the decoded attributes
// against the fp32 module 1.2e-3 -> 3.6e-4 max, 1.4e-4 -> 3.5e-5 mean; degree 8: 850 us / 2.4e-4; degree 5: 775 us / 6.2e-4.
// Every v_pk_*_f16 and v_cvt_pk_* costs 4.2 cycles of a SIMD, v_pk_*_f32 4.6, plain fp32 2.5
// (scripts/probes/f16_rate_probe.hip): the gain is the instruction count (13 packed-fp32 instructions and two med3 per
// pair before), not a faster pipe.  Like v_pk_*_f32, a v_pk_*_f16 that reads the previous packed result needs a wait state:
// left alone the compiler runs the four chains of a 16-byte piece one after the other with an s_nop behind every step
// (929 per slab and head; 67 in lock step).
__device__ __forceinline__ void gelu_h2x4(h2v (&y)[4]) {
  constexpr int DEG = 6;
  constexpr float S[DEG + 1] = {9.546607429e-01f, 2.169445321e-01f, -4.391409802e-01f, 4.247604060e-01f,
                                -9.772952171e-02f, -1.414002710e-01f, 8.190509189e-02f};
  h2v a[4], v[4], p[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = __builtin_elementwise_abs(y[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = __builtin_elementwise_min(a[i], (h2v){2.0f16, 2.0f16});
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = v[i] - (h2v){1.0f16, 1.0f16};
#pragma unroll
  for (int i = 0; i < 4; ++i)
    p[i] = __builtin_elementwise_fma(v[i], (h2v){(_Float16)S[DEG], (_Float16)S[DEG]}, (h2v){(_Float16)S[DEG - 1], (_Float16)S[DEG - 1]});
#pragma unroll
  for (int k = DEG - 2; k >= 0; --k) {
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], v[i], (h2v){(_Float16)S[k], (_Float16)S[k]});
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) y[i] = __builtin_elementwise_fma(a[i], p[i], y[i]);
}
__device__ __forceinline__ h2v cvt_h2(float a, float b) { return __builtin_convertvector((f2v){a, b}, h2v); }   // v_cvt_pk_f16_f32
// the kernel's inputs (plane features, positions, earlier heads' outputs) are clamped to the f16 range
__device__ __forceinline__ float clamp_h(float v) { return __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f); }
__device__ __forceinline__ h16x8 pack8h(const f4& lo, const f4& hi) {
  const h2v p0 = cvt_h2(clamp_h(lo[0]), clamp_h(lo[1])), p1 = cvt_h2(clamp_h(lo[2]), clamp_h(lo[3]));
  const h2v p2 = cvt_h2(clamp_h(hi[0]), clamp_h(hi[1])), p3 = cvt_h2(clamp_h(hi[2]), clamp_h(hi[3]));
  return (h16x8){p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
}

// One hidden layer for the wave's two 16-point column tiles: acc[c][mt] (8 feature tiles x 2 point tiles).
template <int KB /* 32-wide k blocks */, int ROW>
__device__ __forceinline__ void layer_mfma(const unsigned char* __restrict__ w, const float* __restrict__ bias,
                                           const h16x8 (&bin)[2][4], f4 (&acc)[2][8], int lane) {
  const int i = lane & 15, g = lane >> 4;
  // The weight fragments of feature tile mt + 1 are requested from LDS before the MFMAs of tile mt are issued (one tile of
  // look-ahead, KB extra 16-byte registers): with the reads and their MFMAs in the same scheduling region every tile
  // waited out the LDS latency (~200 cycles x 8 tiles x 3 layers per slab and head).
  h16x8 a_cur[KB], a_nxt[KB];
  f4 b_cur, b_nxt;
  {
#pragma unroll
    for (int s = 0; s < KB; ++s) a_cur[s] = *reinterpret_cast<const h16x8*>(w + wslot<ROW>(i, g + 4 * s));
    b_cur = *reinterpret_cast<const f4*>(bias + 4 * g);
  }
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    if (mt + 1 < 8) {
#pragma unroll
      for (int s = 0; s < KB; ++s) a_nxt[s] = *reinterpret_cast<const h16x8*>(w + wslot<ROW>(16 * (mt + 1) + i, g + 4 * s));
      b_nxt = *reinterpret_cast<const f4*>(bias + 16 * (mt + 1) + 4 * g);
    }
    __builtin_amdgcn_sched_barrier(0);
    acc[0][mt] = b_cur; acc[1][mt] = b_cur;   // D rows 4g..4g+3 of this feature tile start from the bias
#pragma unroll
    for (int s = 0; s < KB; ++s) {
      acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur[s], bin[0][s], acc[0][mt], 0, 0, 0);
      acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur[s], bin[1][s], acc[1][mt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the reads of later tiles from being hoisted further (VGPR pressure)
#pragma unroll
    for (int s = 0; s < KB; ++s) a_cur[s] = a_nxt[s];
    b_cur = b_nxt;
  }
}

// (The LDS-table GELU of the reference-precision kernels, ggd_mlp_gelu.inc, was measured here too, against the fp32 polynomial:
// 0.924 against 0.772 ms at 1 M points -- no LDS round trip fits inside a kernel whose LDS port is busy with weight fragments.)
// acc = y = z / 2 of the wave's tiles -> the next layer's B operand gelu(z) (f16).  STORE_Z (training): z = y + y is kept for
// the backward as f16, one 16-byte piece per (point tile, k block) in the blocked Z layout below -- the converted pairs are
// the ones the GELU starts from, so the store costs one v_pk_add_f16 per pair and no second conversion.
template <bool STORE_Z>
__device__ __forceinline__ void gelu_pack(const f4 (&acc)[2][8], h16x8 (&bout)[2][4], _Float16* __restrict__ zl, int64_t p0,
                                          int64_t cend, int lane);

// D-layout tile set (8 feature tiles x 2 point tiles, fp32) <-> bf16 [point][128] in the "Z layout": lane (j = point, g)
// holds features 16*mt + 4g + r; inside every 32-feature block the row stores them at position
//   zpos = 32*(mt/2) + 8*g + 4*(mt%2) + r        (true feature = 32*(mt/2) + 16*(mt%2) + 4*g + r)
// so that a lane's two tiles of a block are 16 contiguous bytes and the four lanes of a point write 64 contiguous bytes
// per instruction (the natural order gives 8-byte pieces, which lean on L2 write combining).  zbuf and dzbuf share the
// layout; the weight-gradient kernel maps positions back to features when it writes dW / db.
// HBM form of zbuf / dzbuf ("blocked Z layout"): per (head, layer) an array of 16-point blocks of 4 KB,
//   block[k][g][j][8]   k = 32-feature block, g = lane group, j = point inside the block, 8 bf16 = the lane's tiles 2k, 2k+1
// i.e. exactly the order in which a wave holds the data: the 64 lanes of ONE store / load instruction touch 1024
// contiguous bytes (the point-major form gave 16 separate 64-byte segments per instruction, half cache lines).  The
// weight-gradient kernel streams the same bytes linearly and only re-interprets a chunk's (point, position).
__device__ __forceinline__ int64_t zpiece(int64_t pt, int k, int g) {
  return (pt >> 4) * (16 * HID) + (int64_t)k * 512 + g * 128 + (pt & 15) * 8;
}
__device__ __forceinline__ size_t zlayer_elems(int N) { return (size_t)((N + 15) & ~15) * HID; }   // one (head, layer) plane

template <bool STORE_Z>
__device__ __forceinline__ void gelu_pack(const f4 (&acc)[2][8], h16x8 (&bout)[2][4], _Float16* __restrict__ zl, int64_t p0,
                                          int64_t cend, int lane) {
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f4& a0 = acc[c][2 * s];
      const f4& a1 = acc[c][2 * s + 1];
      h2v p[4] = {cvt_h2(a0[0], a0[1]), cvt_h2(a0[2], a0[3]), cvt_h2(a1[0], a1[1]), cvt_h2(a1[2], a1[3])};
      if (STORE_Z) {   // Z layout (zpos): the lane's tiles 2s, 2s+1 as ONE 16-byte piece; 4 lanes = 64 B
        const int64_t pt = p0 + 16 * c + j;
        const h2v z0 = p[0] + p[0], z1 = p[1] + p[1], z2 = p[2] + p[2], z3 = p[3] + p[3];
        if (pt < cend)
          *reinterpret_cast<h16x8*>(zl + zpiece(pt, s, g)) = (h16x8){z0[0], z0[1], z1[0], z1[1], z2[0], z2[1], z3[0], z3[1]};
      }
      gelu_h2x4(p);
      bout[c][s] = (h16x8){p[0][0], p[0][1], p[1][0], p[1][1], p[2][0], p[2][1], p[3][0], p[3][1]};
      __builtin_amdgcn_sched_barrier(0);
    }
}

// attrs row (16 floats per point): [0..2] color, [3] opacity, [4..7] rotation, [8..10] activated scale,
// [11..13] xyz, [14..15] unused.  It doubles as the carrier of the earlier heads' outputs between heads: the "info"
// vector a head sees is [position(3), attrs[0 .. n_extra)] with n_extra = 0, 3, 4, 8, 11.
// STORE_Z (training): the pre-activations of the three hidden layers are kept (rounded to f16) for the backward,
// zbuf[head][layer][16-point block][4 KB] (blocked Z layout above).
template <bool STORE_Z>
__global__ __launch_bounds__(FWD_THREADS, FWD_WAVES / 4) void decoder_forward_kernel(const float* __restrict__ feat,
                                                                         const float* __restrict__ pos, int N,
                                                                         const unsigned char* __restrict__ packed,
                                                                         float* attrs, _Float16* __restrict__ zbuf) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;  // HEAD_BYTES: weights + biases of one head
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int j = lane & 15, g = lane >> 4;

  // contiguous chunk of points per workgroup, 32-point slabs round-robin over its waves (the SAME wave revisits
  // the same slab for every head, so the attrs rows it wrote are its own)
  const int64_t per = (((int64_t)N + gridDim.x - 1) / gridDim.x + SLAB - 1) / SLAB * SLAB;
  const int64_t cbeg = (int64_t)blockIdx.x * per;
  const int64_t cend = min((int64_t)N, cbeg + per);

  for (int head = 0; head < NHEAD; ++head) {
    const int n_extra = head == 0 ? 0 : (head == 1 ? 3 : (head == 2 ? 4 : (head == 3 ? 8 : 11)));
    __syncthreads();  // everyone is done with the previous head's weights (and its attrs stores are issued)
    {
      const uint4* src = reinterpret_cast<const uint4*>(packed + (size_t)head * HEAD_BYTES);
      uint4* dst = reinterpret_cast<uint4*>(wl);
      for (int k = tid; k < HEAD_BYTES / 16; k += FWD_THREADS) dst[k] = src[k];
    }
    __threadfence_block();
    __syncthreads();
    const float* b1 = reinterpret_cast<const float*>(wl + OFF_B);
    const float* b2 = b1 + HID;
    const float* b3 = b2 + HID;
    const float* b4 = b3 + HID;

    // (Requesting the next slab's inputs one slab ahead -- 24 VGPRs of raw fp32 in flight, the wait moves to the end of the slab --
    // changes nothing: 0.659 against 0.660 ms.  The loads are not what the two waves of a SIMD wait for.)
    for (int64_t p0 = cbeg + (int64_t)wv * SLAB; p0 < cend; p0 += (int64_t)FWD_WAVES * SLAB) {
      // ---- inputs: k-block 0 = 32 plane features, k-block 1 = info slots 4g..4g+3 (upper half of the block zero)
      h16x8 bin[2][4];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int64_t pt = p0 + 16 * c + j;
        const bool ok = pt < cend;
        f4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0}, inf = {0, 0, 0, 0};
        if (ok) {
          lo = *reinterpret_cast<const f4*>(feat + pt * 32 + 4 * g);
          hi = *reinterpret_cast<const f4*>(feat + pt * 32 + 16 + 4 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int q = 4 * g + e;  // info slot
            float v = 0.0f;
            if (q < 3) v = pos[pt * 3 + q];
            else if (q - 3 < n_extra) v = attrs[pt * 16 + (q - 3)];
            inf[e] = v;
          }
        }
        const f4 z = {0, 0, 0, 0};
        bin[c][0] = pack8h(lo, hi);
        bin[c][1] = pack8h(inf, z);
      }
      f4 acc[2][8];
      h16x8 bh[2][4];
      _Float16* zh = STORE_Z ? zbuf + (size_t)(head * 3) * zlayer_elems(N) : nullptr;
      layer_mfma<2, ROW1>(wl + OFF_W1, b1, bin, acc, lane);
      gelu_pack<STORE_Z>(acc, bh, zh, p0, cend, lane);
      layer_mfma<4, ROW2>(wl + OFF_W2, b2, bh, acc, lane);
      gelu_pack<STORE_Z>(acc, bh, zh + zlayer_elems(N), p0, cend, lane);
      layer_mfma<4, ROW2>(wl + OFF_W3, b3, bh, acc, lane);
      gelu_pack<STORE_Z>(acc, bh, zh + 2 * zlayer_elems(N), p0, cend, lane);
      // ---- output layer: one feature tile (weight rows >= out_dim are zero)
      f4 out[2];
      {
        const f4 bb = *reinterpret_cast<const f4*>(b4 + 4 * g);
        out[0] = bb; out[1] = bb;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const h16x8 a = *reinterpret_cast<const h16x8*>(wl + OFF_W4 + wslot<ROW2>(j, g + 4 * s));
          out[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bh[0][s], out[0], 0, 0, 0);
          out[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bh[1][s], out[1], 0, 0, 0);
        }
      }
      // lane group 0 holds output features 0..3 of point j (D rows 0..3)
      if (g == 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int64_t pt = p0 + 16 * c + j;
          if (pt >= cend) continue;
          float* arow = attrs + pt * 16;
          const f4 o = out[c];
          if (head == 0) {
            arow[0] = o[0]; arow[1] = o[1]; arow[2] = o[2];
          } else if (head == 1) {
            arow[3] = o[0];
          } else if (head == 2) {
            arow[4] = o[0]; arow[5] = o[1]; arow[6] = o[2]; arow[7] = o[3];
          } else if (head == 3) {  // -softplus(s + 5) - 2.5   (torch.nn.Softplus: beta 1, threshold 20)
            // v_exp_f32 / v_log_f32 forms: |error| <= 2e-7 absolute (1 + e rounds, the hardware log2 is good to an ulp) against
            // this tier's 2e-3; libm's expf + log1pf were ~800 instructions per slab on this head -- 9 % of the kernel's issue
            // slots averaged over the heads (0.660 -> 0.633 ms at 1 M points).  The reference-precision kernel (bar 1e-4) does the same.
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              const float v = o[q] + 5.0f;
              const float sp = v > 20.0f ? v : __logf(1.0f + __expf(v));
              arow[8 + q] = -sp - 2.5f;
            }
          } else {                 // xyz = head * 0.01 + position
            arow[11] = o[0] * 0.01f + pos[pt * 3]; arow[12] = o[1] * 0.01f + pos[pt * 3 + 1];
            arow[13] = o[2] * 0.01f + pos[pt * 3 + 2];
            arow[14] = 0.0f; arow[15] = 0.0f;
          }
        }
      }
    }
  }
}

#include "ggd_mlp_bwd.inc"
#include "ggd_mlp_wgrad.inc"
#include "ggd_mlp_pack.inc"
#include "ggd_mlp_hl.inc"

}  // namespace

extern "C" size_t ggd_decoder_packed_bytes(void) { return (size_t)NHEAD * HEAD_BYTES; }

extern "C" int ggd_decoder_pack(ggd_ctx* ctx, void* stream, const float* const* params40, void* packed, void* packed_t) {
  if (!ctx) return GGD_E_INVALID;
  if (!params40 || !packed) return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_pack: NULL pointer");
  ggd_pack_ptrs ptrs;
  for (int i = 0; i < NHEAD * 8; ++i) {
    if (!params40[i]) return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_pack: NULL parameter pointer");
    ptrs.p[i] = params40[i];
  }
  const int total = NHEAD * PACK_PER_HEAD;
  hipLaunchKernelGGL(decoder_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), ptrs,
                     static_cast<unsigned char*>(packed), static_cast<unsigned char*>(packed_t));
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

// the GELU / GELU' tables (ggd_mlp_gelu.inc), built once per context in double precision
static int gelu_tables(ggd_ctx* ctx) {
  if (ctx->gelu_tables) return GGD_OK;
  GGD_HIP(hipMalloc(&ctx->gelu_tables, GT_BYTES));
  hipLaunchKernelGGL(hl_tables_kernel, dim3((GT_N_BWD + 256) / 256), dim3(256), 0, nullptr, static_cast<unsigned char*>(ctx->gelu_tables));
  GGD_HIP(hipGetLastError());
  GGD_HIP(hipStreamSynchronize(nullptr));
  return GGD_OK;
}

static int decoder_forward_impl(ggd_ctx* ctx, void* stream, const float* feat, const float* pos, int32_t N,
                                const void* packed_weights, float* attrs, void* zbuf) {
  if (!ctx) return GGD_E_INVALID;
  if (N < 0) return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_forward: N < 0");
  if (N == 0) return GGD_OK;
  if (!feat || !pos || !packed_weights || !attrs) return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_forward: NULL pointer");
  const size_t lds = HEAD_BYTES;
  if (!(ctx->attr_mask & GGD_ATTR_MLP_FWD)) {
    GGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_forward_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_forward_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->attr_mask |= GGD_ATTR_MLP_FWD;
  }
  int grid = (N + FWD_WAVES * SLAB - 1) / (FWD_WAVES * SLAB);  // at least one slab per wave
  if (grid > 256) grid = 256;
  if (grid < 1) grid = 1;
  if (zbuf)
    hipLaunchKernelGGL(decoder_forward_kernel<true>, dim3(grid), dim3(FWD_THREADS), lds, static_cast<hipStream_t>(stream),
                       feat, pos, N, static_cast<const unsigned char*>(packed_weights), attrs, static_cast<_Float16*>(zbuf));
  else
    hipLaunchKernelGGL(decoder_forward_kernel<false>, dim3(grid), dim3(FWD_THREADS), lds, static_cast<hipStream_t>(stream),
                       feat, pos, N, static_cast<const unsigned char*>(packed_weights), attrs, (_Float16*)nullptr);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

// attrs rows [N][16] <-> the five contiguous per-Gaussian arrays the rasterizer's entry takes (xyz [N][3], scale [N][3],
// rotation [N][4], opacity [N][1], colour [N][3]); one pass each way instead of five strided torch copies per scene.
// MERGE: the gradient row gets zeros where an array is NULL (an attribute the loss did not reach) and in the two unused slots.
template <bool MERGE>
__global__ __launch_bounds__(256) void attrs_split_kernel(float* __restrict__ attrs, int64_t N, float* __restrict__ xyz,
                                                          float* __restrict__ scale, float* __restrict__ rot,
                                                          float* __restrict__ opac, float* __restrict__ color) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  f4* row = reinterpret_cast<f4*>(attrs + n * 16);
  if (!MERGE) {
    const f4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];   // colour 3 | opacity, rotation 4, scale 3 | xyz.x, xyz.yz | 0 0
    color[3 * n] = r0[0]; color[3 * n + 1] = r0[1]; color[3 * n + 2] = r0[2];
    opac[n] = r0[3];
    *reinterpret_cast<f4*>(rot + 4 * n) = r1;
    scale[3 * n] = r2[0]; scale[3 * n + 1] = r2[1]; scale[3 * n + 2] = r2[2];
    xyz[3 * n] = r2[3]; xyz[3 * n + 1] = r3[0]; xyz[3 * n + 2] = r3[1];
  } else {
    f4 r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0;
    if (color) { r0[0] = color[3 * n]; r0[1] = color[3 * n + 1]; r0[2] = color[3 * n + 2]; }
    if (opac) r0[3] = opac[n];
    if (rot) r1 = *reinterpret_cast<const f4*>(rot + 4 * n);
    if (scale) { r2[0] = scale[3 * n]; r2[1] = scale[3 * n + 1]; r2[2] = scale[3 * n + 2]; }
    if (xyz) { r2[3] = xyz[3 * n]; r3[0] = xyz[3 * n + 1]; r3[1] = xyz[3 * n + 2]; }
    row[0] = r0; row[1] = r1; row[2] = r2; row[3] = r3;
  }
}

extern "C" int ggd_attrs_split(ggd_ctx* ctx, void* stream, const float* attrs, int64_t N, float* xyz, float* scale,
                               float* rotation, float* opacity, float* color) {
  if (!ctx) return GGD_E_INVALID;
  if (N <= 0) return GGD_OK;
  if (!attrs || !xyz || !scale || !rotation || !opacity || !color) return ggd_fail(ctx, GGD_E_INVALID, "ggd_attrs_split: NULL pointer");
  hipLaunchKernelGGL(attrs_split_kernel<false>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     const_cast<float*>(attrs), N, xyz, scale, rotation, opacity, color);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

extern "C" int ggd_attrs_merge(ggd_ctx* ctx, void* stream, int64_t N, const float* dxyz, const float* dscale,
                               const float* drotation, const float* dopacity, const float* dcolor, float* dattrs) {
  if (!ctx) return GGD_E_INVALID;
  if (N <= 0) return GGD_OK;
  if (!dattrs) return ggd_fail(ctx, GGD_E_INVALID, "ggd_attrs_merge: NULL pointer");
  hipLaunchKernelGGL(attrs_split_kernel<true>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     dattrs, N, const_cast<float*>(dxyz), const_cast<float*>(dscale), const_cast<float*>(drotation),
                     const_cast<float*>(dopacity), const_cast<float*>(dcolor));
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

extern "C" int ggd_decoder_forward(ggd_ctx* ctx, void* stream, const float* feat, const float* pos, int32_t N,
                                   const void* packed_weights, float* attrs) {
  return decoder_forward_impl(ctx, stream, feat, pos, N, packed_weights, attrs, nullptr);
}

extern "C" size_t ggd_decoder_packed_t_bytes(void) { return (size_t)NHEAD * HEADT_BYTES; }

extern "C" int ggd_decoder_backward(ggd_ctx* ctx, void* stream, int32_t N, const void* packed_t, const float* attrs,
                                    const float* dattrs, const void* zbuf, void* dzbuf, float* dout, float* dfeat,
                                    float* dinfo) {
  if (!ctx) return GGD_E_INVALID;
  if (N < 0) return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_backward: N < 0");
  if (N == 0) return GGD_OK;
  if (!packed_t || !attrs || !dattrs || !zbuf || !dzbuf || !dout || !dfeat || !dinfo)
    return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_backward: NULL pointer");
  if (!(ctx->attr_mask & GGD_ATTR_MLP_BWD)) {
    GGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_backward_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)HEADT_BYTES));
    ctx->attr_mask |= GGD_ATTR_MLP_BWD;
  }
  int grid = (N + MLP_WAVES * SLAB - 1) / (MLP_WAVES * SLAB);
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(decoder_backward_kernel, dim3(grid), dim3(MLP_THREADS), HEADT_BYTES, static_cast<hipStream_t>(stream),
                     N, 0, N, static_cast<const unsigned char*>(packed_t), attrs, dattrs, static_cast<const _Float16*>(zbuf),
                     static_cast<__bf16*>(dzbuf), dout, dfeat, dinfo);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

extern "C" size_t ggd_decoder_zbuf_bytes(int32_t N) {   // 16-point blocks: N rounded up to a multiple of 16
  return (size_t)NHEAD * 3 * (size_t)(N > 0 ? ((N + 15) & ~15) : 0) * HID * 2;
}

extern "C" int ggd_decoder_forward_train(ggd_ctx* ctx, void* stream, const float* feat, const float* pos, int32_t N,
                                         const void* packed_weights, float* attrs, void* zbuf) {
  if (ctx && N > 0 && !zbuf) return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_forward_train: zbuf is NULL");
  return decoder_forward_impl(ctx, stream, feat, pos, N, packed_weights, attrs, zbuf);
}

extern "C" size_t ggd_decoder_wgrad_floats(void) { return (size_t)NHEAD * WG_HEAD_FLOATS; }

extern "C" int ggd_decoder_wgrad(ggd_ctx* ctx, void* stream, int32_t N, const void* zbuf, const void* dzbuf,
                                 const float* dout, const float* feat, const float* pos, const float* attrs,
                                 float* wgrad) {
  if (!ctx) return GGD_E_INVALID;
  if (N < 0) return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_wgrad: N < 0");
  if (N == 0) return GGD_OK;
  if (!zbuf || !dzbuf || !dout || !feat || !pos || !attrs || !wgrad)
    return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_wgrad: NULL pointer");
  if (!(ctx->attr_mask & GGD_ATTR_MLP_WGRAD)) {
    GGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_wgrad_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_LDS));
    ctx->attr_mask |= GGD_ATTR_MLP_WGRAD;
  }
  // split-K over up to 128 point chunks x 20 (head, layer) problems (measured: 32 -> 4.6 ms, 64 -> 3.9, 128 -> 3.7, 256 -> 3.9); at least 4 stages per workgroup
  int chunks = (N + 4 * WG_K - 1) / (4 * WG_K);
  if (chunks > 128) chunks = 128;
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(decoder_wgrad_kernel, dim3(chunks, NHEAD * 4), dim3(WG_THREADS), WG_LDS,
                     static_cast<hipStream_t>(stream), N, 0, N, static_cast<const __bf16*>(zbuf),
                     static_cast<const __bf16*>(dzbuf), dout, feat, pos, attrs, wgrad);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

// Backward + weight gradients in point CHUNKS: the backward kernel of a chunk is followed immediately by the weight-
// gradient kernel of the same chunk, so that the dz / z rows it reads are still in the 256 MB Infinity Cache instead of
// coming back from HBM.  chunk <= 0: one chunk (= ggd_decoder_backward followed by ggd_decoder_wgrad).
extern "C" int ggd_decoder_backward_wgrad(ggd_ctx* ctx, void* stream, int32_t N, int32_t chunk, const void* packed_t,
                                          const float* attrs, const float* dattrs, const void* zbuf, void* dzbuf,
                                          float* dout, float* dfeat, float* dinfo, const float* feat, const float* pos,
                                          float* wgrad) {
  if (!ctx) return GGD_E_INVALID;
  if (N < 0) return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_backward_wgrad: N < 0");
  if (N == 0) return GGD_OK;
  if (!packed_t || !attrs || !dattrs || !zbuf || !dzbuf || !dout || !dfeat || !dinfo || !feat || !pos || !wgrad)
    return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_backward_wgrad: NULL pointer");
  if ((ctx->attr_mask & (GGD_ATTR_MLP_BWD | GGD_ATTR_MLP_WGRAD)) != (GGD_ATTR_MLP_BWD | GGD_ATTR_MLP_WGRAD)) {
    GGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_backward_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)HEADT_BYTES));
    GGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_wgrad_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_LDS));
    ctx->attr_mask |= GGD_ATTR_MLP_BWD | GGD_ATTR_MLP_WGRAD;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (chunk <= 0 || chunk > N) chunk = N;
  chunk = (chunk + 255) / 256 * 256;
  for (int32_t first = 0; first < N; first += chunk) {
    const int32_t last = first + chunk < N ? first + chunk : N;
    const int32_t n = last - first;
    int grid = (n + MLP_WAVES * SLAB - 1) / (MLP_WAVES * SLAB);
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL(decoder_backward_kernel, dim3(grid), dim3(MLP_THREADS), HEADT_BYTES, s, N, first, last,
                       static_cast<const unsigned char*>(packed_t), attrs, dattrs, static_cast<const _Float16*>(zbuf),
                       static_cast<__bf16*>(dzbuf), dout, dfeat, dinfo);
    int chunks = (n + 4 * WG_K - 1) / (4 * WG_K);
    if (chunks > 128) chunks = 128;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(decoder_wgrad_kernel, dim3(chunks, NHEAD * 4), dim3(WG_THREADS), WG_LDS, s, N, first, last,
                       static_cast<const __bf16*>(zbuf), static_cast<const __bf16*>(dzbuf), dout, feat, pos, attrs, wgrad);
  }
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

// ---- the decoder at reference precision (split bf16 operands, csrc/ggd_mlp_hl.inc) -------------------------------------------
extern "C" size_t ggd_decoder_packed_hl_bytes(void) { return (size_t)NHEAD * HLF_HEAD; }
extern "C" size_t ggd_decoder_dzbuf_hl_bytes(int32_t N) { return ggd_decoder_zbuf_bytes(N); }   // dz: one loss-scaled fp16 plane (z: one fp16 plane, ggd_decoder_zbuf_bytes)
extern "C" size_t ggd_decoder_packed_t_hl_bytes(void) { return (size_t)NHEAD * HLT_HEAD; }

extern "C" int ggd_decoder_pack_hl(ggd_ctx* ctx, void* stream, const float* const* params40, void* packed_hl,
                                   void* packed_t_hl) {
  if (!ctx) return GGD_E_INVALID;
  if (!params40 || !packed_hl) return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_pack_hl: NULL pointer");
  ggd_pack_ptrs ptrs;
  for (int i = 0; i < NHEAD * 8; ++i) {
    if (!params40[i]) return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_pack_hl: NULL parameter pointer");
    ptrs.p[i] = params40[i];
  }
  const int total = NHEAD * HLPK_PER_HEAD;
  hipLaunchKernelGGL(decoder_pack_hl_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), ptrs,
                     static_cast<unsigned char*>(packed_hl), static_cast<unsigned char*>(packed_t_hl));
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

static int hl_attributes(ggd_ctx* ctx) {
  if (ctx->attr_mask & GGD_ATTR_MLP_HL) return GGD_OK;
  { const int rt = gelu_tables(ctx); if (rt != GGD_OK) return rt; }
  GGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_forward_hl_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)HL_LDS_FWD));
  GGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_forward_hl_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)HL_LDS_FWD));
  GGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_backward_hl_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)HL_LDS_BWD));
  GGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_wgrad_hl_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)WGH_LDS));
  ctx->attr_mask |= GGD_ATTR_MLP_HL;
  return GGD_OK;
}

extern "C" int ggd_decoder_forward_hl(ggd_ctx* ctx, void* stream, const float* feat, const float* pos, int32_t N,
                                      const void* packed_hl, float* attrs, void* zbuf) {
  if (!ctx) return GGD_E_INVALID;
  if (N < 0) return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_forward_hl: N < 0");
  if (N == 0) return GGD_OK;
  if (!feat || !pos || !packed_hl || !attrs) return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_forward_hl: NULL pointer");
  const int rc = hl_attributes(ctx);
  if (rc != GGD_OK) return rc;
  int grid = (N + FWD_WAVES * SLAB - 1) / (FWD_WAVES * SLAB);
  if (grid > 256) grid = 256;
  if (zbuf)
    hipLaunchKernelGGL(decoder_forward_hl_kernel<true>, dim3(grid), dim3(FWD_THREADS), HL_LDS_FWD,
                       static_cast<hipStream_t>(stream), feat, pos, N, static_cast<const unsigned char*>(packed_hl), attrs,
                       static_cast<__bf16*>(zbuf), static_cast<const unsigned char*>(ctx->gelu_tables));
  else
    hipLaunchKernelGGL(decoder_forward_hl_kernel<false>, dim3(grid), dim3(FWD_THREADS), HL_LDS_FWD,
                       static_cast<hipStream_t>(stream), feat, pos, N, static_cast<const unsigned char*>(packed_hl), attrs,
                       (__bf16*)nullptr, static_cast<const unsigned char*>(ctx->gelu_tables));
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

extern "C" int ggd_decoder_backward_wgrad_hl(ggd_ctx* ctx, void* stream, int32_t N, int32_t chunk, const void* packed_t_hl,
                                             const float* attrs, const float* dattrs, const void* zbuf, void* dzbuf,
                                             float* dout, float* dfeat, float* dinfo, const float* feat, const float* pos,
                                             float* wgrad) {
  if (!ctx) return GGD_E_INVALID;
  if (N < 0) return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_backward_wgrad_hl: N < 0");
  if (N == 0) return GGD_OK;
  if (!packed_t_hl || !attrs || !dattrs || !zbuf || !dzbuf || !dout || !dfeat || !dinfo || !feat || !pos || !wgrad)
    return ggd_fail(ctx, GGD_E_INVALID, "ggd_decoder_backward_wgrad_hl: NULL pointer");
  const int rc = hl_attributes(ctx);
  if (rc != GGD_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // scales of the fp16 dz planes: one exponent per (head, 32-point slab) written by the backward, the largest per head
  // (library scratch: both kernels of this call, nothing in between)
  const int64_t nslab = ((int64_t)N + SLAB - 1) / SLAB;
  const size_t kbytes = ggd_align((size_t)NHEAD * nslab * sizeof(int)) + 256;
  int rck = ggd_reserve_scratch(ctx, kbytes, s);
  if (rck != GGD_OK) return rck;
  int* karr = static_cast<int*>(ctx->scratch);
  int* kref = reinterpret_cast<int*>(static_cast<char*>(ctx->scratch) + ggd_align((size_t)NHEAD * nslab * sizeof(int)));
  GGD_HIP(hipMemsetAsync(kref, 0x80, 64, s));     // every byte 0x80: far below any exponent (atomicMax target); < HL_KNONE is never stored
  if (chunk <= 0 || chunk > N) chunk = N;
  chunk = (chunk + 255) / 256 * 256;
  for (int32_t first = 0; first < N; first += chunk) {
    const int32_t last = first + chunk < N ? first + chunk : N;
    const int32_t n = last - first;
    int grid = (n + MLP_WAVES * SLAB - 1) / (MLP_WAVES * SLAB);
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL(decoder_backward_hl_kernel, dim3(grid), dim3(MLP_THREADS), HL_LDS_BWD, s, N, first, last,
                       static_cast<const unsigned char*>(packed_t_hl), attrs, dattrs, static_cast<const __bf16*>(zbuf),
                       static_cast<__bf16*>(dzbuf), dout, dfeat, dinfo, karr, kref,
                       static_cast<const unsigned char*>(ctx->gelu_tables));
    int chunks = (n + 4 * WG_K - 1) / (4 * WG_K);
    if (chunks > 128) chunks = 128;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(decoder_wgrad_hl_kernel, dim3(chunks, NHEAD * 4), dim3(WG_THREADS), WGH_LDS, s, N, first, last,
                       static_cast<const __bf16*>(zbuf), static_cast<const __bf16*>(dzbuf), dout, feat, pos, attrs, wgrad, karr, kref,
                       static_cast<const unsigned char*>(ctx->gelu_tables));
  }
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}
