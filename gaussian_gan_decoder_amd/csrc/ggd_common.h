// ggd_common.h -- shared declarations of the gfx950 rasterizer library (host ctx + device helpers).
//
// Arithmetic contract: every translation unit is compiled with -ffp-contract=off, so an fp32 expression written
// here is evaluated as written (IEEE mul/add/div/sqrt, correctly rounded) unless it says __builtin_fmaf
// explicitly.  The per-Gaussian stage keeps the operation order of the algorithm's published form so that its
// integer outputs (radii, tiles_touched, depth bits -> sort keys) are reproducible bit-for-bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "ggd_raster.h"

#define GGD_WAVE 64

// Stage ids for the optional hipEvent profiler (ggd_set_profiling).
enum {
  ST_PREPROCESS = 0, ST_SCAN, ST_READBACK, ST_DUPLICATE, ST_SORT, ST_RANGES, ST_BLEND,
  ST_BLEND_BWD, ST_PREPROCESS_BWD, ST_COUNT
};

enum { GGD_ATTR_MLP_FWD = 1, GGD_ATTR_MLP_BWD = 2, GGD_ATTR_MLP_WGRAD = 4, GGD_ATTR_MLP_HL = 64 };

// layout of the debug statistics buffer of the forward blend (ggd_blend_stats / ggd_blend_timeline)
constexpr int GGD_STATS_MODE = 9;            // word: 0 = counters (atomics), 1 = per-wave timeline slots
constexpr int GGD_STATS_BWD = 16;            // eight counters of the backward blend (quarter form), see blend_backward_quarter_kernel
constexpr int GGD_STATS_HEAD = 32;           // first timeline slot (3 words per wave: start, end, listed << 32 | gathered)
constexpr int GGD_STATS_MAX_WAVES = 1 << 17;

// two-launch depth sort (described at ggd_fold below)
constexpr int GGD_MSD_SHIFT = 14, GGD_MSD_BINS = 1024, GGD_MSD_CAP = 12288, GGD_MSD_MAX_TILES = 2048, GGD_MSD_BAN = 64;
constexpr int GGD_MSD_WIN = 32, GGD_MSD_MAX_SHIFT = 16;   // frames whose key ranges form the window; two 8-bit passes finish a bucket
constexpr int GGD_MSD_TARGET = 512;   // buckets the window is spread over: 512 = one resident round of the finish kernel (two workgroups
                                      // per CU); measured at 1 M / 1024^2, sort stage cube / shell: 1024 -> 27.4 / 35.8 us, 512 -> 23.5 / 33.5,
                                      // 256 -> 25.8 / 34.1 (gpurun_out r06b; round 5's fixed bits 14..23: 23.2 / 34.5)

struct ggd_ctx {
  int device = 0;
  void* scratch = nullptr;      // grow-only device workspace (sort histograms, scan block sums, dL_dconic, ...)
  size_t scratch_bytes = 0;
  uint32_t attr_mask = 0;       // GGD_ATTR_* bits: kernels whose dynamic-LDS limit was raised on this ctx's device
  uint32_t* d_words = nullptr;  // small device control block: [0] total R, [1] prefilter trap flag
  uint32_t* h_words = nullptr;  // pinned host mirror
  uint32_t* h_words_dev = nullptr;  // the same memory as the device sees it (the scan writes num_rendered there itself)
  uint32_t* sortctl = nullptr;      // depth sort's control block (histograms, tickets, kept-key count) in its own allocation
  bool sortctl_clean = false;       // cleared by this frame's preprocess and not yet consumed by a sort
  uint32_t* scan_sums = nullptr;    // block sums of a scan that rides on the depth sort (own allocation, grow-only)
  int scan_sums_cap = 0;
  // depth-sort front end folded into the preprocess kernel (ggd_fold below): two control blocks used alternately
  uint32_t* foldctl[2] = {nullptr, nullptr};
  size_t foldctl_cap = 0;           // words per block
  size_t foldctl_dirty[2] = {0, 0}; // words of each block its last user may have written (what the next clear must cover)
  int fold_cur = 0;                 // block the next folding preprocess accumulates into (cleared by the previous one)
  bool fold_poisoned = false;       // a folding launch failed: nothing is known about the blocks -- clear both before the next use
  bool fold_active = false;         // this call's preprocess left histograms + workgroup sums for the sort of the same call
  // the depth keys' top byte (sign + 7 exponent bits) is constant in most scenes: after GGD_FLAT_STREAK such frames in a row
  // the fourth sort pass -- an empty launch, 4.9 us -- is not launched at all; the frame's own histogram says whether that
  // was right (bit 63 of the tagged num_rendered word), and a frame it was wrong for is binned and blended again
  int flat_streak = 0;
  bool spec3 = false;               // this call launched three passes
  bool frame_flat = false;          // this call's top byte was constant (read with num_rendered)
  bool frame_folded = false;        // ... and it ran the folded front end (frame_flat is meaningful)
  unsigned long long spec3_misses = 0;
  // two-launch sort (GGD_OPT_MSD_SORT): decided per frame before the preprocess launch (it selects the histograms to build)
  bool msd_frame = false;           // this call's front end built the two-launch sort's histograms
  bool frame_msd_ok = false;        // ... and they say the two-launch sort was valid for this frame (read with num_rendered)
  int msd_ban = 0;                  // frames to wait before speculating again after a frame it was not valid for
  int msd_oversize_streak = 0;      // consecutive oversized-bucket misses: the pause doubles (8, 16, 32, 64 frames) until a frame succeeds
  unsigned long long msd_frames = 0;
  // ... over a speculated KEY WINDOW (round 6): buckets = (key - msd_lo) >> msd_shift, window and shift fitted to the depth
  // keys of the recent folded frames (their min / max arrive with num_rendered), so a depth range that straddles a binade
  // (2.0: top byte 0x3F | 0x40) no longer falls back to the four-pass sort
  uint32_t msd_lo = 0;              // this frame's window start and bucket shift
  int msd_shift = GGD_MSD_SHIFT;
  uint32_t win_lo[GGD_MSD_WIN], win_hi[GGD_MSD_WIN];   // ring: kept-key min / max of the last folded frames
  int win_n = 0, win_pos = 0;
  uint32_t frame_kmin = 0xffffffffu, frame_kmax = 0u, frame_msd_flags = 0u;   // read with num_rendered
  int msd_buckets = GGD_MSD_TARGET; // buckets the window is spread over (GGD_MSD_BUCKETS: timing experiments)
  // ggd_forward_enqueue ... ggd_forward_collect: the frame whose num_rendered has not been collected yet
  struct { bool valid = false; ggd_params prm; const void* geom = nullptr; void* binning = nullptr; int64_t capacity = 0;
           void* img = nullptr; float* out = nullptr; } pending;
  bool scan_deferred = false;       // geometry_enqueue left the scan to the sort launches of the same call
  void* gelu_tables = nullptr;      // GELU / GELU' interpolation tables of the reference-precision decoder kernels (built on first use)
  uint32_t r_tag = 0;               // sequence number of the single-call forward whose num_rendered the host is waiting for
  bool r_pending = false;           // the host waits for the tagged word (h_words[2..3]), not for the end of the frame
  void* dbg_keys = nullptr;     // debug copy of the unsorted list
  void* dbg_vals = nullptr;
  size_t dbg_cap = 0;
  int opt[GGD_OPT_COUNT] = {3, 1, 1, 1, 1, 1};  // exp: bare v_exp_f32 in the forward blend, compensated 2^x (1-2 ulp) in the backward
  unsigned long long* blend_stats = nullptr;  // debug: device counters filled by the forward blend when non-null
  unsigned long long* stats_buf = nullptr;    // its storage: [0..5] counters, [GGD_STATS_MODE] 1 = per-wave timeline, slots from GGD_STATS_HEAD
  bool profiling = false;
  hipEvent_t ev[2 * ST_COUNT] = {};
  bool ev_used[ST_COUNT] = {};
  std::string err;
};

// The offsets scan of stage a5 as a passenger of the depth sort (single-call forward): its three steps are run by extra
// workgroups appended to the sort's histogram / pass-0 / pass-1 launches -- each step only depends on the previous
// launch, and nothing on the tile-binning paths reads the offsets or the total before the frame ends -- so the scan
// costs no launches of its own (three launches of ~4.8 us each otherwise).
struct ggd_scan_piggy {
  const uint32_t* in = nullptr;   // tiles_touched[n]
  uint32_t* out = nullptr;        // point_offsets[n] (inclusive)
  int64_t n = 0;
  int nb = 0;                     // ggd_scan_blocks(n)
  uint32_t* block_sums = nullptr; // [nb] scratch
  uint32_t* d_total = nullptr;    // device word for the grand total (num_rendered)
  uint32_t* h_total = nullptr;    // device view of the pinned host word (may be NULL)
  unsigned long long* h_tagged = nullptr;   // device view of a pinned 64-bit word that receives (tag << 32 | total): the host
  uint32_t tag = 0;                         // polls it instead of waiting on an event (an event record between two kernels
                                            // costs the GPU a ~6 us bubble)
  // folded front end (ggd_fold): step 1 was done by the preprocess workgroups (256 points each) -- step 2 scans their sums
  const uint4* wg_info = nullptr;           // [n_info] {sum of tiles_touched, kept depth keys, ~min kept key, max kept key} per preprocess workgroup
  int n_info = 0;
  uint32_t* n_valid = nullptr;              // receives the number of kept keys (sum of wg_info[].y)
  int sum_stride = 1;                       // block_sums entries per scan block of step 3 (8 with wg_info: 2048 / 256)
  uint32_t* flat_flag = nullptr;            // folded front end: "the order is in the third pass's output" word of the consumers
  int spec_flat = 0;                        // != 0: only three passes were launched -- set *flat_flag whatever the histogram says
  uint32_t* fold_hist = nullptr;            // the folded front end's histogram replicas: the workgroup that runs step 2 also adds
                                            // replicas 1 .. REPS-1 of passes 1 .. 3 into replica 0 (only pass 0 reads them all)
  int msd = 0;                              // two-launch sort: fold_hist holds [GGD_MSD_BINS | 256] bins per replica, all summed into replica 0
  uint32_t msd_lo = 0;                      // ... its key window (ggd_fold)
  int msd_shift = GGD_MSD_SHIFT;
};

// The depth sort's histogram kernel folded into the preprocess kernel (single-call forward on the tile-binning path): every
// preprocess workgroup adds the digit counts of its kept depth keys to one of GGD_FOLD_REPS replicas of the four 256-bin
// histograms and stores {sum of tiles_touched, kept keys} of its 256 points -- the histogram launch (19 us at 1 M points:
// its 245 workgroups flush into the same 64 lines, and atomic instructions on one line serialise at ~43 ns per 16 lanes) and
// step 1 of the offsets scan disappear.  Control block (words): [REPS * 1024 histograms | 8 tickets | n_valid | flat | pad
// to 64 | status words of the 4 passes]; two blocks alternate, each cleared by the preprocess of the frame before its use.
constexpr int GGD_FOLD_REPS = 16;   // (32 / 16 / 8 replicas: 4114 / 4140 / 4150 frames per second at 1 M / 1024^2; 3907 workgroups over 8 would
                                    // keep one address busy 80 % of the kernel's time, 16 leaves a margin)
constexpr int GGD_FOLD_REP_STRIDE = GGD_MSD_BINS + 256;   // ordinary frames: the four byte histograms [p * 256 + digit] in the first
                                               // 1024 words; two-launch sort: [1024 buckets of the key window | 256 bins of the top byte]
// Two-launch depth sort (round 5; `msd` in ggd_fold / ggd_scan_piggy; the key WINDOW since round 6).  The kept keys of a frame lie
// in a narrow range of the 32-bit key space (fp32 bits of depths between, say, 1.8 and 3.6): over a window [lo, lo + 1024 << shift)
// that contains them they are ordered by ONE most-significant-digit partition and an in-LDS finish instead of three or four
// onesweep passes with their cross-tile look-back (48 -> see DESIGN.md):
//   launch 1  every 4096-key tile partitions ITS OWN keys by bucket = (key - lo) >> shift (1024 buckets, stable, culled keys
//             dropped), writes (key - lo, index) to its own region and a table {offset, count} per (tile, bucket) -- no dependency
//             between tiles at all;
//   launch 2  one 1024-thread workgroup per bucket gathers the bucket's pieces from all tiles in tile order (= index order),
//             orders them by the `shift` low bits of (key - lo) with (up to) two stable 8-bit counting passes in LDS and writes
//             the run of indices to its final place (bucket bases = prefix of the preprocess kernel's 1024-bin histogram).
// Round 5 fixed the window to one binade pair (bucket = key bits 14..23, precondition: a constant top byte).  That fails for a
// depth range that straddles 2.0 -- which the reference's pose sampler produces (camera.py:6-35: radius 2.7, yaw +- 1 rad: the
// unit cube's near corner comes to depth 1.83) -- and left a head-like scene's keys in 154 of the 1024 buckets.  Now the host
// fits the window to the ranges of the last GGD_MSD_WIN folded frames (+ 1/8 margin either side; every frame's kept-key min / max
// arrive with num_rendered) and the shift to the window (<= GGD_MSD_MAX_SHIFT: two 8-bit passes).  (2048 buckets were measured
// in round 5: launch 1 pays + 3 us for the wider digit, profiles/REJECTED.md.)
// The host speculates (after GGD_FLAT_STREAK folded frames); the frame's own front end verifies (no kept key outside the window:
// GGD_FOLD_OUTSIDE == 0; no bucket above GGD_MSD_CAP) and a frame that fails is binned and blended again by the ordinary path (as
// for the skipped fourth pass); a key outside the window is clamped into the last bucket, so the kernels behind always see a
// valid permutation.
// the 64 words behind the histograms: [8 tickets | n_valid | flat | pad | OUTSIDE (word 16, a line of its own): kept keys outside
// the two-launch sort's window | pad]
constexpr int GGD_FOLD_OUTSIDE = GGD_FOLD_REPS * GGD_FOLD_REP_STRIDE + 16;
constexpr int GGD_FOLD_ROWTOT = GGD_FOLD_REPS * GGD_FOLD_REP_STRIDE + 64;   // REPS x 64 words: entries per tile ROW (grids of <= 64
                                                                           // rows), for the row binning's first level
// (The kept keys' range of a frame -- what the two-launch sort's window is fitted to -- travels with the per-workgroup sums in
// wg_info, as plain stores.  Round 6's first forms kept {~min, max} replicas here, updated with atomicMax: 16 replicas packed into
// two 64-byte lines cost the preprocess kernel + 30 us (atomics on one line serialise at ~11 ns), one line per replica behind a
// read-and-compare still + 10 us: a device-scope load + atomic at the END of every workgroup adds 2 - 4 us to its ~8 us lifetime.)
constexpr int GGD_FOLD_HEAD = GGD_FOLD_ROWTOT + GGD_FOLD_REPS * 64;        // words in front of the status words
struct ggd_fold {
  uint32_t* ctl = nullptr;        // this frame's control block (clean)
  uint32_t* clear = nullptr;      // the other block ...
  uint32_t clear_words = 0;       // ... and how much of it the preprocess clears for the next frame
  uint4* wg_info = nullptr;       // [ceil(P / 256)] {sum of tiles_touched, kept keys, ~min kept key, max kept key}
  int rows = 0;                   // != 0: also count the Gaussians per tile row (the grid has <= 64 rows)
  int msd = 0;                    // != 0: histograms of the two-launch sort (1024 buckets of the key window, top byte) instead of the four bytes
  uint32_t msd_lo = 0;            // bucket of a kept key = min((key - msd_lo) >> msd_shift, 1023); a key for which the unclamped
  int msd_shift = GGD_MSD_SHIFT;  // value exceeds 1023 (below the window: the difference wraps) is counted in GGD_FOLD_OUTSIDE
};
// control block: [head | sort status words of the 4 passes | level-1 binning status words]
size_t ggd_fold_ctl_words(int64_t P);
size_t ggd_fold_l1_offset(int64_t P);   // first word of the level-1 status words

int ggd_fail(ggd_ctx* ctx, int code, const std::string& msg);
int ggd_reserve_scratch(ggd_ctx* ctx, size_t bytes, hipStream_t stream);

static inline const char* ggd_basename(const char* p) {
  const char* b = p;
  for (; *p; ++p) if (*p == '/') b = p + 1;
  return b;
}
#define GGD_HIP(call)                                                                                      \
  do {                                                                                                     \
    hipError_t e_ = (call);                                                                                \
    if (e_ != hipSuccess)                                                                                  \
      return ggd_fail(ctx, GGD_E_HIP, std::string(#call) + " (" + ggd_basename(__FILE__) + ":" +            \
                                          std::to_string(__LINE__) + "): " + hipGetErrorString(e_));         \
  } while (0)

struct StageTimer {  // RAII hipEvent pair around one pipeline stage (no-op unless profiling is on)
  ggd_ctx* c; int st; hipStream_t s;
  StageTimer(ggd_ctx* ctx, int stage, hipStream_t stream) : c(ctx), st(stage), s(stream) {
    if (c->profiling) { (void)hipEventRecord(c->ev[2 * st], s); }
  }
  ~StageTimer() {
    if (c->profiling) { (void)hipEventRecord(c->ev[2 * st + 1], s); c->ev_used[st] = true; }
  }
};

static inline size_t ggd_align(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// ---- kernels launched by the C API (defined in the .hip files) ------------------------------------------------
int ggd_launch_preprocess(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const float* means3D,
                          const float* shs, const float* colors_precomp, const float* opacities,
                          const float* scales, const float* rotations, const float* cov3D_precomp,
                          ggd_splat* splat, uint32_t* tiles_touched, uint8_t* clamped, int32_t* radii,
                          uint32_t* depth_keys, uint2* rect, uint32_t* trap_flag, uint32_t* zero_ptr = nullptr,
                          int zero_words = 0,    // zero_words words at zero_ptr are cleared by the first workgroups
                          const ggd_fold* fold = nullptr);
int ggd_launch_mark_visible(ggd_ctx* ctx, hipStream_t s, int P, const float* means3D, const float* view,
                            uint8_t* present);
// inclusive scan of a uint32 array; total written to *d_total (device)
int ggd_launch_inclusive_scan(ggd_ctx* ctx, hipStream_t s, const uint32_t* in, uint32_t* out, int64_t n,
                              uint32_t* d_total, void* tmp, size_t tmp_bytes);
size_t ggd_scan_tmp_bytes(int64_t n);
// same, plus: total also written to h_total (device view of a pinned host word)
int ggd_launch_inclusive_scan_ex(ggd_ctx* ctx, hipStream_t s, const uint32_t* in, uint32_t* out, int64_t n,
                                 uint32_t* d_total, void* tmp, size_t tmp_bytes, uint32_t* h_total);
int ggd_scan_blocks(int64_t n);   // workgroups (= block sums) of the scan over n elements
size_t ggd_sort_ctrl_words();   // words of the depth sort's control block (ggd_launch_sort32_iota's clean_ctl)
int ggd_launch_duplicate(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const uint2* rect,
                         const uint32_t* depth_keys, const uint32_t* tiles_touched, const uint32_t* offsets,
                         uint64_t* keys, uint32_t* vals);
size_t ggd_sort_tmp_bytes(int64_t n);
size_t ggd_sort32_tmp_bytes(int64_t n);
const uint32_t* ggd_sort32_nvalid_ptr(const void* ctl);  // device word: keys kept by ggd_launch_sort32_iota (ctl = its control block: clean_ctl or tmp)
// stable LSD radix sort of (key,val) pairs on key bits [0,nbits); result ends in (keys_a, vals_a); the input must
// have been placed in the buffer ggd_sort_input_is_alt(nbits) says.
int ggd_sort_input_is_alt(int nbits);
int ggd_launch_sort(ggd_ctx* ctx, hipStream_t s, uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b,
                    uint32_t* vals_b, int64_t n, int nbits, void* tmp, size_t tmp_bytes);
// 32-bit-key variant (depth sort of the Gaussians).  keys_src is only read; the result ends in (keys_a, vals_a);
// values start as the identity permutation.  nbits must be a multiple of 16 (even number of passes).
int ggd_launch_sort32_iota(ggd_ctx* ctx, hipStream_t s, const uint32_t* keys_src, uint32_t* keys_a, uint32_t* vals_a,
                           uint32_t* keys_b, uint32_t* vals_b, int64_t n, int nbits, void* tmp, size_t tmp_bytes,
                           uint32_t* clean_ctl = nullptr, const ggd_scan_piggy* piggy = nullptr,
                           bool flag_flat_last = false, bool apply_here = true, const ggd_fold* fold = nullptr,
                           bool skip_last = false);
// flag_flat_last: a constant-digit LAST pass copies nothing -- it sets the word ggd_sort32_flat_ptr(ctl) and the result
//                 stays in (keys_b, vals_b)
// apply_here = false: the caller runs the riding scan's last step (offsets) elsewhere -- ggd_launch_rowbin(apply = ...)
// fold: histograms, kept-key count and the scan's step 1 come from the preprocess kernel (piggy must carry wg_info); no
//       histogram launch.  The workgroup that runs the scan's step 2 reports "top digit constant" in bit 63 of the tagged
//       num_rendered word and in d_total[2].
// skip_last (fold only): the last pass is not launched (the caller expects a constant top digit); the consumers' flat word is
//       set regardless, so that they read the third pass's output (a valid permutation either way)
constexpr int GGD_FLAT_STREAK = 8;
const uint32_t* ggd_sort32_flat_ptr(const void* ctl);
size_t ggd_sort32_msd_table_bytes(int64_t n);
bool ggd_sort32_msd_supported(int64_t n);
int ggd_launch_sort32_msd(ggd_ctx* ctx, hipStream_t s, const uint32_t* keys_src, uint32_t* keys_a, uint32_t* vals_a,
                          uint32_t* keys_b, uint32_t* vals_b, int64_t n, uint32_t* table, const ggd_scan_piggy* piggy,
                          const ggd_fold* fold);
const uint32_t* ggd_fold_nvalid_ptr(const uint32_t* fold_ctl);   // the same two words of a folded front end's control block
const uint32_t* ggd_fold_flat_ptr(const uint32_t* fold_ctl);
// Tile binning (GGD_OPT_BINNING = 1): sorted Gaussian order -> per-tile lists + ranges.
bool ggd_rowbin_supported(int W, int H);
size_t ggd_rowbin_tmp_bytes(int P, uint32_t capacity, int W, int H);
int ggd_launch_rowbin(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const uint2* rect, const uint32_t* order,
                      const uint32_t* n_vis_ptr, uint32_t* list, uint32_t* ranges, uint32_t capacity, void* tmp,
                      size_t tmp_bytes, const uint32_t* order_alt = nullptr, const uint32_t* use_alt = nullptr,
                      const ggd_scan_piggy* apply = nullptr, const uint32_t* fold_rowtot = nullptr, uint32_t* fold_status1 = nullptr);   // apply: step 3 of a riding scan, as appended workgroups of the
                                                                 // last (longest) binning launch
                      // *use_alt != 0 (device): the depth order is in order_alt (see ggd_launch_sort32_iota)
                      // fold_rowtot / fold_status1 (folded front end, grids of <= 64 x 64 tiles): the entries per tile row were
                      // counted by the preprocess kernel -- level 1 is ONE launch (count, look-back over the chunks, scatter)
int ggd_launch_ranges(ggd_ctx* ctx, hipStream_t s, const uint64_t* keys, int64_t n, uint32_t* ranges, int T);
int ggd_launch_blend(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const ggd_splat* splat,
                     const uint32_t* list, const uint32_t* ranges, uint32_t capacity, float* out_color, float* final_T,
                     uint32_t* n_contrib);
// Backward accumulator record (library scratch, one per Gaussian, zeroed with ONE memset): the blend backward adds its
// per-(tile, Gaussian) sums here, the per-Gaussian backward reads it once and writes every caller-visible gradient
// (zeros for culled Gaussians), so the caller's 8 gradient arrays need no zero-fill passes.
constexpr int GGD_ACC_FLOATS = 12;   // 48 B: conic A,B,C | opacity | mean2D x,y | colour r,g,b | 3 pad
constexpr int GGD_ACC_CONIC = 0, GGD_ACC_OPACITY = 3, GGD_ACC_MEAN2D = 4, GGD_ACC_COLOR = 6;
int ggd_launch_blend_backward(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const ggd_splat* splat,
                              const uint32_t* list, const uint32_t* ranges, const float* final_T,
                              const uint32_t* n_contrib, const float* dL_dpix, float* grad_acc /*[P][GGD_ACC_FLOATS]*/);
int ggd_launch_preprocess_backward(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const float* means3D,
                                   const float* shs, const float* colors_precomp, const float* opacities,
                                   float* dL_dopacity, const float* scales,
                                   const float* rotations, const float* cov3D_precomp, const int32_t* radii,
                                   const uint8_t* clamped, const float* grad_acc, float* dL_dmean2D,
                                   float* dL_dcolors, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                                   float* dL_dscales, float* dL_drots);

// ---- device helpers -----------------------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ int ggd_tile_rect(float px, float py, int irad, int gx, int gy, int& minx, int& miny,
                                             int& maxx, int& maxy) {
  // C-style truncation of (p -/+ r [+15]) / 16, clamped to the tile grid.  float->int saturates on gfx950.
  const float fr = (float)irad;
  minx = min(gx, max(0, (int)((px - fr) / 16.0f)));
  miny = min(gy, max(0, (int)((py - fr) / 16.0f)));
  maxx = min(gx, max(0, (int)((px + fr + 15.0f) / 16.0f)));
  maxy = min(gy, max(0, (int)((py + fr + 15.0f) / 16.0f)));
  return (maxx - minx) * (maxy - miny);
}

// Nine wave sums at once (the blend backward's per-Gaussian partials), totals valid in lane 63.  Same six DPP steps as
// ggd_wave_sum_to63, written as v_add_f32_dpp so that each step is ONE instruction per value (through the builtin the
// compiler emits v_mov_b32_dpp + a packed add: 1.5 per value).  The nine chains are interleaved step by step, which
// also provides the wait states a DPP read of a freshly written VGPR needs (the hazard recogniser does not look inside
// inline asm); the leading s_nop covers the producers of the inputs.
#define GGD_DPP9(ctrl)                                                                                          \
  asm volatile("v_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\t"                      \
               "v_add_f32_dpp %2, %2, %2 " ctrl "\n\tv_add_f32_dpp %3, %3, %3 " ctrl "\n\t"                      \
               "v_add_f32_dpp %4, %4, %4 " ctrl "\n\tv_add_f32_dpp %5, %5, %5 " ctrl "\n\t"                      \
               "v_add_f32_dpp %6, %6, %6 " ctrl "\n\tv_add_f32_dpp %7, %7, %7 " ctrl "\n\t"                      \
               "v_add_f32_dpp %8, %8, %8 " ctrl                                                                 \
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), \
                 "+v"(v[8]))
__device__ __forceinline__ void ggd_wave_sum9_to63(float (&v)[9]) {
  asm volatile("s_nop 1");
  GGD_DPP9("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0");
  GGD_DPP9("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0");
  GGD_DPP9("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0");
  GGD_DPP9("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0");
  GGD_DPP9("row_bcast:15 row_mask:0xa bank_mask:0xf");
  GGD_DPP9("row_bcast:31 row_mask:0xc bank_mask:0xf");
  asm volatile("s_nop 1");
}
#undef GGD_DPP9

// Sum over the 64 lanes of a wave with DPP row shifts / broadcasts; the total is valid in lane 63.
// (Hillis-Steele inclusive scan inside each 16-lane row, then row_bcast:15 into rows 1/3 and row_bcast:31 into
// rows 2/3; lanes without a valid source add the `old` operand = 0.)
#define GGD_DPP_ADD(v, ctrl, rowmask)                                                                         \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rowmask, 0xf, false))
__device__ __forceinline__ float ggd_wave_sum_to63(float v) {
  GGD_DPP_ADD(v, 0x111, 0xf);  // row_shr:1
  GGD_DPP_ADD(v, 0x112, 0xf);  // row_shr:2
  GGD_DPP_ADD(v, 0x114, 0xf);  // row_shr:4
  GGD_DPP_ADD(v, 0x118, 0xf);  // row_shr:8
  GGD_DPP_ADD(v, 0x142, 0xa);  // row_bcast:15 -> rows 1, 3
  GGD_DPP_ADD(v, 0x143, 0xc);  // row_bcast:31 -> rows 2, 3
  return v;
}
#endif
