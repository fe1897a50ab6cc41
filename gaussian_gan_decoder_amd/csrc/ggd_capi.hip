// ggd_capi.hip -- the C ABI declared in include/ggd_raster.h (host side: ctx, workspace, stage sequencing).
//
// Stage sequence of one forward: preprocess -> inclusive scan -> read back R ->
// duplicateWithKeys -> stable radix sort -> tile ranges -> blend.  The one host sync is the read-back of
// R = num_rendered, exactly where the CUDA original has it; everything else is enqueued on the caller's stream.
#include <stdlib.h>
#include <string.h>

#include <new>

#include "ggd_common.h"

static std::string g_create_error;

static const char* const kStageNames[ST_COUNT] = {"preprocess", "scan", "readback", "duplicate", "sort",
                                                  "ranges", "blend", "blend_bwd", "preprocess_bwd"};

int ggd_fail(ggd_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg; else g_create_error = msg;
  return code;
}

int ggd_reserve_scratch(ggd_ctx* ctx, size_t bytes, hipStream_t stream) {
  if (bytes <= ctx->scratch_bytes) return GGD_OK;
  // grow-only with head-room so steady-state training never reallocates
  size_t want = bytes + bytes / 4 + (1u << 20);
  if (ctx->scratch) {
    GGD_HIP(hipStreamSynchronize(stream));
    GGD_HIP(hipFree(ctx->scratch));
    ctx->scratch = nullptr; ctx->scratch_bytes = 0;
  }
  hipError_t e = hipMalloc(&ctx->scratch, want);
  if (e != hipSuccess) return ggd_fail(ctx, GGD_E_NOMEM, std::string("hipMalloc scratch: ") + hipGetErrorString(e));
  ctx->scratch_bytes = want;
  return GGD_OK;
}

// ---- layouts -------------------------------------------------------------------------------------------------
extern "C" int ggd_geom_layout(int32_t P, ggd_geom_view* v) {
  if (!v || P < 0) return GGD_E_INVALID;
  size_t off = 0;
  v->splat = off; off += ggd_align((size_t)P * sizeof(ggd_splat));
  v->tiles_touched = off; off += ggd_align((size_t)P * sizeof(uint32_t));
  v->point_offsets = off; off += ggd_align((size_t)P * sizeof(uint32_t));
  v->clamped = off; off += ggd_align((size_t)P);
  v->depth_keys = off; off += ggd_align((size_t)P * sizeof(uint32_t));
  v->rect = off; off += ggd_align((size_t)P * 2 * sizeof(uint32_t));
  v->header = off; off += (P > 0 ? 256 : 0);
  v->total = off;
  return GGD_OK;
}
extern "C" int ggd_binning_layout(int64_t R, ggd_binning_view* v) {
  if (!v || R < 0) return GGD_E_INVALID;
  size_t off = 0;
  // the sorted list comes first: it is the only part the backward reads, so its position does not depend on the R
  // the buffer was laid out for (a capacity in the single-call forward, num_rendered in the two-call form)
  v->list = off; off += ggd_align((size_t)R * sizeof(uint32_t));
  v->list_alt = off; off += ggd_align((size_t)R * sizeof(uint32_t));
  v->keys = off; off += ggd_align((size_t)R * sizeof(uint64_t));
  v->keys_alt = off; off += ggd_align((size_t)R * sizeof(uint64_t));
  v->total = off;
  return GGD_OK;
}
extern "C" int ggd_img_layout(int32_t W, int32_t H, ggd_img_view* v) {
  if (!v || W < 0 || H < 0) return GGD_E_INVALID;
  const size_t T = (size_t)((W + 15) / 16) * ((H + 15) / 16);
  size_t off = 0;
  v->ranges = off; off += ggd_align(T * 2 * sizeof(uint32_t));
  v->final_T = off; off += ggd_align((size_t)W * H * sizeof(float));
  v->n_contrib = off; off += ggd_align((size_t)W * H * sizeof(uint32_t));
  v->total = off;
  return GGD_OK;
}
extern "C" size_t ggd_geom_bytes(int32_t P) { ggd_geom_view v; return ggd_geom_layout(P, &v) == GGD_OK ? v.total : 0; }
extern "C" size_t ggd_binning_bytes(int64_t R) { ggd_binning_view v; return ggd_binning_layout(R, &v) == GGD_OK ? v.total : 0; }
extern "C" size_t ggd_img_bytes(int32_t W, int32_t H) { ggd_img_view v; return ggd_img_layout(W, H, &v) == GGD_OK ? v.total : 0; }

static uint32_t higher_msb(uint32_t n) {  // smallest k with (n >> k) == 0, by bisection from 16
  uint32_t msb = 16, step = 16;
  while (step > 1) {
    step /= 2;
    if (n >> msb) msb += step; else msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}
extern "C" int ggd_sort_bits(int32_t W, int32_t H) {
  const uint32_t T = (uint32_t)((W + 15) / 16) * (uint32_t)((H + 15) / 16);
  return 32 + (int)higher_msb(T);
}

// auto (GGD_OPT_BINNING = 1): the row / column binning path whenever the grid allows it.  It used to start at 0.79 M
// instances (the fixed cost of its launches); since the single-call forward stopped waiting for the end of the frame it
// wins at every size measured (18 k instances: 95 vs 123 us per frame, 170 k: 120 vs 158, 1.25 M: 207 vs 296).
constexpr int64_t GGD_ROWBIN_MIN_R = 1;

// ---- ctx -----------------------------------------------------------------------------------------------------
extern "C" const char* ggd_version(void) { return "ggd-raster 0.1 (gfx950)"; }
extern "C" int ggd_stage_count(void) { return ST_COUNT; }
extern "C" const char* ggd_stage_name(int s) { return (s >= 0 && s < ST_COUNT) ? kStageNames[s] : ""; }

extern "C" ggd_ctx* ggd_create(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    ggd_fail(nullptr, GGD_E_NODEVICE, "no HIP device " + std::to_string(device) + " visible");
    return nullptr;
  }
  ggd_ctx* ctx = new (std::nothrow) ggd_ctx();
  if (!ctx) { ggd_fail(nullptr, GGD_E_NOMEM, "out of host memory"); return nullptr; }
  ctx->device = device;
  if (const char* e = getenv("GGD_EXP_MODE")) { const int v = atoi(e); if (v >= 0 && v <= 3) ctx->opt[GGD_OPT_EXP_MODE] = v; }
  if (const char* e = getenv("GGD_BINNING")) { const int v = atoi(e); if (v >= 0 && v <= 3) ctx->opt[GGD_OPT_BINNING] = v; }
  if (const char* e = getenv("GGD_BLEND_SPLIT")) { const int v = atoi(e); if (v >= 0 && v <= 4) ctx->opt[GGD_OPT_BLEND_SPLIT] = v; }
  if (const char* e = getenv("GGD_BLEND_CULL")) ctx->opt[GGD_OPT_BLEND_CULL] = atoi(e) != 0;
  if (const char* e = getenv("GGD_FOLD")) ctx->opt[GGD_OPT_FOLD] = atoi(e) != 0;
  if (const char* e = getenv("GGD_MSD_SORT")) ctx->opt[GGD_OPT_MSD_SORT] = atoi(e) != 0;
  if (const char* e = getenv("GGD_MSD_BUCKETS")) { const int v = atoi(e); if (v >= 16 && v <= GGD_MSD_BINS) ctx->msd_buckets = v; }   // timing experiments
  int prev = 0;
  (void)hipGetDevice(&prev);
  bool ok = hipSetDevice(device) == hipSuccess &&
            hipMalloc((void**)&ctx->d_words, 256) == hipSuccess &&
            // coherent (fine-grained) pinned memory: the tagged num_rendered word must become visible to the polling host
            // while the kernel that stored it is still running, whatever HIP_HOST_COHERENT says
            hipHostMalloc((void**)&ctx->h_words, 64, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess &&
            hipMemset(ctx->d_words, 0, 256) == hipSuccess;
  // stale pinned memory (e.g. of a destroyed context) must never match a tag: r_tag restarts at 1 for every context
  if (ok) memset(ctx->h_words, 0, 64);
  for (int i = 0; ok && i < 2 * ST_COUNT; ++i) ok = hipEventCreate(&ctx->ev[i]) == hipSuccess;
  // device-side view of the pinned mirror (the scan writes num_rendered there itself: no blit for the read-back) and the
  // depth sort's control block in its own allocation (cleared by the scan: no memset launch in front of the sort);
  // both are optional -- without them the forward falls back to the copy / the memset
  if (ok && hipHostGetDevicePointer((void**)&ctx->h_words_dev, ctx->h_words, 0) != hipSuccess) {
    ctx->h_words_dev = nullptr;
    (void)hipGetLastError();
  }
  if (ok && (hipMalloc((void**)&ctx->sortctl, ggd_sort_ctrl_words() * sizeof(uint32_t)) != hipSuccess ||
             hipMemset(ctx->sortctl, 0, ggd_sort_ctrl_words() * sizeof(uint32_t)) != hipSuccess)) {
    ctx->sortctl = nullptr;
    (void)hipGetLastError();
  }
  (void)hipSetDevice(prev);
  if (!ok) {
    ggd_fail(nullptr, GGD_E_HIP, std::string("ggd_create: ") + hipGetErrorString(hipGetLastError()));
    ggd_destroy(ctx);
    return nullptr;
  }
  return ctx;
}

extern "C" void ggd_destroy(ggd_ctx* ctx) {
  if (!ctx) return;
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->d_words) (void)hipFree(ctx->d_words);
  if (ctx->h_words) (void)hipHostFree(ctx->h_words);
  if (ctx->sortctl) (void)hipFree(ctx->sortctl);
  for (int b = 0; b < 2; ++b) if (ctx->foldctl[b]) (void)hipFree(ctx->foldctl[b]);
  if (ctx->gelu_tables) (void)hipFree(ctx->gelu_tables);
  if (ctx->scan_sums) (void)hipFree(ctx->scan_sums);
  if (ctx->stats_buf) (void)hipFree(ctx->stats_buf);
  if (ctx->dbg_keys) (void)hipFree(ctx->dbg_keys);
  if (ctx->dbg_vals) (void)hipFree(ctx->dbg_vals);
  for (int i = 0; i < 2 * ST_COUNT; ++i)
    if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
  delete ctx;
}

extern "C" const char* ggd_last_error(ggd_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int ggd_set_option(ggd_ctx* ctx, int option, int value) {
  if (!ctx) return GGD_E_INVALID;
  static const int kMax[GGD_OPT_COUNT] = {3, 1, 3, 4, 1, 1};
  if (option < 0 || option >= GGD_OPT_COUNT || value < 0 || value > kMax[option])
    return ggd_fail(ctx, GGD_E_INVALID, "ggd_set_option: unknown option or value");
  ctx->opt[option] = value;
  if (option == GGD_OPT_MSD_SORT || option == GGD_OPT_FOLD) {
    // (re)setting the sort options restarts the cross-frame speculation state: the window of recent key ranges, the pause after
    // a miss, the flat-frame streak -- a caller (or a test) that switches forms gets a defined starting point
    ctx->win_n = 0; ctx->win_pos = 0; ctx->msd_ban = 0; ctx->msd_oversize_streak = 0; ctx->flat_streak = 0;
  }
  return GGD_OK;
}
extern "C" int ggd_blend_stats(ggd_ctx* ctx, int enable, unsigned long long* out) {
  if (!ctx) return GGD_E_INVALID;
  GGD_HIP(hipDeviceSynchronize());
  if (out && ctx->blend_stats) GGD_HIP(hipMemcpy(out, ctx->stats_buf, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (enable) {
    const size_t bytes = (GGD_STATS_HEAD + 3ull * GGD_STATS_MAX_WAVES) * sizeof(unsigned long long);
    if (!ctx->stats_buf) GGD_HIP(hipMalloc((void**)&ctx->stats_buf, bytes));
    GGD_HIP(hipMemset(ctx->stats_buf, 0, bytes));
    if (enable == 2) {   // per-wave timeline instead of the counters
      const unsigned long long one = 1ull;
      GGD_HIP(hipMemcpy(ctx->stats_buf + GGD_STATS_MODE, &one, sizeof(one), hipMemcpyHostToDevice));
    }
    ctx->blend_stats = ctx->stats_buf;
  } else {
    ctx->blend_stats = nullptr;
  }
  return GGD_OK;
}
extern "C" int ggd_blend_backward_stats(ggd_ctx* ctx, unsigned long long* out) {
  if (!ctx || !out) return GGD_E_INVALID;
  if (!ctx->stats_buf) return ggd_fail(ctx, GGD_E_INVALID, "ggd_blend_backward_stats: statistics were never enabled (ggd_blend_stats)");
  GGD_HIP(hipDeviceSynchronize());
  GGD_HIP(hipMemcpy(out, ctx->stats_buf + GGD_STATS_BWD, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return GGD_OK;
}
extern "C" int ggd_blend_timeline(ggd_ctx* ctx, unsigned long long* out, int waves) {
  if (!ctx || !out || waves < 0 || waves > GGD_STATS_MAX_WAVES) return GGD_E_INVALID;
  if (!ctx->stats_buf) return ggd_fail(ctx, GGD_E_INVALID, "ggd_blend_timeline: statistics were never enabled");
  GGD_HIP(hipDeviceSynchronize());
  GGD_HIP(hipMemcpy(out, ctx->stats_buf + GGD_STATS_HEAD, 3ull * waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return GGD_OK;
}

extern "C" int ggd_get_option(ggd_ctx* ctx, int option) {
  if (ctx && option == GGD_STAT_FLAT_STREAK) return ctx->flat_streak;
  if (ctx && option == GGD_STAT_SORT_RERUNS) return (int)(ctx->spec3_misses & 0x7fffffffull);
  if (ctx && option == GGD_STAT_MSD_FRAMES) return (int)(ctx->msd_frames & 0x7fffffffull);
  if (!ctx || option < 0 || option >= GGD_OPT_COUNT) return GGD_E_INVALID;
  return ctx->opt[option];
}

extern "C" int ggd_set_profiling(ggd_ctx* ctx, int enabled) {
  if (!ctx) return GGD_E_INVALID;
  ctx->profiling = enabled != 0;
  for (int i = 0; i < ST_COUNT; ++i) ctx->ev_used[i] = false;
  return GGD_OK;
}

extern "C" int ggd_stage_times(ggd_ctx* ctx, float* ms_out) {
  if (!ctx || !ms_out) return GGD_E_INVALID;
  for (int i = 0; i < ST_COUNT; ++i) {
    ms_out[i] = -1.0f;
    if (ctx->ev_used[i]) {
      GGD_HIP(hipEventSynchronize(ctx->ev[2 * i + 1]));
      float ms = 0.0f;
      GGD_HIP(hipEventElapsedTime(&ms, ctx->ev[2 * i], ctx->ev[2 * i + 1]));
      ms_out[i] = ms;
    }
  }
  return GGD_OK;
}

static int check_params(ggd_ctx* ctx, const ggd_params* prm) {
  if (!ctx) return GGD_E_INVALID;
  if (!prm) return ggd_fail(ctx, GGD_E_INVALID, "params is NULL");
  if (prm->P < 0 || prm->width <= 0 || prm->height <= 0)
    return ggd_fail(ctx, GGD_E_INVALID, "bad P / image size");
  if (!prm->viewmatrix || !prm->projmatrix || !prm->campos || !prm->bg)
    return ggd_fail(ctx, GGD_E_INVALID, "viewmatrix / projmatrix / campos / bg must be device pointers");
  if (prm->sh_degree < 0 || prm->sh_degree > 3) return ggd_fail(ctx, GGD_E_INVALID, "sh_degree must be 0..3");
  if (prm->tanfovx == 0.0f || prm->tanfovy == 0.0f) return ggd_fail(ctx, GGD_E_INVALID, "tanfov must be non-zero");
  return GGD_OK;
}

static int check_inputs(ggd_ctx* ctx, const ggd_params* prm, const float* means3D, const float* shs,
                        const float* colors_precomp, const float* scales, const float* rotations,
                        const float* cov3D_precomp) {
  if (prm->P == 0) return GGD_OK;
  if (!means3D) return ggd_fail(ctx, GGD_E_INVALID, "means3D is NULL");
  if ((shs == nullptr) == (colors_precomp == nullptr))
    return ggd_fail(ctx, GGD_E_INVALID, "Please provide excatly one of either SHs or precomputed colors!");
  if (((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr) ||
      ((scales == nullptr) != (rotations == nullptr)))
    return ggd_fail(ctx, GGD_E_INVALID,
                    "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
  if (shs && (prm->sh_degree + 1) * (prm->sh_degree + 1) > prm->M)
    return ggd_fail(ctx, GGD_E_INVALID, "shs holds fewer coefficients than sh_degree needs");
  return GGD_OK;
}

// ---- forward ---------------------------------------------------------------------------------------------------
// The two-launch depth sort's key window for the next frame: the union of the kept-key ranges of the last folded frames, a margin
// of 1/8 of its width either side, and the smallest shift that spreads it over at most ctx->msd_buckets buckets.  False when no
// range is known yet or the window is too wide for two 8-bit finishing passes.
static bool msd_fit_window(const ggd_ctx* ctx, uint32_t* lo_out, int* shift_out) {
  const int n = ctx->win_n < GGD_MSD_WIN ? ctx->win_n : GGD_MSD_WIN;
  if (n <= 0) return false;
  uint32_t lo = 0xffffffffu, hi = 0u;
  for (int i = 0; i < n; ++i) { lo = ctx->win_lo[i] < lo ? ctx->win_lo[i] : lo; hi = ctx->win_hi[i] > hi ? ctx->win_hi[i] : hi; }
  if (lo > hi) return false;
  const uint64_t margin = ((uint64_t)(hi - lo) >> 3) + 4096u;
  const uint64_t wlo = (uint64_t)lo > margin ? (uint64_t)lo - margin : 0u;
  uint64_t whi = (uint64_t)hi + margin;
  if (whi > 0xfffffffeull) whi = 0xfffffffeull;
  const uint64_t span = whi - wlo;            // bucket of the largest key = span >> shift, must stay below the bucket count
  int shift = 0;
  while ((span >> shift) >= (uint64_t)ctx->msd_buckets) ++shift;
  if (shift > GGD_MSD_MAX_SHIFT) return false;
  *lo_out = (uint32_t)wlo; *shift_out = shift;
  return true;
}

static const char* const kPendingMsg = "a frame enqueued with ggd_forward_enqueue is pending on this context: call ggd_forward_collect first";
// Enqueue the per-Gaussian kernels + scan + the asynchronous read-back of {R, prefilter trap}; no host sync.
static int geometry_enqueue(ggd_ctx* ctx, void* stream, const ggd_params* prm, const float* means3D,
                            const float* shs, const float* colors_precomp, const float* opacities,
                            const float* scales, const float* rotations, const float* cov3D_precomp,
                            void* geom_buf, int32_t* radii, int64_t* num_rendered, bool defer_scan = false) {
  (void)hipGetLastError();   // a sticky error another library left in this thread is not ours to report
  int rc = check_params(ctx, prm);
  if (rc != GGD_OK) return rc;
  rc = check_inputs(ctx, prm, means3D, shs, colors_precomp, scales, rotations, cov3D_precomp);
  if (rc != GGD_OK) return rc;
  if (!num_rendered) return ggd_fail(ctx, GGD_E_INVALID, "num_rendered is NULL");
  *num_rendered = 0;
  ctx->r_pending = false;      // (a previous call that failed half way may have left these set)
  ctx->scan_deferred = false;
  if (prm->P == 0) return GGD_OK;
  if (!geom_buf || !radii || !opacities) return ggd_fail(ctx, GGD_E_INVALID, "geom_buf / radii / opacities is NULL");
  if (prm->raw_attributes && cov3D_precomp)
    return ggd_fail(ctx, GGD_E_INVALID, "raw_attributes needs scales/rotations (no cov3D_precomp)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ggd_geom_view gv;
  ggd_geom_layout(prm->P, &gv);
  char* gb = static_cast<char*>(geom_buf);
  ggd_splat* splat = reinterpret_cast<ggd_splat*>(gb + gv.splat);
  uint32_t* tiles = reinterpret_cast<uint32_t*>(gb + gv.tiles_touched);
  uint32_t* offsets = reinterpret_cast<uint32_t*>(gb + gv.point_offsets);
  uint8_t* clamped = reinterpret_cast<uint8_t*>(gb + gv.clamped);
  uint32_t* depth_keys = reinterpret_cast<uint32_t*>(gb + gv.depth_keys);
  uint2* rect = reinterpret_cast<uint2*>(gb + gv.rect);

  const size_t scan_tmp = ggd_scan_tmp_bytes(prm->P);
  rc = ggd_reserve_scratch(ctx, scan_tmp, s);
  if (rc != GGD_OK) return rc;
  if (prm->prefiltered) GGD_HIP(hipMemsetAsync(ctx->d_words + 1, 0, sizeof(uint32_t), s));
  // single-call forward on a tile-binning path: the scan (offsets + num_rendered) rides on the depth sort's launches, and the
  // sort's histograms + the scan's first step are produced by the preprocess kernel itself (ggd_fold)
  const bool ride = defer_scan && ctx->h_words_dev;
  ggd_fold fold;
  ctx->fold_active = false;
  ctx->msd_frame = false;
  if (ride && ctx->opt[GGD_OPT_FOLD] != 0) {
    const int nwg = (prm->P + 255) / 256;
    if (ctx->scan_sums_cap < 5 * nwg + 64) {   // [exclusive prefixes: nwg words | {sum, kept, ~min key, max key}: nwg uint4]
      if (ctx->scan_sums) (void)hipFree(ctx->scan_sums);
      ctx->scan_sums = nullptr; ctx->scan_sums_cap = 0;
      const int cap = 5 * (nwg + nwg / 2) + 64;
      GGD_HIP(hipMalloc((void**)&ctx->scan_sums, (size_t)cap * sizeof(uint32_t)));
      ctx->scan_sums_cap = cap;
    }
    const size_t need = ggd_fold_ctl_words(prm->P);
    if (ctx->foldctl_cap < need) {   // (grow-only; both blocks start clean)
      GGD_HIP(hipStreamSynchronize(s));
      for (int b = 0; b < 2; ++b) { if (ctx->foldctl[b]) (void)hipFree(ctx->foldctl[b]); ctx->foldctl[b] = nullptr; }
      ctx->foldctl_cap = 0;
      const size_t cap = need + need / 2;
      for (int b = 0; b < 2; ++b) {
        GGD_HIP(hipMalloc((void**)&ctx->foldctl[b], cap * sizeof(uint32_t)));
        GGD_HIP(hipMemsetAsync(ctx->foldctl[b], 0, cap * sizeof(uint32_t), s));
        ctx->foldctl_dirty[b] = 0;
      }
      ctx->foldctl_cap = cap;
      ctx->fold_cur = 0;
    }
    if (ctx->fold_poisoned) {
      for (int b = 0; b < 2; ++b) {
        GGD_HIP(hipMemsetAsync(ctx->foldctl[b], 0, ctx->foldctl_cap * sizeof(uint32_t), s));
        ctx->foldctl_dirty[b] = 0;
      }
      ctx->fold_poisoned = false;
    }
    const int cur = ctx->fold_cur, oth = cur ^ 1;
    fold.ctl = ctx->foldctl[cur];
    fold.clear = ctx->foldctl[oth];
    fold.clear_words = (uint32_t)ctx->foldctl_dirty[oth];
    fold.wg_info = reinterpret_cast<uint4*>(ctx->scan_sums + (((size_t)nwg + 3) & ~(size_t)3));
    fold.rows = ((prm->width + 15) / 16 <= 64 && (prm->height + 15) / 16 <= 64) ? 1 : 0;
    // two-launch depth sort: speculated once the key ranges of GGD_FLAT_STREAK folded frames are known (render_enqueue: folded,
    // tile binning, speculative), decided HERE because it selects the histograms this launch builds
    ctx->msd_frame = ctx->opt[GGD_OPT_MSD_SORT] != 0 && ctx->opt[GGD_OPT_FOLD] == 1 && ctx->win_n >= GGD_FLAT_STREAK &&
                     ctx->msd_ban == 0 && ggd_sort32_msd_supported(prm->P) && msd_fit_window(ctx, &ctx->msd_lo, &ctx->msd_shift);
    fold.msd = ctx->msd_frame ? 1 : 0;
    fold.msd_lo = ctx->msd_lo; fold.msd_shift = ctx->msd_shift;
    ctx->foldctl_dirty[oth] = 0;       // (clean once this launch has run)
    ctx->foldctl_dirty[cur] = need;    // what this frame may write
    ctx->fold_cur = oth;
    ctx->fold_active = true;
  }
  {
    StageTimer t(ctx, ST_PREPROCESS, s);
    const bool old_ctl = !ctx->fold_active && ctx->sortctl;
    rc = ggd_launch_preprocess(ctx, s, *prm, means3D, shs, colors_precomp, opacities, scales, rotations,
                               cov3D_precomp, splat, tiles, shs ? clamped : nullptr, radii, depth_keys, rect,
                               ctx->d_words + 1, old_ctl ? ctx->sortctl : nullptr, old_ctl ? (int)ggd_sort_ctrl_words() : 0,
                               ctx->fold_active ? &fold : nullptr);
    if (rc != GGD_OK) { ctx->fold_poisoned = ctx->fold_active; ctx->fold_active = false; return rc; }
    ctx->sortctl_clean = old_ctl;
  }
  ctx->scan_deferred = false;
  if (ride) {
    const int nb = ggd_scan_blocks(prm->P);
    if (!ctx->fold_active && ctx->scan_sums_cap < nb) {
      if (ctx->scan_sums) (void)hipFree(ctx->scan_sums);
      ctx->scan_sums = nullptr; ctx->scan_sums_cap = 0;
      GGD_HIP(hipMalloc((void**)&ctx->scan_sums, (size_t)(nb + nb / 2 + 64) * sizeof(uint32_t)));
      ctx->scan_sums_cap = nb + nb / 2 + 64;
    }
    ctx->scan_deferred = true;
    if (prm->prefiltered)
      GGD_HIP(hipMemcpyAsync(ctx->h_words + 1, ctx->d_words + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    return GGD_OK;
  }
  {
    StageTimer t(ctx, ST_SCAN, s);
    rc = ggd_launch_inclusive_scan_ex(ctx, s, tiles, offsets, prm->P, ctx->d_words, ctx->scratch, ctx->scratch_bytes,
                                      ctx->h_words_dev);
    if (rc != GGD_OK) return rc;
  }
  {
    StageTimer t(ctx, ST_READBACK, s);
    // with the device view of the pinned mirror the scan has already delivered R; only the prefilter trap word is copied
    if (!ctx->h_words_dev)
      GGD_HIP(hipMemcpyAsync(ctx->h_words, ctx->d_words, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    else if (prm->prefiltered)
      GGD_HIP(hipMemcpyAsync(ctx->h_words + 1, ctx->d_words + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  }
  return GGD_OK;
}

// Wait for the stream and publish R (the one host sync of a forward).
static int geometry_finish(ggd_ctx* ctx, void* stream, const ggd_params* prm, int64_t* num_rendered) {
  // The host needs num_rendered, not the finished frame: in the single-call forward it waits for the word written by the
  // launch that delivers R (early in the depth sort) and returns while binning and blend are still running -- as
  // upstream returns with its render kernels in flight.  Outputs are ordered on the caller's stream as usual.
  if (ctx->r_pending) {
    ctx->r_pending = false;
    // the launch that carries the scan's block-sum step stores (tag << 32 | R) into the pinned mirror; poll for this
    // call's tag (an event record behind that launch would cost the GPU a ~6 us bubble between two kernels).  Every few
    // thousand polls the stream is queried: if it has drained without the tag (a failed launch), fall back to the copy.
    volatile unsigned long long* slot = reinterpret_cast<volatile unsigned long long*>(ctx->h_words + 2);
    const unsigned long long want = ctx->r_tag;   // (30 bits; bit 63 of the word: this frame's top depth digit is constant,
                                                  // bit 62: the two-launch sort's histograms say it was valid for this frame)
    unsigned long long v = *slot;
    for (unsigned spins = 0; ((v >> 32) & 0x3fffffffull) != want; ++spins) {
      __builtin_ia32_pause();
      if ((spins & 0xfff) == 0xfff) {
        const hipError_t q = hipStreamQuery(static_cast<hipStream_t>(stream));
        if (q == hipSuccess) {
          v = *slot;
          if (((v >> 32) & 0x3fffffffull) != want) {
            uint32_t w6[6] = {0u, 0u, 0u, 0u, 0u, 0u};
            GGD_HIP(hipMemcpy(w6, ctx->d_words, sizeof(w6), hipMemcpyDeviceToHost));
            v = ((unsigned long long)(w6[2] & 1u) << 63) | ((unsigned long long)((w6[2] >> 1) & 1u) << 62) | (want << 32) | w6[0];
            ctx->h_words[4] = w6[3]; ctx->h_words[5] = w6[4]; ctx->h_words[6] = w6[5];
          }
          break;
        }
        if (q != hipErrorNotReady)   // the stream is in an error state: do not spin on a word that will never come
          return ggd_fail(ctx, GGD_E_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
      }
      v = *slot;
    }
    ctx->h_words[0] = (uint32_t)v;
    ctx->frame_flat = (v >> 63) != 0ull;
    ctx->frame_msd_ok = ((v >> 62) & 1ull) != 0ull;
    // (stored by the same device thread before the tagged word's release store; x86 loads are not reordered with earlier loads)
    volatile uint32_t* hw = ctx->h_words;
    ctx->frame_kmin = hw[4]; ctx->frame_kmax = hw[5]; ctx->frame_msd_flags = hw[6];
  } else {
    GGD_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  }
  if (prm->prefiltered && ctx->h_words[1] != 0)
    return ggd_fail(ctx, GGD_E_PREFILTER, "Point is filtered although prefiltered is set. This shouldn't happen!");
  *num_rendered = (int64_t)ctx->h_words[0];
  return GGD_OK;
}

extern "C" int ggd_forward_geometry(ggd_ctx* ctx, void* stream, const ggd_params* prm, const float* means3D,
                                    const float* shs, const float* colors_precomp, const float* opacities,
                                    const float* scales, const float* rotations, const float* cov3D_precomp,
                                    void* geom_buf, int32_t* radii, int64_t* num_rendered) {
  if (ctx && ctx->pending.valid) return ggd_fail(ctx, GGD_E_INVALID, kPendingMsg);
  const int rc = geometry_enqueue(ctx, stream, prm, means3D, shs, colors_precomp, opacities, scales, rotations,
                                  cov3D_precomp, geom_buf, radii, num_rendered);
  if (rc != GGD_OK || prm->P == 0) return rc;
  return geometry_finish(ctx, stream, prm, num_rendered);
}

// R = the value binning_buf was laid out for.  speculative: R is only a CAPACITY (the true num_rendered is still on
// the device); only the tile-binning path can run that way (its launch geometry does not depend on R).
static int render_enqueue(ggd_ctx* ctx, void* stream, const ggd_params* prm, const void* geom_buf, int64_t layout_R,
                          int64_t R, void* binning_buf, void* img_buf, float* out_color, bool speculative) {
  (void)hipGetLastError();   // a sticky error another library left in this thread is not ours to report
  // layout_R: what binning_buf was laid out for; R: number of instances to process (== layout_R unless the caller
  // over-allocated; in the speculative case the true count is still on the device and R is only its upper bound)
  int rc = check_params(ctx, prm);
  if (rc != GGD_OK) return rc;
  if (R < 0) return ggd_fail(ctx, GGD_E_INVALID, "num_rendered < 0");
  if (!img_buf || !out_color) return ggd_fail(ctx, GGD_E_INVALID, "img_buf / out_color is NULL");
  if (R > 0 && (!geom_buf || !binning_buf)) return ggd_fail(ctx, GGD_E_INVALID, "geom_buf / binning_buf is NULL");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ggd_geom_view gv; ggd_binning_view bv; ggd_img_view iv;
  ggd_geom_layout(prm->P, &gv); ggd_binning_layout(layout_R, &bv); ggd_img_layout(prm->width, prm->height, &iv);
  const char* gb = static_cast<const char*>(geom_buf);
  char* bb = static_cast<char*>(binning_buf);
  char* ib = static_cast<char*>(img_buf);
  const ggd_splat* splat = reinterpret_cast<const ggd_splat*>(gb + gv.splat);
  const uint32_t* tiles = reinterpret_cast<const uint32_t*>(gb + gv.tiles_touched);
  const uint32_t* offsets = reinterpret_cast<const uint32_t*>(gb + gv.point_offsets);
  const uint2* rect = reinterpret_cast<const uint2*>(gb + gv.rect);
  const uint32_t* depth_keys_all = reinterpret_cast<const uint32_t*>(gb + gv.depth_keys);
  uint64_t* keys = reinterpret_cast<uint64_t*>(bb + bv.keys);
  uint32_t* list = reinterpret_cast<uint32_t*>(bb + bv.list);
  uint64_t* keys_alt = reinterpret_cast<uint64_t*>(bb + bv.keys_alt);
  uint32_t* list_alt = reinterpret_cast<uint32_t*>(bb + bv.list_alt);
  uint32_t* ranges = reinterpret_cast<uint32_t*>(ib + iv.ranges);
  float* final_T = reinterpret_cast<float*>(ib + iv.final_T);
  uint32_t* n_contrib = reinterpret_cast<uint32_t*>(ib + iv.n_contrib);
  const int T = ((prm->width + 15) / 16) * ((prm->height + 15) / 16);
  const int nbits = ggd_sort_bits(prm->width, prm->height);

  const int bmode = ctx->opt[GGD_OPT_BINNING];
  // binning path: 0 = duplicate + radix sort; 3 (2: alias) = two-level row / column binning; 1 = auto: row binning when the
  // grid is <= 255 x 255 tiles (and R is past its fixed costs), else the sort
  const bool rowbin_ok = !prm->debug && ggd_rowbin_supported(prm->width, prm->height);
  const bool rowbin = rowbin_ok && (bmode == 2 || bmode == 3 || (bmode == 1 && R >= GGD_ROWBIN_MIN_R));
  const bool tilebin = rowbin;
  if (speculative && !tilebin) return ggd_fail(ctx, GGD_E_INVALID, "speculative render needs the tile-binning path");
  const uint32_t capacity = R > 0xffffffffll ? 0xffffffffu : (uint32_t)R;
  if (R > 0 && tilebin) {
    // depth-sort the Gaussians once (32-bit keys), then one stable tile-binning pass
    const uint32_t* depth_keys = reinterpret_cast<const uint32_t*>(gb + gv.depth_keys);
    const size_t pairs = ggd_align((size_t)prm->P * sizeof(uint32_t));
    const size_t sort_tmp = ggd_sort32_tmp_bytes(prm->P);
    const size_t bin_tmp = ggd_rowbin_tmp_bytes(prm->P, capacity, prm->width, prm->height);
    const size_t msd_tab = (ctx->msd_frame && ctx->scan_deferred && ctx->fold_active) ? ggd_sort32_msd_table_bytes(prm->P) : 0;
    rc = ggd_reserve_scratch(ctx, 4 * pairs + sort_tmp + bin_tmp + msd_tab, s);
    if (rc != GGD_OK) return rc;
    char* sc = static_cast<char*>(ctx->scratch);
    uint32_t* ka = reinterpret_cast<uint32_t*>(sc);
    uint32_t* va = reinterpret_cast<uint32_t*>(sc + pairs);
    uint32_t* kb = reinterpret_cast<uint32_t*>(sc + 2 * pairs);
    uint32_t* vb = reinterpret_cast<uint32_t*>(sc + 3 * pairs);
    void* tmp = sc + 4 * pairs;
    void* bin_tmp_ptr = sc + 4 * pairs + sort_tmp;  // the sort's histogram block stays alive for the binning pass
    uint32_t* clean_ctl = nullptr;
    ggd_scan_piggy pg;          // a scan that rides on this call's launches (see geometry_enqueue)
    bool riding = false;
    const uint32_t *n_vis_ptr = nullptr, *flat_ptr = nullptr;   // device words: kept keys, "last pass was flat"
    uint32_t* folded_l1 = nullptr;                              // the folded front end's control block, if this call has one
    {
      StageTimer t(ctx, ST_SORT, s);
      // the control block this frame's scan cleared, if nobody has used it since (a second render of the same geometry
      // falls back to the memset)
      clean_ctl = ctx->sortctl_clean ? ctx->sortctl : nullptr;
      ctx->sortctl_clean = false;
      ggd_fold fold;
      const bool folded = ctx->scan_deferred && ctx->fold_active;   // this call's preprocess filled the histograms
      if (ctx->scan_deferred) {   // this call's geometry half left the scan to us
        uint32_t* tiles_w = reinterpret_cast<uint32_t*>(const_cast<char*>(gb) + gv.tiles_touched);
        pg.in = tiles_w; pg.out = reinterpret_cast<uint32_t*>(const_cast<char*>(gb) + gv.point_offsets);
        pg.n = prm->P; pg.nb = ggd_scan_blocks(prm->P); pg.block_sums = ctx->scan_sums;
        pg.d_total = ctx->d_words; pg.h_total = ctx->h_words_dev;
        pg.h_tagged = reinterpret_cast<unsigned long long*>(ctx->h_words_dev + 2);
        ctx->r_tag = (ctx->r_tag + 1u) & 0x3fffffffu;
        if (ctx->r_tag == 0u) ctx->r_tag = 1u;
        pg.tag = ctx->r_tag;
        if (folded) {
          const int nwg = (prm->P + 255) / 256;
          fold.ctl = ctx->foldctl[ctx->fold_cur ^ 1];   // (fold_cur already points at the next frame's block)
          pg.wg_info = reinterpret_cast<const uint4*>(ctx->scan_sums + (((size_t)nwg + 3) & ~(size_t)3));
          pg.n_info = nwg;
          pg.sum_stride = 8;                            // 2048-element scan blocks over 256-point workgroup prefixes
        }
      }
      ctx->fold_active = false;
      riding = ctx->scan_deferred;
      // the fourth pass is an empty launch when the depths' top byte is constant: after GGD_FLAT_STREAK such frames it is not
      // launched; ggd_forward re-renders a frame for which that was wrong (frame_flat arrives with num_rendered)
      ctx->frame_folded = folded;
      const bool msd = folded && ctx->msd_frame;   // (this call's preprocess built the two-launch sort's histograms)
      if (msd && !(rowbin && speculative)) return ggd_fail(ctx, GGD_E_INVALID, "internal: two-launch sort outside the speculative tile-binning path");
      ctx->spec3 = !msd && folded && rowbin && speculative && ctx->flat_streak >= GGD_FLAT_STREAK && ctx->opt[GGD_OPT_FOLD] == 1;
      if (!msd) ctx->msd_frame = false;
      fold.msd = msd ? 1 : 0;
      fold.msd_lo = ctx->msd_lo; fold.msd_shift = ctx->msd_shift;
      if (msd)
        rc = ggd_launch_sort32_msd(ctx, s, depth_keys, ka, va, kb, vb, prm->P,
                                   reinterpret_cast<uint32_t*>(sc + 4 * pairs + sort_tmp + bin_tmp), &pg, &fold);
      else
      rc = ggd_launch_sort32_iota(ctx, s, depth_keys, ka, va, kb, vb, prm->P, 32, tmp, sort_tmp, folded ? nullptr : clean_ctl,
                                  riding ? &pg : nullptr, rowbin, !rowbin, folded ? &fold : nullptr, ctx->spec3);
      ctx->r_pending = riding && rc == GGD_OK;
      ctx->scan_deferred = false;
      if (rc != GGD_OK) return rc;
      if (folded) { n_vis_ptr = ggd_fold_nvalid_ptr(fold.ctl); flat_ptr = ggd_fold_flat_ptr(fold.ctl); folded_l1 = fold.ctl; }
      else {
        const void* ctl = clean_ctl ? static_cast<const void*>(clean_ctl) : tmp;
        n_vis_ptr = ggd_sort32_nvalid_ptr(ctl); flat_ptr = ggd_sort32_flat_ptr(ctl);
      }
    }
    {
      StageTimer t(ctx, ST_DUPLICATE, s);
      // the depth sort dropped the culled Gaussians (key 0xFFFFFFFF) and left the number of kept ones on the device
      const bool l1 = folded_l1 != nullptr && (prm->width + 15) / 16 <= 64 && (prm->height + 15) / 16 <= 64;
      rc = ggd_launch_rowbin(ctx, s, *prm, rect, va, n_vis_ptr, list, ranges, capacity, bin_tmp_ptr, bin_tmp, vb,
                             flat_ptr, riding ? &pg : nullptr, l1 ? folded_l1 + GGD_FOLD_ROWTOT : nullptr,
                             l1 ? folded_l1 + ggd_fold_l1_offset(prm->P) : nullptr);
      if (rc != GGD_OK) return rc;
    }
  } else {
  if (R > 0) {
    const size_t sort_tmp = ggd_sort_tmp_bytes(R);
    rc = ggd_reserve_scratch(ctx, sort_tmp, s);
    if (rc != GGD_OK) return rc;
    const bool to_alt = ggd_sort_input_is_alt(nbits) != 0;
    uint64_t* k0 = to_alt ? keys_alt : keys;
    uint32_t* v0 = to_alt ? list_alt : list;
    {
      StageTimer t(ctx, ST_DUPLICATE, s);
      rc = ggd_launch_duplicate(ctx, s, *prm, rect, depth_keys_all, tiles, offsets, k0, v0);
      if (rc != GGD_OK) return rc;
    }
    if (prm->debug) {
      if (ctx->dbg_cap < (size_t)R) {
        if (ctx->dbg_keys) (void)hipFree(ctx->dbg_keys);
        if (ctx->dbg_vals) (void)hipFree(ctx->dbg_vals);
        ctx->dbg_keys = ctx->dbg_vals = nullptr; ctx->dbg_cap = 0;
        GGD_HIP(hipMalloc(&ctx->dbg_keys, (size_t)R * sizeof(uint64_t)));
        GGD_HIP(hipMalloc(&ctx->dbg_vals, (size_t)R * sizeof(uint32_t)));
        ctx->dbg_cap = (size_t)R;
      }
      GGD_HIP(hipMemcpyAsync(ctx->dbg_keys, k0, (size_t)R * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
      GGD_HIP(hipMemcpyAsync(ctx->dbg_vals, v0, (size_t)R * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    }
    {
      StageTimer t(ctx, ST_SORT, s);
      rc = ggd_launch_sort(ctx, s, keys, list, keys_alt, list_alt, R, nbits, ctx->scratch, ctx->scratch_bytes);
      if (rc != GGD_OK) return rc;
    }
  }
  {
    StageTimer t(ctx, ST_RANGES, s);
    rc = ggd_launch_ranges(ctx, s, keys, R, ranges, T);
    if (rc != GGD_OK) return rc;
  }
  }
  {
    StageTimer t(ctx, ST_BLEND, s);
    rc = ggd_launch_blend(ctx, s, *prm, splat, list, ranges, capacity, out_color, final_T, n_contrib);
    if (rc != GGD_OK) return rc;
  }
  return GGD_OK;
}

extern "C" int ggd_forward_render(ggd_ctx* ctx, void* stream, const ggd_params* prm, const void* geom_buf,
                                  int64_t R, void* binning_buf, void* img_buf, float* out_color) {
  // (the speculation state of a frame between ggd_forward_enqueue and ggd_forward_collect -- spec3, msd_frame, frame_folded, the
  // tagged read-back -- lives on the context: another forward on it would clear that state and a missed speculation would
  // never be rendered again)
  if (ctx && ctx->pending.valid) return ggd_fail(ctx, GGD_E_INVALID, kPendingMsg);
  return render_enqueue(ctx, stream, prm, geom_buf, R, R, binning_buf, img_buf, out_color, false);
}

extern "C" int ggd_forward_can_speculate(ggd_ctx* ctx, const ggd_params* prm, int64_t capacity) {
  if (!ctx || !prm || prm->debug || prm->P <= 0) return 0;
  const int bmode = ctx->opt[GGD_OPT_BINNING];
  return (ggd_rowbin_supported(prm->width, prm->height) &&
          (bmode == 2 || bmode == 3 || (bmode == 1 && capacity >= GGD_ROWBIN_MIN_R))) ? 1 : 0;
}

// The speculative route of the single-call forward in its two halves: everything is enqueued before the host looks at
// num_rendered (the GPU never idles on that read-back) ...
static int forward_spec_enqueue(ggd_ctx* ctx, void* stream, const ggd_params* prm, const float* means3D, const float* shs,
                                const float* colors_precomp, const float* opacities, const float* scales,
                                const float* rotations, const float* cov3D_precomp, void* geom_buf, int32_t* radii,
                                void* binning_buf, int64_t capacity, void* img_buf, float* out_color, int64_t* num_rendered) {
  int rc = geometry_enqueue(ctx, stream, prm, means3D, shs, colors_precomp, opacities, scales, rotations,
                            cov3D_precomp, geom_buf, radii, num_rendered, true);
  if (rc != GGD_OK) return rc;
  ctx->spec3 = false; ctx->frame_folded = false; ctx->frame_flat = false; ctx->frame_msd_ok = false;
  ctx->frame_kmin = 0xffffffffu; ctx->frame_kmax = 0u; ctx->frame_msd_flags = 0u;
  return render_enqueue(ctx, stream, prm, geom_buf, capacity, capacity, binning_buf, img_buf, out_color, true);
}
// ... and the collection of num_rendered (+ "the depth keys' top byte was constant") once the launch that delivers it has run;
// binning and blend may still be running.  A frame that needed the fourth sort pass it did not get is binned and blended again.
static int forward_spec_collect(ggd_ctx* ctx, void* stream, const ggd_params* prm, const void* geom_buf, void* binning_buf,
                                int64_t capacity, void* img_buf, float* out_color, int64_t* num_rendered) {
  int rc = geometry_finish(ctx, stream, prm, num_rendered);
  if (rc != GGD_OK) return rc;
  const bool spec3 = ctx->spec3, msd = ctx->msd_frame;
  ctx->spec3 = false; ctx->msd_frame = false;
  if (ctx->frame_folded) ctx->flat_streak = ctx->frame_flat ? (ctx->flat_streak < (1 << 30) ? ctx->flat_streak + 1 : ctx->flat_streak) : 0;
  if (ctx->frame_folded && ctx->msd_ban > 0) ctx->msd_ban -= 1;
  const bool msd_missed = msd && !ctx->frame_msd_ok;
  const bool msd_oversize = msd_missed && (ctx->frame_msd_flags & 4u) == 0u;   // a bucket above the finish kernel's capacity
  if (msd_oversize) {
    // the window was too coarse for this data -- typically fitted to another scene: forget the older frames' ranges (the window
    // re-forms from this scene's in GGD_FLAT_STREAK frames) -- or the data has > GGD_MSD_CAP equal keys, which no window cures:
    // the pause doubles with every consecutive such miss (8, 16, 32, GGD_MSD_BAN frames)
    ctx->msd_oversize_streak = ctx->msd_oversize_streak < 4 ? ctx->msd_oversize_streak + 1 : 4;
    ctx->msd_ban = GGD_MSD_BAN >> (4 - ctx->msd_oversize_streak);
    ctx->win_n = 0; ctx->win_pos = 0;
  } else if (msd_missed) {
    ctx->msd_ban = 2;   // a key outside the window: this frame's range joins the window below, the next frames fit again
  }
  if (ctx->frame_folded && ctx->frame_kmin <= ctx->frame_kmax) {   // (something was kept) -> the ring of recent key ranges
    ctx->win_lo[ctx->win_pos] = ctx->frame_kmin; ctx->win_hi[ctx->win_pos] = ctx->frame_kmax;
    ctx->win_pos = (ctx->win_pos + 1) % GGD_MSD_WIN;
    if (ctx->win_n < (1 << 30)) ctx->win_n += 1;
  }
  if (msd && !msd_missed) { ctx->msd_frames += 1; ctx->msd_oversize_streak = 0; }
  if (*num_rendered > capacity)
    return ggd_fail(ctx, GGD_E_CAPACITY, "binning_buf capacity is below num_rendered: re-run with a larger buffer");
  if (msd_missed || (spec3 && !ctx->frame_flat)) {   // the short form of the sort did not hold for this frame: bin and blend it again, in full
    ctx->spec3_misses += 1;
    return render_enqueue(ctx, stream, prm, geom_buf, capacity, *num_rendered, binning_buf, img_buf, out_color, false);
  }
  return GGD_OK;
}

extern "C" int ggd_forward(ggd_ctx* ctx, void* stream, const ggd_params* prm, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           const float* rotations, const float* cov3D_precomp, void* geom_buf, int32_t* radii,
                           void* binning_buf, int64_t capacity, void* img_buf, float* out_color,
                           int64_t* num_rendered) {
  if (capacity < 0) return ggd_fail(ctx, GGD_E_INVALID, "capacity < 0");
  if (ctx && ctx->pending.valid) return ggd_fail(ctx, GGD_E_INVALID, kPendingMsg);
  if (prm && prm->P > 0 && capacity > 0 && ggd_forward_can_speculate(ctx, prm, capacity)) {
    const int rc = forward_spec_enqueue(ctx, stream, prm, means3D, shs, colors_precomp, opacities, scales, rotations,
                                        cov3D_precomp, geom_buf, radii, binning_buf, capacity, img_buf, out_color, num_rendered);
    if (rc != GGD_OK) return rc;
    return forward_spec_collect(ctx, stream, prm, geom_buf, binning_buf, capacity, img_buf, out_color, num_rendered);
  }
  int rc = geometry_enqueue(ctx, stream, prm, means3D, shs, colors_precomp, opacities, scales, rotations,
                            cov3D_precomp, geom_buf, radii, num_rendered, false);
  if (rc != GGD_OK) return rc;
  if (prm->P == 0) return render_enqueue(ctx, stream, prm, geom_buf, capacity, 0, binning_buf, img_buf, out_color, false);
  rc = geometry_finish(ctx, stream, prm, num_rendered);
  if (rc != GGD_OK) return rc;
  if (*num_rendered > capacity)
    return ggd_fail(ctx, GGD_E_CAPACITY, "binning_buf capacity is below num_rendered: re-run with a larger buffer");
  return render_enqueue(ctx, stream, prm, geom_buf, capacity, *num_rendered, binning_buf, img_buf, out_color, false);
}

extern "C" int ggd_forward_enqueue(ggd_ctx* ctx, void* stream, const ggd_params* prm, const float* means3D, const float* shs,
                                   const float* colors_precomp, const float* opacities, const float* scales,
                                   const float* rotations, const float* cov3D_precomp, void* geom_buf, int32_t* radii,
                                   void* binning_buf, int64_t capacity, void* img_buf, float* out_color) {
  if (!ctx) return GGD_E_INVALID;
  if (ctx->pending.valid) return ggd_fail(ctx, GGD_E_INVALID, "ggd_forward_enqueue: the previous frame of this context has not been collected");
  if (!prm || prm->P <= 0 || capacity <= 0 || !ggd_forward_can_speculate(ctx, prm, capacity))
    return ggd_fail(ctx, GGD_E_INVALID, "ggd_forward_enqueue needs the tile-binning path, P > 0 and a capacity (ggd_forward_can_speculate)");
  int64_t dummy = 0;
  const int rc = forward_spec_enqueue(ctx, stream, prm, means3D, shs, colors_precomp, opacities, scales, rotations,
                                      cov3D_precomp, geom_buf, radii, binning_buf, capacity, img_buf, out_color, &dummy);
  if (rc != GGD_OK) return rc;
  ctx->pending.valid = true; ctx->pending.prm = *prm; ctx->pending.geom = geom_buf; ctx->pending.binning = binning_buf;
  ctx->pending.capacity = capacity; ctx->pending.img = img_buf; ctx->pending.out = out_color;
  return GGD_OK;
}

extern "C" int ggd_forward_collect(ggd_ctx* ctx, void* stream, int64_t* num_rendered) {
  if (!ctx || !num_rendered) return GGD_E_INVALID;
  if (!ctx->pending.valid) return ggd_fail(ctx, GGD_E_INVALID, "ggd_forward_collect: no frame is pending on this context");
  ctx->pending.valid = false;
  return forward_spec_collect(ctx, stream, &ctx->pending.prm, ctx->pending.geom, ctx->pending.binning, ctx->pending.capacity,
                              ctx->pending.img, ctx->pending.out, num_rendered);
}

// ---- backward --------------------------------------------------------------------------------------------------
extern "C" int ggd_backward(ggd_ctx* ctx, void* stream, const ggd_params* prm, const float* means3D,
                            const float* shs, const float* colors_precomp, const float* opacities, const float* scales,
                            const float* rotations, const float* cov3D_precomp, const int32_t* radii,
                            const void* geom_buf, const void* binning_buf, const void* img_buf, int64_t R,
                            const float* dL_dpix, float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity,
                            float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales,
                            float* dL_drots) {
  (void)hipGetLastError();   // a sticky error another library left in this thread is not ours to report
  int rc = check_params(ctx, prm);
  if (rc != GGD_OK) return rc;
  rc = check_inputs(ctx, prm, means3D, shs, colors_precomp, scales, rotations, cov3D_precomp);
  if (rc != GGD_OK) return rc;
  const int P = prm->P;
  if (P == 0) return GGD_OK;
  if (!radii || !geom_buf || !img_buf || !dL_dpix || !dL_dmeans2D || !dL_dcolors || !dL_dopacity ||
      !dL_dmeans3D || !dL_dcov3D || !dL_dscales || !dL_drots || (prm->M > 0 && !dL_dsh) || (R > 0 && !binning_buf))
    return ggd_fail(ctx, GGD_E_INVALID, "ggd_backward: NULL buffer");
  if (prm->raw_attributes && (!opacities || cov3D_precomp))
    return ggd_fail(ctx, GGD_E_INVALID, "raw_attributes needs opacities and scales/rotations (no cov3D_precomp)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ggd_geom_view gv; ggd_binning_view bv; ggd_img_view iv;
  ggd_geom_layout(P, &gv); ggd_binning_layout(R, &bv); ggd_img_layout(prm->width, prm->height, &iv);
  const char* gb = static_cast<const char*>(geom_buf);
  const char* bb = static_cast<const char*>(binning_buf);
  const char* ib = static_cast<const char*>(img_buf);
  const ggd_splat* splat = reinterpret_cast<const ggd_splat*>(gb + gv.splat);
  const uint8_t* clamped = reinterpret_cast<const uint8_t*>(gb + gv.clamped);
  const uint32_t* list = reinterpret_cast<const uint32_t*>(bb + bv.list);
  const uint32_t* ranges = reinterpret_cast<const uint32_t*>(ib + iv.ranges);
  const float* final_T = reinterpret_cast<const float*>(ib + iv.final_T);
  const uint32_t* n_contrib = reinterpret_cast<const uint32_t*>(ib + iv.n_contrib);

  const size_t acc_bytes = (size_t)P * GGD_ACC_FLOATS * sizeof(float);
  rc = ggd_reserve_scratch(ctx, ggd_align(acc_bytes), s);
  if (rc != GGD_OK) return rc;
  float* grad_acc = static_cast<float*>(ctx->scratch);
  // the only zero-fill of the backward: the accumulator records.  Every caller array is written in full by the
  // per-Gaussian kernel, except the ones a mode never produces (kept zero like upstream's zero-initialised outputs).
  GGD_HIP(hipMemsetAsync(grad_acc, 0, acc_bytes, s));
  if (cov3D_precomp) {
    GGD_HIP(hipMemsetAsync(dL_dscales, 0, (size_t)P * 3 * sizeof(float), s));
    GGD_HIP(hipMemsetAsync(dL_drots, 0, (size_t)P * 4 * sizeof(float), s));
  }
  if (colors_precomp && prm->M > 0 && dL_dsh) GGD_HIP(hipMemsetAsync(dL_dsh, 0, (size_t)P * prm->M * 3 * sizeof(float), s));

  if (R > 0) {
    StageTimer t(ctx, ST_BLEND_BWD, s);
    rc = ggd_launch_blend_backward(ctx, s, *prm, splat, list, ranges, final_T, n_contrib, dL_dpix, grad_acc);
    if (rc != GGD_OK) return rc;
  }
  {
    StageTimer t(ctx, ST_PREPROCESS_BWD, s);
    rc = ggd_launch_preprocess_backward(ctx, s, *prm, means3D, shs, colors_precomp, opacities, dL_dopacity, scales, rotations,
                                        cov3D_precomp, radii, shs ? clamped : nullptr, grad_acc, dL_dmeans2D,
                                        dL_dcolors, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drots);
    if (rc != GGD_OK) return rc;
  }
  return GGD_OK;
}

extern "C" int ggd_mark_visible(ggd_ctx* ctx, void* stream, int32_t P, const float* means3D,
                                const float* viewmatrix, const float* projmatrix, uint8_t* present) {
  (void)projmatrix;  // the frustum test only needs view-space z (upstream computes p_proj and ignores it)
  if (!ctx) return GGD_E_INVALID;
  if (P < 0) return ggd_fail(ctx, GGD_E_INVALID, "P < 0");
  if (P == 0) return GGD_OK;
  if (!means3D || !viewmatrix || !present) return ggd_fail(ctx, GGD_E_INVALID, "ggd_mark_visible: NULL pointer");
  return ggd_launch_mark_visible(ctx, static_cast<hipStream_t>(stream), P, means3D, viewmatrix, present);
}

extern "C" int ggd_debug_unsorted(ggd_ctx* ctx, void* stream, uint64_t* keys, uint32_t* values, int64_t R) {
  if (!ctx) return GGD_E_INVALID;
  if (R < 0 || (size_t)R > ctx->dbg_cap) return ggd_fail(ctx, GGD_E_INVALID, "no debug copy of that size (debug=1?)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (keys && R) GGD_HIP(hipMemcpyAsync(keys, ctx->dbg_keys, (size_t)R * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
  if (values && R) GGD_HIP(hipMemcpyAsync(values, ctx->dbg_vals, (size_t)R * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  return GGD_OK;
}
