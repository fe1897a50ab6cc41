/*
 * ggd_raster.h -- C ABI of the MI355X-native (gfx950) differentiable 3D-Gaussian-splatting rasterizer.
 *
 * This is the drop-in boundary for the one hot path of fraunhoferhhi/gaussian_gan_decoder: the native
 * module `diff_gaussian_rasterization._C` that the reference imports at
 *   gaussian_splatting/gaussian_renderer/__init__.py:14
 * and drives through `GaussianRasterizer(...)` at
 *   gaussian_splatting/gaussian_renderer/__init__.py:53,87-95 (render) and :139,167-175 (render_simple).
 * The CUDA sources behind that module are an EMPTY, unpinned git submodule in the reference tree
 * (.gitmodules:4-6); the three pybind entry points it would export are replaced here one-to-one:
 *
 *   _C.rasterize_gaussians            ->  ggd_forward_geometry() + ggd_forward_render()
 *   _C.rasterize_gaussians_backward   ->  ggd_backward()
 *   _C.mark_visible                   ->  ggd_mark_visible()
 *
 * Plain pointers and sizes only; no torch types.  All `const float*` / `void*` tensor arguments are DEVICE
 * pointers (HIP, gfx950) unless a comment says "host".  `stream` is a hipStream_t passed as void* (NULL = the
 * default stream).  Every function returns 0 on success or a negative GGD_E_* code; the message is available
 * through ggd_last_error().  Nothing here throws.
 *
 * Buffer ownership mirrors the upstream design (geomBuffer / binningBuffer / imgBuffer byte tensors that the
 * autograd function keeps alive between forward and backward): the CALLER allocates them with the sizes
 * returned by ggd_geom_bytes / ggd_binning_bytes / ggd_img_bytes, the library only fills them.  The ctx owns
 * a grow-only scratch workspace (sort ping-pong space, histograms, block sums, a pinned word for the
 * `num_rendered` read-back) and is not re-entrant: one ctx per (device, stream).
 */
#ifndef GGD_RASTER_H
#define GGD_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGD_TILE 16 /* tile edge in pixels (16x16), fixed by the algorithm's contract */

enum {
  GGD_OK = 0,
  GGD_E_INVALID = -1,   /* bad argument (null pointer, negative size, exactly-one-of rule violated, degree) */
  GGD_E_HIP = -2,       /* a HIP runtime call failed */
  GGD_E_NOMEM = -3,     /* workspace allocation failed */
  GGD_E_PREFILTER = -4, /* prefiltered=1 but a Gaussian failed the frustum test (upstream traps here) */
  GGD_E_NODEVICE = -5,  /* no gfx950 device visible */
  GGD_E_CAPACITY = -6   /* ggd_forward: binning_buf capacity below num_rendered (num_rendered is valid: retry) */
};

typedef struct ggd_ctx ggd_ctx;

/* Per-call parameters == the fields of the reference's GaussianRasterizationSettings
 * (constructed at gaussian_renderer/__init__.py:38-51 and :124-137) plus tensor extents. */
typedef struct ggd_params {
  int32_t P;             /* number of Gaussians */
  int32_t M;             /* SH coefficients per channel present in `shs` (stride); 0 when colors_precomp is used */
  int32_t sh_degree;     /* active SH degree D, 0..3 ((D+1)^2 <= M) */
  int32_t width, height; /* image size in pixels */
  float tanfovx, tanfovy;
  float scale_modifier;
  int32_t prefiltered;
  int32_t debug;            /* 1: keep a copy of the unsorted key/value list for ggd_debug_unsorted() */
  const float* viewmatrix;  /* device, 16 floats: world_view_transform flattened row-major (= V^T), cameras.py:85 */
  const float* projmatrix;  /* device, 16 floats: full_proj_transform flattened row-major (= (P V)^T), cameras.py:91 */
  const float* campos;      /* device, 3 floats */
  const float* bg;          /* device, 3 floats */
  int32_t raw_attributes;   /* 1: `scales`, `rotations`, `opacities` are the RAW decoder outputs and the activation
                               prologue of gaussian_model.py:100-121 (exp / L2-normalise (eps 1e-12) / sigmoid) is fused
                               into the per-Gaussian kernels, forward and backward; 0: as upstream */
  int32_t reserved_;
} ggd_params;

/* One record per Gaussian, written by the preprocess kernel and gathered by the blend kernels: exactly what a blended
 * (tile, Gaussian) instance needs, with the per-Gaussian constants of the inner loop formed ONCE here instead of once
 * per instance: the three coefficients of  power = fma(fma(hA, dx, nB*dy), dx, (hC*dy)*dy)  (hA = -A/2, nB = -B,
 * hC = -C/2 are exact rescalings of the conic (A, B, C): A = -2 hA etc. bit for bit), the power-domain cull threshold
 * and the half extents of the axis-aligned box outside of which alpha < 1/255 for sure. */
typedef struct ggd_splat {
  float x, y;               /* pixel-space centre ("means2D" upstream) */
  float hA, nB, hC;         /* -conic.A / 2, -conic.B, -conic.C / 2 (conic = inverse 2D covariance) */
  float thr;                /* ln(1 / (255 opacity)) lowered by a safety margin: power < thr  =>  alpha < 1/255 */
  float opacity;
  float r, g, b;            /* colour after SH evaluation / colors_precomp */
  float ex, ey;             /* half extents of the box {power >= thr}, inflated for fp32 rounding (+inf: keep always) */
} ggd_splat;                /* 48 bytes */

/* Byte offsets of the named arrays inside the caller-owned buffers (for tests / debug taps / bindings). */
typedef struct ggd_geom_view {
  size_t splat;         /* ggd_splat[P] */
  size_t tiles_touched; /* uint32[P] */
  size_t point_offsets; /* uint32[P], inclusive prefix sum of tiles_touched */
  size_t clamped;       /* uint8[P], bit c set <=> colour channel c was clamped at 0 */
  size_t depth_keys;    /* uint32[P]: raw fp32 bits of the view-space depth, 0xFFFFFFFF for culled Gaussians */
  size_t rect;          /* uint32[2*P]: tile rect {minx | maxx << 16, miny | maxy << 16}, zero for culled Gaussians */
  size_t header;        /* uint32[64]: reserved (zeroed) */
  size_t total;
} ggd_geom_view;

typedef struct ggd_binning_view {
  size_t keys;     /* uint64[R] sorted keys:  (tile_id << 32) | depth_bits */
  size_t list;     /* uint32[R] sorted Gaussian indices ("point_list"); ALWAYS at offset 0 of binning_buf, so readers of
                      the list (the backward) do not need to know which R the buffer was laid out for */
  size_t keys_alt; /* uint64[R] ping-pong partner (contents unspecified after the call) */
  size_t list_alt; /* uint32[R] */
  size_t total;
} ggd_binning_view;

typedef struct ggd_img_view {
  size_t ranges;    /* uint32[2*T]: [first, last) into `list` per tile, (0,0) for empty tiles */
  size_t final_T;   /* float[H*W] */
  size_t n_contrib; /* uint32[H*W] */
  size_t total;
} ggd_img_view;

size_t ggd_geom_bytes(int32_t P);
size_t ggd_binning_bytes(int64_t R);
size_t ggd_img_bytes(int32_t width, int32_t height);
int ggd_geom_layout(int32_t P, ggd_geom_view* out);
int ggd_binning_layout(int64_t R, ggd_binning_view* out);
int ggd_img_layout(int32_t width, int32_t height, ggd_img_view* out);

/* Number of key bits the radix sort covers: 32 + msb(#tiles) (11 for 1024 tiles, 13 for 4096). */
int ggd_sort_bits(int32_t width, int32_t height);

ggd_ctx* ggd_create(int device);
void ggd_destroy(ggd_ctx* ctx);
const char* ggd_last_error(ggd_ctx* ctx); /* ctx may be NULL: returns the last creation error */
const char* ggd_version(void);

/*
 * Forward, phase 1 (per Gaussian): frustum cull, cov3D, EWA cov2D, conic, radius, tile rect, SH->RGB;
 * inclusive scan of tiles_touched; reads the total back ("num_rendered", one host sync on `stream`).
 * Exactly one of {shs, colors_precomp} and exactly one of {scales+rotations, cov3D_precomp} must be non-NULL.
 *   means3D[P,3] opacities[P] shs[P,M,3] colors_precomp[P,3] scales[P,3] rotations[P,4](w,x,y,z) cov3D_precomp[P,6]
 *   geom_buf : ggd_geom_bytes(P) bytes, written     radii : int32[P], written     num_rendered : HOST out
 */
int ggd_forward_geometry(ggd_ctx* ctx, void* stream, const ggd_params* prm,
                         const float* means3D, const float* shs, const float* colors_precomp,
                         const float* opacities, const float* scales, const float* rotations,
                         const float* cov3D_precomp,
                         void* geom_buf, int32_t* radii, int64_t* num_rendered);

/*
 * Forward, phase 2 (per instance / per tile): duplicateWithKeys, stable radix sort of (tile|depth) keys,
 * identifyTileRanges, front-to-back alpha blend.  No host sync.
 *   binning_buf : ggd_binning_bytes(num_rendered)   img_buf : ggd_img_bytes(W,H)   out_color : float[3,H,W]
 */
int ggd_forward_render(ggd_ctx* ctx, void* stream, const ggd_params* prm,
                       const void* geom_buf, int64_t num_rendered,
                       void* binning_buf, void* img_buf, float* out_color);

/*
 * Forward in ONE call with a caller-chosen binning capacity (instances): geometry + render are enqueued back to back,
 * so the GPU does not idle while num_rendered travels to the host (the two-phase form above stalls the stream for that
 * round trip).  On the speculative route the call returns as soon as num_rendered has ARRIVED -- the binning and blend
 * kernels may still be running: out_color and the buffers are ordered on `stream` like the result of any asynchronous
 * launch (upstream's forward likewise returns with its render kernels in flight); synchronise the stream before reading
 * them from the host or from another stream.  binning_buf must hold ggd_binning_bytes(capacity).  Returns
 * GGD_E_CAPACITY (with *num_rendered set) when capacity < num_rendered: call again with a larger buffer.  ggd_backward
 * takes the returned num_rendered (the sorted list is at offset 0 of binning_buf for every capacity).  ggd_forward_can_speculate
 * tells whether the call will take the speculative route (tile-binning path) for that capacity.
 */
int ggd_forward(ggd_ctx* ctx, void* stream, const ggd_params* prm,
                const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                const float* scales, const float* rotations, const float* cov3D_precomp,
                void* geom_buf, int32_t* radii, void* binning_buf, int64_t capacity, void* img_buf,
                float* out_color, int64_t* num_rendered);

/*
 * The same forward in two halves, for hosts that keep several frames in flight (one context + stream per frame slot: the
 * latency-bound front of frame k + 1 -- per-Gaussian kernel, depth sort, binning -- runs under the blend of frame k).
 * ggd_forward_enqueue launches the whole frame and returns at once; it needs the speculative route (ggd_forward_can_speculate,
 * P > 0, capacity > 0).  ggd_forward_collect, called any time later on the same context (typically when the slot comes round
 * again, the frame long finished), returns that frame's num_rendered -- or GGD_E_CAPACITY with it, in which case the outputs
 * are not valid and the frame has to be rendered again with a larger buffer.  At most one frame per context may be pending:
 * while one is, ggd_forward / ggd_forward_geometry / ggd_forward_render / ggd_forward_enqueue on that context return
 * GGD_E_INVALID (the pending frame's verification state lives on the context).  The buffers, the camera matrices and the other
 * pointers of `prm` must stay valid until it is collected.
 */
int ggd_forward_enqueue(ggd_ctx* ctx, void* stream, const ggd_params* prm,
                        const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                        const float* scales, const float* rotations, const float* cov3D_precomp,
                        void* geom_buf, int32_t* radii, void* binning_buf, int64_t capacity, void* img_buf, float* out_color);
int ggd_forward_collect(ggd_ctx* ctx, void* stream, int64_t* num_rendered);
int ggd_forward_can_speculate(ggd_ctx* ctx, const ggd_params* prm, int64_t capacity);

/*
 * Backward.  Consumes the three buffers of the matching forward plus the original inputs and dL/d(out_color).
 * All nine gradient outputs are fully written (zero-filled then accumulated) by the library:
 *   dL_dmeans2D[P,3] (xy = NDC-scaled screen gradient, z = 0)   dL_dcolors[P,3]   dL_dopacity[P]
 *   dL_dmeans3D[P,3]   dL_dcov3D[P,6]   dL_dsh[P,M,3] (may be NULL iff M == 0)   dL_dscales[P,3]   dL_drots[P,4]
 * dL_dconic is internal scratch.
 */
int ggd_backward(ggd_ctx* ctx, void* stream, const ggd_params* prm,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* opacities /* only read when prm->raw_attributes */,
                 const float* scales, const float* rotations, const float* cov3D_precomp,
                 const int32_t* radii,
                 const void* geom_buf, const void* binning_buf, const void* img_buf, int64_t num_rendered,
                 const float* dL_dpix,
                 float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity,
                 float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drots);

/* present[i] = 1 iff Gaussian i passes the view-space z > 0.2 frustum test. */
int ggd_mark_visible(ggd_ctx* ctx, void* stream, int32_t P, const float* means3D,
                     const float* viewmatrix, const float* projmatrix, uint8_t* present);

/* Debug tap (prm->debug = 1 on the preceding ggd_forward_render): copies the UNSORTED duplicateWithKeys output
 * to device buffers keys[R] / values[R] supplied by the caller.  Either pointer may be NULL. */
int ggd_debug_unsorted(ggd_ctx* ctx, void* stream, uint64_t* keys, uint32_t* values, int64_t num_rendered);

/* Tuning / experiment knobs (do not change results beyond the documented tolerances).  Returns GGD_E_INVALID for an
 * unknown option or value. */
enum {
  GGD_OPT_EXP_MODE = 0,   /* blend exp(): 0 = ocml expf (<=1 ulp), 1 = native 2^(x*log2e) (fast, ~3 ulp),
                             2 = compensated 2^x (v_exp_f32 + product-residual correction, 1-2 ulp), in both blend kernels;
                             3 (default) = 1 in the forward, 2 in the backward: at 1 M Gaussians / 1024^2 the image stays within
                             5.4e-7 of the fp32 oracle (2: 3.6e-7; the bar is 1e-5) and the forward blend is 7 % faster; the
                             backward's error budget needs the 1-2 ulp class (mode 1 there: gradients 3-4 x outside it) */
  GGD_OPT_BLEND_CULL = 1, /* 1 (default) = skip records whose alpha cannot reach 1/255 anywhere in the tile */
  GGD_OPT_BINNING = 2,    /* how the per-tile sorted lists are built (results are identical):
                             0 = duplicateWithKeys + 64-bit (tile|depth) radix sort + identifyTileRanges,
                             3 = depth-sort the Gaussians once, then TWO 1-D stable binning passes (tile rows, then
                                 columns; grids up to 255 x 255 tiles = 4080 x 4080 pixels, falls back to 0 beyond),
                             2 = alias of 3 (the single-level tile binning it used to select was removed: slower than 3
                                 on every grid 3 supports),
                             1 (default) = auto: 3 wherever it applies, else 0.
                             debug=1 (key taps) always uses 0 */
  GGD_OPT_BLEND_SPLIT = 3, /* backward blend (the forward always runs four 8x8 quarter waves per tile, one pixel per
                             lane): 3 = the four quarter waves of a tile in one workgroup, per-record sums combined in LDS;
                             4 = four independent quarter waves per tile; 1 (default) = auto (= 4 since round 6: equal or faster on every grid measured);
                             0, 2 = aliases of 3 (the one-wave and two-wave forms they selected were removed).  Backward sums
                             differ only in their fp32 summation order. */
  GGD_OPT_FOLD = 4,       /* single-call forward on the tile-binning path: 1 (default) = the per-Gaussian kernel also builds the
                             depth sort's digit histograms and the first step of the offsets scan (no histogram launch);
                             0 = separate histogram launch.  Results are identical. */
  GGD_OPT_MSD_SORT = 5,   /* 1 (default): once the kept depth keys' ranges of 8 single-call frames are known, the depth sort of the
                             single-call forward runs as TWO launches over a speculated key window fitted to those ranges (one
                             partition by (key - window start) >> shift without any dependency between tiles + an in-LDS finish
                             per bucket) instead of three or four onesweep passes; verified by every frame's own front end (no
                             kept key outside the window, no bucket above the finish kernel's capacity), a frame it does not hold
                             for is re-rendered by the ordinary path and its range joins the window.  0 = never.  Results are
                             identical.  Setting this option or GGD_OPT_FOLD (to any value) restarts the speculation state. */
  GGD_OPT_COUNT
};
int ggd_set_option(ggd_ctx* ctx, int option, int value);
/* Debug: blend work counters of the NEXT forward calls: out[0]=records visited, [1]=records culled by the wave-level
 * test, [2]=lanes with a candidate pixel (summed over visited records), [3]=candidate pixels, [4]=sum of list lengths,
 * [5]=the part of [1] that was staged and then rejected by the whole wave inside the blend loop.
 * enable=1 starts (and zeroes) counting, enable=0 stops; out (host, 6 x uint64) may be NULL.
 * enable=2 records, instead of the counters, a per-wave timeline of the next forward blend: ggd_blend_timeline copies, for
 * the first `waves` workgroups of that launch, 3 x uint64 each: start and end on the device's 100 MHz constant clock, and
 * (list length << 32 | list entries gathered before the wave's pixels were all finished).  (The counters are same-address
 * atomics and stretch the kernel; the timeline is one plain store per wave.) */
int ggd_blend_stats(ggd_ctx* ctx, int enable, unsigned long long* out);
/* ... and of the backward blends (quarter form) that ran since ggd_blend_stats(ctx, 1, ..) started counting; call BEFORE stopping:
 * out[0]=list entries walked (sum over quarter waves of the positions up to the wave's last contributor), [1]=records staged after
 * the pre-cull, [2]=staged records some pixel of the wave still needed, [3]=records at least one pixel blended (one 9-sum wave
 * reduction each), [4]=live lanes of those, [5]=rows flushed = 36-byte atomic spans, [6]=gather rounds, [7]=waves with work. */
int ggd_blend_backward_stats(ggd_ctx* ctx, unsigned long long* out);
int ggd_blend_timeline(ggd_ctx* ctx, unsigned long long* out, int waves);
int ggd_get_option(ggd_ctx* ctx, int option);
/* Read-only counters through ggd_get_option (single-call forward with GGD_OPT_FOLD = 1): the depth keys' top byte is constant in
 * most scenes, which makes the fourth sort pass an empty launch; after 8 such frames in a row it is not launched, every
 * frame's own histogram says whether that held, and a frame for which it did not is binned and blended again before
 * ggd_forward returns (results are identical either way). */
enum {
  GGD_STAT_FLAT_STREAK = 100,   /* consecutive frames whose top depth digit was constant */
  GGD_STAT_SORT_RERUNS = 101,   /* frames re-rendered because the speculated short form of the depth sort did not hold */
  GGD_STAT_MSD_FRAMES = 102     /* frames whose depth sort ran as two launches (GGD_OPT_MSD_SORT) */
};

/*
 * Tri-plane feature gather of the per-point decoder (input side of the raster path; replaces the three torch ops
 * sample_from_planes -> grid_sample -> mean(0) of main/decoder_models/sequential_decoder_reverse.py:42-57 and
 * base_decoder.py:22):  out[n,:] = mean over the 3 EG3D planes of the bilinear sample (zero padding,
 * align_corners = False) at (2/box_warp) * pos[n].  planes_cl is CHANNEL-LAST [3][H][W][C], C a power of two <= 64.
 * The backward zero-fills dplanes_cl [3][H][W][C] and scatter-adds dout[N,C]; positions receive no gradient.
 */
int ggd_triplane_forward(ggd_ctx* ctx, void* stream, const float* planes_cl, int32_t C, int32_t H, int32_t W,
                         const float* pos, int32_t N, float box_warp, float* out);
int ggd_triplane_backward(ggd_ctx* ctx, void* stream, int32_t C, int32_t H, int32_t W, const float* pos, int32_t N,
                          float box_warp, const float* dout, float* dplanes_cl);

/*
 * The PanoHead form of the same gather: every plane is a C x D "tri-grid" sampled with a 3-D grid_sample (trilinear,
 * zero padding, align_corners = False) at all three projected coordinates (PanoHead/training/volumetric_rendering/
 * renderer.py:47-58, chosen at sequential_decoder_reverse.py:42-50 when the generator has `triplane_depth`).
 * grids_cl is CHANNEL-LAST [3][D][H][W][C]; axes: 0 = EG3D plane axes, 1 = PanoHead plane axes (they differ in the third
 * plane).  out[n,:] = mean over the 3 grids.  The backward zero-fills dgrids_cl and scatter-adds dout[N,C].
 */
int ggd_trigrid_forward(ggd_ctx* ctx, void* stream, const float* grids_cl, int32_t C, int32_t D, int32_t H, int32_t W,
                        int32_t axes, const float* pos, int32_t N, float box_warp, float* out);
int ggd_trigrid_backward(ggd_ctx* ctx, void* stream, int32_t C, int32_t D, int32_t H, int32_t W, int32_t axes,
                         const float* pos, int32_t N, float box_warp, const float* dout, float* dgrids_cl);

/*
 * The general form of the four entries above (they are thin wrappers of it): D = 0 selects the 2-D tri-plane form
 * (axes must be 0), D >= 1 the tri-grid.  `mod` (NULL = none): a per-(depth, channel) modulation [max(D,1)][C] of the
 * planes -- what is sampled is fl(texel * mod), i.e. the gather of `planes * code[None, :, None, None]` without that
 * product ever being materialised (the training step's stand-in for per-scene planes of a shared backbone,
 * main/decoder_models/sequential_decoder_reverse.py:89-99); the scatter returns the gradient w.r.t. the UNMODULATED
 * planes.  accumulate != 0: dgrids_cl is added to instead of zero-filled first (several scenes into one gradient
 * buffer).  Large N (>= 49152 points, C in {16, 32, 64}): the (point, plane) items are sorted by cell and summed in
 * registers per run of equal cells, one group of atomics per run; otherwise one atomic per (point, tap, channel).
 */
int ggd_planes_gather(ggd_ctx* ctx, void* stream, const float* grids_cl, int32_t C, int32_t D, int32_t H, int32_t W,
                      int32_t axes, const float* mod, const float* pos, int32_t N, float box_warp, float* out);
int ggd_planes_scatter(ggd_ctx* ctx, void* stream, int32_t C, int32_t D, int32_t H, int32_t W, int32_t axes,
                       const float* mod, const float* pos, int32_t N, float box_warp, const float* dout,
                       float* dgrids_cl, int32_t accumulate);

/*
 * GPU iso-surface point sampler -- the position generator of the decoder training step, replacing the CPU marching
 * cubes + trimesh + D2H/H2D hop of main/decoder_utils/target_dataloader.py:96-118,172-176: density grid sigma[n][n][n]
 * ([x][y][z], z fastest, main/marching_cube/sample.py:15-17) -> iso-surface at `level` (reference: 10) by marching
 * tetrahedra -> num_points positions, point i on face (i mod F) of pass (i div F) with barycentric weights rand(3)/sum
 * (:104-110), in the reference's units (index / n - 0.5, :99-101), scaled by clip(1 + thickness * N(0,1), 0, 1)
 * (:113-116).  No host sync: the face count F stays on the device (*num_faces, a DEVICE uint32; 0 -> positions are
 * zero-filled).  Pure function of (sigma, level, seed).  tmp: ggd_surface_tmp_bytes(n) bytes of device scratch.
 */
size_t ggd_surface_tmp_bytes(int32_t n);
int ggd_surface_sample(ggd_ctx* ctx, void* stream, const float* sigma, int32_t n, float level, int32_t num_points,
                       float thickness, uint64_t seed, float* positions, uint32_t* num_faces, void* tmp,
                       size_t tmp_bytes);

/*
 * Fused per-point decoder, inference (bf16 MFMA, fp32 accumulate): the 5 chained `Decoder` MLPs of
 * main/decoder_models/sequential_decoder_reverse.py:68-85 in ONE launch.
 *   feat  [N,32]  mean-of-planes features (ggd_triplane_forward)      pos [N,3] positions
 *   packed_weights : ggd_decoder_packed_bytes() bytes, the LDS image of the 5 heads (bf16 weight rows pre-permuted to
 *                    the MFMA operand order + fp32 biases; built by gaussian_gan_decoder_amd.fused_decoder.pack_weights)
 *   attrs [N,16]  out: [0..2] color, [3] opacity, [4..7] rotation, [8..10] scale (= -softplus(s+5)-2.5),
 *                      [11..13] xyz (= head*0.01 + pos), [14..15] zero
 */
size_t ggd_decoder_packed_bytes(void);
/* Build the weight images on the device in ONE launch: params40 = HOST array of 40 DEVICE pointers, per head (colour,
 * opacity, rotation, scale, xyz) W1 b1 W2 b2 W3 b3 W4 b4 as torch.nn.Linear stores them (fp32, [out][in] row-major;
 * in = 35, 38, 39, 43, 46, out = 3, 1, 4, 3, 3); packed: ggd_decoder_packed_bytes(), packed_t (may be NULL):
 * ggd_decoder_packed_t_bytes().  Training calls it after every optimizer step. */
int ggd_decoder_pack(ggd_ctx* ctx, void* stream, const float* const* params40, void* packed, void* packed_t);
/* attrs rows [N][16] -> the five contiguous arrays the rasterizer entry takes (what the reference assigns to the GaussianModel,
 * main/train_pano2gaussian_decoder.py:223-227), and the gradients back into rows (a NULL array = zeros): one pass each way. */
int ggd_attrs_split(ggd_ctx* ctx, void* stream, const float* attrs, int64_t N, float* xyz, float* scale, float* rotation,
                    float* opacity, float* color);
int ggd_attrs_merge(ggd_ctx* ctx, void* stream, int64_t N, const float* dxyz, const float* dscale, const float* drotation,
                    const float* dopacity, const float* dcolor, float* dattrs);
int ggd_decoder_forward(ggd_ctx* ctx, void* stream, const float* feat, const float* pos, int32_t N,
                        const void* packed_weights, float* attrs);

/*
 * Fused decoder, TRAINING.  ggd_decoder_forward_train == ggd_decoder_forward that also keeps the hidden layers'
 * pre-activations: zbuf[5 heads][3 layers][ceil(N/16) blocks][16 points x 128] bf16 (ggd_decoder_zbuf_bytes(N); opaque:
 * inside a 4 KB block the values sit in the register order of the kernels, see csrc/ggd_mlp.hip).
 * ggd_decoder_backward: given dattrs[N,16] (gradient w.r.t. the attrs rows) it back-propagates through the 5 heads
 * (last first) with the TRANSPOSED weight image packed_t (ggd_decoder_packed_t_bytes(); built by
 * fused_decoder.pack_weights_t) and writes
 *   dzbuf (same size and layout as zbuf)  gradient at every hidden pre-activation
 *   dout  [5][N][4]      fp32  gradient at every head's raw output (columns >= the head's width are zero)
 *   dfeat [N,32]         fp32  gradient w.r.t. the plane features (sum over the 5 heads)
 *   dinfo [N,16]         fp32  scratch (gradient carried between heads through the chained inputs)
 * The weight / bias gradients are reductions over the N points of dz_l^T h_{l-1}; the caller forms them as split-K
 * GEMMs from dzbuf / dout and the activations (fused_decoder.FusedDecoderFn).
 */
size_t ggd_decoder_zbuf_bytes(int32_t N);
size_t ggd_decoder_packed_t_bytes(void);
int ggd_decoder_forward_train(ggd_ctx* ctx, void* stream, const float* feat, const float* pos, int32_t N,
                              const void* packed_weights, float* attrs, void* zbuf);
int ggd_decoder_backward(ggd_ctx* ctx, void* stream, int32_t N, const void* packed_t, const float* attrs,
                         const float* dattrs, const void* zbuf, void* dzbuf, float* dout, float* dfeat, float* dinfo);

/*
 * Weight / bias gradients of the fused decoder: split-K MFMA GEMMs over the points, operands transposed through LDS
 * (ds_read_b64_tr_b16), gelu(z) recomputed on the fly.  ACCUMULATES (fp32 atomics) into wgrad, which the caller
 * zero-initialises: ggd_decoder_wgrad_floats() floats = 5 heads x
 *   { dW1[128][64] db1[128] dW2[128][128] db2[128] dW3[128][128] db3[128] dW4[16][128] db4[16] }
 * (dW1 columns: 32 plane features, 3 position, then the earlier heads' outputs; columns / rows past the layer's real
 * width are padding).  feat / pos / attrs as in ggd_decoder_forward_train, zbuf from it, dzbuf / dout from
 * ggd_decoder_backward.
 */
size_t ggd_decoder_wgrad_floats(void);
int ggd_decoder_wgrad(ggd_ctx* ctx, void* stream, int32_t N, const void* zbuf, const void* dzbuf, const float* dout,
                      const float* feat, const float* pos, const float* attrs, float* wgrad);

/*
 * Fused image losses of the decoder training step and their gradient (main/train_pano2gaussian_decoder.py:246-261
 * without the external-network terms):
 *   loss = w[0]*L1 + w[1]*L2 + w[2]*(1 - SSIM) + w[3]*Sobel
 * with l1_loss / l2_loss / ssim of gaussian_splatting/utils/loss_utils.py:17-63 and sobel_loss of
 * main/loss_utils/sobel_loss.py:19-30.  image, target: device [3,H,W] fp32.  weights4: HOST array of 4 floats.
 * terms5 (device, 5 floats): L1, L2, 1 - SSIM, Sobel, weighted total.  grad_image (device [3,H,W]) = dloss/dimage.
 * tmp: device scratch of ggd_image_loss_tmp_bytes(W, H) bytes.  Three kernel launches, no host sync.
 */
size_t ggd_image_loss_tmp_bytes(int32_t W, int32_t H);
int ggd_image_loss(ggd_ctx* ctx, void* stream, int32_t W, int32_t H, const float* image, const float* target,
                   const float* weights4, float* terms5, float* grad_image, void* tmp, size_t tmp_bytes);

/* ggd_decoder_backward + ggd_decoder_wgrad over point chunks of `chunk` points (<= 0: one chunk), each chunk's weight-
 * gradient kernel launched right behind its backward kernel so that it reads dz / z from the Infinity Cache. */
int ggd_decoder_backward_wgrad(ggd_ctx* ctx, void* stream, int32_t N, int32_t chunk, const void* packed_t,
                               const float* attrs, const float* dattrs, const void* zbuf, void* dzbuf, float* dout,
                               float* dfeat, float* dinfo, const float* feat, const float* pos, float* wgrad);

/*
 * The fused decoder at REFERENCE PRECISION (the reference trains its decoder in fp32, main/decoder_models/base_decoder.py:8-27):
 * same kernels' structure on the bf16 matrix cores with every fp32 operand split into two bf16 numbers (x = hi + lo) and
 * every product evaluated as W_hi x_hi + W_hi x_lo + W_lo x_hi with fp32 accumulation (csrc/ggd_mlp_hl.inc), in the forward,
 * the backward and the weight gradients (outputs within 1e-5 of an fp32 evaluation).
 * The weight images have their own format (hi and lo image per layer): ggd_decoder_pack_hl builds both from the 40
 * parameter tensors (see ggd_decoder_pack).  zbuf: ggd_decoder_zbuf_bytes(N) as in the bf16 form, but holding fp16; dzbuf:
 * TWO bf16 planes (hi | lo), ggd_decoder_dzbuf_hl_bytes(N) = 2 x ggd_decoder_zbuf_bytes(N); dout / dfeat / dinfo / wgrad as
 * in the bf16 entry points (ggd_decoder_forward_hl with zbuf == NULL is the inference form).
 */
size_t ggd_decoder_packed_hl_bytes(void);
size_t ggd_decoder_dzbuf_hl_bytes(int32_t N);
size_t ggd_decoder_packed_t_hl_bytes(void);
int ggd_decoder_pack_hl(ggd_ctx* ctx, void* stream, const float* const* params40, void* packed_hl, void* packed_t_hl);
int ggd_decoder_forward_hl(ggd_ctx* ctx, void* stream, const float* feat, const float* pos, int32_t N,
                           const void* packed_hl, float* attrs, void* zbuf);
int ggd_decoder_backward_wgrad_hl(ggd_ctx* ctx, void* stream, int32_t N, int32_t chunk, const void* packed_t_hl,
                                  const float* attrs, const float* dattrs, const void* zbuf, void* dzbuf, float* dout,
                                  float* dfeat, float* dinfo, const float* feat, const float* pos, float* wgrad);

/* Per-stage device time (ms, hipEvent pairs on `stream`) of the most recent forward_geometry / forward_render /
 * backward call when profiling is on.  Stage names: ggd_stage_name(i), i in [0, ggd_stage_count()). */
int ggd_set_profiling(ggd_ctx* ctx, int enabled);
int ggd_stage_count(void);
const char* ggd_stage_name(int stage);
int ggd_stage_times(ggd_ctx* ctx, float* ms_out /* host, ggd_stage_count() floats, <0 = not run */);

#ifdef __cplusplus
}
#endif
#endif /* GGD_RASTER_H */
