/*
 * splatraster.h — C ABI of the MI355X-native differentiable Gaussian tile rasterizer
 * and of the 3-nearest-neighbour distance kernel used to seed Gaussian scales.
 *
 * This is the drop-in boundary for SplatLoc's native hot path.  The reference binds
 * this path through two CUDA extension modules whose sources are NOT vendored in
 * /root/reference (empty submodule dirs, .gitmodules:1-9):
 *
 *   diff_gauss.GaussianRasterizer.forward / backward
 *       call site  gaussian_splatting/gaussian_renderer/__init__.py:117-126  (forward)
 *       settings   gaussian_splatting/gaussian_renderer/__init__.py:42-55
 *       backward   triggered by train_gaussians.py:229 and :286 (loss.backward())
 *   simple_knn._C.distCUDA2
 *       call site  gaussian_splatting/scene/gaussian_model.py:18,206
 *
 * Everything here is plain C: raw device pointers, sizes, an opaque stream handle
 * (hipStream_t passed as void*), int status codes.  No torch types, no exceptions.
 * All float tensors are fp32, row-major, contiguous, resident in device memory.
 *
 * Memory ownership: the caller owns every buffer.  Scratch state that must survive
 * from forward to backward lives in three caller-allocated opaque buffers
 * (geometry / binning / image), sized by the *_bytes() queries below, so a framework's
 * caching allocator owns all device memory and several frames can be alive at once
 * (train_gaussians.py:195-229 keeps 5 forwards alive before one backward).
 */
#ifndef SPLATRASTER_H
#define SPLATRASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped on every change of a signature or buffer layout; the Python binding refuses a library
 * whose splatraster_abi_version() differs (a stale in-tree .so would otherwise be called through
 * ctypes with mismatched arguments) */
#define SPLATRASTER_ABI_VERSION 13

#define SPLATRASTER_TILE 16 /* tile edge in pixels (16x16 = 256 pixels = 4 wave64) */

/* status codes */
#define SPLATRASTER_OK 0
#define SPLATRASTER_ERR_BAD_ARG 1      /* null pointer / inconsistent sizes / both-or-neither inputs */
#define SPLATRASTER_ERR_HIP 2          /* a HIP runtime call or kernel launch failed (incl. a wedged look-back: hard spin bound -> trap) */
#define SPLATRASTER_ERR_UNSUPPORTED 3  /* e.g. sh_degree > 3 */
#define SPLATRASTER_ERR_OVERFLOW 4     /* tile instance count does not fit the binning buffer */
#define SPLATRASTER_WARN_LOOKBACK_STALL 5 /* splatraster_poll_errors() only: a scan / sort block waited longer than the soft
                                          * spin bound for a predecessor; NOT an error of any frame (results are late, never wrong) */

/*
 * Mirror of diff_gauss.GaussianRasterizationSettings
 * (gaussian_renderer/__init__.py:42-55) minus the tensors, which are passed as pointers.
 */
typedef struct splatraster_settings {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t sh_degree;   /* active SH degree (0..3); only used when shs != NULL */
    int32_t sh_coeffs;   /* M: coefficients per colour channel stored in shs [P, M, 3]; 0 when shs == NULL */
    int32_t channels;    /* C: composited channels. colors_precomp is [P, C]; with shs, C must be 3 */
    int32_t bg_channels; /* number of valid floats behind bg; channels >= bg_channels read 0 (train_gaussians.py:70 passes 3 for C = 4) */
    int32_t prefiltered;
    int32_t debug;
} splatraster_settings;

/* ---- buffer sizing ------------------------------------------------------------------ */

/* per-Gaussian forward state: projected centre, depth, conic+opacity, tile counts,
 * depth-sorted order, instance offsets, SH colours + clamp flags. */
size_t splatraster_geometry_bytes(int32_t P);
/* per-tile-instance state for R = num_rendered instances: sorted point list and the
 * sort's ping-pong buffers, the [tiles] range table, the per-instance payload (32-byte record
 * + quadrant reach mask) and, when channels % 4 != 0, the 16-byte-aligned feature table. */
size_t splatraster_binning_bytes(int32_t P, int64_t R, int32_t width, int32_t height, int32_t channels);
/* per-pixel forward state needed by backward: final transmittance, last contributor. */
size_t splatraster_image_bytes(int32_t width, int32_t height);

/* ---- forward ------------------------------------------------------------------------- */

/*
 * Stage 1 of forward (replaces the preprocess + InclusiveSum part of the reference
 * extension's forward).  Projects all P Gaussians, depth-sorts them, counts the tiles
 * each touches, and returns the total number of (tile, Gaussian) instances in
 * *num_rendered (one stream synchronisation to read a single int64).
 *
 * Exactly one of shs / colors_precomp and exactly one of (scales, rotations) /
 * cov3D_precomp must be non-NULL, as the reference wrapper enforces.
 *
 * radii [P] int32 is an output tensor (returned 4th by GaussianRasterizer.forward).
 */
int splatraster_forward_geometry(const splatraster_settings* s, int32_t P,
                                 const float* means3D,       /* [P,3] */
                                 const float* shs,           /* [P,M,3] or NULL */
                                 const float* opacities,     /* [P,1] */
                                 const float* scales,        /* [P,3] or NULL */
                                 const float* rotations,     /* [P,4] (w,x,y,z) or NULL */
                                 const float* cov3D_precomp, /* [P,6] or NULL */
                                 const float* viewmatrix,    /* [4,4] row-vector convention */
                                 const float* projmatrix,    /* [4,4] view @ proj */
                                 const float* campos,        /* [3] */
                                 void* geometry, int32_t* radii, int64_t* num_rendered,
                                 void* stream);

/*
 * Stage 2 of forward: instance emission, stable tile-bucket radix sort, tile ranges and
 * front-to-back alpha compositing.  R must be the value stage 1 returned and `binning`
 * must hold splatraster_binning_bytes(P, R, W, H, C) bytes.
 *
 * Outputs: out_color [C,H,W], out_depth [1,H,W], out_alpha [1,H,W]
 * (the first three members of the tuple GaussianRasterizer.forward returns,
 * gaussian_renderer/__init__.py:117).
 */
int splatraster_forward_render(const splatraster_settings* s, int32_t P, int64_t R,
                               const float* bg,             /* [bg_channels] */
                               const float* colors_precomp, /* [P,C] or NULL when shs were given */
                               void* geometry, void* binning, void* image,
                               float* out_color, float* out_depth, float* out_alpha,
                               void* stream);

/* ---- backward ------------------------------------------------------------------------ */

/*
 * Full backward (replaces the reference extension's rasterize_gaussians_backward).
 * Gradient outputs are OVERWRITTEN (zeroed inside).  dL_dmeans2D is [P,3] with
 * z = 0 and xy in NDC units (pixel gradient x 0.5*W / 0.5*H), which is what
 * GaussianModel.add_densification_stats reads (gaussian_model.py:677-679).
 * Optional outputs (dL_dshs, dL_dcov3D, dL_dscales, dL_drotations, dL_dcolors) may be
 * NULL when the matching forward input was NULL.
 */
int splatraster_backward(const splatraster_settings* s, int32_t P, int64_t R,
                         const float* bg, const float* means3D, const float* shs,
                         const float* colors_precomp, const float* opacities,
                         const float* scales, const float* rotations,
                         const float* cov3D_precomp, const float* viewmatrix,
                         const float* projmatrix, const float* campos,
                         const int32_t* radii,
                         void* geometry /* read + its backward scratch area is written */,
                         const void* binning, const void* image,
                         const float* out_color, const float* out_depth, const float* out_alpha,
                         const float* dL_dout_color, /* [C,H,W] */
                         const float* dL_dout_depth, /* [1,H,W] or NULL (= zeros) */
                         const float* dL_dout_alpha, /* [1,H,W] or NULL (= zeros) */
                         float* dL_dmeans3D,   /* [P,3] */
                         float* dL_dmeans2D,   /* [P,3] */
                         float* dL_dcolors,    /* [P,C] or NULL */
                         float* dL_dopacities, /* [P,1] */
                         float* dL_dscales,    /* [P,3] or NULL */
                         float* dL_drotations, /* [P,4] or NULL */
                         float* dL_dcov3D,     /* [P,6] or NULL */
                         float* dL_dshs,       /* [P,M,3] or NULL */
                         /* pose-gradient extension (no counterpart in the reference):
                          * exact derivative w.r.t. the camera tensors, or NULL to skip */
                         float* dL_dviewmatrix, /* [4,4] or NULL */
                         float* dL_dprojmatrix, /* [4,4] or NULL */
                         float* dL_dcampos,     /* [3] or NULL (non-zero only with shs) */
                         void* stream);

/* ---- a WINDOW of views in one launch sequence -------------------------------------------- */

/*
 * SplatLoc.map renders `window_size` (5) views of one Gaussian scene, sums their losses and runs ONE
 * backward (train_gaussians.py:195-229); color_refinement and eval_rendering call the rasterizer once per
 * view.  The reference extension has only the per-view call, so that loop is 5 x (preprocess, two sorts,
 * compositing) + autograd's accumulation of 5 gradient sets.  These entry points take the V <= 8 views of a
 * window at once: one preprocess over the P Gaussians for all views, ONE depth sort of the V * P (view,
 * Gaussian) rows, ONE tile sort keyed by (view, tile), one compositing grid over V * tiles * 4 quadrant-waves —
 * a 640x480 frame alone leaves most of an MI355X idle — and a backward that sums the V views into ONE set of
 * parameter gradients.  Results per view are bit-identical to V calls of the functions above (both sorts are
 * stable: every (view, tile) list is in (depth, index) order); only the float-atomic summation order of the
 * gradients differs, as it does from run to run anyway.  All views share the settings' image size, channel
 * count, scale modifier and background; tanfovx / tanfovy are per view (settings->tanfovx/y are ignored here).
 * Colours must be precomputed (`colors_precomp`, SplatLoc's configuration): view-dependent SH colours and the
 * pose-gradient extension stay on the per-view call (SPLATRASTER_ERR_UNSUPPORTED here).
 */
#define SPLATRASTER_MAX_WINDOW_VIEWS 8

typedef struct splatraster_window_view {
    const float* viewmatrix; /* [4,4] device memory, row-vector convention */
    const float* projmatrix; /* [4,4] view @ proj */
    const float* campos;     /* [3] or NULL (unused with precomputed colours) */
    float tanfovx, tanfovy;
    int32_t* radii;          /* [P] int32: forward output (4th member of the tuple), read again by the backward */
    float* out_color;        /* [C,H,W] forward output; read by the backward */
    float* out_depth;        /* [1,H,W] */
    float* out_alpha;        /* [1,H,W] */
    /* backward only (ignored by the forward calls) */
    const float* dL_dout_color; /* [C,H,W]; or [color_grad_channels,H,W] when that is in [1, C) */
    const float* dL_dout_depth; /* [1,H,W] or NULL (= zeros) */
    const float* dL_dout_alpha; /* [1,H,W] or NULL (= zeros) */
    float* dL_dmeans2D;         /* [P,3] output: per view, as GaussianModel.add_densification_stats reads it */
    /* SplatLoc's render() hands the colour buffer out as TWO tensors, render = image[:3] and kp_prob = image[-1]
     * (gaussian_renderer/__init__.py:133-135), and the losses produce their gradients separately — through autograd
     * each arrives zero-padded to [C,H,W] and the two are added (five elementwise kernels per view).  A caller that
     * has the two gradients apart passes them apart: color_grad_channels = C - 1 planes behind dL_dout_color and the
     * last channel's plane behind dL_dout_last, or NULL when that channel did not reach the loss (color_refinement:
     * RGB only, train_gaussians.py:283-285) — the backward then skips the channel altogether.  0 = all C planes are
     * behind dL_dout_color (the plain call).  Wider tables ([rgb | features | kp_score]): any g in [1, C) — the first g
     * planes behind dL_dout_color, the last channel behind dL_dout_last, the channels in between without a gradient
     * (their planes are neither read nor built).  One value per launch. */
    const float* dL_dout_last;  /* [1,H,W] or NULL */
    int32_t color_grad_channels; /* 0 (= C), or g in [1, C) */
} splatraster_window_view;

size_t splatraster_window_geometry_bytes(int32_t P, int32_t n_views);
size_t splatraster_window_binning_bytes(int32_t P, int32_t n_views, int64_t R_total, int32_t width, int32_t height,
                                        int32_t channels);
size_t splatraster_window_image_bytes(int32_t width, int32_t height, int32_t n_views);

/* Stage 1: num_rendered[v] = tile instances of view v (one stream synchronisation for the whole window). */
int splatraster_forward_window_geometry(const splatraster_settings* s, int32_t n_views,
                                        const splatraster_window_view* views, int32_t P,
                                        const float* means3D, const float* opacities, const float* scales,
                                        const float* rotations, const float* cov3D_precomp, void* geometry,
                                        int64_t* num_rendered /* [n_views] */, void* stream);
/* Stage 2: `binning` holds splatraster_window_binning_bytes(P, n_views, sum(num_rendered), W, H, C) bytes. */
int splatraster_forward_window_render(const splatraster_settings* s, int32_t n_views,
                                      const splatraster_window_view* views, int32_t P,
                                      const int64_t* num_rendered, const float* bg, const float* colors_precomp,
                                      void* geometry, void* binning, void* image, void* stream);
/* Backward of the whole window: the parameter gradients are the SUM over the views (written once, overwritten),
 * dL_dmeans2D per view.  `bg`: the background the forward was given (NULL = zeros) — the back-to-front walk starts every pixel
 * at A = bg . g - g_A (round 4; rounds 1-3 took the background's share from the forward's colour planes, which the backward
 * no longer reads). */
int splatraster_backward_window(const splatraster_settings* s, int32_t n_views, const splatraster_window_view* views,
                                int32_t P, const int64_t* num_rendered, const float* bg, const float* means3D,
                                const float* colors_precomp, const float* scales, const float* rotations,
                                const float* cov3D_precomp, void* geometry, const void* binning, const void* image,
                                float* dL_dmeans3D,   /* [P,3] */
                                float* dL_dcolors,    /* [P,C] */
                                float* dL_dopacities, /* [P,1] */
                                float* dL_dscales,    /* [P,3] or NULL */
                                float* dL_drotations, /* [P,4] or NULL */
                                float* dL_dcov3D,     /* [P,6] or NULL */
                                void* stream);

/* The same backward, writing the gradients of SplatLoc's RAW parameters directly (SH degree 0, scales + rotations; what
 * gaussian_model.py:78-105 and gaussian_renderer/__init__.py:84-102 put between the optimiser tensors and the rasterizer call):
 *   scales = exp(scaling), rotations = normalize(rotation), opacities = sigmoid(opacity),
 *   colors_precomp = [clamp_min(C0 f_dc + 0.5, 0) | extra]                                       (C = 3 + extra_channels)
 * The chain through these activations and the sum of the accumulator rows' colour columns run inside the per-Gaussian
 * backward kernel: no dL/dcolors / dL/dopacities / dL/dscales / dL/drotations tensors, no second pass over them.  `scales`,
 * `rotations`, `colors_precomp` are the ACTIVATED tensors the forward was given; results are bit-identical to
 * splatraster_backward_window followed by splatraster_activate_backward.  extra_channels <= 1 (C <= 4: the accumulator rows' colour
 * columns share the 64-byte line of the moments the kernel reads anyway); wider layouts: SPLATRASTER_ERR_UNSUPPORTED — use the plain call. */
/* Stage 1 of the window forward from the RAW parameters: the activations run inside the projection kernel, which uses the activated
 * values and writes them — `scales`, `rotations`, `opacities`, `colors` are OUTPUTS of this call (the render stage takes `colors` as
 * its colors_precomp, the backward `scales` / `rotations`) — bit-identical to splatraster_activate_forward followed by
 * splatraster_forward_window_geometry. */
typedef struct splatraster_raw_forward {
    const float* scaling;  /* [P,3] */
    const float* rotation; /* [P,4] */
    const float* opacity;  /* [P]   logits */
    const float* f_dc;     /* [P,3] */
    const float* extra;    /* [P,extra_channels] or NULL */
    int32_t extra_channels;
    float* scales;         /* [P,3]   out */
    float* rotations;      /* [P,4]   out */
    float* opacities;      /* [P]     out */
    float* colors;         /* [P,3 + extra_channels] out */
} splatraster_raw_forward;
int splatraster_forward_window_geometry_raw(const splatraster_settings* s, int32_t n_views, const splatraster_window_view* views,
                                            int32_t P, const float* means3D, const splatraster_raw_forward* raw, void* geometry,
                                            int64_t* num_rendered /* [n_views] */, void* stream);

typedef struct splatraster_raw_params {
    const float* scaling;  /* [P,3] */
    const float* rotation; /* [P,4] */
    const float* opacity;  /* [P]   logits */
    const float* f_dc;     /* [P,3] */
    int32_t extra_channels;
    float* dL_dscaling;    /* [P,3] */
    float* dL_drotation;   /* [P,4] */
    float* dL_dopacity;    /* [P]   */
    float* dL_df_dc;       /* [P,3] */
    float* dL_dextra;      /* [P,extra_channels] or NULL */
    /* optional (NULL: none): SplatLoc.map's isotropic regulariser (train_gaussians.py:221-228; splatraster_isotropic_loss's row_grad and
     * out) adds reg_weight * reg_out[1] * reg_row_grad[i] to dL/dscales[i, :] before the chain through exp */
    const float* reg_row_grad; /* [P] */
    const float* reg_out;      /* [2] (device) */
    float reg_weight;
} splatraster_raw_params;
int splatraster_backward_window_raw(const splatraster_settings* s, int32_t n_views, const splatraster_window_view* views,
                                    int32_t P, const int64_t* num_rendered, const float* bg, const float* means3D,
                                    const float* colors_precomp, const float* scales, const float* rotations, void* geometry,
                                    const void* binning, const void* image, const splatraster_raw_params* raw,
                                    float* dL_dmeans3D /* [P,3] */, void* stream);

/* ---- auxiliary entry points ---------------------------------------------------------- */

/* present[i] = 1 when Gaussian i passes the near-plane test (view-space z > 0.2).
 * Mirrors GaussianRasterizer.markVisible of the extension's Python wrapper. */
int splatraster_mark_visible(int32_t P, const float* means3D, const float* viewmatrix,
                             const float* projmatrix, uint8_t* present, void* stream);

/* Debug / test introspection: byte offsets of the arrays inside the opaque buffers. */
typedef struct splatraster_geometry_layout {
    size_t rec0;          /* 32-byte records, stride 32: float4 at rec0 + 32 i = pixel x, pixel y, view depth, radius (float, 0 = culled) */
    size_t rec1;          /* = rec0 + 16: float4 at rec1 + 32 i = conic a, b, c, opacity */
    size_t tiles_touched; /* uint32[P] (original index order) */
    size_t depth_order;   /* uint32[P]: Gaussian indices (bits 0..23; bits 24..31 = min(tiles_touched, 255)), stable-sorted by depth bits (culled last) */
    size_t offsets;       /* uint32[P]: inclusive scan of tiles_touched in depth_order */
    size_t rgb;           /* float[3P]: SH colours (only with shs) */
    size_t clamped;       /* uint8[3P]: SH clamp flags (only with shs) */
    size_t total;
} splatraster_geometry_layout;
typedef struct splatraster_binning_layout {
    size_t point_list; /* uint32[R]: Gaussian index per instance, sorted by (tile, depth, index) */
    size_t tile_list;  /* uint32[R]: tile id per sorted instance */
    size_t ranges;     /* uint32[2*tiles]: [start, end) per tile */
    size_t total;
} splatraster_binning_layout;
typedef struct splatraster_image_layout {
    size_t final_T;   /* float[H*W] */
    size_t n_contrib; /* uint32[H*W] */
    size_t total;
} splatraster_image_layout;
int splatraster_get_geometry_layout(int32_t P, splatraster_geometry_layout* out);
int splatraster_get_binning_layout(int32_t P, int64_t R, int32_t width, int32_t height, int32_t channels,
                                   splatraster_binning_layout* out);
int splatraster_get_image_layout(int32_t width, int32_t height, splatraster_image_layout* out);
/* the same for the buffers of a window: arrays indexed by Gaussian hold n_views * P rows (row v * P + i), the
 * instance lists hold rows and GLOBAL tile ids v * tiles + t, `ranges` has 2 * n_views * tiles entries and the
 * per-pixel planes n_views * H * W */
int splatraster_get_window_geometry_layout(int32_t P, int32_t n_views, splatraster_geometry_layout* out);
int splatraster_get_window_binning_layout(int32_t P, int32_t n_views, int64_t R_total, int32_t width, int32_t height,
                                          int32_t channels, splatraster_binning_layout* out);
int splatraster_get_window_image_layout(int32_t width, int32_t height, int32_t n_views, splatraster_image_layout* out);

/* Stable LSD radix sort of (key, value) pairs on key bits [0, key_bits); exposed so the
 * sort can be tested and timed on its own.  tmp must hold splatraster_sort_tmp_bytes(n). */
size_t splatraster_sort_tmp_bytes(int64_t n);
int splatraster_sort_pairs_u32(int64_t n, uint32_t* keys, uint32_t* vals, int32_t key_bits,
                               void* tmp, void* stream);

/* ---- parameter activations + SH / feature packing -------------------- */

/* The elementwise work between SplatLoc's raw optimiser tensors and the rasterizer call, as one
 * kernel each way.  Replaces, with identical results:
 *   scales    = exp(_scaling)             gaussian_model.py:78-80; an isotropic [P,1] model is
 *                                         repeated to 3 columns (gaussian_renderer/__init__.py:73-76)
 *   rotations = normalize(_rotation)      gaussian_model.py:82-84 (eps 1e-12)
 *   opacities = sigmoid(_opacity)         gaussian_model.py:101-103
 *   colors    = cat(clamp_min(eval_sh(active_sh_degree, cat(f_dc, f_rest), normalize(xyz - campos))
 *                             + 0.5, 0), extra)
 *                                         gaussian_renderer/__init__.py:84-102, sh_utils.py:55-118
 * sh_coeffs = (max_sh_degree + 1)^2 >= (active_sh_degree + 1)^2, active_sh_degree in [0, 3];
 * f_dc is [P,1,3], f_rest [P,sh_coeffs-1,3] (NULL when sh_coeffs == 1); scaling_cols is 3 or 1;
 * extra is [P,extras] (SplatLoc: the kp_score column) or NULL with extras == 0. */
int splatraster_activate_forward(int32_t P, int32_t sh_coeffs, int32_t active_sh_degree,
                                 int32_t scaling_cols, int32_t extras,
                                 const float* xyz, const float* f_dc, const float* f_rest,
                                 const float* scaling, const float* rotation, const float* opacity,
                                 const float* extra, const float* campos,
                                 float* scales /* [P,3] */, float* rotations /* [P,4] */,
                                 float* opacities /* [P,1] */, float* colors /* [P,3+extras] */,
                                 void* stream);
/* Gradients w.r.t. the raw tensors given dL/d(scales, rotations, opacities, colors); recomputes
 * the activations from the raw inputs (nothing is saved by the forward).  dL_dxyz receives ONLY
 * the view-direction term of the SH colour (zero at degree 0; the rasterizer's own dL/dmeans3D
 * is added by the caller) and may be NULL; dL_df_rest / dL_dextra may be NULL when empty.
 * The camera centre gets no gradient. */
int splatraster_activate_backward(int32_t P, int32_t sh_coeffs, int32_t active_sh_degree,
                                  int32_t scaling_cols, int32_t extras,
                                  const float* xyz, const float* f_dc, const float* f_rest,
                                  const float* scaling, const float* rotation, const float* opacity,
                                  const float* campos,
                                  const float* dL_dscales, const float* dL_drotations,
                                  const float* dL_dopacities, const float* dL_dcolors,
                                  float* dL_dxyz, float* dL_df_dc, float* dL_df_rest, float* dL_dscaling,
                                  float* dL_drotation, float* dL_dopacity, float* dL_dextra, void* stream);

/* Per-view densification statistics, in place: for every Gaussian with
 * radii > 0:  max_radii2D = max(max_radii2D, radii) (train_gaussians.py:240-244),
 * xyz_gradient_accum += ||viewspace_grad[:2]||, denom += 1 (gaussian_model.py:677-679). */
int splatraster_densification_stats(int32_t P, const float* viewspace_grad /* [P,3] */, const int32_t* radii,
                                    float* xyz_gradient_accum /* [P,1] */, float* denom /* [P,1] */,
                                    float* max_radii2D /* [P] */, void* stream);
/* The same for the n_views <= SPLATRASTER_MAX_WINDOW_VIEWS views of a window in ONE launch, in view order (host arrays
 * of device pointers).  xyz_gradient_accum == denom == NULL: only max_radii2D is updated — the statistics line of
 * SplatLoc.color_refinement (train_gaussians.py:293-294); viewspace_grads may then be NULL. */
int splatraster_densification_stats_window(int32_t P, int32_t n_views, const float* const* viewspace_grads,
                                           const int32_t* const* radii, float* xyz_gradient_accum, float* denom,
                                           float* max_radii2D, void* stream);

/* ---- densify / clone / split / prune + Adam over the parameter groups -------- */

/* The raw (pre-activation) parameter tensors of the reference's GaussianModel, in the group order of
 * GaussianModel.training_setup (gaussian_model.py:254-297): row-major [P, width], fp32, device memory.
 * The same struct carries the Adam moments (exp_avg / exp_avg_sq of each group; NULL = the group has no
 * optimizer state — SplatLoc's marker never receives a gradient in map()). */
typedef struct splatraster_model {
    int32_t P;
    int32_t f_rest_width;  /* 3 * ((max_sh_degree + 1)^2 - 1); 0 in SplatLoc (f_rest is [P, 0, 3]) */
    int32_t marker_width;  /* 1 (0 = absent) */
    int32_t kp_width;      /* columns of _kp_score (1 in SplatLoc) */
    int32_t scaling_width; /* 3, or 1 for an isotropic model */
    float* xyz;            /* [P,3] */
    float* f_dc;           /* [P,1,3] */
    float* f_rest;         /* [P,f_rest_width] */
    float* opacity;        /* [P,1]  logit */
    float* marker;         /* [P,marker_width] */
    float* kp_score;       /* [P,kp_width] */
    float* scaling;        /* [P,scaling_width]  log */
    float* rotation;       /* [P,4]  un-normalised quaternion (w,x,y,z) */
} splatraster_model;

/* GaussianModel.densify_and_prune(max_grad, min_opacity, extent, max_screen_size)
 * (gaussian_model.py:655-675 with densify_and_clone :632-653, densify_and_split :590-630, prune_points
 * :510-526 and the optimizer surgery :477-587) as one compaction:
 *   plan   decides clone / split / prune per row from xyz_gradient_accum / denom, the scales, opacity and
 *          (primitive_reg) the marker, scans the keep flags and returns the new row count (one
 *          device->host read: it sizes the caller's allocations);
 *   apply  writes the re-sized parameter tensors and Adam moments in the reference's row order
 *          [originals | clones | split children copy 0 | copy 1]; new rows get zero moments.
 * The caller zeroes the statistics afterwards (densification_postfix resets xyz_gradient_accum, denom
 * AND max_radii2D, gaussian_model.py:585-587 — which is why max_screen_size can only act as the on/off
 * switch of the world-size prune: use_size_prune = (max_screen_size != 0)).
 * Split children: xyz + R(q/|q|) (z * exp(scaling)), z ~ N(0, I).  unit_noise [2,P,3] supplies z by
 * (copy, source row) when given (tests: the recorded table of tests/golden/densify.npz); otherwise z is
 * drawn from Philox4x32-10 keyed by (seed, draw_id, source row, copy), so data-parallel replicas that pass
 * the same (seed, draw_id) split identically without a broadcast. */
size_t splatraster_densify_workspace_bytes(int32_t P);
int splatraster_densify_plan(const splatraster_model* model, const float* xyz_gradient_accum /* [P,1] */,
                             const float* denom /* [P,1] */, float max_grad, float min_opacity, float extent,
                             float percent_dense, int32_t use_size_prune, int32_t primitive_reg, void* workspace,
                             int32_t* new_P, void* stream);
int splatraster_densify_apply(const splatraster_model* model, const splatraster_model* exp_avg /* or NULL */,
                              const splatraster_model* exp_avg_sq /* or NULL */, const float* unit_noise /* or NULL */,
                              uint64_t seed, uint64_t draw_id, void* workspace /* from plan */, int32_t new_P,
                              splatraster_model* out_model, splatraster_model* out_exp_avg,
                              splatraster_model* out_exp_avg_sq, int32_t* source_row /* [new_P] or NULL */,
                              int32_t* source_kind /* [new_P] or NULL: 0 original, 1 clone, 2 / 3 split child */,
                              void* stream);

/* Key-frame insertion, GaussianModel.extend_from_pcd -> densification_postfix -> cat_tensors_to_optimizer
 * (gaussian_model.py:222-241, :528-587; once per key-frame, train_gaussians.py:173-177): the `extra->P` new rows are
 * appended to the 8 parameter tensors and ZERO rows to the Adam moments of every group that has state — ONE launch
 * instead of 24 torch.cat + 16 zeros_like.  out_* hold model->P + extra->P rows; exp_avg / exp_avg_sq (and their
 * outputs) may be NULL, or carry NULL members for groups without state.  The caller resets the densification
 * statistics, as densification_postfix does (:585-587). */
int splatraster_model_append(const splatraster_model* model, const splatraster_model* exp_avg /* or NULL */,
                             const splatraster_model* exp_avg_sq /* or NULL */, const splatraster_model* extra,
                             splatraster_model* out_model, splatraster_model* out_exp_avg,
                             splatraster_model* out_exp_avg_sq, void* stream);

/* One torch.optim.Adam step (no weight decay, no amsgrad: gaussian_model.py:287) over up to 16 parameter
 * groups in ONE launch.  `step` is the group's step count AFTER this step (>= 1): the bias corrections
 * and 1 - beta are formed on the host in double like torch does (hence the double betas).  A group with grad == NULL is skipped (torch
 * semantics: no state is touched).  row_gate (or NULL): per-ROW gate values; rows with
 * gate > row_gate_threshold see a ZERO gradient — the key-primitive freeze `get_xyz.grad[key_mask] = 0`
 * of train_gaussians.py:231-234 (gate = the marker, threshold 0.005) without its .cpu() sync. */
typedef struct splatraster_adam_group {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    const float* row_gate;
    int64_t numel;
    int32_t row_width; /* elements per row (for row_gate) */
    float lr;
    double step;
} splatraster_adam_group;
int splatraster_adam_step(int32_t n_groups, const splatraster_adam_group* groups, double beta1, double beta2, double eps,
                          float row_gate_threshold, void* stream);
/* splatraster_adam_step and, in the same launch, the statistics line of SplatLoc.color_refinement (train_gaussians.py:293-294):
 * max_radii2D[i] = max(max_radii2D[i], radii[i]) wherever radii[i] > 0 (radii: the int32 [P] output of the frame's forward).
 * n_groups may be 0 (only the statistics). */
int splatraster_adam_step_radii(int32_t n_groups, const splatraster_adam_group* groups, double beta1, double beta2, double eps,
                                float row_gate_threshold, int32_t P, const int32_t* radii, float* max_radii2D, void* stream);

/* Isotropic scale regulariser of SplatLoc.map (train_gaussians.py:222-228):
 *   mask = marker > 0.005;  loss = mean_{mask} | mean_k scaling[i,k] / (0.02 (1 - marker_i)) - 1 |
 * scaling is the ACTIVATED [P, scaling_cols] tensor.  row_grad[i] = d|x_i - 1| / d scaling[i,k] (equal for
 * every k; 0 outside the mask); out[0] = loss, out[1] = 1 / |mask| (both 0 for an empty mask), so
 * d loss / d scaling[i,k] = out[1] * row_grad[i].  No host synchronisation. */
size_t splatraster_isotropic_loss_workspace_bytes(int32_t P);
int splatraster_isotropic_loss(int32_t P, int32_t scaling_cols, const float* scaling, const float* marker,
                               float* row_grad /* [P] */, float* out /* [2] */, void* workspace, void* stream);

/* ---- per-view mapping loss + gradient ---------------------------------- */

/* loss = get_loss_mapping(config, image, depth, viewpoint, opacity) (utils/utils.py:55-82; exposure
 * affine exp(a) image + b unless exposure == NULL, i.e. initialization=True)
 *      + get_loss_marker(config, marker, viewpoint.kp_score) (train_gaussians.py:38-42),
 * the per-view sum of train_gaussians.py:217-218, and its gradient w.r.t. the rendered buffers,
 * in one pass.  image / g_image are 3 planes of H*W floats with plane stride H*W (they may point
 * into a [C,H,W] render and its gradient), marker / g_marker one plane.  kp is the float32
 * key-point score map used as the (soft) BCE target, `gt.view(-1).float()` at train_gaussians.py:40
 * (a bool mask is passed as 0.0 / 1.0).  out[4] = { rgbd loss, marker loss, dL/dexposure_a, dL/dexposure_b } (device). */
size_t splatraster_mapping_loss_workspace_bytes(int32_t pixels);
int splatraster_mapping_loss(int32_t pixels, const float* image, const float* depth, const float* marker,
                             const float* gt_image, const float* gt_depth, const float* kp,
                             float rgb_boundary_threshold, const float* exposure /* [2] = a, b or NULL */,
                             float* g_image, float* g_depth, float* g_marker, float* out /* [4] */,
                             void* workspace, void* stream);

/* The same for the n_views <= SPLATRASTER_MAX_WINDOW_VIEWS views of a window (the loop train_gaussians.py:195-219 sums the
 * per-view losses with weight 1) in ONE launch pair: `views` is a HOST array of per-view device pointers, out is [n_views][4]
 * (device), workspace n_views x splatraster_mapping_loss_workspace_bytes(pixels).  Per-view results are bit-identical to
 * n_views calls of splatraster_mapping_loss (same blocks, same summation order). */
typedef struct splatraster_loss_view {
    const float* image;    /* 3 planes */
    const float* depth;
    const float* marker;
    const float* gt_image; /* 3 planes */
    const float* gt_depth;
    const float* kp;
    const float* exposure; /* [2] = a, b or NULL */
    float* g_image;        /* 3 planes */
    float* g_depth;
    float* g_marker;
} splatraster_loss_view;
int splatraster_mapping_loss_window(int32_t n_views, int32_t pixels, const splatraster_loss_view* views,
                                    float rgb_boundary_threshold, float* out /* [n_views][4] */, void* workspace,
                                    void* stream);

/* Colour-refinement loss (train_gaussians.py:283-285):
 *   loss = (1 - lambda_dssim) l1_loss(image, gt) + lambda_dssim (1 - ssim(image, gt))
 * with l1_loss / ssim of gaussian_splatting/utils/loss_utils.py:21-22, 42-102 (11x11 Gaussian
 * window, sigma 1.5, zero padding), and its gradient w.r.t. image.  image, gt, g_image: [C,H,W].
 * out[3] = { l1, ssim, loss } (device). */
size_t splatraster_refinement_loss_workspace_bytes(int32_t channels, int32_t height, int32_t width);
int splatraster_refinement_loss(int32_t channels, int32_t height, int32_t width, float lambda_dssim,
                                const float* image, const float* gt, float* g_image, float* out /* [3] */,
                                void* workspace, void* stream);

/* Per-frame metrics of eval_rendering (utils/eval_utils.py:45-52): image = clamp(render, 0, 1), mask = gt > 0 per element,
 *   psnr = 20 log10(1 / sqrt(mean_{mask} (image - gt)^2))   (gaussian_splatting/utils/image_utils.py:19-21)
 *   ssim = ssim(image, gt)                                  (loss_utils.py:61-102)
 * render, gt: [C,H,W] (render is clamped on load; nothing is written back).  out[4] = { psnr, ssim, masked mse, mask count }
 * (device).  LPIPS (a learned network, torchmetrics) is not part of this library. */
size_t splatraster_eval_metrics_workspace_bytes(int32_t channels, int32_t height, int32_t width);
int splatraster_eval_metrics(int32_t channels, int32_t height, int32_t width, const float* render, const float* gt,
                             float* out /* [4] */, void* workspace, void* stream);

/* ---- simple_knn._C.distCUDA2 (gaussian_model.py:206) ---------------------------------- */

size_t splatknn_workspace_bytes(int32_t N);
/* out[i] = mean of the squared distances from points[i] to its 3 nearest other points.  Exact: a tiled brute force below
 * 10 000 points, an exact uniform-grid search from there on (SplatLoc passes 5-20 k per key-frame) — same distance
 * arithmetic, bit-identical results.  No host synchronisation. */
int splatknn_dist2(int32_t N, const float* points /* [N,3] */, float* out /* [N] */,
                   void* workspace, void* stream);

/* test hook: point count from which splatknn_dist2 takes the grid search (< 0 restores the default 10 000) */
int splatknn_debug_set_grid_min(int32_t n);

/* ---- pose refinement on the device (build extension, DESIGN.md §6.8: the reference's rasterizer returns no camera gradient
 * and nothing calls its utils/optimization_utils.py:5-66 pose helpers) ------------------------------------------------------
 * L = mean |color - target_color| + depth_weight * mean |depth - target_depth| and its gradient planes in one pass;
 * `*loss_out += L` (a float atomic: zero it before the first call; monitoring value).  target_depth may be NULL (no depth
 * term; g_depth, when given, is zeroed). */
int splatraster_l1_rgbd_loss(int64_t n_color, const float* color, const float* target_color, int64_t n_depth, const float* depth,
                             const float* target_depth, float depth_weight, float* g_color, float* g_depth, float* loss_out,
                             void* stream);
/* One Adam step on a camera pose and the camera tensors of the new pose, without leaving the device.
 * The pose is W2C = T(w, t) @ W2C_init with an axis-angle w and a translation t — at_to_transform_matrix of
 * utils/optimization_utils.py:31-42 (Rodrigues made regular at w = 0) —, the camera tensors are what
 * utils/camera_utils.py:129-139 derives from it: viewmatrix = W2C^T, projmatrix = viewmatrix @ projection_matrix,
 * campos = -R^T t.  state [20 floats, device]: w[3], t[3], Adam exp_avg[6], exp_avg_sq[6], step, pad — all zero at the start.
 * advance != 0: chains dL_dviewmatrix [16], dL_dprojmatrix [16], dL_dcampos [3 or NULL] (splatraster_backward's outputs) to
 * (w, t), takes the torch.optim.Adam step (lr_rot for w, lr_trans for t) and writes the NEW pose's tensors;
 * advance == 0: only writes the tensors of the current state (the first iteration).  All matrices row-major [4, 4]. */
int splatraster_pose_step(const float* dL_dviewmatrix, const float* dL_dprojmatrix, const float* dL_dcampos, const float* W2C_init,
                          const float* projection_matrix, float lr_rot, float lr_trans, float beta1, float beta2, float eps,
                          int advance, float* state, float* viewmatrix, float* projmatrix, float* campos, void* stream);

/* ---- per-stage timing (HIP events on the launch stream) ----------------------------- */

/* Stage ids: every kernel group of the path is bracketed by a hipEvent pair when timing
 * is enabled (process-wide switch; off by default).  An event record between two kernels
 * costs ~10 us of idle GPU on MI355X, so a throughput measurement should select only the
 * stage it needs (splatraster_timing_select) and take the full breakdown in a separate run. */
#define SPLATRASTER_STAGE_PREPROCESS 0     /* preprocess_kernel */
#define SPLATRASTER_STAGE_DEPTH_SORT 1     /* depth keys + P-sized radix sort */
#define SPLATRASTER_STAGE_SCAN 2           /* inclusive scan of tiles_touched */
#define SPLATRASTER_STAGE_EMIT 3           /* emit_kernel */
#define SPLATRASTER_STAGE_TILE_SORT 4      /* R-sized radix sort on tile id */
#define SPLATRASTER_STAGE_RANGES 5         /* clearing the per-tile range table when nothing was emitted; the launch order of small grids (tile_order_kernel) */
#define SPLATRASTER_STAGE_COMPOSITE_FWD 6  /* composite_fwd_kernel */
#define SPLATRASTER_STAGE_COMPOSITE_BWD 7  /* composite_bwd_kernel (the accumulator memset before it is not bracketed) */
#define SPLATRASTER_STAGE_PREPROCESS_BWD 8 /* preprocess_bwd_kernel */
#define SPLATRASTER_STAGE_PAYLOAD 9        /* payload_kernel: per-instance records, reach masks, tile ranges */
#define SPLATRASTER_STAGE_COUNT 10

int splatraster_timing_enable(int on);            /* all stages on / off */
int splatraster_timing_select(uint32_t stage_mask); /* bit s set: time stage s only */
/* Waits for all recorded events, ADDS elapsed milliseconds / launch counts per stage into
 * ms[SPLATRASTER_STAGE_COUNT] / counts[SPLATRASTER_STAGE_COUNT], then clears the records. */
int splatraster_timing_collect(double* ms, int64_t* counts);

/* ---- misc ---------------------------------------------------------------------------- */

/* The decoupled look-backs of the one-pass scan and of the one-sweep radix sort never give up with a
 * partial prefix (that would be a silently mis-sorted frame).  Soft bound: a block that has waited longer
 * than the spin limit raises a flag in host-mapped memory and keeps waiting — the frame stays correct, so
 * forward / backward / sort_pairs still return OK; splatraster_poll_errors() (exact after a stream
 * synchronize) returns SPLATRASTER_WARN_LOOKBACK_STALL once per raised flag, for monitoring.  Hard bound
 * (2^30 spins): the block traps, the kernel aborts and the following HIP calls fail -> SPLATRASTER_ERR_HIP:
 * a wedged device is a fault, never a silent hang. */
int splatraster_poll_errors(void);
/* test hooks: SOFT spin bound of the look-backs (default 1 << 24; 0 makes every block that has to wait at
 * all report), and y[i] = the device's 2^x (the alpha arithmetic shared with the CPU oracle). */
int splatraster_debug_set_spin_limit(uint32_t limit);
/* Deterministic debug mode of the backward (process-wide switch, default off).  The compositing backward sums
 * per-(wave, Gaussian) partials into per-Gaussian rows with float atomics, whose arrival order — and therefore the last bits of
 * every gradient — changes from run to run.  With the switch on,
 *   (1) each partial is converted to fixed point and accumulated with 64-bit INTEGER atomics (associative: the totals are
 *       bit-reproducible); the fixed point is chosen PER ELEMENT, 2^-44 of the largest partial the element receives
 *       (a first pass of the kernel takes that maximum, also an associative reduction) — round 3's single 2^-40 quantised the
 *       small rows of a large scene;
 *   (2) the per-pixel state of the back-to-front walk (T, A) is carried in double; split launches are not used.
 * Two passes of the kernel (~2.5 x slower).  Meant for regression hunting and the strictest tests.  (The NORMAL path is accurate
 * per gradient row since round 4 — it walks the lists back to front like the lineage; rounds 1-3 walked front to back and
 * carried an absolute float32 error in the suffix sum.) */
int splatraster_debug_set_deterministic(int on);
/* A/B hook: launches of the C = 4..15 backward with at most this many quadrant-waves take the small-layout panel
 * variant (HISTORY.md §11); < 0 restores the built-in default. */
int splatraster_debug_set_small_panel_max_waves(int waves);
/* A/B hook: narrow-layout launches (C <= 4) with at most this many quadrant-waves split every list of >= 256 entries in
 * four for the backward (the forward checkpoints every pixel's state at the quarter points of its tile's list; DESIGN.md §6.3, HISTORY.md §11).
 * 0 = never, < 0 or > 6144 = the built-in default 6144 (one 640x480 frame).  Must not change between a forward and its backward. */
int splatraster_debug_set_split_max_waves(int waves);
/* A/B / test hook: the forward of narrow layouts (C <= 4) walks the longest tile lists of a launch with a TEAM of four waves per
 * quadrant (two evaluate alpha for a step of candidates, one runs the transmittance chain a step behind, one accumulates
 * another step behind).  -1 (default): on the launches that are also split for the backward (at most 6144
 * quadrant lists: one 640x480 frame), 0: never, 1: every narrow launch of at most 32768 quadrant lists, 2 (tests): the same and a team for each of the 128 longest lists
 * whether or not they stand out.  Results never depend on it (bit-identical images, T, n_contrib, segment records). */
int splatraster_debug_set_fwd_team(int mode);
/* A/B / test hook: instance count from which the per-instance payload is written with streaming (non-temporal) stores
 * (HISTORY.md §11); < 0 restores the built-in default (8 Mi instances), 0 = always.  Results never depend on it. */
int splatraster_debug_set_payload_stream_min(int64_t instances);
/* A/B / test hook: which front end orders the tile instances.  -1 (default): the binned front end (counting sort by
 * (view, tile) + one LDS sort per tile) for windows of at most 6144 (view, tile) lists — SplatLoc's own
 * 640x480 frames, singly or five at a time — and the two global radix sorts otherwise; 0: the radix sorts always; 1: the
 * binned front end whenever the shape allows (at most 16 384 tiles per view).  Both produce bit-identical point lists,
 * ranges and payloads.  The render stage follows the choice the GEOMETRY stage made for its geometry buffer: a change between the
 * two stages of a forward takes effect at the next geometry stage. */
int splatraster_debug_set_front_end(int mode);
/* A/B / test hook of the binned front end: the per-tile sort launch exists for lists of up to 2048 keys (128 threads, every
 * tile of a 640x480 frame resident at once) and of up to 4096 (256 threads); by default the library follows a hint the kernel
 * raises when it meets a list beyond 2048 keys (longer lists than the chosen launch holds go to the long-list launch either
 * way).  2048 / 4096 force an instantiation, any other value restores the default.  Results never depend on it. */
int splatraster_debug_set_tile_sort_cap(int keys);
/* Measurement hook.  The binned front end's two sort launches (lists up to the tile launch's cap | longer lists, which that launch
 * finds in the scanned table itself) are independent: 0 (default) = both on the caller's stream; -1 = the long-list launch on an
 * internal side stream — forked behind the key scatter, joined before the compositing grid — when a list beyond 2048 keys was seen
 * in the last 64 frames; 1 = always.  Measured a wash at Replica scale (profiles/r06_ab_probes.txt #6).  Results never depend on it. */
int splatraster_debug_set_sort_fork(int mode);
int splatraster_debug_exp2(int64_t n, const float* x, float* y, void* stream);
/* test hook: fills the LDS of every compute unit with `pattern` (e.g. a NaN's bits): enough workgroups of 64 KB each to cover
 * the whole array.  The compositing kernels read rows of their LDS staging buffers that a round did not write (the absent second
 * member of a round's last pair); their results must not depend on what the LDS held before the launch. */
int splatraster_debug_poison_lds(uint32_t pattern, void* stream);

const char* splatraster_error_string(int status);
/* last HIP error text recorded by this thread (empty string when none). */
const char* splatraster_last_hip_error(void);
int splatraster_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SPLATRASTER_H */
