// common.h — shared declarations of the gfx950 rasterizer kernels (internal; the public
// contract is include/splatraster.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/splatraster.h"

namespace sr {

constexpr int TILE = SPLATRASTER_TILE;
constexpr int TILE_PIX = TILE * TILE;
constexpr float NEAR_Z = 0.2f;
constexpr float DILATION = 0.3f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_EPS = 0.0001f;
constexpr int WAVE = 64;
constexpr int PREPROCESS_BLOCK = 256;
static inline int preprocess_blocks(int32_t P) { return (P + PREPROCESS_BLOCK - 1) / PREPROCESS_BLOCK; }

// thread-local last-error text, filled by SR_HIP_CHECK
void set_hip_error(hipError_t e, const char* what);
void set_error_text(const char* text);

// look-back watchdog (scan_sort.hip): flag word in host-mapped memory, polled by every entry point
int lookback_error_init();
int lookback_error_poll();
int lookback_set_spin_limit(uint32_t limit);
// test hook: y[i] = exp2_shared(x[i]) (composite_fwd.hip)
int launch_debug_exp2(int64_t n, const float* x, float* y, hipStream_t stream);
int launch_poison_lds(uint32_t pattern, hipStream_t stream);

#define SR_HIP_CHECK(expr)                                 \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) {                            \
            ::sr::set_hip_error(_e, #expr);                \
            return SPLATRASTER_ERR_HIP;                    \
        }                                                  \
    } while (0)

#define SR_LAUNCH_CHECK()                                  \
    do {                                                   \
        hipError_t _e = hipGetLastError();                 \
        if (_e != hipSuccess) {                            \
            ::sr::set_hip_error(_e, "kernel launch");      \
            return SPLATRASTER_ERR_HIP;                    \
        }                                                  \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- a window of views ------------------------------------------------------------------
// SplatLoc.map renders `window_size` views of ONE Gaussian scene before a single backward
// (train_gaussians.py:195-229).  Every stage of the pipeline therefore works on a WINDOW of V views at once
// (V = 1: the plain GaussianRasterizer call): "row" g = v * P + i is Gaussian i as seen from view v, "global tile"
// v * tiles + t is tile t of view v.  One preprocess, ONE depth sort over the V * P rows, one emission, ONE
// tile sort keyed by the global tile, one compositing grid over V * tiles * 4 quadrant-waves; the per-view
// results are bit-identical to V separate calls (both sorts are stable, so every (view, tile) list is in
// (depth, index) order).  Per-view pointers travel as kernel arguments (the camera tensors stay on the device:
// reading them on the host would cost a synchronisation).
constexpr int MAX_VIEWS = SPLATRASTER_MAX_WINDOW_VIEWS;
struct WinCams {
    const float* view[MAX_VIEWS];    // [4,4] row-vector convention
    const float* proj[MAX_VIEWS];    // [4,4] view @ proj
    const float* campos[MAX_VIEWS];  // [3] or null
    float tanfovx[MAX_VIEWS], tanfovy[MAX_VIEWS];
    int32_t* radii[MAX_VIEWS];       // [P] per view (forward: written; backward: read)
};
struct WinOut {                      // forward outputs
    float* color[MAX_VIEWS];         // [C,H,W]
    float* depth[MAX_VIEWS];         // [1,H,W]
    float* alpha[MAX_VIEWS];         // [1,H,W]
};
struct WinGrad {                     // backward inputs per view
    const float* out_color[MAX_VIEWS];
    const float* out_depth[MAX_VIEWS];
    const float* dL_dcolor[MAX_VIEWS];
    const float* dL_ddepth[MAX_VIEWS];   // null = zeros
    const float* dL_dalpha[MAX_VIEWS];   // null = zeros
    const float* dL_dlast[MAX_VIEWS];    // gradient plane of channel C - 1 when it travels apart (null = zeros); see gc
    float* dL_dmeans2D[MAX_VIEWS];       // [P,3] output
    int gc;                              // channel planes behind dL_dcolor: C, or C - 1 (then channel C - 1 reads dL_dlast)
    const float* bg;                     // background: the walk starts at A = bg . g - g_A; may be null (zeros)
    int bg_channels;
};

// ---- opaque buffer views (n = V * P rows) -------------------------------------------------
struct GeomView {
    float4* rec;             // [2n] 32-byte records: rec[2g] = px, py, depth, radius (float; 0 = culled),
                             //                       rec[2g+1] = conic a, b, c, opacity
    uint32_t* tiles_touched; // [n] row order
    uint32_t* depth_order;   // [n] rows sorted by depth (all views together; culled rows last): row in bits 0..23,
                             //     min(tiles_touched[row], 255) in bits 24..31 (the scan then needs no gather)
    uint32_t* offsets;       // [n] inclusive scan of tiles_touched in depth order
    float* rgb;              // [3P]   (SH colours: single-view calls only)
    uint8_t* clamped;        // [3P]
    uint32_t* sort_keys;     // [n] scratch (depth bits)
    uint32_t* sort_tmp;      // scratch for the n-sized sort + scan partials
    uint32_t* total;         // [4] device-side R (uint64 as two words), written by the scan
    uint32_t* block_tiles;   // [V * preprocess_blocks(P)] per-(view, block) sums of tiles_touched, written by preprocess
    uint32_t* span_owner;    // [SPAN_OWNER_CAP] depth rank owning instance k * EMIT_SPAN, written by the one-pass scan
};
struct BinView {
    uint32_t* point_list; // [R] sorted rows (v * P + i)
    uint32_t* tile_list;  // [R] sorted global tile ids (v * tiles + t)
    uint32_t* ranges;     // [2 * V * tiles]
    uint32_t* keys_tmp;   // [R] unsorted global tile ids
    uint32_t* vals_tmp;   // [R] unsorted rows
    void* sort_tmp;       // radix sort scratch
    float4* irec;         // [2R] per-instance copy of the 32-byte record, in sorted order
    uint32_t* ipack;      // [R] point_list[j] (a row id < 2^24) | reach bits << 24 (bit q: the Gaussian may reach quadrant q of its tile):
                          //     ONE word per entry is all a compositing wave prefetches
    float* featp;         // [P * CP] feature rows padded to CP = roundup4(C) floats (16-byte aligned rows); shared by the views
    float* gacc;          // [V * P * gacc_row_floats(C)] backward gradient accumulator rows (one per row)
    float* pose_acc;      // right behind gacc (one fill zeroes both): POSE_SETS replicated accumulator sets of the camera gradients + the ticket word
    uint32_t* nparts;     // [V * tiles] parts of every (view, tile) list in a split launch (written with the launch order; common.h)
    float* ckpt;          // [V][SPLIT_PARTS_MAX][C + 2][H * W] segment records of the forward (T in front of the segment, its own colours, depth), split launches only
    uint32_t* tile_order; // [V * tiles] global tile ids, longest list first: the launch order of the compositing kernels
};
struct ImgView {
    float* final_T;       // [V][H * W]
    uint32_t* n_contrib;  // [V][H * W]
};

GeomView geom_view(void* base, int32_t P, int32_t V);
BinView bin_view(void* base, int32_t P, int32_t V, int64_t R, int32_t W, int32_t H, int32_t C);
ImgView img_view(void* base, int32_t W, int32_t H, int32_t V);

// ---- stage launchers (each returns a SPLATRASTER_* status) -----------------------------
// RAW-parameter mode of the forward projection (splatraster_forward_window_geometry_raw): the parameter activations of SplatLoc's own
// configuration run inside preprocess_kernel (activation_math.h: the arithmetic of activations.hip) — the kernel reads the raw tensors,
// uses the activated values in registers and writes them (and the packed colour rows) once for the later stages and the backward
struct RawFwd {
    const float* scaling;    // [P,3] log-scales
    const float* rotation;   // [P,4] raw quaternions
    const float* opacity;    // [P]   logits
    const float* f_dc;       // [P,3] SH dc coefficients
    const float* extra;      // [P,E] feature columns (null when E == 0)
    int E;
    float* scales;           // [P,3]   out: exp
    float* rotations;        // [P,4]   out: normalised
    float* opacities;        // [P]     out: sigmoid
    float* colors;           // [P,3+E] out: [clamp_min(C0 f_dc + 0.5, 0) | extra]
};
int launch_preprocess(const splatraster_settings& s, int32_t P, int32_t V, const WinCams& cams, const float* means3D,
                      const float* shs /*V == 1 only*/, const float* opacities, const float* scales, const float* rotations,
                      const float* cov3D_precomp, GeomView g, uint32_t* zero0, uint32_t nzero0, uint32_t* zero1, uint32_t nzero1,
                      hipStream_t stream, bool depth_keys = true /*false: no depth-sort input (binned front end)*/,
                      const RawFwd* raw = nullptr);
// sums the gradient contributions of the V views of a window into ONE set of parameter gradients (written once,
// in view order: deterministic given the accumulator rows); dL/dmeans2D is per view
// RAW-parameter mode of the preprocess backward (SplatLoc's own configuration: SH degree 0, scales + rotations): the chain through the
// parameter activations and the sum of the accumulator rows' colour columns happen in the same kernel — no gather_dcolors launch, no
// activation-backward launch, no activated-gradient tensors (splatraster_backward_window_raw)
struct RawBwd {
    const float* scaling;    // [P,3] log-scales
    const float* rotation;   // [P,4] raw quaternions
    const float* opacity;    // [P]   logits
    const float* f_dc;       // [P,3] SH dc coefficients
    int E;                   // feature columns behind the three colours (C = 3 + E)
    float* d_scaling;        // [P,3]
    float* d_rotation;       // [P,4]
    float* d_opacity;        // [P]
    float* d_f_dc;           // [P,3]
    float* d_extra;          // [P,E] (null when E == 0)
    // optional: the isotropic regulariser of SplatLoc.map (train_gaussians.py:221-228) adds reg_weight * reg_out[1] * reg_row_grad[i] to
    // dL/dscales[i, :] BEFORE the chain through exp (losses.hip: row_grad, out = {loss, 1 / |mask|})
    const float* reg_row_grad;   // [P] or null
    const float* reg_out;        // [2] device
    float reg_weight;
};
int launch_preprocess_bwd(const splatraster_settings& s, int32_t P, int32_t V, const WinCams& cams, const WinGrad& grads,
                          const float* means3D, const float* shs /*V == 1 only*/,
                          const float* scales, const float* rotations, const float* cov3D_precomp,
                          const uint8_t* clamped, const float4* rec, const float* gacc, int C,
                          float* dL_dcolors,
                          float* dL_dmeans3D, float* dL_dopacities, float* dL_dscales,
                          float* dL_drotations, float* dL_dcov3D, float* dL_dshs,
                          float* dL_dview /*[16] or null; V == 1 only*/, float* dL_dproj, float* dL_dcampos,
                          float* pose_acc /*BinView::pose_acc, zeroed by the caller; needed with dL_dview*/, hipStream_t stream,
                          const RawBwd* raw = nullptr);
int launch_mark_visible(int32_t P, const float* means3D, const float* view, uint8_t* present,
                        hipStream_t stream);

size_t sort_tmp_bytes(int64_t n);
size_t sort_zero_bytes(int64_t n, int key_bits);
int sort_pairs_u32(int64_t n, uint32_t* keys, uint32_t* vals, uint32_t* keys_alt, uint32_t* vals_alt,
                   int key_bits, void* tmp, hipStream_t stream, bool* result_in_alt, bool tmp_zeroed = false);
// inclusive scan of in[perm[i] & 0xFFFFFF] (perm may be null; its words carry min(in[row], 255) in bits 24..31:
// scan_sort.hip perm_value) into out[i]; total (u64 as 2 words) optional
size_t scan_tmp_bytes(int64_t n);
size_t scan_state_bytes(int64_t n);
int inclusive_scan_u32(int64_t n, const uint32_t* in, const uint32_t* perm, uint32_t* out, uint32_t* total,
                       void* tmp, hipStream_t stream, bool state_zeroed = false, uint32_t* span_owner = nullptr,
                       uint32_t span = 0, uint32_t span_cap = 0);
constexpr int EMIT_SPAN = 1024;  // instances emitted per workgroup (binning.hip)
constexpr uint32_t SPAN_OWNER_CAP = 1u << 16;  // emit workgroups that get their first rank from the scan (R <= 64 M)

// exclusive scan of in[0, n) in place (in == out); total (u64 as 2 words) optional; state words at tmp zeroed by the caller
// when state_zeroed
int exclusive_scan_u32(int64_t n, uint32_t* inout, uint32_t* total, void* tmp, hipStream_t stream, bool state_zeroed = false);

// ---- tile-binned front end (binsort.hip): counting sort by (view, tile) + one LDS sort per tile ----------------
constexpr int BIN_THREADS = 1024;           // threads of a count / scatter block
// rows of one view per count / scatter block: 2048 (2 per thread), 1024 when that would leave CUs without a block (one frame of a
// 200k-Gaussian map: 96 blocks of 2048 rows on 256 CUs), 8192 (8 per thread) when the frame has so many tiles that the
// (tile, chunk) table would outgrow the rows it describes
#ifndef SR_BIN_SMALL_BLOCKS
#define SR_BIN_SMALL_BLOCKS 128      // launches with fewer 2048-row blocks than this use 1024-row blocks (196k rows: walk + scan 39.5 -> 32.9 us; 500k: 29.8 -> 35.5)
#endif
static inline int bin_rows_per_thread(int32_t P, int32_t V, int tiles)
{
    if (tiles > 2048) return 8;
    const int64_t blocks2 = (int64_t)(V > 0 ? V : 1) * ((P + 2 * BIN_THREADS - 1) / (2 * BIN_THREADS));
    return blocks2 < SR_BIN_SMALL_BLOCKS ? 1 : 2;
}
static inline int bin_chunk_rows(int32_t P, int32_t V, int tiles) { return BIN_THREADS * bin_rows_per_thread(P, V, tiles); }
static inline int bin_chunks(int32_t P, int32_t V, int tiles) { return (P + bin_chunk_rows(P, V, tiles) - 1) / bin_chunk_rows(P, V, tiles); }
constexpr int BIN_MAX_TILES = 16384;        // tiles per view: the LDS histogram of a count / scatter block (64 KB)
#ifndef SR_BIN_AUTO_MAX_TILES
#define SR_BIN_AUTO_MAX_TILES 6144          // (view, tile) lists up to which the binned front end is the default
#endif
constexpr int BIN_AUTO_MAX_TILES = SR_BIN_AUTO_MAX_TILES;
constexpr int BIN_SORT_TILE_NARROW = 2048, BIN_SORT_TILE_WIDE = 4096;   // longest list of the per-tile launch's two instantiations
// (a 1024-key instantiation — one wave, 8.7 KB of LDS — was measured at 1080p, where every list fits: tile-sort stage 0.797 -> 0.785 ms per window
//  of five views; the stage is the payload's gathers and stores there, not residency; profiles/r06_ab_probes.txt #9)
constexpr int BIN_WIDE_FRAMES = 64;         // frames the wide instantiation stays selected after a list beyond 2048 was seen
void set_bin_fork(int mode);                // -1: the long-list sort launch runs on a side stream when the scene has long lists; 0 never; 1 always
void set_bin_tile_cap(int cap);             // test / A-B hook: 2048 or 4096 forces an instantiation, anything else: follow the hint
constexpr int BIN_SORT_BIG = 16384, BIN_BIG_BLOCKS = 128; // the launch for longer lists (keys in 139 KB of LDS, 1024 threads; blocks)
constexpr int BIN_BIG_MINE = 1024;          // long lists one block of that launch can be handed (list k is block k % BIN_BIG_BLOCKS's)
static_assert((size_t)BIN_MAX_TILES * MAX_VIEWS <= (size_t)BIN_BIG_MINE * BIN_BIG_BLOCKS, "every (view, tile) list could be a long one");
void set_bin_mode(int mode);   // -1 auto, 0 radix front end always, 1 binned whenever the shape allows
size_t bin_table_entries(int32_t P, int32_t V, int tiles);
size_t bin_scratch_bytes(int32_t P, int32_t V, int tiles);
bool use_bins(int32_t P, int32_t V, int gx, int gy /*tiles per row / column of a view*/, size_t scratch_bytes);
int launch_bin_count(const splatraster_settings& s, int32_t P, int32_t V, const GeomView& g, uint32_t* table, void* scan_tmp,
                     hipStream_t stream);
int launch_bin_scatter_sort(const splatraster_settings& s, int32_t P, int32_t V, int64_t R, const GeomView& g,
                            const uint32_t* table, const BinView& b, uint64_t* keys, hipStream_t stream);

int launch_emit(const splatraster_settings& s, int32_t P, int32_t V, int64_t R, const GeomView& g, uint32_t* keys,
                uint32_t* vals, uint32_t* ranges, uint32_t nranges, hipStream_t stream);
int launch_ranges_clear(int32_t tiles, uint32_t* ranges, hipStream_t stream);
int launch_payload(const splatraster_settings& s, int32_t V, int64_t R, const GeomView& g, const BinView& b, hipStream_t stream);
// 16-byte aligned copy of the [P, C] feature rows (returns feat itself when C % 4 == 0)
int launch_pad_features(int32_t P, int C, const float* feat, float* featp, hipStream_t stream);
static inline int padded_channels(int C) { return (C + 3) & ~3; }
// Per-Gaussian gradient accumulator row of the backward (64-byte aligned rows so that the
// float atomics of one Gaussian land in as few memory-side requests as possible):
//   [0, C) dL/dfeature, [MO, MO + 7) moment record, padded to a multiple of 16 floats.
// The moments share the last 64-byte line of the features when they fit.
// camera-gradient accumulators of preprocess_bwd_kernel<true>: block b adds its 27 partial sums to set b % POSE_SETS (sixteen sets on
// sixteen 256-byte lines instead of 27 words every block hits), the LAST block to finish (ticket) sums the sets and writes the outputs
constexpr int POSE_SETS = 16, POSE_SET_FLOATS = 64;
constexpr size_t POSE_ACC_BYTES = (size_t)POSE_SETS * POSE_SET_FLOATS * sizeof(float) + 256;   // + the ticket's own line
__host__ __device__ static inline int gacc_moment_offset(int C) { return ((C & 15) + 7 <= 16) ? C : ((C + 15) & ~15); }
__host__ __device__ static inline int gacc_row_floats(int C) { return (gacc_moment_offset(C) + 7 + 15) & ~15; }
// One small frame alone (SplatLoc's color_refinement: 640x480, one view) is 4 800 quadrant-waves on a machine with room
// for ~8 000: every wave starts at once and the compositing kernel lasts as long as its LONGEST list.  For such launches
// (narrow layouts, at most SPLIT_MAX_WAVES quadrant-waves: measured 233 -> 180 us at 4 800 waves, 201 -> 187 us at 12 900,
// 810 -> 790 us at 24 000, and a LOSS at 64 500: 783 -> 871 us) the forward records, per quarter of the tile's list, every pixel's
// transmittance in front of it and the colours / depth that quarter ALONE contributes, and the backward runs SPLIT_PARTS waves
// per quadrant, one per quarter.  The backward walks back to front (round 4): a wave whose quarter ends in front of a pixel's last
// contributor starts from the boundary state T_b, A_b = (sum of the LATER quarters' colours . g + depth g_D + T_final (bg . g - g_A)) / T_b
// — segment sums accumulated from zero, accurate relative to their own magnitude (rounds 1-3 subtracted a prefix from the image:
// an absolute error of 1e-7 |image| in a remainder of size T_b |image|).  The backward only needs the list up to the quadrant's
// deepest contributor (63 % of it on average), so quarters balance better than halves.  The forward itself — T chain,
// n_contrib, final_T, the images — is untouched, so every bit-exact contract holds.  Lists shorter than SPLIT_MIN_LIST
// are not split.
// Round 4: the threshold is 6 144 quadrant-waves (one 640x480 frame is 4 800), not 26 000: the segment sums cost the forward
// 8 % and the back-to-front backward no longer gains from splitting a 24 000-wave window (5 views at 640x480: backward 0.777 ms
// split vs 0.725 ms whole, forward 0.418 vs 0.393 ms), while one frame alone still does (backward 132 -> 102 us).
#ifndef SR_SPLIT_MAX_WAVES
#define SR_SPLIT_MAX_WAVES 6144
#endif
constexpr int SPLIT_MAX_WAVES = SR_SPLIT_MAX_WAVES;
constexpr int SPLIT_MIN_LIST = 256;
#ifndef SR_SPLIT_PARTS
#define SR_SPLIT_PARTS 4
#endif
constexpr int SPLIT_PARTS = SR_SPLIT_PARTS;     // parts of every split list = gridDim.y of the split backward
// Round 6: the LONGEST lists of such a frame get more parts.  On a reconstructed room the lists are not the uniform cloud's: at
// Replica scale (413k rows, 640x480) the mean list has 1 400 entries and 15 - 40 lists per frame exceed 4 096 (up to 7 400) — a
// quarter of one is still 1 900 entries for one wave, and the backward (241 us) waited for those waves: 8 parts for EVERY list
// 241 -> 217 us there but 102 -> 109 us on the uniform cloud, 16 parts 227 / 164 (profiles/r06_ab_probes.txt #5).  So: the first
// SPLIT_EXTRA_TILES lists of the launch order (longest first) are split in 8 parts from SPLIT_LONG entries and in 16 from twice that
// (128 lists from 2 048 entries: 214 us at Replica scale / 134 on the 60-key-frame room / 102 on the uniform cloud; 256 from 1 536:
// 201 / 133 / 103; 512 from 1 024: 195 / 128 / 105 — SplatLoc's maps are rooms: the last one);
// their parts beyond the fourth are walked by EXTRA workgroups at the head of the backward's grid.  How many parts a list has is
// decided once, by the block that computes the launch order (tile_order.h: `nparts[global tile]`), and read by the forward (which
// writes that many segment records) and the backward alike.
constexpr int SPLIT_PARTS_MAX = 4 * SPLIT_PARTS;       // segment records per pixel the checkpoint buffer holds
#ifndef SR_SPLIT_LONG
#define SR_SPLIT_LONG 1024
#endif
#ifndef SR_SPLIT_EXTRA_TILES
#define SR_SPLIT_EXTRA_TILES 512
#endif
constexpr int SPLIT_LONG = SR_SPLIT_LONG, SPLIT_EXTRA_TILES = SR_SPLIT_EXTRA_TILES;
static_assert(SPLIT_EXTRA_TILES % 8 == 0, "the extra workgroups use the quadrant id scheme (8 tiles per 32 ids)");
void set_split_max_waves(int waves);   // A/B hook (< 0: default)
void set_fwd_team(int mode);           // A/B hook: teams of four waves for the longest lists of a narrow launch (-1 automatic, 0 never, 1 whenever possible)
void set_payload_stream_min(int64_t instances);   // test hook (< 0: default)
int split_max_waves();
static inline bool split_lists(int C, int V, int tiles) { return C <= 4 && 4 * V * tiles <= split_max_waves(); }
// parts of a list of `len` entries at position `rank` of the launch order
__host__ __device__ static inline uint32_t split_count(uint32_t len, uint32_t rank)
{
    if (rank >= (uint32_t)SPLIT_EXTRA_TILES || len < (uint32_t)SPLIT_LONG) return (uint32_t)SPLIT_PARTS;
    return len < 2u * (uint32_t)SPLIT_LONG ? 2u * (uint32_t)SPLIT_PARTS : (uint32_t)SPLIT_PARTS_MAX;
}
// entries per part of a list of `len` entries split in `np` parts (a multiple of the 64-entry chunk; len when the list is not split)
__host__ __device__ static inline uint32_t split_part(uint32_t len, uint32_t np = (uint32_t)SPLIT_PARTS)
{
    return len < (uint32_t)SPLIT_MIN_LIST ? len : ((len / np + 63u) & ~63u);
}

int launch_composite_fwd(const splatraster_settings& s, int32_t P, int32_t V, int64_t R, const GeomView& g, const BinView& b,
                         const ImgView& im, const float* featp /*padded rows, shared by the views*/, const float* bg,
                         const WinOut& out, hipStream_t stream);
// launch order of the compositing grids: global tile ids by descending list length (binning.hip) — for launches of a few
// rounds of waves only (a window of 640x480 frames: fwd 0.409 -> 0.397, bwd 0.758 -> 0.744 ms); on large grids the order costs
// more L2 locality between neighbouring tiles than the shorter tail gains (S2: fwd 1.93 -> 1.97, bwd 4.24 -> 4.31 ms)
#ifndef SR_TILE_ORDER
#define SR_TILE_ORDER 1   // 0 = compositing grids always in tile order (A/B)
#endif
constexpr int TILE_ORDER_MAX_WAVES = 32768;
static inline bool use_tile_order(int V, int tiles_per_view) { return SR_TILE_ORDER && 4ll * V * tiles_per_view <= TILE_ORDER_MAX_WAVES; }
int launch_tile_order(const splatraster_settings& s, int32_t V, const BinView& b, hipStream_t stream);
int launch_composite_bwd(const splatraster_settings& s, int32_t P, int32_t V, int64_t R, const GeomView& g,
                         const BinView& b, const ImgView& im, const float* feat, int feat_stride,
                         const WinGrad& grads,
                         float* gacc /*[V * P, gacc_row_floats(C)]: dL/dfeature | moments sum E dx, E dy, E dx^2,
                                      E dx dy, E dy^2, E, w g_D*/,
                         long long* gacc64 /*non-NULL: deterministic fixed-point accumulation into this buffer*/,
                         int det_pass /*deterministic mode: 0 = max pass, 1 = sum pass; else ignored*/,
                         hipStream_t stream);
// test / A-B hook: frames with at most this many quadrant-waves take the small-layout panel variant of the backward
// (< 0: the built-in default)
void set_small_panel_max_waves(int waves);
int launch_fixed_to_float(int64_t n, const long long* src, float* dst, hipStream_t stream);

int launch_activate_fwd(int32_t P, int32_t K, int32_t deg, int32_t SC, int32_t E, const float* xyz,
                        const float* f_dc, const float* f_rest, const float* scaling, const float* rotation,
                        const float* opacity, const float* extra, const float* campos, float* scales,
                        float* rotations, float* opacities, float* colors, hipStream_t stream);
int launch_activate_bwd(int32_t P, int32_t K, int32_t deg, int32_t SC, int32_t E, const float* xyz,
                        const float* f_dc, const float* f_rest, const float* scaling, const float* rotation,
                        const float* opacity, const float* campos, const float* g_scales,
                        const float* g_rotations, const float* g_opacities, const float* g_colors, float* d_xyz,
                        float* d_f_dc, float* d_f_rest, float* d_scaling, float* d_rotation, float* d_opacity,
                        float* d_extra, hipStream_t stream);

struct StatsViews {
    const float* vs_grad[MAX_VIEWS];   // [P,3] per view (unused when accum == null)
    const int32_t* radii[MAX_VIEWS];   // [P] per view
};
int launch_densification_stats(int32_t P, int32_t V, const StatsViews& views, float* accum /*or null*/,
                               float* denom /*or null*/, float* max_radii, hipStream_t stream);
size_t mapping_loss_workspace_bytes(int32_t HW);
int launch_mapping_loss_window(int32_t V, int32_t HW, const splatraster_loss_view* views, float threshold, float* out,
                               void* workspace, hipStream_t stream);
int launch_mapping_loss(int32_t HW, const float* image, const float* depth, const float* marker,
                        const float* gt_image, const float* gt_depth, const float* kp, float threshold,
                        const float* exposure, float* g_image, float* g_depth, float* g_marker, float* out,
                        void* workspace, hipStream_t stream);

size_t refinement_loss_workspace_bytes(int32_t C, int32_t H, int32_t W);
int launch_refinement_loss(int32_t C, int32_t H, int32_t W, float lambda, const float* image, const float* gt,
                           float* g_image, float* out, void* workspace, hipStream_t stream);

size_t eval_metrics_workspace_bytes(int32_t C, int32_t H, int32_t W);
int launch_eval_metrics(int32_t C, int32_t H, int32_t W, const float* image, const float* gt, float* out, void* workspace,
                        hipStream_t stream);

int launch_l1_rgbd_loss(int64_t n_color, const float* color, const float* tgt_c, int64_t n_depth, const float* depth,
                        const float* tgt_d, float depth_weight, float* g_color, float* g_depth, float* loss_out, hipStream_t stream);
int launch_pose_step(const float* dL_dview, const float* dL_dproj, const float* dL_dcampos, const float* W2C0, const float* Pm,
                     float lr_rot, float lr_trans, float beta1, float beta2, float eps, int advance, float* state, float* view_out,
                     float* proj_out, float* campos_out, hipStream_t stream);

int knn_dist2(int32_t N, const float* points, float* out, void* workspace, hipStream_t stream);
size_t knn_workspace_bytes(int32_t N);
void knn_set_grid_min(int n);   // point count from which the exact grid search replaces the tiled brute force (< 0: default)

}  // namespace sr
