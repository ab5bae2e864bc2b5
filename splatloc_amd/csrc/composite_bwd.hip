// composite_bwd.hip — backward of the alpha compositing (SURVEY.md §8a "COMPOSITE bwd").
//
// Replaces the render-backward stage of the rasterizer extension whose backward is
// triggered at train_gaussians.py:229 / :286.
//
// Formulation (round 4: BACK-TO-FRONT, the lineage's direction; rounds 1-3 walked front to back): with
// q_i = f_i . g + z_i g_D (g = dL/dcolor at the pixel), T_i the transmittance in front of Gaussian i and
//     A_i = [ sum_{j>i} alpha_j T_j q_j + T_final (bg . g - g_A) ] / T_{i+1}      (what lies behind i, per unit of light reaching it)
// the derivative is  dL/dalpha_i = T_i (q_i - A_i)  and the state moves towards the camera by
//     T_i = T_{i+1} / (1 - alpha_i),     A_{i-1} = A_i + alpha_i (q_i - A_i),
// started at every pixel's last contributor with T = final_T and A = bg . g - g_A (exact).  One dot product per (pixel, Gaussian),
// linear in (g, g_D, g_A) — channels can be split over several launches (generic C).  Every quantity is O(1)-scaled: errors stay
// RELATIVE to the Gaussian's own gradient.  The front-to-back form of rounds 1-3, S_i = S_total - sum_{j<=i} w_j q_j with S_total =
// out_color . g, carried an ABSOLUTE error of ~1e-7 |S_total| in float32 that was up to 1e-3 of the S_i of a Gaussian behind
// T = 1e-4 (profiles/r04_grad_bars_before_accurate_mode.json); it also needed the forward's C + 1 colour / depth planes, which
// this form does not read (except at the boundaries of split launches, below).  alpha itself is evaluated with the forward's
// arithmetic (composite_common.h), and which entries contribute is the forward's decision (n_contrib), so the two passes agree
// on every alpha >= 1/255 / T < 1e-4 test.
//
// Machine mapping = the forward's (composite_fwd.hip): ONE wave64 = one workgroup = one 8x8
// quadrant, the tile's list streamed from the per-instance payload (mask byte + record),
// candidates' feature rows gathered into LDS, candidates processed two at a time; no
// __syncthreads.  Per (wave, Gaussian) the per-pixel partials are summed over the 64 pixels:
//   * 7 geometric partials (+ channels beyond the first 32) of BOTH Gaussians of a pair:
//     one packed butterfly reduction (permlane32/16 swap + DPP, ~2.2 VALU per value) and one
//     wave-wide float atomic;
//   * NC >= 32: dL/dfeature[g][ch] = sum_pix w[pix][g] * dL/dcolor[pix][ch] for the first 32
//     channels is a dense [16 g x 64 pix] x [64 pix x 32 ch] contraction per group of 16
//     contributing Gaussians: the weights are parked in a 4-KB LDS panel (one ds_write per
//     step), the dL/dcolor panel lives in registers for the whole tile, and the contraction
//     runs on the matrix pipe with v_mfma_f32_16x16x4_f32 — exact fp32 (k-ordered fmaf chain).
#include <type_traits>

#include "composite_common.h"

#ifndef SR_BWD_FS
#define SR_BWD_FS 32  // feature rows staged per round (<= 64)
#endif
#ifndef SR_BWD_STAGE_UNROLL
#define SR_BWD_STAGE_UNROLL 3  // gather iterations in flight together while staging feature rows (A/B on S2, 5 cameras: 2: 0.947, 3: 0.914 ms)
#endif

#ifndef SR_BWD_MINW
#define SR_BWD_MINW 4  // waves per SIMD the register allocator must allow (A/B on S2: 1.035 vs 1.07 ms at 3)
#endif

#ifndef SR_BWD_DOT_CHAINS
#define SR_BWD_DOT_CHAINS 2  // (>= 1) independent accumulators of the 4x4x1 MFMA dot product (A/B: 1: 0.950, 2: 0.928, 3: 0.944 (other box), 6 spills)
#endif
#ifndef SR_BWD_SMALL_PANEL_MAX_WAVES
#define SR_BWD_SMALL_PANEL_MAX_WAVES 6144  // frames with at most this many quadrant-waves use the small-layout panel variant (0 = never)
#endif
#ifndef SR_BWD_DOT_PREFETCH
#define SR_BWD_DOT_PREFETCH 1  // double-buffered LDS reads in the 4x4x1 dot (A/B on S2: 0.8645 -> 0.860 ms)
#endif
// (The timing probes of rounds 2-3 that produce WRONG results by design — hot-row gathers, skipped staging, dropped atomics /
//  butterfly / MFMAs / plane loads — are not part of this translation unit: tools/patches/composite_probes.patch adds them
//  back for tools/ablate.py --patch.)
#ifndef SR_BWD_TM
#define SR_BWD_TM 1   // 1 = transposed moment reduction (small-layout panel variant): E = G dL/dalpha parked in an 8-column LDS panel, reduced by lane = (Gaussian, pixel column)
#endif
#ifndef SR_BWD_MINW_TM
#define SR_BWD_MINW_TM 3  // waves per SIMD of the NC >= 32 kernel WITH the E panel (12.6 KB of LDS per wave: 12 workgroups per CU)
#endif
#ifndef SR_BWD_DOTM_MIN
#define SR_BWD_DOTM_MIN 8  // channels from which the 4x4x1 MFMA dot product is used
#endif

namespace sr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NC, bool SP, bool AUX = true>
struct BwdCfg {
    // Small layouts (NC <= 15: SplatLoc's own C = 4, and 1 / 2 / 3 / 8), variant SP: ALL channels and the depth
    // weight w g_D are ONE 16-column block of the flush contraction, so only the six geometric moments go
    // through the butterfly (12 instead of 22 values per pair at C = 4) — at the price of 91 instead of 59 VGPRs
    // (5 instead of 8 waves per SIMD).  It is chosen per launch: when the frame's quadrant-waves all fit the
    // machine at once anyway (SplatLoc's 640x480: 4 800 waves) the lost occupancy costs nothing and the kernel
    // is 8 % faster (C = 4: 0.252 -> 0.231 ms); on large frames the butterfly variant at full occupancy wins
    // (S1, 1200x680: 0.198 vs 0.238 ms), and below 4 channels the flush costs what it saves.
    static constexpr bool SMALLP = NC <= 15 && SP;
    static constexpr bool MFMA = NC >= 32 || SMALLP;
    // NC in (32, 47]: the channels beyond 32 and the depth weight are a THIRD 16-column block of the flush (16 more
    // MFMAs per 16 Gaussians) instead of 4 of the 10 butterfly values per Gaussian
    static constexpr int NM = NC >= 32 ? 32 : (SMALLP ? NC : 0);   // channels reduced on the matrix pipe
    static constexpr int NB = NC >= 32 ? 2 : 1;         // 16-column blocks of the contraction
    static constexpr bool XD = SMALLP && AUX;              // column NC of the blocks = the depth weight
    static constexpr int NV = NC - NM;         // channels reduced with the packed butterfly
    // TM (small-layout panel variant; measured on the reference layout, 5 views: 0.791 -> 0.762 ms): the six geometric
    // moments leave the butterfly too.  E = G dL/dalpha is parked in an
    // 8-column LDS panel [64 pix][8 Gaussians]; every 8 Gaussians lane (g = l & 7, i = l >> 3) sums its Gaussian over
    // pixel column i (8 reads), forms the moments with ITS Gaussian's mean, and three swap stages fold the 8 columns:
    // ~8 VALU per Gaussian instead of 6 moment products + 13 butterfly instructions.
    static constexpr bool TM = SR_BWD_TM && SMALLP;
    static constexpr int EG = 8;               // Gaussians per E-panel reduction
    static constexpr int ES = 9;               // LDS row stride of the E panel [64 pix][EG]
    // AUX = false: no view of the launch has a depth / alpha gradient (color_refinement, train_gaussians.py:283-285):
    // the depth weight w g_D is identically zero and leaves the reduction
    static constexpr int KV = NV + ((XD || !AUX) ? 0 : 1) + (TM ? 0 : 6);   // butterfly values per Gaussian
    static constexpr int NCP = (NC + 3) & ~3;
    static constexpr int FS = SR_BWD_FS;             // feature rows staged per round
    static constexpr int GROUP = 16;           // Gaussians per MFMA flush (M of v_mfma_f32_16x16x4_f32)
    static constexpr int WS = 17;              // LDS row stride of the weight panel [64 pix][GROUP]
};

// Gradient accumulation.  Normal mode: float atomics (memory-side adds; the order in which the quadrant-waves
// of different tiles reach a Gaussian's row varies from run to run, so sums differ in the last bits).
// DET (splatraster_debug_set_deterministic — the reproducible AND accurate debug mode): every wave-level partial — itself
// computed in a fixed order — is converted to fixed point and added with a 64-bit INTEGER atomic: integer addition is
// associative, so the totals are bit-reproducible whatever the arrival order.  Round 4: the fixed point is chosen PER
// ELEMENT from the largest partial that element receives (round 3 used 2^-40 for everything: a resolution of 9e-13,
// which is 1e-3 of a gradient of 1e-9 — with mean-reduced losses, dL/dout ~ 1 / (H W), most rows of a 500k-Gaussian
// scene are that small).  The kernel therefore runs TWICE in this mode:
//   det_pass 0: atomicMax of the bit pattern of |partial| into the (zeroed) float accumulator — max is associative too;
//   det_pass 1: partial * 2^(170 - biased exponent of that max) added as int64: |partial| * scale < 2^44, and up to 2^18
//               partials per element (4 quadrant-waves per tile a Gaussian touches) cannot overflow 2^62; the resolution
//               is 2^-44 of the element's LARGEST partial — twenty bits below a float's;
//   fixed_to_float_kernel (preprocess_bwd.hip) reads the same exponent and converts back.
constexpr int DET_HEADROOM_EXP = 170;   // scale exponent = DET_HEADROOM_EXP - biased exponent of max |partial|
__device__ __forceinline__ int det_scale_exp(unsigned max_bits)
{
    const int eb = (int)((max_bits >> 23) & 0xffu);
    return DET_HEADROOM_EXP - (eb > 0 ? eb : 1);
}
template <bool DET>
__device__ __forceinline__ void acc_add(float* gacc, long long* gacc64, size_t idx, float v, int det_pass)
{
    if (DET) {
        if (det_pass == 0) {
            atomicMax(reinterpret_cast<unsigned*>(gacc) + idx, __float_as_uint(fabsf(v)));
        } else {
            const unsigned mb = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(gacc) + idx);
            if (((mb >> 23) & 0xffu) == 0xffu) {
                // a NaN or an infinity reached this element (it won pass 0's maximum): no fixed point can hold it.  The element
                // must come back non-finite, like the float atomics of the normal path would leave it — this mode exists to
                // expose such failures, not to flush them to 0 (round-4 advisor finding).  NaN: the recorded maximum says it
                // all; infinities: their signs are counted (low word +inf, high word -inf) for fixed_to_float_kernel
                if (__builtin_isinf(v)) atomicAdd(reinterpret_cast<unsigned long long*>(gacc64) + idx, v > 0.f ? 1ull : (1ull << 32));
                return;
            }
            const double scaled = ldexp((double)v, det_scale_exp(mb));
            atomicAdd(reinterpret_cast<unsigned long long*>(gacc64) + idx, (unsigned long long)__double2ll_rn(scaled));
        }
    } else {
        (void)det_pass;
        atomicAdd(gacc + idx, v);
    }
}

template <int NC, bool SP, bool AUX>
constexpr int bwd_min_waves() { return (BwdCfg<NC, SP, AUX>::TM && NC >= 32) ? SR_BWD_MINW_TM : SR_BWD_MINW; }

template <int NC, bool DET, bool SP, bool AUX>
__global__ void __launch_bounds__(WAVE, (bwd_min_waves<NC, SP, AUX>()))
composite_bwd_kernel(int W, int H, int C_total, int CP4, int c0, int first_pass, int tiles /*per view*/, int V,
                     int P /*rows per view*/,
                     const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ ipack /*id | reach bits << 24*/,
                     const float4* __restrict__ irec, const float4* __restrict__ featp4,
                     WinGrad grads,
                     const float* __restrict__ final_T_all, const uint32_t* __restrict__ n_contrib_all,
                     float* __restrict__ gacc /*[V * P, GROW]*/, int GROW,
                     int MO, long long* __restrict__ gacc64 /*[V * P, GROW] fixed point, DET only*/,
                     const float* __restrict__ ckpt_all /*split launches (common.h; gridDim.y == SPLIT_PARTS): the forward's segment
                                                          records [V][SPLIT_PARTS_MAX][NC + 2][H * W], else null*/,
                     const uint32_t* __restrict__ nparts /*split launches: parts of every (view, tile) list, or null: SPLIT_PARTS*/,
                     int extra_blocks /*split launches: workgroups at the head of the grid that walk parts 4 .. of the longest lists*/,
                     const uint32_t* __restrict__ tile_order /*launch order (binning.hip), or null*/,
                     int det_pass /*DET only: 0 = per-element max of |partial|, 1 = fixed-point sums (acc_add)*/)
{
    using Cfg = BwdCfg<NC, SP, AUX>;
    // DET (the accurate mode): the walk's state T, A and everything derived from it in double; else float
    using ST = typename std::conditional<DET, double, float>::type;
    constexpr int NCP = Cfg::NCP, NM = Cfg::NM, NV = Cfg::NV, KV = Cfg::KV, FS = Cfg::FS;
    constexpr int WS = Cfg::WS, GROUP = Cfg::GROUP;
    constexpr bool MFMA = Cfg::MFMA, XD = Cfg::XD;
    constexpr int NB = Cfg::NB;
    constexpr bool TM = Cfg::TM;
    constexpr int EG = Cfg::EG, ES = Cfg::ES;
    constexpr bool DOTM = NC >= SR_BWD_DOTM_MIN;  // dot products q = f . g on the matrix pipe
    constexpr int PPR = NCP / 4;  // 16-byte pieces per staged row
    static_assert(2 * KV <= WAVE, "at most 25 butterfly-reduced channels per pass");
    // records of the staged candidates only (row-indexed, like s_feat): with the 64-entry chunk
    // records here the workgroup needs 11.2 KB and only 14 fit a CU; at 10 KB all 16 (4 / SIMD) do
    // (+ 1: the second Gaussian of a pair is ALWAYS read from row r0 + 1 — adjacent rows share one address register —
    //  also when the round's last pair has no second member: that row is then stale, and the pair's second half is masked)
    __shared__ __attribute__((aligned(16))) float4 s_rec0[FS + 1];
    __shared__ __attribute__((aligned(16))) float4 s_rec1[FS + 1];
    __shared__ __attribute__((aligned(16))) float s_feat[(FS + (DOTM ? 0 : 1)) * NCP];   // (the MFMA dot product clamps its rows)
    __shared__ uint32_t s_cgid[FS + 1];
    __shared__ float s_w[MFMA ? WAVE * WS : 1];   // matrix-pipe weight panel w[64 pix][GROUP]
    __shared__ uint32_t s_gid[MFMA ? GROUP : 1];  // Gaussian id of every parked panel column
    __shared__ float s_e[TM ? WAVE * ES : 1];     // E panel [64 pix][EG]: column c belongs to weight-panel slot e0 + c
    __shared__ float2 s_emean[TM ? GROUP : 1];    // projected mean of every parked Gaussian

    int gtile, quad;   // global tile = view * tiles + tile: the grid covers the V views of the window
    const int gx = (W + TILE - 1) / TILE;
    int seg_y = (int)blockIdx.y;      // split launches: the part of the list this wave walks
    if (NC <= 4 && (int)blockIdx.x < extra_blocks) {
        // EXTRA workgroups (first in the grid: their lists are the launch's longest): group e / (ids of SPLIT_EXTRA_TILES tiles) walks
        // parts 4 (group + 1) + blockIdx.y of the list at position r of the launch order — if that list has so many parts
        const unsigned per = quadrant_blocks(SPLIT_EXTRA_TILES, gx);
        const unsigned group = blockIdx.x / per;
        int r;
        quadrant_of_block(blockIdx.x - group * per, SPLIT_EXTRA_TILES, gx, r, quad);
        if (r >= SPLIT_EXTRA_TILES || r >= V * tiles) return;
        gtile = (int)tile_order[r];
        seg_y += SPLIT_PARTS * (int)(group + 1u);
        if (seg_y >= (int)nparts[gtile]) return;
    } else {
        quadrant_of_block(blockIdx.x - (NC <= 4 ? (unsigned)extra_blocks : 0u), V * tiles, gx, gtile, quad, tile_order);
        if (gtile >= V * tiles) return;
    }
    const int view = (V == 1) ? 0 : gtile / tiles;      // wave-uniform (scalar)
    const int tile = gtile - view * tiles;
    const uint32_t row0 = (uint32_t)view * (uint32_t)P;  // accumulator rows are per (view, Gaussian), feature rows shared
    const float* __restrict__ dL_dcolor = grads.dL_dcolor[view];
    const float* __restrict__ dL_ddepth = grads.dL_ddepth[view];
    const float* __restrict__ dL_dalpha = grads.dL_dalpha[view];
    const float* __restrict__ dL_dlast = grads.dL_dlast[view];
    const int gc = grads.gc;   // channel planes behind dL_dcolor; channel C - 1 reads dL_dlast when gc < C; channels between: no gradient
    const float* __restrict__ final_T = final_T_all + (size_t)view * H * W;
    const uint32_t* __restrict__ n_contrib = n_contrib_all + (size_t)view * H * W;
    const int lane = threadIdx.x;
    const int qx = (tile % gx) * TILE + (quad & 1) * 8, qy = (tile / gx) * TILE + (quad >> 1) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
    const size_t plane = (size_t)H * W;
    const size_t pix = inside ? (size_t)py * W + px : 0;
    uint32_t beg = ranges[2 * gtile], end0 = ranges[2 * gtile + 1];
    // split launches: wave blockIdx.y of the quadrant takes part blockIdx.y of the tile's list
    const uint32_t list0 = beg;                                // list positions (n_contrib) count from the tile's first entry
    const bool split = !DET && NC <= 4 && ckpt_all != nullptr;   // (the accurate mode starts every list at its head)
    const int seg = split ? seg_y : 0;
    const int np = (split && nparts != nullptr) ? (int)nparts[gtile] : SPLIT_PARTS;   // parts of this list (wave-uniform)
    if (split) {
        const uint32_t part = split_part(end0 - beg, (uint32_t)np);
        beg += (uint32_t)seg * part;
        if (seg < np - 1) end0 = min(end0, beg + part);
        if (beg >= end0) return;
    }
    // (the forward wrote C_total + 2 planes per checkpoint k, taken in front of list entry beg + (k + 1) part: T, its C_total
    //  colours so far, the depth so far — this pass may cover fewer channels)
    // per-pixel constants
    float g[NC];
    // state of the back-to-front walk: T = transmittance BEHIND the entry about to be processed, A = what lies behind it per
    // unit of light (header).  At a pixel's last contributor: T = final_T, A = bg . g - g_A.
    ST A = 0;
    ST T = 1;
    float gD = 0.0f;
    uint32_t last = 0;
    if (inside) {
        last = n_contrib[pix];
        // a split launch's segment that ends in FRONT of this pixel's last contributor starts from the boundary state rebuilt
        // from the forward's segment records (common.h): T_b, and A_b = (what the later segments contribute) / T_b
        const bool from_ckpt = split && seg < np - 1 && list0 + last > end0;
        float s_end = 0.0f;
        // planes [0, gc) behind dL_dcolor, the LAST channel's plane behind dL_dlast, nothing in between.  Branch-free on purpose:
        // every channel loads from a valid address chosen by selects (plane 0 when the channel has no gradient) and the value
        // is selected afterwards — with a branch per channel the set-up is 35 basic blocks and the compiler waits for every
        // second load before issuing the next (seen in the .s: 21 x s_waitcnt vmcnt(0) between the plane loads, backward + 4 %).
        float gvals[NC];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            const int c = c0 + ch;
            const bool lastc = c == C_total - 1 && c >= gc && dL_dlast != nullptr;
            const float* src = lastc ? dL_dlast : dL_dcolor + (size_t)(c < gc ? c : 0) * plane;
            gvals[ch] = src[pix];
        }
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            const int c = c0 + ch;
            const bool take = c < gc || (c == C_total - 1 && dL_dlast != nullptr);
            g[ch] = take ? gvals[ch] : 0.0f;
        }
        {   // s_end = bg . g: AFTER the plane loads and branch-free (clamped index + select) — a conditional scalar load inside the
            // loop above put a branch between the plane loads and serialised them (measured: backward + 6 % on S2)
            const float* bgp = grads.bg ? grads.bg : final_T_all;      // (any readable address; nb = 0 selects 0 below)
            const int nb = grads.bg ? grads.bg_channels : 0;
            const int hi = nb > 0 ? nb - 1 : 0;
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) {
                const int c = c0 + ch;
                const float b = bgp[c < hi ? c : hi];
                s_end = fmaf(c < nb ? b : 0.0f, g[ch], s_end);
            }
        }
        float gA = 0.0f;
        if (AUX && first_pass) {
            gD = dL_ddepth ? dL_ddepth[pix] : 0.0f;
            gA = dL_dalpha ? dL_dalpha[pix] : 0.0f;
            s_end -= gA;
        }
        T = (ST)final_T[pix];
        A = (ST)s_end;
        if constexpr (NC <= 4) {
            if (from_ckpt) {
                // later segments first (the smallest contributions), this boundary's neighbour last
                float sfx = final_T[pix] * s_end;
                const float* rec0 = ckpt_all + (size_t)view * SPLIT_PARTS_MAX * (C_total + 2) * plane + pix;
                for (int k = np - 1; k > seg; --k) {
                    const float* ck = rec0 + (size_t)k * (C_total + 2) * plane;
#pragma unroll
                    for (int ch = 0; ch < NC; ++ch) sfx = fmaf(ck[(size_t)(1 + c0 + ch) * plane], g[ch], sfx);
                    if (AUX) sfx = fmaf(ck[(size_t)(1 + C_total) * plane], gD, sfx);
                }
                const float Tb = rec0[(size_t)(seg + 1) * (C_total + 2) * plane];
                T = (ST)Tb;
                A = (ST)(sfx / Tb);        // (T_b >= 1e-4: the pixel was still active at the boundary)
            }
        }
    } else {
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) g[ch] = 0.0f;
    }
    // B operand of the contraction, kept in registers for the whole tile: for k-step kk and
    // channel half t, lane l holds dL/dcolor[pix = 4 kk + (l >> 4)][ch = 16 t + (l & 15)].
    // Built once by transposing through the (still unused) weight panel.
    float gt[MFMA ? 16 : 1][NB];
    if (MFMA) {
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ch = 0; ch < 16; ++ch) {
                const int c = 16 * t + ch;
                s_w[lane * WS + ch] = c < NC ? g[c < NC ? c : 0] : ((XD && c == NC) ? gD : 0.0f);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) gt[kk][t] = s_w[(4 * kk + (lane >> 4)) * WS + (lane & 15)];
        }
        __builtin_amdgcn_wave_barrier();
    }
    // this quadrant only needs the list up to its deepest contributor
    uint32_t wave_last = last;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, d, WAVE));
    const uint32_t end = min(end0, list0 + wave_last);
    if (NC <= 4) { if (split && beg >= end) return; }   // nothing of this half contributes

    // wave_reduce_pack leaves total k in lane bitreverse6(k); values [0, KV) belong to the first
    // Gaussian of a pair, [KV, 2 KV) to the second; inside a Gaussian: NV colours then 7 geometric
    const int slotv = (int)(__brev((unsigned)lane) >> 26);
    const bool slot_second = slotv >= KV;
    const int sv = slot_second ? slotv - KV : slotv;
    const bool slot_col = sv < NV;
    const bool slot_ok = slotv < 2 * KV;
    const int slot_off = slot_col ? (c0 + NM + sv) : (TM ? MO + 6 : MO + sv - NV);  // float offset inside the Gaussian's row
    int nslot = 0;  // Gaussians parked in the weight panel (wave-uniform)
    int e0 = 0;     // weight-panel slot of the E panel's column 0 (wave-uniform)

    // dL/dfeature of the parked Gaussians: D[g][ch] = sum_pix W[pix][g] * G[pix][ch]
    // (v_mfma_f32_16x16x4_f32: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15],
    //  D[row = 4 (l >> 4) + reg][col = l & 15])
    auto flush_panel = [&](int count) {
        __builtin_amdgcn_wave_barrier();
        f32x4 D[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) D[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int row = (lane >> 4) * WS + (lane & 15);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float a = s_w[4 * kk * WS + row];
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                D[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, gt[kk][t], D[t], 0, 0, 0);
            }
        }
        // column c = 16 t + (l & 15): channel c0 + c below NM, the depth weight (moment slot 6, first pass only) at NM
        const int j0c = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gs = 4 * (lane >> 4) + r;
            if (gs < count) {
                const uint32_t rowi = __umul24(s_gid[gs], (uint32_t)GROW);
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    const int c = 16 * t + j0c;
                    const bool full = 16 * t + 16 <= NM;   // every column of the block is a channel
                    const bool col_ok = full || c < NM || (XD && c == NM && first_pass);
                    const uint32_t col_off = (full || c < NM) ? (uint32_t)(c0 + c) : (uint32_t)(MO + 6);
                    if (col_ok) acc_add<DET>(gacc, gacc64, (size_t)(rowi + col_off), D[t][r], det_pass);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    // Moments of the parked E columns.  Lane (g = l & 7, i = l >> 3) owns Gaussian column g and pixel column i of the
    // quadrant: it reads E[pix = 8 j + i][g] for the 8 rows j, multiplies by ITS Gaussian's offsets d = mean - pixel
    // (dx is constant in the lane), and the first three stages of the packed butterfly (lane bits 5, 4, 3 = the pixel
    // column) fold the 8 columns: lane l ends with moment (l >> 5 & 1) + 2 (l >> 4 & 1) + 4 (l >> 3 & 1) of Gaussian g.
    auto reduce_e = [&](int count) {
        if constexpr (TM) {
            __builtin_amdgcn_wave_barrier();
            const int eg = lane & 7, ei = lane >> 3;
            const float2 mu = s_emean[min(e0 + eg, GROUP - 1)];
            const float dx = mu.x - (float)(qx + ei);
            const float dy0 = mu.y - (float)qy;
            float m[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // sum E, sum E dy, sum E dy^2 (then the six moments)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float e = s_e[(8 * j + ei) * ES + eg];
                const float dy = dy0 - (float)j;
                const float ey = e * dy;
                m[5] += e;
                m[1] += ey;
                m[4] = fmaf(ey, dy, m[4]);
            }
            m[0] = m[5] * dx;    // sum E dx
            m[2] = m[0] * dx;    // sum E dx^2
            m[3] = m[1] * dx;    // sum E dx dy
            const float outv = wave_reduce_hi3<6>(m, lane);
            const int vi = ((lane >> 5) & 1) + 2 * ((lane >> 4) & 1) + 4 * ((lane >> 3) & 1);
            if (eg < count && vi < 6) {
                const size_t di = (size_t)(__umul24(s_gid[e0 + eg], (uint32_t)GROW) + (uint32_t)(MO + vi));
                acc_add<DET>(gacc, gacc64, di, outv, det_pass);
            }
            __builtin_amdgcn_wave_barrier();
        }
    };

    // chunk in flight: the packed word (id | reach bits) of list entry base + lane — ONE register, so that the prefetch really stays in flight
    // while the current chunk is processed.  (Rounds 1-3 also prefetched the 32-byte record into eight registers; at 128 VGPRs
    // the compiler spilled them to scratch right behind the loads — `global_load ... s_waitcnt vmcnt(0) ... scratch_store` —
    // which exposed the whole memory latency once per chunk.  The records of the <= FS candidates are now loaded in the staging
    // round, together with their feature rows: one latency per round, as before, and only for this quadrant's candidates.)
    // (the word stays RAW — id | reach bits << 24: bit and id are extracted where they are consumed, one chunk later; extracting
    //  them here made the compiler wait for the load right behind the prefetch, an exposed memory latency per chunk)
    uint32_t pw = 0;
    auto fetch = [&](uint32_t base, uint32_t& w_) {
        w_ = 0u;
        if (base + (uint32_t)lane < end) w_ = ipack[base + (uint32_t)lane];
    };
    // The list is walked from the quadrant's deepest contributor towards the camera, 64 entries at a time (chunk k = entries
    // [beg + 64 k, beg + 64 k + 64)), the candidates of a chunk from the highest list position down.
    if (beg >= end) return;
    const uint64_t gt_mask = (lane == WAVE - 1) ? 0ull : (~0ull << (lane + 1));   // list positions behind this lane's
    const int nchunks = (int)((end - beg + (WAVE - 1)) / WAVE);
    fetch(beg + (uint32_t)(nchunks - 1) * WAVE, pw);

#pragma unroll 1
    for (int chunk = nchunks - 1; chunk >= 0; --chunk) {
        const uint32_t base = beg + (uint32_t)chunk * WAVE;
        const bool cur_reach = (pw >> (24 + quad)) & 1u;
        uint64_t cand = __builtin_amdgcn_ballot_w64(cur_reach);
        const uint32_t cur_gid = pw & 0xFFFFFFu;
        // the chunk in front: requested now, consumed after this one (the loop's last iteration requests nothing: the guard
        // is per lane, so there is no wave-uniform branch around the loads)
        fetch(chunk > 0 ? base - WAVE : 0xFFFFFF00u, pw);
        const uint32_t idx0 = base - list0;
#pragma unroll 1
        while (cand != 0) {
            // ---- stage the records and feature rows of the LAST <= FS candidates (row 0 = the deepest) ----
            const int ncand = min(FS, (int)__popcll(cand));
            const int rank = __popcll(cand & gt_mask);
            const bool mine = cur_reach && ((cand >> lane) & 1ull) && rank < FS;
            __builtin_amdgcn_wave_barrier();
            if (mine) s_cgid[rank] = cur_gid;
            float4 rec_a = make_float4(0.f, 0.f, 0.f, 0.f), rec_b = rec_a;
            if (mine) {
                const size_t j = (size_t)base + lane;
                rec_a = irec[2 * j];
                rec_b = irec[2 * j + 1];
            }
            __builtin_amdgcn_wave_barrier();
            // 16-byte pieces of the 16-byte-aligned padded rows
#pragma unroll SR_BWD_STAGE_UNROLL
            for (int e = lane; e < ncand * PPR; e += WAVE) {
                const int row = e / PPR, pc = e - row * PPR;
                reinterpret_cast<float4*>(s_feat)[e] = featp4[(size_t)(__umul24(s_cgid[row] - row0, (uint32_t)CP4) + (uint32_t)((c0 >> 2) + pc))];
            }
            if (mine) {
                s_rec0[rank] = rec_a;
                s_rec1[rank] = rec_b;
            }
            __builtin_amdgcn_wave_barrier();
            // One pair of candidates (list positions j0 > j1: the deeper one first; staged rows r0, r1); qm0 / qm1 are
            // their dot products when the matrix pipe already produced them.
            auto process_pair = [&](int j0, int j1, bool has1, int r0, int r1, float qm0, float qm1) {
                const float4 p0 = s_rec0[r0], q0 = s_rec1[r0];
                const float4 p1 = s_rec0[r1], q1 = s_rec1[r1];
                const float dx0 = p0.x - fx, dy0 = p0.y - fy, dx1 = p1.x - fx, dy1 = p1.y - fy;
                const float pw0 = gauss_log2(q0, dx0, dy0), pw1 = gauss_log2(q1, dx1, dy1);  // log2 of the weight
                // (alpha is tested before the min with 0.99: same decision as the forward's, and a NaN from an overflowed power
                //  is a miss — exp2_core; for the same reason a miss zeroes G below, not dL/dalpha)
                const float Gr0 = exp2_core(pw0), Gr1 = exp2_core(pw1);
                const float ar0 = q0.w * Gr0, ar1 = q1.w * Gr1;
                const float al0 = fminf(ALPHA_MAX, ar0), al1 = fminf(ALPHA_MAX, ar1);
                const bool hit0 = (idx0 + (uint32_t)j0 < last) && pw0 <= 0.0f && ar0 >= ALPHA_MIN;
                const bool hit1 = has1 && (idx0 + (uint32_t)j1 < last) && pw1 <= 0.0f && ar1 >= ALPHA_MIN;
                const float G0 = hit0 ? Gr0 : 0.0f, G1 = hit1 ? Gr1 : 0.0f;
                // ---- dot products q = f . g (+ depth) of both Gaussians ----
                float qd0 = qm0, qd1 = qm1;
                if (!DOTM) {
                    const float* f0 = &s_feat[r0 * NCP];
                    const float* f1 = &s_feat[r1 * NCP];
                    qd0 = AUX ? p0.z * gD : 0.0f;
                    qd1 = AUX ? p1.z * gD : 0.0f;
#pragma unroll
                    for (int ch = 0; ch < NC; ++ch) {
                        qd0 += f0[ch] * g[ch];
                        qd1 += f1[ch] * g[ch];
                    }
                    // a round's last pair may have no second member: row r0 + 1 is then stale (or never written: any bit
                    // pattern), and d = q - A with q = NaN would poison the chain — everything else of that half is masked
                    qd1 = has1 ? qd1 : 0.0f;
                }
                // ---- Gaussian 0 (the deeper one), then Gaussian 1: sequential in T and A ----
                // Branch-free: a miss keeps T and A and has w = 0, E = G dA = 0.  T_i = T_{i+1} / (1 - alpha) through the hardware
                // reciprocal (1 ulp; alpha <= 0.99 keeps the argument >= 0.01) — a relative error per step, harmless; the correctly
                // rounded division would be 10 instructions per Gaussian.
                float dA0, dA1, w0, w1;
                if constexpr (DET) {
                    const double Ti0 = hit0 ? T / (1.0 - (double)al0) : T;
                    const double d0 = (double)qd0 - A;
                    dA0 = (float)(Ti0 * d0);
                    const double ah0 = hit0 ? (double)al0 : 0.0;
                    A = fma(ah0, d0, A);
                    w0 = (float)(ah0 * Ti0);
                    const double Ti1 = hit1 ? Ti0 / (1.0 - (double)al1) : Ti0;
                    const double d1 = (double)qd1 - A;
                    dA1 = (float)(Ti1 * d1);
                    const double ah1 = hit1 ? (double)al1 : 0.0;
                    A = fma(ah1, d1, A);
                    w1 = (float)(ah1 * Ti1);
                    T = Ti1;
                } else {
                    const float Ti0 = hit0 ? T * __builtin_amdgcn_rcpf(1.0f - al0) : T;
                    const float d0 = qd0 - A;
                    dA0 = Ti0 * d0;                               // (finite also for a miss; E = G dA = 0 there)
                    const float ah0 = hit0 ? al0 : 0.0f;
                    A = fmaf(ah0, d0, A);
                    w0 = ah0 * Ti0;
                    const float Ti1 = hit1 ? Ti0 * __builtin_amdgcn_rcpf(1.0f - al1) : Ti0;
                    const float d1 = qd1 - A;
                    dA1 = Ti1 * d1;
                    const float ah1 = hit1 ? al1 : 0.0f;
                    A = fmaf(ah1, d1, A);
                    w1 = ah1 * Ti1;
                    T = Ti1;
                }
                const float E0 = G0 * dA0, E1 = G1 * dA1;   // (0 for a miss)
                const uint32_t gi0 = s_cgid[r0], gi1 = s_cgid[r1];
                if constexpr (KV > 0) {
                    float red[2 * KV];
#pragma unroll
                    for (int ch = 0; ch < NV; ++ch) {
                        red[ch] = w0 * g[NM + ch];
                        red[KV + ch] = w1 * g[NM + ch];
                    }
                    // geometric partials as raw moments of E = G dL/dalpha over the pixel offset d;
                    // the per-Gaussian factors (conic, opacity, 0.5 W / 0.5 H) are applied once per
                    // Gaussian in preprocess_bwd instead of once per (pixel, Gaussian) here
                    if constexpr (!TM) {
                        const float Ex0 = E0 * dx0, Ey0 = E0 * dy0, Ex1 = E1 * dx1, Ey1 = E1 * dy1;
                        red[NV + 0] = Ex0;
                        red[NV + 1] = Ey0;
                        red[NV + 2] = Ex0 * dx0;
                        red[NV + 3] = Ex0 * dy0;
                        red[NV + 4] = Ey0 * dy0;
                        red[NV + 5] = E0;
                        red[KV + NV + 0] = Ex1;
                        red[KV + NV + 1] = Ey1;
                        red[KV + NV + 2] = Ex1 * dx1;
                        red[KV + NV + 3] = Ex1 * dy1;
                        red[KV + NV + 4] = Ey1 * dy1;
                        red[KV + NV + 5] = E1;
                    }
                    if constexpr (!XD && AUX) {
                        red[KV - 1] = w0 * gD;
                        red[2 * KV - 1] = w1 * gD;
                    }
                    {
                        const float outv = wave_reduce_pack<2 * KV>(red, lane);
                        const uint32_t gi = slot_second ? gi1 : gi0;
                        const size_t di = (size_t)(__umul24(gi, (uint32_t)GROW) + (uint32_t)slot_off);   // gi < 2^24 (checked on the host)
                        if (slot_ok && (has1 || !slot_second)) acc_add<DET>(gacc, gacc64, di, outv, det_pass);
                    }
                }
                if constexpr (MFMA) {
                    // park the weights (0 for pixels that miss); a pair never straddles a flush
                    s_w[lane * WS + nslot] = w0;
                    s_w[lane * WS + nslot + 1] = w1;
                    if constexpr (TM) {   // and E = G dL/dalpha in the 8-column panel (column = slot - e0)
                        s_e[lane * ES + (nslot - e0)] = E0;
                        s_e[lane * ES + (nslot - e0) + 1] = E1;
                    }
                    if (lane == 0) {
                        s_gid[nslot] = gi0;
                        s_gid[nslot + 1] = gi1;
                        if constexpr (TM) {
                            s_emean[nslot] = make_float2(p0.x, p0.y);
                            s_emean[nslot + 1] = make_float2(p1.x, p1.y);
                        }
                    }
                    nslot += has1 ? 2 : 1;
                    if constexpr (TM) {
                        if (nslot - e0 >= EG - 1 || nslot >= GROUP - 1) {
                            reduce_e(nslot - e0);
                            e0 = nslot;
                        }
                    }
                    if (nslot >= GROUP - 1) {
                        flush_panel(nslot);
                        nslot = 0;
                        e0 = 0;
                    }
                }
            };
            // q[pix][g] = sum_ch F[g][ch] G[pix][ch] + z_g g_D for four staged rows on the matrix pipe:
            // v_mfma_f32_4x4x1_16b_f32 computes D[lane 4b+j][r] += A[lane 4b+r] * B[lane 4b+j]; with A =
            // feature of Gaussian (lane & 3) and B = this pixel's gradient, register r of every lane ends
            // up holding q[own pixel][Gaussian r].
            auto dot4 = [&](int slot4) -> f32x4 {
                const int k = lane & 3;
                const int srow = min(slot4 + k, ncand - 1);
                const float* fr = &s_feat[srow * NCP];
                const float zsel = reinterpret_cast<const float*>(&s_rec0[srow])[2];
                constexpr int CH = SR_BWD_DOT_CHAINS;   // independent accumulation chains
                f32x4 Qc[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) Qc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#if SR_BWD_DOT_PREFETCH
                // the next 16-byte piece of the row is requested before the four MFMAs of the current one
                const float4* f4 = reinterpret_cast<const float4*>(fr);
                float4 cur = f4[0];
#pragma unroll
                for (int i = 0; i < PPR; ++i) {
                    const float4 nxt = f4[i + 1 < PPR ? i + 1 : i];
                    const float v[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ch = 4 * i + q;
                        if (ch < NC) Qc[ch % CH] = __builtin_amdgcn_mfma_f32_4x4x1f32(v[q], g[ch < NC ? ch : 0], Qc[ch % CH], 0, 0, 0);
                    }
                    cur = nxt;
                }
#else
#pragma unroll
                for (int ch = 0; ch < NC; ++ch)
                    Qc[ch % CH] = __builtin_amdgcn_mfma_f32_4x4x1f32(fr[ch], g[ch], Qc[ch % CH], 0, 0, 0);
#endif
                Qc[NC % CH] = __builtin_amdgcn_mfma_f32_4x4x1f32(zsel, gD, Qc[NC % CH], 0, 0, 0);
                f32x4 Q = Qc[0];
#pragma unroll
                for (int c = 1; c < CH; ++c) Q += Qc[c];
                return Q;
            };
#pragma unroll 1
            for (int slot = 0; slot < ncand; slot += (DOTM ? 4 : 2)) {
                // pop the next two (four) candidates, deepest first; hasK are wave-uniform
                const bool has1 = slot + 1 < ncand, has2 = DOTM && slot + 2 < ncand, has3 = DOTM && slot + 3 < ncand;
                const int j0 = 63 - __builtin_clzll(cand);
                cand &= ~(1ull << j0);
                const int j1 = has1 ? 63 - __builtin_clzll(cand) : j0;
                if (has1) cand &= ~(1ull << j1);
                const int j2 = has2 ? 63 - __builtin_clzll(cand) : j1;
                if (has2) cand &= ~(1ull << j2);
                const int j3 = has3 ? 63 - __builtin_clzll(cand) : j2;
                if (has3) cand &= ~(1ull << j3);
                float qm[4] = {0.f, 0.f, 0.f, 0.f};
                if (DOTM) {
                    const f32x4 Q = dot4(slot);
                    qm[0] = Q[0]; qm[1] = Q[1]; qm[2] = Q[2]; qm[3] = Q[3];
                }
                process_pair(j0, j1, has1, slot, slot + 1, qm[0], qm[1]);
                if (has2) process_pair(j2, j3, has3, slot + 2, slot + 3, qm[2], qm[3]);
            }
        }
    }
    if constexpr (TM) { if (nslot - e0 > 0) reduce_e(nslot - e0); }
    if (MFMA && nslot > 0) flush_panel(nslot);
}



static int g_small_panel_max_waves = SR_BWD_SMALL_PANEL_MAX_WAVES;
void set_small_panel_max_waves(int waves) { g_small_panel_max_waves = waves < 0 ? SR_BWD_SMALL_PANEL_MAX_WAVES : waves; }

struct BwdLaunch {
    int P, V;
    const WinGrad* grads;
    const float* ckpt;   // non-null: split launch — SPLIT_PARTS waves per quadrant, wave k > 0 from the forward's segment records (common.h)
    int det_pass;        // deterministic mode: 0 = per-element max pass, 1 = fixed-point sum pass (acc_add)
};

template <int NC, bool DET, bool AUX = true>
static int launch_one_bwd(const splatraster_settings& s, int c0, int first, const GeomView& g,
                          const BinView& b, const ImgView& im, const float* feat, int feat_stride,
                          const BwdLaunch& L, float* gacc, long long* gacc64, hipStream_t stream)
{
    (void)g;
    const int gx = (s.image_width + TILE - 1) / TILE, gy = (s.image_height + TILE - 1) / TILE;
    const int tiles = gx * gy;
    const unsigned blocks = quadrant_blocks(L.V * tiles, gx);  // 4 quadrants per (view, tile) (+ padding of the id space)
    const float* ckpt = (NC <= 4 && c0 == 0 && first) ? L.ckpt : nullptr;
    // split launches with a launch order: three groups of extra workgroups (parts 4 .. 15 of the SPLIT_EXTRA_TILES longest lists) lead the grid
    const bool extras = ckpt != nullptr && use_tile_order(L.V, tiles);
    const unsigned extra_blocks = extras ? (unsigned)(SPLIT_PARTS_MAX / SPLIT_PARTS - 1) * quadrant_blocks(SPLIT_EXTRA_TILES, gx) : 0u;
    const dim3 grid(blocks + extra_blocks, ckpt ? (unsigned)SPLIT_PARTS : 1u);
#define SR_BWD_LAUNCH(SPV)                                                                                          \
    hipLaunchKernelGGL((composite_bwd_kernel<NC, DET, SPV, AUX>), grid, dim3(WAVE), 0, stream, s.image_width,           \
                       s.image_height, feat_stride, padded_channels(feat_stride) / 4, c0, first, tiles, L.V, L.P,      \
                       b.ranges, b.ipack, b.irec, reinterpret_cast<const float4*>(feat), *L.grads,       \
                       im.final_T, im.n_contrib, gacc, gacc_row_floats(s.channels),                                    \
                       gacc_moment_offset(s.channels), gacc64, ckpt, extras ? b.nparts : nullptr, (int)extra_blocks,               \
                       use_tile_order(L.V, tiles) ? b.tile_order : nullptr, L.det_pass)
    if constexpr (NC >= 4 && NC <= 15 && AUX) {   // C = 3 and below: the flush costs what the 8 saved butterfly values gain (A/B: S0 0.036 vs 0.041 ms)
        // per VIEW: small frames (SplatLoc's 640x480) take the panel variant — also as a window of V views (A/B at the
        // reference layout, 5 views: 0.816 vs 0.869 ms); large frames the butterfly variant at full occupancy
        if (4 * tiles <= g_small_panel_max_waves)
            SR_BWD_LAUNCH(true);
        else
            SR_BWD_LAUNCH(false);
    } else {
        SR_BWD_LAUNCH(false);
    }
#undef SR_BWD_LAUNCH
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

int launch_composite_bwd(const splatraster_settings& s, int32_t P, int32_t V, int64_t R, const GeomView& g,
                         const BinView& b, const ImgView& im, const float* feat, int feat_stride,
                         const WinGrad& grads, float* gacc, long long* gacc64, int det_pass, hipStream_t stream)
{
    if (R == 0) return SPLATRASTER_OK;
    const bool det = gacc64 != nullptr;
    int C = s.channels;
    const int tiles_v = ((s.image_width + TILE - 1) / TILE) * ((s.image_height + TILE - 1) / TILE);
    const BwdLaunch L{P, V, &grads, (!det && split_lists(s.channels, V, tiles_v)) ? b.ckpt : nullptr, det_pass};
    // Channels and auxiliary planes that did not reach the loss are not computed: when the last channel's gradient
    // travels apart (grads.gc = C - 1) and NO view has one, the launch covers channels [0, C - 1) only — its dL/dfeature
    // column stays at the zero the accumulator rows were cleared to; without any depth / alpha gradient the RGB kernel
    // drops the depth weight from the dot product and the reduction (color_refinement: train_gaussians.py:283-285).
    bool any_last = false, any_aux = false;
    for (int v = 0; v < V; ++v) {
        any_last = any_last || grads.dL_dlast[v] != nullptr;
        any_aux = any_aux || grads.dL_ddepth[v] != nullptr || grads.dL_dalpha[v] != nullptr;
    }
    if (grads.gc < C && !any_last) C = grads.gc;
    if (C == 3 && !any_aux && !det)   // (the panel variant of this kernel measured equal at 640x480: 0.752 vs 0.760 ms per refinement iteration)
        return launch_one_bwd<3, false, false>(s, 0, 1, g, b, im, feat, feat_stride, L, gacc, gacc64, stream);
#define SR_BWD_ARGS g, b, im, feat, feat_stride, L, gacc, gacc64, stream
#define SR_BWD_ONE(N, c0_, first_) (det ? launch_one_bwd<N, true>(s, c0_, first_, SR_BWD_ARGS) : launch_one_bwd<N, false>(s, c0_, first_, SR_BWD_ARGS))
#define SR_BWD_CASE(N) \
    case N: return SR_BWD_ONE(N, 0, 1);
    switch (C) {
        SR_BWD_CASE(1) SR_BWD_CASE(2) SR_BWD_CASE(3) SR_BWD_CASE(4) SR_BWD_CASE(8) SR_BWD_CASE(16)
        SR_BWD_CASE(32) SR_BWD_CASE(35)
        default: break;
    }
#undef SR_BWD_CASE
    int c0 = 0, first = 1, st = SPLATRASTER_OK;
    while (c0 < C && st == SPLATRASTER_OK) {
        const int left = C - c0;
        if (left >= 32) { st = SR_BWD_ONE(32, c0, first); c0 += 32; }
        else if (left >= 16) { st = SR_BWD_ONE(16, c0, first); c0 += 16; }
        else if (left >= 8) { st = SR_BWD_ONE(8, c0, first); c0 += 8; }
        else if (left >= 4) { st = SR_BWD_ONE(4, c0, first); c0 += 4; }
        else if (left == 3) { st = SR_BWD_ONE(3, c0, first); c0 += 3; }
        else if (left == 2) { st = SR_BWD_ONE(2, c0, first); c0 += 2; }
        else { st = SR_BWD_ONE(1, c0, first); c0 += 1; }
        first = 0;
    }
#undef SR_BWD_ONE
#undef SR_BWD_ARGS
    return st;
}

}  // namespace sr

