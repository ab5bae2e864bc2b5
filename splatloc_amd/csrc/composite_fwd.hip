// composite_fwd.hip — front-to-back alpha compositing of the per-tile depth-sorted lists.
//
// Replaces the render stage of the rasterizer extension behind
// diff_gauss.GaussianRasterizer.forward (gaussian_renderer/__init__.py:117-126; algorithm
// per SURVEY.md §8a "COMPOSITE fwd").
//
// Machine mapping: ONE wave64 = one workgroup = one 8x8 pixel quadrant of a
// 16x16 tile.  The four quadrants of a tile have very different amounts of work; as waves of
// one 256-thread workgroup they met at three barriers per batch and spent > 50 % of their
// cycles waiting.  Here every quadrant walks the tile's list on its own, wave-synchronously,
// with no __syncthreads anywhere:
//   * the list is streamed 64 entries at a time from the per-instance payload written by
//     payload_kernel (binning.hip): packed word (id | reach mask) + 32-byte record, contiguous in sorted
//     order -> independent coalesced loads per lane, issued one chunk ahead;
//   * only entries whose mask says they may reach THIS quadrant become candidates; their
//     feature rows (4*C bytes) are gathered into LDS, at most FS rows per round;
//   * candidates are composited two at a time (twice the ILP of the alpha evaluation, and
//     exactly the K = 2 of the matrix instruction below).
//
// NC >= 32: the accumulation of the first 32 channels, out[pix][ch] += w[pix][g] * F[g][ch],
// is a [32 ch x 2 g] x [2 g x 32 pix] product per candidate pair and runs on the matrix pipe
// (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fmaf chain = the sequential front-to-back
// sum).  A operand: one conflict-free ds_read_b32 of the two staged feature rows; B operand:
// the two weight registers of the pair after ONE v_permlane32_swap.
#include <type_traits>

#include "composite_common.h"

#ifndef SR_FWD_FS
#define SR_FWD_FS 24  // feature rows staged per round (<= 64; A/B on S2, 5 cameras: 12: 0.399, 16: 0.417, 20: 0.411, 24: 0.391, 32: 0.401, 40: 0.417, 48: 0.442 ms)
#endif
// (the wrong-result timing probes of rounds 2-3 live in tools/patches/composite_probes.patch, not here)
#ifndef SR_FWD_STAGE_UNROLL
#define SR_FWD_STAGE_UNROLL 3  // gather iterations in flight together while staging feature rows (A/B on S2: 1: 0.429, 2: 0.456, 3: 0.416, 5: 0.478 ms)
#endif

#ifndef SR_FWD_M4
#define SR_FWD_M4 1  // leftover channels + depth of a wide layout on v_mfma_f32_4x4x1 (0: VALU pair sums)
#endif
#ifndef SR_FWD_MINW
#define SR_FWD_MINW 4  // waves per SIMD the register allocator must allow
#endif

namespace sr {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// In-place half swap: afterwards x = [x.lanes0-31 | y.lanes0-31], y = [x.lanes32-63 | y.lanes32-63].
// Inline asm on purpose: with ROCm 7.2 hipcc, feeding BOTH results of
// __builtin_amdgcn_permlane32_swap to MFMA B operands made the second MFMA read the first
// result's register (seen in the .s; image rows 4-7 of every quadrant repeated rows 0-3).
// The s_nop pads cover VALU-write -> permlane read and permlane write -> MFMA read hazards,
// which the compiler does not insert inside asm statements.
__device__ __forceinline__ void swap_halves(float& x, float& y)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}

// acc += A * B with v_mfma_f32_32x32x2_f32, accumulating IN PLACE.  Through the builtin, hipcc
// (ROCm 7.2) put the result of the MFMA inside the wave-uniform `if (any hit)` into a different
// register block than the accumulator and merged the two paths with s_nop 10 + 32 v_mov per
// candidate pair, exposing the whole 64-cycle MFMA latency; the tied "+v" operand forbids that.
// Hazards the compiler does not see inside asm: VALU-written A/B -> MFMA read (s_nop 1 here);
// MFMA D -> VALU read is covered once by mfma_drain() before the epilogue; D -> next MFMA as C
// needs no wait states.
__device__ __forceinline__ void mfma_acc_32x32x2(f32x16& acc, float a, float b)
{
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_drain(f32x16& x, f32x16& y)
{
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(x), "+v"(y));
}
// acc[r] += a * b per lane group of 4 (v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 outer products, K = 1).  Lane l supplies
// A[block l/4][row l%4] and B[block l/4][column l%4]; its four result registers are rows 0-3 of column l%4.  With a = value
// (l % 4) of a Gaussian's leftover row and b = the lane's own blending weight, register r of lane l accumulates value r for
// the lane's OWN pixel: four per-pixel FMAs as one matrix-pipe instruction (2 passes) instead of twelve VALU instructions per pair.
__device__ __forceinline__ void mfma_acc_4x4x1(f32x4& acc, float a, float b)
{
    asm volatile("s_nop 1\n\tv_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_drain4(f32x4& x)
{
    asm volatile("s_nop 7" : "+v"(x));
}

// Index of the lowest set bit; -1 for an empty mask (s_ff1_i32_b64's own convention, which __builtin_ctzll leaves undefined).
__device__ __forceinline__ int first_bit(uint64_t m)
{
    int j;
    asm("s_ff1_i32_b64 %0, %1" : "=s"(j) : "s"(m));
    return j;
}

template <int NC>
struct FwdCfg {
    static constexpr bool MFMA = NC >= 32;
    static constexpr int NM = MFMA ? 32 : 0;   // channels accumulated on the matrix pipe
    static constexpr int NV = NC - NM;         // channels accumulated with VALU FMAs
    static constexpr int NCP = (NC + 3) & ~3;  // LDS row stride (floats), 16-B aligned rows
    // the leftover channels (< 4) and the depth ride a 4x4x1 matrix instruction: the depth takes the staged row's padding slot
    static constexpr bool M4 = SR_FWD_M4 && MFMA && NV > 0 && NV <= 3 && NM + NV < NCP;
    static constexpr int FS = SR_FWD_FS;             // feature rows staged per round
};

template <int NC>
__global__ void __launch_bounds__(WAVE, SR_FWD_MINW)   // (layouts of <= 4 channels: composite_fwd_narrow_kernel / _mixed_kernel below)
composite_fwd_kernel(int W, int H, int CP4, int c0, int bg_channels, int write_aux, int tiles /*per view*/, int V,
                     int P /*rows per view*/,
                     const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ ipack /*id | reach bits << 24*/,
                     const float4* __restrict__ irec, const float4* __restrict__ featp4,
                     const float* __restrict__ bg, WinOut outs,
                     float* __restrict__ final_T_all, uint32_t* __restrict__ n_contrib_all,
                     const uint32_t* __restrict__ tile_order /*launch order (binning.hip), or null*/)
{
    static_assert(NC > 4, "narrow layouts have their own kernels");
    using Cfg = FwdCfg<NC>;
    constexpr int NCP = Cfg::NCP, NM = Cfg::NM, NV = Cfg::NV, FS = Cfg::FS;
    constexpr bool MFMA = Cfg::MFMA, M4 = Cfg::M4;
    constexpr int PPR = NCP / 4;  // 16-byte pieces per staged row
    // records of the chunk at [1 + j]; entry 0 is the ABSENT candidate (opacity 0: alpha = 0, never live): an odd round's last
    // pair reads it through j1 = -1 (s_ff1 of an empty mask) instead of carrying a wave-uniform "has a second Gaussian" flag
    // through the pair loop (9 scalar + 4 vector instructions per pair of select / mask algebra)
    __shared__ __attribute__((aligned(16))) float4 s_rec0[WAVE + 1];
    __shared__ __attribute__((aligned(16))) float4 s_rec1[WAVE + 1];
    static_assert(FS % 2 == 0, "pairs: an odd round ends below FS, its zeroed row FS - 1 at most");
    __shared__ __attribute__((aligned(16))) float s_feat[FS * NCP];
    __shared__ uint32_t s_cgid[FS];

    int gtile, quad;   // global tile = view * tiles + tile: the grid covers the V views of the window
    const int gx = (W + TILE - 1) / TILE;
    quadrant_of_block(blockIdx.x, V * tiles, gx, gtile, quad, tile_order);
    if (gtile >= V * tiles) return;
    const int view = (V == 1) ? 0 : gtile / tiles;      // wave-uniform (scalar)
    const int tile = gtile - view * tiles;
    const uint32_t row0 = (uint32_t)view * (uint32_t)P;  // the view's first row: feature row of row g is g - row0
    float* __restrict__ out_color = outs.color[view];
    float* __restrict__ out_depth = outs.depth[view];
    float* __restrict__ out_alpha = outs.alpha[view];
    float* __restrict__ final_T = final_T_all + (size_t)view * H * W;
    uint32_t* __restrict__ n_contrib = n_contrib_all + (size_t)view * H * W;
    const int lane = threadIdx.x;
    const int qx = (tile % gx) * TILE + (quad & 1) * 8, qy = (tile / gx) * TILE + (quad >> 1) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
    const uint32_t beg = ranges[2 * gtile], end = ranges[2 * gtile + 1];

    if (lane == 0) {
        s_rec0[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_rec1[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bool active = inside;  // pixel still accumulating
    float T = 1.0f, D = 0.0f;
    float acc[NV > 0 ? NV : 1];
#pragma unroll
    for (int ch = 0; ch < NV; ++ch) acc[ch] = 0.0f;
    // matrix-pipe accumulators D[ch][pix]: accA = pixels (lanes) 0-31 of the wave, accB = 32-63
    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accA[r] = 0.0f; accB[r] = 0.0f; }
    f32x4 acc4 = {0.0f, 0.0f, 0.0f, 0.0f};   // M4: channels NM .. NM + 2 and the depth (register 3) of the lane's own pixel
    uint32_t last = 0;
    const int lb32 = (lane >> 5) * NCP + (lane & 31);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (WAVE - lane));

    // chunk in flight: the packed word (id | reach bits << 24) and the record of list entry base + lane.  The word stays RAW: the
    // quadrant's bit and the id are extracted where they are consumed, one chunk later (extracting the bit here made the compiler
    // wait for the load right behind the prefetch — `s_waitcnt vmcnt(3)` — an exposed memory latency per chunk).
    uint32_t pw = 0;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    auto fetch = [&](uint32_t base, uint32_t& w_, float4& x0, float4& x1) {
        w_ = 0u;
        if (base + (uint32_t)lane < end) {
            const uint32_t j = base + (uint32_t)lane;
            w_ = ipack[j];
            x0 = irec[2 * (size_t)j];
            x1 = irec[2 * (size_t)j + 1];
        }
    };
    fetch(beg, pw, a0, a1);

    bool wave_done = __builtin_amdgcn_ballot_w64(active) == 0;
#pragma unroll 1
    for (uint32_t base = beg; base < end && !wave_done; base += WAVE) {
        const bool cur_reach = (pw >> (24 + quad)) & 1u;
        uint64_t cand = __builtin_amdgcn_ballot_w64(cur_reach);
        const uint32_t cur_gid = pw & 0xFFFFFFu;
        if (cand != 0) {
            __builtin_amdgcn_wave_barrier();
            s_rec0[1 + lane] = a0;
            s_rec1[1 + lane] = a1;
        }
        // next chunk: issued now, consumed after this chunk has been composited
        fetch(base + WAVE, pw, a0, a1);
#pragma unroll 1
        while (cand != 0 && !wave_done) {
            // ---- stage the feature rows of the next <= FS candidates ----
            const int rank = __popcll(cand & lt_mask);
            const int ncand = min(FS, (int)__popcll(cand));
            __builtin_amdgcn_wave_barrier();
            if (cur_reach && ((cand >> lane) & 1ull) && rank < FS) s_cgid[rank] = cur_gid - row0;   // feature row (shared by the views)
            __builtin_amdgcn_wave_barrier();
            // 16-byte pieces of the 16-byte-aligned padded rows
#pragma unroll SR_FWD_STAGE_UNROLL
            for (int e = lane; e < ncand * PPR; e += WAVE) {
                const int row = e / PPR, pc = e - row * PPR;
                reinterpret_cast<float4*>(s_feat)[e] = featp4[(size_t)(__umul24(s_cgid[row], (uint32_t)CP4) + (uint32_t)((c0 >> 2) + pc))];   // ids < 2^24: checked on the host
            }
            __builtin_amdgcn_wave_barrier();
            if (ncand & 1) {      // the absent candidate's feature row: finite, whatever the LDS held
                if (lane < PPR) reinterpret_cast<float4*>(s_feat)[ncand * PPR + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
                __builtin_amdgcn_wave_barrier();
            }
            if constexpr (M4) {   // the candidate's depth into the padding slot of its staged row (after the row's pieces: LDS keeps program order)
                if (cur_reach && ((cand >> lane) & 1ull) && rank < FS) s_feat[rank * NCP + NM + 3] = s_rec0[1 + lane].z;
                __builtin_amdgcn_wave_barrier();
            }
            // ---- composite them front to back, two at a time ----
#pragma unroll 1
            for (int slot = 0; slot < ncand; slot += 2) {
                const int j0 = first_bit(cand);
                cand &= cand - 1;
                const int j1 = first_bit(cand);   // -1 in the last pair of an odd round: the absent candidate
                cand &= cand - 1;                 // (an empty mask stays empty)
                const float4 p0 = s_rec0[1 + j0], q0 = s_rec1[1 + j0];
                const float4 p1 = s_rec0[1 + j1], q1 = s_rec1[1 + j1];
                const int s1 = slot + 1;          // row ncand of an odd round was zeroed while staging
                // matrix operands of the pair: read NOW, beside the records (left to the scheduler, each read sat right in front
                // of the instruction that consumes it: an exposed LDS round trip per operand and pair)
                float a32 = 0.0f, a40 = 0.0f, a41 = 0.0f;
                if constexpr (MFMA) a32 = s_feat[slot * NCP + lb32];  // A[i = ch][k = g]: lanes 32-63 read the pair's second row
                if constexpr (M4) {
                    const int l4 = NM + (lane & 3);
                    a40 = s_feat[slot * NCP + l4];
                    a41 = s_feat[s1 * NCP + l4];
                }
                if constexpr (MFMA) __builtin_amdgcn_sched_barrier(0);
                const float dx0 = p0.x - fx, dy0 = p0.y - fy, dx1 = p1.x - fx, dy1 = p1.y - fy;
                const float pw0 = gauss_log2(q0, dx0, dy0), pw1 = gauss_log2(q1, dx1, dy1);  // log2 of the weight
                // (alpha is tested BEFORE the min with 0.99, like the backward: a NaN from an overflowed power is a miss, which is
                //  the only thing the argument clamp of exp2_shared was there for — composite_common.h)
                const float ar0 = q0.w * exp2_core(pw0), ar1 = q1.w * exp2_core(pw1);
                const float al0 = fminf(ALPHA_MAX, ar0), al1 = fminf(ALPHA_MAX, ar1);
                // Gaussian 0
                const bool live0 = active && pw0 <= 0.0f && ar0 >= ALPHA_MIN;
                const float tT0 = transmit(T, al0);
                const bool hit0 = live0 && tT0 >= T_EPS;
                const bool act1 = active && !(live0 && !hit0);  // transmittance exhausted: pixel finished
                const float w0 = hit0 ? al0 * T : 0.0f;
                const float T1 = hit0 ? tT0 : T;
                // Gaussian 1 (the absent candidate has alpha = 0)
                const bool live1 = act1 && pw1 <= 0.0f && ar1 >= ALPHA_MIN;
                const float tT1 = transmit(T1, al1);
                const bool hit1 = live1 && tT1 >= T_EPS;
                active = act1 && !(live1 && !hit1);
                const float w1 = hit1 ? al1 * T1 : 0.0f;
                T = hit1 ? tT1 : T1;
                const uint32_t idx = base - beg;
                last = hit1 ? idx + (uint32_t)j1 + 1u : (hit0 ? idx + (uint32_t)j0 + 1u : last);
                // No "skip if nobody hit" branch here on purpose: with the reach masks ~95 % of the
                // candidates hit, and a conditional around the accumulation makes hipcc merge the two
                // paths by copying all 32 accumulator registers per pair (seen in the .s).
                const float* f0 = &s_feat[slot * NCP + NM];
                const float* f1 = &s_feat[s1 * NCP + NM];
                if constexpr (M4) {
                    mfma_acc_4x4x1(acc4, a40, w0);
                    mfma_acc_4x4x1(acc4, a41, w1);
                } else {
#pragma unroll
                    for (int ch = 0; ch < NV; ++ch) {
                        // narrow layouts: two FMAs in list order (forward -2.7 % at C = 3 / 4); beside the MFMA accumulation of
                        // wide layouts the pair-sum form (mul + fma + add) measured FASTER (1.90 vs 1.99 ms per window)
                        if (MFMA) acc[ch] += f0[ch] * w0 + f1[ch] * w1;
                        else acc[ch] = fmaf(f1[ch], w1, fmaf(f0[ch], w0, acc[ch]));
                    }
                    if (MFMA) D += p0.z * w0 + p1.z * w1;
                    else D = fmaf(p1.z, w1, fmaf(p0.z, w0, D));
                }
                if (MFMA) {
                    float b0 = w0, b1 = w1;
                    swap_halves(b0, b1);  // b0 -> B for pixels 0-31, b1 -> B for pixels 32-63
                    mfma_acc_32x32x2(accA, a32, b0);
                    mfma_acc_32x32x2(accB, a32, b1);
                }
                if (__builtin_amdgcn_ballot_w64(active) == 0) { wave_done = true; break; }
            }
        }
    }

    const size_t plane = (size_t)H * W;
    if constexpr (M4) {
        mfma_drain4(acc4);
#pragma unroll
        for (int ch = 0; ch < NV; ++ch) acc[ch] = acc4[ch];
        D = acc4[3];
    }
    if (MFMA) {
        mfma_drain(accA, accB);
        // D[ch][pix]: lane l, register r holds channel (r&3) + 8 (r>>2) + 4 (l>>5) of wave pixel
        // (l & 31) [accA] / 32 + (l & 31) [accB]; T of those pixels comes from lanes (l&31), 32+(l&31).
        float TA = T, TB = T;
        swap_halves(TA, TB);
        const int pa = lane & 31, pb = 32 + (lane & 31);
        const int xa = qx + (pa & 7), ya = qy + (pa >> 3), xb = qx + (pb & 7), yb = qy + (pb >> 3);
        const bool ina = xa < W && ya < H, inb = xb < W && yb < H;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float bgc = c < bg_channels ? bg[c] : 0.0f;
            if (ina) out_color[(size_t)c * plane + (size_t)ya * W + xa] = accA[r] + TA * bgc;
            if (inb) out_color[(size_t)c * plane + (size_t)yb * W + xb] = accB[r] + TB * bgc;
        }
    }
    if (inside) {
        const size_t pix = (size_t)py * W + px;
#pragma unroll
        for (int ch = 0; ch < NV; ++ch) {
            const int c = c0 + NM + ch;
            out_color[(size_t)c * plane + pix] = acc[ch] + T * (c < bg_channels ? bg[c] : 0.0f);
        }
        if (write_aux) {
            out_depth[pix] = D;
            out_alpha[pix] = 1.0f - T;
            final_T[pix] = T;
            n_contrib[pix] = last;
        }
    }
}

// ---- narrow layouts (C <= 4: SplatLoc's own RGB / RGB + key-point frames) ----------------------------------------------------
// A row of <= 4 features is ONE 16-byte piece: it rides the chunk prefetch beside the entry's record — lane l requests the word of
// list entry base + 128 + l, and record + feature row of entry base + 64 + l (the row's address needs that entry's word, which was
// requested a chunk earlier) — and the whole chunk sits in LDS when its candidates are composited: no staging rounds (no rank /
// popcount algebra, no gather whose latency a round of 24 candidates has to absorb).  A single 640x480 frame lasts as long as the walk
// of its LONGEST list by one wave that runs almost alone on its SIMD, where every exposed latency counts: the staged form spent
// 70 us per 1 000 list entries there, this one 65 (tools/lone_wave.py; profiles/r05_ab_probes.txt #10).  Arithmetic and its order are those of composite_fwd_kernel.
#define SR_FWD_PARAMS                                                                                                        \
    int W, int H, int CP4, int c0, int bg_channels, int write_aux, int tiles /*per view*/, int V, int P /*rows per view*/,  \
        const uint32_t *__restrict__ ranges, const uint32_t *__restrict__ ipack /*id | reach bits << 24*/,                  \
        const float4 *__restrict__ irec, const float4 *__restrict__ featp4, const float *__restrict__ bg, WinOut outs,      \
        float *__restrict__ final_T_all, uint32_t *__restrict__ n_contrib_all,                                              \
        float *__restrict__ ckpt_all /*split launches (common.h): [V][SPLIT_PARTS_MAX][NC + 2][H * W] segment records, else null*/, \
        const uint32_t *__restrict__ nparts /*split launches: parts of every (view, tile) list (written with the launch order), or null: SPLIT_PARTS*/
#define SR_FWD_ARGS W, H, CP4, c0, bg_channels, write_aux, tiles, V, P, ranges, ipack, irec, featp4, bg, outs, final_T_all, n_contrib_all, ckpt_all, nparts

// one quadrant of global tile `gtile` (= view * tiles + tile) by ONE wave; s_rec0 / s_rec1 / s_fq: WAVE + 1 float4 each, the wave's own
template <int NC>
__device__ __forceinline__ void narrow_quadrant(SR_FWD_PARAMS, int gtile, int quad, int lane, float4* __restrict__ s_rec0,
                                                float4* __restrict__ s_rec1, float4* __restrict__ s_fq)
{
    static_assert(NC >= 1 && NC <= 4, "one 16-byte piece per feature row");
    // chunk entry j at [1 + j]; entry 0 is the ABSENT candidate (opacity 0: alpha = 0, never live), read through j1 = -1 by the
    // last pair of an odd chunk
    const int gx = (W + TILE - 1) / TILE;
    const int view = (V == 1) ? 0 : gtile / tiles;      // wave-uniform (scalar)
    const int tile = gtile - view * tiles;
    const uint32_t row0 = (uint32_t)view * (uint32_t)P;  // the view's first row: feature row of row g is g - row0
    float* __restrict__ out_color = outs.color[view];
    float* __restrict__ out_depth = outs.depth[view];
    float* __restrict__ out_alpha = outs.alpha[view];
    float* __restrict__ final_T = final_T_all + (size_t)view * H * W;
    uint32_t* __restrict__ n_contrib = n_contrib_all + (size_t)view * H * W;
    const int qx = (tile % gx) * TILE + (quad & 1) * 8, qy = (tile / gx) * TILE + (quad >> 1) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
    const uint32_t beg = ranges[2 * gtile], end = ranges[2 * gtile + 1];

    if (lane == 0) {
        s_rec0[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_rec1[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_fq[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bool active = inside;  // pixel still accumulating
    float T = 1.0f, D = 0.0f;
    float acc[NC];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) acc[ch] = 0.0f;
    uint32_t last = 0;

    // in flight: pw2 = the word of entry base + 128 + lane; a0 / a1 / af = record and feature row of entry base + 64 + lane, whose
    // word pw arrived a chunk ago.  The words stay RAW (id | reach bits << 24): bit and id are extracted where they are consumed.
    const uint32_t fo = (uint32_t)(c0 >> 2);
    uint32_t pw = 0, pw2 = 0;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, af = a0;
    auto fetch_word = [&](uint32_t base, uint32_t& w_) {
        w_ = 0u;
        if (base + (uint32_t)lane < end) w_ = ipack[base + (uint32_t)lane];
    };
    auto fetch_entry = [&](uint32_t base, uint32_t w_) {
        if (base + (uint32_t)lane < end) {
            const uint32_t j = base + (uint32_t)lane;
            a0 = irec[2 * (size_t)j];
            a1 = irec[2 * (size_t)j + 1];
            // (only entries that reach this quadrant are ever read back; ids < 2^24: checked on the host)
            if ((w_ >> (24 + quad)) & 1u) af = featp4[(size_t)(__umul24((w_ & 0xFFFFFFu) - row0, (uint32_t)CP4) + fo)];
        }
    };
    fetch_word(beg, pw);
    fetch_word(beg + WAVE, pw2);
    fetch_entry(beg, pw);

    bool wave_done = __builtin_amdgcn_ballot_w64(active) == 0;
    // split launches (common.h): the backward runs SPLIT_PARTS waves per quadrant, one per part of the list.  Segment record k
    // = { T in front of the segment's first entry, the colours and the depth the segment ALONE contributes } — segment sums
    // are accumulated from zero, so they are accurate relative to their own (transmittance-scaled) magnitude; the backward
    // rebuilds "what lies behind a boundary" from the sums of the later segments (a prefix C_k subtracted from the image
    // would carry the image's rounding, 1e-7 |C|, into a remainder of size T_k |C|).
    const int np = (ckpt_all != nullptr && nparts != nullptr) ? (int)nparts[gtile] : SPLIT_PARTS;   // (wave-uniform)
    const uint32_t part = ckpt_all != nullptr ? split_part(end - beg, (uint32_t)np) : 0u;
    uint32_t ck_at = part ? beg + part : 0xFFFFFFFFu;
    int ck_k = 0;
    float sacc[NC], sD = 0.0f;     // the current segment's own sums (split launches only)
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) sacc[ch] = 0.0f;
    auto store_segment = [&](int k, bool with_next_T) {
        if (inside) {
            const size_t pl = (size_t)H * W;
            float* ck = ckpt_all + ((size_t)view * SPLIT_PARTS_MAX + k) * (NC + 2) * pl + (size_t)py * W + px;
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) ck[(size_t)(1 + ch) * pl] = sacc[ch];
            ck[(size_t)(1 + NC) * pl] = sD;
            if (with_next_T) ck[(size_t)(NC + 2) * pl] = T;     // plane 0 of record k + 1
        }
    };
#pragma unroll 1
    for (uint32_t base = beg; base < end && !wave_done; base += WAVE) {
        if (base == ck_at) {   // a segment ends in front of this entry
            store_segment(ck_k, true);
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) sacc[ch] = 0.0f;
            sD = 0.0f;
            ++ck_k;
            ck_at = ck_k < np - 1 ? ck_at + part : 0xFFFFFFFFu;
        }
        uint64_t cand = __builtin_amdgcn_ballot_w64((pw >> (24 + quad)) & 1u);
        if (cand != 0) {
            __builtin_amdgcn_wave_barrier();
            s_rec0[1 + lane] = a0;
            s_rec1[1 + lane] = a1;
            s_fq[1 + lane] = af;
            __builtin_amdgcn_wave_barrier();
        }
        // the next chunk's entries and the word of the one behind it: issued now, consumed after this chunk has been composited
        pw = pw2;
        fetch_entry(base + WAVE, pw);
        fetch_word(base + 2 * WAVE, pw2);
        const uint32_t idx = base - beg;
#pragma unroll 1
        while (cand != 0) {
            // ---- front to back, two at a time ----
            const int j0 = first_bit(cand);
            cand &= cand - 1;
            const int j1 = first_bit(cand);   // -1 in the last pair of an odd chunk: the absent candidate
            cand &= cand - 1;                 // (an empty mask stays empty)
            const float4 p0 = s_rec0[1 + j0], q0 = s_rec1[1 + j0], f0 = s_fq[1 + j0];
            const float4 p1 = s_rec0[1 + j1], q1 = s_rec1[1 + j1], f1 = s_fq[1 + j1];
            const float dx0 = p0.x - fx, dy0 = p0.y - fy, dx1 = p1.x - fx, dy1 = p1.y - fy;
            const float pw0 = gauss_log2(q0, dx0, dy0), pw1 = gauss_log2(q1, dx1, dy1);  // log2 of the weight
            // (alpha is tested BEFORE the min with 0.99, like the backward: a NaN from an overflowed power is a miss)
            const float ar0 = q0.w * exp2_core(pw0), ar1 = q1.w * exp2_core(pw1);
            const float al0 = fminf(ALPHA_MAX, ar0), al1 = fminf(ALPHA_MAX, ar1);
            // Gaussian 0
            const bool live0 = active && pw0 <= 0.0f && ar0 >= ALPHA_MIN;
            const float tT0 = transmit(T, al0);
            const bool hit0 = live0 && tT0 >= T_EPS;
            const bool act1 = active && !(live0 && !hit0);  // transmittance exhausted: pixel finished
            const float w0 = hit0 ? al0 * T : 0.0f;
            const float T1 = hit0 ? tT0 : T;
            // Gaussian 1 (the absent candidate has alpha = 0)
            const bool live1 = act1 && pw1 <= 0.0f && ar1 >= ALPHA_MIN;
            const float tT1 = transmit(T1, al1);
            const bool hit1 = live1 && tT1 >= T_EPS;
            active = act1 && !(live1 && !hit1);
            const float w1 = hit1 ? al1 * T1 : 0.0f;
            T = hit1 ? tT1 : T1;
            last = hit1 ? idx + (uint32_t)j1 + 1u : (hit0 ? idx + (uint32_t)j0 + 1u : last);
            const float fv0[4] = {f0.x, f0.y, f0.z, f0.w}, fv1[4] = {f1.x, f1.y, f1.z, f1.w};
            // two FMAs in list order per channel
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) acc[ch] = fmaf(fv1[ch], w1, fmaf(fv0[ch], w0, acc[ch]));
            D = fmaf(p1.z, w1, fmaf(p0.z, w0, D));
            if (part) {     // (wave-uniform) the segment's own sums: never read by this kernel's images
#pragma unroll
                for (int ch = 0; ch < NC; ++ch) sacc[ch] = fmaf(fv1[ch], w1, fmaf(fv0[ch], w0, sacc[ch]));
                sD = fmaf(p1.z, w1, fmaf(p0.z, w0, sD));
            }
            if (__builtin_amdgcn_ballot_w64(active) == 0) { wave_done = true; break; }
        }
    }

    if (part) {
        store_segment(ck_k, false);
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) sacc[ch] = 0.0f;
        sD = 0.0f;
        for (int k = ck_k + 1; k < np; ++k) store_segment(k, false);   // segments the wave never reached contribute nothing
    }
    if (inside) {
        const size_t plane = (size_t)H * W;
        const size_t pix = (size_t)py * W + px;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            const int c = c0 + ch;
            out_color[(size_t)c * plane + pix] = acc[ch] + T * (c < bg_channels ? bg[c] : 0.0f);
        }
        if (write_aux) {
            out_depth[pix] = D;
            out_alpha[pix] = 1.0f - T;
            final_T[pix] = T;
            n_contrib[pix] = last;
        }
    }
}

template <int NC>
__global__ void __launch_bounds__(WAVE, 7)   // the budget is 72 registers / 7 waves per SIMD (pinned by tests/test_codegen_budget.py); ROCm 7.2 lands
                                             // at 64 / 8 by itself, while a bound of 8 makes the allocator spill the four-channel instantiation
composite_fwd_narrow_kernel(SR_FWD_PARAMS, const uint32_t* __restrict__ tile_order /*launch order (binning.hip), or null*/)
{
    __shared__ __attribute__((aligned(16))) float4 s_rec0[WAVE + 1];
    __shared__ __attribute__((aligned(16))) float4 s_rec1[WAVE + 1];
    __shared__ __attribute__((aligned(16))) float4 s_fq[WAVE + 1];
    int gtile, quad;   // global tile = view * tiles + tile: the grid covers the V views of the window
    quadrant_of_block(blockIdx.x, V * tiles, (W + TILE - 1) / TILE, gtile, quad, tile_order);
    if (gtile >= V * tiles) return;
    narrow_quadrant<NC>(SR_FWD_ARGS, gtile, quad, (int)threadIdx.x, s_rec0, s_rec1, s_fq);
}

// ---- narrow layouts on a frame that does not fill the machine: a TEAM of four waves per quadrant -------------------------------
// One 640x480 frame is 4 800 quadrant lists on a machine with 1 024 SIMDs: the kernel lasts as long as the walk of its longest
// list, by a wave that is alone on its SIMD.  Such a wave issues ONE instruction every four cycles whatever its kind, and the
// one-wave kernel spends 80 of them per pair of candidates (65 us per 1 000 list entries, tools/lone_wave.py) while three
// quarters of the machine idle.  The work of a pair falls into three parts of which only the middle one is sequential:
//   A  (two waves, alternate pairs)  Gaussian weight, 2^x, the alpha tests  ->  alpha, or -1 for a miss, in LDS        42 instr.
//   B  the transmittance chain: live / hit / finished, T, the last contributor  ->  the weight w = alpha T (0: no hit),
//      in place of alpha                                                                                                 ~30
//   C  colours, depth and the split launches' segment sums from w                                                      ~27
// The waves walk the list on their own (own prefetch, own copy of what they need of a chunk) in steps of <= TEAM_STEP
// candidates, B one step behind A and C one behind B, through three rotating buffers of TEAM_STEP x 64 floats, and meet at one
// barrier per step.  Every role keeps its candidates' operands SLOT-ordered in LDS (scattered once per step by the chunk's
// lanes), so that the unrolled pair loops address them with immediates: no bit scans, no address arithmetic per pair.
// Same operations on the same operands in the same order as the one-wave kernel: images, depth, alpha, final_T, n_contrib and
// the segment records are bit-identical (tests/test_gpu_edge_cases.py::test_team_forward_changes_nothing).
#ifndef SR_FWD_TEAM_STEP
#define SR_FWD_TEAM_STEP 32   // candidates per pipeline step (a multiple of 4)
#endif
constexpr int TEAM_STEP = SR_FWD_TEAM_STEP;
static_assert(TEAM_STEP % 4 == 0 && TEAM_STEP >= 4 && TEAM_STEP <= WAVE, "two A waves take alternate pairs");
struct TeamLds {
    float buf[3][TEAM_STEP][WAVE];                            // step s lives in buffer s % 3: alpha (A) -> w (B) -> read by C
    __attribute__((aligned(16))) float4 aq[2][TEAM_STEP];     // A0 / A1: pre-scaled conic, opacity of the step's slots
    __attribute__((aligned(16))) float2 axy[2][TEAM_STEP];    //          centre
    __attribute__((aligned(16))) uint32_t bj[TEAM_STEP];      // B: list index + 1 of the slot's candidate
    __attribute__((aligned(16))) float4 cf[TEAM_STEP];        // C: feature row
    __attribute__((aligned(16))) float cz[TEAM_STEP];         //    depth
    float T[WAVE];                                            // B -> C: the final transmittance
    int fin;                                                  // the last iteration (-1: B has not finished yet)
};
// one quadrant of global tile `gtile` by the FOUR waves of the workgroup (every wave of it must call this)
template <int NC>
__device__ __forceinline__ void team_quadrant(SR_FWD_PARAMS, int gtile, int quad, TeamLds& L)
{
    static_assert(NC >= 1 && NC <= 4, "one 16-byte piece per feature row");
    constexpr int STEP = TEAM_STEP;
    float (&s_buf)[3][STEP][WAVE] = L.buf;
    float4 (&s_aq)[2][STEP] = L.aq;
    float2 (&s_axy)[2][STEP] = L.axy;
    uint32_t (&s_bj)[STEP] = L.bj;
    float4 (&s_cf)[STEP] = L.cf;
    float (&s_cz)[STEP] = L.cz;
    float (&s_T)[WAVE] = L.T;
    int& s_fin = L.fin;

    const int gx = (W + TILE - 1) / TILE;
    const int view = (V == 1) ? 0 : gtile / tiles;
    const int tile = gtile - view * tiles;
    const uint32_t row0 = (uint32_t)view * (uint32_t)P;
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // 0, 1: A   2: B   3: C
    const int lane = threadIdx.x & (WAVE - 1);
    const int qx = (tile % gx) * TILE + (quad & 1) * 8, qy = (tile / gx) * TILE + (quad >> 1) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);
    const bool inside = px < W && py < H;
    const uint32_t beg = ranges[2 * gtile], end = ranges[2 * gtile + 1];
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (WAVE - lane));
    const size_t plane = (size_t)H * W;
    const size_t pix = (size_t)py * W + px;
    const int np = (ckpt_all != nullptr && nparts != nullptr) ? (int)nparts[gtile] : SPLIT_PARTS;   // (workgroup-uniform)
    const uint32_t part = ckpt_all != nullptr ? split_part(end - beg, (uint32_t)np) : 0u;

    if (threadIdx.x == 0) s_fin = -1;
    __syncthreads();

    // ---- every role: the walk over the chunks and the steps of a chunk ----
    // in flight: pw = the word of entry next + lane, pw2 = of entry next + 64 + lane (C: a feature row's address needs its word a
    // chunk ahead of the row), r0 / r1 = what the role keeps of entry next + lane; c0r / c1r = the same of the chunk being handed out
    uint32_t next = beg;          // first entry of the chunk in flight
    uint32_t base = beg;          // first entry of the chunk being handed out
    uint64_t cand = 0;            // its candidates not yet handed out
    uint32_t pw = 0, pw2 = 0;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, c0r = r0, c1r = r0;
    const uint32_t fo = (uint32_t)(c0 >> 2);
    auto fetch_word = [&](uint32_t at, uint32_t& w_) {
        w_ = 0u;
        if (at + (uint32_t)lane < end) w_ = ipack[at + (uint32_t)lane];
    };
    auto fetch_entry = [&](auto role_c, uint32_t at, uint32_t w_) {
        constexpr int R = decltype(role_c)::value;     // 0: A, 2: B, 3: C
        if (R != 2 && at + (uint32_t)lane < end) {
            const uint32_t j = at + (uint32_t)lane;
            if (R == 0) {
                r0 = irec[2 * (size_t)j];          // (x, y, depth, -)
                r1 = irec[2 * (size_t)j + 1];      // conic, opacity
            } else {
                r0.x = reinterpret_cast<const float*>(irec)[8 * (size_t)j + 2];   // depth
                if ((w_ >> (24 + quad)) & 1u) r1 = featp4[(size_t)(__umul24((w_ & 0xFFFFFFu) - row0, (uint32_t)CP4) + fo)];
            }
        }
    };
    // on_chunk(): what a role does when its walk enters the chunk at `base` (B, C: the split launches' segment boundary)
    auto next_step = [&](auto role_c, auto&& on_chunk) -> uint64_t {
        while (cand == 0) {
            if (next >= end) return 0ull;
            base = next;
            on_chunk();
            cand = __builtin_amdgcn_ballot_w64((pw >> (24 + quad)) & 1u);
            c0r = r0;
            c1r = r1;
            next += WAVE;
            pw = pw2;
            fetch_entry(role_c, next, pw);
            fetch_word(next + WAVE, pw2);
        }
        uint64_t m = cand;
        if (__popcll(cand) > STEP) {
            const bool mine = ((cand >> lane) & 1ull) && __popcll(cand & lt_mask) < STEP;
            m = __builtin_amdgcn_ballot_w64(mine);
        }
        cand &= ~m;
        return m;
    };
    fetch_word(beg, pw);
    fetch_word(beg + WAVE, pw2);

    if (role < 2) {
        // ================================ A: alpha of step `it` into buffer it % 3 ================================
        fetch_entry(std::integral_constant<int, 0>{}, beg, pw);
        const float fx = (float)px, fy = (float)py;
        float4* __restrict__ my_q = &s_aq[role][0];
        float2* __restrict__ my_xy = &s_axy[role][0];
        bool a_end = false;
#pragma unroll 1
        for (int it = 0;; ++it) {
            const uint64_t m = (a_end || s_fin >= 0) ? 0ull : next_step(std::integral_constant<int, 0>{}, [] {});
            if (m == 0) a_end = true;
            const int n = (int)__popcll(m);
            if (n) {
                __builtin_amdgcn_wave_barrier();
                if ((m >> lane) & 1ull) {
                    const int slot = (int)__popcll(m & lt_mask);
                    my_q[slot] = c1r;
                    my_xy[slot] = make_float2(c0r.x, c0r.y);
                }
                if (lane == 0 && (n & 1)) {    // the ABSENT candidate of an odd step's last pair: opacity 0, never live
                    my_q[n] = make_float4(0.f, 0.f, 0.f, 0.f);
                    my_xy[n] = make_float2(0.f, 0.f);
                }
                __builtin_amdgcn_wave_barrier();
                float* __restrict__ buf = &s_buf[it % 3][2 * role][lane];
                const float4* __restrict__ qs = my_q + 2 * role;
                const float2* __restrict__ cs = my_xy + 2 * role;
#pragma unroll
                for (int i = 0; i < STEP / 4; ++i) {          // pair 2 i + role: slots 4 i + 2 role, + 1
                    if (4 * i + 2 * role >= n) break;
                    const float4 q0 = qs[4 * i], q1 = qs[4 * i + 1];
                    const float4 cc = *reinterpret_cast<const float4*>(&cs[4 * i]);   // both centres
                    const float dx0 = cc.x - fx, dy0 = cc.y - fy, dx1 = cc.z - fx, dy1 = cc.w - fy;
                    const float pw0 = gauss_log2(q0, dx0, dy0), pw1 = gauss_log2(q1, dx1, dy1);
                    // (alpha is tested BEFORE the min with 0.99: a NaN from an overflowed power is a miss)
                    const float ar0 = q0.w * exp2_core(pw0), ar1 = q1.w * exp2_core(pw1);
                    const float al0 = fminf(ALPHA_MAX, ar0), al1 = fminf(ALPHA_MAX, ar1);
                    buf[(4 * i) * WAVE] = (pw0 <= 0.0f && ar0 >= ALPHA_MIN) ? al0 : -1.0f;
                    buf[(4 * i + 1) * WAVE] = (pw1 <= 0.0f && ar1 >= ALPHA_MIN) ? al1 : -1.0f;
                }
            }
            __syncthreads();
            if (s_fin >= 0 && it >= s_fin) break;
        }
        return;
    }

    if (role == 2) {
        // ================================ B: the transmittance chain of step it - 1, in place ================================
        bool active = inside;
        float T = 1.0f;
        uint32_t last = 0;
        uint32_t ck_at = part ? beg + part : 0xFFFFFFFFu;
        int ck_k = 0;
        auto boundary = [&] {
            if (base == ck_at) {   // a segment ends in front of this entry: plane 0 of record k + 1 = T in front of it
                if (inside) ckpt_all[((size_t)view * SPLIT_PARTS_MAX + ck_k) * (NC + 2) * plane + pix + (size_t)(NC + 2) * plane] = T;
                ++ck_k;
                ck_at = ck_k < np - 1 ? ck_at + part : 0xFFFFFFFFu;
            }
        };
        bool finished = false;
        int fin_it = __builtin_amdgcn_ballot_w64(active) == 0 ? 0 : -1;    // the iteration to announce as the last one
#pragma unroll 1
        for (int it = 0;; ++it) {
            if (it > 0 && fin_it < 0) {
                const uint64_t m = next_step(std::integral_constant<int, 2>{}, boundary);
                const int n = (int)__popcll(m);
                if (n == 0) fin_it = it;                        // the list is exhausted: C's last step is this iteration's
                else {
                    __builtin_amdgcn_wave_barrier();
                    if ((m >> lane) & 1ull) s_bj[__popcll(m & lt_mask)] = base - beg + (uint32_t)lane + 1u;
                    __builtin_amdgcn_wave_barrier();
                    float* __restrict__ buf = &s_buf[(it - 1) % 3][0][lane];
#pragma unroll
                    for (int i = 0; i < STEP / 2; ++i) {      // slots 2 i, 2 i + 1 (the absent candidate of an odd step: A left -1)
                        if (2 * i >= n) break;
                        const float al0 = buf[(2 * i) * WAVE], al1 = buf[(2 * i + 1) * WAVE];
                        const uint2 jj = *reinterpret_cast<const uint2*>(&s_bj[2 * i]);
                        // Gaussian 0
                        const bool live0 = active && al0 >= 0.0f;
                        const float tT0 = transmit(T, al0);
                        const bool hit0 = live0 && tT0 >= T_EPS;
                        const bool act1 = active && !(live0 && !hit0);  // transmittance exhausted: pixel finished
                        const float w0 = hit0 ? al0 * T : 0.0f;
                        const float T1 = hit0 ? tT0 : T;
                        // Gaussian 1
                        const bool live1 = act1 && al1 >= 0.0f;
                        const float tT1 = transmit(T1, al1);
                        const bool hit1 = live1 && tT1 >= T_EPS;
                        active = act1 && !(live1 && !hit1);
                        const float w1 = hit1 ? al1 * T1 : 0.0f;
                        T = hit1 ? tT1 : T1;
                        last = hit1 ? jj.y : (hit0 ? jj.x : last);
                        buf[(2 * i) * WAVE] = w0;
                        buf[(2 * i + 1) * WAVE] = w1;
                    }
                    // (a step is always finished: C reads every slot of it as a weight)
                    if (__builtin_amdgcn_ballot_w64(active) == 0) fin_it = it + 1;   // C composites this step in the next iteration
                }
            }
            if (fin_it >= 0 && !finished) {
                finished = true;
                s_T[lane] = T;
                if (lane == 0) s_fin = fin_it;
            }
            __syncthreads();
            if (fin_it >= 0 && it >= fin_it) break;
        }
        if (inside && write_aux) {
            outs.alpha[view][pix] = 1.0f - T;
            (final_T_all + (size_t)view * plane)[pix] = T;
            (n_contrib_all + (size_t)view * plane)[pix] = last;
        }
        return;
    }

    // ================================ C: colours, depth and segment sums of step it - 2 ================================
    fetch_entry(std::integral_constant<int, 3>{}, beg, pw);
    float D = 0.0f;
    float acc[NC];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) acc[ch] = 0.0f;
    uint32_t ck_at = part ? beg + part : 0xFFFFFFFFu;
    int ck_k = 0;
    float sacc[NC], sD = 0.0f;     // the current segment's own sums (split launches only; common.h)
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) sacc[ch] = 0.0f;
    auto store_sums = [&](int k) {
        if (inside) {
            float* ck = ckpt_all + ((size_t)view * SPLIT_PARTS_MAX + k) * (NC + 2) * plane + pix;
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) ck[(size_t)(1 + ch) * plane] = sacc[ch];
            ck[(size_t)(1 + NC) * plane] = sD;
        }
    };
    auto boundary = [&] {
        if (base == ck_at) {   // a segment ends in front of this entry
            store_sums(ck_k);
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) sacc[ch] = 0.0f;
            sD = 0.0f;
            ++ck_k;
            ck_at = ck_k < np - 1 ? ck_at + part : 0xFFFFFFFFu;
        }
    };
#pragma unroll 1
    for (int it = 0;; ++it) {
        if (it > 1) {
            const uint64_t m = next_step(std::integral_constant<int, 3>{}, boundary);
            const int n = (int)__popcll(m);
            if (n) {
                __builtin_amdgcn_wave_barrier();
                if ((m >> lane) & 1ull) {
                    const int slot = (int)__popcll(m & lt_mask);
                    s_cf[slot] = c1r;
                    s_cz[slot] = c0r.x;
                }
                if (lane == 0 && (n & 1)) {    // the absent candidate's row: finite, whatever the LDS held (its weight is 0)
                    s_cf[n] = make_float4(0.f, 0.f, 0.f, 0.f);
                    s_cz[n] = 0.0f;
                }
                __builtin_amdgcn_wave_barrier();
                const float* __restrict__ buf = &s_buf[(it - 2) % 3][0][lane];
#pragma unroll
                for (int i = 0; i < STEP / 2; ++i) {
                    if (2 * i >= n) break;
                    const float w0 = buf[(2 * i) * WAVE], w1 = buf[(2 * i + 1) * WAVE];
                    const float4 f0 = s_cf[2 * i], f1 = s_cf[2 * i + 1];
                    const float2 zz = *reinterpret_cast<const float2*>(&s_cz[2 * i]);
                    const float fv0[4] = {f0.x, f0.y, f0.z, f0.w}, fv1[4] = {f1.x, f1.y, f1.z, f1.w};
                    // two FMAs in list order per channel
#pragma unroll
                    for (int ch = 0; ch < NC; ++ch) acc[ch] = fmaf(fv1[ch], w1, fmaf(fv0[ch], w0, acc[ch]));
                    D = fmaf(zz.y, w1, fmaf(zz.x, w0, D));
                    if (part) {     // (wave-uniform) the segment's own sums
#pragma unroll
                        for (int ch = 0; ch < NC; ++ch) sacc[ch] = fmaf(fv1[ch], w1, fmaf(fv0[ch], w0, sacc[ch]));
                        sD = fmaf(zz.y, w1, fmaf(zz.x, w0, sD));
                    }
                }
            }
        }
        __syncthreads();
        if (s_fin >= 0 && it >= s_fin) break;
    }
    if (part) {
        store_sums(ck_k);
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) sacc[ch] = 0.0f;
        sD = 0.0f;
        for (int k = ck_k + 1; k < np; ++k) store_sums(k);   // segments the walk never reached contribute nothing
    }
    if (inside) {
        const float T = s_T[lane];
        float* __restrict__ out_color = outs.color[view];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            const int c = c0 + ch;
            out_color[(size_t)c * plane + pix] = acc[ch] + T * (c < bg_channels ? bg[c] : 0.0f);
        }
        if (write_aux) outs.depth[view][pix] = D;
    }
}

// ---- the narrow-layout launch of a frame that does not fill the machine ----
// Workgroups of four waves over the launch order (longest list first, binning.hip): the TEAM_MAX longest lists — if they stand out:
// at least TEAM_MIN_LIST entries and 1.25 x the list a quarter down the order (or any such list of a frame with fewer quadrants
// than the machine has SIMDs) — are walked by a team per quadrant (four workgroups
// per tile, launched first), every other tile by one workgroup with one wave per quadrant (narrow_quadrant; its chunk copies live in
// the team's buffers).  A team costs ~1.7 x the SIMD time of a lone wave and finishes its list 2.3 x sooner: worth it for the lists
// the whole launch waits for, a loss for all of them (a uniform cloud: forward 69 -> 86 us with a team for every list).
#ifndef SR_FWD_TEAM_MAX
#define SR_FWD_TEAM_MAX 128   // (32: room 97.6 / Replica scale 180.1 us of forward; 64: 84.7 / 157.7; 128: 82.3 / 147.1; 256: 81.5 / 141.0, uniform cloud 68.1 -> 71.4)
#endif
#ifndef SR_FWD_TEAM_MIN_LIST
#define SR_FWD_TEAM_MIN_LIST 512
#endif
constexpr int TEAM_MAX = SR_FWD_TEAM_MAX, TEAM_MIN_LIST = SR_FWD_TEAM_MIN_LIST;
__device__ __forceinline__ bool team_list(int r, int T, const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ order, int force)
{
    if (r >= TEAM_MAX || r >= T) return false;
    if (force) return true;     // (test hook: every list of the first TEAM_MAX, whatever its length)
    const uint32_t t = order[r];
    const uint32_t len = ranges[2 * t + 1] - ranges[2 * t];
    if (len < (uint32_t)TEAM_MIN_LIST) return false;
    if (4 * T <= 1024) return true;          // fewer quadrants than SIMDs: every wave is alone anyway
    const uint32_t q = order[T >> 2];
    return 4u * len >= 5u * (ranges[2 * q + 1] - ranges[2 * q]);
}
static_assert(4 * 3 * (WAVE + 1) * sizeof(float4) <= sizeof(float) * 3 * TEAM_STEP * WAVE, "the four waves' chunk copies fit the team's buffers");
template <int NC>
__global__ void __launch_bounds__(4 * WAVE, 5)   // 27 KB of LDS: five workgroups = five waves per SIMD
composite_fwd_mixed_kernel(SR_FWD_PARAMS, const uint32_t* __restrict__ tile_order /*launch order (binning.hip): required*/, int force)
{
    __shared__ __attribute__((aligned(16))) TeamLds L;
    const int T = V * tiles;
    const int b = (int)blockIdx.x;
    if (b < 4 * TEAM_MAX) {
        const int r = b >> 2;
        if (!team_list(r, T, ranges, tile_order, force)) return;
        team_quadrant<NC>(SR_FWD_ARGS, (int)tile_order[r], b & 3, L);
        return;
    }
    const int r = b - 4 * TEAM_MAX;
    if (r >= T || team_list(r, T, ranges, tile_order, force)) return;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float4* mine = reinterpret_cast<float4*>(&L.buf[0][0][0]) + w * 3 * (WAVE + 1);
    narrow_quadrant<NC>(SR_FWD_ARGS, (int)tile_order[r], w, (int)(threadIdx.x & (WAVE - 1)), mine, mine + (WAVE + 1), mine + 2 * (WAVE + 1));
}

__global__ void debug_exp2_kernel(int64_t n, const float* __restrict__ x, float* __restrict__ y)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = exp2_shared(x[i]);
}

// test hook (splatraster_debug_poison_lds): 64 KB workgroups, two per compute unit resident at a time, several rounds of them
__global__ void __launch_bounds__(256) poison_lds_kernel(uint32_t pattern, uint32_t* __restrict__ sink)
{
    __shared__ uint32_t s_all[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) s_all[i] = pattern;
    __syncthreads();
    // keep the stores: the buffer is read back through a data-dependent index
    const uint32_t v = s_all[(pattern + threadIdx.x * 61u) & 16383u];
    if (v != pattern) sink[0] = v;
    __builtin_amdgcn_s_sleep(64);   // stay resident for a moment so that the blocks spread over every compute unit
}

int launch_poison_lds(uint32_t pattern, hipStream_t stream)
{
    static uint32_t* sink = nullptr;
    if (!sink && hipMalloc(&sink, 256) != hipSuccess) return SPLATRASTER_ERR_HIP;
    hipLaunchKernelGGL(poison_lds_kernel, dim3(256 * 2 * 4), dim3(256), 0, stream, pattern, sink);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

int launch_debug_exp2(int64_t n, const float* x, float* y, hipStream_t stream)
{
    hipLaunchKernelGGL(debug_exp2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, x, y);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

struct FwdLaunch {
    int P, V;
    const WinOut* outs;
    float* ckpt;   // non-null: split launch (the backward runs SPLIT_PARTS waves per quadrant: the forward records every quarter's own sums)
    bool team;     // narrow layouts on a frame that does not fill the machine: composite_fwd_mixed_kernel (teams of four waves for the longest lists)
};

static int g_fwd_team = -1;   // -1: automatic (the split launches' condition), 0: never, 1: every narrow launch that has a launch order, 2: and a team for each of its first TEAM_MAX lists
void set_fwd_team(int mode) { g_fwd_team = mode; }

template <int NC>
static int launch_one(const splatraster_settings& s, int c0, int write_aux, const GeomView& g,
                      const BinView& b, const ImgView& im, const float* featp, int feat_stride,
                      const float* bg, const FwdLaunch& L, hipStream_t stream)
{
    (void)g;
    const int gx = (s.image_width + TILE - 1) / TILE, gy = (s.image_height + TILE - 1) / TILE;
    const int tiles = gx * gy;
    const unsigned blocks = quadrant_blocks(L.V * tiles, gx);  // 4 quadrants per (view, tile) (+ padding of the id space)
    if constexpr (NC <= 4) {
        if (L.team && use_tile_order(L.V, tiles)) {
            hipLaunchKernelGGL(composite_fwd_mixed_kernel<NC>, dim3((unsigned)(L.V * tiles + 4 * TEAM_MAX)), dim3(4 * WAVE), 0, stream,
                               s.image_width, s.image_height, padded_channels(feat_stride) / 4, c0, s.bg_channels, write_aux, tiles, L.V, L.P,
                               b.ranges, b.ipack, b.irec, reinterpret_cast<const float4*>(featp), bg, *L.outs, im.final_T, im.n_contrib,
                               (c0 == 0 && write_aux) ? L.ckpt : nullptr, b.nparts, b.tile_order, g_fwd_team == 2 ? 1 : 0);
        } else {
            hipLaunchKernelGGL(composite_fwd_narrow_kernel<NC>, dim3(blocks), dim3(WAVE), 0, stream, s.image_width,
                               s.image_height, padded_channels(feat_stride) / 4, c0, s.bg_channels, write_aux, tiles, L.V, L.P, b.ranges,
                               b.ipack, b.irec, reinterpret_cast<const float4*>(featp), bg, *L.outs, im.final_T,
                               im.n_contrib, (c0 == 0 && write_aux) ? L.ckpt : nullptr, use_tile_order(L.V, tiles) ? b.nparts : nullptr,
                               use_tile_order(L.V, tiles) ? b.tile_order : nullptr);
        }
    } else {
        hipLaunchKernelGGL(composite_fwd_kernel<NC>, dim3(blocks), dim3(WAVE), 0, stream, s.image_width,
                           s.image_height, padded_channels(feat_stride) / 4, c0, s.bg_channels, write_aux, tiles, L.V, L.P, b.ranges,
                           b.ipack, b.irec, reinterpret_cast<const float4*>(featp), bg, *L.outs, im.final_T,
                           im.n_contrib, use_tile_order(L.V, tiles) ? b.tile_order : nullptr);
    }
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

int launch_composite_fwd(const splatraster_settings& s, int32_t P, int32_t V, int64_t R, const GeomView& g, const BinView& b,
                         const ImgView& im, const float* featp, const float* bg, const WinOut& outs, hipStream_t stream)
{
    (void)R;
    const int C = s.channels;
    const int tiles_v = ((s.image_width + TILE - 1) / TILE) * ((s.image_height + TILE - 1) / TILE);
    const bool team = g_fwd_team < 0 ? split_lists(C, V, tiles_v) : (g_fwd_team != 0 && C <= 4);
    const FwdLaunch L{P, V, &outs, split_lists(C, V, tiles_v) ? b.ckpt : nullptr, team};
    int c0 = 0, aux = 1, st = SPLATRASTER_OK;
#define SR_FWD_CASE(N)                                                                              \
    case N:                                                                                         \
        return launch_one<N>(s, 0, 1, g, b, im, featp, C, bg, L, stream);
    switch (C) {
        SR_FWD_CASE(1) SR_FWD_CASE(2) SR_FWD_CASE(3) SR_FWD_CASE(4) SR_FWD_CASE(8) SR_FWD_CASE(16)
        SR_FWD_CASE(32) SR_FWD_CASE(35)
        default: break;
    }
#undef SR_FWD_CASE
    // generic channel count: chunked passes (alpha is re-evaluated per chunk)
    while (c0 < C && st == SPLATRASTER_OK) {
        const int left = C - c0;
        if (left >= 32) { st = launch_one<32>(s, c0, aux, g, b, im, featp, C, bg, L, stream); c0 += 32; }
        else if (left >= 16) { st = launch_one<16>(s, c0, aux, g, b, im, featp, C, bg, L, stream); c0 += 16; }
        else if (left >= 8) { st = launch_one<8>(s, c0, aux, g, b, im, featp, C, bg, L, stream); c0 += 8; }
        else if (left >= 4) { st = launch_one<4>(s, c0, aux, g, b, im, featp, C, bg, L, stream); c0 += 4; }
        else if (left == 3) { st = launch_one<3>(s, c0, aux, g, b, im, featp, C, bg, L, stream); c0 += 3; }
        else if (left == 2) { st = launch_one<2>(s, c0, aux, g, b, im, featp, C, bg, L, stream); c0 += 2; }
        else { st = launch_one<1>(s, c0, aux, g, b, im, featp, C, bg, L, stream); c0 += 1; }
        aux = 0;
    }
    return st;
}

}  // namespace sr

