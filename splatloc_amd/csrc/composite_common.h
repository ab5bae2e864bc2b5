// composite_common.h — pieces shared by the forward and backward compositing kernels.
#pragma once
#include "common.h"

namespace sr {

// Conservative reach test of one projected Gaussian against the four 8x8 pixel quadrants of
// a 16x16 tile (bit q set <=> the Gaussian MAY reach alpha >= 1/255 at some pixel centre of
// quadrant q = (qx | qy << 1)).
//
// alpha = o * exp(power) >= 1/255  <=>  q(d) := A dx^2 + 2 B dx dy + C dy^2 <= 2 ln(255 o),
// and the minimum of the convex quadratic q over the quadrant's box of pixel centres is
// 0 if the centre lies inside, else attained on one of the four edges at the clamped 1-D
// minimiser.  The continuous box minimum is <= the minimum over pixel centres, and an
// absolute + relative slack far above fp32 / v_exp_f32 error is added, so a cleared bit
// proves that no pixel of the quadrant would pass the alpha >= 1/255 test: skipping the
// Gaussian for that wave leaves the image bit-identical.
__device__ __forceinline__ unsigned quadrant_reach_mask(const float4 r0, const float4 r1, float tile_x0,
                                                        float tile_y0)
{
    const float A = r1.x, B = r1.y, C = r1.z;
    const float lim0 = 2.0f * __logf(255.0f * r1.w);
    const float lim = lim0 + 0.02f + 1e-4f * fabsf(lim0);
    if (!(lim > 0.0f)) return 0u;
    const float nBrA = -B * __frcp_rn(A), nBrC = -B * __frcp_rn(C);
    unsigned mask = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float bx0 = tile_x0 + (float)((q & 1) * 8), by0 = tile_y0 + (float)((q >> 1) * 8);
        // d = mu - pixel: u in [u0, u1], v in [v0, v1]
        const float u0 = r0.x - (bx0 + 7.0f), u1 = r0.x - bx0;
        const float v0 = r0.y - (by0 + 7.0f), v1 = r0.y - by0;
        float qmin;
        if (u0 <= 0.0f && u1 >= 0.0f && v0 <= 0.0f && v1 >= 0.0f) {
            qmin = 0.0f;
        } else {
            qmin = 3.0e38f;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float ue = e ? u1 : u0;
                const float vs = fminf(v1, fmaxf(v0, nBrC * ue));
                qmin = fminf(qmin, A * ue * ue + 2.0f * B * ue * vs + C * vs * vs);
                const float ve = e ? v1 : v0;
                const float us = fminf(u1, fmaxf(u0, nBrA * ve));
                qmin = fminf(qmin, A * us * us + 2.0f * B * us * ve + C * ve * ve);
            }
        }
        if (qmin <= lim) mask |= 1u << q;
    }
    return mask;
}

// Same test for ONE quadrant whose first pixel centre is (bx0, by0).
__device__ __forceinline__ bool quadrant_reach(const float4 r0, const float4 r1, float bx0, float by0)
{
    const float A = r1.x, B = r1.y, C = r1.z;
    const float lim0 = 2.0f * __logf(255.0f * r1.w);
    const float lim = lim0 + 0.02f + 1e-4f * fabsf(lim0);
    const float u0 = r0.x - (bx0 + 7.0f), u1 = r0.x - bx0;
    const float v0 = r0.y - (by0 + 7.0f), v1 = r0.y - by0;
    float qmin = 0.0f;
    if (!(u0 <= 0.0f && u1 >= 0.0f && v0 <= 0.0f && v1 >= 0.0f)) {
        const float nBrA = -B * __frcp_rn(A), nBrC = -B * __frcp_rn(C);
        qmin = 3.0e38f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float ue = e ? u1 : u0;
            const float vs = fminf(v1, fmaxf(v0, nBrC * ue));
            qmin = fminf(qmin, A * ue * ue + 2.0f * B * ue * vs + C * vs * vs);
            const float ve = e ? v1 : v0;
            const float us = fminf(u1, fmaxf(u0, nBrA * ve));
            qmin = fminf(qmin, A * us * us + 2.0f * B * us * ve + C * ve * ve);
        }
    }
    return (lim > 0.0f) && (qmin <= lim);
}

// The per-instance payload stores the conic pre-scaled for the compositing kernels:
//   q = (-0.5 log2(e) A, -log2(e) B, -0.5 log2(e) C, opacity)
// so that log2 of the Gaussian weight is a 5-operation polynomial and the weight itself one
// v_exp_f32 (no separate -0.5 and log2(e) multiplies per (pixel, Gaussian)).  Forward and backward
// call the SAME function with explicit fmaf: their alpha values are bit-identical.
constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ float4 payload_conic(const float4& conic_opacity)
{
    return make_float4(-0.5f * LOG2E * conic_opacity.x, -LOG2E * conic_opacity.y, -0.5f * LOG2E * conic_opacity.z,
                       conic_opacity.w);
}
__device__ __forceinline__ float gauss_log2(const float4& q, float dx, float dy)
{
    return fmaf(dx, fmaf(q.y, dy, q.x * dx), (q.z * dy) * dy);
}
// 2^x as ONE explicit operation sequence that the CPU oracle restates instruction for instruction
// (its orc_exp2; the product never links it), so that alpha, the transmittance chain and therefore every
// alpha >= 1/255 / T < 1e-4 decision, n_contrib and final_T are BIT-IDENTICAL on both sides:
//   n = rint(x) (ties to even: v_rndne_f32), f = x - n in [-0.5, 0.5] (exact),
//   p = degree-5 minimax polynomial of 2^f (Horner, fmaf), result = ldexp(p, n) (v_ldexp_f32).
// c0 = 1 exactly (2^n is exact at the integers); max relative error 1.7e-7 (1.4 ulp) — the accuracy
// class of a libm expf.
// v_exp_f32 (the hardware approximation, 1 quarter-rate instruction instead of 9 full-rate ones) is
// not reproducible off the GPU; -DSR_EXP2_HW selects it for A/B timing only (cost: DESIGN.md §6.1).
constexpr float EXP2_C0 = 1.0f, EXP2_C1 = 0.6931470036506653f, EXP2_C2 = 0.24022242426872253f,
                EXP2_C3 = 0.05550733581185341f, EXP2_C4 = 0.009671512991189957f, EXP2_C5 = 0.001326472731307149f;
// The arithmetic without the argument clamp, for the compositing kernels (backward -2 % on S2 / S1 in round 3; the forward
// follows in round 4, two instructions per pair fewer, +0.3 % on S2): there the clamp's only job (an argument of -inf or
// NaN, reachable through an overflowing conic * d^2) is done for free by testing alpha BEFORE the min with 0.99 — NaN fails
// the >= 1/255 test, the Gaussian is a miss exactly as with the clamped 2^-200 = 0 — and by zeroing G, not dL/dalpha, for a
// miss in the backward.  For every finite argument the value is the clamped function's, bit for bit: below -200 both flush to
// 0 in v_ldexp_f32, above 0 the pixel is skipped (power > 0) whatever comes out.  v_cvt_i32_f32 saturates and maps NaN to 0;
// it is written as asm because an out-of-range float -> int conversion is undefined in C++.
__device__ __forceinline__ float exp2_core(float x)
{
#ifdef SR_EXP2_HW
    return __builtin_amdgcn_exp2f(x);
#else
    const float n = __builtin_rintf(x);
    const float f = x - n;
    float p = fmaf(EXP2_C5, f, EXP2_C4);
    p = fmaf(p, f, EXP2_C3);
    p = fmaf(p, f, EXP2_C2);
    p = fmaf(p, f, EXP2_C1);
    p = fmaf(p, f, EXP2_C0);
    int e;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(e) : "v"(n));
    return __builtin_amdgcn_ldexpf(p, e);
#endif
}
__device__ __forceinline__ float exp2_shared(float x)
{
    // clamp first, exactly like orc_exp2: x = -inf would make f = NaN (and min(0.99, o * NaN) = 0.99, a spurious
    // opaque hit), and (int)n of an out-of-range float is undefined.  2^-200 flushes to 0 in ldexp either way.
    return exp2_core(fminf(fmaxf(x, -200.0f), 200.0f));
}
// T (1 - alpha) with two roundings, never a contracted fma: the oracle's transmittance chain
__device__ __forceinline__ float transmit(float T, float alpha)
{
#pragma clang fp contract(off)
    const float om = 1.0f - alpha;
    return T * om;
}

// Block id -> (global tile, quadrant) for the wave-per-quadrant kernels.  Observed dispatch: block b -> XCD b % 8, in
// order of b within an XCD.  The four quadrants of a tile get ids with the same (id % 8): they run on the same
// XCD and share its L2 for the tile's list; consecutive tiles are interleaved over the XCDs, which balances the
// load and spreads the atomics of one Gaussian over time (mapping every XCD to its own contiguous band of tiles
// was measured slower: forward 0.476 vs 0.419 ms, backward 0.982 vs 0.853 ms on S2).  Speed only, never correctness.
// `order` (binning.hip tile_order_kernel; null = identity) turns the tile slot into the tile that runs there: longest lists first.
__device__ __forceinline__ void quadrant_of_block(unsigned id, int tiles, int gx, int& tile, int& quad,
                                                  const uint32_t* __restrict__ order = nullptr)
{
    (void)gx;
    tile = (int)((id >> 5) * 8u + (id & 7u));
    quad = (int)((id >> 3) & 3u);
    if (order && tile < tiles) tile = (int)order[tile];
}
// blocks to launch so that every tile's four quadrants get an id
__host__ __device__ static inline unsigned quadrant_blocks(int tiles, int gx)
{
    (void)gx;
    return (unsigned)((tiles + 7) / 8) * 32u;
}

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// ---- DPP helpers (wave64 = 4 rows of 16 lanes) -----------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float v)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}

// ---- packed butterfly reduction -----------------------------------------------------------
// Reduces K per-lane values over the 64 lanes of a wave and leaves total k in lane
// bitreverse6(k): every stage halves the lane span of each value AND merges two registers
// into one, so the whole reduction costs ~2.2 VALU per value instead of 6 DPP adds + a
// readlane/select gather per value.
//   stage 1 (lane bit 5): v_permlane32_swap + add      (2 instr per pair)
//   stage 2 (lane bit 4): v_permlane16_swap + add      (2 instr per pair)
//   stage 3 (lane bit 3): two bank-masked v_add_f32_dpp row_ror:8       (fold_lane_bit3)
//   stage 4 (lane bit 2): two bank-masked v_add_f32_dpp row_ror:12 / 4   (fold_lane_bit2)
//   stage 5 (lane bit 1): select / select / add quad_perm[2,3,0,1]
//   stage 6 (lane bit 0): select / select / add quad_perm[1,0,3,2]
__device__ __forceinline__ float as_f(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned as_u(float f) { return __builtin_bit_cast(unsigned, f); }

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_mov_old(float old, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                 __builtin_bit_cast(int, v), CTRL, 0xf,
                                                                 BANK_MASK, false));
}

// One stage of the packed butterfly on lane bit 3 / bit 2: lanes with the bit clear end with x + x[partner], lanes with it
// set with y + y[partner] (partner = lane ^ 8 / lane ^ 4).  Both bits are constant inside a DPP bank (4 lanes), so each half is
// ONE bank-masked v_add_f32_dpp on the untouched inputs — 2 instructions per output instead of select / select / (moves) / add
// (3 for bit 3, 6 for bit 2).  s_nop 1 = the two wait states a DPP read of a just-written VGPR needs (the compiler cannot see
// into the asm).
#ifndef SR_BUTTERFLY_ASM
#define SR_BUTTERFLY_ASM 1   // 0 = the select / move formulation through compiler builtins (A/B)
#endif
__device__ __forceinline__ float fold_lane_bit3(float x, float y)
{
#if !SR_BUTTERFLY_ASM
    const bool lo = (__lane_id() & 8) == 0;
    const float keep = lo ? x : y, send = lo ? y : x;
    return keep + dpp_get<0x128, 0xf>(send);
#endif
    float d;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc"
        : "=&v"(d)
        : "v"(x), "v"(y));
    return d;
}
__device__ __forceinline__ float fold_lane_bit2(float x, float y)
{
#if !SR_BUTTERFLY_ASM
    const bool lo = (__lane_id() & 4) == 0;
    const float keep = lo ? x : y, send = lo ? y : x;
    float t = dpp_mov_old<0x12C, 0x5>(0.0f, send);
    t = dpp_mov_old<0x124, 0xA>(t, send);
    return keep + t;
#endif
    float d;   // banks 0, 2 (bit clear) read lane + 4 = row_ror:12; banks 1, 3 read lane - 4 = row_ror:4
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %1, %1 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xa"
        : "=&v"(d)
        : "v"(x), "v"(y));
    return d;
}

template <int K>
__device__ __forceinline__ float wave_reduce_pack(const float (&v)[K], int lane)
{
    static_assert(K >= 1 && K <= 64, "at most 64 values per wave");
    constexpr int N1 = (K + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2,
                  N5 = (N4 + 1) / 2, N6 = (N5 + 1) / 2;
    static_assert(N6 == 1, "");
    float a[N1];
#pragma unroll
    for (int m = 0; m < N1; ++m) {
        const float x = v[2 * m];
        const float y = (2 * m + 1 < K) ? v[2 * m + 1] : 0.0f;
        const auto r = __builtin_amdgcn_permlane32_swap(as_u(x), as_u(y), false, false);
        a[m] = as_f(r[0]) + as_f(r[1]);
    }
    float b[N2];
#pragma unroll
    for (int m = 0; m < N2; ++m) {
        const float x = a[2 * m];
        const float y = (2 * m + 1 < N1) ? a[2 * m + 1] : 0.0f;
        const auto r = __builtin_amdgcn_permlane16_swap(as_u(x), as_u(y), false, false);
        b[m] = as_f(r[0]) + as_f(r[1]);
    }
    float c[N3];
#pragma unroll
    // A stage's last register may have no partner value (odd count): the lanes that would hold it then belong to value
    // indices >= K, which no caller consumes (and which never feed a lane of an existing index: later stages pair lanes
    // that agree in all higher bits) — there x + x[partner] in every lane is one instruction.
    for (int m = 0; m < N3; ++m)
        c[m] = (2 * m + 1 < N2) ? fold_lane_bit3(b[2 * m], b[2 * m + 1 < N2 ? 2 * m + 1 : 0]) : b[2 * m] + dpp_get<0x128, 0xf>(b[2 * m]);
    float d[N4];
#pragma unroll
    for (int m = 0; m < N4; ++m)
        d[m] = (2 * m + 1 < N3) ? fold_lane_bit2(c[2 * m], c[2 * m + 1 < N3 ? 2 * m + 1 : 0]) : fold_lane_bit2(c[2 * m], c[2 * m]);
    float e[N5];
    {
        const bool lo = (lane & 2) == 0;
#pragma unroll
        for (int m = 0; m < N5; ++m) {
            const float x = d[2 * m];
            if (2 * m + 1 < N4) {
                const float y = d[2 * m + 1 < N4 ? 2 * m + 1 : 0];
                const float keep = lo ? x : y, send = lo ? y : x;
                e[m] = keep + dpp_get<0x4E, 0xf>(send);  // quad_perm [2,3,0,1]
            } else {
                e[m] = x + dpp_get<0x4E, 0xf>(x);
            }
        }
    }
    const float x6 = e[0];
    if (N5 > 1) {
        const bool lo1 = (lane & 1) == 0;
        const float y6 = e[N5 > 1 ? 1 : 0];
        const float keep6 = lo1 ? x6 : y6, send6 = lo1 ? y6 : x6;
        return keep6 + dpp_get<0xB1, 0xf>(send6);  // quad_perm [1,0,3,2]
    }
    return x6 + dpp_get<0xB1, 0xf>(x6);
}

// The first three stages of wave_reduce_pack only: K <= 8 per-lane values are summed over lane bits 5, 4, 3 (lanes that
// differ in bits 0..2 stay separate); lane l ends with the total of value (l >> 5 & 1) + 2 (l >> 4 & 1) + 4 (l >> 3 & 1).
template <int K>
__device__ __forceinline__ float wave_reduce_hi3(const float (&v)[K], int lane)
{
    static_assert(K >= 1 && K <= 8, "at most 8 values");
    constexpr int N1 = (K + 1) / 2, N2 = (N1 + 1) / 2;
    float a[N1];
#pragma unroll
    for (int m = 0; m < N1; ++m) {
        const float x = v[2 * m];
        const float y = (2 * m + 1 < K) ? v[2 * m + 1] : 0.0f;
        const auto r = __builtin_amdgcn_permlane32_swap(as_u(x), as_u(y), false, false);
        a[m] = as_f(r[0]) + as_f(r[1]);
    }
    float b[N2];
#pragma unroll
    for (int m = 0; m < N2; ++m) {
        const float x = a[2 * m];
        const float y = (2 * m + 1 < N1) ? a[2 * m + 1] : 0.0f;
        const auto r = __builtin_amdgcn_permlane16_swap(as_u(x), as_u(y), false, false);
        b[m] = as_f(r[0]) + as_f(r[1]);
    }
    (void)lane;
    if (N2 > 1) return fold_lane_bit3(b[0], b[N2 > 1 ? 1 : 0]);
    return b[0] + dpp_get<0x128, 0xf>(b[0]);
}

}  // namespace sr
