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

// Block id -> (tile, quadrant) for the wave-per-quadrant kernels: the four quadrants of a tile
// get ids with the same (id % 8), i.e. they run on the same XCD (observed dispatch: block b ->
// XCD b % 8) and share that XCD's L2 for the tile's list; speed only, never correctness.
__device__ __forceinline__ void quadrant_of_block(unsigned id, int& tile, int& quad)
{
    tile = (int)((id >> 5) * 8u + (id & 7u));
    quad = (int)((id >> 3) & 3u);
}

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

}  // namespace sr
