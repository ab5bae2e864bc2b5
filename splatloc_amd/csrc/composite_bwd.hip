// composite_bwd.hip — backward of the alpha compositing (SURVEY.md §8a "COMPOSITE bwd").
//
// Replaces the render-backward stage of the rasterizer extension whose backward is
// triggered at train_gaussians.py:229 / :286.
//
// Formulation (differs from the lineage's back-to-front accum_rec recurrences, same
// derivative): with w_i = alpha_i T_i, q_i = f_i . g + z_i g_D  (g = dL/dcolor at the pixel)
//     S_total = out_color . g + out_depth g_D - T_final g_A
//     S_i     = S_total - sum_{j<=i} w_j q_j            (suffix sum incl. background/alpha terms)
//     dL/dalpha_i = T_i q_i - S_i / (1 - alpha_i)
// so the pass runs FRONT-TO-BACK exactly like the forward (same arithmetic for alpha and
// T, no division T/(1-alpha) to undo transmittance) and needs one dot product per
// (pixel, Gaussian) instead of three per-channel recurrences.  Everything is linear in
// (g, g_D, g_A), so channels can be split over several launches (generic C).
//
// One 256-thread workgroup per 16x16 tile, wave w = 8x8 quadrant w.  Per (wave, Gaussian)
// the per-pixel partials must be summed over the wave's 64 pixels:
//   * 7 geometric partials (+ channels beyond the first 32): packed butterfly reduction
//     (permlane32/16 swap + DPP, ~2.2 VALU per value), one wave-wide float atomic;
//   * NC >= 32: dL/dfeature[g][ch] = sum_pix w[pix][g] * dL/dcolor[pix][ch] for the first 32
//     channels is a dense [32 g x 64 pix] x [64 pix x 32 ch] contraction per group of 32
//     contributing Gaussians: the weights are parked in LDS (one ds_write per step) and the
//     contraction runs on the matrix pipe with v_mfma_f32_32x32x2_f32 — exact fp32 (a k-ordered
//     fmaf chain), concurrent with the VALU work of the other resident waves.
#include "composite_common.h"

#ifndef SR_BWD_MFMA_BATCH
#define SR_BWD_MFMA_BATCH 128  // A/B on S2: 128 + 4 waves/SIMD = 1.54 ms vs 64/3 = 1.68 ms
#endif
#ifndef SR_BWD_MINW
#define SR_BWD_MINW 4  // waves per SIMD the register allocator must allow (5 spills: 5.9 ms)
#endif

namespace sr {

constexpr int CB_THREADS = 256;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NC>
struct BwdCfg {
    static constexpr bool MFMA = NC >= 32;
    static constexpr int NM = MFMA ? 32 : 0;   // channels reduced on the matrix pipe
    static constexpr int NV = NC - NM;         // channels reduced with the packed butterfly
    static constexpr int KRED = NV + 7;
    static constexpr int NCP = (NC + 3) & ~3;
    static constexpr int BATCH = MFMA ? SR_BWD_MFMA_BATCH : ((NC > 16) ? 128 : 256);
    static constexpr int GROUP = 16;           // Gaussians per MFMA flush (M of v_mfma_f32_16x16x4_f32)
    static constexpr int WS = 17;              // LDS row stride of the per-wave weight panel [64 pix][GROUP]
};

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
//   stage 3 (lane bit 3): select / select / add row_ror:8
//   stage 4 (lane bit 2): select / select / two bank-masked row_ror movs / add
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
    {
        const bool lo = (lane & 8) == 0;
#pragma unroll
        for (int m = 0; m < N3; ++m) {
            const float x = b[2 * m];
            const float y = (2 * m + 1 < N2) ? b[2 * m + 1] : 0.0f;
            const float keep = lo ? x : y, send = lo ? y : x;
            c[m] = keep + dpp_get<0x128, 0xf>(send);  // row_ror:8
        }
    }
    float d[N4];
    {
        const bool lo = (lane & 4) == 0;
#pragma unroll
        for (int m = 0; m < N4; ++m) {
            const float x = c[2 * m];
            const float y = (2 * m + 1 < N3) ? c[2 * m + 1] : 0.0f;
            const float keep = lo ? x : y, send = lo ? y : x;
            // partner = lane ^ 4: banks 0,2 read lane+4 (row_ror:12), banks 1,3 read lane-4 (row_ror:4)
            float t = dpp_mov_old<0x12C, 0x5>(0.0f, send);
            t = dpp_mov_old<0x124, 0xA>(t, send);
            d[m] = keep + t;
        }
    }
    float e[N5];
    {
        const bool lo = (lane & 2) == 0;
#pragma unroll
        for (int m = 0; m < N5; ++m) {
            const float x = d[2 * m];
            const float y = (2 * m + 1 < N4) ? d[2 * m + 1] : 0.0f;
            const float keep = lo ? x : y, send = lo ? y : x;
            e[m] = keep + dpp_get<0x4E, 0xf>(send);  // quad_perm [2,3,0,1]
        }
    }
    const bool lo1 = (lane & 1) == 0;
    const float x6 = e[0];
    const float y6 = (N5 > 1) ? e[N5 > 1 ? 1 : 0] : 0.0f;
    const float keep6 = lo1 ? x6 : y6, send6 = lo1 ? y6 : x6;
    return keep6 + dpp_get<0xB1, 0xf>(send6);  // quad_perm [1,0,3,2]
}

template <int NC>
__global__ void __launch_bounds__(CB_THREADS, SR_BWD_MINW)
composite_bwd_kernel(int W, int H, int C_total, int c0, int first_pass,
                     const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                     const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                     const float* __restrict__ feat, const float* __restrict__ out_color,
                     const float* __restrict__ out_depth, const float* __restrict__ final_T,
                     const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor,
                     const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha,
                     float* __restrict__ ggrad /*[P,8]*/, float* __restrict__ dcolors /*[P,C_total]*/)
{
    using Cfg = BwdCfg<NC>;
    constexpr int NCP = Cfg::NCP, BATCH = Cfg::BATCH, NM = Cfg::NM, NV = Cfg::NV, KRED = Cfg::KRED;
    constexpr int WS = Cfg::WS, GROUP = Cfg::GROUP;
    constexpr bool MFMA = Cfg::MFMA;
    static_assert(KRED <= WAVE, "at most 57 butterfly-reduced channels per pass");
    __shared__ __attribute__((aligned(16))) float4 s_rec0[BATCH];
    __shared__ __attribute__((aligned(16))) float4 s_rec1[BATCH];
    __shared__ __attribute__((aligned(16))) float s_feat[BATCH * NCP];
    __shared__ uint32_t s_id[BATCH];
    __shared__ uint64_t s_cand[4][BATCH / WAVE];
    __shared__ uint8_t s_any[BATCH];
    __shared__ uint32_t s_max[CB_THREADS / WAVE];
    // matrix-pipe weight panel, one per wave: w[64 pix][GROUP]
    __shared__ float s_w[MFMA ? 4 * WAVE * WS : 1];
    __shared__ uint32_t s_gid[MFMA ? 4 * GROUP : 1];

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int gx = (W + TILE - 1) / TILE;
    const int tile = blockIdx.y * gx + blockIdx.x;
    const int px = blockIdx.x * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = blockIdx.y * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
    const size_t plane = (size_t)H * W;
    const size_t pix = inside ? (size_t)py * W + px : 0;

    const uint32_t beg = ranges[2 * tile], end = ranges[2 * tile + 1];

    // per-pixel constants
    float g[NC];
    float S = 0.0f;  // running suffix sum
    float gD = 0.0f;
    uint32_t last = 0;
    if (inside) {
        last = n_contrib[pix];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            g[ch] = dL_dcolor[(size_t)(c0 + ch) * plane + pix];
            S += out_color[(size_t)(c0 + ch) * plane + pix] * g[ch];
        }
        if (first_pass) {
            gD = dL_ddepth ? dL_ddepth[pix] : 0.0f;
            const float gA = dL_dalpha ? dL_dalpha[pix] : 0.0f;
            S += out_depth[pix] * gD - final_T[pix] * gA;
        }
    } else {
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) g[ch] = 0.0f;
    }
    float* my_w = &s_w[MFMA ? wave * WAVE * WS : 0];
    uint32_t* my_gid = &s_gid[MFMA ? wave * GROUP : 0];
    // B operand of the contraction, kept in registers for the whole tile: for k-step kk and
    // channel half t, lane l holds dL/dcolor[pix = 4 kk + (l >> 4)][ch = 16 t + (l & 15)].
    // Built once by transposing through the (still unused) weight panel.
    float gt[MFMA ? 16 : 1][2];
    if (MFMA) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ch = 0; ch < 16; ++ch) my_w[lane * WS + ch] = g[16 * t + ch];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) gt[kk][t] = my_w[(4 * kk + (lane >> 4)) * WS + (lane & 15)];
        }
        __builtin_amdgcn_wave_barrier();
    }
    // the tile only needs the list up to its deepest contributor
    uint32_t max_last = last;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) max_last = max(max_last, (uint32_t)__shfl_xor((int)max_last, d, WAVE));
    if (lane == 0) s_max[wave] = max_last;
    __syncthreads();
    const uint32_t tile_last = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    const uint32_t wave_last = max_last;
    int todo = (int)min(end - beg, tile_last);

    float T = 1.0f;
    uint32_t contributor = 0;
    const float halfW = 0.5f * (float)W, halfH = 0.5f * (float)H;
    // wave_reduce_pack leaves total k in lane bitreverse6(k)
    const int slot = (int)(__brev((unsigned)lane) >> 26);
    const bool slot_col = slot < NV;
    const bool slot_ok = slot < KRED;
    const int slot_off = slot_col ? (c0 + NM + slot) : (slot - NV);
    int nslot = 0;  // Gaussians parked in the weight panel (wave-uniform)

    // dL/dfeature of the parked Gaussians: D[g][ch] = sum_pix W[pix][g] * G[pix][ch]
    // (v_mfma_f32_16x16x4_f32: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15],
    //  D[row = 4 (l >> 4) + reg][col = l & 15])
    auto flush_panel = [&](int count) {
        __builtin_amdgcn_wave_barrier();
        f32x4 D0 = {0.f, 0.f, 0.f, 0.f}, D1 = {0.f, 0.f, 0.f, 0.f};
        const int row = (lane >> 4) * WS + (lane & 15);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float a = my_w[4 * kk * WS + row];
            D0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, gt[kk][0], D0, 0, 0, 0);
            D1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, gt[kk][1], D1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gs = 4 * (lane >> 4) + r;
            if (gs < count) {
                float* dst = dcolors + (size_t)my_gid[gs] * C_total + c0 + (lane & 15);
                atomicAdd(dst, D0[r]);
                atomicAdd(dst + 16, D1[r]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    for (uint32_t base = beg; todo > 0; base += BATCH, todo -= BATCH) {
        const int nb = todo < BATCH ? todo : BATCH;
        __syncthreads();
        unsigned m4 = 0u;
        if (tid < nb) {
            const uint32_t gi = point_list[base + tid];
            const float4 a0 = rec0[gi], a1 = rec1[gi];
            s_id[tid] = gi;
            s_rec0[tid] = a0;
            s_rec1[tid] = a1;
            m4 = quadrant_reach_mask(a0, a1, (float)(blockIdx.x * TILE), (float)(blockIdx.y * TILE));
            s_any[tid] = (uint8_t)m4;
        }
        if (wave < BATCH / WAVE) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t bal = __ballot((m4 >> q) & 1u);
                if (lane == 0) s_cand[q][wave] = bal;
            }
        }
        __syncthreads();
        for (int e = tid; e < nb * NC; e += CB_THREADS) {
            const int row = e / NC, ch = e - row * NC;
            if (s_any[row]) s_feat[row * NCP + ch] = feat[(size_t)s_id[row] * C_total + c0 + ch];
        }
        __syncthreads();
        const int nw = (int)min((uint32_t)nb, wave_last > contributor ? wave_last - contributor : 0u);
#pragma unroll 1
        for (int k = 0; k * WAVE < nw; ++k) {
            uint64_t cand = uniform_u64(s_cand[wave][k]);
            const int lim = nw - k * WAVE;  // only list positions below the wave's deepest contributor
            if (lim < WAVE) cand &= (1ull << lim) - 1ull;
            while (cand) {
                const int j = k * WAVE + __builtin_ctzll(cand);
                cand &= cand - 1;
                const float4 r0 = s_rec0[j];
                const float4 r1 = s_rec1[j];
                const float dx = r0.x - fx, dy = r0.y - fy;
                const float power = -0.5f * (r1.x * dx * dx + r1.z * dy * dy) - r1.y * dx * dy;
                const float G = __expf(power);
                const float alpha = fminf(ALPHA_MAX, r1.w * G);
                const bool hit = (contributor + (uint32_t)j < last) && power <= 0.0f && alpha >= ALPHA_MIN;
                if (!__any(hit)) continue;
                const float* f = &s_feat[j * NCP];
                const float w = hit ? alpha * T : 0.0f;
                float q = r0.z * gD;
#pragma unroll
                for (int ch = 0; ch < NC; ++ch) q += f[ch] * g[ch];
                const float one_m = 1.0f - alpha;
                float dL_dalpha_i = 0.0f;
                if (hit) {
                    S -= w * q;
                    dL_dalpha_i = T * q - S * __frcp_rn(one_m);
                    T *= one_m;
                }
                const float dL_dG = r1.w * dL_dalpha_i;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * r1.x - gdy * r1.y;
                const float dG_ddely = -gdy * r1.z - gdx * r1.y;
                float red[KRED];
#pragma unroll
                for (int ch = 0; ch < NV; ++ch) red[ch] = w * g[NM + ch];
                red[NV + 0] = dL_dG * dG_ddelx * halfW;
                red[NV + 1] = dL_dG * dG_ddely * halfH;
                red[NV + 2] = -0.5f * gdx * dx * dL_dG;
                red[NV + 3] = -gdx * dy * dL_dG;
                red[NV + 4] = -0.5f * gdy * dy * dL_dG;
                red[NV + 5] = G * dL_dalpha_i;
                red[NV + 6] = w * gD;
                const float outv = wave_reduce_pack<KRED>(red, lane);
                const uint32_t gi = s_id[j];
                float* dst = slot_col ? (dcolors + (size_t)gi * C_total + slot_off)
                                      : (ggrad + (size_t)gi * 8 + slot_off);
                if (slot_ok) atomicAdd(dst, outv);
                if (MFMA) {
                    my_w[lane * WS + nslot] = w;  // park the weights (0 for pixels that miss)
                    if (lane == 0) my_gid[nslot] = gi;
                    if (++nslot == GROUP) {
                        flush_panel(GROUP);
                        nslot = 0;
                    }
                }
            }
        }
        contributor += (uint32_t)nb;
    }
    if (MFMA && nslot > 0) flush_panel(nslot);
}

template <int NC>
static int launch_one_bwd(const splatraster_settings& s, int c0, int first, const GeomView& g,
                          const BinView& b, const ImgView& im, const float* feat, int feat_stride,
                          const float* out_color, const float* out_depth, const float* dL_dcolor,
                          const float* dL_ddepth, const float* dL_dalpha, float* ggrad, float* dcolors,
                          hipStream_t stream)
{
    const int gx = (s.image_width + TILE - 1) / TILE, gy = (s.image_height + TILE - 1) / TILE;
    hipLaunchKernelGGL(composite_bwd_kernel<NC>, dim3(gx, gy), dim3(CB_THREADS), 0, stream, s.image_width,
                       s.image_height, feat_stride, c0, first, b.ranges, b.point_list, g.rec0, g.rec1, feat,
                       out_color, out_depth, im.final_T, im.n_contrib, dL_dcolor, dL_ddepth, dL_dalpha, ggrad,
                       dcolors);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

int launch_composite_bwd(const splatraster_settings& s, int32_t P, int64_t R, const GeomView& g,
                         const BinView& b, const ImgView& im, const float* feat, int feat_stride,
                         const float* out_color, const float* out_depth, const float* dL_dcolor,
                         const float* dL_ddepth, const float* dL_dalpha, float* ggrad, float* dcolors,
                         hipStream_t stream)
{
    (void)P;
    if (R == 0) return SPLATRASTER_OK;
    const int C = s.channels;
#define SR_BWD_ARGS g, b, im, feat, feat_stride, out_color, out_depth, dL_dcolor, dL_ddepth, dL_dalpha, ggrad, dcolors, stream
#define SR_BWD_CASE(N) \
    case N: return launch_one_bwd<N>(s, 0, 1, SR_BWD_ARGS);
    switch (C) {
        SR_BWD_CASE(1) SR_BWD_CASE(2) SR_BWD_CASE(3) SR_BWD_CASE(4) SR_BWD_CASE(8) SR_BWD_CASE(16)
        SR_BWD_CASE(32) SR_BWD_CASE(35)
        default: break;
    }
#undef SR_BWD_CASE
    int c0 = 0, first = 1, st = SPLATRASTER_OK;
    while (c0 < C && st == SPLATRASTER_OK) {
        const int left = C - c0;
        if (left >= 32) { st = launch_one_bwd<32>(s, c0, first, SR_BWD_ARGS); c0 += 32; }
        else if (left >= 16) { st = launch_one_bwd<16>(s, c0, first, SR_BWD_ARGS); c0 += 16; }
        else if (left >= 8) { st = launch_one_bwd<8>(s, c0, first, SR_BWD_ARGS); c0 += 8; }
        else if (left >= 4) { st = launch_one_bwd<4>(s, c0, first, SR_BWD_ARGS); c0 += 4; }
        else if (left == 3) { st = launch_one_bwd<3>(s, c0, first, SR_BWD_ARGS); c0 += 3; }
        else if (left == 2) { st = launch_one_bwd<2>(s, c0, first, SR_BWD_ARGS); c0 += 2; }
        else { st = launch_one_bwd<1>(s, c0, first, SR_BWD_ARGS); c0 += 1; }
        first = 0;
    }
#undef SR_BWD_ARGS
    return st;
}

}  // namespace sr
