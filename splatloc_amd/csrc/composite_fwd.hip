// composite_fwd.hip — front-to-back alpha compositing of the per-tile depth-sorted lists.
//
// Replaces the render stage of the rasterizer extension behind
// diff_gauss.GaussianRasterizer.forward (gaussian_renderer/__init__.py:117-126; algorithm
// per SURVEY.md §8a "COMPOSITE fwd").  One 256-thread workgroup (4 wave64) per 16x16
// tile; wave w owns the 8x8 pixel quadrant (w&1, w>>1) so that a Gaussian that misses a
// quadrant is skipped by the whole wave with one ballot.  The tile's list is consumed in
// batches staged in LDS: 32-byte projected records + the feature rows (4*C bytes each),
// fetched with coalesced global loads and read back as wave-uniform (broadcast) LDS reads.
// VALU/LDS-bound; no MFMA (no dense contraction in this form).
#include "composite_common.h"

#ifndef SR_FWD_ABLATE
#define SR_FWD_ABLATE 0  // perf ablation switch (tools/ablate.py); 0 = product
#endif

namespace sr {

constexpr int CF_THREADS = 256;

template <int NC>
struct FwdCfg {
    static constexpr int NCP = (NC + 3) & ~3;           // LDS row stride (floats), 16-B aligned rows
    static constexpr int BATCH = (NC > 16) ? 128 : 256; // Gaussians staged per round
};

template <int NC>
__global__ void __launch_bounds__(CF_THREADS)
composite_fwd_kernel(int W, int H, int C_total, int c0, int bg_channels, int write_aux,
                     const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                     const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                     const float* __restrict__ feat, const float* __restrict__ bg,
                     float* __restrict__ out_color, float* __restrict__ out_depth,
                     float* __restrict__ out_alpha, float* __restrict__ final_T,
                     uint32_t* __restrict__ n_contrib)
{
    constexpr int NCP = FwdCfg<NC>::NCP;
    constexpr int BATCH = FwdCfg<NC>::BATCH;
    __shared__ __attribute__((aligned(16))) float4 s_rec0[BATCH];
    __shared__ __attribute__((aligned(16))) float4 s_rec1[BATCH];
    __shared__ __attribute__((aligned(16))) float s_feat[BATCH * NCP];
    __shared__ uint32_t s_id[BATCH];
    __shared__ uint64_t s_cand[4][BATCH / WAVE];  // per quadrant: candidate bitmask of the batch
    __shared__ uint8_t s_any[BATCH];              // row reaches at least one quadrant

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int gx = (W + TILE - 1) / TILE;
    const int tile = blockIdx.y * gx + blockIdx.x;
    const int px = blockIdx.x * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = blockIdx.y * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;

    const uint32_t beg = ranges[2 * tile], end = ranges[2 * tile + 1];
    int todo = (int)(end - beg);

    bool done = !inside;
    float T = 1.0f, D = 0.0f;
    float acc[NC];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) acc[ch] = 0.0f;
    uint32_t contributor = 0, last = 0;

    for (uint32_t base = beg; todo > 0; base += BATCH, todo -= BATCH) {
        if (__syncthreads_count(done) == CF_THREADS) break;
        const int nb = todo < BATCH ? todo : BATCH;
        // ---- stage ids + records ----
        unsigned m4 = 0u;
        if (tid < nb) {
            const uint32_t g = point_list[base + tid];
            const float4 a0 = rec0[g], a1 = rec1[g];
            s_id[tid] = g;
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
        // ---- stage feature rows: consecutive threads walk consecutive floats of a row ----
#if SR_FWD_ABLATE != 3
        for (int e = tid; e < nb * NC; e += CF_THREADS) {
            const int row = e / NC, ch = e - row * NC;
            if (s_any[row]) s_feat[row * NCP + ch] = feat[(size_t)s_id[row] * C_total + c0 + ch];
        }
#endif
        __syncthreads();
        bool wave_done = __all(done);
#pragma unroll 1
        for (int k = 0; k < BATCH / WAVE && !wave_done; ++k) {
            uint64_t cand = uniform_u64(s_cand[wave][k]);
            while (cand) {
                const int j = k * WAVE + __builtin_ctzll(cand);
                cand &= cand - 1;
                const float4 r0 = s_rec0[j];
                const float4 r1 = s_rec1[j];
                const float dx = r0.x - fx, dy = r0.y - fy;
                const float power = -0.5f * (r1.x * dx * dx + r1.z * dy * dy) - r1.y * dx * dy;
                const float alpha = fminf(ALPHA_MAX, r1.w * __expf(power));
                const float test_T = T * (1.0f - alpha);
                const bool live = !done && power <= 0.0f && alpha >= ALPHA_MIN;
                const bool hit = live && test_T >= T_EPS;
                if (live && !hit) done = true;  // transmittance exhausted: pixel finished
                if (__any(live && !hit) && __all(done)) { wave_done = true; break; }
                if (__any(hit)) {
                    const float w = hit ? alpha * T : 0.0f;
                    const float* f = &s_feat[j * NCP];
#if SR_FWD_ABLATE != 2
#pragma unroll
                    for (int ch = 0; ch < NC; ++ch) acc[ch] += f[ch] * w;
#else
                    acc[0] += f[0] * w;
#endif
                    D += r0.z * w;
                    if (hit) {
                        T = test_T;
                        last = contributor + (uint32_t)j + 1u;
                    }
                }
            }
        }
        contributor += (uint32_t)nb;
    }

#if SR_FWD_ABLATE == 1
    {
        float keep = 0.f;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) keep += acc[ch];
        if (keep == 123.456f) out_color[0] = keep;
    }
#endif
    if (inside) {
        const size_t pix = (size_t)py * W + px;
        const size_t plane = (size_t)H * W;
#if SR_FWD_ABLATE != 1
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            const int c = c0 + ch;
            out_color[(size_t)c * plane + pix] = acc[ch] + T * (c < bg_channels ? bg[c] : 0.0f);
        }
#endif
        if (write_aux) {
            out_depth[pix] = D;
            out_alpha[pix] = 1.0f - T;
            final_T[pix] = T;
            n_contrib[pix] = last;
        }
    }
}

template <int NC>
static int launch_one(const splatraster_settings& s, int c0, int write_aux, const GeomView& g,
                      const BinView& b, const ImgView& im, const float* feat, int feat_stride,
                      const float* bg, float* out_color, float* out_depth, float* out_alpha,
                      hipStream_t stream)
{
    const int gx = (s.image_width + TILE - 1) / TILE, gy = (s.image_height + TILE - 1) / TILE;
    hipLaunchKernelGGL(composite_fwd_kernel<NC>, dim3(gx, gy), dim3(CF_THREADS), 0, stream, s.image_width,
                       s.image_height, feat_stride, c0, s.bg_channels, write_aux, b.ranges, b.point_list,
                       g.rec0, g.rec1, feat, bg, out_color, out_depth, out_alpha, im.final_T, im.n_contrib);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

int launch_composite_fwd(const splatraster_settings& s, int64_t R, const GeomView& g, const BinView& b,
                         const ImgView& im, const float* feat, const float* bg, float* out_color,
                         float* out_depth, float* out_alpha, hipStream_t stream)
{
    (void)R;
    const int C = s.channels;
    int c0 = 0, aux = 1, st = SPLATRASTER_OK;
#define SR_FWD_CASE(N)                                                                              \
    case N:                                                                                         \
        return launch_one<N>(s, 0, 1, g, b, im, feat, C, bg, out_color, out_depth, out_alpha, stream);
    switch (C) {
        SR_FWD_CASE(1) SR_FWD_CASE(2) SR_FWD_CASE(3) SR_FWD_CASE(4) SR_FWD_CASE(8) SR_FWD_CASE(16)
        SR_FWD_CASE(32) SR_FWD_CASE(35)
        default: break;
    }
#undef SR_FWD_CASE
    // generic channel count: chunked passes (alpha is re-evaluated per chunk)
    while (c0 < C && st == SPLATRASTER_OK) {
        const int left = C - c0;
        if (left >= 32) { st = launch_one<32>(s, c0, aux, g, b, im, feat, C, bg, out_color, out_depth, out_alpha, stream); c0 += 32; }
        else if (left >= 16) { st = launch_one<16>(s, c0, aux, g, b, im, feat, C, bg, out_color, out_depth, out_alpha, stream); c0 += 16; }
        else if (left >= 8) { st = launch_one<8>(s, c0, aux, g, b, im, feat, C, bg, out_color, out_depth, out_alpha, stream); c0 += 8; }
        else if (left >= 4) { st = launch_one<4>(s, c0, aux, g, b, im, feat, C, bg, out_color, out_depth, out_alpha, stream); c0 += 4; }
        else if (left == 3) { st = launch_one<3>(s, c0, aux, g, b, im, feat, C, bg, out_color, out_depth, out_alpha, stream); c0 += 3; }
        else if (left == 2) { st = launch_one<2>(s, c0, aux, g, b, im, feat, C, bg, out_color, out_depth, out_alpha, stream); c0 += 2; }
        else { st = launch_one<1>(s, c0, aux, g, b, im, feat, C, bg, out_color, out_depth, out_alpha, stream); c0 += 1; }
        aux = 0;
    }
    return st;
}

}  // namespace sr
