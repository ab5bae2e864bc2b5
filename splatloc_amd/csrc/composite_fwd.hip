// composite_fwd.hip — front-to-back alpha compositing of the per-tile depth-sorted lists.
//
// Replaces the render stage of the rasterizer extension behind
// diff_gauss.GaussianRasterizer.forward (gaussian_renderer/__init__.py:117-126; algorithm
// per SURVEY.md §8a "COMPOSITE fwd").  One 256-thread workgroup (4 wave64) per 16x16
// tile; wave w owns the 8x8 pixel quadrant (w&1, w>>1) so that a Gaussian that misses a
// quadrant is skipped by the whole wave with one ballot.  The tile's list is consumed in
// batches staged in LDS: 32-byte projected records + the feature rows (4*C bytes each),
// fetched with coalesced global loads and read back as wave-uniform (broadcast) LDS reads.
// VALU-bound.
//
// NC >= 32: the accumulation of the first 32 channels, out[pix][ch] += w[pix][g] * F[g][ch],
// is a [32 ch x 2 g] x [2 g x 32 pix] product per pair of contributing Gaussians and runs on
// the matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fmaf chain = the sequential
// front-to-back sum).  A operand: one conflict-free ds_read_b32 of the two staged feature
// rows; B operand: the two weight registers of the pair after ONE v_permlane32_swap
// (lanes 0-31 <- pixels 0-31 / 32-63 of Gaussian 0, lanes 32-63 <- of Gaussian 1).
#include "composite_common.h"

namespace sr {

constexpr int CF_THREADS = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

template <int NC>
struct FwdCfg {
    static constexpr bool MFMA = NC >= 32;
    static constexpr int NM = MFMA ? 32 : 0;            // channels accumulated on the matrix pipe
    static constexpr int NV = NC - NM;                  // channels accumulated with VALU FMAs
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
    constexpr bool MFMA = FwdCfg<NC>::MFMA;
    constexpr int NM = FwdCfg<NC>::NM, NV = FwdCfg<NC>::NV;
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
    float acc[NV > 0 ? NV : 1];  // VALU-accumulated channels (all of them when NC < 32)
#pragma unroll
    for (int ch = 0; ch < NV; ++ch) acc[ch] = 0.0f;
    // matrix-pipe accumulators D[ch][pix]: accA = pixels (lanes) 0-31 of the wave, accB = 32-63
    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accA[r] = 0.0f; accB[r] = 0.0f; }
    float w_pend = 0.0f;  // weights of a contributing Gaussian waiting for its pair partner
    int j_pend = -1;      // its row in the staged batch (wave-uniform), -1 = none
    auto mfma_pair = [&](int j0, float w0, int j1, float w1) {
        const float a = s_feat[((lane >> 5) ? j1 : j0) * NCP + (lane & 31)];  // A[i = ch][k = g]
        swap_halves(w0, w1);  // w0 -> B for pixels 0-31, w1 -> B for pixels 32-63
        accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w0, accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w1, accB, 0, 0, 0);
    };
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
        for (int e = tid; e < nb * NC; e += CF_THREADS) {
            const int row = e / NC, ch = e - row * NC;
            if (s_any[row]) s_feat[row * NCP + ch] = feat[(size_t)s_id[row] * C_total + c0 + ch];
        }
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
                    const float* f = &s_feat[j * NCP + NM];
#pragma unroll
                    for (int ch = 0; ch < NV; ++ch) acc[ch] += f[ch] * w;
                    D += r0.z * w;
                    if (MFMA) {
                        if (j_pend < 0) {
                            w_pend = w;
                            j_pend = j;
                        } else {
                            mfma_pair(j_pend, w_pend, j, w);
                            j_pend = -1;
                        }
                    }
                    if (hit) {
                        T = test_T;
                        last = contributor + (uint32_t)j + 1u;
                    }
                }
            }
        }
        // the staged rows are about to be overwritten: retire an unpaired Gaussian (partner weight 0)
        if (MFMA && j_pend >= 0) {
            mfma_pair(j_pend, w_pend, j_pend, 0.0f);
            j_pend = -1;
        }
        contributor += (uint32_t)nb;
    }

    if (MFMA) {
        // D[ch][pix]: lane l, register r holds channel (r&3) + 8 (r>>2) + 4 (l>>5) of wave pixel
        // (l & 31) [accA] / 32 + (l & 31) [accB]; T of those pixels comes from lanes (l&31), 32+(l&31).
        float TA = T, TB = T;
        swap_halves(TA, TB);
        const int qx = blockIdx.x * TILE + (wave & 1) * 8, qy = blockIdx.y * TILE + (wave >> 1) * 8;
        const int pa = lane & 31, pb = 32 + (lane & 31);
        const int xa = qx + (pa & 7), ya = qy + (pa >> 3), xb = qx + (pb & 7), yb = qy + (pb >> 3);
        const bool ina = xa < W && ya < H, inb = xb < W && yb < H;
        const size_t plane = (size_t)H * W;
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
        const size_t plane = (size_t)H * W;
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
