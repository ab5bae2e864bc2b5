// losses.hip — SplatLoc's per-view mapping loss and its gradient w.r.t. the rendered buffers in
// ONE pass over the pixels.  Replaces the chain of elementwise torch kernels
// (masks, exposure affine, abs, three means, sigmoid, BCE) and the equally long autograd chain of
//   utils/utils.py:55-82     get_loss_mapping / get_loss_mapping_rgbd
//   train_gaussians.py:38-42 get_loss_marker
// as summed per view at train_gaussians.py:217-218:
//   loss = mean |m_rgb x - m_rgb gt| + mean |m_d depth - m_d gt_depth| + mean BCE(sigmoid(marker), kp)
//   x = exp(a) image + b,  m_rgb = sum_c gt[c] > threshold,  m_d = gt_depth > 0.01
// HBM-bound: reads 8 planes + a byte mask, writes 5 planes per pixel.  The five sums (three loss
// terms, dL/da, dL/db) are reduced per block in double and finished by a one-block kernel:
// deterministic, no atomics.
#include "common.h"

namespace sr {

constexpr int LOSS_BLOCK = 256;
constexpr int LOSS_SUMS = 5;  // l1 rgb, l1 depth, bce, d/da, d/db

__device__ __forceinline__ float sgn(float v) { return (float)((v > 0.f) - (v < 0.f)); }

__global__ void __launch_bounds__(LOSS_BLOCK)
mapping_loss_kernel(int HW, const float* __restrict__ image /*3 planes*/, const float* __restrict__ depth,
                    const float* __restrict__ marker, const float* __restrict__ gt_image,
                    const float* __restrict__ gt_depth, const uint8_t* __restrict__ kp, float threshold,
                    const float* __restrict__ exposure /*[2] = a, b or NULL*/, float* __restrict__ g_image,
                    float* __restrict__ g_depth, float* __restrict__ g_marker, double* __restrict__ partial)
{
    const float ea = exposure ? expf(exposure[0]) : 1.0f;
    const float eb = exposure ? exposure[1] : 0.0f;
    const float inv_rgb = 1.0f / (3.0f * (float)HW), inv_n = 1.0f / (float)HW;
    double acc[LOSS_SUMS] = {0, 0, 0, 0, 0};
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
        const float t0 = gt_image[p], t1 = gt_image[HW + p], t2 = gt_image[2 * HW + p];
        const float m = ((t0 + t1) + t2) > threshold ? 1.0f : 0.0f;
        const float gt[3] = {t0, t1, t2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = image[c * HW + p];
            const float x = exposure ? ea * v + eb : v;
            const float diff = x * m - gt[c] * m;
            const float gx = sgn(diff) * m * inv_rgb;
            acc[0] += fabsf(diff);
            acc[3] += (double)(gx * v);
            acc[4] += (double)gx;
            g_image[c * HW + p] = gx * ea;
        }
        const float gd = gt_depth[p];
        const float md = gd > 0.01f ? 1.0f : 0.0f;
        const float dd = depth[p] * md - gd * md;
        acc[1] += fabsf(dd);
        g_depth[p] = sgn(dd) * md * inv_n;
        const float s = 1.0f / (1.0f + expf(-marker[p]));
        const float y = kp[p] ? 1.0f : 0.0f;
        const float lp = fmaxf(logf(s), -100.0f), lq = fmaxf(logf(1.0f - s), -100.0f);
        acc[2] += (double)(-(y * lp + (1.0f - y) * lq));
        const float sq = s * (1.0f - s);
        g_marker[p] = (s - y) / fmaxf(sq, 1e-12f) * sq * inv_n;
    }
    __shared__ double s_red[LOSS_BLOCK / WAVE][LOSS_SUMS];
#pragma unroll
    for (int k = 0; k < LOSS_SUMS; ++k) {
        double v = acc[k];
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) v += __shfl_xor(v, d, WAVE);
        if ((threadIdx.x & (WAVE - 1)) == 0) s_red[threadIdx.x / WAVE][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < LOSS_SUMS) {
        double v = 0;
        for (int w = 0; w < LOSS_BLOCK / WAVE; ++w) v += s_red[w][threadIdx.x];
        partial[(size_t)blockIdx.x * LOSS_SUMS + threadIdx.x] = v;
    }
}

__global__ void __launch_bounds__(LOSS_BLOCK)
mapping_loss_finish_kernel(int blocks, int HW, const double* __restrict__ partial,
                           const float* __restrict__ exposure, float* __restrict__ out /*[4]*/)
{
    __shared__ double s_red[LOSS_BLOCK / WAVE][LOSS_SUMS];
    double acc[LOSS_SUMS] = {0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < blocks; b += blockDim.x)
        for (int k = 0; k < LOSS_SUMS; ++k) acc[k] += partial[(size_t)b * LOSS_SUMS + k];
#pragma unroll
    for (int k = 0; k < LOSS_SUMS; ++k) {
        double v = acc[k];
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) v += __shfl_xor(v, d, WAVE);
        if ((threadIdx.x & (WAVE - 1)) == 0) s_red[threadIdx.x / WAVE][k] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[LOSS_SUMS];
        for (int k = 0; k < LOSS_SUMS; ++k) {
            t[k] = 0;
            for (int w = 0; w < LOSS_BLOCK / WAVE; ++w) t[k] += s_red[w][k];
        }
        const double n = (double)HW;
        out[0] = (float)(t[0] / (3.0 * n) + t[1] / n);                 // get_loss_mapping
        out[1] = (float)(t[2] / n);                                    // get_loss_marker
        out[2] = exposure ? (float)(t[3] * (double)expf(exposure[0])) : 0.0f;  // dL/d exposure_a
        out[3] = exposure ? (float)t[4] : 0.0f;                        // dL/d exposure_b
    }
}

size_t mapping_loss_workspace_bytes(int32_t HW)
{
    int blocks = (HW + LOSS_BLOCK - 1) / LOSS_BLOCK;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (size_t)blocks * LOSS_SUMS * sizeof(double);
}

int launch_mapping_loss(int32_t HW, const float* image, const float* depth, const float* marker,
                        const float* gt_image, const float* gt_depth, const uint8_t* kp, float threshold,
                        const float* exposure, float* g_image, float* g_depth, float* g_marker, float* out,
                        void* workspace, hipStream_t stream)
{
    int blocks = (HW + LOSS_BLOCK - 1) / LOSS_BLOCK;
    if (blocks > 2048) blocks = 2048;
    double* partial = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(mapping_loss_kernel, dim3(blocks), dim3(LOSS_BLOCK), 0, stream, HW, image, depth, marker,
                       gt_image, gt_depth, kp, threshold, exposure, g_image, g_depth, g_marker, partial);
    SR_LAUNCH_CHECK();
    hipLaunchKernelGGL(mapping_loss_finish_kernel, dim3(1), dim3(LOSS_BLOCK), 0, stream, blocks, HW, partial,
                       exposure, out);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

}  // namespace sr
