// losses.hip — SplatLoc's per-view mapping loss and its gradient w.r.t. the rendered buffers in
// ONE pass over the pixels.  Replaces the chain of elementwise torch kernels
// (masks, exposure affine, abs, three means, sigmoid, BCE) and the equally long autograd chain of
//   utils/utils.py:55-82     get_loss_mapping / get_loss_mapping_rgbd
//   train_gaussians.py:38-42 get_loss_marker
// as summed per view at train_gaussians.py:217-218:
//   loss = mean |m_rgb x - m_rgb gt| + mean |m_d depth - m_d gt_depth| + mean BCE(sigmoid(marker), kp)
//   x = exp(a) image + b,  m_rgb = sum_c gt[c] > threshold,  m_d = gt_depth > 0.01
// kp is the float score map used as a SOFT BCE target (gt.view(-1).float(), train_gaussians.py:40).
// HBM-bound: reads 9 planes, writes 5 planes per pixel.  The five sums (three loss
// terms, dL/da, dL/db) are reduced per block in double and finished by a one-block kernel:
// deterministic, no atomics.
#include "common.h"

namespace sr {

constexpr int LOSS_BLOCK = 256;
constexpr int LOSS_SUMS = 5;  // l1 rgb, l1 depth, bce, d/da, d/db

__device__ __forceinline__ float sgn(float v) { return (float)((v > 0.f) - (v < 0.f)); }

// the views of a window in ONE launch pair (blockIdx.y = view): every view's blocks, partial sums and their order are those
// of the single-view launch, so the window's per-view results are bit-identical to V separate calls
struct LossViews {
    const float* image[MAX_VIEWS]; const float* depth[MAX_VIEWS]; const float* marker[MAX_VIEWS];
    const float* gt_image[MAX_VIEWS]; const float* gt_depth[MAX_VIEWS]; const float* kp[MAX_VIEWS];
    const float* exposure[MAX_VIEWS];   // [2] = a, b or NULL
    float* g_image[MAX_VIEWS]; float* g_depth[MAX_VIEWS]; float* g_marker[MAX_VIEWS];
};

__global__ void __launch_bounds__(LOSS_BLOCK)
mapping_loss_kernel(int HW, LossViews lv, float threshold, double* __restrict__ partial_all /*[V][gridDim.x][LOSS_SUMS]*/)
{
    const int view = blockIdx.y;
    const float* __restrict__ image = lv.image[view];
    const float* __restrict__ depth = lv.depth[view];
    const float* __restrict__ marker = lv.marker[view];
    const float* __restrict__ gt_image = lv.gt_image[view];
    const float* __restrict__ gt_depth = lv.gt_depth[view];
    const float* __restrict__ kp = lv.kp[view];
    const float* __restrict__ exposure = lv.exposure[view];
    float* __restrict__ g_image = lv.g_image[view];
    float* __restrict__ g_depth = lv.g_depth[view];
    float* __restrict__ g_marker = lv.g_marker[view];
    double* __restrict__ partial = partial_all + (size_t)view * gridDim.x * LOSS_SUMS;
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
        const float y = kp[p];   // soft target: the raw key-point score map (utils/dataset.py:94, camera_utils.py:75)
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
mapping_loss_finish_kernel(int blocks, int HW, const double* __restrict__ partial_all, LossViews lv,
                           float* __restrict__ out_all /*[V][4]*/)
{
    const int view = blockIdx.x;
    const double* __restrict__ partial = partial_all + (size_t)view * blocks * LOSS_SUMS;
    const float* __restrict__ exposure = lv.exposure[view];
    float* __restrict__ out = out_all + 4 * view;
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

static int loss_blocks(int32_t HW)
{
    int blocks = (HW + LOSS_BLOCK - 1) / LOSS_BLOCK;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return blocks;
}

size_t mapping_loss_workspace_bytes(int32_t HW) { return (size_t)loss_blocks(HW) * LOSS_SUMS * sizeof(double); }

// V views (host array of per-view pointers), out [V][4], workspace V x mapping_loss_workspace_bytes(HW)
int launch_mapping_loss_window(int32_t V, int32_t HW, const splatraster_loss_view* views, float threshold, float* out,
                               void* workspace, hipStream_t stream)
{
    LossViews lv{};
    for (int v = 0; v < V; ++v) {
        lv.image[v] = views[v].image; lv.depth[v] = views[v].depth; lv.marker[v] = views[v].marker;
        lv.gt_image[v] = views[v].gt_image; lv.gt_depth[v] = views[v].gt_depth; lv.kp[v] = views[v].kp;
        lv.exposure[v] = views[v].exposure;
        lv.g_image[v] = views[v].g_image; lv.g_depth[v] = views[v].g_depth; lv.g_marker[v] = views[v].g_marker;
    }
    const int blocks = loss_blocks(HW);
    double* partial = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(mapping_loss_kernel, dim3(blocks, V), dim3(LOSS_BLOCK), 0, stream, HW, lv, threshold, partial);
    SR_LAUNCH_CHECK();
    hipLaunchKernelGGL(mapping_loss_finish_kernel, dim3(V), dim3(LOSS_BLOCK), 0, stream, blocks, HW, partial, lv, out);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

int launch_mapping_loss(int32_t HW, const float* image, const float* depth, const float* marker,
                        const float* gt_image, const float* gt_depth, const float* kp, float threshold,
                        const float* exposure, float* g_image, float* g_depth, float* g_marker, float* out,
                        void* workspace, hipStream_t stream)
{
    const splatraster_loss_view one{image, depth, marker, gt_image, gt_depth, kp, exposure, g_image, g_depth, g_marker};
    return launch_mapping_loss_window(1, HW, &one, threshold, out, workspace, stream);
}


// ---------------------------------------------------------------------------------------------
// Colour-refinement loss (train_gaussians.py:283-285):
//     loss = (1 - lambda) l1_loss(image, gt) + lambda (1 - ssim(image, gt))
// l1_loss / ssim: gaussian_splatting/utils/loss_utils.py:21-22, 42-102 (11x11 Gaussian window,
// sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2, mean over all elements).
// The reference runs 5 dense 11x11 grouped convolutions forward and their autograd backward;
// here the window is applied separably out of LDS, in two kernels:
//   refine_fwd: per 16x16 tile (+5 halo) blur(x, y, x^2, y^2, xy) -> SSIM map value and its
//               partials dm/dmu1, dm/ds1, dm/ds12 (s1 = blur(x^2), s12 = blur(xy)); tile sums of m
//               and |x - y|
//   refine_bwd: dL/dx = (1-lambda) sign(x-y)/n - lambda/n (blur(dm/dmu1) + 2 x blur(dm/ds1) + y blur(dm/ds12))
// (the window is symmetric, so the adjoint of the zero-padded convolution is the same blur).
// ---------------------------------------------------------------------------------------------
constexpr int RT = 16;            // tile edge
constexpr int RH = 5;             // window radius
constexpr int RW = 2 * RH + 1;    // 11 taps
constexpr int RE = RT + 2 * RH;   // 26: tile + halo

struct RefineWindow { float w[RW]; };

// Round 5: both kernels work on 32x16 tiles with REGISTER-BLOCKED passes — a horizontal work item loads 14 consecutive taps
// once and produces 4 adjacent outputs, a vertical one loads 12 rows and produces 2 (the 16x16 / one-output-per-thread form
// read LDS 22 + 55 times per output): refine_fwd 22.1 -> 18.4 us, refine_bwd 16.1 -> 14.4 us (profiles/r05_ab_probes.txt #8;
// what is left is the load -> barrier -> pass -> barrier -> pass -> store chain, not arithmetic).  Every output is still
// accumulated tap by tap in ascending order with the same fused multiply-adds: the maps and the gradient are bit-identical.
constexpr int FW = 32, FH = 16;               // tile of the refinement kernels
constexpr int FEW = FW + 2 * RH, FEH = FH + 2 * RH;   // 42 x 26 with halo
constexpr int FHG = FW / 4;                   // horizontal work items per row (4 outputs each)

__global__ void __launch_bounds__(256)
refine_fwd_kernel(int H, int W, const float* __restrict__ img, const float* __restrict__ gt, RefineWindow win,
                  float* __restrict__ dm_dmu1, float* __restrict__ dm_ds1, float* __restrict__ dm_ds12,
                  double* __restrict__ partial /*[blocks][2]: sum m, sum |x-y|*/)
{
    __shared__ float s_x[FEH][FEW + 1], s_y[FEH][FEW + 1];
    __shared__ float s_h[5][FEH][FW + 1];
    __shared__ double s_red[256 / WAVE][2];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * FW, y0 = blockIdx.y * FH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    for (int e = tid; e < FEH * FEW; e += 256) {
        const int r = e / FEW, c = e - r * FEW;
        const int gy = y0 + r - RH, gx = x0 + c - RH;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        s_x[r][c] = in ? img[plane + (size_t)gy * W + gx] : 0.0f;
        s_y[r][c] = in ? gt[plane + (size_t)gy * W + gx] : 0.0f;
    }
    __syncthreads();
    if (tid < FEH * FHG) {   // horizontal pass: FEH rows x FHG groups of 4 columns
        const int r = tid / FHG, c0 = (tid - r * FHG) * 4;
        float xv[RW + 3], yv[RW + 3], xx[RW + 3], yy[RW + 3], xy[RW + 3];
#pragma unroll
        for (int k = 0; k < RW + 3; ++k) {
            xv[k] = s_x[r][c0 + k];
            yv[k] = s_y[r][c0 + k];
            xx[k] = xv[k] * xv[k];
            yy[k] = yv[k] * yv[k];
            xy[k] = xv[k] * yv[k];
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
            for (int k = 0; k < RW; ++k) {
                const float wk = win.w[k];
                a0 += wk * xv[o + k];
                a1 += wk * yv[o + k];
                a2 += wk * xx[o + k];
                a3 += wk * yy[o + k];
                a4 += wk * xy[o + k];
            }
            s_h[0][r][c0 + o] = a0; s_h[1][r][c0 + o] = a1; s_h[2][r][c0 + o] = a2; s_h[3][r][c0 + o] = a3; s_h[4][r][c0 + o] = a4;
        }
    }
    __syncthreads();
    const int tx = tid & (FW - 1), ty0 = (tid / FW) * 2;   // two vertically adjacent outputs per thread
    const int gx = x0 + tx;
    double msum = 0.0, lsum = 0.0;
    float col[5][RW + 1];
#pragma unroll
    for (int k = 0; k < RW + 1; ++k)
#pragma unroll
        for (int q = 0; q < 5; ++q) col[q][k] = s_h[q][ty0 + k][tx];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int ty = ty0 + o, gy = y0 + ty;
        if (gy < H && gx < W) {
            float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < RW; ++k) {
                const float wk = win.w[k];
#pragma unroll
                for (int q = 0; q < 5; ++q) v[q] += wk * col[q][o + k];
            }
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float mu1 = v[0], mu2 = v[1];
            const float sig1 = v[2] - mu1 * mu1, sig2 = v[3] - mu2 * mu2, sig12 = v[4] - mu1 * mu2;
            const float A = 2.0f * mu1 * mu2 + C1, B = 2.0f * sig12 + C2;
            const float Cc = mu1 * mu1 + mu2 * mu2 + C1, D = sig1 + sig2 + C2;
            const float icd = 1.0f / (Cc * D);
            const float m = A * B * icd;
            const size_t oo = plane + (size_t)gy * W + gx;
            dm_dmu1[oo] = (2.0f * mu2 * (B - A) * Cc * D - A * B * 2.0f * mu1 * (D - Cc)) * icd * icd;
            dm_ds1[oo] = -m / D;
            dm_ds12[oo] = 2.0f * A * icd;
            msum += (double)m;
            lsum += (double)fabsf(s_x[ty + RH][tx + RH] - s_y[ty + RH][tx + RH]);
        }
    }
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        msum += __shfl_xor(msum, d, WAVE);
        lsum += __shfl_xor(lsum, d, WAVE);
    }
    if ((tid & (WAVE - 1)) == 0) { s_red[tid / WAVE][0] = msum; s_red[tid / WAVE][1] = lsum; }
    __syncthreads();
    if (tid == 0) {
        double a = 0, bsum = 0;
        for (int w = 0; w < 256 / WAVE; ++w) { a += s_red[w][0]; bsum += s_red[w][1]; }
        const size_t bid = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * bid] = a;
        partial[2 * bid + 1] = bsum;
    }
}

// the loss values out of refine_fwd_kernel's per-block partial sums (a fixed order: deterministic): run by ONE block of
// refine_bwd_kernel — the values depend on the forward launch only, and a launch of their own cost the stream 4.7 us
__device__ __forceinline__ void refine_finish_block(int blocks, double n_total, float lambda, const double* __restrict__ partial,
                                                    float* __restrict__ out /*[3] = l1, ssim, loss*/)
{
    __shared__ double s_red[LOSS_BLOCK / WAVE][2];
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < blocks; i += LOSS_BLOCK) { a += partial[2 * (size_t)i]; b += partial[2 * (size_t)i + 1]; }
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) { a += __shfl_xor(a, d, WAVE); b += __shfl_xor(b, d, WAVE); }
    if ((threadIdx.x & (WAVE - 1)) == 0) { s_red[threadIdx.x / WAVE][0] = a; s_red[threadIdx.x / WAVE][1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = 0, l = 0;
        for (int w = 0; w < LOSS_BLOCK / WAVE; ++w) { m += s_red[w][0]; l += s_red[w][1]; }
        const double ssim = m / n_total, l1 = l / n_total;
        out[0] = (float)l1;
        out[1] = (float)ssim;
        out[2] = (float)((1.0 - (double)lambda) * l1 + (double)lambda * (1.0 - ssim));
    }
}

__global__ void __launch_bounds__(256)
refine_bwd_kernel(int H, int W, float n_total, float lambda, const float* __restrict__ img,
                  const float* __restrict__ gt, RefineWindow win, const float* __restrict__ dm_dmu1,
                  const float* __restrict__ dm_ds1, const float* __restrict__ dm_ds12, float* __restrict__ g_img,
                  int blocks, double n_total_d, const double* __restrict__ partial, float* __restrict__ out)
{
    __shared__ float s_m[3][FEH][FEW + 1];
    __shared__ float s_h[3][FEH][FW + 1];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * FW, y0 = blockIdx.y * FH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    for (int e = tid; e < FEH * FEW; e += 256) {
        const int r = e / FEW, c = e - r * FEW;
        const int gy = y0 + r - RH, gx = x0 + c - RH;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t o = plane + (size_t)gy * W + gx;
        s_m[0][r][c] = in ? dm_dmu1[o] : 0.0f;
        s_m[1][r][c] = in ? dm_ds1[o] : 0.0f;
        s_m[2][r][c] = in ? dm_ds12[o] : 0.0f;
    }
    __syncthreads();
    if (tid < FEH * FHG) {
        const int r = tid / FHG, c0 = (tid - r * FHG) * 4;
        float mv[3][RW + 3];
#pragma unroll
        for (int k = 0; k < RW + 3; ++k)
#pragma unroll
            for (int q = 0; q < 3; ++q) mv[q][k] = s_m[q][r][c0 + k];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int k = 0; k < RW; ++k) {
                const float wk = win.w[k];
                a0 += wk * mv[0][o + k];
                a1 += wk * mv[1][o + k];
                a2 += wk * mv[2][o + k];
            }
            s_h[0][r][c0 + o] = a0; s_h[1][r][c0 + o] = a1; s_h[2][r][c0 + o] = a2;
        }
    }
    __syncthreads();
    const int tx = tid & (FW - 1), ty0 = (tid / FW) * 2;
    const int gx = x0 + tx;
    float col[3][RW + 1];
#pragma unroll
    for (int k = 0; k < RW + 1; ++k)
#pragma unroll
        for (int q = 0; q < 3; ++q) col[q][k] = s_h[q][ty0 + k][tx];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int gy = y0 + ty0 + o;
        if (gy >= H || gx >= W) continue;
        float b0 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const float wk = win.w[k];
            b0 += wk * col[0][o + k];
            b1 += wk * col[1][o + k];
            b2 += wk * col[2][o + k];
        }
        const size_t oo = plane + (size_t)gy * W + gx;
        const float x = img[oo], y = gt[oo];
        const float inv_n = 1.0f / n_total;
        g_img[oo] = (1.0f - lambda) * sgn(x - y) * inv_n - lambda * inv_n * (b0 + 2.0f * x * b1 + y * b2);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) refine_finish_block(blocks, n_total_d, lambda, partial, out);
}

static RefineWindow refine_window()
{
    // loss_utils.py:42-49: exp in double -> float32 tensor -> divided by its float32 sum
    RefineWindow win;
    float sum = 0.f;
    for (int k = 0; k < RW; ++k) {
        win.w[k] = (float)exp(-(double)((k - RH) * (k - RH)) / (2.0 * 1.5 * 1.5));
        sum += win.w[k];
    }
    for (int k = 0; k < RW; ++k) win.w[k] /= sum;
    return win;
}

size_t refinement_loss_workspace_bytes(int32_t C, int32_t H, int32_t W)
{
    const size_t blocks = (size_t)((W + FW - 1) / FW) * ((H + FH - 1) / FH) * (size_t)C;
    return align_up(3 * sizeof(float) * (size_t)C * H * W, 256) + blocks * 2 * sizeof(double);
}

// ---------------------------------------------------------------------------------------------
// eval_rendering's per-frame metrics (utils/eval_utils.py:22-72):
//     image = clamp(render, 0, 1);  mask = gt > 0 (per element)
//     psnr  = 20 log10(1 / sqrt(mean_{mask} (image - gt)^2))          gaussian_splatting/utils/image_utils.py:19-21
//     ssim  = ssim(image, gt)                                          loss_utils.py:61-102 (same window as above)
// one pass over the frame: the tile loader clamps, the SSIM map is summed (its partial derivatives are not needed),
// and the masked squared error and the mask count ride in the same block partials (double, deterministic).
// ---------------------------------------------------------------------------------------------
constexpr int EVAL_SUMS = 3;   // sum ssim map, sum mask (x - y)^2, sum mask

__global__ void __launch_bounds__(RT * RT)
eval_metrics_kernel(int H, int W, const float* __restrict__ img, const float* __restrict__ gt, RefineWindow win,
                    double* __restrict__ partial /*[blocks][EVAL_SUMS]*/)
{
    __shared__ float s_x[RE][RE + 1], s_y[RE][RE + 1];
    __shared__ float s_h[5][RE][RT + 1];
    __shared__ double s_red[RT * RT / WAVE][EVAL_SUMS];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * RT, y0 = blockIdx.y * RT;
    const size_t plane = (size_t)blockIdx.z * H * W;
    for (int e = tid; e < RE * RE; e += RT * RT) {
        const int r = e / RE, c = e - r * RE;
        const int gy = y0 + r - RH, gx = x0 + c - RH;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        s_x[r][c] = in ? fminf(fmaxf(img[plane + (size_t)gy * W + gx], 0.0f), 1.0f) : 0.0f;   // torch.clamp(rendering, 0.0, 1.0)
        s_y[r][c] = in ? gt[plane + (size_t)gy * W + gx] : 0.0f;
    }
    __syncthreads();
    for (int e = tid; e < RE * RT; e += RT * RT) {
        const int r = e / RT, c = e - r * RT;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const float xv = s_x[r][c + k], yv = s_y[r][c + k], wk = win.w[k];
            a0 += wk * xv;
            a1 += wk * yv;
            a2 += wk * (xv * xv);
            a3 += wk * (yv * yv);
            a4 += wk * (xv * yv);
        }
        s_h[0][r][c] = a0; s_h[1][r][c] = a1; s_h[2][r][c] = a2; s_h[3][r][c] = a3; s_h[4][r][c] = a4;
    }
    __syncthreads();
    const int ty = tid / RT, tx = tid - ty * RT;
    const int gy = y0 + ty, gx = x0 + tx;
    double acc[EVAL_SUMS] = {0.0, 0.0, 0.0};
    if (gy < H && gx < W) {
        float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const float wk = win.w[k];
#pragma unroll
            for (int q = 0; q < 5; ++q) v[q] += wk * s_h[q][ty + k][tx];
        }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float mu1 = v[0], mu2 = v[1];
        const float sig1 = v[2] - mu1 * mu1, sig2 = v[3] - mu2 * mu2, sig12 = v[4] - mu1 * mu2;
        const float A = 2.0f * mu1 * mu2 + C1, B = 2.0f * sig12 + C2;
        const float Cc = mu1 * mu1 + mu2 * mu2 + C1, D = sig1 + sig2 + C2;
        acc[0] = (double)(A * B / (Cc * D));
        const float x = s_x[ty + RH][tx + RH], y = s_y[ty + RH][tx + RH];
        if (y > 0.0f) {
            const float d = x - y;
            acc[1] = (double)(d * d);
            acc[2] = 1.0;
        }
    }
#pragma unroll
    for (int k = 0; k < EVAL_SUMS; ++k) {
        double t = acc[k];
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) t += __shfl_xor(t, d, WAVE);
        if ((tid & (WAVE - 1)) == 0) s_red[tid / WAVE][k] = t;
    }
    __syncthreads();
    if (tid < EVAL_SUMS) {
        double t = 0;
        for (int w = 0; w < RT * RT / WAVE; ++w) t += s_red[w][tid];
        const size_t bid = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[EVAL_SUMS * bid + tid] = t;
    }
}

__global__ void __launch_bounds__(LOSS_BLOCK)
eval_metrics_finish_kernel(int blocks, double n_total, const double* __restrict__ partial,
                           float* __restrict__ out /*[4] = psnr, ssim, masked mse, mask count*/)
{
    __shared__ double s_red[LOSS_BLOCK / WAVE][EVAL_SUMS];
    double acc[EVAL_SUMS] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < blocks; i += blockDim.x)
        for (int k = 0; k < EVAL_SUMS; ++k) acc[k] += partial[EVAL_SUMS * (size_t)i + k];
#pragma unroll
    for (int k = 0; k < EVAL_SUMS; ++k) {
        double t = acc[k];
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) t += __shfl_xor(t, d, WAVE);
        if ((threadIdx.x & (WAVE - 1)) == 0) s_red[threadIdx.x / WAVE][k] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[EVAL_SUMS];
        for (int k = 0; k < EVAL_SUMS; ++k) {
            t[k] = 0;
            for (int w = 0; w < LOSS_BLOCK / WAVE; ++w) t[k] += s_red[w][k];
        }
        const double mse = t[2] > 0 ? t[1] / t[2] : 0.0 / 0.0;     // an empty mask: the reference's mean over nothing is NaN
        out[0] = (float)(20.0 * log10(1.0 / sqrt(mse)));
        out[1] = (float)(t[0] / n_total);
        out[2] = (float)mse;
        out[3] = (float)t[2];
    }
}

size_t eval_metrics_workspace_bytes(int32_t C, int32_t H, int32_t W)
{
    const size_t blocks = (size_t)((W + RT - 1) / RT) * ((H + RT - 1) / RT) * (size_t)C;
    return blocks * EVAL_SUMS * sizeof(double);
}

int launch_eval_metrics(int32_t C, int32_t H, int32_t W, const float* image, const float* gt, float* out, void* workspace,
                        hipStream_t stream)
{
    const dim3 grid((W + RT - 1) / RT, (H + RT - 1) / RT, C);
    const int blocks = (int)(grid.x * grid.y * grid.z);
    double* partial = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(eval_metrics_kernel, grid, dim3(RT * RT), 0, stream, H, W, image, gt, refine_window(), partial);
    SR_LAUNCH_CHECK();
    hipLaunchKernelGGL(eval_metrics_finish_kernel, dim3(1), dim3(LOSS_BLOCK), 0, stream, blocks, (double)C * H * W, partial, out);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

int launch_refinement_loss(int32_t C, int32_t H, int32_t W, float lambda, const float* image, const float* gt,
                           float* g_image, float* out, void* workspace, hipStream_t stream)
{
    const size_t n = (size_t)C * H * W;
    float* maps = reinterpret_cast<float*>(workspace);
    double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + align_up(3 * sizeof(float) * n, 256));
    const dim3 grid((W + FW - 1) / FW, (H + FH - 1) / FH, C);
    const int blocks = (int)(grid.x * grid.y * grid.z);
    const RefineWindow win = refine_window();
    hipLaunchKernelGGL(refine_fwd_kernel, grid, dim3(256), 0, stream, H, W, image, gt, win, maps, maps + n,
                       maps + 2 * n, partial);
    SR_LAUNCH_CHECK();
    hipLaunchKernelGGL(refine_bwd_kernel, grid, dim3(256), 0, stream, H, W, (float)n, lambda, image, gt, win, maps,
                       maps + n, maps + 2 * n, g_image, blocks, (double)n, partial, out);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

}  // namespace sr
