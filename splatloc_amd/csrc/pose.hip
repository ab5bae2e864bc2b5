// pose.hip — the two small kernels that close the pose-refinement loop on the device (splatloc_amd/pose.py refine_pose):
//   l1_rgbd_loss_kernel  L = mean |colour - target| + w_d mean |depth - target_d| and its gradient planes, one pass;
//   pose_step_kernel     ONE thread: chains dL/dviewmatrix, dL/dprojmatrix, dL/dcampos (what the rasterizer's backward
//                        returns, preprocess_bwd.hip) through  view = (T(w, t) W2C0)^T,  proj = view P,  campos = -R^T t  to the
//                        six pose parameters (axis-angle w, translation t; utils/optimization_utils.py:31-42's
//                        at_to_transform_matrix), takes an Adam step on them and writes the NEXT iteration's camera tensors.
// The reference has no such loop (its rasterizer returns no camera gradient); with these two kernels an
// iteration is a launch sequence without a torch operator, an autograd graph or a host read of the gradient
// (round 4's refine_pose: ~60 tiny torch kernels and a graph per iteration for 6 numbers).
#include "common.h"

namespace sr {

constexpr int L1_THREADS = 256;

// one plane set (colour or depth): |x - t| w summed, the gradient plane written; 16 bytes per access when the three pointers allow it
__device__ __forceinline__ float l1_region(int64_t n, const float* __restrict__ x, const float* __restrict__ t, float w,
                                           float* __restrict__ g, int64_t tid, int64_t nthreads)
{
    float acc = 0.0f;
    auto one = [&](float xv, float tv) -> float {
        const float d = xv - tv;
        acc += fabsf(d) * w;
        return d > 0.0f ? w : (d < 0.0f ? -w : 0.0f);
    };
    const bool vec = (((uintptr_t)x | (uintptr_t)t | (uintptr_t)g) & 15u) == 0;
    const int64_t nq = vec ? n / 4 : 0;
    for (int64_t q = tid; q < nq; q += nthreads) {
        const float4 xv = reinterpret_cast<const float4*>(x)[q], tv = reinterpret_cast<const float4*>(t)[q];
        float4 gv;
        gv.x = one(xv.x, tv.x);
        gv.y = one(xv.y, tv.y);
        gv.z = one(xv.z, tv.z);
        gv.w = one(xv.w, tv.w);
        if (g) reinterpret_cast<float4*>(g)[q] = gv;
    }
    for (int64_t e = 4 * nq + tid; e < n; e += nthreads) {
        const float gv = one(x[e], t[e]);
        if (g) g[e] = gv;
    }
    return acc;
}

// (round 5's first version: one element per thread and iteration, 1 200 blocks each ending in an atomic on the one loss word —
//  23.6 us for 1.2 M elements, most of it the atomics' queue; now at most 256 blocks and 16-byte accesses)
__global__ void __launch_bounds__(L1_THREADS)
l1_rgbd_loss_kernel(int64_t n_color, const float* __restrict__ color, const float* __restrict__ tgt_c, int64_t n_depth,
                    const float* __restrict__ depth, const float* __restrict__ tgt_d, float depth_weight,
                    float* __restrict__ g_color, float* __restrict__ g_depth, float* __restrict__ loss_out)
{
    __shared__ float s_sum[L1_THREADS / WAVE];
    const float wc = 1.0f / (float)n_color, wd = (tgt_d && n_depth) ? depth_weight / (float)n_depth : 0.0f;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
    float acc = l1_region(n_color, color, tgt_c, wc, g_color, tid, nthreads);
    if (tgt_d) {
        acc += l1_region(n_depth, depth, tgt_d, wd, g_depth, tid, nthreads);
    } else if (g_depth) {
        for (int64_t e = tid; e < n_depth; e += nthreads) g_depth[e] = 0.0f;
    }
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0) s_sum[threadIdx.x / WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < L1_THREADS / WAVE; ++k) t += s_sum[k];
        atomicAdd(loss_out, t);
    }
}

int launch_l1_rgbd_loss(int64_t n_color, const float* color, const float* tgt_c, int64_t n_depth, const float* depth,
                        const float* tgt_d, float depth_weight, float* g_color, float* g_depth, float* loss_out,
                        hipStream_t stream)
{
    const int64_t n = n_color + n_depth;
    int blocks = (int)((n + L1_THREADS * 16 - 1) / (L1_THREADS * 16));
    blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
    hipLaunchKernelGGL(l1_rgbd_loss_kernel, dim3(blocks), dim3(L1_THREADS), 0, stream, n_color, color, tgt_c, n_depth, depth,
                       tgt_d, depth_weight, g_color, g_depth, loss_out);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

// forward-mode derivative with respect to the three components of w
struct D3 {
    double v, d[3];
};
__device__ inline D3 dconst(double c) { return {c, {0.0, 0.0, 0.0}}; }
__device__ inline D3 operator+(D3 a, D3 b) { return {a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}}; }
__device__ inline D3 operator-(D3 a, D3 b) { return {a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}}; }
__device__ inline D3 operator*(D3 a, D3 b)
{
    return {a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2]}};
}
__device__ inline D3 operator/(D3 a, D3 b)
{
    const double q = a.v / b.v, ib = 1.0 / b.v;
    return {q, {(a.d[0] - q * b.d[0]) * ib, (a.d[1] - q * b.d[1]) * ib, (a.d[2] - q * b.d[2]) * ib}};
}
__device__ inline D3 dfun(D3 a, double f, double df) { return {f, {df * a.d[0], df * a.d[1], df * a.d[2]}}; }

// R(w) = I + a [w]x + b [w]x^2, a = sin t / t, b = (1 - cos t) / t^2 (series below t^2 = 1e-12): splatloc_amd/pose.py
// axis_angle_to_matrix, i.e. utils/optimization_utils.py:5-22 made regular at w = 0
__device__ void rodrigues(const double w[3], D3 R[3][3])
{
    D3 x[3];
    for (int k = 0; k < 3; ++k) { x[k] = dconst(w[k]); x[k].d[k] = 1.0; }
    const D3 t2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    D3 a, b;
    if (t2.v < 1e-12) {
        a = dconst(1.0) - t2 / dconst(6.0);
        b = dconst(0.5) - t2 / dconst(24.0);
    } else {
        const D3 t = dfun(t2, sqrt(t2.v), 0.5 / sqrt(t2.v));
        a = dfun(t, sin(t.v), cos(t.v)) / t;
        b = (dconst(1.0) - dfun(t, cos(t.v), -sin(t.v))) / t2;
    }
    const D3 z = dconst(0.0);
    const D3 K[3][3] = {{z, z - x[2], x[1]}, {x[2], z, z - x[0]}, {z - x[1], x[0], z}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            D3 kk = K[i][0] * K[0][j] + K[i][1] * K[1][j] + K[i][2] * K[2][j];
            R[i][j] = dconst(i == j ? 1.0 : 0.0) + a * K[i][j] + b * kk;
        }
}

// state: w[3], t[3], exp_avg[6], exp_avg_sq[6], step, (pad)
__global__ void pose_step_kernel(const float* __restrict__ dL_dview, const float* __restrict__ dL_dproj,
                                 const float* __restrict__ dL_dcampos, const float* __restrict__ W2C0,
                                 const float* __restrict__ Pm, float lr_rot, float lr_trans, float beta1, float beta2, float eps,
                                 int advance, float* __restrict__ state, float* __restrict__ view_out,
                                 float* __restrict__ proj_out, float* __restrict__ campos_out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double w[3], t[3];
    for (int k = 0; k < 3; ++k) { w[k] = state[k]; t[k] = state[3 + k]; }
    double A[4][4], P4[4][4];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) { A[r][c] = W2C0[4 * r + c]; P4[r][c] = Pm[4 * r + c]; }
    if (advance) {
        D3 R[3][3];
        rodrigues(w, R);
        // W2C = T A, T = [R t; 0 1]
        double M[4][4];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) M[r][c] = R[r][0].v * A[0][c] + R[r][1].v * A[1][c] + R[r][2].v * A[2][c] + t[r] * A[3][c];
        for (int c = 0; c < 4; ++c) M[3][c] = A[3][c];
        // dL/dview_total = dL/dview + dL/dproj P^T ; dL/dW2C = its transpose
        double GW[4][4];
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                double g = dL_dview[4 * r + c];
                for (int k = 0; k < 4; ++k) g += (double)dL_dproj[4 * r + k] * P4[c][k];
                GW[c][r] = g;   // view[r][c] = W2C[c][r]
            }
        if (dL_dcampos) {   // campos_b = -sum_a Rc[a][b] tc[a], Rc = W2C[:3,:3], tc = W2C[:3,3]
            for (int a = 0; a < 3; ++a) {
                double s = 0.0;
                for (int b = 0; b < 3; ++b) {
                    GW[a][b] += -M[a][3] * (double)dL_dcampos[b];
                    s += M[a][b] * (double)dL_dcampos[b];
                }
                GW[a][3] += -s;
            }
        }
        // dL/dT = dL/dW2C A^T (rows 0..2)
        double GT[3][4];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) GT[r][c] = GW[r][0] * A[c][0] + GW[r][1] * A[c][1] + GW[r][2] * A[c][2] + GW[r][3] * A[c][3];
        double g[6] = {0, 0, 0, GT[0][3], GT[1][3], GT[2][3]};
        for (int k = 0; k < 3; ++k)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) g[k] += GT[i][j] * R[i][j].d[k];
        // torch.optim.Adam (no weight decay, no amsgrad): bias-corrected moments, eps added to the corrected denominator
        const double step = (double)state[18] + 1.0;
        const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
        for (int k = 0; k < 6; ++k) {
            const double m = (double)beta1 * state[6 + k] + (1.0 - (double)beta1) * g[k];
            const double v = (double)beta2 * state[12 + k] + (1.0 - (double)beta2) * g[k] * g[k];
            state[6 + k] = (float)m;
            state[12 + k] = (float)v;
            const double lr = k < 3 ? lr_rot : lr_trans;
            const double upd = (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + (double)eps);
            if (k < 3) w[k] -= upd; else t[k - 3] -= upd;
        }
        state[18] = (float)step;
        for (int k = 0; k < 3; ++k) { state[k] = (float)w[k]; state[3 + k] = (float)t[k]; w[k] = state[k]; t[k] = state[3 + k]; }
    }
    // the camera tensors of the (new) pose, float32 like the caller's torch code would produce them
    D3 R[3][3];
    rodrigues(w, R);
    float M[4][4];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c)
            M[r][c] = (float)(R[r][0].v * A[0][c] + R[r][1].v * A[1][c] + R[r][2].v * A[2][c] + t[r] * A[3][c]);
    for (int c = 0; c < 4; ++c) M[3][c] = (float)A[3][c];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) view_out[4 * r + c] = M[c][r];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += (double)M[k][r] * P4[k][c];
            proj_out[4 * r + c] = (float)s;
        }
    if (campos_out)
        for (int b = 0; b < 3; ++b)
            campos_out[b] = (float)(-((double)M[0][b] * M[0][3] + (double)M[1][b] * M[1][3] + (double)M[2][b] * M[2][3]));
}

int launch_pose_step(const float* dL_dview, const float* dL_dproj, const float* dL_dcampos, const float* W2C0, const float* Pm,
                     float lr_rot, float lr_trans, float beta1, float beta2, float eps, int advance, float* state, float* view_out,
                     float* proj_out, float* campos_out, hipStream_t stream)
{
    hipLaunchKernelGGL(pose_step_kernel, dim3(1), dim3(64), 0, stream, dL_dview, dL_dproj, dL_dcampos, W2C0, Pm, lr_rot, lr_trans,
                       beta1, beta2, eps, advance, state, view_out, proj_out, campos_out);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

}  // namespace sr
