// activations.hip — the parameter activations and SH / feature packing that sit between the raw
// optimiser tensors and the rasterizer call, as ONE kernel forward and ONE backward
// (SURVEY.md §8f-1).  Replaces the ~10 elementwise launches (and their P-sized round trips
// through HBM) of
//   gaussian_model.py:78-105            exp(_scaling), normalize(_rotation), sigmoid(_opacity), cat(f_dc, f_rest)
//   gaussian_renderer/__init__.py:73-102 repeat of an isotropic scale, dir = normalize(xyz - campos),
//                                        clamp_min(eval_sh + 0.5, 0), cat(rgb, kp_score)
//   sh_utils.py:55-118                  eval_sh
// HBM-bound: every array is streamed once (reads 44 + 12 K + 4 E bytes per Gaussian, writes
// 44 + 4 E forward; about twice that backward).  Accurate expf / sqrtf / division (no fast-math
// intrinsics): the outputs track torch's to the last ulp or two.
#include "activation_math.h"

namespace sr {

namespace {

constexpr float A_C0 = 0.28209479177387814f;
constexpr float A_C1 = 0.4886025119029199f;
__device__ const float A_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                  -1.0925484305920792f, 0.5462742152960396f};
__device__ const float A_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                  0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                  -0.5900435899266435f};

// SH basis B_k(d) of eval_sh = sum_k B_k sh_k, k < (deg + 1)^2 <= 16
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* B)
{
    B[0] = A_C0;
    if (deg > 0) {
        B[1] = -A_C1 * y;
        B[2] = A_C1 * z;
        B[3] = -A_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = A_C2[0] * xy;
            B[5] = A_C2[1] * yz;
            B[6] = A_C2[2] * (2.0f * zz - xx - yy);
            B[7] = A_C2[3] * xz;
            B[8] = A_C2[4] * (xx - yy);
            if (deg > 2) {
                B[9] = A_C3[0] * y * (3.0f * xx - yy);
                B[10] = A_C3[1] * xy * z;
                B[11] = A_C3[2] * y * (4.0f * zz - xx - yy);
                B[12] = A_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                B[13] = A_C3[4] * x * (4.0f * zz - xx - yy);
                B[14] = A_C3[5] * z * (xx - yy);
                B[15] = A_C3[6] * x * (xx - 3.0f * yy);
            }
        }
    }
}

// dB_k/d(x, y, z), k >= 1
__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, float (*dB)[3])
{
    dB[0][0] = dB[0][1] = dB[0][2] = 0.0f;
    if (deg > 0) {
        dB[1][0] = 0.f; dB[1][1] = -A_C1; dB[1][2] = 0.f;
        dB[2][0] = 0.f; dB[2][1] = 0.f; dB[2][2] = A_C1;
        dB[3][0] = -A_C1; dB[3][1] = 0.f; dB[3][2] = 0.f;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            dB[4][0] = A_C2[0] * y; dB[4][1] = A_C2[0] * x; dB[4][2] = 0.f;
            dB[5][0] = 0.f; dB[5][1] = A_C2[1] * z; dB[5][2] = A_C2[1] * y;
            dB[6][0] = A_C2[2] * -2.0f * x; dB[6][1] = A_C2[2] * -2.0f * y; dB[6][2] = A_C2[2] * 4.0f * z;
            dB[7][0] = A_C2[3] * z; dB[7][1] = 0.f; dB[7][2] = A_C2[3] * x;
            dB[8][0] = A_C2[4] * 2.0f * x; dB[8][1] = A_C2[4] * -2.0f * y; dB[8][2] = 0.f;
            if (deg > 2) {
                dB[9][0] = A_C3[0] * 6.0f * xy; dB[9][1] = A_C3[0] * (3.0f * xx - 3.0f * yy); dB[9][2] = 0.f;
                dB[10][0] = A_C3[1] * yz; dB[10][1] = A_C3[1] * xz; dB[10][2] = A_C3[1] * xy;
                dB[11][0] = A_C3[2] * -2.0f * xy; dB[11][1] = A_C3[2] * (4.0f * zz - xx - 3.0f * yy);
                dB[11][2] = A_C3[2] * 8.0f * yz;
                dB[12][0] = A_C3[3] * -6.0f * xz; dB[12][1] = A_C3[3] * -6.0f * yz;
                dB[12][2] = A_C3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
                dB[13][0] = A_C3[4] * (4.0f * zz - 3.0f * xx - yy); dB[13][1] = A_C3[4] * -2.0f * xy;
                dB[13][2] = A_C3[4] * 8.0f * xz;
                dB[14][0] = A_C3[5] * 2.0f * xz; dB[14][1] = A_C3[5] * -2.0f * yz; dB[14][2] = A_C3[5] * (xx - yy);
                dB[15][0] = A_C3[6] * (3.0f * xx - 3.0f * yy); dB[15][1] = A_C3[6] * -6.0f * xy; dB[15][2] = 0.f;
            }
        }
    }
}

__device__ __forceinline__ const float* sh_row(const float* f_dc, const float* f_rest, int i, int K, int k)
{
    return k == 0 ? f_dc + 3 * (size_t)i : f_rest + 3 * ((size_t)i * (K - 1) + (k - 1));
}

}  // namespace

// rgb channel c of Gaussian i before the clamp: eval_sh + 0.5
__device__ __forceinline__ float sh_channel(int i, int c, int K, int deg, const float* __restrict__ xyz,
                                            const float* __restrict__ f_dc, const float* __restrict__ f_rest,
                                            const float* __restrict__ campos, float* B /*[16] out*/)
{
    if (deg == 0) {
        B[0] = A_C0;
        return act_rgb_raw_deg0(f_dc[3 * (size_t)i + c]);
    }
    const float vx = xyz[3 * (size_t)i] - campos[0], vy = xyz[3 * (size_t)i + 1] - campos[1],
                vz = xyz[3 * (size_t)i + 2] - campos[2];
    const float len = sqrtf((vx * vx + vy * vy) + vz * vz);
    sh_basis(deg, vx / len, vy / len, vz / len, B);
    const int M = (deg + 1) * (deg + 1);
    float v = 0.f;
    for (int k = 0; k < M; ++k) v += B[k] * sh_row(f_dc, f_rest, i, K, k)[c];
    return v + 0.5f;
}

// Work item t is BOTH elements 4t..4t+3 of the packed colour table [P, 3 + E] (wide rows: a
// thread per 16 bytes keeps the loads and stores of the table coalesced) and, for t < P,
// Gaussian t of the narrow per-Gaussian tensors.
__global__ void __launch_bounds__(256)
activate_fwd_kernel(int P, int K, int deg, int SC, int E, const float* __restrict__ xyz,
                    const float* __restrict__ f_dc, const float* __restrict__ f_rest,
                    const float* __restrict__ scaling, const float* __restrict__ rotation,
                    const float* __restrict__ opacity, const float* __restrict__ extra,
                    const float* __restrict__ campos, float* __restrict__ scales,
                    float* __restrict__ rotations, float* __restrict__ opacities, float* __restrict__ colors)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int CW = 3 + E;
    const size_t total = (size_t)P * CW;
    // four consecutive elements of the colour table per thread: one 16-byte store, 16 bytes of
    // loads in flight per lane
    if (4 * t < total) {
        int row = (int)((4 * t) / (unsigned)CW), col = (int)(4 * t - (size_t)row * CW);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (4 * t + e < total) {
                if (col >= 3) {
                    v[e] = extra[(size_t)row * E + (col - 3)];
                } else {
                    float B[16];
                    v[e] = fmaxf(sh_channel(row, col, K, deg, xyz, f_dc, f_rest, campos, B), 0.0f);
                }
            }
            if (++col == CW) { col = 0; ++row; }
        }
        if (4 * t + 3 < total) {
            reinterpret_cast<float4*>(colors)[t] = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int e = 0; e < 4 && 4 * t + e < total; ++e) colors[4 * t + e] = v[e];
        }
    }
    if (t >= (size_t)P) return;
    const int i = (int)t;
    // scales = exp(_scaling) (an isotropic [P,1] model is repeated)
#pragma unroll
    for (int c = 0; c < 3; ++c) scales[3 * (size_t)i + c] = expf(scaling[(size_t)i * SC + (SC == 3 ? c : 0)]);
    // rotations = q / max(||q||, 1e-12)
    {
        reinterpret_cast<float4*>(rotations)[i] = act_normalize(reinterpret_cast<const float4*>(rotation)[i]);
    }
    opacities[i] = act_sigmoid(opacity[i]);
}

__global__ void __launch_bounds__(256)
activate_bwd_kernel(int P, int K, int deg, int SC, int E, const float* __restrict__ xyz,
                    const float* __restrict__ f_dc, const float* __restrict__ f_rest,
                    const float* __restrict__ scaling, const float* __restrict__ rotation,
                    const float* __restrict__ opacity, const float* __restrict__ campos,
                    const float* __restrict__ g_scales, const float* __restrict__ g_rotations,
                    const float* __restrict__ g_opacities, const float* __restrict__ g_colors,
                    float* __restrict__ d_xyz, float* __restrict__ d_f_dc, float* __restrict__ d_f_rest,
                    float* __restrict__ d_scaling, float* __restrict__ d_rotation,
                    float* __restrict__ d_opacity, float* __restrict__ d_extra)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int CW = 3 + E;
    const int M = (deg + 1) * (deg + 1);
    // ---- elements 4t .. 4t+3 of dL/dcolors: d cat, d clamp_min, d eval_sh w.r.t. the coefficients ----
    const size_t total = (size_t)P * CW;
    if (4 * t < total) {
        float gq[4];
        if (4 * t + 3 < total) {
            const float4 q = reinterpret_cast<const float4*>(g_colors)[t];
            gq[0] = q.x; gq[1] = q.y; gq[2] = q.z; gq[3] = q.w;
        } else {
            for (int e = 0; e < 4; ++e) gq[e] = 4 * t + e < total ? g_colors[4 * t + e] : 0.0f;
        }
        int row = (int)((4 * t) / (unsigned)CW), col = (int)(4 * t - (size_t)row * CW);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (4 * t + e < total) {
                if (col >= 3) {
                    d_extra[(size_t)row * E + (col - 3)] = gq[e];
                } else {
                    float B[16];
                    const float raw = sh_channel(row, col, K, deg, xyz, f_dc, f_rest, campos, B);
                    const float g = raw >= 0.0f ? gq[e] : 0.0f;  // clamp_min passes the gradient at >= 0
                    d_f_dc[3 * (size_t)row + col] = B[0] * g;
                    for (int k = 1; k < K; ++k)
                        d_f_rest[3 * ((size_t)row * (K - 1) + (k - 1)) + col] = k < M ? B[k] * g : 0.0f;
                }
            }
            if (++col == CW) { col = 0; ++row; }
        }
    }
    if (t >= (size_t)P) return;
    const int i = (int)t;
    // d exp
    {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float s = expf(scaling[(size_t)i * SC + (SC == 3 ? c : 0)]);
            const float v = g_scales[3 * (size_t)i + c] * s;
            if (SC == 3) d_scaling[3 * (size_t)i + c] = v; else acc += v;
        }
        if (SC != 3) d_scaling[i] = acc;
    }
    // d normalize: (g - u (u . g)) / n
    {
        const float4 q = reinterpret_cast<const float4*>(rotation)[i];
        const float4 g = reinterpret_cast<const float4*>(g_rotations)[i];
        reinterpret_cast<float4*>(d_rotation)[i] = act_normalize_bwd(q, g);
    }
    // d sigmoid
    {
        d_opacity[i] = act_sigmoid_bwd(g_opacities[i], opacity[i]);
    }
    // view-direction term of the SH colour (degree > 0 only)
    if (d_xyz) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (deg > 0) {
            const float vx = xyz[3 * (size_t)i] - campos[0], vy = xyz[3 * (size_t)i + 1] - campos[1],
                        vz = xyz[3 * (size_t)i + 2] - campos[2];
            const float len = sqrtf((vx * vx + vy * vy) + vz * vz);
            const float x = vx / len, y = vy / len, z = vz / len;
            float B[16];
            sh_basis(deg, x, y, z, B);
            float g[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float raw = 0.f;
                for (int k = 0; k < M; ++k) raw += B[k] * sh_row(f_dc, f_rest, i, K, k)[c];
                g[c] = (raw + 0.5f >= 0.0f) ? g_colors[(size_t)i * CW + c] : 0.0f;
            }
            float dB[16][3];
            sh_basis_grad(deg, x, y, z, dB);
            for (int k = 1; k < M; ++k) {
                const float* s = sh_row(f_dc, f_rest, i, K, k);
                const float w = (s[0] * g[0] + s[1] * g[1]) + s[2] * g[2];
                gx += dB[k][0] * w;
                gy += dB[k][1] * w;
                gz += dB[k][2] * w;
            }
            // through dir = v / |v|
            const float dg = (x * gx + y * gy) + z * gz;
            gx = (gx - x * dg) / len;
            gy = (gy - y * dg) / len;
            gz = (gz - z * dg) / len;
        }
        d_xyz[3 * (size_t)i] = gx;
        d_xyz[3 * (size_t)i + 1] = gy;
        d_xyz[3 * (size_t)i + 2] = gz;
    }
}

// Per-view densification statistics, in place, one launch and no host round trip (the reference
// indexes with boolean masks, i.e. a nonzero + device->host sync per statement):
//   vis = radii > 0
//   max_radii2D[vis] = max(max_radii2D[vis], radii[vis])                 train_gaussians.py:240-244
//   xyz_gradient_accum[vis] += ||viewspace_grad[vis, :2]||;  denom[vis] += 1   gaussian_model.py:677-679
__global__ void __launch_bounds__(256)
densification_stats_kernel(int P, int V, StatsViews views, float* __restrict__ accum, float* __restrict__ denom,
                           float* __restrict__ max_radii)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    // the views of a window in view order (the reference's loop order, train_gaussians.py:238-245): one
    // read-modify-write of the three statistics per Gaussian instead of one per (view, Gaussian)
    float a = 0.0f, n = 0.0f, m = 0.0f;
    bool any = false;
#pragma unroll 1
    for (int v = 0; v < V; ++v) {
        const int r = views.radii[v][i];
        if (r <= 0) continue;
        if (!any) { any = true; m = max_radii[i]; if (accum) { a = accum[i]; n = denom[i]; } }
        if (accum) {    // color_refinement updates max_radii2D only (train_gaussians.py:293-294)
            const float gx = views.vs_grad[v][3 * (size_t)i], gy = views.vs_grad[v][3 * (size_t)i + 1];
            a += sqrtf(gx * gx + gy * gy);
            n += 1.0f;
        }
        m = fmaxf(m, (float)r);
    }
    if (any) {
        max_radii[i] = m;
        if (accum) { accum[i] = a; denom[i] = n; }
    }
}

int launch_densification_stats(int32_t P, int32_t V, const StatsViews& views, float* accum, float* denom,
                               float* max_radii, hipStream_t stream)
{
    if (P == 0 || V == 0) return SPLATRASTER_OK;
    hipLaunchKernelGGL(densification_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, V, views,
                       accum, denom, max_radii);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

int launch_activate_fwd(int32_t P, int32_t K, int32_t deg, int32_t SC, int32_t E, const float* xyz,
                        const float* f_dc, const float* f_rest, const float* scaling, const float* rotation,
                        const float* opacity, const float* extra, const float* campos, float* scales,
                        float* rotations, float* opacities, float* colors, hipStream_t stream)
{
    if (P == 0) return SPLATRASTER_OK;
    const size_t quads = ((size_t)P * (3 + E) + 3) / 4;
    const size_t items = quads > (size_t)P ? quads : (size_t)P;
    hipLaunchKernelGGL(activate_fwd_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, P, K, deg, SC, E, xyz,
                       f_dc, f_rest, scaling, rotation, opacity, extra, campos, scales, rotations, opacities,
                       colors);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

int launch_activate_bwd(int32_t P, int32_t K, int32_t deg, int32_t SC, int32_t E, const float* xyz,
                        const float* f_dc, const float* f_rest, const float* scaling, const float* rotation,
                        const float* opacity, const float* campos, const float* g_scales,
                        const float* g_rotations, const float* g_opacities, const float* g_colors, float* d_xyz,
                        float* d_f_dc, float* d_f_rest, float* d_scaling, float* d_rotation, float* d_opacity,
                        float* d_extra, hipStream_t stream)
{
    if (P == 0) return SPLATRASTER_OK;
    const size_t quads = ((size_t)P * (3 + E) + 3) / 4;
    const size_t items = quads > (size_t)P ? quads : (size_t)P;
    hipLaunchKernelGGL(activate_bwd_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, P, K, deg, SC, E, xyz,
                       f_dc, f_rest, scaling, rotation, opacity, campos, g_scales, g_rotations, g_opacities,
                       g_colors, d_xyz, d_f_dc, d_f_rest, d_scaling, d_rotation, d_opacity, d_extra);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

}  // namespace sr
