// preprocess.hip — per-Gaussian forward projection (EWA splatting) and its backward.
//
// Replaces the preprocess stage of the rasterizer extension SplatLoc calls at
// gaussian_splatting/gaussian_renderer/__init__.py:117-126 (source un-vendored; algorithm
// per SURVEY.md §8a).  One thread per Gaussian, 256-thread blocks (4 wave64), pure
// streaming: every input array is read once with unit-stride-per-lane addresses, every
// output written once.  HBM-bound: 44+4C B read, 48 B written per Gaussian.
//
// THIS FILE IS COMPILED WITH -ffp-contract=off: the forward below must round exactly
// like the CPU oracle so that radii / tile counts / depth keys are bit-identical.
#include "activation_math.h"

namespace sr {

__device__ __forceinline__ int f2i_sat(float v)
{
    if (!(v > -1.0e9f)) v = -1.0e9f;
    if (!(v < 1.0e9f)) v = 1.0e9f;
    return (int)v;
}

__constant__ float SH_C0 = 0.28209479177387814f;
__constant__ float SH_C1 = 0.4886025119029199f;
__constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
__constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

__device__ __forceinline__ void cov3d_from_scale_rot(const float* s3, float mod, const float* q, float* c6)
{
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    float R[3][3];
    R[0][0] = 1.f - 2.f * (y * y + z * z);
    R[0][1] = 2.f * (x * y - r * z);
    R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z);
    R[1][1] = 1.f - 2.f * (x * x + z * z);
    R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y);
    R[2][1] = 2.f * (y * z + r * x);
    R[2][2] = 1.f - 2.f * (x * x + y * y);
    const float s[3] = {mod * s3[0], mod * s3[1], mod * s3[2]};
    float L[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) L[j][i] = R[j][i] * s[i];
    c6[0] = (L[0][0] * L[0][0] + L[0][1] * L[0][1]) + L[0][2] * L[0][2];
    c6[1] = (L[0][0] * L[1][0] + L[0][1] * L[1][1]) + L[0][2] * L[1][2];
    c6[2] = (L[0][0] * L[2][0] + L[0][1] * L[2][1]) + L[0][2] * L[2][2];
    c6[3] = (L[1][0] * L[1][0] + L[1][1] * L[1][1]) + L[1][2] * L[1][2];
    c6[4] = (L[1][0] * L[2][0] + L[1][1] * L[2][1]) + L[1][2] * L[2][2];
    c6[5] = (L[2][0] * L[2][0] + L[2][1] * L[2][1]) + L[2][2] * L[2][2];
}

__device__ void sh_to_rgb(int deg, const float* sh, float px, float py, float pz, const float* campos,
                          float* rgb, uint8_t* clamped)
{
    const float dx = px - campos[0], dy = py - campos[1], dz = pz - campos[2];
    const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float x = dx / len, y = dy / len, z = dz / len;
    for (int c = 0; c < 3; ++c) {
        float res = SH_C0 * sh[0 * 3 + c];
        if (deg > 0) {
            res = res - SH_C1 * y * sh[1 * 3 + c] + SH_C1 * z * sh[2 * 3 + c] - SH_C1 * x * sh[3 * 3 + c];
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * sh[4 * 3 + c] + SH_C2[1] * yz * sh[5 * 3 + c] +
                      SH_C2[2] * (2.f * zz - xx - yy) * sh[6 * 3 + c] + SH_C2[3] * xz * sh[7 * 3 + c] +
                      SH_C2[4] * (xx - yy) * sh[8 * 3 + c];
                if (deg > 2) {
                    res = res + SH_C3[0] * y * (3.f * xx - yy) * sh[9 * 3 + c] +
                          SH_C3[1] * xy * z * sh[10 * 3 + c] +
                          SH_C3[2] * y * (4.f * zz - xx - yy) * sh[11 * 3 + c] +
                          SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * sh[12 * 3 + c] +
                          SH_C3[4] * x * (4.f * zz - xx - yy) * sh[13 * 3 + c] +
                          SH_C3[5] * z * (xx - yy) * sh[14 * 3 + c] +
                          SH_C3[6] * x * (xx - 3.f * yy) * sh[15 * 3 + c];
                }
            }
        }
        res += 0.5f;
        clamped[c] = (uint8_t)(res < 0.f);
        rgb[c] = res < 0.f ? 0.f : res;
    }
}

// One thread per Gaussian, looping over the V views of the window: the view-independent work (loading the
// parameters, the 3D covariance) is done once, then every view projects it with its own camera.  Row
// g = v * P + i of the geometry buffer is Gaussian i seen from view v.
__global__ void __launch_bounds__(PREPROCESS_BLOCK)
preprocess_kernel(int P, int V, int W, int H, float scale_modifier, int sh_degree, int sh_coeffs, WinCams cams,
                  const float* __restrict__ means3D, const float* __restrict__ shs,
                  const float* __restrict__ opacities, const float* __restrict__ scales,
                  const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
                  float4* __restrict__ rec, uint32_t* __restrict__ tiles_touched,
                  float* __restrict__ rgb, uint8_t* __restrict__ clamped,
                  uint32_t* __restrict__ block_tiles /*[V][gridDim.x] per-(view, block) sums of tiles_touched*/,
                  uint32_t* __restrict__ sort_keys, uint32_t* __restrict__ sort_vals /*depth-sort input*/,
                  uint32_t* __restrict__ zero0, uint32_t nzero0, uint32_t* __restrict__ zero1, uint32_t nzero1,
                  RawFwd raw /*raw.scaling != null: the parameter activations run here (common.h)*/)
{
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    // state words of the depth sort's look-back and of the one-pass scan must be zero when those
    // kernels start: cleared here instead of by two extra memset launches
    for (uint32_t w = (uint32_t)gi; w < nzero0; w += gridDim.x * blockDim.x) zero0[w] = 0u;
    for (uint32_t w = (uint32_t)gi; w < nzero1; w += gridDim.x * blockDim.x) zero1[w] = 0u;
    const bool live = gi < P;
    const int i = live ? gi : P - 1;  // padding lanes recompute the last Gaussian and store nothing
    const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
    float opacity;
    float c6[6];
    if (raw.scaling) {
        // raw parameters: exp / normalize / sigmoid / SH degree 0 + clamp + [rgb | extra] packing with activations.hip's own arithmetic
        // (activation_math.h: bit-identical to activate_fwd_kernel in front of the plain kernel); the activated values are used here
        // and written once for the later stages (the compositing kernels gather the colour rows, the backward reads scales / rotations)
        const float s3[3] = {expf(raw.scaling[3 * (size_t)i]), expf(raw.scaling[3 * (size_t)i + 1]), expf(raw.scaling[3 * (size_t)i + 2])};
        const float4 qv = act_normalize(reinterpret_cast<const float4*>(raw.rotation)[i]);
        opacity = act_sigmoid(raw.opacity[i]);
        const float q[4] = {qv.x, qv.y, qv.z, qv.w};
        cov3d_from_scale_rot(s3, scale_modifier, q, c6);
        if (live) {
#pragma unroll
            for (int k = 0; k < 3; ++k) raw.scales[3 * (size_t)i + k] = s3[k];
            reinterpret_cast<float4*>(raw.rotations)[i] = qv;
            raw.opacities[i] = opacity;
            const int CW = 3 + raw.E;
            float* crow = raw.colors + (size_t)i * CW;
#pragma unroll
            for (int c = 0; c < 3; ++c) crow[c] = fmaxf(act_rgb_raw_deg0(raw.f_dc[3 * (size_t)i + c]), 0.0f);
            for (int e = 0; e < raw.E; ++e) crow[3 + e] = raw.extra[(size_t)i * raw.E + e];
        }
    } else {
    opacity = opacities[i];
    if (cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * i + k];
    } else {
        const float s3[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        const float4 qv = reinterpret_cast<const float4*>(rotations)[i];
        const float q[4] = {qv.x, qv.y, qv.z, qv.w};
        cov3d_from_scale_rot(s3, scale_modifier, q, c6);
    }
    }
    const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    __shared__ uint32_t s_part[MAX_VIEWS][PREPROCESS_BLOCK / WAVE];

#pragma unroll 1
    for (int v = 0; v < V; ++v) {
        // camera tensors: wave-uniform addresses -> scalar loads, SGPR-resident
        const float* __restrict__ view = cams.view[v];
        const float* __restrict__ proj = cams.proj[v];
        const float* __restrict__ campos_p = cams.campos[v];
        const float tanfovx = cams.tanfovx[v], tanfovy = cams.tanfovy[v];
        float Vm[16], PM[16], campos[3];
#pragma unroll
        for (int k = 0; k < 16; ++k) { Vm[k] = view[k]; PM[k] = proj[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) campos[k] = campos_p ? campos_p[k] : 0.f;
        int32_t out_radius = 0;
        uint32_t out_tiles = 0;
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = make_float4(0.f, 0.f, 0.f, 0.f);

        const float tx0 = ((Vm[0] * px + Vm[4] * py) + Vm[8] * pz) + Vm[12];
        const float ty0 = ((Vm[1] * px + Vm[5] * py) + Vm[9] * pz) + Vm[13];
        const float tz = ((Vm[2] * px + Vm[6] * py) + Vm[10] * pz) + Vm[14];
        if (live && tz > NEAR_Z) {
            const float hx = ((PM[0] * px + PM[4] * py) + PM[8] * pz) + PM[12];
            const float hy = ((PM[1] * px + PM[5] * py) + PM[9] * pz) + PM[13];
            const float hw = ((PM[3] * px + PM[7] * py) + PM[11] * pz) + PM[15];
            const float p_w = 1.0f / (hw + 0.0000001f);
            const float ndc_x = hx * p_w, ndc_y = hy * p_w;
            const float focal_x = (float)W / (2.0f * tanfovx);
            const float focal_y = (float)H / (2.0f * tanfovy);
            const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
            const float txtz = tx0 / tz, tytz = ty0 / tz;
            const float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
            const float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
            const float j00 = focal_x / tz, j02 = -(focal_x * tx) / (tz * tz);
            const float j11 = focal_y / tz, j12 = -(focal_y * ty) / (tz * tz);
            float A0[3], A1[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                A0[c] = j00 * Vm[4 * c + 0] + j02 * Vm[4 * c + 2];
                A1[c] = j11 * Vm[4 * c + 1] + j12 * Vm[4 * c + 2];
            }
            float SA0[3], SA1[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                SA0[j] = (S[j][0] * A0[0] + S[j][1] * A0[1]) + S[j][2] * A0[2];
                SA1[j] = (S[j][0] * A1[0] + S[j][1] * A1[1]) + S[j][2] * A1[2];
            }
            const float a = ((A0[0] * SA0[0] + A0[1] * SA0[1]) + A0[2] * SA0[2]) + DILATION;
            const float b = (A0[0] * SA1[0] + A0[1] * SA1[1]) + A0[2] * SA1[2];
            const float c = ((A1[0] * SA1[0] + A1[1] * SA1[1]) + A1[2] * SA1[2]) + DILATION;
            const float det = a * c - b * b;
            if (det != 0.0f) {
                const float det_inv = 1.0f / det;
                const float mid = 0.5f * (a + c);
                const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
                const float lambda1 = mid + disc, lambda2 = mid - disc;
                const float rad_f = ceilf(3.0f * sqrtf(fmaxf(lambda1, lambda2)));
                const int my_radius = f2i_sat(rad_f);
                const float pixx = ((ndc_x + 1.0f) * (float)W - 1.0f) * 0.5f;
                const float pixy = ((ndc_y + 1.0f) * (float)H - 1.0f) * 0.5f;
                const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
                const float rf = (float)my_radius;
                const int rminx = min(gx, max(0, f2i_sat((pixx - rf) / (float)TILE)));
                const int rminy = min(gy, max(0, f2i_sat((pixy - rf) / (float)TILE)));
                const int rmaxx = min(gx, max(0, f2i_sat((pixx + rf + (float)(TILE - 1)) / (float)TILE)));
                const int rmaxy = min(gy, max(0, f2i_sat((pixy + rf + (float)(TILE - 1)) / (float)TILE)));
                const int area = (rmaxx - rminx) * (rmaxy - rminy);
                if (area > 0) {
                    if (shs) {   // single-view calls only (V == 1: row == i)
                        sh_to_rgb(sh_degree, shs + (size_t)i * sh_coeffs * 3, px, py, pz, campos,
                                  rgb + 3 * (size_t)i, clamped + 3 * (size_t)i);
                    }
                    out_radius = my_radius;
                    out_tiles = (uint32_t)area;
                    r0 = make_float4(pixx, pixy, tz, (float)my_radius);
                    r1 = make_float4(c * det_inv, -b * det_inv, a * det_inv, opacity);
                }
            }
        }
        if (live) {
            const size_t g = (size_t)v * P + i;
            cams.radii[v][i] = out_radius;
            tiles_touched[g] = out_tiles;
            rec[2 * g] = r0;
            rec[2 * g + 1] = r1;
            // input of the depth sort: view depth > 0.2 for every visible Gaussian, so its IEEE bits
            // order like the value; culled rows sort to the end
            if (sort_keys) {   // (null: the binned front end, binsort.hip, orders the instances per tile instead)
                sort_keys[g] = out_radius > 0 ? __float_as_uint(tz) : 0xFFFFFFFFu;
                sort_vals[g] = (uint32_t)g | (min(out_tiles, 255u) << 24);   // (rows < 2^24; the count rides along for the scan)
            }
        }
        uint32_t t = out_tiles;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) t += (uint32_t)__shfl_xor((int)t, d, WAVE);
        if ((threadIdx.x & (WAVE - 1)) == 0) s_part[v][threadIdx.x / WAVE] = t;
    }
    // The instance count R = sum of tiles_touched is needed on the HOST (it sizes the binning
    // buffer).  Per-(view, block) sums written here (no atomics, nothing to zero) are copied out and
    // added up by the host while the depth sort and the scan are still executing (capi.hip).
    __syncthreads();
    if ((int)threadIdx.x < V) {
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < PREPROCESS_BLOCK / WAVE; ++k) sum += s_part[threadIdx.x][k];
        block_tiles[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = sum;
    }
}

int launch_preprocess(const splatraster_settings& s, int32_t P, int32_t V, const WinCams& cams, const float* means3D,
                      const float* shs, const float* opacities, const float* scales, const float* rotations,
                      const float* cov3D_precomp, GeomView g, uint32_t* zero0, uint32_t nzero0,
                      uint32_t* zero1, uint32_t nzero1, hipStream_t stream, bool depth_keys, const RawFwd* raw)
{
    if (P == 0) return SPLATRASTER_OK;
    const int blocks = preprocess_blocks(P);
    hipLaunchKernelGGL(preprocess_kernel, dim3(blocks), dim3(PREPROCESS_BLOCK), 0, stream, P, V, s.image_width,
                       s.image_height, s.scale_modifier, s.sh_degree, s.sh_coeffs, cams,
                       means3D, shs, opacities, scales, rotations, cov3D_precomp, g.rec,
                       g.tiles_touched, g.rgb, g.clamped, g.block_tiles, depth_keys ? g.sort_keys : nullptr,
                       depth_keys ? g.depth_order : nullptr, zero0, nzero0, zero1, nzero1, (raw ? *raw : RawFwd{}));
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view,
                                    uint8_t* __restrict__ present)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float v2 = view[2], v6 = view[6], v10 = view[10], v14 = view[14];
    const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
    const float tz = ((v2 * px + v6 * py) + v10 * pz) + v14;
    present[i] = (uint8_t)(tz > NEAR_Z);
}

int launch_mark_visible(int32_t P, const float* means3D, const float* view, uint8_t* present,
                        hipStream_t stream)
{
    if (P == 0) return SPLATRASTER_OK;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, means3D, view,
                       present);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

}  // namespace sr
