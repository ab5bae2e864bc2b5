// preprocess_bwd.hip — per-Gaussian backward of the projection (SURVEY.md §8a
// "PREPROCESS bwd"): conic -> cov2D -> (Sigma3, view-space mean) -> scale / quaternion /
// mean3D, NDC mean2D -> mean3D through the projection, depth -> mean3D, SH -> (sh, mean3D).
//
// One thread per Gaussian; reads the 32-byte record of gradient MOMENTS accumulated by the
// compositing backward, with E = G dL/dalpha and d = mu2D - pixel summed over (pixel, tile):
//   (sum E dx, sum E dy, sum E dx^2, sum E dx dy, sum E dy^2, sum E, sum w g_D, pad)
// turns them into dL/dmean2D (NDC), dL/dconic, dL/dopacity, dL/ddepth with the per-Gaussian
// factors (conic, opacity, 0.5 W, 0.5 H), and chains through the projection.  Plus the forward
// inputs; writes every gradient tensor once.
// HBM-bound: ~100 B read, ~70 B written per Gaussian.
#include "composite_common.h"
#include "activation_math.h"

namespace sr {

__constant__ float BSH_C0 = 0.28209479177387814f;
__constant__ float BSH_C1 = 0.4886025119029199f;
__constant__ float BSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f};
__constant__ float BSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                -0.5900435899266435f};

__device__ void sh_backward(int deg, int M, const float* sh, float* dsh, const uint8_t* clamped,
                            const float* drgb_in, float ox, float oy, float oz, float* dmean)
{
    const float len = sqrtf(ox * ox + oy * oy + oz * oz);
    const float x = ox / len, y = oy / len, z = oz / len;
    float ddir[3] = {0.f, 0.f, 0.f};
    for (int ch = 0; ch < 3; ++ch) {
        const float d = clamped[ch] ? 0.f : drgb_in[ch];
        float ddx = 0.f, ddy = 0.f, ddz = 0.f;
#define SHV(k) sh[(k) * 3 + ch]
        dsh[0 * 3 + ch] = BSH_C0 * d;
        if (deg > 0) {
            dsh[1 * 3 + ch] = -BSH_C1 * y * d;
            dsh[2 * 3 + ch] = BSH_C1 * z * d;
            dsh[3 * 3 + ch] = -BSH_C1 * x * d;
            ddx = -BSH_C1 * SHV(3);
            ddy = -BSH_C1 * SHV(1);
            ddz = BSH_C1 * SHV(2);
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                dsh[4 * 3 + ch] = BSH_C2[0] * xy * d;
                dsh[5 * 3 + ch] = BSH_C2[1] * yz * d;
                dsh[6 * 3 + ch] = BSH_C2[2] * (2.f * zz - xx - yy) * d;
                dsh[7 * 3 + ch] = BSH_C2[3] * xz * d;
                dsh[8 * 3 + ch] = BSH_C2[4] * (xx - yy) * d;
                ddx += BSH_C2[0] * y * SHV(4) + BSH_C2[2] * 2.f * -x * SHV(6) + BSH_C2[3] * z * SHV(7) + BSH_C2[4] * 2.f * x * SHV(8);
                ddy += BSH_C2[0] * x * SHV(4) + BSH_C2[1] * z * SHV(5) + BSH_C2[2] * 2.f * -y * SHV(6) + BSH_C2[4] * 2.f * -y * SHV(8);
                ddz += BSH_C2[1] * y * SHV(5) + BSH_C2[2] * 4.f * z * SHV(6) + BSH_C2[3] * x * SHV(7);
                if (deg > 2) {
                    dsh[9 * 3 + ch] = BSH_C3[0] * y * (3.f * xx - yy) * d;
                    dsh[10 * 3 + ch] = BSH_C3[1] * xy * z * d;
                    dsh[11 * 3 + ch] = BSH_C3[2] * y * (4.f * zz - xx - yy) * d;
                    dsh[12 * 3 + ch] = BSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * d;
                    dsh[13 * 3 + ch] = BSH_C3[4] * x * (4.f * zz - xx - yy) * d;
                    dsh[14 * 3 + ch] = BSH_C3[5] * z * (xx - yy) * d;
                    dsh[15 * 3 + ch] = BSH_C3[6] * x * (xx - 3.f * yy) * d;
                    ddx += BSH_C3[0] * SHV(9) * 6.f * xy + BSH_C3[1] * SHV(10) * yz + BSH_C3[2] * SHV(11) * -2.f * xy +
                           BSH_C3[3] * SHV(12) * -6.f * xz + BSH_C3[4] * SHV(13) * (-3.f * xx + 4.f * zz - yy) +
                           BSH_C3[5] * SHV(14) * 2.f * xz + BSH_C3[6] * SHV(15) * 3.f * (xx - yy);
                    ddy += BSH_C3[0] * SHV(9) * 3.f * (xx - yy) + BSH_C3[1] * SHV(10) * xz +
                           BSH_C3[2] * SHV(11) * (-3.f * yy + 4.f * zz - xx) + BSH_C3[3] * SHV(12) * -6.f * yz +
                           BSH_C3[4] * SHV(13) * -2.f * xy + BSH_C3[5] * SHV(14) * -2.f * yz +
                           BSH_C3[6] * SHV(15) * -6.f * xy;
                    ddz += BSH_C3[1] * SHV(10) * xy + BSH_C3[2] * SHV(11) * 8.f * yz +
                           BSH_C3[3] * SHV(12) * 3.f * (2.f * zz - xx - yy) + BSH_C3[4] * SHV(13) * 8.f * xz +
                           BSH_C3[5] * SHV(14) * (xx - yy);
                }
            }
        }
#undef SHV
        for (int k = (deg + 1) * (deg + 1); k < M; ++k) dsh[k * 3 + ch] = 0.f;
        ddir[0] += ddx * d;
        ddir[1] += ddy * d;
        ddir[2] += ddz * d;
    }
    const float dot = x * ddir[0] + y * ddir[1] + z * ddir[2];
    dmean[0] += (ddir[0] - x * dot) / len;
    dmean[1] += (ddir[1] - y * dot) / len;
    dmean[2] += (ddir[2] - z * dot) / len;
}

// One thread per Gaussian, looping over the V views of the window: the contributions of every view (accumulator
// row g = v * P + i, forward record rec[g], the view's camera) are summed in view order into ONE set of parameter
// gradients, written once — no per-view gradient tensors, no accumulation kernels, a deterministic sum.
// dL/dmeans2D stays per view (GaussianModel.add_densification_stats reads it per view).
template <bool POSE, bool RAW = false>
__global__ void __launch_bounds__(256)
preprocess_bwd_kernel(int P, int V, int W, int H, float mod, int sh_degree, int M, WinCams cams, WinGrad grads,
                      const float* __restrict__ means3D, const float* __restrict__ shs,
                      const float* __restrict__ scales, const float* __restrict__ rotations,
                      const float* __restrict__ cov3D_precomp,
                      const uint8_t* __restrict__ clamped,
                      const float4* __restrict__ rec, const float* __restrict__ gacc, int C, int GROW, int MO,
                      float* __restrict__ dL_dmeans3D,
                      float* __restrict__ dL_dopacities, float* __restrict__ dL_dscales,
                      float* __restrict__ dL_drotations, float* __restrict__ dL_dcov3D,
                      float* __restrict__ dL_dshs, float* __restrict__ dL_dview, float* __restrict__ dL_dproj,
                      float* __restrict__ dL_dcampos, float* __restrict__ pose_acc /*POSE: zeroed sets + ticket (common.h)*/,
                      RawBwd raw /*RAW: the raw parameters and their gradient outputs (common.h)*/)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // pose partials of this Gaussian: dV[4c + r] (r < 3), dPM[4c + k] (k = 0, 1, 3), dcampos   (V == 1 only)
    float pose[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) pose[k] = 0.f;
    float dmean[3] = {0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float drot[4] = {0.f, 0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dop = 0.f;
    float csum[4] = {0.f, 0.f, 0.f, 0.f};   // RAW: dL/dcolours of this Gaussian (a view that does not see it left its row at +0: skipping it is exact)
    if (i < P) {
    bool any_visible = false;
    const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
    // 3D covariance (recomputed; same formula as the forward) — view independent
    float c6[6];
    float Rm[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, sc[3] = {0.f, 0.f, 0.f};
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * i + k];
    } else {
        qv = reinterpret_cast<const float4*>(rotations)[i];
        const float r = qv.x, x = qv.y, y = qv.z, z = qv.w;
        Rm[0][0] = 1.f - 2.f * (y * y + z * z); Rm[0][1] = 2.f * (x * y - r * z); Rm[0][2] = 2.f * (x * z + r * y);
        Rm[1][0] = 2.f * (x * y + r * z); Rm[1][1] = 1.f - 2.f * (x * x + z * z); Rm[1][2] = 2.f * (y * z - r * x);
        Rm[2][0] = 2.f * (x * z - r * y); Rm[2][1] = 2.f * (y * z + r * x); Rm[2][2] = 1.f - 2.f * (x * x + y * y);
        sc[0] = mod * scales[3 * i]; sc[1] = mod * scales[3 * i + 1]; sc[2] = mod * scales[3 * i + 2];
        float L[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) L[j][k] = Rm[j][k] * sc[k];
        c6[0] = L[0][0] * L[0][0] + L[0][1] * L[0][1] + L[0][2] * L[0][2];
        c6[1] = L[0][0] * L[1][0] + L[0][1] * L[1][1] + L[0][2] * L[1][2];
        c6[2] = L[0][0] * L[2][0] + L[0][1] * L[2][1] + L[0][2] * L[2][2];
        c6[3] = L[1][0] * L[1][0] + L[1][1] * L[1][1] + L[1][2] * L[1][2];
        c6[4] = L[1][0] * L[2][0] + L[1][1] * L[2][1] + L[1][2] * L[2][2];
        c6[5] = L[2][0] * L[2][0] + L[2][1] * L[2][1] + L[2][2] * L[2][2];
    }
    const float S3[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    float G3s[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};   // dL/dSigma3 summed over the views

#pragma unroll 1
    for (int v = 0; v < V; ++v) {
    const float* __restrict__ view = cams.view[v];
    const float* __restrict__ proj = cams.proj[v];
    const float* __restrict__ campos_p = cams.campos[v];
    const float tanfovx = cams.tanfovx[v], tanfovy = cams.tanfovy[v];
    float Vm[16], PM[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { Vm[k] = view[k]; PM[k] = proj[k]; }
    const size_t gr = (size_t)v * P + i;   // row of (view, Gaussian)
    float dm2x = 0.f, dm2y = 0.f;
    const bool visible = cams.radii[v][i] > 0;
    if (visible) {
        any_visible = true;
        const float* mrow = gacc + gr * GROW + MO;  // moment record of this row
        if constexpr (RAW) {    // the row's colour columns (C <= 4: the moments' own 64-byte line), summed in view order like gather_dcolors_kernel
            const float4 cr = *reinterpret_cast<const float4*>(gacc + gr * GROW);
            csum[0] += cr.x; csum[1] += cr.y; csum[2] += cr.z; csum[3] += cr.w;
        }
        const float4 g0 = make_float4(mrow[0], mrow[1], mrow[2], mrow[3]);
        const float4 g1 = make_float4(mrow[4], mrow[5], mrow[6], 0.f);
        const float4 con = rec[2 * gr + 1];  // conic a, b, c, opacity of the forward
        // power = -1/2 (A dx^2 + C dy^2) - B dx dy, alpha = o G:
        dm2x = -0.5f * (float)W * con.w * (con.x * g0.x + con.y * g0.y);
        dm2y = -0.5f * (float)H * con.w * (con.z * g0.y + con.y * g0.x);
        const float gA = -0.5f * con.w * g0.z, gB = -con.w * g0.w, gC = -0.5f * con.w * g1.x;
        dop += g1.y;
        const float gdepth = g1.z;
        const float tx0 = Vm[0] * px + Vm[4] * py + Vm[8] * pz + Vm[12];
        const float ty0 = Vm[1] * px + Vm[5] * py + Vm[9] * pz + Vm[13];
        const float tz = Vm[2] * px + Vm[6] * py + Vm[10] * pz + Vm[14];
        const float focal_x = (float)W / (2.0f * tanfovx), focal_y = (float)H / (2.0f * tanfovy);
        const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
        const float txtz = tx0 / tz, tytz = ty0 / tz;
        const float xg = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const float yg = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
        const float itz = 1.0f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
        const float J00 = focal_x * itz, J02 = -(focal_x * tx) * itz2;
        const float J11 = focal_y * itz, J12 = -(focal_y * ty) * itz2;
        // Wv[r][c] = V[4c + r]
        float A0[3], A1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            A0[c] = J00 * Vm[4 * c + 0] + J02 * Vm[4 * c + 2];
            A1[c] = J11 * Vm[4 * c + 1] + J12 * Vm[4 * c + 2];
        }
        float SA0[3], SA1[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            SA0[j] = S3[j][0] * A0[0] + S3[j][1] * A0[1] + S3[j][2] * A0[2];
            SA1[j] = S3[j][0] * A1[0] + S3[j][1] * A1[1] + S3[j][2] * A1[2];
        }
        const float a = A0[0] * SA0[0] + A0[1] * SA0[1] + A0[2] * SA0[2] + DILATION;
        const float b = A0[0] * SA1[0] + A0[1] * SA1[1] + A0[2] * SA1[2];
        const float c = A1[0] * SA1[0] + A1[1] * SA1[1] + A1[2] * SA1[2] + DILATION;
        const float det = a * c - b * b;
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        if (det != 0.f) {
            const float d2 = 1.0f / (det * det);
            dL_da = (-c * c * gA + b * c * gB - b * b * gC) * d2;
            dL_db = (2.f * b * c * gA - (det + 2.f * b * b) * gB + 2.f * a * b * gC) * d2;
            dL_dc = (-b * b * gA + a * b * gB - a * a * gC) * d2;
        }
        const float G2[2][2] = {{dL_da, 0.5f * dL_db}, {0.5f * dL_db, dL_dc}};
        // dL/dSigma3 (full symmetric) = A^T G2 A — linear in the view's contribution, chained to scale / quaternion
        // once after the loop
        float GA0[3], GA1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            GA0[k] = G2[0][0] * A0[k] + G2[0][1] * A1[k];
            GA1[k] = G2[1][0] * A0[k] + G2[1][1] * A1[k];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) G3s[j][k] += A0[j] * GA0[k] + A1[j] * GA1[k];
        // dL/dJ = 2 G2 J Sigma_v with J Sigma_v = (A Sigma3) Wv^T ; (A Sigma3)[r][k] = SA_r[k]
        float JS0[3], JS1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // column k of Sigma_v side: sum_c SA[c] * Wv[k][c]
            JS0[k] = SA0[0] * Vm[0 + k] + SA0[1] * Vm[4 + k] + SA0[2] * Vm[8 + k];
            JS1[k] = SA1[0] * Vm[0 + k] + SA1[1] * Vm[4 + k] + SA1[2] * Vm[8 + k];
        }
        const float dJ00 = 2.f * (G2[0][0] * JS0[0] + G2[0][1] * JS1[0]);
        const float dJ02 = 2.f * (G2[0][0] * JS0[2] + G2[0][1] * JS1[2]);
        const float dJ11 = 2.f * (G2[1][0] * JS0[1] + G2[1][1] * JS1[1]);
        const float dJ12 = 2.f * (G2[1][0] * JS0[2] + G2[1][1] * JS1[2]);
        const float dtx = xg * (-focal_x * itz2 * dJ02);
        const float dty = yg * (-focal_y * itz2 * dJ12);
        const float dtz = -focal_x * itz2 * dJ00 - focal_y * itz2 * dJ11 + 2.f * focal_x * tx * itz3 * dJ02 +
                          2.f * focal_y * ty * itz3 * dJ12;
#pragma unroll
        for (int k = 0; k < 3; ++k)  // Wv^T [dtx dty dtz]: Wv[r][k] = V[4k + r]
            dmean[k] += Vm[4 * k + 0] * dtx + Vm[4 * k + 1] * dty + Vm[4 * k + 2] * (dtz + gdepth);
        if (POSE) {
            // t = Wv p + trans (V[4c + r] multiplies p[c] into t[r]); cov2D = A Sigma3 A^T with
            // A = J Wv: dL/dA = 2 G2 A Sigma3, dL/dWv = J^T dL/dA
            const float dt[3] = {dtx, dty, dtz + gdepth};
            const float pp[3] = {px, py, pz};
            float dA0[3], dA1[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                dA0[k] = 2.f * (G2[0][0] * SA0[k] + G2[0][1] * SA1[k]);
                dA1[k] = 2.f * (G2[1][0] * SA0[k] + G2[1][1] * SA1[k]);
            }
            // J = [[J00, 0, J02], [0, J11, J12]]
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                pose[3 * c + 0] = dt[0] * pp[c] + J00 * dA0[c];
                pose[3 * c + 1] = dt[1] * pp[c] + J11 * dA1[c];
                pose[3 * c + 2] = dt[2] * pp[c] + J02 * dA0[c] + J12 * dA1[c];
            }
            pose[9] = dt[0];
            pose[10] = dt[1];
            pose[11] = dt[2];
        }
        // NDC mean2D -> mean3D
        const float hx = PM[0] * px + PM[4] * py + PM[8] * pz + PM[12];
        const float hy = PM[1] * px + PM[5] * py + PM[9] * pz + PM[13];
        const float hw = PM[3] * px + PM[7] * py + PM[11] * pz + PM[15];
        const float mw = 1.0f / (hw + 0.0000001f);
        const float mul1 = hx * mw * mw, mul2 = hy * mw * mw;
        dmean[0] += (PM[0] * mw - PM[3] * mul1) * dm2x + (PM[1] * mw - PM[3] * mul2) * dm2y;
        dmean[1] += (PM[4] * mw - PM[7] * mul1) * dm2x + (PM[5] * mw - PM[7] * mul2) * dm2y;
        dmean[2] += (PM[8] * mw - PM[11] * mul1) * dm2x + (PM[9] * mw - PM[11] * mul2) * dm2y;
        if (POSE) {
            const float dh[3] = {dm2x * mw, dm2y * mw, -(mul1 * dm2x + mul2 * dm2y)};  // d/d(hx, hy, hw)
            const float p4[4] = {px, py, pz, 1.f};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 3; ++j) pose[12 + 3 * c + j] = dh[j] * p4[c];
        }
        if (shs) {   // single-view calls only
            const float sm0 = dmean[0], sm1 = dmean[1], sm2 = dmean[2];
            sh_backward(sh_degree, M, shs + (size_t)i * 3 * M, dL_dshs + (size_t)i * 3 * M, clamped + 3 * (size_t)i,
                        gacc + gr * GROW, px - campos_p[0], py - campos_p[1], pz - campos_p[2], dmean);
            if (POSE) {  // direction = normalize(p - campos)
                pose[24] = -(dmean[0] - sm0);
                pose[25] = -(dmean[1] - sm1);
                pose[26] = -(dmean[2] - sm2);
            }
        }
    } else if (shs && dL_dshs) {
        for (int k = 0; k < 3 * M; ++k) dL_dshs[(size_t)i * 3 * M + k] = 0.f;
    }
    float* __restrict__ dm2 = grads.dL_dmeans2D[v];
    dm2[3 * i] = dm2x;
    dm2[3 * i + 1] = dm2y;
    dm2[3 * i + 2] = 0.f;
    }  // views

    if (any_visible) {
        if (cov3D_precomp) {
            dcov[0] = G3s[0][0]; dcov[1] = 2.f * G3s[0][1]; dcov[2] = 2.f * G3s[0][2];
            dcov[3] = G3s[1][1]; dcov[4] = 2.f * G3s[1][2]; dcov[5] = G3s[2][2];
        } else {
            // Sigma3 = L L^T, L = R diag(s)  =>  dL/dL = 2 G3 L
            float dR[3][3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float ds = 0.f;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float dLjk = 2.f * (G3s[j][0] * Rm[0][k] + G3s[j][1] * Rm[1][k] + G3s[j][2] * Rm[2][k]) * sc[k];
                    ds += dLjk * Rm[j][k];
                    dR[j][k] = dLjk * sc[k];
                }
                dscale[k] = ds * mod;
            }
            const float r = qv.x, x = qv.y, y = qv.z, z = qv.w;
            drot[0] = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
            drot[1] = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - 2.f * x * dR[2][2]);
            drot[2] = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - 2.f * y * dR[2][2]);
            drot[3] = 2.f * (-2.f * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2.f * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
        }
    }
    }  // i < P
    if (POSE) {
        // 27 partials summed over the wave with the packed butterfly, then over the block in LDS, then one atomic per value per
        // block into set blockIdx.x % POSE_SETS (every block on the same 27 words queued the atomics of 2 000 blocks on three
        // lines: 13 us of this kernel at 500k Gaussians); the last block to take a ticket sums the sets and writes the outputs,
        // all 16 + 16 + 3 entries of them — nothing to zero beforehand but the sets, which lie behind the accumulator rows
        __shared__ float s_pose[4][32];
        __shared__ bool s_last;
        const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
        const float tot = wave_reduce_pack<27>(pose, lane);
        const int slot = (int)(__brev((unsigned)lane) >> 26);
        if (slot < 27) s_pose[w][slot] = tot;
        __syncthreads();
        if (threadIdx.x < 27) {
            const int k = threadIdx.x;
            const float v = s_pose[0][k] + s_pose[1][k] + s_pose[2][k] + s_pose[3][k];
            const float before = atomicAdd(&pose_acc[(blockIdx.x & (POSE_SETS - 1)) * POSE_SET_FLOATS + k], v);
            asm volatile("" ::"v"(before));     // (returned: the addition is done at the memory side before the ticket below)
        }
        __syncthreads();
        // two-level ticket (2 000 increments of ONE word would queue for ~40 us): the set's own counter in the last word of its
        // line, then — by the last block of every set — the counter behind the sets
        if (threadIdx.x == 0) {
            const unsigned q = blockIdx.x & (POSE_SETS - 1);
            const unsigned in_set = (gridDim.x - q + (POSE_SETS - 1)) / POSE_SETS;       // blocks that add to set q
            const unsigned nsets = gridDim.x < (unsigned)POSE_SETS ? gridDim.x : (unsigned)POSE_SETS;
            unsigned* set_ticket = reinterpret_cast<unsigned*>(pose_acc + q * POSE_SET_FLOATS + (POSE_SET_FLOATS - 1));
            unsigned* ticket = reinterpret_cast<unsigned*>(pose_acc + POSE_SETS * POSE_SET_FLOATS);
            // release / acquire at agent scope on both tickets: this block's additions (ordered before this thread by the
            // barrier above) happen-before the winner's loads of the sets by the memory model, not only by today's codegen
            bool last = false;
            if (__hip_atomic_fetch_add(set_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == in_set - 1)
                last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nsets - 1;
            s_last = last;
        }
        __syncthreads();
        if (s_last && threadIdx.x < 35) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int e = threadIdx.x;      // output entry: dV[0..15], dPM[16..31], dcampos[32..34]
            int k = -1;                     // its partial (dV[4c + r], r < 3: 3c + r; dPM[4c + j], j = 0, 1, 3: 12 + 3c + (j == 3 ? 2 : j))
            if (e < 16) { if ((e & 3) < 3) k = 3 * (e >> 2) + (e & 3); }
            else if (e < 32) { const int j = (e - 16) & 3; if (j != 2) k = 12 + 3 * ((e - 16) >> 2) + (j == 3 ? 2 : j); }
            else k = 24 + (e - 32);
            float sum = 0.0f;
            if (k >= 0) {
                for (int q = 0; q < POSE_SETS; ++q)
                    sum += __hip_atomic_load(&pose_acc[q * POSE_SET_FLOATS + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (e < 16) dL_dview[e] = sum;
            else if (e < 32) dL_dproj[e - 16] = sum;
            else if (dL_dcampos) dL_dcampos[e - 32] = sum;
        }
    }
    // the per-Gaussian gradients are written AFTER the camera sums went out: the set atomics return into registers and nothing
    // but loads precedes them — behind the stores below, waiting for them would wait for the stores as well (vmcnt counts both)
    if (i < P) {
#pragma unroll
    for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = dmean[k];
    if constexpr (RAW) {
        // the chain through the activations, with activations.hip's own arithmetic (activation_math.h): bit-identical to
        // gather_dcolors_kernel + activate_bwd_kernel behind the plain kernel
        if (raw.reg_row_grad) {     // torch's `d_sca + ((w * out[1]) * row_grad)`: three separately rounded operations (activation_math.h)
            const float o1 = raw.reg_out[1], rg = raw.reg_row_grad[i];
#pragma unroll
            for (int k = 0; k < 3; ++k) dscale[k] = act_add_scaled(dscale[k], raw.reg_weight, o1, rg);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) raw.d_scaling[3 * i + k] = dscale[k] * expf(raw.scaling[3 * (size_t)i + k]);
        reinterpret_cast<float4*>(raw.d_rotation)[i] =
            act_normalize_bwd(reinterpret_cast<const float4*>(raw.rotation)[i], make_float4(drot[0], drot[1], drot[2], drot[3]));
        raw.d_opacity[i] = act_sigmoid_bwd(dop, raw.opacity[i]);
        // dL/dcolours (csum: summed in the view loop), then d cat / d clamp_min / d eval_sh (degree 0)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = act_rgb_raw_deg0(raw.f_dc[3 * (size_t)i + c]) >= 0.0f ? csum[c] : 0.0f;
            raw.d_f_dc[3 * (size_t)i + c] = ACT_SH_C0 * g;
        }
        if (raw.E) raw.d_extra[i] = csum[3];
        return;
    }
    dL_dopacities[i] = dop;
    if (dL_dscales) {
#pragma unroll
        for (int k = 0; k < 3; ++k) dL_dscales[3 * i + k] = dscale[k];
    }
    if (dL_drotations) reinterpret_cast<float4*>(dL_drotations)[i] = make_float4(drot[0], drot[1], drot[2], drot[3]);
    if (dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = dcov[k];
    }
    }
}

// dL/dcolors [P, C] out of the 64-byte aligned accumulator rows: one thread per element,
// contiguous reads inside a row, fully coalesced writes.
// deterministic debug mode: the fixed-point accumulator rows -> the float rows the kernels below read
// (dst holds, per element, the bit pattern of the largest |partial| that went into src — composite_bwd.hip acc_add:
//  the fixed point of the element is 2^-(170 - its biased exponent); the same expression is evaluated here)
__global__ void __launch_bounds__(256)
fixed_to_float_kernel(int64_t n, const long long* __restrict__ src, float* __restrict__ dst)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const unsigned mb = reinterpret_cast<const unsigned*>(dst)[e];
    const int eb = (int)((mb >> 23) & 0xffu);
    if (eb == 255) {   // a non-finite partial (composite_bwd.hip acc_add): NaN, or the infinity whose sign(s) were counted
        const unsigned long long cnt = (unsigned long long)src[e];
        const bool pos = (cnt & 0xffffffffull) != 0, neg = (cnt >> 32) != 0;
        dst[e] = ((mb & 0x7fffffu) || pos == neg) ? __builtin_nanf("") : (neg ? -__builtin_inff() : __builtin_inff());
        return;
    }
    dst[e] = (float)ldexp((double)src[e], -(170 - (eb > 0 ? eb : 1)));
}

int launch_fixed_to_float(int64_t n, const long long* src, float* dst, hipStream_t stream)
{
    if (n <= 0) return SPLATRASTER_OK;
    hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, src, dst);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

__global__ void __launch_bounds__(256)
gather_dcolors_kernel(int64_t n, int C, int GROW, int V, int64_t P, const float* __restrict__ gacc,
                      float* __restrict__ dL_dcolors)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int64_t i = e / C;
    const int ch = (int)(e - i * C);
    float sum = gacc[i * GROW + ch];
    for (int v = 1; v < V; ++v) sum += gacc[(v * P + i) * GROW + ch];   // the feature table is shared by the views
    dL_dcolors[e] = sum;
}

int launch_preprocess_bwd(const splatraster_settings& s, int32_t P, int32_t V, const WinCams& cams, const WinGrad& grads,
                          const float* means3D, const float* shs, const float* scales, const float* rotations,
                          const float* cov3D_precomp, const uint8_t* clamped, const float4* rec, const float* gacc, int C,
                          float* dL_dcolors, float* dL_dmeans3D, float* dL_dopacities, float* dL_dscales,
                          float* dL_drotations, float* dL_dcov3D, float* dL_dshs, float* dL_dview,
                          float* dL_dproj, float* dL_dcampos, float* pose_acc, hipStream_t stream, const RawBwd* raw)
{
    if (P == 0) return SPLATRASTER_OK;
    if (dL_dcolors && !raw) {
        const int64_t n = (int64_t)P * C;
        hipLaunchKernelGGL(gather_dcolors_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, C,
                           gacc_row_floats(C), V, (int64_t)P, gacc, dL_dcolors);
        SR_LAUNCH_CHECK();
    }
    const bool pose = dL_dview && dL_dproj;      // (the accumulator sets behind gacc were zeroed with the rows: capi.hip)
    if (pose && !pose_acc) return SPLATRASTER_ERR_BAD_ARG;
#define SR_PBWD_ARGS                                                                                              \
    P, V, s.image_width, s.image_height, s.scale_modifier, s.sh_degree, s.sh_coeffs, cams, grads, means3D,         \
        shs, scales, rotations, cov3D_precomp, clamped, rec, gacc, C, gacc_row_floats(C),                          \
        gacc_moment_offset(C), dL_dmeans3D, dL_dopacities, dL_dscales, dL_drotations,                              \
        dL_dcov3D, dL_dshs, dL_dview, dL_dproj, dL_dcampos, pose_acc, (raw ? *raw : RawBwd{})
    if (raw) {
        if (pose || shs || cov3D_precomp || !scales || !rotations || raw->E > 1) return SPLATRASTER_ERR_UNSUPPORTED;   // (C <= 4: the colour columns share the moments' line)
        hipLaunchKernelGGL((preprocess_bwd_kernel<false, true>), dim3((P + 255) / 256), dim3(256), 0, stream, SR_PBWD_ARGS);
    } else if (pose)
        hipLaunchKernelGGL(preprocess_bwd_kernel<true>, dim3((P + 255) / 256), dim3(256), 0, stream, SR_PBWD_ARGS);
    else
        hipLaunchKernelGGL(preprocess_bwd_kernel<false>, dim3((P + 255) / 256), dim3(256), 0, stream, SR_PBWD_ARGS);
#undef SR_PBWD_ARGS
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

}  // namespace sr
