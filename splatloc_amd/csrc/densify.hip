// densify.hip — SplatLoc's densify / clone / split / prune with optimizer-state surgery as ONE
// compaction on the device, the Adam step over all parameter groups as ONE launch, and the isotropic
// scale regulariser.  Replaces
//   GaussianModel.densify_and_prune            gaussian_model.py:655-675
//     densify_and_clone / densify_and_split    gaussian_model.py:632-653 / 590-630
//     prune_points, _prune_optimizer           gaussian_model.py:492-526
//     cat_tensors_to_optimizer, densification_postfix   gaussian_model.py:528-587
//   torch.optim.Adam(l, lr=0.0, eps=1e-15).step()       gaussian_model.py:254-300, train_gaussians.py:265
//   the isotropic regulariser + key-primitive gradient gate    train_gaussians.py:222-234
// i.e. three boolean-mask `cat`s and two boolean-mask prunes over 8 parameter tensors and 2x7 Adam
// moment tensors (~90 small launches, a device->host sync per mask) by: flags -> one scan -> one gather.
//
// Row order of the result = the reference's: surviving originals (in order), surviving clones (in
// selection order), surviving "copy 0" split children, surviving "copy 1" split children.  The four
// classes are four sections of ONE flag array of 4 P entries, so a single exclusive scan yields every
// destination.  HBM-bound integer/byte work; nothing here is reshaped into a GEMM.
#include "common.h"

namespace sr {

constexpr int DN_BLOCK = 256;

struct DensifyHyper {
    float max_grad, min_opacity, size_split /* percent_dense * extent */, size_prune /* 0.1 * extent */;
    int use_size_prune, primitive_reg;
};

__device__ __forceinline__ float row_max_exp(const float* __restrict__ scaling, int SC, size_t i, float* s)
{
    float m = -3.0e38f;
#pragma unroll 3
    for (int k = 0; k < SC; ++k) {
        s[k] = expf(scaling[(size_t)SC * i + k]);
        m = fmaxf(m, s[k]);
    }
    return m;
}

// flags[sec * P + i], sec = 0 original kept, 1 clone kept, 2 / 3 split child kept
__global__ void __launch_bounds__(DN_BLOCK)
densify_flags_kernel(int P, int SC, const float* __restrict__ scaling, const float* __restrict__ opacity,
                     const float* __restrict__ marker, const float* __restrict__ accum,
                     const float* __restrict__ denom, DensifyHyper h, uint32_t* __restrict__ flags)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float g = accum[i] / denom[i];                  // grads = xyz_gradient_accum / denom
    if (g != g) g = 0.0f;                           // grads[grads.isnan()] = 0
    g = fabsf(g);                                   // torch.norm over the 1-wide last dim
    float s[3];
    const float smax = row_max_exp(scaling, SC, (size_t)i, s);
    const bool hot = g >= h.max_grad;
    const bool clone = hot && smax <= h.size_split;
    const bool split = hot && smax > h.size_split;
    const float op = 1.0f / (1.0f + expf(-opacity[i]));
    const bool gate = !h.primitive_reg || marker[i] <= 0.005f;   // key primitives are never pruned
    const bool low = op < h.min_opacity;
    const bool prune_self = (low || (h.use_size_prune && smax > h.size_prune)) && gate;
    // a child's stored scale is log(s / 1.6); the prune looks at exp of that.  torch on the GPU evaluates
    // `t / 1.6` (gaussian_model.py:603, a Python scalar divisor) as t * (1 / 1.6f), not as a division: mirrored,
    // so the prune / size thresholds see the same last bit as the reference on its own device
    float cmax = -3.0e38f;
#pragma unroll 3
    for (int k = 0; k < SC; ++k) cmax = fmaxf(cmax, expf(logf(s[k] * (1.0f / 1.6f))));
    const bool prune_child = (low || (h.use_size_prune && cmax > h.size_prune)) && gate;
    flags[i] = (!split && !prune_self) ? 1u : 0u;
    flags[(size_t)P + i] = (clone && !prune_self) ? 1u : 0u;
    const uint32_t c = (split && !prune_child) ? 1u : 0u;
    flags[2 * (size_t)P + i] = c;
    flags[3 * (size_t)P + i] = c;
}

// ---- counter-based normal draws (Philox4x32-10 + Box-Muller): a replica that runs the same step with
// the same seed draws the same children without any broadcast --------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi, uint32_t (&out)[4])
{
    uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi, (uint32_t)(ctr_hi >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = c[k];
}
__device__ __forceinline__ void normal3(uint64_t seed, uint64_t stream_id, uint32_t row, uint32_t copy, float (&z)[3])
{
    uint32_t r[4];
    philox4x32(seed, ((uint64_t)copy << 32) | row, stream_id, r);
    const float u0 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
    const float u2 = ((float)(r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = (float)(r[3] >> 8) * (1.0f / 16777216.0f);
    const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
    z[0] = ra * cosf(6.28318530717958647692f * u1);
    z[1] = ra * sinf(6.28318530717958647692f * u1);
    z[2] = rb * cosf(6.28318530717958647692f * u3);
}

struct DensifyGroups {           // the 8 groups of GaussianModel.training_setup, raw (pre-activation) tensors
    const float* in[8];          // xyz, f_dc, f_rest, opacity, marker, kp_score, scaling, rotation
    const float* m_in[8];        // exp_avg (NULL: the group has no Adam state yet)
    const float* v_in[8];
    float* out[8];
    float* m_out[8];
    float* v_out[8];
    int width[8];                // floats per row
};
enum { G_XYZ = 0, G_FDC, G_FREST, G_OPACITY, G_MARKER, G_KP, G_SCALING, G_ROTATION };

__global__ void __launch_bounds__(DN_BLOCK)
densify_gather_kernel(int P, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ incl, DensifyGroups G,
                      const float* __restrict__ unit_noise /*[2,P,3] or NULL*/, uint64_t seed, uint64_t stream_id,
                      int32_t* __restrict__ source_row, int32_t* __restrict__ source_kind)
{
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= 4 * (size_t)P) return;
    if (!flags[s]) return;
    const size_t d = incl[s] - 1u;                // exclusive prefix = destination row
    const int sec = (int)(s / (size_t)P);
    const size_t i = s - (size_t)sec * P;
    if (source_row) source_row[d] = (int32_t)i;
    if (source_kind) source_kind[d] = sec;
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) {
        const int w = G.width[gq];
        if (w == 0 || !G.in[gq]) continue;
        const float* src = G.in[gq] + (size_t)w * i;
        float* dst = G.out[gq] + (size_t)w * d;
        if (sec >= 2 && gq == G_XYZ) {
            // new_xyz = R(q / |q|) (z * exp(scaling)) + xyz     gaussian_model.py:598-601
            const int SC = G.width[G_SCALING];
            float z[3];
            if (unit_noise) {
                const float* u = unit_noise + ((size_t)(sec - 2) * P + i) * 3;
                z[0] = u[0]; z[1] = u[1]; z[2] = u[2];
            } else {
                normal3(seed, stream_id, (uint32_t)i, (uint32_t)(sec - 2), z);
            }
            float smp[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) smp[k] = z[k] * expf(G.in[G_SCALING][(size_t)SC * i + (SC == 1 ? 0 : k)]);
            const float* q = G.in[G_ROTATION] + 4 * i;
            const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            const float r = q[0] / n, x = q[1] / n, y = q[2] / n, zq = q[3] / n;
            const float R00 = 1.f - 2.f * (y * y + zq * zq), R01 = 2.f * (x * y - r * zq), R02 = 2.f * (x * zq + r * y);
            const float R10 = 2.f * (x * y + r * zq), R11 = 1.f - 2.f * (x * x + zq * zq), R12 = 2.f * (y * zq - r * x);
            const float R20 = 2.f * (x * zq - r * y), R21 = 2.f * (y * zq + r * x), R22 = 1.f - 2.f * (x * x + y * y);
            dst[0] = (R00 * smp[0] + R01 * smp[1] + R02 * smp[2]) + src[0];
            dst[1] = (R10 * smp[0] + R11 * smp[1] + R12 * smp[2]) + src[1];
            dst[2] = (R20 * smp[0] + R21 * smp[1] + R22 * smp[2]) + src[2];
        } else if (sec >= 2 && gq == G_SCALING) {
            for (int k = 0; k < w; ++k) dst[k] = logf(expf(src[k]) * (1.0f / 1.6f));   // / (0.8 * N), N = 2, as torch divides by a scalar
        } else {
            for (int k = 0; k < w; ++k) dst[k] = src[k];
        }
        if (G.m_in[gq]) {   // Adam moments: originals keep theirs, every new row starts from zero
            const float* ms = G.m_in[gq] + (size_t)w * i;
            const float* vs = G.v_in[gq] + (size_t)w * i;
            float* md = G.m_out[gq] + (size_t)w * d;
            float* vd = G.v_out[gq] + (size_t)w * d;
            for (int k = 0; k < w; ++k) {
                md[k] = sec == 0 ? ms[k] : 0.0f;
                vd[k] = sec == 0 ? vs[k] : 0.0f;
            }
        }
    }
}

size_t densify_workspace_bytes(int32_t P)
{
    const size_t n = 4 * (size_t)(P > 0 ? P : 1);
    return align_up(4 * n, 256) * 2 + align_up(scan_tmp_bytes((int64_t)n), 256) + 256;
}

struct DensifyWs { uint32_t* flags; uint32_t* incl; void* scan_tmp; uint32_t* total; };
static DensifyWs densify_ws(void* base, int32_t P)
{
    const size_t n = 4 * (size_t)(P > 0 ? P : 1);
    char* b = reinterpret_cast<char*>(base);
    DensifyWs w;
    w.flags = reinterpret_cast<uint32_t*>(b);
    w.incl = reinterpret_cast<uint32_t*>(b + align_up(4 * n, 256));
    w.scan_tmp = b + 2 * align_up(4 * n, 256);
    w.total = reinterpret_cast<uint32_t*>(b + 2 * align_up(4 * n, 256) + align_up(scan_tmp_bytes((int64_t)n), 256));
    return w;
}

int densify_plan(int32_t P, int SC, const float* scaling, const float* opacity, const float* marker, const float* accum,
                 const float* denom, float max_grad, float min_opacity, float extent, float percent_dense,
                 int use_size_prune, int primitive_reg, void* workspace, int32_t* new_P, hipStream_t stream)
{
    DensifyWs w = densify_ws(workspace, P);
    DensifyHyper h{max_grad, min_opacity, percent_dense * extent, 0.1f * extent, use_size_prune, primitive_reg};
    hipLaunchKernelGGL(densify_flags_kernel, dim3((P + DN_BLOCK - 1) / DN_BLOCK), dim3(DN_BLOCK), 0, stream, P, SC, scaling,
                       opacity, marker, accum, denom, h, w.flags);
    SR_LAUNCH_CHECK();
    int st = inclusive_scan_u32(4 * (int64_t)P, w.flags, nullptr, w.incl, w.total, w.scan_tmp, stream);
    if (st) return st;
    uint32_t total[2] = {0, 0};
    SR_HIP_CHECK(hipMemcpyAsync(total, w.total, sizeof(total), hipMemcpyDeviceToHost, stream));
    SR_HIP_CHECK(hipStreamSynchronize(stream));   // the new row count sizes the caller's allocations
    *new_P = (int32_t)total[0];
    return SPLATRASTER_OK;
}

int densify_apply(int32_t P, const DensifyGroups& G, const float* unit_noise, uint64_t seed, uint64_t stream_id,
                  void* workspace, int32_t* source_row, int32_t* source_kind, hipStream_t stream)
{
    DensifyWs w = densify_ws(workspace, P);
    const size_t n = 4 * (size_t)P;
    hipLaunchKernelGGL(densify_gather_kernel, dim3((unsigned)((n + DN_BLOCK - 1) / DN_BLOCK)), dim3(DN_BLOCK), 0, stream, P,
                       w.flags, w.incl, G, unit_noise, seed, stream_id, source_row, source_kind);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

// ---- key-frame insertion: rows appended to every group, zero Adam moments for the new rows ---------------------
// GaussianModel.extend_from_pcd -> densification_postfix -> cat_tensors_to_optimizer (gaussian_model.py:222-241,
// :528-587): torch.cat per parameter tensor and per Adam moment (24 cat launches + 16 zeros_like) as ONE launch.
// G.in = the model (P rows), G.m_in / G.v_in its moments, `extra` = the N new rows; outputs hold P + N rows.
struct AppendRows { const float* extra[8]; };
__global__ void __launch_bounds__(DN_BLOCK)
model_append_kernel(int P, int N, DensifyGroups G, AppendRows X)
{
    const size_t d = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= (size_t)P + (size_t)N) return;
    const bool old = d < (size_t)P;
    const size_t r = old ? d : d - (size_t)P;
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) {
        const int w = G.width[gq];
        if (w == 0 || !G.out[gq]) continue;
        const float* src = (old ? G.in[gq] : X.extra[gq]) + (size_t)w * r;
        float* dst = G.out[gq] + (size_t)w * d;
        for (int k = 0; k < w; ++k) dst[k] = src[k];
        if (G.m_out[gq]) {
            float* md = G.m_out[gq] + (size_t)w * d;
            float* vd = G.v_out[gq] + (size_t)w * d;
            for (int k = 0; k < w; ++k) {
                md[k] = old ? G.m_in[gq][(size_t)w * r + k] : 0.0f;
                vd[k] = old ? G.v_in[gq][(size_t)w * r + k] : 0.0f;
            }
        }
    }
}

int model_append(int32_t P, int32_t N, const DensifyGroups& G, const AppendRows& X, hipStream_t stream)
{
    const size_t n = (size_t)P + (size_t)N;
    if (n == 0) return SPLATRASTER_OK;
    hipLaunchKernelGGL(model_append_kernel, dim3((unsigned)((n + DN_BLOCK - 1) / DN_BLOCK)), dim3(DN_BLOCK), 0, stream, P, N, G, X);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

// ---- Adam over all groups, one launch --------------------------------------------------------------
constexpr int ADAM_MAX_GROUPS = 16;
struct AdamTable {
    float* param[ADAM_MAX_GROUPS];
    const float* grad[ADAM_MAX_GROUPS];
    float* m[ADAM_MAX_GROUPS];
    float* v[ADAM_MAX_GROUPS];
    const float* gate[ADAM_MAX_GROUPS];        // per-ROW gate (or NULL): rows with gate > gate_thr get gradient 0
    unsigned long long end[ADAM_MAX_GROUPS];   // exclusive end of the group's element range in the flat index space
    float step_size[ADAM_MAX_GROUPS];          // lr / (1 - beta1^t)
    float inv_bc2_sqrt[ADAM_MAX_GROUPS];       // 1 / sqrt(1 - beta2^t)
    int row_width[ADAM_MAX_GROUPS];
    int n;
    float beta1, beta2, eps, gate_thr;
    float omb1, omb2;                          // 1 - beta formed in double on the host, like torch's python floats
    // optional tail of the launch: max_radii[i] = max(max_radii[i], radii[i]) for radii[i] > 0 — the statistics line of
    // SplatLoc.color_refinement (train_gaussians.py:293-294), a launch of its own otherwise
    int radii_n;
    const int* radii;
    float* max_radii;
};

__device__ __forceinline__ void radii_update(const AdamTable& T, unsigned long long i)
{
    const int r = T.radii[i];
    if (r > 0) T.max_radii[i] = fmaxf(T.max_radii[i], (float)r);
}

__device__ __forceinline__ void adam_element(const AdamTable& T, int gq, float g, float& m, float& v, float& p)
{
    // torch.optim.Adam: exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m = m + (g - m) * T.omb1;
    v = v * T.beta2 + (T.omb2 * g) * g;
    const float denom = sqrtf(v) * T.inv_bc2_sqrt[gq] + T.eps;
    p -= T.step_size[gq] * (m / denom);
}

// scalar form: any alignment
__global__ void __launch_bounds__(256)
adam_kernel(AdamTable T, unsigned long long total)
{
    for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < total + (unsigned long long)T.radii_n;
         e += (unsigned long long)gridDim.x * blockDim.x) {
        if (e >= total) { radii_update(T, e - total); continue; }
        int gq = 0;
#pragma unroll 1
        while (gq + 1 < T.n && e >= T.end[gq]) ++gq;
        const unsigned long long k = e - (gq ? T.end[gq - 1] : 0ull);
        float g = T.grad[gq][k];
        if (T.gate[gq] && T.gate[gq][k / (unsigned)T.row_width[gq]] > T.gate_thr) g = 0.0f;   // train_gaussians.py:231-234
        float m = T.m[gq][k], v = T.v[gq][k], p = T.param[gq][k];
        adam_element(T, gq, g, m, v, p);
        T.m[gq][k] = m;
        T.v[gq][k] = v;
        T.param[gq][k] = p;
    }
}

// 16-byte form (every tensor of every group 16-byte aligned — torch allocations are): the flat index space counts QUADS of four
// consecutive elements of a group (the last quad of a group may be partial); per element the arithmetic is adam_element's, so the
// two kernels agree bit for bit.  Seven streams of 4-byte accesses per thread left the step at 3.4 TB/s of its 28 bytes per element.
struct AdamQuads {
    unsigned long long qend[ADAM_MAX_GROUPS];   // exclusive end of the group's quad range
    long long numel[ADAM_MAX_GROUPS];
};
__global__ void __launch_bounds__(256)
adam_quad_kernel(AdamTable T, AdamQuads Q, unsigned long long total_quads)
{
    const unsigned long long radii_quads = ((unsigned long long)T.radii_n + 3ull) / 4ull;
    for (unsigned long long q = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; q < total_quads + radii_quads;
         q += (unsigned long long)gridDim.x * blockDim.x) {
        if (q >= total_quads) {
            const unsigned long long i0 = (q - total_quads) * 4ull;
            for (unsigned long long i = i0; i < i0 + 4ull && i < (unsigned long long)T.radii_n; ++i) radii_update(T, i);
            continue;
        }
        int gq = 0;
#pragma unroll 1
        while (gq + 1 < T.n && q >= Q.qend[gq]) ++gq;
        const unsigned long long k0 = (q - (gq ? Q.qend[gq - 1] : 0ull)) * 4ull;
        const long long left = Q.numel[gq] - (long long)k0;
        const float* gate = T.gate[gq];
        unsigned long long row = 0;
        int rem = 0;
        const int rw = T.row_width[gq];
        if (gate) {
            row = k0 / (unsigned)rw;
            rem = (int)(k0 - row * (unsigned)rw);
        }
        if (left >= 4) {
            float4 g4 = *reinterpret_cast<const float4*>(T.grad[gq] + k0);
            float4 m4 = *reinterpret_cast<const float4*>(T.m[gq] + k0);
            float4 v4 = *reinterpret_cast<const float4*>(T.v[gq] + k0);
            float4 p4 = *reinterpret_cast<const float4*>(T.param[gq] + k0);
            float g[4] = {g4.x, g4.y, g4.z, g4.w}, m[4] = {m4.x, m4.y, m4.z, m4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w},
                  p[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (gate) {
                    if (gate[row] > T.gate_thr) g[j] = 0.0f;   // train_gaussians.py:231-234
                    if (++rem == rw) { rem = 0; ++row; }
                }
                adam_element(T, gq, g[j], m[j], v[j], p[j]);
            }
            *reinterpret_cast<float4*>(T.m[gq] + k0) = make_float4(m[0], m[1], m[2], m[3]);
            *reinterpret_cast<float4*>(T.v[gq] + k0) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(T.param[gq] + k0) = make_float4(p[0], p[1], p[2], p[3]);
        } else {
            for (long long j = 0; j < left; ++j) {
                const unsigned long long k = k0 + (unsigned long long)j;
                float g = T.grad[gq][k];
                if (gate) {
                    if (gate[row] > T.gate_thr) g = 0.0f;
                    if (++rem == rw) { rem = 0; ++row; }
                }
                float m = T.m[gq][k], v = T.v[gq][k], p = T.param[gq][k];
                adam_element(T, gq, g, m, v, p);
                T.m[gq][k] = m;
                T.v[gq][k] = v;
                T.param[gq][k] = p;
            }
        }
    }
}

int adam_step(int n, const splatraster_adam_group* groups, double beta1, double beta2, double eps, float gate_thr,
              int32_t radii_n, const int32_t* radii, float* max_radii, hipStream_t stream)
{
    AdamTable T{};
    AdamQuads Q{};
    unsigned long long total = 0, total_quads = 0;
    int used = 0;
    bool aligned = true;
    for (int k = 0; k < n; ++k) {
        const splatraster_adam_group& g = groups[k];
        if (g.numel <= 0 || !g.grad) continue;     // a group without a gradient is skipped (torch semantics)
        if (!g.param || !g.exp_avg || !g.exp_avg_sq || g.step < 1.0 || g.row_width < 1) return SPLATRASTER_ERR_BAD_ARG;
        total += (unsigned long long)g.numel;
        total_quads += ((unsigned long long)g.numel + 3ull) / 4ull;
        T.param[used] = g.param; T.grad[used] = g.grad; T.m[used] = g.exp_avg; T.v[used] = g.exp_avg_sq;
        T.gate[used] = g.row_gate; T.end[used] = total; T.row_width[used] = g.row_width;
        Q.qend[used] = total_quads; Q.numel[used] = g.numel;
        aligned = aligned && (((uintptr_t)g.param | (uintptr_t)g.grad | (uintptr_t)g.exp_avg | (uintptr_t)g.exp_avg_sq) & 15u) == 0;
        const double bc1 = 1.0 - pow(beta1, g.step), bc2 = 1.0 - pow(beta2, g.step);
        T.step_size[used] = (float)((double)g.lr / bc1);
        T.inv_bc2_sqrt[used] = (float)(1.0 / sqrt(bc2));
        ++used;
    }
    const bool with_radii = radii_n > 0 && radii && max_radii;
    if (used == 0 && !with_radii) return SPLATRASTER_OK;
    T.radii_n = with_radii ? radii_n : 0;
    T.radii = radii;
    T.max_radii = max_radii;
    T.n = used; T.beta1 = (float)beta1; T.beta2 = (float)beta2; T.eps = (float)eps; T.gate_thr = gate_thr;
    T.omb1 = (float)(1.0 - beta1);   // torch: python-float `1 - beta` rounded to fp32 once
    T.omb2 = (float)(1.0 - beta2);
    if (aligned) {
        unsigned long long blocks = (total_quads + ((unsigned long long)T.radii_n + 3ull) / 4ull + 255) / 256;
        if (blocks > 4096) blocks = 4096;   // grid-stride: 16 workgroups per CU
        hipLaunchKernelGGL(adam_quad_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, T, Q, total_quads);
    } else {
        unsigned long long blocks = (total + (unsigned long long)T.radii_n + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, T, total);
    }
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

// ---- isotropic scale regulariser (train_gaussians.py:222-228) ----------------------------------------
//   mask = marker > 0.005 ;  x_i = mean_k(scaling[i, k]) / (0.02 (1 - marker_i)) ;  loss = mean_{mask} |x_i - 1|
// forward: per-row d|x - 1|/d scaling[i, k] (0 outside the mask) + deterministic block partials of
// (sum |x - 1|, count); finish: out[0] = loss, out[1] = 1 / count (0 when the mask is empty: the
// reference's mean over an empty tensor is NaN and poisons the step — here it contributes nothing).
__global__ void __launch_bounds__(256)
isotropic_fwd_kernel(int P, int SC, const float* __restrict__ scaling /*activated [P,SC]*/, const float* __restrict__ marker,
                     float* __restrict__ row_grad /*[P]*/, double* __restrict__ partial /*[blocks,2]*/)
{
    double acc = 0.0, cnt = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
        const float mk = marker[i];
        float rg = 0.0f;
        if (mk > 0.005f) {
            float s = 0.0f;
            for (int k = 0; k < SC; ++k) s += scaling[(size_t)SC * i + k];
            const float inv = 1.0f / (0.02f * (1.0f - mk));
            const float x = (s / (float)SC) * inv - 1.0f;
            acc += (double)fabsf(x);
            cnt += 1.0;
            rg = (float)((x > 0.f) - (x < 0.f)) * inv / (float)SC;
        }
        row_grad[i] = rg;
    }
    __shared__ double s_red[256 / WAVE][2];
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        acc += __shfl_xor(acc, d, WAVE);
        cnt += __shfl_xor(cnt, d, WAVE);
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) { s_red[threadIdx.x / WAVE][0] = acc; s_red[threadIdx.x / WAVE][1] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, c = 0;
        for (int w = 0; w < 256 / WAVE; ++w) { a += s_red[w][0]; c += s_red[w][1]; }
        partial[2 * (size_t)blockIdx.x] = a;
        partial[2 * (size_t)blockIdx.x + 1] = c;
    }
}
__global__ void isotropic_finish_kernel(int blocks, const double* __restrict__ partial, float* __restrict__ out)
{
    // one wave: lane l sums partials l, l + 64, ... in order, then a fixed butterfly — deterministic, and 16 dependent
    // loads per lane instead of 1024 in one thread (81 -> 4 us at 500k Gaussians)
    if (blockIdx.x != 0) return;
    double a = 0, c = 0;
    for (int b = threadIdx.x; b < blocks; b += WAVE) { a += partial[2 * (size_t)b]; c += partial[2 * (size_t)b + 1]; }
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        a += __shfl_xor(a, d, WAVE);
        c += __shfl_xor(c, d, WAVE);
    }
    if (threadIdx.x != 0) return;
    out[0] = c > 0 ? (float)(a / c) : 0.0f;
    out[1] = c > 0 ? (float)(1.0 / c) : 0.0f;
}

static int isotropic_blocks(int32_t P)
{
    int b = (P + 255) / 256;
    return b < 1 ? 1 : (b > 1024 ? 1024 : b);
}
size_t isotropic_workspace_bytes(int32_t P) { return (size_t)isotropic_blocks(P) * 2 * sizeof(double); }

int launch_isotropic(int32_t P, int SC, const float* scaling, const float* marker, float* row_grad, float* out,
                     void* workspace, hipStream_t stream)
{
    const int blocks = isotropic_blocks(P);
    double* partial = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(isotropic_fwd_kernel, dim3(blocks), dim3(256), 0, stream, P, SC, scaling, marker, row_grad, partial);
    SR_LAUNCH_CHECK();
    hipLaunchKernelGGL(isotropic_finish_kernel, dim3(1), dim3(64), 0, stream, blocks, partial, out);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

}  // namespace sr

using namespace sr;

extern "C" {

size_t splatraster_densify_workspace_bytes(int32_t P) { return densify_workspace_bytes(P); }

static int check_model(const splatraster_model* m, int need_all)
{
    if (!m || m->P < 0 || m->f_rest_width < 0 || m->kp_width < 0 || m->marker_width < 0) return SPLATRASTER_ERR_BAD_ARG;
    if (m->scaling_width != 1 && m->scaling_width != 3) return SPLATRASTER_ERR_BAD_ARG;
    if (m->P == 0 || !need_all) return SPLATRASTER_OK;
    if (!m->xyz || !m->f_dc || !m->opacity || !m->scaling || !m->rotation) return SPLATRASTER_ERR_BAD_ARG;
    if ((m->f_rest_width > 0 && !m->f_rest) || (m->kp_width > 0 && !m->kp_score) || (m->marker_width > 0 && !m->marker))
        return SPLATRASTER_ERR_BAD_ARG;
    return SPLATRASTER_OK;
}

int splatraster_densify_plan(const splatraster_model* model, const float* xyz_gradient_accum, const float* denom,
                             float max_grad, float min_opacity, float extent, float percent_dense,
                             int32_t use_size_prune, int32_t primitive_reg, void* workspace, int32_t* new_P, void* stream)
{
    int st = check_model(model, 1);
    if (st) return st;
    if (!new_P || !(max_grad > 0.0f)) return SPLATRASTER_ERR_BAD_ARG;   // clones are kept out of the split by their zero gradient
    *new_P = 0;
    if (model->P == 0) return SPLATRASTER_OK;
    if (!xyz_gradient_accum || !denom || !workspace) return SPLATRASTER_ERR_BAD_ARG;
    if (primitive_reg && (model->marker_width != 1 || !model->marker)) return SPLATRASTER_ERR_BAD_ARG;
    if (4 * (int64_t)model->P >= ((int64_t)1 << 31)) return SPLATRASTER_ERR_OVERFLOW;
    st = lookback_error_init();
    if (st) return st;
    st = densify_plan(model->P, model->scaling_width, model->scaling, model->opacity, model->marker, xyz_gradient_accum,
                      denom, max_grad, min_opacity, extent, percent_dense, use_size_prune, primitive_reg, workspace, new_P,
                      reinterpret_cast<hipStream_t>(stream));
    return st;
}

int splatraster_densify_apply(const splatraster_model* model, const splatraster_model* exp_avg,
                              const splatraster_model* exp_avg_sq, const float* unit_noise, uint64_t seed,
                              uint64_t draw_id, void* workspace, int32_t new_P, splatraster_model* out_model,
                              splatraster_model* out_exp_avg, splatraster_model* out_exp_avg_sq, int32_t* source_row,
                              int32_t* source_kind, void* stream)
{
    int st = check_model(model, 1);
    if (st) return st;
    if (new_P < 0 || !out_model) return SPLATRASTER_ERR_BAD_ARG;
    if (model->P == 0 || new_P == 0) return SPLATRASTER_OK;
    if (!workspace || (exp_avg == nullptr) != (exp_avg_sq == nullptr)) return SPLATRASTER_ERR_BAD_ARG;
    if (exp_avg && (!out_exp_avg || !out_exp_avg_sq)) return SPLATRASTER_ERR_BAD_ARG;
    const int widths[8] = {3, 3, model->f_rest_width, 1, model->marker_width, model->kp_width, model->scaling_width, 4};
    auto ptrs = [](const splatraster_model* m, const float* (&p)[8]) {
        p[0] = m->xyz; p[1] = m->f_dc; p[2] = m->f_rest; p[3] = m->opacity; p[4] = m->marker; p[5] = m->kp_score;
        p[6] = m->scaling; p[7] = m->rotation;
    };
    DensifyGroups G{};
    const float *in[8], *out[8], *mi[8] = {}, *vi[8] = {}, *mo[8] = {}, *vo[8] = {};
    ptrs(model, in);
    ptrs(out_model, out);
    if (exp_avg) { ptrs(exp_avg, mi); ptrs(exp_avg_sq, vi); ptrs(out_exp_avg, mo); ptrs(out_exp_avg_sq, vo); }
    for (int k = 0; k < 8; ++k) {
        G.width[k] = widths[k];
        G.in[k] = in[k];
        G.out[k] = const_cast<float*>(out[k]);
        if (widths[k] > 0 && in[k] && !out[k]) return SPLATRASTER_ERR_BAD_ARG;
        // a group carries Adam state only when BOTH moments are given (the marker never has any in map())
        const bool has = mi[k] && vi[k];
        if (has && (!mo[k] || !vo[k])) return SPLATRASTER_ERR_BAD_ARG;
        G.m_in[k] = has ? mi[k] : nullptr;
        G.v_in[k] = has ? vi[k] : nullptr;
        G.m_out[k] = has ? const_cast<float*>(mo[k]) : nullptr;
        G.v_out[k] = has ? const_cast<float*>(vo[k]) : nullptr;
    }
    return densify_apply(model->P, G, unit_noise, seed, draw_id, workspace, source_row, source_kind,
                         reinterpret_cast<hipStream_t>(stream));
}

int splatraster_model_append(const splatraster_model* model, const splatraster_model* exp_avg,
                             const splatraster_model* exp_avg_sq, const splatraster_model* extra,
                             splatraster_model* out_model, splatraster_model* out_exp_avg,
                             splatraster_model* out_exp_avg_sq, void* stream)
{
    int st = check_model(model, 1);
    if (st) return st;
    st = check_model(extra, 1);
    if (st) return st;
    if (!out_model || (exp_avg == nullptr) != (exp_avg_sq == nullptr)) return SPLATRASTER_ERR_BAD_ARG;
    if (exp_avg && (!out_exp_avg || !out_exp_avg_sq)) return SPLATRASTER_ERR_BAD_ARG;
    if (model->f_rest_width != extra->f_rest_width || model->marker_width != extra->marker_width ||
        model->kp_width != extra->kp_width || model->scaling_width != extra->scaling_width)
        return SPLATRASTER_ERR_BAD_ARG;
    if ((int64_t)model->P + (int64_t)extra->P >= ((int64_t)1 << 31)) return SPLATRASTER_ERR_OVERFLOW;
    const int widths[8] = {3, 3, model->f_rest_width, 1, model->marker_width, model->kp_width, model->scaling_width, 4};
    auto ptrs = [](const splatraster_model* m, const float* (&p)[8]) {
        p[0] = m->xyz; p[1] = m->f_dc; p[2] = m->f_rest; p[3] = m->opacity; p[4] = m->marker; p[5] = m->kp_score;
        p[6] = m->scaling; p[7] = m->rotation;
    };
    DensifyGroups G{};
    AppendRows X{};
    const float *in[8], *ex[8], *out[8], *mi[8] = {}, *vi[8] = {}, *mo[8] = {}, *vo[8] = {};
    ptrs(model, in);
    ptrs(extra, ex);
    ptrs(out_model, out);
    if (exp_avg) { ptrs(exp_avg, mi); ptrs(exp_avg_sq, vi); ptrs(out_exp_avg, mo); ptrs(out_exp_avg_sq, vo); }
    for (int k = 0; k < 8; ++k) {
        G.width[k] = widths[k];
        G.in[k] = in[k];
        X.extra[k] = ex[k];
        G.out[k] = const_cast<float*>(out[k]);
        if (widths[k] > 0 && !out[k]) return SPLATRASTER_ERR_BAD_ARG;
        // a group carries Adam state only when BOTH moments are given (the marker never has any in map())
        const bool has = mi[k] && vi[k];
        if (has && (!mo[k] || !vo[k])) return SPLATRASTER_ERR_BAD_ARG;
        G.m_in[k] = has ? mi[k] : nullptr;
        G.v_in[k] = has ? vi[k] : nullptr;
        G.m_out[k] = has ? const_cast<float*>(mo[k]) : nullptr;
        G.v_out[k] = has ? const_cast<float*>(vo[k]) : nullptr;
    }
    return model_append(model->P, extra->P, G, X, reinterpret_cast<hipStream_t>(stream));
}

int splatraster_adam_step(int32_t n_groups, const splatraster_adam_group* groups, double beta1, double beta2, double eps,
                          float row_gate_threshold, void* stream)
{
    if (n_groups < 0 || n_groups > ADAM_MAX_GROUPS || (n_groups > 0 && !groups)) return SPLATRASTER_ERR_BAD_ARG;
    if (n_groups == 0) return SPLATRASTER_OK;
    return adam_step(n_groups, groups, beta1, beta2, eps, row_gate_threshold, 0, nullptr, nullptr, reinterpret_cast<hipStream_t>(stream));
}

int splatraster_adam_step_radii(int32_t n_groups, const splatraster_adam_group* groups, double beta1, double beta2, double eps,
                                float row_gate_threshold, int32_t P, const int32_t* radii, float* max_radii2D, void* stream)
{
    if (n_groups < 0 || n_groups > ADAM_MAX_GROUPS || (n_groups > 0 && !groups) || P < 0 || (P > 0 && (!radii || !max_radii2D)))
        return SPLATRASTER_ERR_BAD_ARG;
    return adam_step(n_groups, groups, beta1, beta2, eps, row_gate_threshold, P, radii, max_radii2D, reinterpret_cast<hipStream_t>(stream));
}

size_t splatraster_isotropic_loss_workspace_bytes(int32_t P) { return isotropic_workspace_bytes(P); }

int splatraster_isotropic_loss(int32_t P, int32_t scaling_cols, const float* scaling, const float* marker,
                               float* row_grad, float* out, void* workspace, void* stream)
{
    if (P < 0 || (scaling_cols != 1 && scaling_cols != 3)) return SPLATRASTER_ERR_BAD_ARG;
    if (!out || !workspace) return SPLATRASTER_ERR_BAD_ARG;
    if (P > 0 && (!scaling || !marker || !row_grad)) return SPLATRASTER_ERR_BAD_ARG;
    return launch_isotropic(P, scaling_cols, scaling, marker, row_grad, out, workspace,
                            reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
