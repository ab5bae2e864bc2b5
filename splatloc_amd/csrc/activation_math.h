// activation_math.h — the parameter activations' arithmetic as shared device functions (gaussian_model.py:78-105,
// gaussian_renderer/__init__.py:84-102 at SH degree 0): activations.hip uses them in its own kernels, preprocess_bwd.hip in
// the backward that writes RAW-parameter gradients directly (no activation-backward launch).  ONE definition, compiled with
// the same flags in both translation units, so that the two paths round identically (tests/test_gpu_activations.py holds the
// fused backward against the two-kernel one bit for bit).  preprocess.hip is compiled with -ffp-contract=off (its projection must round
// like the CPU oracle's): every function here states `#pragma clang fp contract(fast)` itself, so that it contracts the same way in
// every translation unit.
#pragma once
#include "common.h"

namespace sr {

constexpr float ACT_SH_C0 = 0.28209479177387814f;

// normalize: n = max(||q||, 1e-12)
__device__ __forceinline__ float act_quat_norm(const float4 q)
{
#pragma clang fp contract(fast)
    return fmaxf(sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w), 1e-12f);
}

// d normalize: (g - u (u . g)) / n with u = q / n
__device__ __forceinline__ float4 act_normalize_bwd(const float4 q, const float4 g)
{
#pragma clang fp contract(fast)
    const float n = act_quat_norm(q);
    const float ux = q.x / n, uy = q.y / n, uz = q.z / n, uw = q.w / n;
    const float ug = ((ux * g.x + uy * g.y) + uz * g.z) + uw * g.w;
    return make_float4((g.x - ux * ug) / n, (g.y - uy * ug) / n, (g.z - uz * ug) / n, (g.w - uw * ug) / n);
}

__device__ __forceinline__ float act_sigmoid(const float x)
{
#pragma clang fp contract(fast)
    return 1.0f / (1.0f + expf(-x));
}

// d sigmoid given the upstream gradient and the logit
__device__ __forceinline__ float act_sigmoid_bwd(const float g, const float logit)
{
#pragma clang fp contract(fast)
    const float s = act_sigmoid(logit);
    return g * s * (1.0f - s);
}

// SH degree 0: rgb before the clamp = C0 f_dc + 0.5
// (two roundings — what torch's `C0 * sh + 0.5` computes — under `contract(off)`: HIP's __fmul_rn / __fadd_rn are plain `*` / `+` and
//  contract like them; the first version of this header left the expression to the translation unit's flag and 13 % of the colour
//  elements came out 1 ulp apart between activations.hip and preprocess.hip)
__device__ __forceinline__ float act_rgb_raw_deg0(const float f_dc)
{
#pragma clang fp contract(off)
    const float p = ACT_SH_C0 * f_dc;
    return p + 0.5f;
}

// a + (w s) r with every operation rounded separately (torch's `a + ((w * s) * r)` on float32 tensors)
__device__ __forceinline__ float act_add_scaled(const float a, const float w, const float s, const float r)
{
#pragma clang fp contract(off)
    const float ws = w * s;
    const float t = ws * r;
    return a + t;
}

// q / max(||q||, 1e-12)
__device__ __forceinline__ float4 act_normalize(const float4 q)
{
#pragma clang fp contract(fast)
    const float n = act_quat_norm(q);
    return make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
}

}  // namespace sr
