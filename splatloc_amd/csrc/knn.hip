// knn.hip — simple_knn._C.distCUDA2 (gaussian_splatting/scene/gaussian_model.py:18,206):
// mean squared distance from every point to its 3 nearest other points.
//
// MI355X-first design: exact tiled brute force.  N is 5-20 k per key-frame in SplatLoc
// (SURVEY.md §2a), i.e. <= 4e8 distance evaluations (~10 VALU each) — far below a
// millisecond on 256 CUs, so the lineage's Morton sort + box pruning is not needed at
// these sizes.  Candidates are staged through LDS in 256-point tiles (coalesced loads,
// broadcast reads); the candidate range is split over gridDim.y slices so small N still
// fills the chip, and a second tiny kernel merges the per-slice top-3.
// Compiled with -ffp-contract=off: d2 = (dx*dx + dy*dy) + dz*dz rounds like the oracle.
#include "common.h"

namespace sr {

constexpr int KNN_THREADS = 256;

__device__ __forceinline__ void top3_insert(float d, float& b0, float& b1, float& b2)
{
    if (d < b2) {
        if (d < b1) {
            b2 = b1;
            if (d < b0) { b1 = b0; b0 = d; } else { b1 = d; }
        } else {
            b2 = d;
        }
    }
}

__global__ void __launch_bounds__(KNN_THREADS)
knn_partial_kernel(int N, int slice_len, const float* __restrict__ pts, float* __restrict__ partial)
{
    __shared__ float sx[KNN_THREADS], sy[KNN_THREADS], sz[KNN_THREADS];
    const int i = blockIdx.x * KNN_THREADS + threadIdx.x;
    const bool valid = i < N;
    const float x = valid ? pts[3 * i] : 0.f, y = valid ? pts[3 * i + 1] : 0.f, z = valid ? pts[3 * i + 2] : 0.f;
    float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
    const int jbeg = blockIdx.y * slice_len;
    const int jend = min(N, jbeg + slice_len);
    for (int base = jbeg; base < jend; base += KNN_THREADS) {
        const int j = base + threadIdx.x;
        __syncthreads();
        if (j < jend) { sx[threadIdx.x] = pts[3 * j]; sy[threadIdx.x] = pts[3 * j + 1]; sz[threadIdx.x] = pts[3 * j + 2]; }
        __syncthreads();
        const int cnt = min(KNN_THREADS, jend - base);
        for (int k = 0; k < cnt; ++k) {
            const float dx = sx[k] - x, dy = sy[k] - y, dz = sz[k] - z;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (base + k != i) top3_insert(d, b0, b1, b2);
        }
    }
    if (valid) {
        float* o = partial + ((size_t)blockIdx.y * N + i) * 3;
        o[0] = b0; o[1] = b1; o[2] = b2;
    }
}

__global__ void __launch_bounds__(KNN_THREADS)
knn_merge_kernel(int N, int slices, const float* __restrict__ partial, float* __restrict__ out)
{
    const int i = blockIdx.x * KNN_THREADS + threadIdx.x;
    if (i >= N) return;
    float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
    for (int s = 0; s < slices; ++s) {
        const float* p = partial + ((size_t)s * N + i) * 3;
        top3_insert(p[0], b0, b1, b2);
        top3_insert(p[1], b0, b1, b2);
        top3_insert(p[2], b0, b1, b2);
    }
    float sum = 0.f;
    sum += (b0 == INFINITY) ? 0.f : b0;
    sum += (b1 == INFINITY) ? 0.f : b1;
    sum += (b2 == INFINITY) ? 0.f : b2;
    out[i] = sum / 3.0f;
}

static int knn_slices(int N)
{
    const int qblocks = (N + KNN_THREADS - 1) / KNN_THREADS;
    int s = (2048 + qblocks - 1) / qblocks;          // aim for >= 2048 workgroups
    const int max_s = (N + KNN_THREADS - 1) / KNN_THREADS;  // at least one tile per slice
    if (s > max_s) s = max_s;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return s;
}

size_t knn_workspace_bytes(int32_t N)
{
    const size_t n = (size_t)(N > 0 ? N : 1);
    return align_up((size_t)knn_slices((int)n) * n * 3 * sizeof(float), 256);
}

int knn_dist2(int32_t N, const float* points, float* out, void* workspace, hipStream_t stream)
{
    const int slices = knn_slices(N);
    int slice_len = (N + slices - 1) / slices;
    slice_len = (slice_len + KNN_THREADS - 1) / KNN_THREADS * KNN_THREADS;
    const int qblocks = (N + KNN_THREADS - 1) / KNN_THREADS;
    float* partial = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(knn_partial_kernel, dim3(qblocks, slices), dim3(KNN_THREADS), 0, stream, N, slice_len,
                       points, partial);
    SR_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_merge_kernel, dim3(qblocks), dim3(KNN_THREADS), 0, stream, N, slices, partial, out);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

}  // namespace sr
