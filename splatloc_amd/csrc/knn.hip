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
//
// From KNN_GRID_MIN points on (and for a whole map at once: 500 k points would be 2.5e11 evaluations) the same EXACT result
// comes from a uniform grid: bounding box -> cells of ~4 points -> counting sort of the points by cell -> every
// point searches the rings of cells around its own until no unvisited cell can hold anything nearer than its current
// third-nearest.  Same distance arithmetic and the same 3 smallest values as the brute force, so the output is
// bit-identical (tests/test_gpu_parity.py::test_dist2_grid_equals_brute_force); no host synchronisation.
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
knn_partial_kernel(int N, int slice_len, const float* __restrict__ pts, float* __restrict__ partial,
                   const uint32_t* __restrict__ gate /*null, or: run only when *gate > gate_limit (fallback of an overloaded grid)*/,
                   uint32_t gate_limit)
{
    if (gate && !(*gate > gate_limit)) return;
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
            if (base + k != i && d < INFINITY) top3_insert(d, b0, b1, b2);
        }
    }
    if (valid) {
        float* o = partial + ((size_t)blockIdx.y * N + i) * 3;
        o[0] = b0; o[1] = b1; o[2] = b2;
    }
}

__global__ void __launch_bounds__(KNN_THREADS)
knn_merge_kernel(int N, int slices, const float* __restrict__ partial, float* __restrict__ out,
                 const uint32_t* __restrict__ gate, uint32_t gate_limit)
{
    if (gate && !(*gate > gate_limit)) return;
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

// ---- exact grid search ------------------------------------------------------------------------------------------
constexpr int KNN_GRID_MIN_DEFAULT = 10000;   // measured crossover on MI355X (profiles/r03_knn.json): 5 k 0.11 vs 0.16 ms, 20 k 0.34 vs 0.19 ms
static int g_knn_grid_min = KNN_GRID_MIN_DEFAULT;
void knn_set_grid_min(int n) { g_knn_grid_min = n < 0 ? KNN_GRID_MIN_DEFAULT : n; }

struct KnnGrid {            // written by knn_grid_setup_kernel, read by the others (device memory)
    float ox, oy, oz;       // origin (bounding-box minimum)
    float h, inv_h;         // cell edge
    int nx, ny, nz;
    uint32_t cells;
};
// A degenerate cloud (many duplicates, or a few far outliers that inflate the box while everything else falls into a handful
// of cells) would make every query scan one huge cell serially.  The count pass records the largest cell; beyond
// knn_overload_limit(N) points in one cell the grid query kernel returns at once and the tiled brute force — launched behind
// it, returning at once otherwise — produces the (identical) result.  Decided on the device: no host synchronisation.
// The flag word lives at the END of the workspace, outside both paths' buffers (they alias each other).
__host__ __device__ static inline uint32_t knn_overload_limit(int N)
{
    const uint32_t a = 4096u, b = (uint32_t)N / 64u;
    return a > b ? a : b;
}
constexpr int KNN_BBOX_BLOCKS = 256;

__global__ void __launch_bounds__(KNN_THREADS)
knn_bbox_kernel(int N, const float* __restrict__ pts, float* __restrict__ part /*[KNN_BBOX_BLOCKS][6]*/)
{
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * KNN_THREADS + threadIdx.x; i < N; i += gridDim.x * KNN_THREADS)
#pragma unroll
        for (int k = 0; k < 3; ++k) {   // (a non-finite coordinate must not inflate the box: such a point is nobody's neighbour)
            const float v = pts[3 * i + k];
            if (isfinite(v)) { lo[k] = fminf(lo[k], v); hi[k] = fmaxf(hi[k], v); }
        }
    __shared__ float s[KNN_THREADS / WAVE][6];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int d = 1; d < WAVE; d <<= 1) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], d, WAVE)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d, WAVE)); }
    if ((threadIdx.x & (WAVE - 1)) == 0)
#pragma unroll
        for (int k = 0; k < 3; ++k) { s[threadIdx.x / WAVE][k] = lo[k]; s[threadIdx.x / WAVE][3 + k] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s[0][threadIdx.x];
        for (int w = 1; w < KNN_THREADS / WAVE; ++w) v = threadIdx.x < 3 ? fminf(v, s[w][threadIdx.x]) : fmaxf(v, s[w][threadIdx.x]);
        part[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// one thread: bounding box -> cell edge such that a cell holds ~4 points and the grid has at most max_cells cells
__global__ void knn_grid_setup_kernel(int N, int nparts, const float* __restrict__ part, uint32_t max_cells, KnnGrid* __restrict__ G)
{
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = 0; b < nparts; ++b)
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], part[b * 6 + k]); hi[k] = fmaxf(hi[k], part[b * 6 + 3 + k]); }
    float ext[3];
    float emax = 0.f;
    for (int k = 0; k < 3; ++k) { ext[k] = hi[k] - lo[k]; if (!(ext[k] >= 0.f) || !isfinite(ext[k])) ext[k] = 0.f; emax = fmaxf(emax, ext[k]); }
    if (!(emax > 0.f)) emax = 1.f;
    for (int k = 0; k < 3; ++k) ext[k] = fmaxf(ext[k], 1e-3f * emax);     // flat clouds: a thin but non-degenerate box
    float h = cbrtf(ext[0] * ext[1] * ext[2] * 4.0f / (float)N);
    if (!(h > 0.f) || !isfinite(h)) h = emax;
    int nx, ny, nz;
    for (int it = 0; it < 64; ++it) {
        nx = (int)fminf(ceilf(ext[0] / h) + 1.f, 2048.f); ny = (int)fminf(ceilf(ext[1] / h) + 1.f, 2048.f);
        nz = (int)fminf(ceilf(ext[2] / h) + 1.f, 2048.f);
        if ((uint64_t)nx * ny * nz <= max_cells && ceilf(ext[0] / h) < 2047.f && ceilf(ext[1] / h) < 2047.f && ceilf(ext[2] / h) < 2047.f) break;
        h *= 1.26f;
    }
    if ((uint64_t)nx * ny * nz > max_cells) { nx = ny = nz = 1; h = 2.f * emax; }
    G->ox = lo[0]; G->oy = lo[1]; G->oz = lo[2];
    G->h = h; G->inv_h = 1.0f / h;
    G->nx = nx; G->ny = ny; G->nz = nz;
    G->cells = (uint32_t)(nx * ny * nz);
}

__device__ __forceinline__ void knn_cell_of(const KnnGrid& g, float x, float y, float z, int& ix, int& iy, int& iz)
{
    ix = min(g.nx - 1, max(0, (int)((x - g.ox) * g.inv_h)));
    iy = min(g.ny - 1, max(0, (int)((y - g.oy) * g.inv_h)));
    iz = min(g.nz - 1, max(0, (int)((z - g.oz) * g.inv_h)));
}

__global__ void __launch_bounds__(KNN_THREADS)
knn_count_kernel(int N, const float* __restrict__ pts, const KnnGrid* __restrict__ Gp, uint32_t* __restrict__ cell_of,
                 uint32_t* __restrict__ count, uint32_t* __restrict__ max_count)
{
    const int i = blockIdx.x * KNN_THREADS + threadIdx.x;
    if (i >= N) return;
    const KnnGrid g = *Gp;
    int ix, iy, iz;
    knn_cell_of(g, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], ix, iy, iz);
    const uint32_t c = (uint32_t)ix + (uint32_t)g.nx * ((uint32_t)iy + (uint32_t)g.ny * (uint32_t)iz);
    cell_of[i] = c;
    const uint32_t now = atomicAdd(&count[c], 1u) + 1u;
    if (now > knn_overload_limit(N)) atomicMax(max_count, now);     // (rare: only cells already beyond the limit touch the word)
}

// sorted[start[c] + k] = (x, y, z, original index) of the k-th point that reached cell c (any order inside a cell)
__global__ void __launch_bounds__(KNN_THREADS)
knn_scatter_kernel(int N, const float* __restrict__ pts, const uint32_t* __restrict__ cell_of,
                   const uint32_t* __restrict__ incl /*inclusive scan of count*/, uint32_t* __restrict__ fill,
                   float4* __restrict__ sorted)
{
    const int i = blockIdx.x * KNN_THREADS + threadIdx.x;
    if (i >= N) return;
    const uint32_t c = cell_of[i];
    const uint32_t start = c ? incl[c - 1] : 0u;
    const uint32_t k = atomicAdd(&fill[c], 1u);
    sorted[start + k] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], __uint_as_float((uint32_t)i));
}

__global__ void __launch_bounds__(KNN_THREADS)
knn_grid_query_kernel(int N, const float4* __restrict__ sorted, const uint32_t* __restrict__ incl,
                      const KnnGrid* __restrict__ Gp, const uint32_t* __restrict__ max_count, float* __restrict__ out)
{
    const int s = blockIdx.x * KNN_THREADS + threadIdx.x;   // queries in cell order: neighbouring threads walk neighbouring cells
    if (s >= N) return;
    if (*max_count > knn_overload_limit(N)) return;          // overloaded grid: the brute force behind this launch answers
    const KnnGrid g = *Gp;
    const float4 q = sorted[s];
    const uint32_t self = __float_as_uint(q.w);
    int cx, cy, cz;
    knn_cell_of(g, q.x, q.y, q.z, cx, cy, cz);
    float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
    const int rmax = max(g.nx, max(g.ny, g.nz));
    for (int r = 0; r <= rmax; ++r) {
        const int z0 = max(0, cz - r), z1 = min(g.nz - 1, cz + r);
        const int y0 = max(0, cy - r), y1 = min(g.ny - 1, cy + r);
        const int x0 = max(0, cx - r), x1 = min(g.nx - 1, cx + r);
        for (int iz = z0; iz <= z1; ++iz)
            for (int iy = y0; iy <= y1; ++iy) {
                const bool shell_row = (iz == cz - r) || (iz == cz + r) || (iy == cy - r) || (iy == cy + r);
                // inside a shell row every cell of the row belongs to ring r; otherwise only its two end cells do
                const int step = shell_row ? 1 : max(1, 2 * r);
                for (int ix = shell_row ? x0 : cx - r; ix <= x1; ix += step) {
                    if (ix < x0) continue;
                    const uint32_t c = (uint32_t)ix + (uint32_t)g.nx * ((uint32_t)iy + (uint32_t)g.ny * (uint32_t)iz);
                    const uint32_t beg = c ? incl[c - 1] : 0u, end = incl[c];
                    for (uint32_t j = beg; j < end; ++j) {
                        const float4 p = sorted[j];
                        const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
                        const float d = (dx * dx + dy * dy) + dz * dz;
                        if (__float_as_uint(p.w) != self && d < INFINITY) top3_insert(d, b0, b1, b2);   // (d < inf: a non-finite point is nobody's neighbour)
                    }
                }
            }
        // every unvisited point lies outside the cube of cells [c - r, c + r]^3 (clipped to the grid: beyond the grid
        // there are no points).  Distance from q to the nearest face that still has cells behind it, with a safety
        // margin for the rounding of the cell assignment:
        float m = INFINITY;
        if (cx - r > 0) m = fminf(m, q.x - (g.ox + (float)(cx - r) * g.h));
        if (cx + r < g.nx - 1) m = fminf(m, (g.ox + (float)(cx + r + 1) * g.h) - q.x);
        if (cy - r > 0) m = fminf(m, q.y - (g.oy + (float)(cy - r) * g.h));
        if (cy + r < g.ny - 1) m = fminf(m, (g.oy + (float)(cy + r + 1) * g.h) - q.y);
        if (cz - r > 0) m = fminf(m, q.z - (g.oz + (float)(cz - r) * g.h));
        if (cz + r < g.nz - 1) m = fminf(m, (g.oz + (float)(cz + r + 1) * g.h) - q.z);
        if (m == INFINITY) break;                       // the whole grid has been visited
        m = fmaxf(0.f, m - 1e-3f * g.h);
        if (b2 < (m * m) * 0.9999f) break;              // nothing unvisited can be nearer than the current third-nearest
    }
    float sum = 0.f;
    sum += (b0 == INFINITY) ? 0.f : b0;
    sum += (b1 == INFINITY) ? 0.f : b1;
    sum += (b2 == INFINITY) ? 0.f : b2;
    out[self] = sum / 3.0f;
}

static uint32_t knn_max_cells(int N)
{
    uint32_t c = 1024;
    while (c < (uint32_t)N / 2u && c < (1u << 22)) c <<= 1;
    return c;
}

struct KnnGridWs { KnnGrid* G; float* part; uint32_t* count; uint32_t* fill; uint32_t* incl; uint32_t* cell_of; float4* sorted; void* scan_tmp; size_t bytes, zero_bytes; };
static KnnGridWs knn_grid_ws(void* base, int N)
{
    const size_t n = (size_t)(N > 0 ? N : 1), cells = knn_max_cells(N);
    char* b = reinterpret_cast<char*>(base);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    KnnGridWs w;
    // count and fill first: one memset clears both
    w.count = reinterpret_cast<uint32_t*>(b + take(4 * cells));
    w.fill = reinterpret_cast<uint32_t*>(b + take(4 * cells));
    w.zero_bytes = o;
    w.G = reinterpret_cast<KnnGrid*>(b + take(sizeof(KnnGrid)));
    w.part = reinterpret_cast<float*>(b + take(sizeof(float) * 6 * KNN_BBOX_BLOCKS));
    w.incl = reinterpret_cast<uint32_t*>(b + take(4 * cells));
    w.cell_of = reinterpret_cast<uint32_t*>(b + take(4 * n));
    w.sorted = reinterpret_cast<float4*>(b + take(16 * n));
    w.scan_tmp = b + take(scan_tmp_bytes((int64_t)cells));
    w.bytes = o;
    return w;
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


// flag word of the grid path (largest overloaded cell, see KnnGrid): the last 256 bytes of the workspace
static uint32_t* knn_flag_word(void* workspace, int32_t N)
{
    return reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) + knn_workspace_bytes(N) - 256);
}

// the tiled brute force; `gate` non-null: every block returns at once unless *gate > gate_limit
static int knn_brute_dist2(int32_t N, const float* points, float* out, void* workspace, const uint32_t* gate, uint32_t gate_limit,
                           hipStream_t stream)
{
    const int slices = knn_slices(N);
    int slice_len = (N + slices - 1) / slices;
    slice_len = (slice_len + KNN_THREADS - 1) / KNN_THREADS * KNN_THREADS;
    const int qblocks = (N + KNN_THREADS - 1) / KNN_THREADS;
    float* partial = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(knn_partial_kernel, dim3(qblocks, slices), dim3(KNN_THREADS), 0, stream, N, slice_len,
                       points, partial, gate, gate_limit);
    SR_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_merge_kernel, dim3(qblocks), dim3(KNN_THREADS), 0, stream, N, slices, partial, out, gate, gate_limit);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

static int knn_grid_dist2(int32_t N, const float* points, float* out, void* workspace, hipStream_t stream)
{
    KnnGridWs w = knn_grid_ws(workspace, N);
    const uint32_t cells = knn_max_cells(N);
    const int blocks = (N + KNN_THREADS - 1) / KNN_THREADS;
    SR_HIP_CHECK(hipMemsetAsync(w.count, 0, w.zero_bytes, stream));
    uint32_t* flag = knn_flag_word(workspace, N);
    SR_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(KNN_BBOX_BLOCKS), dim3(KNN_THREADS), 0, stream, N, points, w.part);
    SR_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_grid_setup_kernel, dim3(1), dim3(1), 0, stream, N, KNN_BBOX_BLOCKS, w.part, cells, w.G);
    SR_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_count_kernel, dim3(blocks), dim3(KNN_THREADS), 0, stream, N, points, w.G, w.cell_of, w.count, flag);
    SR_LAUNCH_CHECK();
    // the scan covers all `cells` slots (the grid uses a prefix of them; the rest hold zero counts)
    int st = lookback_error_init();
    if (st) return st;
    st = inclusive_scan_u32((int64_t)cells, w.count, nullptr, w.incl, nullptr, w.scan_tmp, stream);
    if (st) return st;
    hipLaunchKernelGGL(knn_scatter_kernel, dim3(blocks), dim3(KNN_THREADS), 0, stream, N, points, w.cell_of, w.incl, w.fill, w.sorted);
    SR_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_grid_query_kernel, dim3(blocks), dim3(KNN_THREADS), 0, stream, N, w.sorted, w.incl, w.G, flag, out);
    SR_LAUNCH_CHECK();
    // an overloaded grid (degenerate cloud) is answered by the brute force instead; otherwise these blocks return at once
    return knn_brute_dist2(N, points, out, workspace, flag, knn_overload_limit(N), stream);
}

size_t knn_workspace_bytes(int32_t N)
{
    const size_t n = (size_t)(N > 0 ? N : 1);
    const size_t brute = align_up((size_t)knn_slices((int)n) * n * 3 * sizeof(float), 256);
    const size_t grid = knn_grid_ws(nullptr, (int)n).bytes;
    return (brute > grid ? brute : grid) + 256;     // either path may be taken (the threshold is a run-time knob); + the flag word
}

int knn_dist2(int32_t N, const float* points, float* out, void* workspace, hipStream_t stream)
{
    if (N >= g_knn_grid_min && N >= 8) return knn_grid_dist2(N, points, out, workspace, stream);
    return knn_brute_dist2(N, points, out, workspace, nullptr, 0u, stream);
}

}  // namespace sr
