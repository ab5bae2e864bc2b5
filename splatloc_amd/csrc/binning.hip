// binning.hip — depth keys, tile-instance emission and per-tile ranges.
//
// The lineage sorts R = sum(tiles_touched) 64-bit (tile | depth) keys with 6 radix passes.
// Here the same ordering (tile, depth bits, Gaussian index) is produced with far less
// traffic:
//   1. stable sort of the P Gaussians by depth bits (32-bit keys, P elements; the keys are
//      written by preprocess_kernel);
//   2. instances are emitted IN DEPTH ORDER, one thread per instance (coalesced writes);
//   3. a stable sort of the R instances on the tile id only (<= 16 bits, 2 passes).
// Because both sorts are stable, ties resolve exactly as in the 64-bit formulation.
#include "tile_order.h"
#include "composite_common.h"

namespace sr {

__device__ __forceinline__ int f2i_sat_b(float v)
{
    if (!(v > -1.0e9f)) v = -1.0e9f;
    if (!(v < 1.0e9f)) v = 1.0e9f;
    return (int)v;
}

// Instance j belongs to the Gaussian of depth rank r = first r with offsets[r] > j (offsets =
// inclusive scan of tiles_touched in depth order); k = j - offsets[r-1] is its tile slot inside
// the Gaussian's rect.  A workgroup emits EMIT_SPAN consecutive instances: the ranks they
// belong to are consecutive too (every visible Gaussian has >= 1 tile; culled ones sort to the
// end), so the workgroup finds its first rank with a cooperative 256-ary search (3 dependent
// loads at P = 500k instead of 19 per thread), stages <= EMIT_SPAN + 1 offsets in LDS and every
// thread finishes with a binary search in LDS.
constexpr int EMIT_BLOCK = 256;
constexpr int EMIT_IPT = 4;
static_assert(EMIT_SPAN == EMIT_BLOCK * EMIT_IPT, "common.h: EMIT_SPAN");

// A window of V views: the n = V * P rows of all views are in ONE depth order; row g = v * P + i emits the
// global tile ids v * tiles + t of its rect, so the single stable tile sort that follows leaves every
// (view, tile) list in (depth, index) order — exactly what V separate calls produce.
__global__ void __launch_bounds__(EMIT_BLOCK)
emit_kernel(int64_t R, int P /*rows: V * P*/, int Pv /*Gaussians per view*/, int tiles_per_view, int W, int H,
            const uint32_t* __restrict__ offsets,
            const uint32_t* __restrict__ depth_order, const float4* __restrict__ rec,
            uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
            const uint32_t* __restrict__ span_owner /* first rank of span k, or NULL */,
            uint32_t* __restrict__ ranges, uint32_t nranges /* words to clear for payload_kernel */)
{
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < nranges; w += gridDim.x * blockDim.x) ranges[w] = 0u;
    __shared__ uint32_t s_off[EMIT_SPAN + 1];  // s_off[q] = offsets[r_lo - 1 + q] (0 before the first)
    __shared__ uint32_t s_gid[EMIT_SPAN], s_org[EMIT_SPAN], s_w[EMIT_SPAN];
    __shared__ int s_min;
    const int t = threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.x * EMIT_SPAN;
    const uint32_t j0u = (uint32_t)j0;
    // ---- first rank with offsets[r] > j0: handed over by the scan, else searched for ----
    int lo = 0, hi = P;
    if (span_owner && blockIdx.x < SPAN_OWNER_CAP) {
        lo = (int)span_owner[blockIdx.x];
        hi = lo + 1;
    }
    while (hi - lo > 1) {
        const int n = hi - lo;
        const int step = (n + EMIT_BLOCK - 1) / EMIT_BLOCK;
        if (t == 0) s_min = EMIT_BLOCK;
        __syncthreads();
        const int last = lo + (t + 1) * step - 1;  // last element of this thread's segment
        if (last >= hi - 1 || offsets[last] > j0u) atomicMin(&s_min, t);
        __syncthreads();
        const int seg = s_min;  // the segment whose last element is the first one > j0 (or the tail)
        __syncthreads();
        lo = lo + seg * step;
        hi = min(hi, lo + step);
    }
    const int r_lo = lo;  // (hi - lo == 1: offsets[lo] > j0 because j0 < R = offsets[P - 1])
    // ---- stage the offsets of ranks r_lo - 1 ... r_lo + EMIT_SPAN - 1 ----
    for (int q = t; q <= EMIT_SPAN; q += EMIT_BLOCK) {
        const int r = r_lo - 1 + q;
        s_off[q] = r < 0 ? 0u : offsets[min(r, P - 1)];
    }
    __syncthreads();
    // ---- per-rank data (id, rect origin, rect width) of the ranks this span touches: one gather
    //      per Gaussian instead of one per instance ----
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const uint32_t j_last = (uint32_t)(min(R, j0 + (int64_t)EMIT_SPAN) - 1);
    int nr;  // ranks r_lo ... r_lo + nr - 1
    {
        int a = 1, b = EMIT_SPAN + 1;
        while (a < b) {
            const int mid = (a + b) >> 1;
            if (s_off[mid] > j_last) b = mid; else a = mid + 1;
        }
        nr = a;
    }
    for (int q = t; q < nr; q += EMIT_BLOCK) {
        const uint32_t g = depth_order[r_lo + q] & 0xFFFFFFu;
        const float4 p = rec[2 * g];
        const float rf = p.w;  // integer-valued radius stored by preprocess
        const int rminx = min(gx, max(0, f2i_sat_b((p.x - rf) / (float)TILE)));
        const int rminy = min(gy, max(0, f2i_sat_b((p.y - rf) / (float)TILE)));
        const int rmaxx = min(gx, max(0, f2i_sat_b((p.x + rf + (float)(TILE - 1)) / (float)TILE)));
        s_gid[q] = g;
        const uint32_t v = (P == Pv) ? 0u : g / (uint32_t)Pv;   // once per row, not per instance
        s_org[q] = v * (uint32_t)tiles_per_view + (uint32_t)rminy * (uint32_t)gx + (uint32_t)rminx;  // global id of the rect's first tile
        s_w[q] = (uint32_t)(rmaxx - rminx);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < EMIT_IPT; ++it) {
        const int64_t j = j0 + it * EMIT_BLOCK + t;
        if (j < R) {
            const uint32_t ju = (uint32_t)j;
            int a = 1, b = nr + 1;  // first q in [1, nr] with s_off[q] > j  (rank r_lo - 1 + q)
            while (a < b) {
                const int mid = (a + b) >> 1;
                if (s_off[mid] > ju) b = mid; else a = mid + 1;
            }
            const uint32_t k = ju - s_off[a - 1];
            const uint32_t w = s_w[a - 1];
            const uint32_t row = k / w;
            keys[j] = s_org[a - 1] + row * (uint32_t)gx + (k - row * w);
            vals[j] = s_gid[a - 1];
        }
    }
}

int launch_emit(const splatraster_settings& s, int32_t P, int32_t V, int64_t R, const GeomView& g, uint32_t* keys,
                uint32_t* vals, uint32_t* ranges, uint32_t nranges, hipStream_t stream)
{
    if (R == 0) return SPLATRASTER_OK;
    const int64_t blocks = (R + EMIT_SPAN - 1) / EMIT_SPAN;
    const int n = P * V;
    const int tiles = ((s.image_width + TILE - 1) / TILE) * ((s.image_height + TILE - 1) / TILE);
    hipLaunchKernelGGL(emit_kernel, dim3((unsigned)blocks), dim3(EMIT_BLOCK), 0, stream, R, n, P, tiles, s.image_width,
                       s.image_height, g.offsets, g.depth_order, g.rec, keys, vals,
                       scan_state_bytes(n) ? g.span_owner : nullptr, ranges, nranges);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

// Zeroes the per-tile range table (empty tiles keep [0, 0)); the boundaries themselves are written
// by payload_kernel, which walks the sorted list anyway.
int launch_ranges_clear(int32_t tiles, uint32_t* ranges, hipStream_t stream)
{
    SR_HIP_CHECK(hipMemsetAsync(ranges, 0, sizeof(uint32_t) * 2 * (size_t)tiles, stream));
    return SPLATRASTER_OK;
}

// Per-instance payload, written once per frame in sorted order so that every later reader
// (forward and backward compositing, four quadrant-waves per tile) streams it with coalesced
// loads instead of chasing id -> record through 16-byte gathers scattered over HBM:
//   irec[2j] = (pixel x, y, depth, radius) of point_list[j], irec[2j+1] = its conic pre-scaled for
//   the compositing kernels (payload_conic) and opacity;  ipack[j] = point_list[j] | reach bits << 24 (ids < 2^24: checked on the host).
#ifndef SR_PAYLOAD_NT_MIN
#define SR_PAYLOAD_NT_MIN (8ll << 20)   // instances from which the payload is written with streaming stores (36 B each: 288 MB)
#endif
static int64_t g_payload_stream_min = SR_PAYLOAD_NT_MIN;
void set_payload_stream_min(int64_t instances) { g_payload_stream_min = instances < 0 ? (int64_t)SR_PAYLOAD_NT_MIN : instances; }

template <bool NT>
__global__ void __launch_bounds__(256)
payload_kernel(int64_t R, int gx, int tiles_per_view, int V, const uint32_t* __restrict__ point_list,
               const uint32_t* __restrict__ tile_list, const float4* __restrict__ rec,
               float4* __restrict__ irec, uint32_t* __restrict__ ipack, uint32_t* __restrict__ ranges)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= R) return;
    const uint32_t g = point_list[j], t = tile_list[j];
    const float4 a0 = rec[2 * (size_t)g], a1 = rec[2 * (size_t)g + 1];  // one 32-byte gather
    uint32_t tl = t;   // tile inside its view (V <= 8: a few compares instead of a division)
#pragma unroll 1
    for (int v = 1; v < V && tl >= (uint32_t)tiles_per_view; ++v) tl -= (uint32_t)tiles_per_view;
    const uint32_t ty = tl / (uint32_t)gx, tx = tl - ty * (uint32_t)gx;
    const float4 c1 = payload_conic(a1);   // pre-scaled for gauss_log2 (composite_common.h)
    const uint32_t m = g | (quadrant_reach_mask(a0, a1, (float)(tx * TILE), (float)(ty * TILE)) << 24);
    if (NT) {
        // Streaming stores for lists larger than the memory-side cache (S2 window: 20 M instances, 657 MB): what is written
        // here is next read after the whole list has gone by, and must not push the 32-byte records this kernel gathers
        // out of the caches (S2: 0.436 -> 0.377 ms).  Smaller lists (a window of 640x480 frames: 6 M instances) are read
        // back FROM the caches by the compositing kernels: there plain stores win (0.122 vs 0.142 ms).
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        f32x4_t* out = reinterpret_cast<f32x4_t*>(irec) + 2 * j;
        __builtin_nontemporal_store((f32x4_t){a0.x, a0.y, a0.z, a0.w}, out);
        __builtin_nontemporal_store((f32x4_t){c1.x, c1.y, c1.z, c1.w}, out + 1);
        __builtin_nontemporal_store(m, ipack + j);
    } else {
        irec[2 * j] = a0;
        irec[2 * j + 1] = c1;
        ipack[j] = m;
    }
    // per-tile [start, end) of the sorted list (the table was zeroed for the empty tiles)
    if (j == 0 || tile_list[j - 1] != t) ranges[2 * t] = (uint32_t)j;
    if (j == R - 1 || tile_list[j + 1] != t) ranges[2 * t + 1] = (uint32_t)(j + 1);
}

int launch_payload(const splatraster_settings& s, int32_t V, int64_t R, const GeomView& g, const BinView& b, hipStream_t stream)
{
    if (R == 0) return SPLATRASTER_OK;
    const int gx = (s.image_width + TILE - 1) / TILE, gy = (s.image_height + TILE - 1) / TILE;
    if (R >= g_payload_stream_min)
        hipLaunchKernelGGL(payload_kernel<true>, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, stream, R, gx, gx * gy, V,
                           b.point_list, b.tile_list, g.rec, b.irec, b.ipack, b.ranges);
    else
        hipLaunchKernelGGL(payload_kernel<false>, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, stream, R, gx, gx * gy, V,
                           b.point_list, b.tile_list, g.rec, b.irec, b.ipack, b.ranges);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

// Launch order of the compositing grids.  A window of SplatLoc's own 640x480 frames is 24 000 quadrant-waves on a machine
// with ~8 000 wave slots: three rounds, and in tile order the last round starts waves of every length — the kernel then
// waits for the longest of them (measured life times, list-scheduling model: 412 us in tile order, 363 us longest first,
// 338 us ideal).  One small block buckets the (view, tile) lists by length (16 entries per bucket, longest first); the
// compositing kernels map block -> tile_order[block's tile slot].  Which tile a wave works on changes, never what it computes.
// Only for launches of a few rounds (use_tile_order, common.h).
__global__ void __launch_bounds__(ORDER_THREADS)
tile_order_kernel(int T, const uint32_t* __restrict__ ranges, uint32_t* __restrict__ order, uint32_t* __restrict__ nparts)
{
    __shared__ uint32_t s_cnt[ORDER_BUCKETS];
    __shared__ uint32_t s_wsum[ORDER_THREADS / WAVE];
    tile_order_block(T, [&](int i) { return ranges[2 * i + 1] - ranges[2 * i]; }, order, s_cnt, s_wsum, nparts);
}

int launch_tile_order(const splatraster_settings& s, int32_t V, const BinView& b, hipStream_t stream)
{
    const int gx = (s.image_width + TILE - 1) / TILE, gy = (s.image_height + TILE - 1) / TILE;
    if (!use_tile_order(V, gx * gy)) return SPLATRASTER_OK;
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(ORDER_THREADS), 0, stream, V * gx * gy, b.ranges, b.tile_order, b.nparts);
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

// Feature rows are user input [P, C]; with C % 4 != 0 (C = 35: 140-byte rows) they are only
// 4-byte aligned, which forces 4-byte gathers in the compositing kernels.  One streaming pass
// per frame pads them to 16-byte aligned rows of CP = roundup4(C) floats (pad = 0).
__global__ void __launch_bounds__(256)
pad_features_kernel(int64_t n4 /* P * CP / 4 */, int C, int PPR, const float* __restrict__ feat,
                    float4* __restrict__ featp4)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte piece per thread
    if (e >= n4) return;
    const int64_t row = e / PPR;
    const int c = (int)(e - row * PPR) * 4;
    const float* src = feat + row * C + c;
    float4 v;
    v.x = src[0];
    v.y = c + 1 < C ? src[1] : 0.0f;
    v.z = c + 2 < C ? src[2] : 0.0f;
    v.w = c + 3 < C ? src[3] : 0.0f;
    featp4[e] = v;
}

int launch_pad_features(int32_t P, int C, const float* feat, float* featp, hipStream_t stream)
{
    const int CP = padded_channels(C);
    const int64_t n4 = (int64_t)P * CP / 4;
    if (n4 == 0 || CP == C) return SPLATRASTER_OK;
    hipLaunchKernelGGL(pad_features_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, n4, C, CP / 4,
                       feat, reinterpret_cast<float4*>(featp));
    SR_LAUNCH_CHECK();
    return SPLATRASTER_OK;
}

}  // namespace sr
