// binsort.hip — the tile-binned front end: counting sort of the tile instances by (view, tile) + one LDS sort per tile.
//
// The radix front end (binning.hip) orders the instances with two global sorts: the P rows by depth (4 one-sweep passes),
// then the R instances by tile id (2 passes) — 14 launches whose chains of look-backs and kernel boundaries are what a small
// frame pays for (SplatLoc's own 640x480: 170 of the 540 us of a color_refinement iteration, train_gaussians.py:269-297).
// Here the same (tile, depth bits, row) order comes out of four launches, none of which waits for a predecessor block:
//   1. bin_walk_kernel<.., false>  one block per chunk of 2048 rows of one view (8192 rows on frames of more than 2048
//                          tiles, where the table would outgrow the rows): LDS histogram of the tiles its rects cover
//                          -> table[(view, tile)][chunk] (every entry written: nothing to zero)
//   2. exclusive scan of the table (scan_sort.hip): entry = first slot of (tile, chunk) in the final list; the grand
//                          total is R on the device
//   3. bin_walk_kernel<.., true>   the same walk; an LDS atomic on the (tile, chunk) cursor hands out the slot; writes the 64-bit
//                          key (depth bits << 32 | row).  The order inside a (tile, chunk) piece is whatever the LDS atomics
//                          made it — the next kernel fixes it
//   4. bin_sort_tile_kernel  one block per (view, tile): the list is loaded into LDS and sorted by a register-blocked
//                          bitonic network whose compare-exchange is v_min_f64 / v_max_f64 on the keys read as doubles (see
//                          "the first sort launch" below), one thread per 16 keys, then written out as the payload the
//                          compositing kernels stream (irec, ipack) together with point_list, tile_list and the tile's range.
//                          Two instantiations: lists up to 2048 keys (128 threads, 17 KB of LDS: every tile of a 640x480
//                          frame resident at once) and up to 4096 (256 threads, 35 KB); the host picks by a hint the kernel
//                          itself raises in host-mapped memory when it meets a list beyond 2048 (a reconstructed room has a
//                          few dozen such tiles, the uniform S* clouds none): a hint, never a correctness matter.
//                          (History of this kernel, 1200 lists of ~1000 keys: the whole network in LDS with integer
//                          compare-exchanges 57 us; 256 threads x 4 keys in registers, DPP exchanges 48 - 60 us (VALU-issue-bound:
//                          3 700 instructions per wave, a wave64 VALU instruction occupies its SIMD for 4 cycles); one wave x
//                          32 keys per lane 73 us; this form 39 us, of which ~20 us are the 92 MB of payload traffic.)
// (depth bits, row) is unique inside a tile, so the sorted list is THE (tile, depth, index) order of the lineage's stable
// 64-bit sort — bit-identical point lists and ranges to the radix front end (tests/test_gpu_binsort.py runs both).
// Lists beyond the first launch's capacity are sorted by a second launch that finds them in the scanned table itself (1024 threads, the
// same LDS network on up to 16 384 keys in 139 KB of LDS; beyond 16 384 keys a normalised network runs on the global buffer: slow, correct).
//
// Compiled with -ffp-contract=off like preprocess.hip / binning.hip: the rect arithmetic must round like preprocess's.
#include <mutex>

#include "composite_common.h"
#include "tile_order.h"
#ifdef SR_BIN_TIMING   // tools/micro/tilesort_bench.hip: s_memtime stamps of every phase of a block
namespace sr { __device__ unsigned long long* g_bin_dbg = nullptr; }
#define SR_STAMP(k) do { if (threadIdx.x == 0 && g_bin_dbg) g_bin_dbg[8 * blockIdx.x + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SR_STAMP(k) do { } while (0)
#endif

namespace sr {

__device__ __forceinline__ int f2i_sat_c(float v)
{
    if (!(v > -1.0e9f)) v = -1.0e9f;
    if (!(v < 1.0e9f)) v = 1.0e9f;
    return (int)v;
}

// tile rect of a row from its preprocess record (pixel x, y, depth, radius) — the arithmetic of preprocess_kernel
struct Rect { int x0, y0, x1, y1; };
__device__ __forceinline__ Rect rect_of(const float4 p, int gx, int gy)
{
    const float rf = p.w;
    Rect r;
    r.x0 = min(gx, max(0, f2i_sat_c((p.x - rf) / (float)TILE)));
    r.y0 = min(gy, max(0, f2i_sat_c((p.y - rf) / (float)TILE)));
    r.x1 = min(gx, max(0, f2i_sat_c((p.x + rf + (float)(TILE - 1)) / (float)TILE)));
    r.y1 = min(gy, max(0, f2i_sat_c((p.y + rf + (float)(TILE - 1)) / (float)TILE)));
    return r;
}

__device__ __forceinline__ void wave_order()
{
    // LDS operations of one wave execute in issue order; this keeps the compiler from moving them across the point
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// grid (chunks, V), 1024 threads, ROWS rows of the view per thread; dynamic LDS: tiles words.  16 waves per block and at most
// a few dependent steps per thread: the walk is a chain of LDS atomic round trips (the returned slot addresses the store), so
// it is latency that more resident waves hide, not bandwidth (4 waves x 8 rows per block: 13 / 21 us for count / scatter at
// 500k rows, 640x480; 16 waves x 2 rows: 9 / 16 us).
// A rect of up to BIN_INLINE_TILES tiles is walked by the thread that owns the row; a larger one is queued in LDS and walked
// by a whole WAVE (lanes stride over its tiles) after the barrier: on a reconstructed room — walls seen at grazing angles,
// Gaussians over hundreds of tiles — the thread-serial walk of such a rect was the block's critical path (33 / 48 us for
// count / scatter on 196k rows where the uniform 500k-row cloud took 9 / 16 us).
constexpr int BIN_INLINE_TILES = 8, BIN_QUEUE = 1024;
template <int ROWS, bool SCATTER>
__global__ void __launch_bounds__(BIN_THREADS)
bin_walk_kernel(int P, int tiles, int gx, int gy, int nchunk, const float4* __restrict__ rec,
                uint32_t* __restrict__ table /*[(v * tiles + t) * nchunk + chunk]: COUNT out / exclusive prefix in*/,
                uint64_t* __restrict__ keys /*SCATTER: [R]*/)
{
    extern __shared__ uint32_t s_bin[];
    __shared__ uint64_t s_qkey[BIN_QUEUE];
    __shared__ uint32_t s_qrect[BIN_QUEUE];   // x0 | y0 << 8 | width << 16 | height << 24 (at most 255 tiles a side: BIN_MAX_TILES)
    __shared__ uint32_t s_qn;
    const int v = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
    uint32_t* col = table + (size_t)v * tiles * nchunk + chunk;
    if (t == 0) s_qn = 0u;
    // A block's rows are 64-row groups INTERLEAVED over the whole view (group j of the block = group j nchunk + chunk of the view):
    // a map's rows are spatially coherent — one 2048-row run can be a near wall whose every Gaussian covers hundreds of tiles —
    // and a block of consecutive rows then walks a hundred times the rects of its neighbours (Replica scale: scatter 49 us).
    auto row_of = [&](int k) { return ((k * (BIN_THREADS / WAVE) + (t >> 6)) * nchunk + chunk) * WAVE + (t & (WAVE - 1)); };
    float4 p[ROWS];   // every row's record is requested before the first one is used
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int i = row_of(k);
        p[k] = i < P ? rec[2 * ((size_t)v * P + i)] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int k = t; k < tiles; k += BIN_THREADS) s_bin[k] = SCATTER ? col[(size_t)k * nchunk] : 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        if (!(p[k].w > 0.0f)) continue;   // culled rows (and the padding beyond P) carry radius 0
        const uint32_t g = (uint32_t)v * (uint32_t)P + (uint32_t)row_of(k);
        const Rect r = rect_of(p[k], gx, gy);
        const uint64_t key = ((uint64_t)__float_as_uint(p[k].z) << 32) | g;
        const int w = r.x1 - r.x0, h = r.y1 - r.y0;
        if (w * h > BIN_INLINE_TILES) {
            const uint32_t q = atomicAdd(&s_qn, 1u);
            if (q < (uint32_t)BIN_QUEUE) {
                s_qkey[q] = key;
                s_qrect[q] = (uint32_t)r.x0 | ((uint32_t)r.y0 << 8) | ((uint32_t)w << 16) | ((uint32_t)h << 24);
                continue;
            }   // (queue full: walked here)
        }
        for (int y = r.y0; y < r.y1; ++y)
            for (int x = r.x0; x < r.x1; ++x) {
                const uint32_t slot = atomicAdd(&s_bin[y * gx + x], 1u);
                if (SCATTER) keys[slot] = key;
            }
    }
    __syncthreads();
    {   // the queued rects: one wave each, lanes stride over the tiles
        const uint32_t nq = min(s_qn, (uint32_t)BIN_QUEUE);
        const int lane = t & (WAVE - 1);
        for (uint32_t q = (uint32_t)t / WAVE; q < nq; q += BIN_THREADS / WAVE) {
            const uint64_t key = s_qkey[q];
            const uint32_t rc = s_qrect[q];
            const int x0 = (int)(rc & 255u), y0 = (int)((rc >> 8) & 255u), w = (int)((rc >> 16) & 255u), h = (int)(rc >> 24);
            for (int e = lane; e < w * h; e += WAVE) {
                const int yy = e / w, xx = e - yy * w;
                const uint32_t slot = atomicAdd(&s_bin[(y0 + yy) * gx + x0 + xx], 1u);
                if (SCATTER) keys[slot] = key;
            }
        }
    }
    if (!SCATTER) {
        __syncthreads();
        for (int k = t; k < tiles; k += BIN_THREADS) col[(size_t)k * nchunk] = s_bin[k];
    }
}

// ---- per-tile sort -------------------------------------------------------------------------
// what payload_kernel (binning.hip) writes for the sorted list of one (view, tile), plus the lists themselves; row_at(q) =
// the row of list position q.  The record gathers of U positions are requested before the first one is used.
template <int THREADS, typename RowAt>
__device__ __forceinline__ void write_tile(RowAt row_at, uint32_t n, uint32_t start, uint32_t gt, int gx, int tiles,
                                           const float4* __restrict__ rec, const BinView& b)
{
    constexpr int U = 8;
    const uint32_t tl = gt % (uint32_t)tiles;
    const uint32_t ty = tl / (uint32_t)gx, tx = tl - ty * (uint32_t)gx;
    const float x0 = (float)(tx * TILE), y0 = (float)(ty * TILE);
    for (uint32_t q0 = threadIdx.x; q0 < n; q0 += THREADS * U) {
        uint32_t g[U];
        float4 a0[U], a1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t q = q0 + u * THREADS;
            g[u] = row_at(q < n ? q : q0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a0[u] = rec[2 * (size_t)g[u]];
            a1[u] = rec[2 * (size_t)g[u] + 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t q = q0 + u * THREADS;
            if (q < n) {
                const size_t j = (size_t)start + q;
                b.irec[2 * j] = a0[u];
                b.irec[2 * j + 1] = payload_conic(a1[u]);
                b.ipack[j] = g[u] | (quadrant_reach_mask(a0[u], a1[u], x0, y0) << 24);
                b.point_list[j] = g[u];
                b.tile_list[j] = gt;
            }
        }
    }
}

// loads the list (blocked), sorts it in registers and leaves the rows in LDS at pad32(list position)
// Normalised bitonic network on n keys in GLOBAL memory (lists beyond the 16 384 keys the long-list launch holds in
// registers): merge size k = 2, 4, ...; the first step of a merge pairs i with its MIRROR inside the k-block, the following
// steps pair i with i + j, j = k/4 ... 1.  Every compare-exchange is ascending (minimum to the lower index), so elements
// beyond n behave as +infinity without being stored: a pair whose upper index is >= n is skipped.  One block, a workgroup
// barrier (and with it an L2 round trip) per step: slow, correct, rare.
template <int THREADS>
__device__ void bitonic_sort_global(uint64_t* s, uint32_t n)
{
    uint32_t N = 2;
    while (N < n) N <<= 1;
    const uint32_t half = N >> 1;
    auto step = [&](bool mirror, uint32_t kj) {
        for (uint32_t p = threadIdx.x; p < half; p += THREADS) {
            uint32_t lo, hi;
            if (mirror) {
                const uint32_t h = kj >> 1, blk = p / h, off = p & (h - 1);
                lo = blk * kj + off;
                hi = blk * kj + (kj - 1 - off);
            } else {
                lo = ((p & ~(kj - 1)) << 1) | (p & (kj - 1));
                hi = lo | kj;
            }
            if (hi < n) {
                const uint64_t a = s[lo], b = s[hi];
                if (a > b) { s[lo] = b; s[hi] = a; }
            }
        }
        __syncthreads();
    };
    __syncthreads();
    for (uint32_t k = 2; k <= N; k <<= 1) {
        step(true, k);
        for (uint32_t j = k >> 2; j > 0; j >>= 1) step(false, j);
    }
}

__device__ __forceinline__ void tile_span(uint32_t gt, int gtiles, int nchunk, const uint32_t* __restrict__ table,
                                          const uint32_t* __restrict__ total, uint32_t& start, uint32_t& n)
{
    start = table[(size_t)gt * nchunk];
    const uint32_t end = (gt + 1 < (uint32_t)gtiles) ? table[(size_t)(gt + 1) * nchunk] : total[0];
    n = end - start;
}

// ---- the first sort launch: a register-blocked bitonic network on 64-bit keys compared as DOUBLES -----------------------
// A key is (bits of a positive float depth) << 32 | row.  Read as an IEEE double that is a positive NORMAL number (the
// double's exponent field is the float's sign and top exponent bits: never 0, never 0x7ff), and positive doubles order like
// their bit patterns — so v_min_f64 / v_max_f64 are a 64-bit compare-exchange in TWO instructions (the integer form: one
// compare + four selects, and a scalar xor where the direction varies).  Pads are +infinity.  The network is the NORMALISED
// bitonic network (every compare-exchange puts the minimum at the lower index: no directions): a merge of size k = 2^m starts
// with a MIRROR step (i <-> i ^ (k - 1)) followed by half-cleaners of distance k/4 ... 1.  The keys live in LDS; a pass loads 16
// of them per thread, chosen so that the next (up to) four steps stay inside the thread's registers, and stores them back:
//   A   the first pass: 16 consecutive keys, merges 2 ... 16 complete (10 steps);
//   B   start of a merge k >= 32: 8 keys of the lower half of a k-block (stride k/16) and their 8 mirror images: the mirror
//       step and the half-cleaners k/4, k/8, k/16;
//   C   16 keys of stride 2^p: half-cleaners 2^(p+3) ... 2^p (the last pass of a merge: the low R <= 4 bits of 16 consecutive keys).
// 15 passes for 1024 keys: 880 min / max instructions and 480 LDS accesses per wave, where the DPP form issued 5 800 VALU
// instructions.  LDS index of key i: i + (i >> 4) — one pad per 16 makes every pass's 8-byte accesses conflict-free.
__device__ __forceinline__ uint32_t pidx(uint32_t i) { return i + (i >> 4); }
__device__ __forceinline__ void ce(double& a, double& b)
{
    double lo, hi;   // (inline asm: fmin / fmax would add canonicalising v_max_f64 x, x for signalling NaNs that cannot occur)
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    a = lo;
    b = hi;
}
// half-cleaners on register bits R-1 ... 0 of the 16 registers
template <int R>
__device__ __forceinline__ void half_cleaners(double (&x)[16])
{
#pragma unroll
    for (int bit = R - 1; bit >= 0; --bit) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (!(c & (1 << bit))) ce(x[c], x[c | (1 << bit)]);
    }
}
__device__ __forceinline__ void sort16(double (&x)[16])
{
#pragma unroll
    for (int k = 2; k <= 16; k <<= 1) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if ((c ^ (k - 1)) > c) ce(x[c], x[c ^ (k - 1)]);   // mirror inside the k-block
#pragma unroll
        for (int j = k >> 2; j > 0; j >>= 1) {
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (!(c & j)) ce(x[c], x[c | j]);
        }
    }
}
template <bool MULTI_WAVE>
__device__ __forceinline__ void pass_sync()
{
    if (MULTI_WAVE) __syncthreads(); else wave_order();
}
#define SR_LOAD16(ADDR)  _Pragma("unroll") for (int c = 0; c < 16; ++c) x[c] = s[pidx(ADDR)]
#define SR_STORE16(ADDR) _Pragma("unroll") for (int c = 0; c < 16; ++c) s[pidx(ADDR)] = x[c]

// sorts the N = 2^M keys (16 <= N, N / 16 <= blockDim.x working threads) at s[pidx(0 ... N-1)]
template <bool MULTI_WAVE>
__device__ void sort_lds_f64(double* s, uint32_t M)
{
    const uint32_t t = threadIdx.x;
    const bool work = t < (1u << (M - 4));
    double x[16];
    if (work) {
        double* row = s + 17u * t;   // pidx(16 t + c) = 17 t + c
#pragma unroll
        for (int c = 0; c < 16; ++c) x[c] = row[c];
        sort16(x);
#pragma unroll
        for (int c = 0; c < 16; ++c) row[c] = x[c];
    }
#pragma unroll 1
    for (uint32_t m = 5; m <= M; ++m) {
        pass_sync<MULTI_WAVE>();
        uint32_t g = m - 4;   // free low bits of the merge's first pass; then: low bits still to be cleaned
        if (work) {
            const uint32_t low = t & ((1u << g) - 1u), blk = t >> g;
            const uint32_t l0 = (blk << m) + low, u0 = (blk << m) + (1u << (m - 1)) + ((1u << g) - 1u - low);
            // registers 0..7: L_b = l0 + (b << g); registers 8..15: U'_b = u0 + (b << g); mirror pairs L_b <-> U'_(7-b)
            if (g >= 4) {   // (b << g) has no bits below 16: the pad term is linear in b
                const uint32_t pl = pidx(l0), pu = pidx(u0), st = (1u << g) + (1u << (g - 4));
#pragma unroll
                for (int c = 0; c < 16; ++c) x[c] = s[(c < 8 ? pl : pu) + (uint32_t)(c & 7) * st];
#pragma unroll
                for (int b8 = 0; b8 < 8; ++b8) ce(x[b8], x[15 - b8]);
                half_cleaners<3>(x);
#pragma unroll
                for (int c = 0; c < 16; ++c) s[(c < 8 ? pl : pu) + (uint32_t)(c & 7) * st] = x[c];
            } else {
                SR_LOAD16((c < 8 ? l0 : u0) + ((uint32_t)(c & 7) << g));
#pragma unroll
                for (int b8 = 0; b8 < 8; ++b8) ce(x[b8], x[15 - b8]);
                half_cleaners<3>(x);
                SR_STORE16((c < 8 ? l0 : u0) + ((uint32_t)(c & 7) << g));
            }
        }
        while (g > 0) {
            pass_sync<MULTI_WAVE>();
            if (g >= 4) {
                const uint32_t p = g - 4;
                if (work) {
                    const uint32_t base = ((t >> p) << (p + 4)) | (t & ((1u << p) - 1u));
                    if (p >= 4) {
                        const uint32_t pb = pidx(base), st = (1u << p) + (1u << (p - 4));
#pragma unroll
                        for (int c = 0; c < 16; ++c) x[c] = s[pb + (uint32_t)c * st];
                        half_cleaners<4>(x);
#pragma unroll
                        for (int c = 0; c < 16; ++c) s[pb + (uint32_t)c * st] = x[c];
                    } else {
                        SR_LOAD16(base + ((uint32_t)c << p));
                        half_cleaners<4>(x);
                        SR_STORE16(base + ((uint32_t)c << p));
                    }
                }
                g -= 4;
            } else {
                if (work) {
                    double* row = s + 17u * t;
#pragma unroll
                    for (int c = 0; c < 16; ++c) x[c] = row[c];
                    if (g == 3) half_cleaners<3>(x);
                    else if (g == 2) half_cleaners<2>(x);
                    else half_cleaners<1>(x);
#pragma unroll
                    for (int c = 0; c < 16; ++c) row[c] = x[c];
                }
                g = 0;
            }
        }
    }
    pass_sync<MULTI_WAVE>();
}

// host-mapped word (one per device): raised by the tile kernel when it meets a list beyond BIN_SORT_TILE_NARROW keys
__device__ uint32_t* g_bin_wide_sink = nullptr;

// First sort launch: one block of CAP / 16 threads per global tile, of which ceil(n / 1024) waves work on the list (the
// others leave at once): a list of up to 1024 keys is one wave's alone.  Lists longer than CAP are the second launch's
// (bin_sort_big_kernel finds them itself from the scanned table: the two launches share no state and may run side by side).
template <int CAP>
__global__ void __launch_bounds__(CAP / 16)
bin_sort_tile_kernel(int gtiles, int tiles, int gx, int nchunk, const uint32_t* __restrict__ table /*exclusive prefix*/,
                     const uint32_t* __restrict__ total /*[0]: R*/, const uint64_t* __restrict__ keys,
                     const float4* __restrict__ rec, BinView b)
{
    __shared__ double s_keys[CAP + CAP / 16];
    const uint32_t gt = blockIdx.x, t = threadIdx.x;
    uint32_t start, n;
    SR_STAMP(0);
#ifdef SR_BIN_TIMING
    if (threadIdx.x == 0 && g_bin_dbg) {
        g_bin_dbg[8 * blockIdx.x + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
        g_bin_dbg[8 * blockIdx.x + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
    }
#endif
    tile_span(gt, gtiles, nchunk, table, total, start, n);
    if (t == 0) {   // empty tiles keep [0, 0), like the radix front end's zeroed table
        b.ranges[2 * gt] = n ? start : 0u;
        b.ranges[2 * gt + 1] = n ? start + n : 0u;
        if (n > (uint32_t)BIN_SORT_TILE_NARROW) {   // tell the host that the wide instantiation pays on this scene
            uint32_t* sink = g_bin_wide_sink;
            if (sink) __hip_atomic_store(sink, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (n == 0 || n > (uint32_t)CAP) return;
    uint32_t M = 4;
    while ((1u << M) < n) ++M;
    const uint32_t N = 1u << M, T = N <= 1024u ? 64u : N / 16u;   // working threads: one per 16 keys, at least a wave
    const bool two = T > 64u;
    if (t >= T) return;
    const uint64_t* src = keys + start;
    uint64_t* s_bits = reinterpret_cast<uint64_t*>(s_keys);
    for (uint32_t i0 = t; i0 < N; i0 += 8 * T) {   // (N is a multiple of 16; eight loads in flight per lane)
        uint64_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t i = i0 + u * T;
            v[u] = src[i < n ? i : n - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t i = i0 + u * T;
            if (i < N) s_bits[pidx(i)] = i < n ? v[u] : 0x7FF0000000000000ull;   // pad: +infinity
        }
    }
    SR_STAMP(1);
    if (two) { __syncthreads(); sort_lds_f64<true>(s_keys, M); } else { wave_order(); sort_lds_f64<false>(s_keys, M); }
    SR_STAMP(3);
    auto row_at = [&](uint32_t q) { return (uint32_t)s_bits[pidx(q)] & 0xFFFFFFu; };
    if (T == 64u) write_tile<64>(row_at, n, start, gt, gx, tiles, rec, b);
    else if (T == 128u || CAP <= 2048) write_tile<128>(row_at, n, start, gt, gx, tiles, rec, b);
    else write_tile<256>(row_at, n, start, gt, gx, tiles, rec, b);
#ifdef SR_BIN_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SR_STAMP(5);
}

// Second sort launch: list i is block i % gridDim.x's if it is longer than the tile launch's cap; up to BIN_SORT_BIG keys with the SAME LDS network as the tile
// kernel (139 KB of the CU's 160 KB: one block per CU), the global network beyond.  (Round 5 first held these lists in the
// registers of the block — 8 or 16 keys per thread, 64-bit integer compares, every distance beyond a wave exchanged through LDS
// four registers at a time: ~70 us per list, 87 us of a refinement iteration at Replica scale where 15 - 40 lists of a frame
// exceed 4 096 keys; profiles/r05_ab_probes.txt #11.)
__global__ void __launch_bounds__(1024)
bin_sort_big_kernel(int gtiles, int tiles, int gx, int nchunk, int cap /*the tile launch's CAP: longer lists are this launch's*/,
                    const uint32_t* __restrict__ table, const uint32_t* __restrict__ total, uint64_t* __restrict__ keys,
                    const float4* __restrict__ rec, BinView b,
                    uint32_t* __restrict__ order /*launch order of the compositing grids (tile_order.h), or null*/,
                    int ranges_final /*the tile launch has finished (same stream): its range table may be read*/)
{
    __shared__ double s_keys[BIN_SORT_BIG + BIN_SORT_BIG / 16];
    __shared__ uint32_t s_mine[BIN_BIG_MINE];   // this block's share of the long lists
    __shared__ uint32_t s_nmine;
    uint64_t* s_bits = reinterpret_cast<uint64_t*>(s_keys);
    const uint32_t t = threadIdx.x;
    static_assert(ORDER_THREADS == 1024, "the last block of this launch computes the launch order");
    if (order && blockIdx.x == gridDim.x - 1) {
        __shared__ uint32_t s_cnt[ORDER_BUCKETS];
        __shared__ uint32_t s_wsum[ORDER_THREADS / WAVE];
        if (ranges_final)   // (contiguous 8-byte entries)
            tile_order_block(gtiles, [&](int i) { return b.ranges[2 * i + 1] - b.ranges[2 * i]; }, order, s_cnt, s_wsum, b.nparts);
        else                // beside the tile launch: from the scanned table, like the lists below — nothing that launch writes is read here
            tile_order_block(gtiles, [&](int i) { uint32_t st_, n_; tile_span((uint32_t)i, gtiles, nchunk, table, total, st_, n_); return n_; },
                             order, s_cnt, s_wsum, b.nparts);
        __syncthreads();
    }
    // The long lists: list i is block i % gridDim.x's (neighbouring tiles — a near wall's — go to different blocks); every block
    // looks at its own gtiles / gridDim.x lengths in the scanned table.  (Round 5: the tile launch appended the long lists to a
    // list in global memory with an atomic counter — this launch then had to wait for that one, 40 us of a Replica-scale frame,
    // and the counter had to be reset per render.)
    if (t == 0) s_nmine = 0u;
    __syncthreads();
    for (uint32_t i = blockIdx.x + t * gridDim.x; i < (uint32_t)gtiles; i += gridDim.x * 1024u) {
        uint32_t st_, n_;
        tile_span(i, gtiles, nchunk, table, total, st_, n_);
        if (n_ > (uint32_t)cap) {
            const uint32_t slot = atomicAdd(&s_nmine, 1u);
            if (slot < (uint32_t)BIN_BIG_MINE) s_mine[slot] = i;
        }
    }
    __syncthreads();
    const uint32_t nwork = min(s_nmine, (uint32_t)BIN_BIG_MINE);
    for (uint32_t wi = 0; wi < nwork; ++wi) {
        const uint32_t gt = s_mine[wi];
        uint32_t start, n;
        tile_span(gt, gtiles, nchunk, table, total, start, n);
        if (n <= (uint32_t)BIN_SORT_BIG) {
            uint32_t M = 4;
            while ((1u << M) < n) ++M;
            const uint32_t N = 1u << M;
            const uint64_t* src = keys + start;
            for (uint32_t i0 = t; i0 < N; i0 += 8 * 1024) {   // eight loads in flight per lane
                uint64_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t i = i0 + u * 1024;
                    v[u] = src[i < n ? i : n - 1];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t i = i0 + u * 1024;
                    if (i < N) s_bits[pidx(i)] = i < n ? v[u] : 0x7FF0000000000000ull;   // pad: +infinity
                }
            }
            __syncthreads();
            sort_lds_f64<true>(s_keys, M);
            write_tile<1024>([&](uint32_t q) { return (uint32_t)s_bits[pidx(q)] & 0xFFFFFFu; }, n, start, gt, gx, tiles, rec, b);
        } else {
            uint64_t* list = keys + start;
            bitonic_sort_global<1024>(list, n);
            write_tile<1024>([&](uint32_t q) { return (uint32_t)list[q] & 0xFFFFFFu; }, n, start, gt, gx, tiles, rec, b);
        }
        __syncthreads();   // the next work item reuses the LDS
    }
}

// ---- host side -------------------------------------------------------------------------------
static int g_bin_mode = -1;   // -1 auto, 0 never, 1 whenever the shape allows
void set_bin_mode(int mode) { g_bin_mode = mode < 0 ? -1 : (mode > 1 ? 1 : mode); }

size_t bin_table_entries(int32_t P, int32_t V, int tiles)
{
    return (size_t)(V > 0 ? V : 1) * (size_t)tiles * (size_t)bin_chunks(P, V, tiles);
}

size_t bin_scratch_bytes(int32_t P, int32_t V, int tiles)
{
    const size_t entries = bin_table_entries(P, V, tiles);
    return align_up(entries * sizeof(uint32_t), 256) + scan_tmp_bytes((int64_t)entries);
}

bool use_bins(int32_t P, int32_t V, int gx, int gy, size_t scratch_bytes)
{
    const int tiles = gx * gy;
    if (g_bin_mode == 0 || P <= 0 || tiles <= 0 || tiles > BIN_MAX_TILES || gx > 255 || gy > 255) return false;
    if (bin_table_entries(P, V, tiles) >= ((size_t)1 << 31)) return false;
    if (bin_scratch_bytes(P, V, tiles) > scratch_bytes) return false;
    if (g_bin_mode == 1) return true;
    return (int64_t)V * tiles <= BIN_AUTO_MAX_TILES;
}

// geometry stage: per-(tile, chunk) counts and their exclusive scan (the state words of the scan at scan_tmp must be zero:
// preprocess_kernel clears them)
int launch_bin_count(const splatraster_settings& s, int32_t P, int32_t V, const GeomView& g, uint32_t* table, void* scan_tmp,
                     hipStream_t stream)
{
    const int gx = (s.image_width + TILE - 1) / TILE, gy = (s.image_height + TILE - 1) / TILE;
    const int tiles = gx * gy, nchunk = bin_chunks(P, V, tiles);
#define SR_BIN_WALK(ROWS, SC, ...) hipLaunchKernelGGL((bin_walk_kernel<ROWS, SC>), dim3((unsigned)nchunk, (unsigned)V), dim3(BIN_THREADS), \
                                                      (size_t)tiles * sizeof(uint32_t), stream, P, tiles, gx, gy, nchunk, g.rec, __VA_ARGS__)
    switch (bin_rows_per_thread(P, V, tiles)) {
        case 1: SR_BIN_WALK(1, false, table, (uint64_t*)nullptr); break;
        case 2: SR_BIN_WALK(2, false, table, (uint64_t*)nullptr); break;
        default: SR_BIN_WALK(8, false, table, (uint64_t*)nullptr); break;
    }
    SR_LAUNCH_CHECK();
    return exclusive_scan_u32((int64_t)bin_table_entries(P, V, tiles), table, g.total, scan_tmp, stream, true);
}

// which instantiation of the tile kernel: 0 = follow the hint (default), else BIN_SORT_TILE_NARROW / BIN_SORT_TILE_WIDE forced
static int g_tile_cap = 0;
void set_bin_tile_cap(int cap) { g_tile_cap = (cap == BIN_SORT_TILE_NARROW || cap == BIN_SORT_TILE_WIDE) ? cap : 0; }
static uint32_t* g_wide_host[64] = {};     // pinned, host-mapped hint word per device
static int g_wide_left[64] = {};           // frames the wide instantiation stays selected after the last raised hint
static std::mutex g_wide_mu;

static int wide_hint_init(int dev)
{
    std::lock_guard<std::mutex> lk(g_wide_mu);
    if (g_wide_host[dev]) return SPLATRASTER_OK;
    uint32_t* h = nullptr;
    SR_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h), 64, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent));
    *h = 0u;
    uint32_t* d = nullptr;
    SR_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d), h, 0));
    SR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_bin_wide_sink), &d, sizeof(d)));
    g_wide_host[dev] = h;
    return SPLATRASTER_OK;
}

// 0 (default): both sort launches on the caller's stream; -1: the long-list launch on the side stream when the scene has long lists; 1: always.
// Measured a wash (profiles/r06_ab_probes.txt #6: 4 x 9 interleaved regions at Replica scale: 599.5 us forked, 596.5 serial): the fork / join
// costs what the overlap of 40 us of one-block sorts with a 58-us launch of 1 200 blocks gains.  The hook stays for measurements.
static int g_bin_fork = 0;
void set_bin_fork(int mode) { g_bin_fork = mode < 0 ? -1 : (mode > 1 ? 1 : mode); }

// one side stream + fork / join events per host thread (events are re-recorded per call: another thread's call must never sit
// between this thread's record and wait)
struct SideStream { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; int device = -1; };
static thread_local SideStream t_side;
static int side_stream(int dev, SideStream** out)
{
    SideStream& s = t_side;
    if (!s.stream || s.device != dev) {
        if (s.stream) { (void)hipStreamDestroy(s.stream); (void)hipEventDestroy(s.fork); (void)hipEventDestroy(s.join); }
        s = SideStream{};
        SR_HIP_CHECK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        SR_HIP_CHECK(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming));
        SR_HIP_CHECK(hipEventCreateWithFlags(&s.join, hipEventDisableTiming));
        s.device = dev;
    }
    *out = &s;
    return SPLATRASTER_OK;
}

// render stage: scatter the keys, sort every tile's list, write the payload
int launch_bin_scatter_sort(const splatraster_settings& s, int32_t P, int32_t V, int64_t R, const GeomView& g,
                            const uint32_t* table, const BinView& b, uint64_t* keys, hipStream_t stream)
{
    const int gx = (s.image_width + TILE - 1) / TILE, gy = (s.image_height + TILE - 1) / TILE;
    const int tiles = gx * gy, nchunk = bin_chunks(P, V, tiles);
    const int gtiles = V * tiles;
    (void)R;
    switch (bin_rows_per_thread(P, V, tiles)) {
        case 1: SR_BIN_WALK(1, true, const_cast<uint32_t*>(table), keys); break;
        case 2: SR_BIN_WALK(2, true, const_cast<uint32_t*>(table), keys); break;
        default: SR_BIN_WALK(8, true, const_cast<uint32_t*>(table), keys); break;
    }
#undef SR_BIN_WALK
    SR_LAUNCH_CHECK();
    int dev = 0;
    SR_HIP_CHECK(hipGetDevice(&dev));
    bool wide = g_tile_cap == BIN_SORT_TILE_WIDE;
    bool long_lists = false;   // a list beyond the narrow instantiation was seen in the last BIN_WIDE_FRAMES frames
    if (dev >= 0 && dev < 64) {
        int st = wide_hint_init(dev);
        if (st) return st;
        volatile uint32_t* h = g_wide_host[dev];
        if (*h) { *h = 0u; g_wide_left[dev] = BIN_WIDE_FRAMES; }      // (benign race between host threads: a hint)
        long_lists = g_wide_left[dev] > 0;
        if (g_tile_cap == 0) wide = long_lists;
        if (g_wide_left[dev] > 0) --g_wide_left[dev];
    }
    // The two sort launches are independent (the long-list launch finds its lists in the scanned table itself), so on a scene
    // that HAS long lists they run side by side: the long-list launch on this thread's side stream, forked behind the scatter
    // and joined before the compositing grids (round 5: 40 us of a Replica-scale frame were one 8 192-key sort AFTER the tile
    // launch had finished).  Without long lists the second launch only computes the launch order: same stream, no events.
    const int tile_cap = wide ? BIN_SORT_TILE_WIDE : BIN_SORT_TILE_NARROW;
    const bool fork = g_bin_fork == 1 || (g_bin_fork < 0 && long_lists);
    hipStream_t side = stream;
    SideStream* ss = nullptr;
    if (fork) {
        int st = side_stream(dev, &ss);
        if (st) return st;
        side = ss->stream;
        SR_HIP_CHECK(hipEventRecord(ss->fork, stream));
        SR_HIP_CHECK(hipStreamWaitEvent(side, ss->fork, 0));
        // the long lists first: they are the launch sequence's longest blocks
        hipLaunchKernelGGL(bin_sort_big_kernel, dim3(BIN_BIG_BLOCKS), dim3(1024), 0, side, gtiles, tiles, gx, nchunk,
                           tile_cap, table, g.total, keys, g.rec, b,
                           use_tile_order(V, tiles) ? b.tile_order : (uint32_t*)nullptr, 0);
        SR_LAUNCH_CHECK();
        SR_HIP_CHECK(hipEventRecord(ss->join, side));
    }
    if (wide)
        hipLaunchKernelGGL(bin_sort_tile_kernel<BIN_SORT_TILE_WIDE>, dim3((unsigned)gtiles), dim3(BIN_SORT_TILE_WIDE / 16), 0, stream,
                           gtiles, tiles, gx, nchunk, table, g.total, keys, g.rec, b);
    else
        hipLaunchKernelGGL(bin_sort_tile_kernel<BIN_SORT_TILE_NARROW>, dim3((unsigned)gtiles), dim3(BIN_SORT_TILE_NARROW / 16), 0,
                           stream, gtiles, tiles, gx, nchunk, table, g.total, keys, g.rec, b);
    SR_LAUNCH_CHECK();
    if (fork) {
        SR_HIP_CHECK(hipStreamWaitEvent(stream, ss->join, 0));
    } else {
        hipLaunchKernelGGL(bin_sort_big_kernel, dim3(BIN_BIG_BLOCKS), dim3(1024), 0, stream, gtiles, tiles, gx, nchunk,
                           tile_cap, table, g.total, keys, g.rec, b,
                           use_tile_order(V, tiles) ? b.tile_order : (uint32_t*)nullptr, 1);
        SR_LAUNCH_CHECK();
    }
    return SPLATRASTER_OK;
}

}  // namespace sr
