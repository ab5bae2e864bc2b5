// Micro-benchmark of the per-tile sort kernels of csrc/binsort.hip on synthetic lists (no Python, no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DSR_BIN_TIMING -I splatloc_amd/csrc \
//         tools/micro/tilesort_bench.hip -o gpurun_out/tilesort_bench && gpurun_out/tilesort_bench [tiles] [mean] [sigma]
// Prints the kernel time (HIP events) and, with -DSR_BIN_TIMING, the mean / max s_memtime span of every phase of a block.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#ifndef SR_BENCH_CAP
#define SR_BENCH_CAP 2048   // -DSR_BENCH_CAP=4096: the wide instantiation
#endif
#include "../../splatloc_amd/csrc/binsort.hip"

namespace sr {   // the pieces of the library this translation unit does not carry
void set_hip_error(hipError_t, const char*) {}
size_t scan_tmp_bytes(int64_t) { return 0; }
int exclusive_scan_u32(int64_t, uint32_t*, uint32_t*, void*, hipStream_t, bool) { return 0; }
}  // namespace sr

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const int tiles = argc > 1 ? atoi(argv[1]) : 1200;
    const double mean = argc > 2 ? atof(argv[2]) : 1005.0, sigma = argc > 3 ? atof(argv[3]) : 40.0;
    const int P = 500000, gx = 40;
    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd(mean, sigma);
    std::vector<uint32_t> table(tiles + 1);
    uint32_t R = 0;
    for (int t = 0; t < tiles; ++t) { table[t] = R; R += (uint32_t)std::max(0.0, nd(rng)); }
    table[tiles] = R;
    std::vector<uint64_t> keys(R);
    std::uniform_real_distribution<float> depth(0.2f, 50.0f);   // keys as the pipeline makes them: bits of a positive depth | row
    for (auto& k : keys) {
        float z = depth(rng);
        if ((rng() & 15) == 0) z = 1.25f;   // exact depth ties: the row decides
        uint32_t zb;
        memcpy(&zb, &z, 4);
        k = ((uint64_t)zb << 32) | (rng() % P);
    }
    std::vector<float> rec(8 * (size_t)P);
    for (auto& v : rec) v = (float)(rng() % 1000) / 100.0f + 0.1f;
    uint32_t *d_table, *d_total, *d_big_list;
    uint64_t* d_keys;
    float4* d_rec;
    CK(hipMalloc(&d_table, 4 * (tiles + 1)));
    CK(hipMalloc(&d_total, 16));
    CK(hipMalloc(&d_big_list, 4 * tiles));
    CK(hipMalloc(&d_keys, 8 * (size_t)R));
    CK(hipMalloc(&d_rec, 32 * (size_t)P));
    CK(hipMemcpy(d_table, table.data(), 4 * (tiles + 1), hipMemcpyHostToDevice));
    uint32_t tot[4] = {R, 0, 0, 0};
    CK(hipMemcpy(d_total, tot, 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rec, rec.data(), 32 * (size_t)P, hipMemcpyHostToDevice));
    sr::BinView b{};
    CK(hipMalloc(&b.point_list, 4 * (size_t)R));
    CK(hipMalloc(&b.tile_list, 4 * (size_t)R));
    CK(hipMalloc(&b.ranges, 8 * tiles));
    CK(hipMalloc(&b.irec, 32 * (size_t)R));
    CK(hipMalloc(&b.ipack, 4 * (size_t)R));
#ifdef SR_BIN_TIMING
    unsigned long long* d_dbg;
    CK(hipMalloc(&d_dbg, 8 * 8 * (size_t)tiles));
    CK(hipMemset(d_dbg, 0, 8 * 8 * (size_t)tiles));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(sr::g_bin_dbg), &d_dbg, sizeof(d_dbg)));
#endif
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0.f;
    const int reps = 20;
    for (int it = 0; it < reps + 3; ++it) {
        CK(hipMemcpy(d_keys, keys.data(), 8 * (size_t)R, hipMemcpyHostToDevice));
        CK(hipMemset(d_total + 2, 0, 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(sr::bin_sort_tile_kernel<SR_BENCH_CAP>, dim3(tiles), dim3(SR_BENCH_CAP / 16), 0, 0, tiles, tiles, gx, 1, d_table, d_total, d_keys,
                           d_rec, b, d_total + 2, d_big_list);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 3) { best = std::min(best, ms); sum += ms; }
    }
    printf("tiles %d  R %u  mean list %.0f  tile kernel: best %.1f us  mean %.1f us\n", tiles, R, (double)R / tiles, best * 1e3, sum / reps * 1e3);
    // check: sorted order
    std::vector<uint32_t> pl(R);
    CK(hipMemcpy(pl.data(), b.point_list, 4 * (size_t)R, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int t = 0; t < tiles; ++t) {
        std::vector<uint64_t> ref(keys.begin() + table[t], keys.begin() + table[t + 1]);
        std::sort(ref.begin(), ref.end());
        for (size_t q = 0; q < ref.size(); ++q) bad += ((uint32_t)(ref[q] & 0xFFFFFF) != pl[table[t] + q]);
    }
    printf("mismatches vs std::sort: %zu\n", bad);
#ifdef SR_BIN_TIMING
    std::vector<unsigned long long> dbg(8 * (size_t)tiles);
    CK(hipMemcpy(dbg.data(), d_dbg, 8 * 8 * (size_t)tiles, hipMemcpyDeviceToHost));
    const char* names[] = {"span + keys to LDS", "-", "sort", "-", "payload", "", ""};
    for (int t = 0; t < tiles; ++t) { dbg[8 * t + 2] = dbg[8 * t + 1]; dbg[8 * t + 4] = dbg[8 * t + 3]; }
    unsigned long long t0min = ~0ull, t5max = 0;
    for (int t = 0; t < tiles; ++t) { t0min = std::min(t0min, dbg[8 * t]); t5max = std::max(t5max, dbg[8 * t + 5]); }
    for (int ph = 0; ph < 5; ++ph) {
        double s = 0, mx = 0;
        for (int t = 0; t < tiles; ++t) { const double d = (double)(dbg[8 * t + ph + 1] - dbg[8 * t + ph]); s += d; mx = std::max(mx, d); }
        printf("  phase %-26s mean %8.0f  max %8.0f ticks\n", names[ph], s / tiles, mx);
    }
    {   // where did the wave 0 of every block run?  HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13; XCC_ID 3:0
        int simd_hist[4] = {0, 0, 0, 0};
        std::vector<int> per_simd(8 * 8 * 2 * 16 * 4, 0);
        for (int t = 0; t < tiles; ++t) {
            const unsigned hw = (unsigned)dbg[8 * t + 6], xcc = (unsigned)dbg[8 * t + 7] & 15u;
            const unsigned simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
            simd_hist[simd]++;
            per_simd[(((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd]++;
        }
        int used = 0, mx = 0;
        for (int v : per_simd) { used += v > 0; mx = std::max(mx, v); }
        printf("  wave 0 of the blocks by SIMD id: %d %d %d %d; distinct (xcc, se, sh, cu, simd) slots used %d, most blocks on one SIMD %d\n",
               simd_hist[0], simd_hist[1], simd_hist[2], simd_hist[3], used, mx);
    }
    double st = 0;
    for (int t = 0; t < tiles; ++t) st += (double)(dbg[8 * t] - t0min);
    printf("  block start after kernel start: mean %.0f ticks; kernel span %llu ticks (s_memtime: 100 MHz)\n", st / tiles, t5max - t0min);
#endif
    return bad != 0;
}
