// tile_order.h — the launch order of the compositing grids as a block-level routine (binning.hip: a launch of its own behind the
// radix front end; binsort.hip: the last block of the big-list launch runs it, one launch less per frame of the binned front end).
#pragma once
#include "common.h"

namespace sr {

constexpr int ORDER_THREADS = 1024, ORDER_BUCKETS = 1024;

// One block of ORDER_THREADS threads buckets the T (view, tile) lists by length (16 entries per bucket, longest first) and writes
// the tile ids in that order.  s_cnt: ORDER_BUCKETS words, s_wsum: ORDER_THREADS / WAVE words of LDS.
// `len_at(i)`: length of list i (the range table of the radix front end; the scanned (tile, chunk) table of the binned one, whose
// ranges may still be in flight when this runs beside the tile kernel).
template <typename LenAt>
__device__ __forceinline__ void tile_order_block(int T, LenAt len_at, uint32_t* __restrict__ order, uint32_t* s_cnt, uint32_t* s_wsum,
                                                 uint32_t* __restrict__ nparts = nullptr /*[T]: parts of list i in a split launch (common.h)*/)
{
    const int t = threadIdx.x;
    s_cnt[t] = 0u;
    __syncthreads();
    for (int i = t; i < T; i += ORDER_THREADS) {
        const uint32_t len = len_at(i);
        atomicAdd(&s_cnt[ORDER_BUCKETS - 1 - min((uint32_t)(ORDER_BUCKETS - 1), len >> 4)], 1u);
    }
    __syncthreads();
    // exclusive scan of the 1024 bucket counts (one per thread)
    const uint32_t c = s_cnt[t];
    uint32_t incl = c;
    const int lane = t & (WAVE - 1), w = t / WAVE;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d, WAVE);
        if (lane >= d) incl += o;
    }
    if (lane == WAVE - 1) s_wsum[w] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int k = 0; k < w; ++k) before += s_wsum[k];
    __syncthreads();
    s_cnt[t] = before + incl - c;
    __syncthreads();
    for (int i = t; i < T; i += ORDER_THREADS) {
        const uint32_t len = len_at(i);
        const uint32_t pos = atomicAdd(&s_cnt[ORDER_BUCKETS - 1 - min((uint32_t)(ORDER_BUCKETS - 1), len >> 4)], 1u);
        order[pos] = (uint32_t)i;
        if (nparts) nparts[i] = split_count(len, pos);
    }
}

}  // namespace sr
