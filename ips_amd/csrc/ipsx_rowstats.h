// ipsx_rowstats.h - the LayerNorm row moments (mean, rstd) of one feature row by ONE wavefront, in the contract's order:
// 64 strided partial sums in ascending order + xor butterfly, centred second moment (oracle/ips_oracle.cpp orc_projector).
// Shared by row_stats_kernel (aggregate.hip) and the logits + statistics launch (scorer.hip).
#pragma once

#include "ipsx_math.h"

namespace ipsx {

// a row of up to 64 * RS_MAX floats stays in registers between the two passes (every element is read ONCE, 256 B per
// wave-instruction), longer rows are re-read
constexpr int RS_MAX = 32;

__device__ __forceinline__ float2 row_stats_wave(const float* __restrict__ xr, int d, float eps, int lane) {
    float s = 0.0f, q = 0.0f, mean;
    if (d <= 64 * RS_MAX) {
        float v[RS_MAX];
#pragma unroll
        for (int k = 0; k < RS_MAX; ++k) v[k] = (lane + 64 * k < d) ? xr[lane + 64 * k] : 0.0f;
#pragma unroll
        for (int k = 0; k < RS_MAX; ++k) if (lane + 64 * k < d) s = s + v[k];
        mean = wave_butterfly_sum(s) / (float)d;
#pragma unroll
        for (int k = 0; k < RS_MAX; ++k)
            if (lane + 64 * k < d) { const float c = v[k] - mean; const float c2 = c * c; q = q + c2; }
    } else {
        for (int i = lane; i < d; i += 64) s = s + xr[i];
        mean = wave_butterfly_sum(s) / (float)d;
        for (int i = lane; i < d; i += 64) { const float c = xr[i] - mean; const float c2 = c * c; q = q + c2; }
    }
    const float var = wave_butterfly_sum(q) / (float)d;
    const float rstd = 1.0f / __builtin_sqrtf(var + eps);
    return make_float2(mean, rstd);
}

// NR rows at once by one wavefront (d <= 64 * RS_MAX): the loads of all NR rows are in flight before the first add, and the
// NR reductions run side by side - per row exactly the operations of row_stats_wave, in its order.  For a caller that has
// nothing else in flight to hide a row's 2 us of load latency behind (projector_stream_kernel: 16 rows per wavefront
// and tile, one after the other 32 us of a 300 us tile).  xr[q] == nullptr: no such row, st[q] is left alone.
template <int NR>
__device__ __forceinline__ void row_stats_wave_n(const float* const (&xr)[NR], int d, float eps, int lane, float2 (&st)[NR]) {
    float v[NR][RS_MAX];
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
        for (int k = 0; k < RS_MAX; ++k) v[q][k] = (xr[q] && lane + 64 * k < d) ? xr[q][lane + 64 * k] : 0.0f;
    float mean[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < RS_MAX; ++k) if (lane + 64 * k < d) s = s + v[q][k];
        mean[q] = wave_butterfly_sum(s) / (float)d;
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        float qq = 0.0f;
#pragma unroll
        for (int k = 0; k < RS_MAX; ++k)
            if (lane + 64 * k < d) { const float c = v[q][k] - mean[q]; const float c2 = c * c; qq = qq + c2; }
        const float var = wave_butterfly_sum(qq) / (float)d;
        const float rstd = 1.0f / __builtin_sqrtf(var + eps);
        if (xr[q]) st[q] = make_float2(mean[q], rstd);
    }
}

}  // namespace ipsx
