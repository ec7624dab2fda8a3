// ipsx_rowstats.h - LayerNorm row moments (mean, rstd).
//  * row_stats_wave: one feature row by ONE wavefront - 64 strided partial sums in ascending order + xor butterfly,
//    centred second moment (the transformer's LayerNorms, oracle layernorm_row);
//  * RowMoments / row_moments_wave32 (round 5): the PROJECTOR's moments, taken off the operand stream of its GEMM - lane
//    (row, half h) holds the four consecutive k = 8g + 4h + j of k-group g, so a row's sums run as eight chains (h, j)
//    over g ascending, folded ((c0 + c1) + (c2 + c3)) per half and half 0 + half 1; var = E[x^2] - mean^2, recentred
//    where that cancels (RM_RECENTRE below; oracle/ips_oracle.cpp projector_moments).  Shared by projector_stream_kernel / conv_nhwc_kernel<NORM>
//    (conv_nhwc.hip), row_moments_kernel (aggregate.hip) and the logits + statistics launch (scorer.hip).
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

// ---- the projector's moments (see the header comment): per lane the four j-chains of its half of the row
typedef float rm_f32x2 __attribute__((ext_vector_type(2)));
typedef float rm_f32x4 __attribute__((ext_vector_type(4)));

struct RowMoments {
    rm_f32x2 s01, s23, q01, q23;
};

__device__ __forceinline__ void rm_zero(RowMoments& m) {
    m.s01 = rm_f32x2{0.0f, 0.0f}; m.s23 = m.s01; m.q01 = m.s01; m.q23 = m.s01;
}

// one k-group's four values of this lane: two packed adds, two packed fmas (v_pk_add_f32 / v_pk_fma_f32 - the roundings
// of the scalar operations, half the instructions)
// (spelled out: the compiler splits the vector forms into four v_add_f32 + four v_fma_f32 - eight VALU instructions per
//  stage of the projector's GEMM instead of four, each of them matrix-pipe time)
__device__ __forceinline__ rm_f32x2 rm_pk_add(rm_f32x2 a, rm_f32x2 b) {
    rm_f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ rm_f32x2 rm_pk_fma(rm_f32x2 a, rm_f32x2 b, rm_f32x2 c) {
    rm_f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ void rm_add(RowMoments& m, rm_f32x4 v) {
    const rm_f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
    m.s01 = rm_pk_add(m.s01, lo);
    m.s23 = rm_pk_add(m.s23, hi);
    m.q01 = rm_pk_fma(lo, lo, m.q01);
    m.q23 = rm_pk_fma(hi, hi, m.q23);
}

// Round 6 (advisor, round 5).  E[x^2] - mean^2 loses (mean / std)^2 * 2^-24 of the variance, and the folded form
// acc - mean * colsum cancels the same way one step later.  A row whose one-pass variance is below 1 / RM_RECENTRE of its
// E[x^2] (mean^2 / var > 16: |mean| > 4 std) is therefore CENTRED, as nn.LayerNorm itself does (ips_net.py:56): its second moment is taken
// again over d = x - mean (rm_recentred: the same eight chains, the mean refined by E[d]), and its Linear runs
// on the centred row, acc = chain over (x[c] - mean') * w[o][c], t = acc (centred_row_dot below; oracle projector_moments /
// orc_projector).  The row's statistics say so in the SIGN of rstd: (mean, -rstd) = centred.  Rows of well-conditioned
// features - every fixture, every CAMELYON-like slide - never take these paths and keep their bits.
constexpr float RM_RECENTRE = 17.0f;

// the centred moments of the rows marked `centred` (their lanes read the row again; the others ride along on zeros):
// over d = x - mean the same eight chains sum d and d * d; mean' = mean + E[d] (the first mean carries the rounding of
// 2,048 additions of values that do not cancel - a constant row's residual d is exact, so mean' is the constant itself),
// var = E[d^2] - E[d]^2, which no longer cancels.  NOT inlined - a rare path that must not cost the GEMM kernels around it
// registers.  -> (mean', var)
__device__ __attribute__((noinline)) float2 rm_recentred(const float* __restrict__ xr, bool centred, float mean, float var,
                                                          int d, int lane) {
    const rm_f32x4* p = reinterpret_cast<const rm_f32x4*>(xr);
    const rm_f32x4 mv = {mean, mean, mean, mean};
    rm_f32x4 q = {0.0f, 0.0f, 0.0f, 0.0f}, sd = q;
    const int kgs = d >> 3;
    int g = 0;
    for (; g + 8 <= kgs; g += 8) {
        rm_f32x4 v[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) v[w] = centred ? p[2 * (g + w)] : mv;
#pragma unroll
        for (int w = 0; w < 8; ++w) { const rm_f32x4 c = v[w] - mv; sd = sd + c; q = __builtin_elementwise_fma(c, c, q); }
    }
    for (; g < kgs; ++g) { const rm_f32x4 c = (centred ? p[2 * g] : mv) - mv; sd = sd + c; q = __builtin_elementwise_fma(c, c, q); }
    float s1 = (sd[0] + sd[1]) + (sd[2] + sd[3]);
    float s2 = (q[0] + q[1]) + (q[2] + q[3]);
    s1 = s1 + lane_xor_f32<32>(s1, lane);
    s2 = s2 + lane_xor_f32<32>(s2, lane);
    const float dm = s1 / (float)d;
    float v2 = __builtin_fmaf(-dm, dm, s2 / (float)d);
    v2 = v2 > 0.0f ? v2 : 0.0f;
    return centred ? make_float2(mean + dm, v2) : make_float2(mean, var);
}

// (mean, +-rstd) of the row whose halves lanes i and i + 32 hold - the same bits in both (a + b == b + a).  xr: this lane's
// row + 4 * half (any readable row where there is none); wave-uniform control flow: every lane of the wave calls it.
__device__ __forceinline__ float2 rm_finish(const RowMoments& m, int d, float eps, int lane, const float* __restrict__ xr) {
    float t = (m.s01[0] + m.s01[1]) + (m.s23[0] + m.s23[1]);
    float u = (m.q01[0] + m.q01[1]) + (m.q23[0] + m.q23[1]);
    t = t + lane_xor_f32<32>(t, lane);
    u = u + lane_xor_f32<32>(u, lane);
    float mean = t / (float)d;
    const float ex2 = u / (float)d;
    float var = __builtin_fmaf(-mean, mean, ex2);
    var = var > 0.0f ? var : 0.0f;
    const bool centred = var * RM_RECENTRE < ex2;
    if (__builtin_amdgcn_ballot_w64(centred) != 0ull) {               // rare: a row whose mean dwarfs its spread
        const float2 mv = rm_recentred(xr, centred, mean, var, d, lane);
        mean = mv.x;
        var = mv.y;
    }
    const float rstd = 1.0f / __builtin_sqrtf(var + eps);
    return make_float2(mean, centred ? -rstd : rstd);
}

// One output of a CENTRED row's Linear: the contract's fma chain (per k-group the matrix cores' order 0,4,1,5,2,6,3,7) over
// (x[c] - mean) * w[o][c], weights in the packed B-operand layout [C_out/32][K/8][64 lanes][4].  Plain vector code: rows
// that need it are rare, and the chain is the same whoever runs it.
__device__ __forceinline__ float centred_row_dot(const float* __restrict__ xrow, float mean, const float* __restrict__ wp,
                                                 int kgs, int o) {
    const rm_f32x4* w4 = reinterpret_cast<const rm_f32x4*>(wp) + (size_t)(o >> 5) * kgs * 64 + (o & 31);
    const rm_f32x4* x4 = reinterpret_cast<const rm_f32x4*>(xrow);
    float acc = 0.0f;
#pragma unroll 2
    for (int kg = 0; kg < kgs; ++kg) {
        const rm_f32x4 xa = x4[2 * kg], xb = x4[2 * kg + 1];
        const rm_f32x4 wa = w4[(size_t)kg * 64], wb = w4[(size_t)kg * 64 + 32];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc = __builtin_fmaf(xa[j] - mean, wa[j], acc);
            acc = __builtin_fmaf(xb[j] - mean, wb[j], acc);
        }
    }
    return acc;
}

// the moments of rows row0 .. row0 + 31 (those below n) by one wavefront: lane l -> row row0 + (l & 31), half l >> 5;
// d % 8 == 0.  Eight 16-byte loads in flight per lane.
__device__ __forceinline__ float2 row_moments_wave32(const float* __restrict__ x, long long row0, long long n, int d, float eps,
                                                     int lane) {
    const long long row = row0 + (lane & 31);
    const rm_f32x4* p = reinterpret_cast<const rm_f32x4*>(x + (size_t)(row < n ? row : n - 1) * d + 4 * (lane >> 5));
    RowMoments m;
    rm_zero(m);
    const int kgs = d >> 3;
    int g = 0;
    for (; g + 8 <= kgs; g += 8) {
        rm_f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[2 * (g + u)];
#pragma unroll
        for (int u = 0; u < 8; ++u) rm_add(m, v[u]);
    }
    for (; g < kgs; ++g) rm_add(m, p[2 * g]);
    return rm_finish(m, d, eps, lane, reinterpret_cast<const float*>(p));
}

}  // namespace ipsx
