// ipsx_math.h - device arithmetic shared by every kernel of libipsx.
//
// The sequences here are the "arithmetic contract" of include/ipsx.h and are
// restated operation for operation in oracle/ips_oracle.cpp (det_expf, wave_sum64,
// rank_key).  Every operation is an exactly rounded IEEE fp32 operation, the
// library is compiled with -ffp-contract=off and without fast-math, so host and
// device produce identical bits.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define IPSX_WAVE 64

namespace ipsx {

__device__ __forceinline__ float as_float(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t as_u32(float f) { return __float_as_uint(f); }

// exp(): Cody-Waite reduction by ln2 = C1 + C2, degree-5 Horner polynomial,
// exponent rebuilt from bits (two-step scaling in the subnormal range).
__device__ __forceinline__ float det_expf(float x) {
    // branch-free form of the oracle's det_expf (same operations on the lanes that take its main path, selects for
    // the special cases): divergent early returns would serialise a wavefront whose lanes mix ranges
    const bool is_nan = x != x, big = x > 88.72f, small = x < -104.0f;
    const float xc = (is_nan || big || small) ? 0.0f : x;
    float n = __builtin_rintf(xc * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, xc);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r);
    y = y + 1.0f;
    const int ni = (int)n;
    // y * 2^ni, in two exact power-of-two steps where 2^ni itself is not a normal float
    const bool sub = ni < -126, top = ni > 127;
    const int e1 = sub ? ni + 127 + 64 : (top ? ni + 126 : ni + 127);
    const float s2 = sub ? 5.42101086242752217e-20f /* 2^-64 */ : (top ? 2.0f : 1.0f);
    float res = (y * as_float((uint32_t)e1 << 23)) * s2;
    res = small ? 0.0f : res;
    res = big ? __builtin_huge_valf() : res;
    return is_nan ? x : res;
}

// det_expf for arguments that are never positive (a logit minus its row maximum: <= 0, or NaN when the row holds a NaN or
// an infinity): the same operations as det_expf on every such input - the overflow tests, whose branches cannot be
// taken, are the only thing left out.
__device__ __forceinline__ float det_expf_np(float x) {
    const bool is_nan = x != x, small = x < -104.0f;
    const float xc = (is_nan || small) ? 0.0f : x;
    float n = __builtin_rintf(xc * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, xc);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r);
    y = y + 1.0f;
    const int ni = (int)n;
    const bool sub = ni < -126;
    const int e1 = sub ? ni + 127 + 64 : ni + 127;
    const float s2 = sub ? 5.42101086242752217e-20f /* 2^-64 */ : 1.0f;
    float res = (y * as_float((uint32_t)e1 << 23)) * s2;
    res = small ? 0.0f : res;
    return is_nan ? x : res;
}

// order-preserving integer image of a float for max reductions with one v_max_u32 per step: larger float <=> larger
// key, every NaN -> 0xFFFFFFFF (a NaN wins, as in nanmax); 0 is below every float (the neutral element)
__device__ __forceinline__ uint32_t max_key(float x) {
    const uint32_t u = as_u32(x);
    const uint32_t k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return x != x ? 0xFFFFFFFFu : k;
}

__device__ __forceinline__ float max_key_value(uint32_t k) {
    if (k == 0xFFFFFFFFu) return as_float(0x7FC00000u);
    return as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// xor-butterfly sum over the 64 lanes, offsets 32,16,...,1: every lane ends with
// the same total (the second half of wave_sum64 of the oracle).
__device__ __forceinline__ float wave_butterfly_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

// max over the wave with the oracle's NaN rule (a NaN wins)
__device__ __forceinline__ float nanmax(float a, float b) { return (b > a || b != b) ? b : a; }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = nanmax(v, __shfl_xor(v, off, 64));
    return v;
}

// ranking key: score descending, NaN first, -0 == +0, ties -> earlier position
__device__ __forceinline__ uint64_t rank_key(float s, uint32_t pos) {
    uint32_t u;
    if (s != s) {
        u = 0xFFFFFFFFu;
    } else {
        s = s + 0.0f;
        u = as_u32(s);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    return ((uint64_t)u << 32) | (uint64_t)(0xFFFFFFFFu - pos);
}

__device__ __forceinline__ uint32_t key_pos(uint64_t key) { return 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull); }

__device__ __forceinline__ float key_score(uint64_t key) {
    uint32_t u = (uint32_t)(key >> 32);
    if (u == 0xFFFFFFFFu) return as_float(0x7FC00000u);
    return as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}

}  // namespace ipsx
