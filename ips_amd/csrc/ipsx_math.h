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
    const uint32_t k = u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);      // negative: ~u, else u | sign bit - no compare
    return x != x ? 0xFFFFFFFFu : k;
}

__device__ __forceinline__ float max_key_value(uint32_t k) {
    if (k == 0xFFFFFFFFu) return as_float(0x7FC00000u);
    return as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// The value lane ^ J holds, register to register (no LDS crossbar as __shfl_xor's ds_bpermute takes): strides 1, 2, 4
// and 8 stay inside a DPP row of 16 lanes (quad permutes, a row shift pair, a row rotate); 16 and 32 are gfx950's
// v_permlane16_swap / v_permlane32_swap of the value with itself (which leave rows (0,0,2,2) | (1,1,3,3), halves
// (lo,lo) | (hi,hi) in the two results) and one select.
template <int J>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t v, int lane) {
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "stride");
    const int x = (int)v;
    if (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
    if (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
    if (J == 4) {                                                // row_shl:4 (lane i <- i+4) / row_shr:4 (lane i <- i-4)
        const int up = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xF, 0xF, true);
        const int down = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
        return (uint32_t)((lane & 4) ? down : up);
    }
    if (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, true);      // row_ror:8
    if (J == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    }
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (lane & 32) ? r[0] : r[1];
}

template <int J>
__device__ __forceinline__ float lane_xor_f32(float v, int lane) { return as_float(lane_xor_u32<J>(as_u32(v), lane)); }

template <int J>
__device__ __forceinline__ uint64_t lane_xor_u64(uint64_t v, int lane) {
    return ((uint64_t)lane_xor_u32<J>((uint32_t)(v >> 32), lane) << 32) | lane_xor_u32<J>((uint32_t)v, lane);
}

// xor-butterfly sum over the 64 lanes, offsets 32,16,...,1: every lane ends with
// the same total (the second half of wave_sum64 of the oracle).
__device__ __forceinline__ float wave_butterfly_sum(float v) {
    const int lane = (int)__lane_id();
    v = v + lane_xor_f32<32>(v, lane); v = v + lane_xor_f32<16>(v, lane); v = v + lane_xor_f32<8>(v, lane);
    v = v + lane_xor_f32<4>(v, lane); v = v + lane_xor_f32<2>(v, lane); v = v + lane_xor_f32<1>(v, lane);
    return v;
}

// max over the wave with the oracle's NaN rule (a NaN wins)
__device__ __forceinline__ float nanmax(float a, float b) { return (b > a || b != b) ? b : a; }

__device__ __forceinline__ float wave_max(float v) {
    const int lane = (int)__lane_id();
    v = nanmax(v, lane_xor_f32<32>(v, lane)); v = nanmax(v, lane_xor_f32<16>(v, lane)); v = nanmax(v, lane_xor_f32<8>(v, lane));
    v = nanmax(v, lane_xor_f32<4>(v, lane)); v = nanmax(v, lane_xor_f32<2>(v, lane)); v = nanmax(v, lane_xor_f32<1>(v, lane));
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
