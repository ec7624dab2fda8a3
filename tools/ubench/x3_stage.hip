// Micro-benchmark 3: the 8x8-stage pipeline of the split-bf16 ("fp32 as 3 bf16 terms, 6 products") trunk:
// per stage (K-step of 16) 6 x ds_read_b128 (activation planes) + 6 x global_load_dwordx4 (weight planes) + 24 x
// v_mfma_f32_32x32x16_bf16 (768 matrix-pipe cycles).  One wave per SIMD (112 KB of LDS per workgroup).
//   hipcc --offload-arch=gfx950 -O3 -o x3_stage x3_stage.hip && ./x3_stage
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)
constexpr int PL1 = 144, RB1 = 3 * PL1, ZR1 = 64, SLABX = (ZR1 + 1) * RB1;
__device__ __forceinline__ void zero(f32x16& v) { for (int r = 0; r < 16; ++r) v[r] = 0.f; }
struct XTap { const char* s0; const char* s1; };
struct XOp { uint4 p[2][3]; };
__device__ __forceinline__ XTap x_tap(int tap, const char* S, int i, int half) {
    const int t3 = tap / 3; const int dy = t3 - 1, dx = tap - 3 * t3 - 1; const int x = i & 7, y0 = i >> 3;
    const bool okx = (unsigned)(x + dx) < 8u;
    const bool ok0 = okx && (unsigned)(y0 + dy) < 8u, ok1 = okx && (unsigned)(y0 + 4 + dy) < 8u;
    const int p0 = i + dy * 8 + dx; XTap d;
    d.s0 = S + (ok0 ? p0 : ZR1) * RB1 + 16 * half; d.s1 = S + (ok1 ? p0 + 32 : ZR1) * RB1 + 16 * half; return d;
}
template <int KS> __device__ __forceinline__ void x_load(XOp& st, const XTap& d) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        st.p[0][pl] = *reinterpret_cast<const uint4*>(d.s0 + pl * PL1 + KS * 32);
        st.p[1][pl] = *reinterpret_cast<const uint4*>(d.s1 + pl * PL1 + KS * 32);
    }
}
template <int G> __device__ __forceinline__ void x_loadw(XOp& w, const char* wb, unsigned loff, int g) {
    g = g < G ? g : G - 1;
    const char* p = wb + (size_t)g * 3072;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        w.p[0][pl] = *reinterpret_cast<const uint4*>(p + pl * 1024 + loff);
        w.p[1][pl] = *reinterpret_cast<const uint4*>(p + (size_t)G * 3072 + pl * 1024 + loff);
    }
}
// acc[rt][ct] += sum over the 6 significant plane pairs; A = weights (rows = out channels), B = activations (cols = pixels)
__device__ __forceinline__ void x_mma(const XOp& w, const XOp& x, f32x16 (&acc)[2][2]) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = MFMA16(w.p[rt][PA[q]], x.p[ct][PB[q]], acc[rt][ct]);
}
// 24 MFMAs with the 12 loads of the next stage spread between them: 2 MFMA, 1 DS read, 2 MFMA, 1 VMEM read, ...
#define SGM(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define SGD(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define SGV(n) __builtin_amdgcn_sched_group_barrier(0x020, n, 0)
#ifndef VARIANT
#define VARIANT 0
#endif
#if VARIANT == 0
#define GROUPS() _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) { SGM(2); SGD(1); SGM(2); SGV(1); }
#elif VARIANT == 1      // loads every 1.5 MFMA, 6 MFMA at the end
#define GROUPS() _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) { SGM(2); SGD(1); SGM(1); SGV(1); } SGM(6);
#elif VARIANT == 2      // 4 MFMA at the end
#define GROUPS() _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) { SGM(2); SGD(1); SGM(2); SGV(1); } SGM(1); SGD(1); SGM(1); SGV(1); SGM(1); SGD(1); SGM(1); SGV(1); SGM(4);
#elif VARIANT == 3      // LDS reads first half, global loads second half
#define GROUPS() _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) { SGM(2); SGD(1); } _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) { SGM(2); SGV(1); }
#elif VARIANT == 4      // global first, LDS second
#define GROUPS() _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) { SGM(2); SGV(1); } _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) { SGM(2); SGD(1); }
#elif VARIANT == 5      // 2 MFMA lead-in, then even, 2 at the end
#define GROUPS() SGM(2); _Pragma("unroll") for (int q_ = 0; q_ < 5; ++q_) { SGD(1); SGM(2); SGV(1); SGM(2); } SGD(1); SGM(1); SGV(1); SGM(1);
#endif
template <int MODE>
__device__ __forceinline__ void conv_l1x(const void* wp, const char* S, f32x16 (&acc)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
    const char* w = reinterpret_cast<const char*>(wp);
    const unsigned lo = lane * 16;
    XTap cur = x_tap(0, S, i, half);
    XOp xa, xb, w0, w1, w2;
    x_loadw<36>(w0, w, lo, 0); x_loadw<36>(w1, w, lo, 1); x_loadw<36>(w2, w, lo, 1);
    x_load<0>(xa, cur); x_load<0>(xb, cur);
    int g = 0;
#pragma unroll 1
    for (int tap = 0; tap < 9; tap += 3) {      // 3 taps = 12 stages per trip so the 3-slot weight ring closes
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const XTap nxt = x_tap(tap + t < 8 ? tap + t + 1 : 8, S, i, half);
            if (MODE == 0 || MODE == 2) {
                // slot pattern over 4 stages of a tap: ring index (4t + s) % 3
#define STAGE(S_, XC, XN, LOADN, WC, WF)                                           \
    if (MODE == 0) { LOADN; x_loadw<36>(WF, w, lo, g + 2); SB(); x_mma(WC, XC, acc); SB(); }      \
    else { LOADN; x_loadw<36>(WF, w, lo, g + 2); x_mma(WC, XC, acc); GROUPS(); SB(); }            \
    ++g;
                if (t == 0) {
                    STAGE(0, xa, xb, x_load<1>(xb, cur), w0, w2) STAGE(1, xb, xa, x_load<2>(xa, cur), w1, w0)
                    STAGE(2, xa, xb, x_load<3>(xb, cur), w2, w1) STAGE(3, xb, xa, x_load<0>(xa, nxt), w0, w2)
                } else if (t == 1) {
                    STAGE(0, xa, xb, x_load<1>(xb, cur), w1, w0) STAGE(1, xb, xa, x_load<2>(xa, cur), w2, w1)
                    STAGE(2, xa, xb, x_load<3>(xb, cur), w0, w2) STAGE(3, xb, xa, x_load<0>(xa, nxt), w1, w0)
                } else {
                    STAGE(0, xa, xb, x_load<1>(xb, cur), w2, w1) STAGE(1, xb, xa, x_load<2>(xa, cur), w0, w2)
                    STAGE(2, xa, xb, x_load<3>(xb, cur), w1, w0) STAGE(3, xb, xa, x_load<0>(xa, nxt), w2, w1)
                }
            } else {                              // MODE 1: MFMAs only
#pragma unroll
                for (int s = 0; s < 4; ++s) { x_mma(w0, xa, acc); SB(); }
            }
            cur = nxt;
        }
    }
}
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const void* w, const float* seed, float* out, unsigned long long* cyc, int reps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* S = lds + wave * SLABX;
    for (int z = lane; z < SLABX / 4; z += 64) reinterpret_cast<unsigned*>(S)[z] = 0x3C003C00u + (z & 0xFF);
    __syncthreads();
    f32x16 acc[2][2], idn[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) { zero(acc[a][b]); for (int r = 0; r < 16; ++r) idn[a][b][r] = seed[r + lane]; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) conv_l1x<MODE>(reinterpret_cast<const char*>(w) + (size_t)(r & 3) * 2 * 36 * 3072, S, acc, lane);
    const unsigned long long m1 = __builtin_amdgcn_s_memtime();
    (void)t0;
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r] + idn[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = m1 - m0;
}
int main() {
    const int reps = 8, blocks = 256 * 4;
    void* w; float *seed, *out; unsigned long long* cyc;
    const size_t wbytes = (size_t)4 * 2 * 36 * 3072;
    hipMalloc(&w, wbytes); hipMemset(w, 0x3C, wbytes);
    hipMalloc(&seed, 4096); hipMemset(seed, 0, 4096);
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 4 * 8);
    const size_t ldsb = 4 * SLABX;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    for (int mode = 0; mode < 3; ++mode) {
        for (int it = 0; it < 3; ++it) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 256, ldsb>>>(w, seed, out, cyc, reps); else if (mode == 1) k<1><<<blocks, 256, ldsb>>>(w, seed, out, cyc, reps); else k<2><<<blocks, 256, ldsb>>>(w, seed, out, cyc, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(blocks * 4);
            hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            // s_memtime ticks at 100 MHz; convert with the kernel time instead: report per-stage pipe cycles from wall time
            const double stages = (double)reps * 36;
            const double waves_per_simd = (double)blocks * 4 / 1024.0;
            const double cyc_per_stage = ms * 1e-3 * 2.4e9 / (stages * waves_per_simd);
            printf("mode %d (%s): %.3f ms, %.0f cycles/stage @2.4GHz, s_memtime/stage %.0f (ideal 768)\n", mode,
                   mode == 0 ? "full pipeline" : mode == 1 ? "MFMA only" : "loads spread", ms, cyc_per_stage, (double)h[h.size() / 2] / stages);
        }
    }
    hipError_t e = hipGetLastError(); printf("%s\n", hipGetErrorString(e));
    return 0;
}
