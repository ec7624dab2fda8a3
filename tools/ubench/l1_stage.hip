// Micro-benchmark 2: the actual layer1 pipeline of fused_trunk.hip (copied) with parts knocked out.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)
constexpr int CS1 = 68;
struct L1Tap { const float* s0; const float* s1; bool ok0, ok1; };
struct L1Stage { float a[2][4]; bool ok0, ok1; };
__device__ __forceinline__ void zero(f32x16& v) { for (int r = 0; r < 16; ++r) v[r] = 0.f; }
__device__ __forceinline__ L1Tap l1_tap(int tap, const float* S, int i, int half) {
    const int t3 = tap / 3; const int dy = t3 - 1, dx = tap - 3 * t3 - 1; const int x = i & 7, y0 = i >> 3;
    const bool okx = (unsigned)(x + dx) < 8u; L1Tap d;
    d.ok0 = okx && (unsigned)(y0 + dy) < 8u; d.ok1 = okx && (unsigned)(y0 + 4 + dy) < 8u;
    const int p0 = i + dy * 8 + dx;
    d.s0 = S + half * CS1 + (d.ok0 ? p0 : 0); d.s1 = S + half * CS1 + (d.ok1 ? p0 + 32 : 0); return d;
}
template <int MODE, int CG> __device__ __forceinline__ void l1_load(L1Stage& st, const L1Tap& d) {
    st.ok0 = d.ok0; st.ok1 = d.ok1;
    if (MODE == 2 || MODE == 3) return;               // no A loads
#pragma unroll
    for (int j = 0; j < 4; ++j) { st.a[0][j] = d.s0[(CG * 8 + 2 * j) * CS1]; st.a[1][j] = d.s1[(CG * 8 + 2 * j) * CS1]; }
}
template <int MODE> __device__ __forceinline__ void l1_loadb(float4 (&b)[2], const float4* w, int g) {
    if (MODE == 1 || MODE == 3) return;               // no B loads
    g = g < 72 ? g : 71; b[0] = w[g * 64]; b[1] = w[(72 + g) * 64];
}
template <int MODE> __device__ __forceinline__ void l1_mma(const L1Stage& st, const float4 (&b)[2], f32x16 (&acc)[2][2]) {
    const float bb0[4] = {b[0].x, b[0].y, b[0].z, b[0].w}; const float bb1[4] = {b[1].x, b[1].y, b[1].z, b[1].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a0 = st.ok0 ? st.a[0][j] : 0.0f, a1 = st.ok1 ? st.a[1][j] : 0.0f;
        if (MODE == 5) { a0 = st.a[0][j]; a1 = st.a[1][j]; }     // no selects
        acc[0][0] = MFMA(a0, bb0[j], acc[0][0]); acc[0][1] = MFMA(a0, bb1[j], acc[0][1]);
        acc[1][0] = MFMA(a1, bb0[j], acc[1][0]); acc[1][1] = MFMA(a1, bb1[j], acc[1][1]);
    }
}
template <int MODE>
__device__ __forceinline__ void conv_l1(const float* __restrict__ wp, const float* S, f32x16 (&acc)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
    const float4* w = reinterpret_cast<const float4*>(wp) + lane;
    L1Tap cur = l1_tap(0, S, i, half);
    L1Stage sa, sb; float4 b0[2], b1[2], b2[2], b3[2];
    for (int q = 0; q < 2; ++q) { b0[q] = w[q * 64]; b1[q] = w[(q + 2) * 64]; b2[q] = w[(q + 4) * 64]; b3[q] = w[(q + 6) * 64]; }
    for (int j = 0; j < 4; ++j) { sa.a[0][j] = S[j]; sa.a[1][j] = S[j + 8]; sb.a[0][j] = S[j + 16]; sb.a[1][j] = S[j + 24]; }
    l1_load<MODE, 0>(sa, cur);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const L1Tap nxt = l1_tap(tap < 8 ? tap + 1 : 8, S, i, half);
        const int g = tap * 8;
        l1_load<MODE, 1>(sb, cur); l1_loadb<MODE>(b3, w, g + 3);  SB(); l1_mma<MODE>(sa, b0, acc); SB();
        l1_load<MODE, 2>(sa, cur); l1_loadb<MODE>(b0, w, g + 4);  SB(); l1_mma<MODE>(sb, b1, acc); SB();
        l1_load<MODE, 3>(sb, cur); l1_loadb<MODE>(b1, w, g + 5);  SB(); l1_mma<MODE>(sa, b2, acc); SB();
        l1_load<MODE, 4>(sa, cur); l1_loadb<MODE>(b2, w, g + 6);  SB(); l1_mma<MODE>(sb, b3, acc); SB();
        l1_load<MODE, 5>(sb, cur); l1_loadb<MODE>(b3, w, g + 7);  SB(); l1_mma<MODE>(sa, b0, acc); SB();
        l1_load<MODE, 6>(sa, cur); l1_loadb<MODE>(b0, w, g + 8);  SB(); l1_mma<MODE>(sb, b1, acc); SB();
        l1_load<MODE, 7>(sb, cur); l1_loadb<MODE>(b1, w, g + 9);  SB(); l1_mma<MODE>(sa, b2, acc); SB();
        l1_load<MODE, 0>(sa, nxt); l1_loadb<MODE>(b2, w, g + 10); SB(); l1_mma<MODE>(sb, b3, acc); SB();
        cur = nxt;
    }
}
__device__ __forceinline__ void l1_mask(L1Stage& st) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { st.a[0][j] = st.ok0 ? st.a[0][j] : 0.0f; st.a[1][j] = st.ok1 ? st.a[1][j] : 0.0f; }
}
__device__ __forceinline__ void l1_mma2(const L1Stage& st, const float4 (&b)[2], f32x16 (&acc)[2][2]) {
    const float bb0[4] = {b[0].x, b[0].y, b[0].z, b[0].w}; const float bb1[4] = {b[1].x, b[1].y, b[1].z, b[1].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc[0][0] = MFMA(st.a[0][j], bb0[j], acc[0][0]); acc[0][1] = MFMA(st.a[0][j], bb1[j], acc[0][1]);
        acc[1][0] = MFMA(st.a[1][j], bb0[j], acc[1][0]); acc[1][1] = MFMA(st.a[1][j], bb1[j], acc[1][1]);
    }
}
// new scheme: masks applied in place one stage ahead; weight ring refills the slot consumed 2 stages ago
__device__ __forceinline__ void conv_l1_v2(const float* __restrict__ wp, const float* S, f32x16 (&acc)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
    const float4* w = reinterpret_cast<const float4*>(wp) + lane;
    L1Tap cur = l1_tap(0, S, i, half);
    L1Stage sa, sb; float4 b0[2], b1[2], b2[2], b3[2];
    l1_loadb<0>(b0, w, 0); l1_loadb<0>(b1, w, 1); l1_loadb<0>(b3, w, 0);
    l1_load<0, 0>(sa, cur); l1_mask(sa);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const L1Tap nxt = l1_tap(tap < 8 ? tap + 1 : 8, S, i, half);
        const int g = tap * 8;
        l1_load<0, 1>(sb, cur); l1_loadb<0>(b2, w, g + 2);  SB(); l1_mma2(sa, b0, acc); SB(); l1_mask(sb); SB();
        l1_load<0, 2>(sa, cur); l1_loadb<0>(b3, w, g + 3);  SB(); l1_mma2(sb, b1, acc); SB(); l1_mask(sa); SB();
        l1_load<0, 3>(sb, cur); l1_loadb<0>(b0, w, g + 4);  SB(); l1_mma2(sa, b2, acc); SB(); l1_mask(sb); SB();
        l1_load<0, 4>(sa, cur); l1_loadb<0>(b1, w, g + 5);  SB(); l1_mma2(sb, b3, acc); SB(); l1_mask(sa); SB();
        l1_load<0, 5>(sb, cur); l1_loadb<0>(b2, w, g + 6);  SB(); l1_mma2(sa, b0, acc); SB(); l1_mask(sb); SB();
        l1_load<0, 6>(sa, cur); l1_loadb<0>(b3, w, g + 7);  SB(); l1_mma2(sb, b1, acc); SB(); l1_mask(sa); SB();
        l1_load<0, 7>(sb, cur); l1_loadb<0>(b0, w, g + 8);  SB(); l1_mma2(sa, b2, acc); SB(); l1_mask(sb); SB();
        l1_load<0, 0>(sa, nxt); l1_loadb<0>(b1, w, g + 9);  SB(); l1_mma2(sb, b3, acc); SB(); l1_mask(sa); SB();
        cur = nxt;
    }
}
// M7: one phase per stage - the next stage's LDS reads, the weight refill and the in-place masks are
// placed in the gaps of this stage's 16 MFMAs, order pinned by sched_barrier after every item.
template <int CGN>
__device__ __forceinline__ void l1_stage(const L1Stage& c, L1Stage& n, const L1Tap& dn, const float4 (&b)[2],
                                         float4 (&bf)[2], const float4* w, int gfill, f32x16 (&acc)[2][2]) {
    const float bb0[4] = {b[0].x, b[0].y, b[0].z, b[0].w}; const float bb1[4] = {b[1].x, b[1].y, b[1].z, b[1].w};
    gfill = gfill < 72 ? gfill : 71;
    n.ok0 = dn.ok0; n.ok1 = dn.ok1;
    // k-step 0: MFMAs + the four LDS read pairs of the next stage
    acc[0][0] = MFMA(c.a[0][0], bb0[0], acc[0][0]); SB();
    n.a[0][0] = dn.s0[(CGN * 8 + 0) * CS1]; n.a[0][1] = dn.s0[(CGN * 8 + 2) * CS1]; SB();
    acc[0][1] = MFMA(c.a[0][0], bb1[0], acc[0][1]); SB();
    n.a[1][0] = dn.s1[(CGN * 8 + 0) * CS1]; n.a[1][1] = dn.s1[(CGN * 8 + 2) * CS1]; SB();
    acc[1][0] = MFMA(c.a[1][0], bb0[0], acc[1][0]); SB();
    n.a[0][2] = dn.s0[(CGN * 8 + 4) * CS1]; n.a[0][3] = dn.s0[(CGN * 8 + 6) * CS1]; SB();
    acc[1][1] = MFMA(c.a[1][0], bb1[0], acc[1][1]); SB();
    n.a[1][2] = dn.s1[(CGN * 8 + 4) * CS1]; n.a[1][3] = dn.s1[(CGN * 8 + 6) * CS1]; SB();
    // k-step 1: MFMAs + weight refill
    acc[0][0] = MFMA(c.a[0][1], bb0[1], acc[0][0]); SB();
    bf[0] = w[gfill * 64]; SB();
    acc[0][1] = MFMA(c.a[0][1], bb1[1], acc[0][1]); SB();
    bf[1] = w[(72 + gfill) * 64]; SB();
    acc[1][0] = MFMA(c.a[1][1], bb0[1], acc[1][0]); SB();
    acc[1][1] = MFMA(c.a[1][1], bb1[1], acc[1][1]); SB();
    // k-step 2: MFMAs + in-place halo masks of the next stage's operands
    acc[0][0] = MFMA(c.a[0][2], bb0[2], acc[0][0]); SB();
    n.a[0][0] = n.ok0 ? n.a[0][0] : 0.0f; n.a[0][1] = n.ok0 ? n.a[0][1] : 0.0f; SB();
    acc[0][1] = MFMA(c.a[0][2], bb1[2], acc[0][1]); SB();
    n.a[1][0] = n.ok1 ? n.a[1][0] : 0.0f; n.a[1][1] = n.ok1 ? n.a[1][1] : 0.0f; SB();
    acc[1][0] = MFMA(c.a[1][2], bb0[2], acc[1][0]); SB();
    n.a[0][2] = n.ok0 ? n.a[0][2] : 0.0f; n.a[0][3] = n.ok0 ? n.a[0][3] : 0.0f; SB();
    acc[1][1] = MFMA(c.a[1][2], bb1[2], acc[1][1]); SB();
    n.a[1][2] = n.ok1 ? n.a[1][2] : 0.0f; n.a[1][3] = n.ok1 ? n.a[1][3] : 0.0f; SB();
    // k-step 3
    acc[0][0] = MFMA(c.a[0][3], bb0[3], acc[0][0]); SB();
    acc[0][1] = MFMA(c.a[0][3], bb1[3], acc[0][1]); SB();
    acc[1][0] = MFMA(c.a[1][3], bb0[3], acc[1][0]); SB();
    acc[1][1] = MFMA(c.a[1][3], bb1[3], acc[1][1]); SB();
}
__device__ __forceinline__ void conv_l1_v3(const float* __restrict__ wp, const float* S, f32x16 (&acc)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
    const float4* w = reinterpret_cast<const float4*>(wp) + lane;
    L1Tap cur = l1_tap(0, S, i, half);
    L1Stage sa, sb; float4 b0[2], b1[2], b2[2], b3[2];
    l1_loadb<0>(b0, w, 0); l1_loadb<0>(b1, w, 1); l1_loadb<0>(b3, w, 0);
    l1_load<0, 0>(sa, cur); l1_mask(sa);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const L1Tap nxt = l1_tap(tap < 8 ? tap + 1 : 8, S, i, half);
        const int g = tap * 8;
        l1_stage<1>(sa, sb, cur, b0, b2, w, g + 2, acc);
        l1_stage<2>(sb, sa, cur, b1, b3, w, g + 3, acc);
        l1_stage<3>(sa, sb, cur, b2, b0, w, g + 4, acc);
        l1_stage<4>(sb, sa, cur, b3, b1, w, g + 5, acc);
        l1_stage<5>(sa, sb, cur, b0, b2, w, g + 6, acc);
        l1_stage<6>(sb, sa, cur, b1, b3, w, g + 7, acc);
        l1_stage<7>(sa, sb, cur, b2, b0, w, g + 8, acc);
        l1_stage<0>(sb, sa, nxt, b3, b1, w, g + 9, acc);
        cur = nxt;
    }
}
// M8: pixel-major LDS image [pix][68] with a zero pixel row (index 64): one ds_read_b128 per m-tile
// per stage, no halo selects; weights ring distance 2.
struct P8Tap { const float* s0; const float* s1; };
struct P8Stage { float4 a0, a1; };
__device__ __forceinline__ P8Tap p8_tap(int tap, const float* S, int i, int half) {
    const int t3 = tap / 3; const int dy = t3 - 1, dx = tap - 3 * t3 - 1; const int x = i & 7, y0 = i >> 3;
    const bool okx = (unsigned)(x + dx) < 8u;
    const bool ok0 = okx && (unsigned)(y0 + dy) < 8u, ok1 = okx && (unsigned)(y0 + 4 + dy) < 8u;
    const int p0 = i + dy * 8 + dx; P8Tap d;
    d.s0 = S + (ok0 ? p0 : 64) * 68 + 4 * half; d.s1 = S + (ok1 ? p0 + 32 : 64) * 68 + 4 * half; return d;
}
template <int CG> __device__ __forceinline__ void p8_load(P8Stage& st, const P8Tap& d) {
    st.a0 = *reinterpret_cast<const float4*>(d.s0 + CG * 8); st.a1 = *reinterpret_cast<const float4*>(d.s1 + CG * 8);
}
__device__ __forceinline__ void p8_mma(const P8Stage& st, const float4 (&b)[2], f32x16 (&acc)[2][2]) {
    const float a0[4] = {st.a0.x, st.a0.y, st.a0.z, st.a0.w}, a1[4] = {st.a1.x, st.a1.y, st.a1.z, st.a1.w};
    const float bb0[4] = {b[0].x, b[0].y, b[0].z, b[0].w}; const float bb1[4] = {b[1].x, b[1].y, b[1].z, b[1].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc[0][0] = MFMA(a0[j], bb0[j], acc[0][0]); acc[0][1] = MFMA(a0[j], bb1[j], acc[0][1]);
        acc[1][0] = MFMA(a1[j], bb0[j], acc[1][0]); acc[1][1] = MFMA(a1[j], bb1[j], acc[1][1]);
    }
}
__device__ __forceinline__ void conv_l1_v4(const float* __restrict__ wp, const float* S, f32x16 (&acc)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
    const float4* w = reinterpret_cast<const float4*>(wp) + lane;
    P8Tap cur = p8_tap(0, S, i, half);
    P8Stage sa, sb; float4 b0[2], b1[2], b2[2], b3[2];
    l1_loadb<0>(b0, w, 0); l1_loadb<0>(b1, w, 1); l1_loadb<0>(b3, w, 0);
    p8_load<0>(sa, cur);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const P8Tap nxt = p8_tap(tap < 8 ? tap + 1 : 8, S, i, half);
        const int g = tap * 8;
        p8_load<1>(sb, cur); l1_loadb<0>(b2, w, g + 2);  SB(); p8_mma(sa, b0, acc); SB();
        p8_load<2>(sa, cur); l1_loadb<0>(b3, w, g + 3);  SB(); p8_mma(sb, b1, acc); SB();
        p8_load<3>(sb, cur); l1_loadb<0>(b0, w, g + 4);  SB(); p8_mma(sa, b2, acc); SB();
        p8_load<4>(sa, cur); l1_loadb<0>(b1, w, g + 5);  SB(); p8_mma(sb, b3, acc); SB();
        p8_load<5>(sb, cur); l1_loadb<0>(b2, w, g + 6);  SB(); p8_mma(sa, b0, acc); SB();
        p8_load<6>(sa, cur); l1_loadb<0>(b3, w, g + 7);  SB(); p8_mma(sb, b1, acc); SB();
        p8_load<7>(sb, cur); l1_loadb<0>(b0, w, g + 8);  SB(); p8_mma(sa, b2, acc); SB();
        p8_load<0>(sa, nxt); l1_loadb<0>(b1, w, g + 9);  SB(); p8_mma(sb, b3, acc); SB();
        cur = nxt;
    }
}
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ wp, const float* gin, float* gout, unsigned long long* cyc) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* S = lds + wave * 65 * CS1;
    for (int i = lane; i < 65 * CS1; i += 64) S[i] = gin[i & 4095];
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) zero(acc[a][b]);
    SB(); const unsigned long long t0 = __builtin_amdgcn_s_memtime(); SB();
#pragma unroll 1
    for (int rep = 0; rep < 8; ++rep) { if (MODE == 6) conv_l1_v2(wp, S, acc, lane); else if (MODE == 7) conv_l1_v3(wp, S, acc, lane); else if (MODE == 8) conv_l1_v4(wp, S, acc, lane); else conv_l1<MODE>(wp, S, acc, lane); }
    SB(); const unsigned long long t1 = __builtin_amdgcn_s_memtime(); SB();
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    gout[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
template <int MODE> void run(const char* name, float* wp, float* gin, float* gout, unsigned long long* cyc, int blocks, size_t lds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int it = 0; it < 3; ++it) k<MODE><<<blocks, 256, lds>>>(wp, gin, gout, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-40s blocks %4d: median %8.1f cycles/stage (1024 ideal)\n", name, blocks, (double)h[h.size() / 2] / (8 * 72));
}
int main() {
    float *wp, *gin, *gout; unsigned long long* cyc;
    hipMalloc(&wp, 2 * 72 * 64 * 16 + 65536); hipMalloc(&gin, 1 << 16); hipMalloc(&gout, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 4 * 8);
    std::vector<float> h(2 * 72 * 64 * 4 + 16384);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(wp, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(gin, h.data(), 1 << 16, hipMemcpyHostToDevice);
    for (int two = 0; two < 2; ++two) {
        const int blocks = two ? 512 : 256; const size_t lds = two ? 4 * 65 * 68 * 4 : 100 * 1024;
        printf("---- %d wave(s) per SIMD\n", two + 1);
        run<0>("M0 real layer1 pipeline", wp, gin, gout, cyc, blocks, lds);
        run<1>("M1 no B (weight) loads", wp, gin, gout, cyc, blocks, lds);
        run<2>("M2 no A (LDS) loads", wp, gin, gout, cyc, blocks, lds);
        run<3>("M3 no loads at all", wp, gin, gout, cyc, blocks, lds);
        run<5>("M5 real, no selects", wp, gin, gout, cyc, blocks, lds);
        run<6>("M6 masks one stage ahead, ring dist 2", wp, gin, gout, cyc, blocks, lds);
        run<7>("M7 loads+masks inside the MFMA gaps", wp, gin, gout, cyc, blocks, lds);
        run<8>("M8 pixel-major image, b128 reads, no masks", wp, gin, gout, cyc, blocks, lds);
    }
    return 0;
}
