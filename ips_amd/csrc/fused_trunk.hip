// fused_trunk.hip - the whole patch encoder of the Megapixel-MNIST configuration in ONE
// kernel, activations resident in LDS.
//
// Trunk (reference architecture/ips_net.py:17-52 with config/mnist_config.yml, 32-px
// patches): conv7x7/2(1->64)+BN+ReLU -> maxpool3x3/2 -> 2 x BasicBlock(64) @8x8 ->
// BasicBlock(64->128, /2, 1x1 projection) -> BasicBlock(128) @4x4 -> global average pool.
// 18,628,608 MAC per patch; 4 KiB in, 512 B out: arithmetic intensity ~8 kFLOP/B, so the
// bound is the fp32 matrix pipe (157 TFLOP/s), not HBM.
//
// One workgroup = 4 wavefronts = 4 patches, 68 KiB of LDS (one 17 KiB slab per patch), two
// workgroups per CU (2 waves per SIMD).  Every contraction is v_mfma_f32_32x32x2_f32 in the
// canonical k order, so the embeddings are bit-identical to the layer-by-layer kernels of
// conv.hip and to the oracle.
//
//   stem      wave = patch.  The 32x32 input sits in the slab; 8 tiles of 2 output rows
//             (32 px) x 64 channels; BN+ReLU in registers; the 3x3/2 max-pool is done ON
//             the accumulators (one lane-half exchange), and its output lands in exactly the
//             register layout of the 8x8 stage's MFMA C tile - it is the first block's identity.
//   layer1    wave = patch: 64 px x 64 ch = 2x2 accumulators.  A from the slab ([c][pix],
//             channel stride 68 floats, halo by select), B = pre-packed weights streamed from
//             L2 (16 B per lane per 4 k-steps).  conv1 output overwrites the slab in place
//             (its input is dead: the identity lives in registers).
//   layer2    the 4 waves share the 4 patches: M = 4 x 16 px = 2 tiles, wave w owns output
//             channels 32w..32w+31, so each weight is fetched once per workgroup.
//   avgpool   sequential 16-term sums from the slab (the oracle's order).
//
// No global-memory round trips between layers: HBM traffic is the 4 KiB patch, the 512 B
// embedding and the L2-resident 2.7 MB of weights.

#include "ipsx_common.h"
#include "ipsx_math.h"

namespace ipsx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CS1 = 68;              // channel stride of the 64ch x 8x8 stage (floats)
constexpr int CS2 = 20;              // channel stride of the 128ch x 4x4 stage
constexpr int SLAB = 64 * CS1;       // floats per patch slab (17,408 B); >= 128*CS2 and >= 1024

struct FusedArgs {
    const float* patches;
    float* emb;
    long long n;
    const float *w_stem, *a_stem, *s_stem;
    const float *w[8], *al[8], *sh[8];       // l1.0.c1 l1.0.c2 l1.1.c1 l1.1.c2 l2.0.c1 l2.0.c2 l2.1.c1 l2.1.c2
    const float *w_down, *a_down, *s_down;
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Ordering point for a slab that only ONE wavefront touches (stem and 8x8 stage: wave = patch).
// DS operations of a wavefront are executed in issue order, so a later ds_read of another lane
// sees an earlier ds_write; what is needed is that the compiler keeps the order and that reads
// issued before are complete before the slab is overwritten.
__device__ __forceinline__ void wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void zero(f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.0f;
}

// ------------------------------------------------------------------ stem + max-pool
// S: this wave's slab holding the 32x32 input at [0, 1024).  On return idn[mt][nt] holds the
// pooled 8x8x64 activation in MFMA C layout (lane = channel, rows = pixels).
__device__ __forceinline__ void stem_pool(const FusedArgs& a, const float* S, f32x16 (&idn)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
    const int ox = i & 15;
    const float4* wp0 = reinterpret_cast<const float4*>(a.w_stem) + lane;      // n-tile 0, 7 k-groups
    const float4* wp1 = wp0 + 7 * 64;
    const float al0 = a.a_stem[i], sh0 = a.s_stem[i], al1 = a.a_stem[32 + i], sh1 = a.s_stem[32 + i];
    // column validity of the 7 taps for this lane's output column (tile independent)
    unsigned colmask = 0;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) colmask |= ((unsigned)(2 * ox + kx - 3) < 32u ? 1u : 0u) << kx;

    float prev[2][16];                                                         // stem row 2t-1, per n-tile
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int x = 0; x < 16; ++x) prev[nt][x] = -__builtin_huge_valf();

#pragma unroll 1
    for (int t = 0; t < 8; ++t) {
        const int oy = 2 * t + (i >> 4);
        unsigned rowmask = 0;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) rowmask |= ((unsigned)(2 * oy + ky - 3) < 32u ? 1u : 0u) << ky;
        const int base = (2 * oy - 3) * 32 + (2 * ox - 3);
        // the 3.5 KiB of stem weights are re-read per tile (L1/L2 resident) instead of pinning 56
        // VGPRs across the pooling code; the opaque zero keeps the loads inside the loop
        int opq = 0;
        asm volatile("" : "+s"(opq));
        float bw0[28], bw1[28];
#pragma unroll
        for (int kg = 0; kg < 7; ++kg) {
            const float4 v0 = wp0[(kg + opq) * 64], v1 = wp1[(kg + opq) * 64];
            bw0[4 * kg] = v0.x; bw0[4 * kg + 1] = v0.y; bw0[4 * kg + 2] = v0.z; bw0[4 * kg + 3] = v0.w;
            bw1[4 * kg] = v1.x; bw1[4 * kg + 1] = v1.y; bw1[4 * kg + 2] = v1.z; bw1[4 * kg + 3] = v1.w;
        }
        f32x16 acc0, acc1;
        zero(acc0); zero(acc1);
        float av[25];
        bool okv[25];
#pragma unroll
        for (int s = 0; s < 25; ++s) {                    // k = 2s + half; k = 49 is zero padding
            const int k0 = 2 * s, k1 = 2 * s + 1;
            const int ky0 = k0 / 7, kx0 = k0 % 7;
            const int ky1 = k1 < 49 ? k1 / 7 : 0, kx1 = k1 < 49 ? k1 % 7 : 0;
            const int ky = half ? ky1 : ky0, kx = half ? kx1 : kx0;
            bool ok = ((rowmask >> ky) & (colmask >> kx) & 1u) != 0;
            if (k1 >= 49) ok = ok && !half;
            okv[s] = ok;
            av[s] = S[ok ? base + ky * 32 + kx : 0];
        }
        __builtin_amdgcn_sched_barrier(0);                 // all 25 LDS reads in flight before the MFMAs
#pragma unroll
        for (int s = 0; s < 25; ++s) {
            const float v = okv[s] ? av[s] : 0.0f;
            acc0 = MFMA(v, bw0[s], acc0);
            acc1 = MFMA(v, bw1[s], acc1);
        }
        // BN + ReLU, then give every lane all 32 pixels of its channel (swap lane halves)
        float own[2][16], oth[2][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v0 = __builtin_fmaf(acc0[r], al0, sh0), v1 = __builtin_fmaf(acc1[r], al1, sh1);
            own[0][r] = v0 > 0.0f ? v0 : 0.0f;
            own[1][r] = v1 > 0.0f ? v1 : 0.0f;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oth[nt][r] = __shfl_xor(own[nt][r], 32, 64);
        float pooled[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            // tile pixel p (0..31): row = p >> 4, col = p & 15; C row p lives in lane-half (p>>2)&1, reg (p&3)+4*(p>>3)
            float row0[16], row1[16];
#pragma unroll
            for (int p = 0; p < 32; ++p) {
                const int reg = (p & 3) + 4 * (p >> 3), hp = (p >> 2) & 1;
                const float v = (hp == half) ? own[nt][reg] : oth[nt][reg];
                if (p < 16) row0[p] = v; else row1[p - 16] = v;
            }
            // pooled row t, columns 4*half .. 4*half+3: max over rows {2t-1, 2t, 2t+1} x cols {2x-1, 2x, 2x+1}
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float m0 = -__builtin_huge_valf(), m1 = m0;          // the two candidates for px = q and px = 4+q
#pragma unroll
                for (int d = -1; d <= 1; ++d) {
                    const int c0 = 2 * q + d, c1 = 2 * (4 + q) + d;
                    if (c0 >= 0) { m0 = nanmax(m0, prev[nt][c0]); m0 = nanmax(m0, row0[c0]); m0 = nanmax(m0, row1[c0]); }
                    if (c1 < 16) { m1 = nanmax(m1, prev[nt][c1]); m1 = nanmax(m1, row0[c1]); m1 = nanmax(m1, row1[c1]); }
                }
                pooled[nt][q] = half ? m1 : m0;
            }
#pragma unroll
            for (int x = 0; x < 16; ++x) prev[nt][x] = row1[x];
        }
        // pooled row t -> C-layout registers of the 8x8 stage: pix = 8t + 4*half + q  <=>  [mt = t>>2][reg = 4*(t&3)+q]
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            if (tt == t) {                                   // wave-uniform; static register indices
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) idn[tt >> 2][nt][4 * (tt & 3) + q] = pooled[nt][q];
            }
        }
    }
}

// write a 64px x 64ch wave tile (C layout) into the slab as [c][pix], 16 B per store
__device__ __forceinline__ void store_l1(float* S, const f32x16 (&v)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float* d = S + (nt * 32 + i) * CS1 + mt * 32 + 8 * q + 4 * half;
                *reinterpret_cast<float4*>(d) =
                    make_float4(v[mt][nt][4 * q], v[mt][nt][4 * q + 1], v[mt][nt][4 * q + 2], v[mt][nt][4 * q + 3]);
            }
}

// ------------------------------------------------------------------ 8x8 stage, wave = patch
// acc = conv3x3(S) over K = 9*64 in tap-major order.  wp: packed weights (2 n-tiles x 72 k-groups).
//
// Software pipeline over k-groups (8 k = 4 MFMA steps = 16 MFMAs = 1024 matrix-pipe cycles):
// the LDS reads and the 2 KiB weight fetch of group g+1 are issued before the MFMAs of group g;
// sched_barrier keeps the compiler from sinking them back next to their uses.  Everything that
// depends only on the tap (halo mask, source pixel) is computed once per tap; inside a tap every
// address is base + immediate.
struct L1Tap {
    const float* s0;    // slab pointer of this lane's source pixel, m-tile 0 (halo -> pixel 0)
    const float* s1;    // m-tile 1
    bool ok0, ok1;
};

struct L1Stage {
    float a[2][4];      // raw A values [m-tile][k-step]
    bool ok0, ok1;
    float4 b[2];        // weights of the two n-tiles
};

__device__ __forceinline__ L1Tap l1_tap(int tap, const float* S, int i, int half) {
    const int t3 = tap / 3;
    const int dy = t3 - 1, dx = tap - 3 * t3 - 1;
    const int x = i & 7, y0 = i >> 3;
    const bool okx = (unsigned)(x + dx) < 8u;
    L1Tap d;
    d.ok0 = okx && (unsigned)(y0 + dy) < 8u;
    d.ok1 = okx && (unsigned)(y0 + 4 + dy) < 8u;
    const int p0 = i + dy * 8 + dx;
    d.s0 = S + half * CS1 + (d.ok0 ? p0 : 0);
    d.s1 = S + half * CS1 + (d.ok1 ? p0 + 32 : 0);
    return d;
}

template <int CG>
__device__ __forceinline__ void l1_load(L1Stage& st, const L1Tap& d, const float4* w) {
    st.ok0 = d.ok0; st.ok1 = d.ok1;
    st.b[0] = w[CG * 64];
    st.b[1] = w[72 * 64 + CG * 64];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        st.a[0][j] = d.s0[(CG * 8 + 2 * j) * CS1];
        st.a[1][j] = d.s1[(CG * 8 + 2 * j) * CS1];
    }
}

__device__ __forceinline__ void l1_mma(const L1Stage& st, f32x16 (&acc)[2][2]) {
    const float bb0[4] = {st.b[0].x, st.b[0].y, st.b[0].z, st.b[0].w};
    const float bb1[4] = {st.b[1].x, st.b[1].y, st.b[1].z, st.b[1].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = st.ok0 ? st.a[0][j] : 0.0f;
        const float a1 = st.ok1 ? st.a[1][j] : 0.0f;
        acc[0][0] = MFMA(a0, bb0[j], acc[0][0]);
        acc[0][1] = MFMA(a0, bb1[j], acc[0][1]);
        acc[1][0] = MFMA(a1, bb0[j], acc[1][0]);
        acc[1][1] = MFMA(a1, bb1[j], acc[1][1]);
    }
}

#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void conv_l1(const float* __restrict__ wp, const float* S, f32x16 (&acc)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
    const float4* w = reinterpret_cast<const float4*>(wp) + lane;      // advances 8 k-groups per tap
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) zero(acc[mt][nt]);
    L1Tap cur = l1_tap(0, S, i, half);
    L1Stage sa, sb;
    l1_load<0>(sa, cur, w);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const L1Tap nxt = l1_tap(tap < 8 ? tap + 1 : 8, S, i, half);
        const float4* wn = tap < 8 ? w + 8 * 64 : w;                   // last prefetch re-reads valid memory
        l1_load<1>(sb, cur, w); SB(); l1_mma(sa, acc); SB();
        l1_load<2>(sa, cur, w); SB(); l1_mma(sb, acc); SB();
        l1_load<3>(sb, cur, w); SB(); l1_mma(sa, acc); SB();
        l1_load<4>(sa, cur, w); SB(); l1_mma(sb, acc); SB();
        l1_load<5>(sb, cur, w); SB(); l1_mma(sa, acc); SB();
        l1_load<6>(sa, cur, w); SB(); l1_mma(sb, acc); SB();
        l1_load<7>(sb, cur, w); SB(); l1_mma(sa, acc); SB();
        l1_load<0>(sa, nxt, wn); SB(); l1_mma(sb, acc); SB();
        cur = nxt;
        w = wn;
    }
}

// ------------------------------------------------------------------ 4x4 stage, 4 waves x 4 patches
// M rows: tile mt = patches 2mt, 2mt+1; row i -> patch 2mt + (i>>4), pixel i & 15.
// Wave `wave` accumulates output channels 32*wave .. 32*wave+31 for both tiles.
// Input: CIN channels of WIN x WIN pixels with channel stride CS in every patch slab.
// Same pipeline, stage = 2 packed k-groups (8 MFMA steps x 2 tiles = 16 MFMAs).
struct L2Tap {
    const float* s0;    // tile 0 source (patch i>>4), halo -> pixel 0
    const float* s1;    // tile 1 source (patch 2 + (i>>4))
    bool ok;
};

struct L2Stage {
    float a[2][8];
    bool ok;
    float4 b[2];
};

template <int WIN, int STRIDE, int KS>
__device__ __forceinline__ L2Tap l2_tap(int tap, const float* S0, int oy, int ox) {
    constexpr int PAD = KS / 2;
    const int ky = tap / KS, kx = tap - ky * KS;
    const int iy = oy * STRIDE + ky - PAD, ix = ox * STRIDE + kx - PAD;
    L2Tap d;
    d.ok = (unsigned)iy < (unsigned)WIN && (unsigned)ix < (unsigned)WIN;
    d.s0 = S0 + (d.ok ? iy * WIN + ix : 0);
    d.s1 = d.s0 + 2 * SLAB;
    return d;
}

template <int CS, int C2>
__device__ __forceinline__ void l2_load(L2Stage& st, const L2Tap& d, const float4* w) {
    st.ok = d.ok;
    st.b[0] = w[(2 * C2) * 64];
    st.b[1] = w[(2 * C2 + 1) * 64];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        st.a[0][j] = d.s0[(C2 * 16 + 2 * j) * CS];
        st.a[1][j] = d.s1[(C2 * 16 + 2 * j) * CS];
    }
}

__device__ __forceinline__ void l2_mma(const L2Stage& st, f32x16 (&acc)[2]) {
    const float bb[8] = {st.b[0].x, st.b[0].y, st.b[0].z, st.b[0].w, st.b[1].x, st.b[1].y, st.b[1].z, st.b[1].w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float a0 = st.ok ? st.a[0][j] : 0.0f;
        const float a1 = st.ok ? st.a[1][j] : 0.0f;
        acc[0] = MFMA(a0, bb[j], acc[0]);
        acc[1] = MFMA(a1, bb[j], acc[1]);
    }
}

template <int CIN, int WIN, int CS, int STRIDE, int KS>
__device__ __forceinline__ void conv_l2(const float* __restrict__ wp, const float* lds, f32x16 (&acc)[2], int lane,
                                        int wave) {
    constexpr int KGS = KS * KS * CIN / 8, PER_TAP = CIN / 16, TAPS = KS * KS;
    static_assert(PER_TAP == 4 || PER_TAP == 8, "stage schedule is written for 64 or 128 input channels");
    const int i = lane & 31, half = lane >> 5;
    const int pix = i & 15, oy = pix >> 2, ox = pix & 3;
    const float* S0 = lds + (i >> 4) * SLAB + half * CS;
    const float4* w = reinterpret_cast<const float4*>(wp) + (size_t)wave * KGS * 64 + lane;   // +2*PER_TAP groups per tap
    zero(acc[0]); zero(acc[1]);
    L2Tap cur = l2_tap<WIN, STRIDE, KS>(0, S0, oy, ox);
    L2Stage sa, sb;
    l2_load<CS, 0>(sa, cur, w);
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
        const L2Tap nxt = l2_tap<WIN, STRIDE, KS>(tap < TAPS - 1 ? tap + 1 : TAPS - 1, S0, oy, ox);
        const float4* wn = tap < TAPS - 1 ? w + 2 * PER_TAP * 64 : w;
        l2_load<CS, 1>(sb, cur, w); SB(); l2_mma(sa, acc); SB();
        l2_load<CS, 2>(sa, cur, w); SB(); l2_mma(sb, acc); SB();
        l2_load<CS, 3>(sb, cur, w); SB(); l2_mma(sa, acc); SB();
        if (PER_TAP == 8) {
            l2_load<CS, 4>(sa, cur, w); SB(); l2_mma(sb, acc); SB();
            l2_load<CS, 5>(sb, cur, w); SB(); l2_mma(sa, acc); SB();
            l2_load<CS, 6>(sa, cur, w); SB(); l2_mma(sb, acc); SB();
            l2_load<CS, 7>(sb, cur, w); SB(); l2_mma(sa, acc); SB();
        }
        l2_load<CS, 0>(sa, nxt, wn); SB(); l2_mma(sb, acc); SB();
        cur = nxt;
        w = wn;
    }
}

// write the wave's 2 tiles (C layout) into the slabs in 4x4-stage layout [c][pix] (stride CS2)
__device__ __forceinline__ void store_l2(float* lds, const f32x16 (&v)[2], int lane, int wave) {
    const int i = lane & 31, half = lane >> 5;
    const int n = 32 * wave + i;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float* d = lds + (2 * mt + (q >> 1)) * SLAB + n * CS2 + 8 * (q & 1) + 4 * half;
            *reinterpret_cast<float4*>(d) = make_float4(v[mt][4 * q], v[mt][4 * q + 1], v[mt][4 * q + 2], v[mt][4 * q + 3]);
        }
}

__global__ __launch_bounds__(256, 2) void fused_trunk_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // 4 slabs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31;
    const long long p_first = (long long)blockIdx.x * 4;
    long long pi = p_first + wave;
    if (pi >= a.n) pi = a.n - 1;                                          // tail: recompute a valid patch, store nothing
    float* S = lds + wave * SLAB;

    // ---- input patch -> slab (coalesced 16 B loads)
    const float4* src = reinterpret_cast<const float4*>(a.patches + (size_t)pi * 1024);
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<float4*>(S)[k * 64 + lane] = src[k * 64 + lane];
    wave_fence();

    // ---- stem + pool: result in registers = identity of block 1
    f32x16 idn[2][2], acc[2][2];
    stem_pool(a, S, idn, lane);
    wave_fence();                                                      // the input is dead
    store_l1(S, idn, lane);
    wave_fence();

    // ---- layer1: two BasicBlocks at 8x8, wave = patch
#pragma unroll 1
    for (int blk = 0; blk < 2; ++blk) {
        // conv1 -> BN -> ReLU, written over its own input (identity is in registers)
        conv_l1(a.w[2 * blk], S, acc, lane);
        {
            const float* al = a.al[2 * blk];
            const float* sh = a.sh[2 * blk];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float A = al[nt * 32 + i], B = sh[nt * 32 + i];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = __builtin_fmaf(acc[mt][nt][r], A, B);
                        acc[mt][nt][r] = v > 0.0f ? v : 0.0f;
                    }
            }
        }
        wave_fence();
        store_l1(S, acc, lane);
        wave_fence();
        // conv2 -> BN -> += identity -> ReLU
        conv_l1(a.w[2 * blk + 1], S, acc, lane);
        {
            const float* al = a.al[2 * blk + 1];
            const float* sh = a.sh[2 * blk + 1];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float A = al[nt * 32 + i], B = sh[nt * 32 + i];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = __builtin_fmaf(acc[mt][nt][r], A, B);
                        v = v + idn[mt][nt][r];
                        idn[mt][nt][r] = v > 0.0f ? v : 0.0f;
                    }
            }
        }
        wave_fence();
        store_l1(S, idn, lane);
        __syncthreads();                                                  // layer2 reads all four slabs
    }

    // ---- layer2 block 0: conv3x3/2 (64->128) and the 1x1/2 projection read the 8x8 stage
    f32x16 t2[2], id2[2];
    const int n2 = 32 * wave + i;
    conv_l2<64, 8, CS1, 2, 3>(a.w[4], lds, t2, lane, wave);
    conv_l2<64, 8, CS1, 2, 1>(a.w_down, lds, id2, lane, wave);
    {
        const float A = a.al[4][n2], B = a.sh[4][n2], Ad = a.a_down[n2], Bd = a.s_down[n2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = __builtin_fmaf(t2[mt][r], A, B);
                t2[mt][r] = v > 0.0f ? v : 0.0f;
                id2[mt][r] = __builtin_fmaf(id2[mt][r], Ad, Bd);
            }
    }
    __syncthreads();
    store_l2(lds, t2, lane, wave);
    __syncthreads();
    // conv2 of block 0, then block 1 (conv1, conv2), all 128->128 at 4x4
#pragma unroll 1
    for (int cv = 5; cv < 8; ++cv) {
        conv_l2<128, 4, CS2, 1, 3>(a.w[cv], lds, t2, lane, wave);
        const float A = a.al[cv][n2], B = a.sh[cv][n2];
        const bool plain = (cv == 6);                                     // block 1 conv1: BN + ReLU only
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = __builtin_fmaf(t2[mt][r], A, B);
                if (!plain) v = v + id2[mt][r];
                v = v > 0.0f ? v : 0.0f;
                t2[mt][r] = v;
                if (!plain) id2[mt][r] = v;
            }
        __syncthreads();
        store_l2(lds, t2, lane, wave);
        __syncthreads();
    }

    // ---- global average pool over the 16 pixels, sequential order
    for (int o = threadIdx.x; o < 4 * 128; o += 256) {
        const int pl = o >> 7, n = o & 127;
        const float* s = lds + pl * SLAB + n * CS2;
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum = sum + s[k];
        if (p_first + pl < a.n) a.emb[(size_t)(p_first + pl) * 128 + n] = sum / 16.0f;
    }
}

static bool is_conv(const ipsx_conv& c, int ci, int co, int k, int s, int p) {
    return c.c_in == ci && c.c_out == co && c.kh == k && c.kw == k && c.stride == s && c.pad == p && c.w_packed &&
           c.alpha && c.shift;
}

bool fused_trunk_supported(const ipsx_trunk* t) {
    if (!t || t->c_in != 1 || t->h != 32 || t->w != 32 || t->n_block != 4 || !t->blocks) return false;
    if (!is_conv(t->stem, 1, 64, 7, 2, 3)) return false;
    const ipsx_block* b = t->blocks;
    for (int k = 0; k < 4; ++k)
        if (b[k].n_conv != 2) return false;
    for (int k = 0; k < 2; ++k)
        if (b[k].has_down || !is_conv(b[k].conv[0], 64, 64, 3, 1, 1) || !is_conv(b[k].conv[1], 64, 64, 3, 1, 1)) return false;
    if (!b[2].has_down || !is_conv(b[2].down, 64, 128, 1, 2, 0) || !is_conv(b[2].conv[0], 64, 128, 3, 2, 1) ||
        !is_conv(b[2].conv[1], 128, 128, 3, 1, 1))
        return false;
    if (b[3].has_down || !is_conv(b[3].conv[0], 128, 128, 3, 1, 1) || !is_conv(b[3].conv[1], 128, 128, 3, 1, 1)) return false;
    const char* off = getenv("IPSX_NO_FUSED");
    return !(off && off[0] == '1');
}

int fused_trunk_encode(const ipsx_trunk* t, const float* patches, int64_t n, float* emb, hipStream_t s) {
    FusedArgs a;
    a.patches = patches; a.emb = emb; a.n = n;
    a.w_stem = t->stem.w_packed; a.a_stem = t->stem.alpha; a.s_stem = t->stem.shift;
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 2; ++j) {
            a.w[2 * k + j] = t->blocks[k].conv[j].w_packed;
            a.al[2 * k + j] = t->blocks[k].conv[j].alpha;
            a.sh[2 * k + j] = t->blocks[k].conv[j].shift;
        }
    a.w_down = t->blocks[2].down.w_packed; a.a_down = t->blocks[2].down.alpha; a.s_down = t->blocks[2].down.shift;
    const size_t lds = (size_t)4 * SLAB * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_trunk_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    fused_trunk_kernel<<<dim3((unsigned)cdiv(n, 4)), dim3(256), lds, s>>>(a);
    return launched("fused_trunk");
}

}  // namespace ipsx
