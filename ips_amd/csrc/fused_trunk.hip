// fused_trunk.hip - the whole patch encoder of the Megapixel-MNIST configuration in ONE
// kernel, activations resident in LDS.
//
// Trunk (reference architecture/ips_net.py:17-52 with config/mnist_config.yml, 32-px
// patches): conv7x7/2(1->64)+BN+ReLU -> maxpool3x3/2 -> 2 x BasicBlock(64) @8x8 ->
// BasicBlock(64->128, /2, 1x1 projection) -> BasicBlock(128) @4x4 -> global average pool.
// 18,628,608 MAC per patch; 4 KiB in, 512 B out: arithmetic intensity ~8 kFLOP/B, so the
// bound is the fp32 matrix pipe (157 TFLOP/s), not HBM.
//
// One workgroup = 4 wavefronts = 4 patches, 69 KiB of LDS (one 17.3 KiB slab per patch), two
// workgroups per CU.  Every contraction is v_mfma_f32_32x32x2_f32 in the contract's k order, so
// the embeddings are bit-identical to the layer-by-layer kernels of conv.hip and to the oracle.
//
// What the instruction stream is shaped by (measured, tools/ubench): for the fp32 MFMA every
// OTHER instruction a wave issues costs matrix-pipe time (~8-10 cycles each, wherever it is
// placed), and a VALU write to a register that an MFMA issued just before reads as SrcA/B stalls
// for that MFMA.  So the kernel minimises instructions per MFMA:
//   * the LDS image of a stage is PIXEL-major, [pix][channel] with 4 floats of row padding, so a
//     lane fetches the 4 consecutive k of its half of an 8-group with ONE ds_read_b128
//     (that is why the contract's k order inside a group is 0,4,1,5,2,6,3,7);
//   * one extra all-zero pixel row per slab: halo lanes point at it - no select anywhere;
//   * weights come pre-packed from L2, 16 B per lane per 4 k-steps, through a 4-slot register
//     ring refilled two stages (2 x 1024 pipe cycles) ahead; LDS operands one stage ahead;
//     sched_barrier pins loads-then-16-MFMAs per stage.
//
//   stem      wave = patch.  The zero-padded 38x38 input sits in the slab; 8 tiles of 2 output rows
//             (32 px) x 64 channels; BN+ReLU in registers; the 3x3/2 max-pool is done ON the
//             accumulators (one lane-half exchange of column maxima), and its output lands in
//             exactly the register layout of the 8x8 stage's MFMA C tile - the first block's identity.
//   layer1    wave = patch: 64 px x 64 ch = 2x2 accumulators, no workgroup barriers (the slab is
//             wave-private).  conv1 output overwrites the slab in place (the identity lives in registers).
//   layer2    the 4 waves share the 4 patches: M = 4 x 16 px = 2 tiles, wave w owns output
//             channels 32w..32w+31, so each weight is fetched once per workgroup.
//   avgpool   sequential 16-term sums from the slab (the oracle's order).
//
// No global-memory round trips between layers: HBM traffic is the 4 KiB patch, the 512 B
// embedding and the L2-resident 2.7 MB of weights.

#include <algorithm>

#include "ipsx_common.h"
#include "ipsx_math.h"

namespace ipsx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PS1 = 68;              // floats per pixel row of the 8x8 stage: 64 channels + 4 pad
constexpr int ZP1 = 64;              // index of its all-zero pixel row
constexpr int PS2 = 132;             // 4x4 stage: 128 channels + 4 pad
constexpr int ZP2 = 16;
constexpr int SLAB = (ZP1 + 1) * PS1;   // floats per patch slab (17,680 B); >= (ZP2+1)*PS2 and >= 38*38

struct FusedArgs {
    const float* patches;
    float* emb;
    long long n;
    const int* index;        // optional: patch j of this launch is patches[index[j]] (blank-patch dedup)
    const int* count;        // optional: device-side number of valid entries of index (<= n)
    const float *w_stem, *a_stem, *s_stem;
    const float *w[8], *al[8], *sh[8];       // l1.0.c1 l1.0.c2 l1.1.c1 l1.1.c2 l2.0.c1 l2.0.c2 l2.1.c1 l2.1.c2
    const float *w_down, *a_down, *s_down;
    const void *wh[8], *wh_down, *wh_stem;   // bf16 operand streams of the same convolutions (precision 1 and 2)
    int in_dtype;            // storage type of `patches`: 0 float32, 1 bfloat16, 2 float16 (split trunks only)
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// bf16 matrix pipe, used by the split trunks (fused_trunk_split.h) and their stem
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA16(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
// plane pairs of the split product in issue order; 3 planes: (w.lo, x.hi) (w.hi, x.lo) (w.mid, x.mid) (w.mid, x.hi)
// (w.hi, x.mid) (w.hi, x.hi) - the six significant ones, small terms first; 1 plane: the single bf16 product
__device__ __forceinline__ constexpr int pair_w(int pl, int q) { return pl == 3 ? (q == 0 ? 2 : (q == 2 || q == 3) ? 1 : 0) : 0; }
__device__ __forceinline__ constexpr int pair_x(int pl, int q) { return pl == 3 ? (q == 1 ? 2 : (q == 2 || q == 4) ? 1 : 0) : 0; }

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
// S: this wave's slab holding the ZERO-PADDED input as P[38][38] (image at rows/cols 3..34), so
// every tap address is base + immediate and needs no halo mask.  On return idn[mt][nt] holds
// the pooled 8x8x64 activation in MFMA C layout (lane = channel, rows = pixels).
//
// PL = 0: the exact path - fp32 image, v_mfma_f32_32x32x2_f32 in the contract's k order.
// PL = 1 / 3 (split trunks): the padded image is PL bf16 planes of [38][SPW] elements and the contraction runs on
// v_mfma_f32_32x32x16_bf16 with K laid out as k = 8 ky + kx (kx = 7 and ky = 7 carry zero weights): a lane half's
// 8 k of a K-step are then 8 CONSECUTIVE pixels of one image row - four 4-byte LDS reads - instead of 25 scalar taps,
// and the 49-tap contraction costs 4 K-steps x NP products x 32 cycles per tile against 25 x 64.
constexpr int PW = 38;               // padded input width
constexpr int SPW = 40;              // row pitch (elements) of the bf16 planes of the padded input
constexpr int SPLANE = PW * SPW * 2; // bytes per plane

__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// v_permlane32_swap: lo_keeps = (a of this lane | b of the lane 32 below), hi_keeps = (a of the lane 32 above | b of this
// lane), for the lower | upper lane half - the two cross-half exchanges of the pool in ONE instruction, no LDS round trip
__device__ __forceinline__ void half_swap(float a, float b, float& lo_keeps, float& hi_keeps) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    lo_keeps = __uint_as_float(r[0]);
    hi_keeps = __uint_as_float(r[1]);
}

template <int PL>
__device__ __forceinline__ void stem_pool(const FusedArgs& a, const void* Sv, f32x16 (&idn)[2][2], int lane) {
    constexpr int PLN = PL > 0 ? PL : 1, NP = PL == 3 ? 6 : 1;
    const float* S = reinterpret_cast<const float*>(Sv);
    const char* Sp = reinterpret_cast<const char*>(Sv);
    const int i = lane & 31, half = lane >> 5;
    const int ox = i & 15;
    float bw0[28], bw1[28];                                                    // PL = 0: weights of the 28 k-steps
    uint4 wq[2][4][PLN];                                                       // PL > 0: [n-tile][K-step][plane]
    if constexpr (PL == 0) {
        const float4* wp0 = reinterpret_cast<const float4*>(a.w_stem) + lane;  // n-tile 0, 7 k-groups
        const float4* wp1 = wp0 + 7 * 64;
#pragma unroll
        for (int kg = 0; kg < 7; ++kg) {
            const float4 v0 = wp0[kg * 64], v1 = wp1[kg * 64];
            bw0[4 * kg] = v0.x; bw0[4 * kg + 1] = v0.y; bw0[4 * kg + 2] = v0.z; bw0[4 * kg + 3] = v0.w;
            bw1[4 * kg] = v1.x; bw1[4 * kg + 1] = v1.y; bw1[4 * kg + 2] = v1.z; bw1[4 * kg + 3] = v1.w;
        }
    } else {
        const char* wb = reinterpret_cast<const char*>(a.wh_stem) + lane * 16;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int pl = 0; pl < PLN; ++pl)
                    wq[nt][ks][pl] = *reinterpret_cast<const uint4*>(wb + ((nt * 4 + ks) * PLN + pl) * 1024);
    }
    const float al0 = a.a_stem[i], sh0 = a.s_stem[i], al1 = a.a_stem[32 + i], sh1 = a.s_stem[32 + i];

    // previous stem row (2t-1) of this lane's 8 own columns, per n-tile; row -1 is padding (-inf)
    float prev[2][8];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int x = 0; x < 8; ++x) prev[nt][x] = -__builtin_huge_valf();

    // (the split trunks unroll the eight row pairs: with t a constant the pooled row lands in its registers directly - as a
    //  loop the placement below is a chain of uniform branches whose joins cost ~260 register moves per row pair, 8 k
    //  cycles per patch: nothing against the fp32 trunk's 600 k, a third of the bf16 trunk's stem)
    constexpr int UNROLL_T = PL == 0 ? 1 : 8;
#pragma unroll UNROLL_T
    for (int t = 0; t < 8; ++t) {
        // lane's output pixel: row 2t + (i>>4), col ox.  Step (kg, j) of the contract feeds
        // k = 8kg + 4*half + j -> tap (k/7, k%7): the upper half's tap is 4 columns right of the
        // lower half's, or, when that leaves the 7-wide row (kx0 >= 3), 3 columns left one row down.
        const int oy = 2 * t + (i >> 4);
        f32x16 acc0, acc1;
        if constexpr (PL == 0) {
            const float* base = S + (2 * oy) * PW + 2 * ox;
            const float* baseN = base + half * 4;
            const float* baseW = base + half * (PW - 3);
            float av[25];
#pragma unroll
            for (int st = 0; st < 25; ++st) {              // steps 0..23: k-groups 0..5; step 24: k = 48 (+ padding)
                const int k0 = 8 * (st >> 2) + (st & 3), ky0 = k0 / 7, kx0 = k0 % 7;
                av[st] = (kx0 <= 2) ? baseN[ky0 * PW + kx0] : baseW[ky0 * PW + kx0];
            }
            av[24] = half ? 0.0f : av[24];                 // k = 52 does not exist (zero weight): feed a clean 0
            __builtin_amdgcn_sched_barrier(0);             // all 25 LDS reads in flight before the MFMAs
            zero(acc0); zero(acc1);
#pragma unroll
            for (int st = 0; st < 25; ++st) {
                acc0 = MFMA(av[st], bw0[st], acc0);
                acc1 = MFMA(av[st], bw1[st], acc1);
            }
        } else {
            // K-step ks, half h: image row 2 oy + 2 ks + h, columns 2 ox .. 2 ox + 7 (4-byte aligned)
            uint4 aq[4][PLN];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int pl = 0; pl < PLN; ++pl) {
                    const unsigned* p = reinterpret_cast<const unsigned*>(Sp + pl * SPLANE + ((2 * oy + 2 * ks + half) * SPW + 2 * ox) * 2);
                    aq[ks][pl] = make_uint4(p[0], p[1], p[2], p[3]);
                }
            __builtin_amdgcn_sched_barrier(0);
            zero(acc0); zero(acc1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    acc0 = MFMA16(aq[ks][pair_x(PL, q)], wq[0][ks][pair_w(PL, q)], acc0);
                    acc1 = MFMA16(aq[ks][pair_x(PL, q)], wq[1][ks][pair_w(PL, q)], acc1);
                }
        }
        // BN + ReLU; column maxima over stem rows {2t-1, 2t, 2t+1} for the 8 columns this lane holds:
        // C reg r -> tile row r>>3, column (r&3) + 8*((r>>2)&1) + 4*half.
        float pooled[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = nt ? __builtin_fmaf(acc1[r], al1, sh1) : __builtin_fmaf(acc0[r], al0, sh0);
                v[r] = x > 0.0f ? x : 0.0f;
            }
            float own[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                own[j] = max3(prev[nt][j], v[j], v[8 + j]);                 // columns 4*half + j
                own[4 + j] = max3(prev[nt][4 + j], v[4 + j], v[12 + j]);    // columns 8 + 4*half + j
                prev[nt][j] = v[8 + j];
                prev[nt][4 + j] = v[12 + j];
            }
            // pooled columns 4*half + q need stem columns 8*half - 1 .. 8*half + 7 (oth = the other lane half's own):
            //   half 0: [-inf, own0..3 (cols 0-3), oth0..3 (cols 4-7)]
            //   half 1: [own3 (col 7), oth4..7 (cols 8-11), own4..7 (cols 12-15)]
            float L[9];
            L[0] = half ? own[3] : -__builtin_huge_valf();
#pragma unroll
            for (int j = 0; j < 4; ++j) half_swap(own[j], own[4 + j], L[1 + j], L[5 + j]);
#pragma unroll
            for (int q = 0; q < 4; ++q) pooled[nt][q] = max3(L[2 * q], L[2 * q + 1], L[2 * q + 2]);
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

// write a 64px x 64ch wave tile (C layout: lane = channel, regs = pixels) into the slab as [pix][c]
__device__ __forceinline__ void store_l1(float* S, const f32x16 (&v)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                S[pix * PS1 + nt * 32 + i] = v[mt][nt][r];
            }
}

#define SB() __builtin_amdgcn_sched_barrier(0)

// ------------------------------------------------------------------ 8x8 stage, wave = patch
// acc = conv3x3(S) over K = 9*64.  wp: packed weights (2 n-tiles x 72 k-groups).
// Stage = one k-group (8 k = 4 MFMA steps x 2x2 tiles = 16 MFMAs = 1024 matrix-pipe cycles).
struct L1Tap {
    const float* s0;    // this lane's source pixel row (+ 4*half), m-tile 0; halo -> zero row
    const float* s1;    // m-tile 1
};

struct L1Stage {
    float4 a0, a1;      // 4 consecutive k of this lane's half, per m-tile
};

__device__ __forceinline__ L1Tap l1_tap(int tap, const float* S, int i, int half) {
    const int t3 = tap / 3;
    const int dy = t3 - 1, dx = tap - 3 * t3 - 1;
    const int x = i & 7, y0 = i >> 3;
    const bool okx = (unsigned)(x + dx) < 8u;
    const bool ok0 = okx && (unsigned)(y0 + dy) < 8u;
    const bool ok1 = okx && (unsigned)(y0 + 4 + dy) < 8u;
    const int p0 = i + dy * 8 + dx;
    L1Tap d;
    d.s0 = S + (ok0 ? p0 : ZP1) * PS1 + 4 * half;
    d.s1 = S + (ok1 ? p0 + 32 : ZP1) * PS1 + 4 * half;
    return d;
}

template <int CG>
__device__ __forceinline__ void l1_load(L1Stage& st, const L1Tap& d) {
    st.a0 = *reinterpret_cast<const float4*>(d.s0 + CG * 8);
    st.a1 = *reinterpret_cast<const float4*>(d.s1 + CG * 8);
}

// weights of k-group g (clamped: the tail prefetches re-read the last group) for both n-tiles.
// wb is wave-uniform (scalar base + scalar group offset), loff = lane*16 the only vector part.
__device__ __forceinline__ void l1_loadb(float4 (&b)[2], const char* wb, unsigned loff, int g) {
    g = g < 72 ? g : 71;
    const char* p = wb + (size_t)g * 1024;
    b[0] = *reinterpret_cast<const float4*>(p + loff);
    b[1] = *reinterpret_cast<const float4*>(p + 72 * 1024 + loff);
}

__device__ __forceinline__ void l1_mma(const L1Stage& st, const float4 (&b)[2], f32x16 (&acc)[2][2]) {
    const float a0[4] = {st.a0.x, st.a0.y, st.a0.z, st.a0.w}, a1[4] = {st.a1.x, st.a1.y, st.a1.z, st.a1.w};
    const float bb0[4] = {b[0].x, b[0].y, b[0].z, b[0].w};
    const float bb1[4] = {b[1].x, b[1].y, b[1].z, b[1].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc[0][0] = MFMA(a0[j], bb0[j], acc[0][0]);
        acc[0][1] = MFMA(a0[j], bb1[j], acc[0][1]);
        acc[1][0] = MFMA(a1[j], bb0[j], acc[1][0]);
        acc[1][1] = MFMA(a1[j], bb1[j], acc[1][1]);
    }
}

// IPSX_SPREAD = 1 (default): the loads of the next stage are spread between the 16 MFMAs of this one with
// sched_group_barrier instead of standing in front of them (IPSX_SPREAD = 0).  Measured on the headline workload
// (ms per step): in front 11.59; early (2 MFMA between loads, then 8) 11.59; even (4,1,4,1,...) 11.27; the patterns
// below - 3 (8x8 stage) / 2 (4x4 stage) MFMA between loads, 4 MFMA at the end - 11.12.  The MFMA order, i.e. the
// arithmetic, is the same in every variant.
#ifndef IPSX_SPREAD
#define IPSX_SPREAD 1
#endif
#if IPSX_SPREAD
#define SG_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define SG_LDS(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define SG_VMEM(n) __builtin_amdgcn_sched_group_barrier(0x020, n, 0)
#define L1_PRE()
#define L1_POST() \
    SG_MFMA(3); SG_LDS(1); SG_MFMA(3); SG_VMEM(1); SG_MFMA(3); SG_LDS(1); SG_MFMA(3); SG_VMEM(1); SG_MFMA(4); SB();
#define L2_PRE()
#define L2_POST()                                                                                         \
    SG_MFMA(2); SG_LDS(1); SG_MFMA(2); SG_LDS(1); SG_MFMA(2); SG_VMEM(1); SG_MFMA(2); SG_LDS(1); SG_MFMA(2); \
    SG_LDS(1); SG_MFMA(2); SG_VMEM(1); SG_MFMA(4); SB();
#else
#define L1_PRE() SB()
#define L1_POST() SB()
#define L2_PRE() SB()
#define L2_POST() SB()
#endif

__device__ __forceinline__ void conv_l1(const float* __restrict__ wp, const float* S, f32x16 (&acc)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
    const char* w = reinterpret_cast<const char*>(wp);
    const unsigned lo = lane * 16;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) zero(acc[mt][nt]);
    L1Tap cur = l1_tap(0, S, i, half);
    L1Stage sa, sb;
    float4 b0[2], b1[2], b2[2], b3[2];       // weight ring: slot = stage & 3, refilled 2 stages ahead
    l1_loadb(b0, w, lo, 0);
    l1_loadb(b1, w, lo, 1);
    l1_load<0>(sa, cur);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const L1Tap nxt = l1_tap(tap < 8 ? tap + 1 : 8, S, i, half);
        const int g = tap * 8;
        l1_load<1>(sb, cur); l1_loadb(b2, w, lo, g + 2); L1_PRE(); l1_mma(sa, b0, acc); L1_POST();
        l1_load<2>(sa, cur); l1_loadb(b3, w, lo, g + 3); L1_PRE(); l1_mma(sb, b1, acc); L1_POST();
        l1_load<3>(sb, cur); l1_loadb(b0, w, lo, g + 4); L1_PRE(); l1_mma(sa, b2, acc); L1_POST();
        l1_load<4>(sa, cur); l1_loadb(b1, w, lo, g + 5); L1_PRE(); l1_mma(sb, b3, acc); L1_POST();
        l1_load<5>(sb, cur); l1_loadb(b2, w, lo, g + 6); L1_PRE(); l1_mma(sa, b0, acc); L1_POST();
        l1_load<6>(sa, cur); l1_loadb(b3, w, lo, g + 7); L1_PRE(); l1_mma(sb, b1, acc); L1_POST();
        l1_load<7>(sb, cur); l1_loadb(b0, w, lo, g + 8); L1_PRE(); l1_mma(sa, b2, acc); L1_POST();
        l1_load<0>(sa, nxt); l1_loadb(b1, w, lo, g + 9); L1_PRE(); l1_mma(sb, b3, acc); L1_POST();
        cur = nxt;
    }
}

// ------------------------------------------------------------------ 4x4 stage, 4 waves x 4 patches
// M rows: tile mt = patches 2mt, 2mt+1; row i -> patch 2mt + (i>>4), pixel i & 15.
// Wave `wave` accumulates output channels 32*wave .. 32*wave+31 for both tiles.
// Input: CIN channels of WIN x WIN pixels, pixel-major with row stride PS and zero row ZP.
// Stage = 2 packed k-groups (16 k = 8 MFMA steps x 2 tiles = 16 MFMAs).
struct L2Tap {
    const float* s0;    // tile 0 source pixel row (+ 4*half), halo -> zero row
    const float* s1;    // tile 1 (two slabs further)
};

struct L2Stage {
    float4 a[2][2];     // [tile][k-group of the pair]
};

template <int WIN, int PS, int ZP, int STRIDE, int KS>
__device__ __forceinline__ L2Tap l2_tap(int tap, const float* S0, int oy, int ox) {
    constexpr int PAD = KS / 2;
    const int ky = tap / KS, kx = tap - ky * KS;
    const int iy = oy * STRIDE + ky - PAD, ix = ox * STRIDE + kx - PAD;
    const bool ok = (unsigned)iy < (unsigned)WIN && (unsigned)ix < (unsigned)WIN;
    L2Tap d;
    d.s0 = S0 + (ok ? iy * WIN + ix : ZP) * PS;
    d.s1 = d.s0 + 2 * SLAB;
    return d;
}

template <int C2>
__device__ __forceinline__ void l2_load(L2Stage& st, const L2Tap& d) {
    st.a[0][0] = *reinterpret_cast<const float4*>(d.s0 + C2 * 16);
    st.a[0][1] = *reinterpret_cast<const float4*>(d.s0 + C2 * 16 + 8);
    st.a[1][0] = *reinterpret_cast<const float4*>(d.s1 + C2 * 16);
    st.a[1][1] = *reinterpret_cast<const float4*>(d.s1 + C2 * 16 + 8);
}

template <int G2>
__device__ __forceinline__ void l2_loadb(float4 (&b)[2], const char* wb, unsigned loff, int g2) {
    g2 = g2 < G2 ? g2 : G2 - 1;
    const char* p = wb + (size_t)g2 * 2048;
    b[0] = *reinterpret_cast<const float4*>(p + loff);
    b[1] = *reinterpret_cast<const float4*>(p + 1024 + loff);
}

__device__ __forceinline__ void l2_mma(const L2Stage& st, const float4 (&b)[2], f32x16 (&acc)[2]) {
    const float a0[8] = {st.a[0][0].x, st.a[0][0].y, st.a[0][0].z, st.a[0][0].w,
                         st.a[0][1].x, st.a[0][1].y, st.a[0][1].z, st.a[0][1].w};
    const float a1[8] = {st.a[1][0].x, st.a[1][0].y, st.a[1][0].z, st.a[1][0].w,
                         st.a[1][1].x, st.a[1][1].y, st.a[1][1].z, st.a[1][1].w};
    const float bb[8] = {b[0].x, b[0].y, b[0].z, b[0].w, b[1].x, b[1].y, b[1].z, b[1].w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc[0] = MFMA(a0[j], bb[j], acc[0]);
        acc[1] = MFMA(a1[j], bb[j], acc[1]);
    }
}

template <int CIN, int WIN, int PS, int ZP, int STRIDE, int KS>
__device__ __forceinline__ void conv_l2(const float* __restrict__ wp, const float* lds, f32x16 (&acc)[2], int lane,
                                        int wave) {
    constexpr int KGS = KS * KS * CIN / 8, G2 = KGS / 2, PER_TAP = CIN / 16, TAPS = KS * KS;
    static_assert(PER_TAP == 4 || PER_TAP == 8, "stage schedule is written for 64 or 128 input channels");
    const int i = lane & 31, half = lane >> 5;
    const int pix = i & 15, oy = pix >> 2, ox = pix & 3;
    const float* S0 = lds + (i >> 4) * SLAB + 4 * half;
    const char* w = reinterpret_cast<const char*>(wp) + (size_t)__builtin_amdgcn_readfirstlane(wave) * KGS * 1024;
    const unsigned lo = lane * 16;
    zero(acc[0]); zero(acc[1]);
    L2Tap cur = l2_tap<WIN, PS, ZP, STRIDE, KS>(0, S0, oy, ox);
    L2Stage sa, sb;
    float4 b0[2], b1[2], b2[2], b3[2];
    l2_loadb<G2>(b0, w, lo, 0);
    l2_loadb<G2>(b1, w, lo, 1);
    l2_load<0>(sa, cur);
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
        const L2Tap nxt = l2_tap<WIN, PS, ZP, STRIDE, KS>(tap < TAPS - 1 ? tap + 1 : TAPS - 1, S0, oy, ox);
        const int g = tap * PER_TAP;
        if (PER_TAP == 8) {
            l2_load<1>(sb, cur); l2_loadb<G2>(b2, w, lo, g + 2); L2_PRE(); l2_mma(sa, b0, acc); L2_POST();
            l2_load<2>(sa, cur); l2_loadb<G2>(b3, w, lo, g + 3); L2_PRE(); l2_mma(sb, b1, acc); L2_POST();
            l2_load<3>(sb, cur); l2_loadb<G2>(b0, w, lo, g + 4); L2_PRE(); l2_mma(sa, b2, acc); L2_POST();
            l2_load<4>(sa, cur); l2_loadb<G2>(b1, w, lo, g + 5); L2_PRE(); l2_mma(sb, b3, acc); L2_POST();
            l2_load<5>(sb, cur); l2_loadb<G2>(b2, w, lo, g + 6); L2_PRE(); l2_mma(sa, b0, acc); L2_POST();
            l2_load<6>(sa, cur); l2_loadb<G2>(b3, w, lo, g + 7); L2_PRE(); l2_mma(sb, b1, acc); L2_POST();
            l2_load<7>(sb, cur); l2_loadb<G2>(b0, w, lo, g + 8); L2_PRE(); l2_mma(sa, b2, acc); L2_POST();
            l2_load<0>(sa, nxt); l2_loadb<G2>(b1, w, lo, g + 9); L2_PRE(); l2_mma(sb, b3, acc); L2_POST();
        } else {
            l2_load<1>(sb, cur); l2_loadb<G2>(b2, w, lo, g + 2); L2_PRE(); l2_mma(sa, b0, acc); L2_POST();
            l2_load<2>(sa, cur); l2_loadb<G2>(b3, w, lo, g + 3); L2_PRE(); l2_mma(sb, b1, acc); L2_POST();
            l2_load<3>(sb, cur); l2_loadb<G2>(b0, w, lo, g + 4); L2_PRE(); l2_mma(sa, b2, acc); L2_POST();
            l2_load<0>(sa, nxt); l2_loadb<G2>(b1, w, lo, g + 5); L2_PRE(); l2_mma(sb, b3, acc); L2_POST();
        }
        cur = nxt;
    }
}

// write the wave's 2 tiles (C layout) into the slabs in 4x4-stage layout [pix][c] (row stride PS2)
__device__ __forceinline__ void store_l2(float* lds, const f32x16 (&v)[2], int lane, int wave) {
    const int i = lane & 31, half = lane >> 5;
    const int n = 32 * wave + i;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pl = 2 * mt + (r >> 3), pix = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;
            lds[pl * SLAB + pix * PS2 + n] = v[mt][r];
        }
}

// STAMP = true is the diagnostic build: every wavefront records s_memtime at its phase boundaries
// into a buffer of its own (never into an output); the product launches STAMP = false.
#define IPSX_STAMP(k)                                                                      \
    do {                                                                                   \
        if (STAMP) {                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                             \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                    \
            if (lane == 0) stamps[((size_t)blockIdx.x * 4 + wave) * 16 + (k)] = t_;        \
            __builtin_amdgcn_sched_barrier(0);                                             \
        }                                                                                  \
    } while (0)

// four patches p_first .. p_first + 3 by the workgroup's four wavefronts (the body of fused_trunk_kernel); KEEP: the four
// embeddings are also left in lds[0 .. 511] (behind a barrier) for a caller that goes on with them
// (fused_trunk_stream_kernel: the logits)
template <bool STAMP, bool KEEP>
__device__ __forceinline__ void trunk_quad_tile(const FusedArgs& a, long long p_first, long long n_valid, float* lds,
                                                unsigned long long* stamps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31;
    long long pi = p_first + wave;
    if (pi >= n_valid) pi = n_valid - 1;                                  // tail: recompute a valid patch, store nothing
    if (a.index) pi = a.index[pi];
    float* S = lds + wave * SLAB;
    IPSX_STAMP(0);

    // ---- input patch -> slab as a zero-padded 38x38 image (coalesced 16 B global loads)
    {
        const float4* src = reinterpret_cast<const float4*>(a.patches + (size_t)pi * 1024);
        float4 px[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) px[k] = src[k * 64 + lane];
        for (int z = lane; z < (PW * PW + 3) / 4; z += 64) reinterpret_cast<float4*>(S)[z] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z = lane; z < PS1; z += 64) S[ZP1 * PS1 + z] = 0.0f;          // zero pixel row of the 8x8 stage
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = (k * 64 + lane) * 4, y = e >> 5, x = e & 31;       // 4 pixels of row y starting at x
            float* d = S + (y + 3) * PW + x + 3;
            d[0] = px[k].x; d[1] = px[k].y; d[2] = px[k].z; d[3] = px[k].w;
        }
    }
    wave_fence();

    // ---- stem + pool: result in registers = identity of block 1
    f32x16 idn[2][2], acc[2][2];
    IPSX_STAMP(1);
    stem_pool<0>(a, S, idn, lane);
    IPSX_STAMP(2);
    wave_fence();                                                      // the input is dead
    store_l1(S, idn, lane);
    wave_fence();

    // ---- layer1: two BasicBlocks at 8x8, wave = patch
#pragma unroll 1
    for (int blk = 0; blk < 2; ++blk) {
        // conv1 -> BN -> ReLU, written over its own input (identity is in registers)
        conv_l1(a.w[2 * blk], S, acc, lane);
        IPSX_STAMP(3 + 4 * blk);
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
        IPSX_STAMP(4 + 4 * blk);
        // conv2 -> BN -> += identity -> ReLU
        conv_l1(a.w[2 * blk + 1], S, acc, lane);
        IPSX_STAMP(5 + 4 * blk);
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
        IPSX_STAMP(6 + 4 * blk);
    }

    // ---- layer2 block 0: conv3x3/2 (64->128) and the 1x1/2 projection read the 8x8 stage
    f32x16 t2[2], id2[2];
    const int n2 = 32 * wave + i;
    conv_l2<64, 8, PS1, ZP1, 2, 3>(a.w[4], lds, t2, lane, wave);
    conv_l2<64, 8, PS1, ZP1, 2, 1>(a.w_down, lds, id2, lane, wave);
    IPSX_STAMP(11);
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
    for (int z = lane; z < PS2; z += 64) lds[wave * SLAB + ZP2 * PS2 + z] = 0.0f;   // zero pixel row of the 4x4 stage
    __syncthreads();
    // conv2 of block 0, then block 1 (conv1, conv2), all 128->128 at 4x4
#pragma unroll 1
    for (int cv = 5; cv < 8; ++cv) {
        conv_l2<128, 4, PS2, ZP2, 1, 3>(a.w[cv], lds, t2, lane, wave);
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
        IPSX_STAMP(7 + cv);
    }

    // ---- global average pool over the 16 pixels, sequential order
    float keep[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int o = threadIdx.x + 256 * h;
        const int pl = o >> 7, n = o & 127;
        const float* s = lds + pl * SLAB + n;
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum = sum + s[k * PS2];
        keep[h] = sum / 16.0f;
        if (p_first + pl < n_valid) a.emb[(size_t)(p_first + pl) * 128 + n] = keep[h];
    }
    if (KEEP) {
        __syncthreads();                                                  // every sum has been read
        lds[threadIdx.x] = keep[0];
        lds[threadIdx.x + 256] = keep[1];
        __syncthreads();
    }
    IPSX_STAMP(15);
}

template <bool STAMP>
__global__ __launch_bounds__(256, 2) void fused_trunk_kernel(FusedArgs a, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // 4 slabs
    const long long p_first = (long long)blockIdx.x * 4;
    const long long n_valid = a.count ? (long long)*a.count : a.n;        // compacted launches read their length on device
    if (p_first >= n_valid) return;                                       // workgroup-uniform
    trunk_quad_tile<STAMP, false>(a, p_first, n_valid, lds, stamps);
}

#include "fused_trunk_split.h"
#include "fused_trunk_bf16v2.h"
#include "fused_trunk_bf16v3.h"
#include "fused_trunk_pair.h"

// ------------------------------------------------------------------ the stages as stand-alone convolutions (training step)
// The training step's forward and data-gradient convolutions on the maps of 32-px patches - 64 -> 64 at 8x8, 128 -> 128 at
// 4x4, and the two strided layers between them - are the very stages of the fused trunk: a patch's map in LDS (pixel-major
// with a zero halo row), operands by 16-byte LDS reads, weights streamed from L2 through the register ring.  Run layer
// by layer (the backward pass needs every layer's input), each still reads its input once from HBM and has the map in LDS
// for all nine taps, where the batch-tiled conv_nhwc_kernel fetches every activation once per tap from L2 (0.45 of the fp32
// MFMA peak at 1,024 patches; these: see DESIGN 5.7).  Plain convolutions (BatchNorm is its own autograd node in training).
struct LdsConvArgs {
    const float* x;        // (n, WIN, WIN, CIN) channels-last
    float* y;              // (n, WOUT, WOUT, COUT) channels-last
    const float* wp;       // ipsx_pack_conv_weight layout
    long long n;
    // BatchNorm batch statistics off the accumulators (training step; ipsx_conv2d_lds_nhwc_stats): per workgroup - a slab of
    // 4 patches' output rows - and output channel  sum (y - shift), sum (y - shift)^2  -> stats[blockIdx][0 | 1][COUT], what
    // bn_reduce_kernel<false> would compute in a pass of its own over y (csrc/bn_train.hip; the combination over the slabs
    // stays bn_finalize_kernel's, fp64 in slab order).  shift: a per-channel constant near the mean (the BatchNorm's running
    // mean), NULL = 0.  stats NULL: a plain convolution.
    float* stats;
    const float* shift;
};

// sum and sum of squares of (v - k) over this lane's 16 registers of one accumulator tile
__device__ __forceinline__ void tile_moments(const f32x16& v, float k, float& s1, float& s2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float d = v[r] - k;
        s1 = s1 + d;
        s2 = __builtin_fmaf(d, d, s2);
    }
}

// 64 -> 64, 3x3, stride 1, 8x8 maps: wave = patch (conv_l1)
__global__ __launch_bounds__(256, 2) void conv_lds_l1_kernel(LdsConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // 4 slabs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long p = (long long)blockIdx.x * 4 + wave;
    const bool live = p < a.n;
    if (!live && !a.stats) return;                                        // (no workgroup barrier without statistics)
    float* S = lds + wave * SLAB;
    const int i = lane & 31, half = lane >> 5;
    float s1[2] = {0.0f, 0.0f}, s2[2] = {0.0f, 0.0f};
    if (live) {
        const float4* src = reinterpret_cast<const float4*>(a.x + (size_t)p * 4096);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int e = (k * 64 + lane) * 4;                           // pixel e / 64, channel e % 64
            *reinterpret_cast<float4*>(S + (e >> 6) * PS1 + (e & 63)) = src[k * 64 + lane];
        }
        for (int z = lane; z < PS1; z += 64) S[ZP1 * PS1 + z] = 0.0f;
        wave_fence();
        f32x16 acc[2][2];
        conv_l1(a.wp, S, acc, lane);
        float* dst = a.y + (size_t)p * 4096;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dst[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + nt * 32 + i] = acc[mt][nt][r];
        if (a.stats) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float k = a.shift ? a.shift[nt * 32 + i] : 0.0f;
                tile_moments(acc[0][nt], k, s1[nt], s2[nt]);             // rows in a fixed order: deterministic
                tile_moments(acc[1][nt], k, s1[nt], s2[nt]);
            }
        }
    }
    if (a.stats) {
        // the lane halves hold different pixels of the same channel; then the four patches of the slab, in patch order
        wave_fence();                                                     // (this wave's reads of its slab are over)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            s1[nt] = s1[nt] + __shfl_xor(s1[nt], 32, 64);
            s2[nt] = s2[nt] + __shfl_xor(s2[nt], 32, 64);
            if (half == 0) { S[nt * 32 + i] = s1[nt]; S[64 + nt * 32 + i] = s2[nt]; }
        }
        __syncthreads();
        if (wave == 0) {
            float t1 = lds[lane], t2 = lds[64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) { t1 = t1 + lds[w * SLAB + lane]; t2 = t2 + lds[w * SLAB + 64 + lane]; }
            a.stats[(size_t)blockIdx.x * 128 + lane] = t1;
            a.stats[(size_t)blockIdx.x * 128 + 64 + lane] = t2;
        }
    }
}

// CIN -> 128 onto 4x4 maps (conv_l2): 4 patches per workgroup, wave w owns output channels 32 w .. 32 w + 31
template <int CIN, int WIN, int PS, int ZP, int STRIDE, int KS>
__global__ __launch_bounds__(256, 2) void conv_lds_l2_kernel(LdsConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // 4 slabs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long p_first = (long long)blockIdx.x * 4;
    constexpr int MAP = WIN * WIN * CIN;                                 // floats per patch
    for (int q = threadIdx.x; q < 4 * MAP / 4; q += 256) {               // the four maps, 16 bytes at a time
        const int pl = q / (MAP / 4), e = (q - pl * (MAP / 4)) * 4;      // patch, element inside its map: pixel e / CIN, channel e % CIN
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (p_first + pl < a.n) v = *reinterpret_cast<const float4*>(a.x + (size_t)(p_first + pl) * MAP + e);
        *reinterpret_cast<float4*>(lds + pl * SLAB + (e / CIN) * PS + (e % CIN)) = v;
    }
    for (int z = lane; z < PS; z += 64) lds[wave * SLAB + ZP * PS + z] = 0.0f;
    __syncthreads();
    f32x16 acc[2];
    conv_l2<CIN, WIN, PS, ZP, STRIDE, KS>(a.wp, lds, acc, lane, wave);
    const int i = lane & 31, half = lane >> 5, n = 32 * wave + i;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pl = 2 * mt + (r >> 3), pix = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;
            if (p_first + pl < a.n) a.y[((size_t)(p_first + pl) * 16 + pix) * 128 + n] = acc[mt][r];
        }
    if (a.stats) {
        // this wave owns channel n for all 64 rows of the slab (4 patches x 16 pixels): half of them on each lane half
        const float k = a.shift ? a.shift[n] : 0.0f;
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = (p_first + 2 * mt + (r >> 3) < a.n) ? acc[mt][r] - k : 0.0f;
                s1 = s1 + d;
                s2 = __builtin_fmaf(d, d, s2);
            }
        s1 = s1 + __shfl_xor(s1, 32, 64);
        s2 = s2 + __shfl_xor(s2, 32, 64);
        if (half == 0) {
            a.stats[(size_t)blockIdx.x * 256 + n] = s1;
            a.stats[(size_t)blockIdx.x * 256 + 128 + n] = s2;
        }
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

static int g_bf16_build = 0;      // diagnostic (ipsx_dbg_bf16_build): 0 the default build of the bf16 trunk (the third), 1 / 2 / 3 that build
static int g_pair_mode = 0;       // diagnostic (ipsx_dbg_fused_trunk_pair): 0 the rule below, 1 never, 2 every patch through the pair kernel

static int device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int v = 0;
        cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return cus[dev];
}

static int fused_launch(const ipsx_trunk* t, const float* patches, int64_t n, float* emb,
                        unsigned long long* stamps, hipStream_t s, const int* index = nullptr,
                        const int* count = nullptr) {
    FusedArgs a;
    a.patches = patches; a.emb = emb; a.n = n; a.index = index; a.count = count;
    a.in_dtype = t->patch_dtype;
    if (t->patch_dtype != 0 && !(t->precision == 1 || t->precision == 2))
        return fail(IPSX_EINVAL, "fused trunk: half-precision patch storage goes with precision 1 (bf16) or 2 (fp32x3)");
    if (t->patch_dtype < 0 || t->patch_dtype > 2) return fail(IPSX_EINVAL, "fused trunk: patch_dtype %d", t->patch_dtype);
    a.w_stem = t->stem.w_packed; a.a_stem = t->stem.alpha; a.s_stem = t->stem.shift;
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 2; ++j) {
            a.w[2 * k + j] = t->blocks[k].conv[j].w_packed;
            a.al[2 * k + j] = t->blocks[k].conv[j].alpha;
            a.sh[2 * k + j] = t->blocks[k].conv[j].shift;
        }
    a.w_down = t->blocks[2].down.w_packed; a.a_down = t->blocks[2].down.alpha; a.s_down = t->blocks[2].down.shift;
    bool bf16 = t->precision != 0;
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 2; ++j) {
            a.wh[2 * k + j] = t->blocks[k].conv[j].w_packed_bf16;
            bf16 = bf16 && a.wh[2 * k + j];
        }
    a.wh_down = t->blocks[2].down.w_packed_bf16;
    a.wh_stem = t->stem.w_packed_bf16;
    bf16 = bf16 && a.wh_down && a.wh_stem;
    if (t->precision != 0 && !bf16) return fail(IPSX_EINVAL, "fused trunk: precision %d needs w_packed_bf16 on the stem (ipsx_pack_stem_weight_split) and every block conv", t->precision);
    if (t->precision == 2 || t->precision == 1) {      // the split trunks on the bf16 matrix pipe (fused_trunk_split.h)
        const bool x3 = t->precision == 2;
        const size_t ldsx = (size_t)4 * (x3 ? XL<3>::SLAB : XL<1>::SLAB);
        static bool attr_split = false;
        if (!attr_split) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_trunk_x3_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 4 * XL<3>::SLAB);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_trunk_x3_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 4 * XL<3>::SLAB);
            attr_split = true;
        }
        const dim3 grid((unsigned)cdiv(n, 4)), block(256);
        // the bf16 trunk's builds (round 6): IPSX_BF16_BUILD = 1 first (fused_trunk_split.h), 2 second, 3 third (default)
        static const int bf16_env = [] { const char* e = getenv("IPSX_BF16_BUILD"); return e && e[0] >= '1' && e[0] <= '3' ? e[0] - '0' : 0; }();
        const int bf16_build = g_bf16_build ? g_bf16_build : (bf16_env ? bf16_env : 3);
        if (!x3 && bf16_build == 3) {
            // the third build (fused_trunk_bf16v3.h): eight patches per workgroup, the 4x4 stage once over all eight
            static bool attr_v3 = false;
            if (!attr_v3) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_trunk_bf16v3_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_trunk_bf16v3_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS);
                attr_v3 = true;
            }
            if (stamps) fused_trunk_bf16v3_kernel<true><<<dim3((unsigned)cdiv(n, 8)), block, V3_LDS, s>>>(a, stamps);
            else fused_trunk_bf16v3_kernel<false><<<dim3((unsigned)cdiv(n, 8)), block, V3_LDS, s>>>(a, nullptr);
            return launched("fused_trunk_bf16v3");
        }
        if (!x3 && bf16_build != 1) {
            static bool attr_v2 = false;
            if (!attr_v2) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_trunk_bf16v2_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_trunk_bf16v2_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
                attr_v2 = true;
            }
            // One workgroup per quad by default.  IPSX_BF16_PERSIST=1: two workgroups per unit take the launch's quads in
            // turn (no dispatch gap between quads, the next quad's pixels prefetched) - faster alone (+5 %), but a selection
            // loop that is to run BESIDE the launch needs a whole unit's registers (16 waves of 128) and finds none until the
            // launch is over: measured inside ips() 26.5 -> 22.2 M patches/s, so it is not the default
            static const bool persist = [] { const char* e = getenv("IPSX_BF16_PERSIST"); return e && e[0] == '1'; }();
            const dim3 pgrid((unsigned)(persist ? std::min<int64_t>(cdiv(n, 4), 2 * (int64_t)device_cus()) : cdiv(n, 4)));
            // (diagnostic: IPSX_BF16_ONE_WG=1 asks for more LDS than two workgroups get - ONE workgroup, one wave per SIMD)
            static const size_t lds_v2 = [] { const char* e = getenv("IPSX_BF16_ONE_WG"); return (size_t)(e && e[0] == '1' ? 100 * 1024 : V2_LDS); }();
            if (stamps) fused_trunk_bf16v2_kernel<true><<<pgrid, block, lds_v2, s>>>(a, stamps);
            else fused_trunk_bf16v2_kernel<false><<<pgrid, block, lds_v2, s>>>(a, nullptr);
            return launched("fused_trunk_bf16v2");
        }
        if (x3 && stamps) fused_trunk_x3_kernel<true><<<grid, block, ldsx, s>>>(a, stamps);
        else if (x3) fused_trunk_x3_kernel<false><<<grid, block, ldsx, s>>>(a, nullptr);
        else if (stamps) fused_trunk_bf16_kernel<true><<<grid, block, ldsx, s>>>(a, stamps);
        else fused_trunk_bf16_kernel<false><<<grid, block, ldsx, s>>>(a, nullptr);
        return launched(x3 ? "fused_trunk_x3" : "fused_trunk_bf16");
    }
    const size_t lds = (size_t)4 * SLAB * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_trunk_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_trunk_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int cus = device_cus();
    // What is left over after the whole rounds (8 patches per CU) goes to the two-wavefronts-per-patch kernel when that takes
    // fewer wavefront slots per SIMD: at most a quarter round (one workgroup per CU, one wavefront per SIMD, half a patch
    // each) or between a half and three quarters (three per SIMD) - fused_trunk_pair.h; a device-side count (blank-patch
    // dedup) leaves the launch as it is.
    int64_t rest = 0;
    if (!stamps && !count && g_pair_mode != 1) {
        const int64_t round = 8 * (int64_t)cus;
        rest = g_pair_mode == 2 ? n : n % round;
        // measured, patches -> ms: one wavefront per patch 1024 0.29, 2048 0.54; two per patch 512 0.15, 1024 0.28, 1536 0.43
        // (three workgroups per CU), 2048 0.57
        if (g_pair_mode != 2 && !(rest <= round / 4 || (rest > round / 2 && rest <= 3 * round / 4))) rest = 0;
    }
    const int64_t n_full = n - rest;
    a.n = n_full;
    if (stamps)
        fused_trunk_kernel<true><<<dim3((unsigned)cdiv(n, 4)), dim3(256), lds, s>>>(a, stamps);
    else if (n_full)
        fused_trunk_kernel<false><<<dim3((unsigned)cdiv(n_full, 4)), dim3(256), lds, s>>>(a, nullptr);
    if (rest) {
        a.n = rest;
        if (index) a.index = index + n_full;
        else a.patches = patches + (size_t)n_full * 1024;
        a.emb = emb + (size_t)n_full * 128;
        fused_trunk_pair_kernel<<<dim3((unsigned)cdiv(rest, 2)), dim3(256), lds / 2, s>>>(a);
    }
    return launched("fused_trunk");
}

// One image through the fused trunk as ONE persistent launch that feeds a resident selection loop (fused_trunk_stream_kernel,
// fused_trunk_pair.h).  workgroups <= 0: one per compute unit but the loop's and a few to spare.
int fused_trunk_stream(const ipsx_trunk* t, const float* patches, int64_t n, float* emb, const float* pos, const float* v_packed,
                       int r, float* logits, int32_t* ctl, int32_t* ready, int workgroups, int quad_pulls, hipStream_t s) {
    if (t->precision != 0 || t->patch_dtype != 0) return fail(IPSX_EINVAL, "trunk_stream: the exact fp32 trunk only");
    TrunkStreamArgs a;
    a.f.patches = patches; a.f.emb = emb; a.f.n = n; a.f.index = nullptr; a.f.count = nullptr; a.f.in_dtype = 0;
    a.f.w_stem = t->stem.w_packed; a.f.a_stem = t->stem.alpha; a.f.s_stem = t->stem.shift;
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 2; ++j) {
            a.f.w[2 * k + j] = t->blocks[k].conv[j].w_packed;
            a.f.al[2 * k + j] = t->blocks[k].conv[j].alpha;
            a.f.sh[2 * k + j] = t->blocks[k].conv[j].shift;
            a.f.wh[2 * k + j] = nullptr;
        }
    a.f.w_down = t->blocks[2].down.w_packed; a.f.a_down = t->blocks[2].down.alpha; a.f.s_down = t->blocks[2].down.shift;
    a.f.wh_down = nullptr; a.f.wh_stem = nullptr;
    a.pos = pos; a.vp = v_packed; a.R = r; a.logits = logits; a.ctl = ctl; a.ready = ready;
    a.n_pairs = (unsigned)cdiv(n, 2);
    // Four patches per pull run at the trunk's full rate (0.29 ms per pull at one workgroup per unit), two per pull at 0.84 of
    // it (0.155 ms): as many four-patch pulls per workgroup as leave at least one round of two-patch pulls for the rest.
    {
        const int wgs_ = workgroups > 0 ? workgroups : device_cus() - 1;
        const int64_t q = quad_pulls >= 0 ? quad_pulls : std::max<int64_t>(0, (n - 1) / (4 * (int64_t)wgs_));
        a.quad_pulls = (int)std::min<int64_t>(q, 1 << 20);
    }
    // one workgroup per compute unit (two patches at a time, one wavefront per SIMD: the shortest time from pull to
    // publication); the LDS request - more than half a unit's - keeps two of them off one unit and all of them off the loop's
    const size_t lds = 96 << 10;                           // (four slabs are 69 KB)
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_trunk_stream_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    // (more workgroups than free units: one that is placed late starts late or finds nothing left - with fewer, the units
    //  the dispatcher fails to use at once make a sixth round of 2,500 patches: 1.03 against 0.92 ms per image)
    const int wgs = workgroups > 0 ? workgroups : device_cus() + 8;
    fused_trunk_stream_kernel<<<dim3((unsigned)wgs), dim3(256), lds, s>>>(a);
    return launched("fused_trunk_stream");
}

int fused_trunk_encode(const ipsx_trunk* t, const float* patches, int64_t n, float* emb, hipStream_t s) {
    return fused_launch(t, patches, n, emb, nullptr, s);
}

int fused_trunk_encode_indexed(const ipsx_trunk* t, const float* patches, int64_t n_max, const int* index,
                               const int* count, float* emb, hipStream_t s) {
    return fused_launch(t, patches, n_max, emb, nullptr, s, index, count);
}

}  // namespace ipsx

// which LDS-resident stage kernel covers this plain convolution of channels-last maps (0: none)
static int lds_conv_kind(int c_in, int c_out, int k, int stride, int pad, int h, int w) {
    if (h != w) return 0;
    if (c_in == 64 && c_out == 64 && k == 3 && stride == 1 && pad == 1 && h == 8) return 1;
    if (c_in == 128 && c_out == 128 && k == 3 && stride == 1 && pad == 1 && h == 4) return 2;
    if (c_in == 64 && c_out == 128 && k == 3 && stride == 2 && pad == 1 && h == 8) return 3;
    if (c_in == 64 && c_out == 128 && k == 1 && stride == 2 && pad == 0 && h == 8) return 4;
    return 0;
}

IPSX_API int ipsx_conv2d_lds_nhwc_supported(int c_in, int c_out, int k, int stride, int pad, int h, int w) {
    return lds_conv_kind(c_in, c_out, k, stride, pad, h, w) != 0 ? 1 : 0;
}

IPSX_API int ipsx_conv2d_lds_nhwc(const ipsx_conv* cv, const float* x, float* y, int64_t n, int h, int w, void* stream) {
    return ipsx_conv2d_lds_nhwc_stats(cv, x, y, n, h, w, nullptr, nullptr, stream);
}

IPSX_API int64_t ipsx_conv2d_lds_nhwc_stats_slabs(int64_t n) { return n > 0 ? ipsx::cdiv(n, 4) : 0; }

IPSX_API int ipsx_conv2d_lds_nhwc_stats(const ipsx_conv* cv, const float* x, float* y, int64_t n, int h, int w, const float* shift,
                                        float* partial, void* stream) {
    IPSX_REQUIRE(cv && cv->w_packed && x && y && n >= 0, "conv2d_lds_nhwc: bad arguments");
    const int kind = cv->kh == cv->kw ? lds_conv_kind(cv->c_in, cv->c_out, cv->kh, cv->stride, cv->pad, h, w) : 0;
    IPSX_REQUIRE(kind != 0, "conv2d_lds_nhwc: %d -> %d, %dx%d / %d on %dx%d maps is not one of the fused trunk's stages", cv->c_in,
                 cv->c_out, cv->kh, cv->kw, cv->stride, h, w);
    if (n == 0) return IPSX_OK;
    ipsx::LdsConvArgs a;
    a.x = x; a.y = y; a.wp = cv->w_packed; a.n = n;
    a.stats = partial; a.shift = partial ? shift : nullptr;
    const size_t lds = (size_t)4 * ipsx::SLAB * sizeof(float);
    const dim3 grid((unsigned)ipsx::cdiv(n, 4)), block(256);
    hipStream_t s = ipsx::as_stream(stream);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ipsx::conv_lds_l1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ipsx::conv_lds_l2_kernel<128, 4, ipsx::PS2, ipsx::ZP2, 1, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ipsx::conv_lds_l2_kernel<64, 8, ipsx::PS1, ipsx::ZP1, 2, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ipsx::conv_lds_l2_kernel<64, 8, ipsx::PS1, ipsx::ZP1, 2, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    if (kind == 1) ipsx::conv_lds_l1_kernel<<<grid, block, lds, s>>>(a);
    else if (kind == 2) ipsx::conv_lds_l2_kernel<128, 4, ipsx::PS2, ipsx::ZP2, 1, 3><<<grid, block, lds, s>>>(a);
    else if (kind == 3) ipsx::conv_lds_l2_kernel<64, 8, ipsx::PS1, ipsx::ZP1, 2, 3><<<grid, block, lds, s>>>(a);
    else ipsx::conv_lds_l2_kernel<64, 8, ipsx::PS1, ipsx::ZP1, 2, 1><<<grid, block, lds, s>>>(a);
    return ipsx::launched("conv2d_lds_nhwc");
}

IPSX_API size_t ipsx_packed_conv_weight_bf16_bytes(int c_out, int c_in, int kh, int kw) {
    return (size_t)ipsx::cdiv(c_out, 32) * (size_t)ipsx::cdiv((int64_t)kh * kw * c_in, 16) * 1024;
}

IPSX_API int ipsx_pack_conv_weight_bf16(const float* w, int c_out, int c_in, int kh, int kw, void* packed, void* stream) {
    IPSX_REQUIRE(w && packed && c_out > 0 && c_in > 0 && kh > 0 && kw > 0, "pack_conv_weight_bf16: bad arguments");
    const size_t total = ipsx_packed_conv_weight_bf16_bytes(c_out, c_in, kh, kw) / 2;
    const int ksteps = (int)ipsx::cdiv((int64_t)kh * kw * c_in, 16);
    ipsx::pack_conv_weight_split_kernel<1><<<dim3((unsigned)ipsx::cdiv(total, 256)), dim3(256), 0, ipsx::as_stream(stream)>>>(
        w, c_out, c_in, kh, kw, ksteps, total, static_cast<unsigned short*>(packed));
    return ipsx::launched("pack_conv_weight_bf16");
}

IPSX_API size_t ipsx_packed_conv_weight_x3_bytes(int c_out, int c_in, int kh, int kw) {
    return 3 * ipsx_packed_conv_weight_bf16_bytes(c_out, c_in, kh, kw);
}

IPSX_API int ipsx_pack_conv_weight_x3(const float* w, int c_out, int c_in, int kh, int kw, void* packed, void* stream) {
    IPSX_REQUIRE(w && packed && c_out > 0 && c_in > 0 && kh > 0 && kw > 0, "pack_conv_weight_x3: bad arguments");
    const size_t total = ipsx_packed_conv_weight_x3_bytes(c_out, c_in, kh, kw) / 2;
    const int ksteps = (int)ipsx::cdiv((int64_t)kh * kw * c_in, 16);
    ipsx::pack_conv_weight_split_kernel<3><<<dim3((unsigned)ipsx::cdiv(total, 256)), dim3(256), 0, ipsx::as_stream(stream)>>>(
        w, c_out, c_in, kh, kw, ksteps, total, static_cast<unsigned short*>(packed));
    return ipsx::launched("pack_conv_weight_x3");
}

IPSX_API size_t ipsx_packed_stem_weight_split_bytes(int c_out, int planes) {
    return (planes == 1 || planes == 3) ? (size_t)ipsx::cdiv(c_out, 32) * 4 * planes * 1024 : 0;
}

IPSX_API int ipsx_pack_stem_weight_split(const float* w, int c_out, int planes, void* packed, void* stream) {
    IPSX_REQUIRE(w && packed && c_out > 0 && (planes == 1 || planes == 3), "pack_stem_weight_split: bad arguments");
    const size_t total = ipsx_packed_stem_weight_split_bytes(c_out, planes) / 2;
    const dim3 grid((unsigned)ipsx::cdiv(total, 256)), block(256);
    if (planes == 3)
        ipsx::pack_stem_weight_split_kernel<3><<<grid, block, 0, ipsx::as_stream(stream)>>>(w, c_out, total, static_cast<unsigned short*>(packed));
    else
        ipsx::pack_stem_weight_split_kernel<1><<<grid, block, 0, ipsx::as_stream(stream)>>>(w, c_out, total, static_cast<unsigned short*>(packed));
    return ipsx::launched("pack_stem_weight_split");
}

// Diagnostic switch (not part of include/ipsx.h; tests/test_hip_kernels.py, tools): which of the two exact fp32 kernels
// encodes - 0 the product's rule, 1 fused_trunk_kernel only, 2 fused_trunk_pair_kernel only.
extern "C" __attribute__((visibility("default"))) void ipsx_dbg_fused_trunk_pair(int mode) { ipsx::g_pair_mode = mode; }
// which build of the bf16 trunk IPSX_PRECISION=bf16 launches: 0 the default (the third, fused_trunk_bf16v3.h), 1 the first
// (fused_trunk_split.h), 2 the second (fused_trunk_bf16v2.h), 3 the third
extern "C" __attribute__((visibility("default"))) void ipsx_dbg_bf16_build(int build) { ipsx::g_bf16_build = build; }

// Diagnostic entry point (not part of include/ipsx.h): the fused trunk with s_memtime stamps,
// 16 x uint64 per wavefront = per patch, in launch order.  Used by tools/fused_stamps.py only.
extern "C" __attribute__((visibility("default"))) int ipsx_dbg_fused_trunk_stamps(
    const ipsx_trunk* t, const float* patches, int64_t n, float* emb, unsigned long long* stamps, void* stream) {
    if (!ipsx::fused_trunk_supported(t)) return ipsx::fail(IPSX_EINVAL, "trunk is not the fused shape");
    return ipsx::fused_launch(t, patches, n, emb, stamps, ipsx::as_stream(stream));
}
