// fused_stage.hip - the 64-channel residual stage of a ResNet-18 trunk (layer1: BasicBlocks 64 -> 64, 3x3, stride 1) on
// SMALL feature maps in ONE kernel, activations resident in LDS.
//
// Reference: the `layer1` entry of the nn.Sequential that architecture/ips_net.py:35-50 builds, at the reference's shipped
// Megapixel-MNIST patch size (config/mnist_config.yml:34-38: 50-px patches -> 25x25 after the stem -> 13x13 after the
// max-pool).  Layer by layer (conv_nhwc.hip) these four convolutions are the slowest part of that trunk: every
// activation is fetched once per tap from L2 and feeds only 64 output channels - 32 FLOP per operand byte, bound by the
// L2-to-CU traffic, 0.69 of the fp32 MFMA peak - and they are 47 % of its arithmetic.  Here a workgroup keeps G = 3
// patches' 13x13x64 maps in LDS (508 pixel rows of 68 floats: 135 KiB, one workgroup per compute unit) and runs all four
// convolutions on them in place:
//   * the 3 x 169 = 507 pixel rows are PACKED into 16 M-tiles of 32 (99 % full - a tile may straddle two patches: every
//     lane computes its own source rows);
//   * wave w owns M-tiles 4 w .. 4 w + 3 and both n-tiles (64 output channels): 4 x 2 accumulators; a stage = one k-group
//     (8 k) = 4 ds_read_b128 + two 16-byte weight loads + 32 MFMAs, operands one stage ahead, weights two (the data path of
//     fused_trunk.hip's 8x8 stage with twice the M per wave.  First version: 8 M-tiles x 1 n-tile per wave - 8 LDS reads
//     per stage, 64 operand registers, and the compiler parked operands in AccVGPRs: 128 v_accvgpr_read per 256 MFMAs,
//     1.5 other instructions per MFMA, 0.77 of peak);
//   * halo lanes read an all-zero pixel row (no select), per-lane tap validity is a 9-bit mask computed once;
//   * the identity of a block is NOT held in registers (there are none left): block 1's is the kernel's own input in
//     global memory, block 2's the output of block 1, which goes to the output buffer anyway - each thread re-reads exactly
//     the elements it wrote.
// Arithmetic: v_mfma_f32_32x32x2_f32 in the contract's k order (tap-major, 0,4,1,5,2,6,3,7 inside a group), padded taps
// contribute fma(0, w, acc), BatchNorm = fma(acc, alpha, shift), + identity, ReLU - bit-identical to conv_nhwc.hip and to
// the oracle.  Algorithmic work: 4 x 169 x 576 x 64 MAC per patch; bytes 43 KB in + 43 KB out per patch.

#include "ipsx_common.h"
#include "ipsx_math.h"

namespace ipsx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define FS_SB() __builtin_amdgcn_sched_barrier(0)
#define FS_SG_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define FS_SG_LDS(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define FS_SG_VMEM(n) __builtin_amdgcn_sched_group_barrier(0x020, n, 0)

struct StageArgs {
    const float* x;          // (n, H, W, 64) channels-last
    float* y;                // (n, H, W, 64)
    long long n;
    int n_block;             // BasicBlocks (1 or 2)
    const float *w[4], *al[4], *sh[4];   // b0.c1 b0.c2 b1.c1 b1.c2: packed weights (2 n-tiles x 72 k-groups), BN alpha / shift
    unsigned long long* stamps;          // diagnostic (ipsx_dbg_fused_stage_stamps): s_memtime of wave 0 of workgroup 0 at its
                                         // phase boundaries, or nullptr
};

#define FS_STAMP(k)                                                                                \
    do {                                                                                           \
        if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[k] = __builtin_amdgcn_s_memtime(); \
    } while (0)

template <int H, int W, int G>
struct FS {
    static constexpr int P = H * W;               // pixels per patch
    static constexpr int ROWS = G * P;            // packed pixel rows of a workgroup
    static constexpr int MT = (ROWS + 31) / 32;   // M-tiles
    static constexpr int MTW = (MT + 3) / 4;      // M-tiles per wave (each wave: MTW M-tiles x both n-tiles)
    static constexpr int PS = 68;                 // floats per pixel row: 64 channels + 4 pad (conflict-free b128 reads)
    static constexpr int TROWS = MT * 32;         // rows the tiles cover: [ROWS, TROWS) are written (garbage) and never read
    static constexpr int ZROW = TROWS;            // the all-zero pixel row
    static constexpr int LDS_BYTES = (TROWS + 1) * PS * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "stage does not fit the LDS");
};

template <int MTW>
struct FSOperands {
    float4 a[MTW];
};

// one 3x3 convolution 64 -> 64 over the slab: acc[t][nt] = this wave's M-tiles x the two n-tiles
template <int H, int W, int G>
__device__ __forceinline__ void fs_conv(const float* __restrict__ wp, const char* lds, const unsigned (&rowb)[FS<H, W, G>::MTW],
                                        const unsigned (&mask)[FS<H, W, G>::MTW], f32x16 (&acc)[FS<H, W, G>::MTW][2], int lane) {
    using F = FS<H, W, G>;
    constexpr int MTW = F::MTW;
    const int half = lane >> 5;
    const char* w = reinterpret_cast<const char*>(wp) + lane * 16;
    const unsigned zb = (unsigned)F::ZROW * F::PS * 4 + 16u * half;
#pragma unroll
    for (int t = 0; t < MTW; ++t)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][nt][r] = 0.0f;
    unsigned off[MTW];
#define FS_TAP(tap)                                                                            \
    do {                                                                                       \
        const int dy_ = (tap) / 3 - 1, dx_ = (tap) % 3 - 1;                                    \
        const int delta_ = (dy_ * W + dx_) * F::PS * 4;                                        \
        _Pragma("unroll") for (int t = 0; t < MTW; ++t)                                        \
            off[t] = ((mask[t] >> (tap)) & 1u) ? (unsigned)((int)rowb[t] + delta_) : zb;       \
    } while (0)
#define FS_LOAD(S, CG)                                                                         \
    do {                                                                                       \
        _Pragma("unroll") for (int t = 0; t < MTW; ++t)                                        \
            S.a[t] = *reinterpret_cast<const float4*>(lds + off[t] + (CG) * 32);               \
    } while (0)
#define FS_LOADB(B, g)                                                                         \
    do {                                                                                       \
        const int g_ = (g) < 72 ? (g) : 71;                                                    \
        B[0] = *reinterpret_cast<const float4*>(w + (size_t)g_ * 1024);                        \
        B[1] = *reinterpret_cast<const float4*>(w + (size_t)(72 + g_) * 1024);                 \
    } while (0)
#define FS_MMA(S, B)                                                                           \
    do {                                                                                       \
        const float b0_[4] = {B[0].x, B[0].y, B[0].z, B[0].w}, b1_[4] = {B[1].x, B[1].y, B[1].z, B[1].w}; \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                        \
            _Pragma("unroll") for (int t = 0; t < MTW; ++t) {                                  \
                const float av_ = j == 0 ? S.a[t].x : (j == 1 ? S.a[t].y : (j == 2 ? S.a[t].z : S.a[t].w)); \
                acc[t][0] = FS_MFMA(av_, b0_[j], acc[t][0]);                                   \
                acc[t][1] = FS_MFMA(av_, b1_[j], acc[t][1]);                                   \
            }                                                                                  \
        }                                                                                      \
    } while (0)
// the MTW LDS loads and the two weight loads of the next stage, spread between this stage's 8 * MTW MFMAs
#define FS_POST()                                                                              \
    do {                                                                                       \
        _Pragma("unroll") for (int t = 0; t < MTW; ++t) { FS_SG_MFMA(4); FS_SG_LDS(1); }       \
        FS_SG_MFMA(MTW); FS_SG_VMEM(1); FS_SG_MFMA(MTW); FS_SG_VMEM(1); FS_SG_MFMA(2 * MTW);   \
        FS_SB();                                                                               \
    } while (0)
    FSOperands<MTW> sa, sb;
    float4 b0[2], b1[2], b2[2], b3[2];           // weight ring (both n-tiles): slot = stage & 3, refilled two stages ahead
    FS_TAP(0);
    FS_LOADB(b0, 0);
    FS_LOADB(b1, 1);
    FS_LOAD(sa, 0);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const int g = tap * 8;
        FS_LOAD(sb, 1); FS_LOADB(b2, g + 2); FS_MMA(sa, b0); FS_POST();
        FS_LOAD(sa, 2); FS_LOADB(b3, g + 3); FS_MMA(sb, b1); FS_POST();
        FS_LOAD(sb, 3); FS_LOADB(b0, g + 4); FS_MMA(sa, b2); FS_POST();
        FS_LOAD(sa, 4); FS_LOADB(b1, g + 5); FS_MMA(sb, b3); FS_POST();
        FS_LOAD(sb, 5); FS_LOADB(b2, g + 6); FS_MMA(sa, b0); FS_POST();
        FS_LOAD(sa, 6); FS_LOADB(b3, g + 7); FS_MMA(sb, b1); FS_POST();
        FS_LOAD(sb, 7); FS_LOADB(b0, g + 8); FS_MMA(sa, b2); FS_POST();
        {   // the next tap's source rows (the last tap re-reads its own: the prefetch past the end is never used)
            const int nx = tap < 8 ? tap + 1 : 8;
            const int dy_ = nx / 3 - 1, dx_ = nx % 3 - 1;
            const int delta_ = (dy_ * W + dx_) * F::PS * 4;
#pragma unroll
            for (int t = 0; t < MTW; ++t) off[t] = ((mask[t] >> nx) & 1u) ? (unsigned)((int)rowb[t] + delta_) : zb;
        }
        FS_LOAD(sa, 0); FS_LOADB(b1, g + 9); FS_MMA(sb, b3); FS_POST();
    }
#undef FS_TAP
#undef FS_LOAD
#undef FS_LOADB
#undef FS_MMA
#undef FS_POST
}

template <int H, int W, int G>
__global__ __launch_bounds__(256, 1) void fused_stage64_kernel(StageArgs a) {
    using F = FS<H, W, G>;
    constexpr int MTW = F::MTW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, half = lane >> 5;
    const int mt0 = wave * MTW;
    const long long p0 = (long long)blockIdx.x * G;
    const int np = (int)(a.n - p0 < G ? a.n - p0 : G);                  // patches of this workgroup
    const int rows_valid = np * F::P;
    const float* xin = a.x + (size_t)p0 * F::P * 64;
    float* yout = a.y + (size_t)p0 * F::P * 64;
    FS_STAMP(0);

    // global accesses are raw buffer operations on a buffer that ends with the last VALID row: one 32-bit offset per lane
    // (the per-element part is a scalar offset) instead of 128 64-bit addresses, and the rows of a short last workgroup
    // read as zeros / are not stored by the hardware's bounds check
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, rows_valid * 256, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(yout, 0, rows_valid * 256, 0x00020000);
    // ---- the patches' maps -> LDS rows [pix][68]; rows beyond the valid ones and the zero row: zeros
    {   // (8 loads of a thread in flight: one at a time this was 38 k cycles of a workgroup's 730 k)
        typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(rx, 0, 0, 0)) vec16;
        constexpr int TOTAL = (F::TROWS + 1) * 16;
        for (int e0 = threadIdx.x; e0 < TOTAL; e0 += 256 * 8) {
            vec16 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + 256 * u;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, (e < F::TROWS * 16) ? e * 16 : (int)0x7fffffff, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + 256 * u;
                if (e < TOTAL) *reinterpret_cast<vec16*>(lds + (e >> 4) * F::PS + 4 * (e & 15)) = v[u];
            }
        }
    }
    // ---- per lane and M-tile: byte offset of its own pixel row (+ its half's 16 bytes) and the 9-bit tap mask
    unsigned rowb[MTW], mask[MTW];
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
        const int m = (mt0 + t) * 32 + i;
        const bool valid = m < rows_valid;
        const int pix = m % F::P, y = pix / W, x = pix % W;
        unsigned mk = 0u;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            if (valid && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W) mk |= 1u << tap;
        }
        mask[t] = mk;
        rowb[t] = (unsigned)m * F::PS * 4 + 16u * half;
    }
    __syncthreads();
    FS_STAMP(1);

    f32x16 acc[MTW][2];
    const char* ldsb = reinterpret_cast<const char*>(lds);
#pragma unroll 1
    for (int cv = 0; cv < 2 * a.n_block; ++cv) {
        fs_conv<H, W, G>(a.w[cv], ldsb, rowb, mask, acc, lane);
        FS_STAMP(2 + 2 * cv);
        const float A0 = a.al[cv][i], B0 = a.sh[cv][i], A1 = a.al[cv][32 + i], B1 = a.sh[cv][32 + i];
        const bool second = (cv & 1) != 0;                             // conv2 of a block: + identity, result also to y
        const bool last = cv + 1 == 2 * a.n_block;                     // nothing reads the slab after the last convolution
        const __amdgpu_buffer_rsrc_t ri = cv == 1 ? rx : ry;           // block 1: the kernel's input; block 2: block 1's output
        const int voff = ((mt0 * 32 + 4 * half) * 64 + i) * 4;         // this lane's part of every element offset (n-tile 0)
        // the identity of unit (t, nt) is requested one unit ahead (two 16-register buffers): its L2 round trip - 16 strided
        // 4-byte loads per lane - hides behind the previous unit's arithmetic and stores instead of 8 exposed waits per conv
        float idA[16], idB[16];
#define FS_IDLOAD(BUF, u)                                                                          \
        do {                                                                                       \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)                                         \
                BUF[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(           \
                    ri, voff, (((u) >> 1) * 32 + (r & 3) + 8 * (r >> 2)) * 256 + ((u) & 1) * 128, 0)); \
        } while (0)
#define FS_UNIT(BUF, u)                                                                            \
        do {                                                                                       \
            constexpr int t_ = (u) >> 1, nt_ = (u) & 1;                                            \
            const float A = nt_ ? A1 : A0, B = nt_ ? B1 : B0;                                      \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                       \
                const int m = (mt0 + t_) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;                 \
                float v = __builtin_fmaf(acc[t_][nt_][r], A, B);                                   \
                if (second) v = v + BUF[r];                                                        \
                v = v > 0.0f ? v : 0.0f;                                                           \
                if (!last) lds[m * F::PS + nt_ * 32 + i] = v;                                      \
                if (second)                                                                        \
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, voff, \
                                                          (t_ * 32 + (r & 3) + 8 * (r >> 2)) * 256 + nt_ * 128, 0); \
            }                                                                                      \
            FS_SB();                                                                               \
        } while (0)
        if (second) FS_IDLOAD(idA, 0);
        __syncthreads();                                               // every wave has read what it needs of the old map
        static_assert(MTW == 4, "the epilogue below is written out for 4 M-tiles per wave");
        if (second) FS_IDLOAD(idB, 1);
        FS_UNIT(idA, 0);
        if (second) FS_IDLOAD(idA, 2);
        FS_UNIT(idB, 1);
        if (second) FS_IDLOAD(idB, 3);
        FS_UNIT(idA, 2);
        if (second) FS_IDLOAD(idA, 4);
        FS_UNIT(idB, 3);
        if (second) FS_IDLOAD(idB, 5);
        FS_UNIT(idA, 4);
        if (second) FS_IDLOAD(idA, 6);
        FS_UNIT(idB, 5);
        if (second) FS_IDLOAD(idB, 7);
        FS_UNIT(idA, 6);
        FS_UNIT(idB, 7);
#undef FS_IDLOAD
#undef FS_UNIT
        if (last) FS_STAMP(3 + 2 * cv);
        __syncthreads();
        if (!last) FS_STAMP(3 + 2 * cv);
    }
}

// ------------------------------------------------------------------------------------------------ stem + max-pool, 50 px
// conv7x7/2 (1 -> 64) + BatchNorm + ReLU + max-pool 3x3/2 of a 1x50x50 patch in one kernel, the 25x25x64 stem output never
// in memory: wave = patch, the zero-padded 56x56 input in LDS (every tap address = base + immediate, no halo mask - the
// scheme of fused_trunk.hip's stem_pool), one M-tile = ONE stem row (25 of 32 pixel lanes), 25 MFMA k-steps x 2 n-tiles per
// row; rows go through in pairs (2t, 2t+1): with the previous pair's last row that is the vertical window of pooled row t;
// the horizontal window needs the other lane half's columns (one exchange of 16 registers per n-tile), and the 13 pooled
// columns x 64 channels leave as channels-last rows for the stage kernel above.  Replaces conv_any_kernel (the gather-bound
// generic stem: 0.36 of peak, 1.13 ms per 14,400 patches) + maxpool_3x3s2_nhwc_kernel (0.63 ms).
// Exactness: the same fma chain per stem output as conv.hip / the oracle (k order 0,4,1,5,2,6,3,7 inside a group, K = 49);
// max-pool padding is -inf in the contract and 0 here - identical, because every window holds a real post-ReLU value >= 0.
struct StemArgs {
    const float* patches;     // (n, 1, 50, 50)
    float* y;                 // (n, 13, 13, 64) channels-last
    long long n;
    const float *w, *al, *sh; // packed stem weights (2 n-tiles x 7 k-groups), BN alpha / shift
};

constexpr int SPW = 56;                   // padded input width (50 + 2 x 3)
constexpr int SP_SLAB = SPW * SPW + 64;   // floats per wave (+ slack: lanes 25..31 of a row read past its end)

__device__ __forceinline__ float fs_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

__global__ __launch_bounds__(256, 2) void stem_pool50_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, half = lane >> 5;
    long long p = (long long)blockIdx.x * 4 + wave;
    const bool live = p < a.n;
    if (!live) p = a.n - 1;                                  // tail: recompute the last patch, store nothing
    float* S = lds + wave * SP_SLAB;
    // ---- the patch -> LDS, zero-padded (image at rows / columns 3..52)
    for (int z = lane; z < SP_SLAB / 4; z += 64) reinterpret_cast<float4*>(S)[z] = make_float4(0.f, 0.f, 0.f, 0.f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        const float2* src = reinterpret_cast<const float2*>(a.patches + (size_t)p * 2500);
        for (int e0 = lane; e0 < 1250; e0 += 64 * 5) {
            float2 v[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) v[u] = src[e0 + 64 * u < 1250 ? e0 + 64 * u : e0];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int e = e0 + 64 * u;
                if (e < 1250) {
                    const int yy = e / 25, xx = 2 * (e - yy * 25);
                    S[(yy + 3) * SPW + xx + 3] = v[u].x;
                    S[(yy + 3) * SPW + xx + 4] = v[u].y;
                }
            }
        }
    }
    // ---- weights of the 25 k-steps (k = 8 kg + 4 half + j; step 24 is k = 48 / the non-existent 52), BN of this lane's channels
    float bw0[28], bw1[28];
    {
        const float4* wp0 = reinterpret_cast<const float4*>(a.w) + lane;
        const float4* wp1 = wp0 + 7 * 64;
#pragma unroll
        for (int kg = 0; kg < 7; ++kg) {
            const float4 v0 = wp0[kg * 64], v1 = wp1[kg * 64];
            bw0[4 * kg] = v0.x; bw0[4 * kg + 1] = v0.y; bw0[4 * kg + 2] = v0.z; bw0[4 * kg + 3] = v0.w;
            bw1[4 * kg] = v1.x; bw1[4 * kg + 1] = v1.y; bw1[4 * kg + 2] = v1.z; bw1[4 * kg + 3] = v1.w;
        }
    }
    const float al0 = a.al[i], sh0 = a.sh[i], al1 = a.al[32 + i], sh1 = a.sh[32 + i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // one stem row `oy` -> BN + ReLU'd values of this lane's 16 columns x = 4 half + (r & 3) + 8 (r >> 2), per n-tile;
    // columns >= 25 do not exist (their lanes read beside the image): 0
    auto stem_row = [&](int oy, float (&v0)[16], float (&v1)[16]) {
        const float* base = S + (2 * oy) * SPW + 2 * i;
        const float* baseN = base + half * 4;
        const float* baseW = base + half * (SPW - 3);
        float av[25];
#pragma unroll
        for (int st = 0; st < 25; ++st) {
            const int k0 = 8 * (st >> 2) + (st & 3), ky0 = k0 / 7, kx0 = k0 % 7;
            av[st] = (kx0 <= 2) ? baseN[ky0 * SPW + kx0] : baseW[ky0 * SPW + kx0];
        }
        av[24] = half ? 0.0f : av[24];
        FS_SB();
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
#pragma unroll
        for (int st = 0; st < 25; ++st) {
            acc0 = FS_MFMA(av[st], bw0[st], acc0);
            acc1 = FS_MFMA(av[st], bw1[st], acc1);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool beyond = (r >> 2) == 3 && (half != 0 || (r & 3) >= 1);        // column >= 25
            const float x0 = __builtin_fmaf(acc0[r], al0, sh0), x1 = __builtin_fmaf(acc1[r], al1, sh1);
            v0[r] = (beyond || !(x0 > 0.0f)) ? 0.0f : x0;
            v1[r] = (beyond || !(x1 > 0.0f)) ? 0.0f : x1;
        }
    };

    float prev0[16], prev1[16];                              // stem row 2t - 1 (row -1: padding)
#pragma unroll
    for (int r = 0; r < 16; ++r) { prev0[r] = 0.0f; prev1[r] = 0.0f; }
    float* yout = a.y + (size_t)p * 169 * 64;
#pragma unroll 1
    for (int t = 0; t < 13; ++t) {
        float va0[16], va1[16], vb0[16], vb1[16];
        stem_row(2 * t, va0, va1);
        if (t < 12) {
            stem_row(2 * t + 1, vb0, vb1);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { vb0[r] = 0.0f; vb1[r] = 0.0f; }            // row 25: padding
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float vm[16], oth[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                vm[r] = nt ? fs_max3(prev1[r], va1[r], vb1[r]) : fs_max3(prev0[r], va0[r], vb0[r]);
                oth[r] = lane_xor_f32<32>(vm[r], lane);
            }
            // the row's columns -1 .. 25 in order: column x lives in half (x >> 2) & 1, register (x & 3) + 4 (x >> 3)
            float full[27];
            full[0] = 0.0f;
#pragma unroll
            for (int x = 0; x < 26; ++x) {
                const int own = (x >> 2) & 1, r = (x & 3) + 4 * (x >> 3);
                full[1 + x] = x >= 25 ? 0.0f : ((half == own) ? vm[r] : oth[r]);
            }
            // pooled columns: half 0 stores tx = 0..6, half 1 stores tx = 7..12
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const float lo = fs_max3(full[2 * j], full[2 * j + 1], full[2 * j + 2]);
                const float hi = j < 6 ? fs_max3(full[2 * (7 + j)], full[2 * (7 + j) + 1], full[2 * (7 + j) + 2]) : 0.0f;
                const int tx = half ? 7 + j : j;
                if (live && tx < 13) yout[(size_t)(t * 13 + tx) * 64 + nt * 32 + i] = half ? hi : lo;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { prev0[r] = vb0[r]; prev1[r] = vb1[r]; }
    }
}

// ------------------------------------------------------------------------------------------------ stem + max-pool, 3 x 100 px
// The traffic-sign configuration's stem (config/traffic_config.yml: 3x100x100 patches -> 50x50x64 -> 25x25x64 pooled), same
// scheme as stem_pool50_kernel with three input planes: ONE patch per workgroup (the zero-padded 3 x 106 x 106 image fills
// 132 KiB of LDS), its four wavefronts take bands of 7 / 6 / 6 / 6 pooled rows (each computes the stem row above its band
// itself); a stem row is 2 M-tiles (50 of 64 pixel lanes) x 2 n-tiles x 75 MFMA k-steps (K = 147, k = 3 tap + channel:
// every tap address is base + a constant that depends on the lane half only); rows go through in pairs, the pool runs on the
// accumulators.  Replaces conv_any_kernel (0.36 of peak) + maxpool_3x3s2_nhwc_kernel: 11.5 % of that trunk's kernel time.
struct Stem3Args {
    const float* patches;     // (n, 3, 100, 100)
    float* y;                 // (n, 25, 25, 64) channels-last
    long long n;
    const float *w, *al, *sh; // packed stem weights (2 n-tiles x 19 k-groups), BN alpha / shift
};

constexpr int S3W = 106;                       // padded width / height
constexpr int S3PLANE = S3W * S3W;             // floats per channel plane
constexpr int S3_FLOATS = 3 * S3PLANE + 64;    // + slack: pixel lanes 50..63 of the last row read past the image

// offset (floats) of contraction index k inside the padded planes, relative to the receptive field's top-left corner
__host__ __device__ constexpr int s3_off(int k) {
    return k >= 147 ? 0 : (k % 3) * S3PLANE + ((k / 3) / 7) * S3W + (k / 3) % 7;
}

__global__ __launch_bounds__(256, 1) void stem_pool100x3_kernel(Stem3Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, half = lane >> 5;
    const long long p = blockIdx.x;
    float* S = lds;
    for (int z = threadIdx.x; z < S3_FLOATS / 4; z += 256) reinterpret_cast<float4*>(S)[z] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    {   // 3 x 100 x 100 floats, rows of 25 float4 -> image at rows / columns 3..102 of every plane
        const float4* src = reinterpret_cast<const float4*>(a.patches + (size_t)p * 30000);
        for (int e0 = threadIdx.x; e0 < 7500; e0 += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[e0 + 256 * u < 7500 ? e0 + 256 * u : e0];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + 256 * u;
                if (e < 7500) {
                    const int c = e / 2500, rem = e - c * 2500, yy = rem / 25, xx = 4 * (rem - yy * 25);
                    float* d = S + c * S3PLANE + (yy + 3) * S3W + xx + 3;
                    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                }
            }
        }
    }
    // weights: 19 k-groups x 2 n-tiles of 16 bytes per lane (k = 8 g + 4 half + j; zeros from k = 147 on), streamed from
    // L2 one group ahead of its MFMAs (152 registers if held - the first version did, and spilled); BN of this lane's channels
    const float4* wp0 = reinterpret_cast<const float4*>(a.w) + lane;
    const float4* wp1 = wp0 + 19 * 64;
    const float al0 = a.al[i], sh0 = a.sh[i], al1 = a.al[32 + i], sh1 = a.sh[32 + i];
    __syncthreads();

    // one stem row `oy` -> BN + ReLU'd values, v[mt][nt][r] = column 32 mt + 4 half + (r & 3) + 8 (r >> 2); columns >= 50: 0
    auto stem_row = [&](int oy, float (&v)[2][2][16]) {
        const float* base = S + (2 * oy) * S3W + 2 * i;
        f32x16 acc[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
        float4 w0 = wp0[0], w1 = wp1[0];
        float a0[4], a1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = half ? s3_off(4 + j) : s3_off(j);
            a0[j] = base[off];
            a1[j] = base[off + 64];
        }
#pragma unroll
        for (int g = 0; g < 19; ++g) {
            // operands of the NEXT k-group are requested before this group's MFMAs
            float4 nw0 = w0, nw1 = w1;
            float n0[4] = {0.f, 0.f, 0.f, 0.f}, n1[4] = {0.f, 0.f, 0.f, 0.f};
            if (g + 1 < 19) {
                nw0 = wp0[(g + 1) * 64];
                nw1 = wp1[(g + 1) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (g + 1 == 18 && j == 3) continue;                           // k = 147 / 151: both halves beyond K
                    const int off = half ? s3_off(8 * (g + 1) + 4 + j) : s3_off(8 * (g + 1) + j);
                    n0[j] = base[off];
                    n1[j] = base[off + 64];
                }
            }
            const float b0[4] = {w0.x, w0.y, w0.z, w0.w}, b1[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (g == 18 && j == 3) continue;
                acc[0][0] = FS_MFMA(a0[j], b0[j], acc[0][0]);
                acc[0][1] = FS_MFMA(a0[j], b1[j], acc[0][1]);
                acc[1][0] = FS_MFMA(a1[j], b0[j], acc[1][0]);
                acc[1][1] = FS_MFMA(a1[j], b1[j], acc[1][1]);
            }
            w0 = nw0; w1 = nw1;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a0[j] = n0[j]; a1[j] = n1[j]; }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // column >= 50 <=> tile 1 and 4 half + (r & 3) + 8 (r >> 2) >= 18
                const bool beyond = mt == 1 && ((r >> 2) == 3 || ((r >> 2) == 2 && (half != 0 || (r & 3) >= 2)));
                const float x0 = __builtin_fmaf(acc[mt][0][r], al0, sh0), x1 = __builtin_fmaf(acc[mt][1][r], al1, sh1);
                v[mt][0][r] = (beyond || !(x0 > 0.0f)) ? 0.0f : x0;
                v[mt][1][r] = (beyond || !(x1 > 0.0f)) ? 0.0f : x1;
            }
    };

    const int t0 = wave == 0 ? 0 : 1 + 6 * wave, t1 = 7 + 6 * wave;      // pooled rows [t0, t1): 0-6, 7-12, 13-18, 19-24
    float prev[2][2][16];
    if (t0 > 0) {
        stem_row(2 * t0 - 1, prev);
    } else {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) prev[mt][nt][r] = 0.0f;                // row -1: padding
    }
    float* yout = a.y + (size_t)p * 625 * 64;
#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
        float vb[2][2][16];
        {   // vertical window, as the rows arrive: prev <- max(prev, row 2t), then max with row 2t + 1
            float va[2][2][16];
            stem_row(2 * t, va);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) prev[mt][nt][r] = __builtin_fmaxf(prev[mt][nt][r], va[mt][nt][r]);
        }
        stem_row(2 * t + 1, vb);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float vm[2][16], oth[2][16];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    vm[mt][r] = __builtin_fmaxf(prev[mt][nt][r], vb[mt][nt][r]);
                    oth[mt][r] = lane_xor_f32<32>(vm[mt][r], lane);
                }
            // the row's columns -1 .. 50 in order: column x lives in tile x >> 5, half (x >> 2) & 1, register (x & 3) + 4 ((x & 31) >> 3)
            float full[52];
            full[0] = 0.0f;
#pragma unroll
            for (int x = 0; x < 51; ++x) {
                const int mt = x >> 5, own = (x >> 2) & 1, r = (x & 3) + 4 * ((x & 31) >> 3);
                full[1 + x] = x >= 50 ? 0.0f : ((half == own) ? vm[mt & 1][r] : oth[mt & 1][r]);
            }
            // pooled columns: half 0 stores tx = 0..12, half 1 stores tx = 13..24
#pragma unroll
            for (int j = 0; j < 13; ++j) {
                const float lo = fs_max3(full[2 * j], full[2 * j + 1], full[2 * j + 2]);
                const float hi = j < 12 ? fs_max3(full[2 * (13 + j)], full[2 * (13 + j) + 1], full[2 * (13 + j) + 2]) : 0.0f;
                const int tx = half ? 13 + j : j;
                if (tx < 25) yout[(size_t)(t * 25 + tx) * 64 + nt * 32 + i] = half ? hi : lo;
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) prev[mt][nt][r] = vb[mt][nt][r];
    }
}

static bool stem_pool100x3_supported(const ipsx_trunk* t) {
    const char* e = getenv("IPSX_NO_FUSED");
    if (e && e[0] == '1') return false;
    const ipsx_conv& c = t->stem;
    return t->c_in == 3 && t->h == 100 && t->w == 100 && t->patch_dtype == 0 && c.c_in == 3 && c.c_out == 64 && c.kh == 7 &&
           c.kw == 7 && c.stride == 2 && c.pad == 3 && c.w_packed && c.alpha && c.shift;
}

static bool stem_pool50_supported(const ipsx_trunk* t) {
    const char* e = getenv("IPSX_NO_FUSED");
    if (e && e[0] == '1') return false;
    const ipsx_conv& c = t->stem;
    return t->c_in == 1 && t->h == 50 && t->w == 50 && t->patch_dtype == 0 && c.c_in == 1 && c.c_out == 64 && c.kh == 7 &&
           c.kw == 7 && c.stride == 2 && c.pad == 3 && c.w_packed && c.alpha && c.shift;
}

bool fused_stem_pool50_covers(const ipsx_trunk* t) { return t && stem_pool50_supported(t); }
bool fused_stem_pool100x3_covers(const ipsx_trunk* t) { return t && stem_pool100x3_supported(t); }

// stem + max-pool of 1x50x50 patches -> (n, 13, 13, 64) channels-last; returns 1 when it ran, 0 when the trunk is another shape
int fused_stem_pool50(const ipsx_trunk* t, const float* patches, float* y, int64_t n, hipStream_t s) {
    if (t && stem_pool100x3_supported(t)) {                            // the traffic-sign stem: one patch per workgroup
        if (n <= 0) return 1;
        Stem3Args a3;
        a3.patches = patches; a3.y = y; a3.n = n;
        a3.w = t->stem.w_packed; a3.al = t->stem.alpha; a3.sh = t->stem.shift;
        static bool attr3 = false;
        if (!attr3) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_pool100x3_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(S3_FLOATS * sizeof(float)));
            attr3 = true;
        }
        stem_pool100x3_kernel<<<dim3((unsigned)n), dim3(256), S3_FLOATS * sizeof(float), s>>>(a3);
        return launched("stem_pool100x3") == IPSX_OK ? 1 : -1;
    }
    if (!t || !stem_pool50_supported(t)) return 0;
    if (n <= 0) return 1;
    StemArgs a;
    a.patches = patches; a.y = y; a.n = n;
    a.w = t->stem.w_packed; a.al = t->stem.alpha; a.sh = t->stem.shift;
    stem_pool50_kernel<<<dim3((unsigned)cdiv(n, 4)), dim3(256), 4 * SP_SLAB * sizeof(float), s>>>(a);
    return launched("stem_pool50") == IPSX_OK ? 1 : -1;
}

// the leading run of plain 64 -> 64 BasicBlocks of `blocks` that the kernel covers on an h x w map (0 = none)
int fused_stage64_blocks(const ipsx_block* blocks, int n_block, int h, int w) {
    const char* e = getenv("IPSX_NO_FUSED");           // (read per call: tests switch it)
    const bool off = e && e[0] == '1';
    if (off || !blocks || !(h == 13 && w == 13)) return 0;
    int k = 0;
    for (; k < n_block && k < 2; ++k) {
        const ipsx_block& b = blocks[k];
        if (b.n_conv != 2 || b.has_down) break;
        bool ok = true;
        for (int j = 0; j < 2; ++j) {
            const ipsx_conv& c = b.conv[j];
            ok = ok && c.c_in == 64 && c.c_out == 64 && c.kh == 3 && c.kw == 3 && c.stride == 1 && c.pad == 1 && c.w_packed &&
                 c.alpha && c.shift;
        }
        if (!ok) break;
    }
    return k;
}

static unsigned long long* g_stage_stamps = nullptr;     // diagnostic only (ipsx_dbg_fused_stage_stamps)

int fused_stage64(const ipsx_block* blocks, int n_block, const float* x, float* y, int64_t n, int h, int w, hipStream_t s) {
    IPSX_REQUIRE(blocks && x && y && n >= 0 && x != y, "fused_stage64: bad arguments");
    IPSX_REQUIRE(fused_stage64_blocks(blocks, n_block, h, w) == n_block && n_block >= 1, "fused_stage64: shape not covered");
    if (n == 0) return IPSX_OK;
    StageArgs a;
    a.x = x; a.y = y; a.n = n; a.n_block = n_block;
    a.stamps = g_stage_stamps;
    for (int k = 0; k < 4; ++k) { a.w[k] = nullptr; a.al[k] = nullptr; a.sh[k] = nullptr; }
    for (int k = 0; k < n_block; ++k)
        for (int j = 0; j < 2; ++j) {
            a.w[2 * k + j] = blocks[k].conv[j].w_packed;
            a.al[2 * k + j] = blocks[k].conv[j].alpha;
            a.sh[2 * k + j] = blocks[k].conv[j].shift;
        }
    using F = FS<13, 13, 3>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_stage64_kernel<13, 13, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, F::LDS_BYTES);
        attr = true;
    }
    fused_stage64_kernel<13, 13, 3><<<dim3((unsigned)cdiv(n, 3)), dim3(256), F::LDS_BYTES, s>>>(a);
    return launched("fused_stage64");
}

}  // namespace ipsx

// Diagnostic entry point (not part of include/ipsx.h): a device buffer of 16 uint64 that the next fused_stage64 launches
// fill with the s_memtime of workgroup 0 / wave 0 at its phase boundaries (tools/stage_stamps.py); NULL switches it off.
extern "C" __attribute__((visibility("default"))) void ipsx_dbg_fused_stage_stamps(unsigned long long* buf) {
    ipsx::g_stage_stamps = buf;
}
