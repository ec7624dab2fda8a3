// dgrad_s2.hip - the data gradient of the 32-px trunk's strided 3x3 convolution (64 -> 128 channels, stride 2, 8x8 -> 4x4 maps)
// for the TRAINING step (reference: loss.backward() of training/iterative.py:157-163 through layer2[0].conv1 of the
// torchvision ResNet that architecture/ips_net.py:17-52 builds).
//
//     dx[iy][ix][ci] = sum over (ky, kx, co) with (iy + 1 - ky), (ix + 1 - kx) even of
//                      dy[(iy + 1 - ky) / 2][(ix + 1 - kx) / 2][co] * W[co][ci][ky][kx]
//
// Until round 5 this ran as a stride-1 convolution of dy SPREAD over a zero 8x8 map (csrc/conv_nhwc.hip): three of four
// multiplications by a zero, 110 us at 1,024 patches for 2.4 GFLOP of work.  Here the input pixels are taken by PARITY
// CLASS (iy % 2, ix % 2): a class has 16 pixels per patch and 1, 2, 2 or 4 taps that reach a dy pixel at all - nine taps in
// all instead of thirty-six.  A workgroup takes 4 patches (dy maps in LDS, pixel-major with a zero row for the taps that
// leave the map - the fused trunk's 4x4 stage layout), wavefront (pp, ct) the 32 rows = two patches x 16 pixels of every
// class in turn for the 32 input channels of column tile ct: 72 stages of 8 v_mfma_f32_32x32x2_f32, weights (the packed
// data-gradient form: rotated by 180 degrees, transposed) streamed from L2 two stages ahead, operands by 16-byte LDS reads.
// The sum over a tap's 128 channels runs on two accumulators (even / odd channel groups): fp32 rounding differs from the
// spread convolution's single chain - the training path is tolerance-tested (tests/test_hip_train.py), not bit-pinned.

#include "ipsx_common.h"

namespace ipsx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define DG_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int DG_PS = 132;                     // floats per dy pixel row in LDS: 128 channels + 4 pad
constexpr int DG_ZP = 16;                      // the all-zero pixel row
constexpr int DG_SLAB = (DG_ZP + 1) * DG_PS;   // floats per patch

struct DgTap { int tp, oy, ox, last; };        // packed tap index (2 - ky) * 3 + (2 - kx); dy pixel offset; last tap of its class
// classes (iy % 2, ix % 2) = (0,0), (0,1), (1,0), (1,1); an even coordinate meets tap 1 at offset 0, an odd one taps 0 (offset
// +1) and 2 (offset 0)
__device__ constexpr DgTap DG[9] = {{4, 0, 0, 1}, {5, 0, 1, 0}, {3, 0, 0, 1}, {7, 1, 0, 0}, {1, 0, 0, 1},
                                    {8, 1, 1, 0}, {6, 1, 0, 0}, {2, 0, 1, 0}, {0, 0, 0, 1}};
__device__ constexpr int DG_CLASS[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};

struct DgradArgs {
    const float* dy;       // (n, 4, 4, 128) channels-last
    float* dx;             // (n, 8, 8, 64) channels-last
    const float* wp;       // packed data-gradient weights: 2 n-tiles x (9 taps x 16 k-groups) x 64 lanes x 4
    long long n;
};

// KS = 3: the 3x3 / 2 layer (pad 1).  KS = 1: the 1x1 / 2 projection beside it (pad 0: dx[2a][2b] = W^T dy[a][b], the other three
// parity classes receive no gradient: zeros) - it ran as a 1x1 convolution over the spread map too.
template <int KS>
__global__ __launch_bounds__(256, 2) void dgrad_s2_lds_kernel(DgradArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[4 * DG_SLAB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, half = lane >> 5;
    const long long p_first = (long long)blockIdx.x * 4;
    for (int q = threadIdx.x; q < 4 * 2048 / 4; q += 256) {              // the four dy maps, 16 bytes at a time
        const int pl = q >> 9, e = (q & 511) * 4;                         // patch, element: pixel e / 128, channel e % 128
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (p_first + pl < a.n) v = *reinterpret_cast<const float4*>(a.dy + (size_t)(p_first + pl) * 2048 + e);
        *reinterpret_cast<float4*>(lds + pl * DG_SLAB + (e >> 7) * DG_PS + (e & 127)) = v;
    }
    for (int z = lane; z < DG_PS; z += 64) lds[wave * DG_SLAB + DG_ZP * DG_PS + z] = 0.0f;
    __syncthreads();
    const int pp = wave >> 1, ct = wave & 1;
    const int pl = 2 * pp + (i >> 4), ca = (i & 15) >> 2, cb = i & 3;     // this lane's row: patch, class pixel (ca, cb)
    const float* Sp = lds + pl * DG_SLAB + 4 * half;
    constexpr int NTAP = KS * KS, NST = 8 * NTAP;                       // taps, stages
    const char* w = reinterpret_cast<const char*>(a.wp) + (size_t)ct * (NTAP * 16) * 1024 + lane * 16;
    const float* arow[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) {
        const int sy = ca + (KS == 3 ? DG[e].oy : 0), sx = cb + (KS == 3 ? DG[e].ox : 0);
        arow[e] = Sp + ((sy < 4 && sx < 4) ? sy * 4 + sx : DG_ZP) * DG_PS;
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    float4 A[2][2], B[4][2];
    // stage s = 8 e + q: tap e, channel groups 2 q, 2 q + 1 (16 channels)
#define DG_LOADA(buf, s) do { const float* r_ = arow[(s) >> 3] + ((s) & 7) * 16;                      \
        A[buf][0] = *reinterpret_cast<const float4*>(r_); A[buf][1] = *reinterpret_cast<const float4*>(r_ + 8); } while (0)
#define DG_LOADB(buf, s) do { const char* p_ = w + (size_t)((KS == 3 ? DG[(s) >> 3].tp : 0) * 8 + ((s) & 7)) * 2048;   \
        B[buf][0] = *reinterpret_cast<const float4*>(p_); B[buf][1] = *reinterpret_cast<const float4*>(p_ + 1024); } while (0)
    DG_LOADB(0, 0);
    DG_LOADB(1, 1);
    DG_LOADA(0, 0);
    // (nine taps spelled out, eight stages each: one 72-trip loop is beyond what the unroller takes, and every index here must be
    //  a compile-time constant - the rings live in registers)
#define DG_TAP(E)                                                                                                   \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                                 \
        constexpr int e_ = (E);                                                                                     \
        const int s = 8 * e_ + q;                                                                                   \
        if (s + 1 < NST) DG_LOADA((s + 1) & 1, s + 1);                                                              \
        if (s + 2 < NST) DG_LOADB((s + 2) & 3, s + 2);                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        {                                                                                                           \
            const float4 a0 = A[s & 1][0], a1 = A[s & 1][1], b0 = B[s & 3][0], b1 = B[s & 3][1];                    \
            acc0 = DG_MFMA(a0.x, b0.x, acc0); acc1 = DG_MFMA(a1.x, b1.x, acc1);                                     \
            acc0 = DG_MFMA(a0.y, b0.y, acc0); acc1 = DG_MFMA(a1.y, b1.y, acc1);                                     \
            acc0 = DG_MFMA(a0.z, b0.z, acc0); acc1 = DG_MFMA(a1.z, b1.z, acc1);                                     \
            acc0 = DG_MFMA(a0.w, b0.w, acc0); acc1 = DG_MFMA(a1.w, b1.w, acc1);                                     \
        }                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    }                                                                                                               \
    if (KS == 1 || DG[(E)].last) {                                                                                  \
        /* the class is complete: C-layout register r = tile row (r & 3) + 8 (r >> 2) + 4 half = (patch of the pair, class pixel) */ \
        constexpr int cls_ = DG_CLASS[(E)], py_ = cls_ >> 1, px_ = cls_ & 1;                                        \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                            \
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;                                                       \
            const long long p = p_first + 2 * pp + (rr >> 4);                                                       \
            const int oy = 2 * ((rr & 15) >> 2) + py_, ox = 2 * (rr & 3) + px_;                                     \
            if (p < a.n) a.dx[((size_t)p * 64 + oy * 8 + ox) * 64 + ct * 32 + i] = acc0[r] + acc1[r];               \
            acc0[r] = 0.0f;                                                                                         \
            acc1[r] = 0.0f;                                                                                         \
        }                                                                                                           \
    }
    DG_TAP(0)
    if constexpr (KS == 3) {
        DG_TAP(1) DG_TAP(2) DG_TAP(3) DG_TAP(4) DG_TAP(5) DG_TAP(6) DG_TAP(7) DG_TAP(8)
    } else {
        // the pixels no tap reaches
#pragma unroll
        for (int cls = 1; cls < 4; ++cls)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;
                const long long p = p_first + 2 * pp + (rr >> 4);
                const int oy = 2 * ((rr & 15) >> 2) + (cls >> 1), ox = 2 * (rr & 3) + (cls & 1);
                if (p < a.n) a.dx[((size_t)p * 64 + oy * 8 + ox) * 64 + ct * 32 + i] = 0.0f;
            }
    }
#undef DG_TAP
#undef DG_LOADA
#undef DG_LOADB
}

}  // namespace ipsx

using namespace ipsx;

IPSX_API int ipsx_conv2d_dgrad_s2_lds_nhwc_supported(int c_in, int c_out, int k, int stride, int pad, int h, int w) {
    // (of the FORWARD convolution: c_in -> c_out on h x w maps)
    if (!(c_in == 64 && c_out == 128 && stride == 2 && h == 8 && w == 8)) return 0;
    return ((k == 3 && pad == 1) || (k == 1 && pad == 0)) ? 1 : 0;
}

IPSX_API int ipsx_conv2d_dgrad_s2_lds_nhwc(const float* w_packed_dgrad, int k, const float* dy, float* dx, int64_t n, void* stream) {
    IPSX_REQUIRE(w_packed_dgrad && dy && dx && n >= 0 && (k == 3 || k == 1), "conv2d_dgrad_s2_lds_nhwc: bad arguments");
    if (n == 0) return IPSX_OK;
    DgradArgs a;
    a.dy = dy; a.dx = dx; a.wp = w_packed_dgrad; a.n = n;
    if (k == 3) dgrad_s2_lds_kernel<3><<<dim3((unsigned)cdiv(n, 4)), dim3(256), 0, as_stream(stream)>>>(a);
    else dgrad_s2_lds_kernel<1><<<dim3((unsigned)cdiv(n, 4)), dim3(256), 0, as_stream(stream)>>>(a);
    return launched("conv2d_dgrad_s2_lds_nhwc");
}
