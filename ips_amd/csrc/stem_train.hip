// stem_train.hip - the 1-channel 7x7 / 2 stem convolution of the 32-px trunk as a stand-alone layer for the TRAINING step
// (reference: training/iterative.py:158-163 -> architecture/ips_net.py:273 under net.train(); the stem is the `conv1` that
// ips_net.py:29-31 puts in front of torchvision's ResNet for 1-channel patches).
//
// The no-grad path never runs this layer alone (csrc/fused_trunk.hip fuses it with BatchNorm, ReLU and the max-pool); the
// training step needs its raw output for the batch-statistics BatchNorm behind it and ran it on conv_any_kernel - the generic
// direct convolution, 57 us for 1,024 patches, a fifth of the step's largest convolution's time for a hundredth of its work.
// Here: the fused trunk's stem WITHOUT its epilogue - wave = patch, the zero-padded 38 x 38 image in LDS, eight tiles of two
// output rows (32 pixels) x 64 channels, 25 v_mfma_f32_32x32x2_f32 per tile and channel half in the contract's k order
// (k = 8 g + 4 half + j <-> tap (k / 7, k % 7): the same bits as conv_any_kernel and the oracle), output channels-last.

#include "ipsx_common.h"

namespace ipsx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ST_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int STP = 38;                      // padded image: rows / columns 3..34 hold the patch
constexpr int ST_SLAB = STP * STP + 2;       // floats per wave (+ 2: the last tap of the last pixel reads one row further)

// stats (may be NULL) / shift: the BatchNorm batch statistics of the output off the accumulators, per workgroup = slab of 4
// patches: stats[blockIdx][0 | 1][64] = sum (y - shift), sum (y - shift)^2 - see ipsx_conv2d_lds_nhwc_stats
__global__ __launch_bounds__(256, 2) void stem7x7s2_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                float* __restrict__ y, long long n, const float* __restrict__ shift,
                                                                float* __restrict__ stats) {
    __shared__ float lds[4 * ST_SLAB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, half = lane >> 5;
    const long long p = (long long)blockIdx.x * 4 + wave;
    const bool live = p < n;
    if (!live && !stats) return;                                          // (wave-private slab: no workgroup barrier without statistics)
    float* S = lds + wave * ST_SLAB;
    float s1[2] = {0.0f, 0.0f}, s2[2] = {0.0f, 0.0f};
    if (live) {
    const float k0s = (stats && shift) ? shift[i] : 0.0f, k1s = (stats && shift) ? shift[32 + i] : 0.0f;
    for (int e = lane; e < ST_SLAB; e += 64) S[e] = 0.0f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float4* src = reinterpret_cast<const float4*>(x + (size_t)p * 1024);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = (k * 64 + lane) * 4, r = e >> 5, c = e & 31;       // four pixels of row r
        const float4 v = src[k * 64 + lane];
        float* d = S + (r + 3) * STP + c + 3;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    // weights of the 25 steps that carry taps (k-groups 0..6 of the packed stream: K = 49 padded to 56), both channel halves
    float bw0[28], bw1[28];
    {
        const float4* wp0 = reinterpret_cast<const float4*>(wp) + lane;
        const float4* wp1 = wp0 + 7 * 64;
#pragma unroll
        for (int kg = 0; kg < 7; ++kg) {
            const float4 v0 = wp0[kg * 64], v1 = wp1[kg * 64];
            bw0[4 * kg] = v0.x; bw0[4 * kg + 1] = v0.y; bw0[4 * kg + 2] = v0.z; bw0[4 * kg + 3] = v0.w;
            bw1[4 * kg] = v1.x; bw1[4 * kg + 1] = v1.y; bw1[4 * kg + 2] = v1.z; bw1[4 * kg + 3] = v1.w;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int ox = i & 15;
    float* out = y + (size_t)p * 256 * 64;
#pragma unroll 1
    for (int t = 0; t < 8; ++t) {
        // lane's output pixel: row 2 t + (i >> 4), column ox.  Step (g, j) feeds k = 8 g + 4 half + j -> tap (k / 7, k % 7): the
        // upper half's tap is 4 columns right of the lower half's or, where that leaves the 7-wide row, 3 left and one row down
        const int oy = 2 * t + (i >> 4);
        const float* base = S + (2 * oy) * STP + 2 * ox;
        const float* baseN = base + half * 4;
        const float* baseW = base + half * (STP - 3);
        float av[25];
#pragma unroll
        for (int st = 0; st < 25; ++st) {
            const int k0 = 8 * (st >> 2) + (st & 3), ky0 = k0 / 7, kx0 = k0 % 7;
            av[st] = (kx0 <= 2) ? baseN[ky0 * STP + kx0] : baseW[ky0 * STP + kx0];
        }
        av[24] = half ? 0.0f : av[24];                                   // k = 52 does not exist (zero weight): a clean 0
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
#pragma unroll
        for (int st = 0; st < 25; ++st) {
            acc0 = ST_MFMA(av[st], bw0[st], acc0);
            acc1 = ST_MFMA(av[st], bw1[st], acc1);
        }
        // C layout: register r = tile row (r & 3) + 8 (r >> 2) + 4 half, lane i = channel; tile row q = pixel 32 t + q
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pix = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            out[pix * 64 + i] = acc0[r];
            out[pix * 64 + 32 + i] = acc1[r];
        }
        if (stats) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d0 = acc0[r] - k0s, d1 = acc1[r] - k1s;
                s1[0] = s1[0] + d0; s2[0] = __builtin_fmaf(d0, d0, s2[0]);
                s1[1] = s1[1] + d1; s2[1] = __builtin_fmaf(d1, d1, s2[1]);
            }
        }
    }
    }
    if (stats) {
        // the lane halves hold different pixels of the same channel; then the slab's four patches in patch order
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
            for (int w = 1; w < 4; ++w) { t1 = t1 + lds[w * ST_SLAB + lane]; t2 = t2 + lds[w * ST_SLAB + 64 + lane]; }
            stats[(size_t)blockIdx.x * 128 + lane] = t1;
            stats[(size_t)blockIdx.x * 128 + 64 + lane] = t2;
        }
    }
}

}  // namespace ipsx

using namespace ipsx;

IPSX_API int ipsx_stem7x7s2_nhwc_supported(int c_in, int c_out, int kh, int kw, int stride, int pad, int h, int w) {
    return (c_in == 1 && c_out == 64 && kh == 7 && kw == 7 && stride == 2 && pad == 3 && h == 32 && w == 32) ? 1 : 0;
}

IPSX_API int ipsx_stem7x7s2_nhwc(const ipsx_conv* cv, const float* x, float* y, int64_t n, const float* shift, float* partial,
                                 void* stream) {
    IPSX_REQUIRE(cv && cv->w_packed && x && y && n >= 0, "stem7x7s2_nhwc: bad arguments");
    IPSX_REQUIRE(ipsx_stem7x7s2_nhwc_supported(cv->c_in, cv->c_out, cv->kh, cv->kw, cv->stride, cv->pad, 32, 32),
                 "stem7x7s2_nhwc: the 1 -> 64 channel 7x7 / 2 stem on 32 x 32 patches only");
    if (n == 0) return IPSX_OK;
    stem7x7s2_nhwc_kernel<<<dim3((unsigned)cdiv(n, 4)), dim3(256), 0, as_stream(stream)>>>(x, cv->w_packed, y, (long long)n, partial ? shift : nullptr,
                                                                                               partial);
    return launched("stem7x7s2_nhwc");
}
