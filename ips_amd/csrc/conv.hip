// conv.hip - the patch encoder of IPSNet (reference architecture/ips_net.py:17-52)
// as implicit-GEMM convolutions on the fp32 matrix cores of gfx950.
//
// GEMM view of a convolution: M = (patch, oy, ox) output pixels, N = C_out,
// K = (ky, kx, c) in tap-major order.  One wavefront owns a 64(M) x 64(N) output
// tile = 2x2 accumulators of v_mfma_f32_32x32x2_f32; a workgroup is 4 wavefronts
// stacked along M (256 x 64).  v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain in
// k order, so the result is the oracle's (oracle/ips_oracle.cpp orc_conv2d_affine)
// bit for bit.
//
//  * A operand (activations, NCHW fp32): lane l holds pixel (l & 31) of the tile
//    and, in step j of k-group g, k = 8g + 4*(l >> 5) + j; it is fetched straight from global/L2 with the tap
//    offset applied (im2col on the fly); padded taps read a valid address and are
//    zeroed by a select, so there is no divergent control flow around the MFMAs.
//  * B operand (weights): pre-packed by ipsx_pack_conv_weight into the exact
//    per-lane order [C_out/32][K/8][64 lanes][4 k-steps], so one 16-byte load per
//    lane (1 KiB per wavefront, fully coalesced, L2 resident) feeds 4 k-steps.
//  * epilogue: BatchNorm affine, residual add, ReLU in registers; each lane owns 4
//    consecutive pixels of one channel per accumulator quad -> 16-byte stores.
//
// Algorithmic cost: 2*K*C_out flop per output pixel; bytes: input + output once.

#include <algorithm>

#include "ipsx_common.h"
#include "ipsx_math.h"

namespace ipsx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* x;
    const float* wp;
    const float* alpha;
    const float* shift;
    const float* res;
    float* y;
    unsigned m_total;     // n * ho * wo  (< 2^31, the driver chunks)
    int c_in, h, w, c_out, ho, wo, kh, kw, stride, pad, relu;
    int kgs;              // packed k-groups = ceil(K/8)
    int out_nhwc;         // write y as [pixel][c_out] (channels-last) instead of NCHW
};

// ------------------------------------------------------------------ packing
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, int c_out, int c_in, int kh, int kw,
                                        int kgs, size_t total, float* __restrict__ packed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 3);
    const int lane = (int)((i >> 2) & 63);
    const size_t g = i >> 8;               // nt * kgs + kg
    const int kg = (int)(g % kgs);
    const int nt = (int)(g / kgs);
    const int n = nt * 32 + (lane & 31);
    const int k = kg * 8 + 4 * (lane >> 5) + j;          // lane half h, step j  <->  k = 8g + 4h + j
    const int K = kh * kw * c_in;
    float v = 0.0f;
    if (n < c_out && k < K) {
        const int tap = k / c_in, c = k - tap * c_in;
        v = w[((size_t)n * c_in + c) * kh * kw + tap];
    }
    packed[i] = v;
}

// the same from ANY strided view of the weight (element strides, possibly negative): a channels-last tensor as it lies,
// or - base at the last tap, tap strides negated, channel strides swapped - the weights rotated by 180 degrees and
// transposed that turn a convolution into its own data gradient (training step, csrc/conv_wgrad.hip)
__global__ void pack_conv_weight_strided_kernel(const float* __restrict__ w, long long base, int c_out, int c_in, int kh, int kw,
                                                long long sn, long long sc, long long sky, long long skx, int kgs, size_t total,
                                                float* __restrict__ packed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 3);
    const int lane = (int)((i >> 2) & 63);
    const size_t g = i >> 8;
    const int kg = (int)(g % kgs);
    const int nt = (int)(g / kgs);
    const int n = nt * 32 + (lane & 31);
    const int k = kg * 8 + 4 * (lane >> 5) + j;
    const int K = kh * kw * c_in;
    float v = 0.0f;
    if (n < c_out && k < K) {
        const int tap = k / c_in, c = k - tap * c_in;
        const int ky = tap / kw, kx = tap - ky * kw;
        v = w[base + n * sn + c * sc + ky * sky + kx * skx];
    }
    packed[i] = v;
}

// several strided packings as ONE launch (the training step packs every convolution's weights, forward and data-gradient
// form, after each optimizer step: 19 launches of ~4 us were 3 % of the step's device time); a block finds its job in the
// table of first blocks
constexpr int PACK_BATCH_MAX = 32;
struct PackBatch {
    const float* w[PACK_BATCH_MAX];
    float* packed[PACK_BATCH_MAX];
    long long base[PACK_BATCH_MAX], sn[PACK_BATCH_MAX], sc[PACK_BATCH_MAX], sky[PACK_BATCH_MAX], skx[PACK_BATCH_MAX];
    int c_out[PACK_BATCH_MAX], c_in[PACK_BATCH_MAX], kh[PACK_BATCH_MAX], kw[PACK_BATCH_MAX];
    unsigned first_block[PACK_BATCH_MAX + 1];
    int n;
};

__global__ void pack_conv_weight_batch_kernel(PackBatch b) {
    int job = 0;
    while (job + 1 < b.n && blockIdx.x >= b.first_block[job + 1]) ++job;          // (block-uniform)
    const int c_out = b.c_out[job], c_in = b.c_in[job], kh = b.kh[job], kw = b.kw[job];
    const int K = kh * kw * c_in, kgs = (K + 7) / 8;
    const size_t total = (size_t)((c_out + 31) / 32) * kgs * 256;
    const size_t i = (size_t)(blockIdx.x - b.first_block[job]) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 3);
    const int lane = (int)((i >> 2) & 63);
    const size_t g = i >> 8;
    const int kg = (int)(g % kgs);
    const int nt = (int)(g / kgs);
    const int n = nt * 32 + (lane & 31);
    const int k = kg * 8 + 4 * (lane >> 5) + j;
    float v = 0.0f;
    if (n < c_out && k < K) {
        const int tap = k / c_in, c = k - tap * c_in;
        const int ky = tap / kw, kx = tap - ky * kw;
        v = b.w[job][b.base[job] + n * b.sn[job] + c * b.sc[job] + ky * b.sky[job] + kx * b.skx[job]];
    }
    b.packed[job][i] = v;
}

__global__ void bn_affine_kernel(const float* gamma, const float* beta, const float* mean, const float* var,
                                 const float* lin_bias, float eps, int c, float* alpha, float* shift) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    float invstd = 1.0f / __builtin_sqrtf(var[i] + eps);
    float a = gamma[i] * invstd;
    float m = mean[i] * a;
    float s = beta[i] - m;
    if (lin_bias) s = __builtin_fmaf(lin_bias[i], a, s);
    alpha[i] = a;
    shift[i] = s;
}

// ------------------------------------------------------------------ conv kernels
struct PixelCtx {
    const float* base;    // first element of this pixel's patch
    int iy0, ix0;         // top-left input coordinate of the receptive field
    bool valid;
};

__device__ __forceinline__ PixelCtx pixel_ctx(const ConvArgs& a, unsigned m) {
    PixelCtx p;
    p.valid = m < a.m_total;
    const unsigned howo = (unsigned)(a.ho * a.wo);
    const unsigned mm = p.valid ? m : 0u;
    const unsigned img = mm / howo, pix = mm - img * howo;
    const unsigned oy = pix / (unsigned)a.wo, ox = pix - oy * (unsigned)a.wo;
    p.base = a.x + (size_t)img * a.c_in * a.h * a.w;
    p.iy0 = (int)oy * a.stride - a.pad;
    p.ix0 = (int)ox * a.stride - a.pad;
    return p;
}

__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[2][2], unsigned m_base,
                                              int n_base, int lane) {
    const unsigned howo = (unsigned)(a.ho * a.wo);
    const int half = lane >> 5;
    const bool vec4 = (howo & 3u) == 0 && !a.out_nhwc;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n_base + nt * 32 + (lane & 31);
        if (n >= a.c_out) continue;
        const float al = a.alpha ? a.alpha[n] : 1.0f;
        const float sh = a.shift ? a.shift[n] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned m0 = m_base + mt * 32 + 8 * q + 4 * half;   // rows m0..m0+3 = regs 4q..4q+3
                if (m0 >= a.m_total) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = acc[mt][nt][4 * q + r];
                    if (a.alpha) t = __builtin_fmaf(t, al, sh);
                    else if (a.shift) t = t + sh;
                    v[r] = t;
                }
                if (vec4) {   // 4 consecutive pixels of one patch, 16-byte aligned
                    const unsigned img = m0 / howo, pix = m0 - img * howo;
                    const size_t idx = ((size_t)img * a.c_out + n) * howo + pix;
                    if (a.res) {
                        const float4 rr = *reinterpret_cast<const float4*>(a.res + idx);
                        v[0] = v[0] + rr.x; v[1] = v[1] + rr.y; v[2] = v[2] + rr.z; v[3] = v[3] + rr.w;
                    }
                    if (a.relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
                    }
                    *reinterpret_cast<float4*>(a.y + idx) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned m = m0 + r;
                        if (m >= a.m_total) continue;
                        const unsigned img = m / howo, pix = m - img * howo;
                        const size_t idx = a.out_nhwc ? (size_t)m * a.c_out + n : ((size_t)img * a.c_out + n) * howo + pix;
                        float t = v[r];
                        if (a.res) t = t + a.res[idx];
                        if (a.relu) t = t > 0.0f ? t : 0.0f;
                        a.y[idx] = t;
                    }
                }
            }
        }
    }
}

// Any C_in (the 1- and 3-channel stems, and the public NCHW entry point).  The (channel, ky, kx) of every k is decoded ONCE per workgroup into an LDS
// table (offset inside the image, tap position), so the inner loop has no integer divisions: per element one table
// read (the lanes of a half share the address), the bounds test, the gather.
__global__ __launch_bounds__(256) void conv_any_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) int ktab[];     // [kgs*8][2]: {c*hw + ky*w + kx, ky | kx << 8 | valid << 16}
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    const int hw = a.h * a.w;
    const int K = a.kh * a.kw * a.c_in;
    for (int k = threadIdx.x; k < a.kgs * 8; k += 256) {
        const bool kin = k < K;
        const int kk = kin ? k : 0;
        const int tap = kk / a.c_in, c = kk - tap * a.c_in;
        const int ky = tap / a.kw, kx = tap - ky * a.kw;
        ktab[2 * k] = c * hw + ky * a.w + kx;
        ktab[2 * k + 1] = ky | (kx << 8) | ((kin ? 1 : 0) << 16);
    }
    __syncthreads();
    const unsigned m_base = blockIdx.x * 256u + wave * 64u;
    if (m_base >= a.m_total) return;
    const int n_base = blockIdx.y * 64;
    const PixelCtx p0 = pixel_ctx(a, m_base + (lane & 31));
    const PixelCtx p1 = pixel_ctx(a, m_base + 32 + (lane & 31));
    // offset of (channel 0, tap 0) of each pixel's window inside its image; table offsets are added to it
    const int o0 = p0.iy0 * a.w + p0.ix0, o1 = p1.iy0 * a.w + p1.ix0;
    const float4* wp0 = reinterpret_cast<const float4*>(a.wp) + ((size_t)(blockIdx.y * 2) * a.kgs) * 64 + lane;
    const float4* wp1 = wp0 + (size_t)a.kgs * 64;
    const bool n1 = n_base + 32 < a.c_out;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Which taps of a pixel's window lie inside the image, as one bit per kernel row / column (kh, kw <= 31): the bounds
    // test of an operand element is then two shifts and an AND instead of two adds and two compares - the gather is
    // VALU-bound (one scalar load per element), and every instruction it saves is matrix-pipe time.
    unsigned rm0 = 0, cm0 = 0, rm1 = 0, cm1 = 0;
    for (int t = 0; t < a.kh; ++t) {
        rm0 |= (p0.valid && (unsigned)(p0.iy0 + t) < (unsigned)a.h) ? (1u << t) : 0u;
        rm1 |= (p1.valid && (unsigned)(p1.iy0 + t) < (unsigned)a.h) ? (1u << t) : 0u;
    }
    for (int t = 0; t < a.kw; ++t) {
        cm0 |= ((unsigned)(p0.ix0 + t) < (unsigned)a.w) ? (1u << t) : 0u;
        cm1 |= ((unsigned)(p1.ix0 + t) < (unsigned)a.w) ? (1u << t) : 0u;
    }
    for (int kg = 0; kg < a.kgs; ++kg) {
        const float4 b0 = *wp0;
        const float4 b1 = n1 ? *wp1 : make_float4(0.f, 0.f, 0.f, 0.f);
        wp0 += 64; wp1 += 64;
        const float bb0[4] = {b0.x, b0.y, b0.z, b0.w};
        const float bb1[4] = {b1.x, b1.y, b1.z, b1.w};
        const int4 ta = *reinterpret_cast<const int4*>(ktab + 2 * (kg * 8 + 4 * half));          // entries j = 0, 1
        const int4 tb = *reinterpret_cast<const int4*>(ktab + 2 * (kg * 8 + 4 * half) + 4);      // entries j = 2, 3
        const int off[4] = {ta.x, ta.z, tb.x, tb.z}, pk[4] = {ta.y, ta.w, tb.y, tb.w};
        float a0[4], a1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned ky = (unsigned)pk[j] & 0xFFu, kx = ((unsigned)pk[j] >> 8) & 0xFFu, kin = (unsigned)pk[j] >> 16;
            const bool ok0 = ((rm0 >> ky) & (cm0 >> kx) & kin) != 0u;
            const bool ok1 = ((rm1 >> ky) & (cm1 >> kx) & kin) != 0u;
            const float v0 = p0.base[ok0 ? o0 + off[j] : 0], v1 = p1.base[ok1 ? o1 + off[j] : 0];
            a0[j] = ok0 ? v0 : 0.0f;
            a1[j] = ok1 ? v1 : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bb0[j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bb1[j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bb0[j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bb1[j], acc[1][1], 0, 0, 0);
        }
    }
    conv_epilogue(a, acc, m_base, n_base, lane);
}

// ------------------------------------------------------------------ pooling
__global__ void maxpool_3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int h,
                                     int w, int ho, int wo) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % wo);
    const size_t t = i / wo;
    const int oy = (int)(t % ho);
    const size_t pc = t / ho;
    const float* src = x + pc * (size_t)h * w;
    float m = -__builtin_huge_valf();
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
            if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
            m = nanmax(m, src[iy * w + ix]);
        }
    y[i] = m;
}

__global__ void avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int hw) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float* src = x + i * (size_t)hw;
    float s = 0.0f;
    for (int j = 0; j < hw; ++j) s = s + src[j];
    y[i] = s / (float)hw;
}

static int check_conv(const ipsx_conv* cv) {
    IPSX_REQUIRE(cv && cv->w_packed, "conv: missing packed weights");
    IPSX_REQUIRE(cv->c_in > 0 && cv->c_out > 0 && cv->kh > 0 && cv->kw > 0 && cv->stride > 0 && cv->pad >= 0,
                 "conv: bad geometry c_in=%d c_out=%d k=%dx%d s=%d p=%d", cv->c_in, cv->c_out, cv->kh, cv->kw,
                 cv->stride, cv->pad);
    return IPSX_OK;
}

}  // namespace ipsx

using namespace ipsx;

IPSX_API size_t ipsx_packed_conv_weight_elems(int c_out, int c_in, int kh, int kw) {
    const size_t nts = (size_t)cdiv(c_out, 32), kgs = (size_t)cdiv((int64_t)kh * kw * c_in, 8);
    return nts * kgs * 256;
}

IPSX_API int ipsx_pack_conv_weight(const float* w, int c_out, int c_in, int kh, int kw, float* packed,
                                   void* stream) {
    IPSX_REQUIRE(w && packed && c_out > 0 && c_in > 0 && kh > 0 && kw > 0, "pack_conv_weight: bad arguments");
    const size_t total = ipsx_packed_conv_weight_elems(c_out, c_in, kh, kw);
    const int kgs = (int)cdiv((int64_t)kh * kw * c_in, 8);
    pack_conv_weight_kernel<<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream)>>>(
        w, c_out, c_in, kh, kw, kgs, total, packed);
    return launched("pack_conv_weight");
}

IPSX_API int ipsx_pack_conv_weight_strided(const float* w, int64_t base, int c_out, int c_in, int kh, int kw, int64_t s_out,
                                           int64_t s_in, int64_t s_ky, int64_t s_kx, float* packed, void* stream) {
    IPSX_REQUIRE(w && packed && c_out > 0 && c_in > 0 && kh > 0 && kw > 0, "pack_conv_weight_strided: bad arguments");
    const size_t total = ipsx_packed_conv_weight_elems(c_out, c_in, kh, kw);
    const int kgs = (int)cdiv((int64_t)kh * kw * c_in, 8);
    pack_conv_weight_strided_kernel<<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream)>>>(
        w, base, c_out, c_in, kh, kw, s_out, s_in, s_ky, s_kx, kgs, total, packed);
    return launched("pack_conv_weight_strided");
}

IPSX_API int ipsx_pack_conv_weights_batch(const ipsx_pack_job* jobs, int n_jobs, void* stream) {
    IPSX_REQUIRE(jobs && n_jobs > 0 && n_jobs <= PACK_BATCH_MAX, "pack_conv_weights_batch: 1..%d jobs", PACK_BATCH_MAX);
    PackBatch b;
    unsigned blocks = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const ipsx_pack_job& j = jobs[k];
        IPSX_REQUIRE(j.w && j.packed && j.c_out > 0 && j.c_in > 0 && j.kh > 0 && j.kw > 0, "pack_conv_weights_batch: bad job %d", k);
        b.w[k] = j.w; b.packed[k] = j.packed; b.base[k] = j.base;
        b.sn[k] = j.s_out; b.sc[k] = j.s_in; b.sky[k] = j.s_ky; b.skx[k] = j.s_kx;
        b.c_out[k] = j.c_out; b.c_in[k] = j.c_in; b.kh[k] = j.kh; b.kw[k] = j.kw;
        b.first_block[k] = blocks;
        blocks += (unsigned)cdiv(ipsx_packed_conv_weight_elems(j.c_out, j.c_in, j.kh, j.kw), 256);
    }
    b.first_block[n_jobs] = blocks;
    b.n = n_jobs;
    pack_conv_weight_batch_kernel<<<dim3(blocks), dim3(256), 0, as_stream(stream)>>>(b);
    return launched("pack_conv_weights_batch");
}

IPSX_API int ipsx_bn_affine(const float* gamma, const float* beta, const float* mean, const float* var,
                            const float* lin_bias, float eps, int c, float* alpha, float* shift, void* stream) {
    IPSX_REQUIRE(gamma && beta && mean && var && alpha && shift && c > 0, "bn_affine: bad arguments");
    bn_affine_kernel<<<dim3((unsigned)cdiv(c, 256)), dim3(256), 0, as_stream(stream)>>>(
        gamma, beta, mean, var, lin_bias, eps, c, alpha, shift);
    return launched("bn_affine");
}

namespace ipsx {
int conv2d_affine_impl(const ipsx_conv* cv, const float* x, const float* residual, float* y, int64_t n, int h,
                       int w, int relu, int out_nhwc, void* stream);
}

IPSX_API int ipsx_conv2d_affine(const ipsx_conv* cv, const float* x, const float* residual, float* y,
                                int64_t n, int h, int w, int relu, void* stream) {
    return conv2d_affine_impl(cv, x, residual, y, n, h, w, relu, 0, stream);
}

IPSX_API int ipsx_conv2d_affine_to_nhwc(const ipsx_conv* cv, const float* x, const float* residual, float* y,
                                        int64_t n, int h, int w, int relu, void* stream) {
    return conv2d_affine_impl(cv, x, residual, y, n, h, w, relu, 1, stream);
}

int ipsx::conv2d_affine_impl(const ipsx_conv* cv, const float* x, const float* residual, float* y, int64_t n,
                             int h, int w, int relu, int out_nhwc, void* stream) {
    IPSX_TRY(check_conv(cv));
    IPSX_REQUIRE(x && y && n >= 0 && h > 0 && w > 0, "conv2d_affine: bad arguments");
    if (n == 0) return IPSX_OK;
    const int ho = conv_out(h, cv->kh, cv->stride, cv->pad), wo = conv_out(w, cv->kw, cv->stride, cv->pad);
    IPSX_REQUIRE(ho > 0 && wo > 0, "conv2d_affine: empty output (%dx%d input, %dx%d kernel)", h, w, cv->kh, cv->kw);
    IPSX_REQUIRE(cv->kh <= 31 && cv->kw <= 31, "conv2d_affine: kernel %dx%d larger than 31x31", cv->kh, cv->kw);
    const int64_t howo = (int64_t)ho * wo;
    // keep every launch below 2^31 output pixels / input elements per image group
    const int64_t per = std::max<int64_t>(1, std::min<int64_t>(n, ((int64_t)1 << 30) / std::max<int64_t>(howo, 1)));
    for (int64_t i0 = 0; i0 < n; i0 += per) {
        const int64_t cnt = std::min(per, n - i0);
        ConvArgs a;
        a.x = x + (size_t)i0 * cv->c_in * h * w;
        a.y = y + (size_t)i0 * cv->c_out * howo;
        a.res = residual ? residual + (size_t)i0 * cv->c_out * howo : nullptr;
        a.wp = cv->w_packed; a.alpha = cv->alpha; a.shift = cv->shift;
        a.m_total = (unsigned)(cnt * howo);
        a.c_in = cv->c_in; a.h = h; a.w = w; a.c_out = cv->c_out; a.ho = ho; a.wo = wo;
        a.kh = cv->kh; a.kw = cv->kw; a.stride = cv->stride; a.pad = cv->pad; a.relu = relu;
        a.kgs = (int)cdiv((int64_t)cv->kh * cv->kw * cv->c_in, 8);
        a.out_nhwc = out_nhwc;
        dim3 grid((unsigned)cdiv(a.m_total, 256), (unsigned)cdiv(cv->c_out, 64));
        const size_t table = (size_t)a.kgs * 8 * 2 * sizeof(int);
        IPSX_REQUIRE(table <= 64 * 1024, "conv2d_affine: K = %d does not fit the k table", a.kgs * 8);
        conv_any_kernel<<<grid, dim3(256), table, as_stream(stream)>>>(a);
        IPSX_TRY(launched("conv2d_affine"));
    }
    return IPSX_OK;
}

IPSX_API int ipsx_maxpool_3x3s2(const float* x, float* y, int64_t n, int c, int h, int w, void* stream) {
    IPSX_REQUIRE(x && y && n >= 0 && c > 0 && h > 0 && w > 0, "maxpool: bad arguments");
    const int ho = conv_out(h, 3, 2, 1), wo = conv_out(w, 3, 2, 1);
    const size_t total = (size_t)n * c * ho * wo;
    if (!total) return IPSX_OK;
    maxpool_3x3s2_kernel<<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream)>>>(x, y, total, h, w,
                                                                                                ho, wo);
    return launched("maxpool");
}

IPSX_API int ipsx_avgpool(const float* x, float* y, int64_t n, int c, int hw, void* stream) {
    IPSX_REQUIRE(x && y && n >= 0 && c > 0 && hw > 0, "avgpool: bad arguments");
    const size_t total = (size_t)n * c;
    if (!total) return IPSX_OK;
    avgpool_kernel<<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream)>>>(x, y, total, hw);
    return launched("avgpool");
}
