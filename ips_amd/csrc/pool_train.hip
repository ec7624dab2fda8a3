// pool_train.hip - the backward pass of the trunk's nn.MaxPool2d(3, 2, 1) on channels-last activations, for the TRAINING step
// (reference: loss.backward() of training/iterative.py:157-163 through the `maxpool` of the torchvision ResNet that
// architecture/ips_net.py:17-52 builds; the forward pass is ipsx_maxpool_3x3s2_nhwc of csrc/conv_nhwc.hip).
//
// Same result as ATen's max_pool2d_with_indices_backward, bit for bit: the gradient of a window goes to its FIRST maximum in
// row-major scan order (`val > maxval || isnan(val)` replaces; the scan starts at the window's first pixel inside the map) -
// with post-ReLU inputs most windows have several equal zeros, so the rule matters.  No index tensor: the maximum's position
// is found again from x.  ATen: a zero fill of dx and a scatter with atomics, 10 + 58 us at 1,024 maps of 16 x 16 x 64; here a
// workgroup holds 32 channels of one map in LDS, finds every window's first maximum once, and every input pixel then adds the
// gradients of the (one, two or four) windows that chose it in the windows' row-major order - ATen's order of additions - and
// is written once.

#include "ipsx_common.h"

namespace ipsx {

constexpr int PT_H = 16, PT_HO = 8, PT_C = 32;       // map size, pooled size, channels per workgroup

__global__ __launch_bounds__(256) void maxpool_3x3s2_bwd_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     float* __restrict__ dx, int c) {
    __shared__ __attribute__((aligned(16))) float xs[PT_H * PT_H * PT_C];
    __shared__ __attribute__((aligned(16))) float gs[PT_HO * PT_HO * PT_C];     // dy of the 64 windows
    __shared__ __attribute__((aligned(16))) unsigned char arg[PT_HO * PT_HO * PT_C];   // pixel (0..255) of every window's first maximum
    const int chunks = c / PT_C;
    const size_t map = (size_t)blockIdx.x / chunks;
    const int c0 = (int)(blockIdx.x % chunks) * PT_C;
    const int t = threadIdx.x, j = t & 7;
#pragma unroll
    for (int k = 0; k < 8; ++k) {                                    // 256 pixels x 32 channels: 8 float4 per pixel
        const int px = (t >> 3) + 32 * k;
        *reinterpret_cast<float4*>(xs + px * PT_C + 4 * j) =
            *reinterpret_cast<const float4*>(x + (map * (PT_H * PT_H) + px) * c + c0 + 4 * j);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {                                    // 64 windows x 32 channels
        const int w = (t >> 3) + 32 * k;
        *reinterpret_cast<float4*>(gs + w * PT_C + 4 * j) =
            *reinterpret_cast<const float4*>(dy + (map * (PT_HO * PT_HO) + w) * c + c0 + 4 * j);
    }
    __syncthreads();
    // pass A: every window's first maximum (ATen's scan: row-major from the first pixel inside the map, `>` or NaN replaces);
    // a thread takes four channels of a window
    const float4* xs4 = reinterpret_cast<const float4*>(xs);
    const float4* gs4 = reinterpret_cast<const float4*>(gs);
    unsigned* arg4 = reinterpret_cast<unsigned*>(arg);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int w = (t >> 3) + 32 * k, oy = w >> 3, ox = w & 7;
        const int y0 = max(2 * oy - 1, 0), x0 = max(2 * ox - 1, 0);
        const int y1 = min(2 * oy + 1, PT_H - 1), x1 = min(2 * ox + 1, PT_H - 1);
        float m[4];
        unsigned am[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { m[q] = -__builtin_huge_valf(); am[q] = (unsigned)(y0 * PT_H + x0); }
        for (int iy = y0; iy <= y1; ++iy)
            for (int ix = x0; ix <= x1; ++ix) {
                const unsigned px = (unsigned)(iy * PT_H + ix);
                const float4 v4 = xs4[px * (PT_C / 4) + j];
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (v[q] > m[q] || v[q] != v[q]) { m[q] = v[q]; am[q] = px; }
            }
        arg4[w * (PT_C / 4) + j] = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
    }
    __syncthreads();
    // pass B: every input pixel adds the gradients of the windows that chose it, in the windows' row-major order (ATen's
    // order of additions: pixels on odd rows / columns lie in two windows each way); four channels per thread, 16-byte stores
#pragma unroll 2
    for (int k = 0; k < 8; ++k) {
        const int px = (t >> 3) + 32 * k, iy = px >> 4, ix = px & 15;
        const int oy0 = iy >> 1, oy1 = min((iy + 1) >> 1, PT_HO - 1), ox0 = ix >> 1, ox1 = min((ix + 1) >> 1, PT_HO - 1);
        float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int oy = oy0; oy <= oy1; ++oy)
            for (int ox = ox0; ox <= ox1; ++ox) {
                const int w = oy * PT_HO + ox;
                const unsigned am = arg4[w * (PT_C / 4) + j];
                const float4 g4 = gs4[w * (PT_C / 4) + j];
                const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (((am >> (8 * q)) & 255u) == (unsigned)px) sum[q] = sum[q] + g[q];
            }
        *reinterpret_cast<float4*>(dx + (map * (PT_H * PT_H) + px) * c + c0 + 4 * j) = make_float4(sum[0], sum[1], sum[2], sum[3]);
    }
}

}  // namespace ipsx

using namespace ipsx;

IPSX_API int ipsx_maxpool_3x3s2_bwd_nhwc_supported(int c, int h, int w) {
    return (h == PT_H && w == PT_H && c > 0 && c % PT_C == 0) ? 1 : 0;
}

IPSX_API int ipsx_maxpool_3x3s2_bwd_nhwc(const float* x, const float* dy, float* dx, int64_t n, int c, int h, int w, void* stream) {
    IPSX_REQUIRE(x && dy && dx && n >= 0, "maxpool_bwd_nhwc: bad arguments");
    IPSX_REQUIRE(ipsx_maxpool_3x3s2_bwd_nhwc_supported(c, h, w), "maxpool_bwd_nhwc: 16 x 16 maps, a multiple of 32 channels (got %d x %d x %d)",
                 h, w, c);
    IPSX_REQUIRE(n * (c / PT_C) < ((int64_t)1 << 31), "maxpool_bwd_nhwc: too many maps");
    if (n == 0) return IPSX_OK;
    maxpool_3x3s2_bwd_nhwc_kernel<<<dim3((unsigned)(n * (c / PT_C))), dim3(256), 0, as_stream(stream)>>>(x, dy, dx, c);
    return launched("maxpool_bwd_nhwc");
}
