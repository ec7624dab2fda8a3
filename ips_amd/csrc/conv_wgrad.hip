// conv_wgrad.hip - the weight gradient of a convolution on channels-last activations, for the training step's encoder
// (SURVEY.md section 8 f, N-b; reference: loss.backward() of training/iterative.py:157-163 through the torchvision
// BasicBlocks of architecture/ips_net.py:264-283, which stock PyTorch hands to MIOpen).
//
//     dW[co][ky][kx][ci] = sum over (img, oy, ox) of dy[img, oy, ox, co] * x[img, s oy + ky - p, s ox + kx - p, ci]
//
// is the GEMM  C (C_out x K) = dy^T (C_out x P) . im2col(x) (P x K),  K = kh kw C_in, with the REDUCTION over the P output
// pixels - on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: two pixels per instruction).  Both operands are read where
// they lie: a lane's A value is dy[pixel][co0 + i], its B value x[shifted pixel][ci0 + i] - 128 contiguous bytes per
// half-wave and operand, raw buffer loads whose bounds check turns padding taps and ragged ends into zeros (no im2col
// buffer, no transposition).  One fp32 MFMA is 64 matrix-pipe cycles, so one 4-byte load per MFMA and lane is far below
// what the memory path delivers; operands are requested three steps ahead (ring of 4).
//
// Work split: a workgroup owns one 64 x 64 block of C - (tap, 64 input channels, 64 output channels) - for a contiguous
// range of images ("split"); each of its 8 wavefronts keeps one PIXEL PAIR of the map and walks the images (offsets and
// padding tests are loop invariants: a step is four loads with a scalar image offset and four MFMAs), accumulating the
// whole 64 x 64 block (2 x 2 accumulator tiles); the wavefronts' blocks are added through LDS in wavefront order.  Splits are sized so that blocks x splits ~ the compute units; the per-split partial blocks go to a
// caller workspace and conv_wgrad_reduce_kernel adds them in split order: the result is deterministic (same bits run to
// run), in the memory order [co][ky][kx][ci] - the layout of a channels-last weight tensor.

#include <algorithm>

#include "ipsx_common.h"
#include "ipsx_math.h"

namespace ipsx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define WG_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr unsigned kWgOob = 0x80000000u;
constexpr int WG_WAVES = 8;

struct WgradArgs {
    const float* x;                // (n, h, w, c_in)
    const float* dy;               // (n, ho, wo, c_out)
    float* partial;                // [splits][c_out][K]
    unsigned x_bytes, dy_bytes;
    int n, h, w, c_in, ho, wo, c_out, kh, kw, stride, pad;
    int ci_blocks, co_blocks, groups;      // groups = kh kw ci_blocks co_blocks
    int imgs_per_split;
    int K;
};

struct WgradStage {
    float a0, a1, b0, b1;
};

__device__ __forceinline__ float wg_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

__global__ __launch_bounds__(WG_WAVES * 64) void conv_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];          // [WG_WAVES][64][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, i = lane & 31;
    const int g = (int)(blockIdx.x % (unsigned)a.groups), split = (int)(blockIdx.x / (unsigned)a.groups);
    const int cob = g % a.co_blocks;
    const int g1 = g / a.co_blocks;
    const int cib = g1 % a.ci_blocks, tap = g1 / a.ci_blocks;
    const int ky = tap / a.kw, kx = tap - ky * a.kw;
    const int co0 = cob * 64, ci0 = cib * 64;
    const int img_lo = split * a.imgs_per_split, img_hi = min(a.n, img_lo + a.imgs_per_split);
    const int howo = a.ho * a.wo, npair = (howo + 1) >> 1;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)a.dy_bytes, 0x00020000);
    // Work items of the workgroup: (pixel pair of the map, sub-range of the split's images) - a wavefront keeps ONE pixel
    // pair and walks the images, so its lanes' offsets inside an image (and whether their tap lies in the padding) are
    // loop invariants and a step is four loads whose image offset is a scalar, and four MFMAs - no vector arithmetic
    // (with fp32 MFMAs every other vector instruction is matrix-pipe time).  Maps of fewer than 8 pixel pairs: the images
    // are cut into sub-ranges so that all 8 wavefronts have an item.
    const int nsub = npair >= WG_WAVES ? 1 : WG_WAVES / npair;
    const int n_items = npair * nsub;
    const int n_img = img_hi - img_lo, sub_imgs = (n_img + nsub - 1) / nsub;
    const unsigned ystride = (unsigned)(howo * a.c_out) * 4u, xstride = (unsigned)(a.h * a.w * a.c_in) * 4u;      // bytes per image

    f32x16 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.0f;
    // the wavefront's steps, flattened over its items: (item, image) -> next; past the end: out-of-range loads (zeros)
    int item = wave;
    unsigned voffy = kWgOob, voffx = kWgOob, soffy = 0u, soffx = 0u;
    int left = 0;                                                // images left in the current item
    auto open_item = [&]() {
        voffy = voffx = kWgOob;
        left = 0;
        if (item < n_items) {
            const int slot = item % npair, sub = item / npair;
            const int lo = img_lo + sub * sub_imgs, hi = min(img_hi, lo + sub_imgs);
            const int q = 2 * slot + half;
            const int oy = q / a.wo, ox = q - oy * a.wo;
            const int iy = oy * a.stride + ky - a.pad, ix = ox * a.stride + kx - a.pad;
            const bool okq = q < howo;
            const bool okx = okq && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
            voffy = okq ? ((unsigned)q * (unsigned)a.c_out + (unsigned)(co0 + i)) * 4u : kWgOob;
            voffx = okx ? ((unsigned)(iy * a.w + ix) * (unsigned)a.c_in + (unsigned)(ci0 + i)) * 4u : kWgOob;
            soffy = (unsigned)lo * ystride;
            soffx = (unsigned)lo * xstride;
            left = hi > lo ? hi - lo : 0;
        }
    };
    open_item();
    while (left == 0 && item < n_items) { item += WG_WAVES; open_item(); }
    int steps = 0;                                               // this wavefront's steps in all (wave-uniform)
    for (int it = wave; it < n_items; it += WG_WAVES) {
        const int sub = it / npair;
        const int lo = img_lo + sub * sub_imgs, hi = min(img_hi, lo + sub_imgs);
        steps += hi > lo ? hi - lo : 0;
    }
    auto issue = [&](WgradStage& st) {
        const bool live = left > 0;
        const unsigned vy = live ? voffy : kWgOob, vx = live ? voffx : kWgOob;
        st.a0 = wg_load(ry, vy, soffy);
        st.a1 = wg_load(ry, vy, soffy + 128u);
        st.b0 = wg_load(rx, vx, soffx);
        st.b1 = wg_load(rx, vx, soffx + 128u);
        soffy += ystride;
        soffx += xstride;
        if (--left <= 0) {                                       // (wave-uniform)
            do { item += WG_WAVES; open_item(); } while (left == 0 && item < n_items);
        }
    };
    auto mma = [&](const WgradStage& st) {
        acc[0][0] = WG_MFMA(st.a0, st.b0, acc[0][0]);
        acc[0][1] = WG_MFMA(st.a0, st.b1, acc[0][1]);
        acc[1][0] = WG_MFMA(st.a1, st.b0, acc[1][0]);
        acc[1][1] = WG_MFMA(st.a1, st.b1, acc[1][1]);
    };
    WgradStage s0, s1, s2, s3;
    issue(s0);
    issue(s1);
    issue(s2);
#pragma unroll 1
    for (int t = 0; t < steps; t += 4) {
        issue(s3); mma(s0);
        issue(s0); mma(s1);
        issue(s1); mma(s2);
        issue(s2); mma(s3);
    }
    // the 8 wavefronts' blocks, added in wavefront order
    float* mine = wg_lds + wave * 4096;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mine[(32 * u + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + 32 * v + i] = acc[u][v][r];
    __syncthreads();
    float* out = a.partial + (size_t)split * a.c_out * a.K;
    for (int e = threadIdx.x; e < 4096; e += WG_WAVES * 64) {
        float s = wg_lds[e];
#pragma unroll
        for (int wv = 1; wv < WG_WAVES; ++wv) s = s + wg_lds[wv * 4096 + e];
        const int row = e >> 6, col = e & 63;
        out[(size_t)(co0 + row) * a.K + (size_t)tap * a.c_in + ci0 + col] = s;
    }
}

// 3x3 kernels: ONE WAVEFRONT PER TAP.  conv_wgrad_kernel gives a workgroup's wavefronts different pixels of the same tap -
// nothing is shared between them and every MFMA needs 256 B of fresh operands from L2 (measured: 4.4 TB/s of L2 -> CU
// traffic, matrix pipe at 0.45).  Here the wavefronts of a workgroup walk the SAME (pixel pair, image) sequence, each for
// its own tap: the dy operand is the same line for all of them and the x operands are the taps' shifted views of one
// neighbourhood, so all but the first request of a line is served by the compute unit's L1.  Eight wavefronts (two per
// SIMD, evenly) take taps 0..7; the ninth tap is shared out among them by pixel pair afterwards (a second accumulator
// block, added through LDS in wavefront order).  Partial blocks per split as before.
__device__ __forceinline__ void wgrad_walk(const WgradArgs& a, __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t ry, int tap, int co0,
                                           int ci0, int img_lo, int n_img, int slot0, int slot_step, int half, int i,
                                           f32x16 (&acc)[2][2]) {
    const int ky = tap / a.kw, kx = tap - ky * a.kw;
    const int howo = a.ho * a.wo, npair = (howo + 1) >> 1;
    const unsigned ystride = (unsigned)(howo * a.c_out) * 4u, xstride = (unsigned)(a.h * a.w * a.c_in) * 4u;
    int slot = slot0, left = 0;
    unsigned voffy = kWgOob, voffx = kWgOob, soffy = 0u, soffx = 0u;
    auto open_slot = [&]() {
        voffy = voffx = kWgOob;
        left = 0;
        if (slot < npair && n_img > 0) {
            const int q = 2 * slot + half;
            const int oy = q / a.wo, ox = q - oy * a.wo;
            const int iy = oy * a.stride + ky - a.pad, ix = ox * a.stride + kx - a.pad;
            const bool okq = q < howo;
            const bool okx = okq && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
            voffy = okq ? ((unsigned)q * (unsigned)a.c_out + (unsigned)(co0 + i)) * 4u : kWgOob;
            voffx = okx ? ((unsigned)(iy * a.w + ix) * (unsigned)a.c_in + (unsigned)(ci0 + i)) * 4u : kWgOob;
            soffy = (unsigned)img_lo * ystride;
            soffx = (unsigned)img_lo * xstride;
            left = n_img;
        }
    };
    open_slot();
    const int my_slots = slot0 < npair ? (npair - slot0 + slot_step - 1) / slot_step : 0;
    const int steps = n_img > 0 ? my_slots * n_img : 0;
    auto issue = [&](WgradStage& st) {
        const bool live = left > 0;
        const unsigned vy = live ? voffy : kWgOob, vx = live ? voffx : kWgOob;
        st.a0 = wg_load(ry, vy, soffy);
        st.a1 = wg_load(ry, vy, soffy + 128u);
        st.b0 = wg_load(rx, vx, soffx);
        st.b1 = wg_load(rx, vx, soffx + 128u);
        soffy += ystride;
        soffx += xstride;
        if (--left <= 0) { slot += slot_step; open_slot(); }     // (wave-uniform)
    };
    auto mma = [&](const WgradStage& st) {
        acc[0][0] = WG_MFMA(st.a0, st.b0, acc[0][0]);
        acc[0][1] = WG_MFMA(st.a0, st.b1, acc[0][1]);
        acc[1][0] = WG_MFMA(st.a1, st.b0, acc[1][0]);
        acc[1][1] = WG_MFMA(st.a1, st.b1, acc[1][1]);
    };
    WgradStage s0, s1, s2, s3;
    issue(s0);
    issue(s1);
    issue(s2);
#pragma unroll 1
    for (int t = 0; t < steps; t += 4) {
        issue(s3); mma(s0);
        issue(s0); mma(s1);
        issue(s1); mma(s2);
        issue(s2); mma(s3);
    }
}

__global__ __launch_bounds__(WG_WAVES * 64) void conv_wgrad_taps_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];          // [WG_WAVES][64][64]: the shared ninth tap
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, i = lane & 31;
    const int g = (int)(blockIdx.x % (unsigned)a.groups), split = (int)(blockIdx.x / (unsigned)a.groups);       // groups = ci_blocks * co_blocks
    const int cob = g % a.co_blocks, cib = g / a.co_blocks;
    const int co0 = cob * 64, ci0 = cib * 64;
    const int img_lo = split * a.imgs_per_split, img_hi = min(a.n, img_lo + a.imgs_per_split);
    const int n_img = img_hi > img_lo ? img_hi - img_lo : 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)a.dy_bytes, 0x00020000);
    float* out = a.partial + (size_t)split * a.c_out * a.K;
    f32x16 acc[2][2];
    auto zero = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.0f;
    };
    // taps 0..7: this wavefront's own tap, every pixel pair
    zero();
    wgrad_walk(a, rx, ry, wave, co0, ci0, img_lo, n_img, 0, 1, half, i, acc);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(size_t)(co0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * half) * a.K + (size_t)wave * a.c_in + ci0 + 32 * v + i] = acc[u][v][r];
    // tap 8: pixel pairs wave, wave + 8, ... of every image; the eight partial blocks added in wavefront order
    zero();
    wgrad_walk(a, rx, ry, 8, co0, ci0, img_lo, n_img, wave, WG_WAVES, half, i, acc);
    float* mine = wg_lds + wave * 4096;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mine[(32 * u + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + 32 * v + i] = acc[u][v][r];
    __syncthreads();
    for (int e = threadIdx.x; e < 4096; e += WG_WAVES * 64) {
        float s = wg_lds[e];
#pragma unroll
        for (int wv = 1; wv < WG_WAVES; ++wv) s = s + wg_lds[wv * 4096 + e];
        out[(size_t)(co0 + (e >> 6)) * a.K + (size_t)8 * a.c_in + ci0 + (e & 63)] = s;
    }
}

// One input channel (the 7x7 stem of the 1-channel trunks): K = kh kw <= 64 taps are the "channels" of the B operand - lane
// i of B-tile v reads x at the position its OWN tap 32 v + i points to (a 4-byte gather inside one small image), taps
// beyond kh kw read nothing.  Wavefronts split the pixel pairs, blocks added through LDS as in conv_wgrad_kernel.
__global__ __launch_bounds__(WG_WAVES * 64) void conv_wgrad_stem_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];          // [WG_WAVES][64][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, i = lane & 31;
    const int cob = (int)(blockIdx.x % (unsigned)a.groups), split = (int)(blockIdx.x / (unsigned)a.groups);     // groups = co_blocks
    const int co0 = cob * 64;
    const int img_lo = split * a.imgs_per_split, img_hi = min(a.n, img_lo + a.imgs_per_split);
    const int n_img = img_hi > img_lo ? img_hi - img_lo : 0;
    const int howo = a.ho * a.wo, npair = (howo + 1) >> 1, taps = a.kh * a.kw;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)a.dy_bytes, 0x00020000);
    const unsigned ystride = (unsigned)(howo * a.c_out) * 4u, xstride = (unsigned)(a.h * a.w) * 4u;
    const int t0 = i, t1 = 32 + i;                              // this lane's taps in the two B tiles
    const int ky0 = t0 / a.kw, kx0 = t0 - ky0 * a.kw, ky1 = t1 / a.kw, kx1 = t1 - ky1 * a.kw;
    f32x16 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.0f;
    int slot = wave, left = 0;
    unsigned voffy = kWgOob, voffx0 = kWgOob, voffx1 = kWgOob, soffy = 0u, soffx = 0u;
    auto open_slot = [&]() {
        voffy = voffx0 = voffx1 = kWgOob;
        left = 0;
        if (slot < npair && n_img > 0) {
            const int q = 2 * slot + half;
            const int oy = q / a.wo, ox = q - oy * a.wo;
            const bool okq = q < howo;
            const int iy0 = oy * a.stride + ky0 - a.pad, ix0 = ox * a.stride + kx0 - a.pad;
            const int iy1 = oy * a.stride + ky1 - a.pad, ix1 = ox * a.stride + kx1 - a.pad;
            const bool ok0 = okq && t0 < taps && (unsigned)iy0 < (unsigned)a.h && (unsigned)ix0 < (unsigned)a.w;
            const bool ok1 = okq && t1 < taps && (unsigned)iy1 < (unsigned)a.h && (unsigned)ix1 < (unsigned)a.w;
            voffy = okq ? ((unsigned)q * (unsigned)a.c_out + (unsigned)(co0 + i)) * 4u : kWgOob;
            voffx0 = ok0 ? (unsigned)(iy0 * a.w + ix0) * 4u : kWgOob;
            voffx1 = ok1 ? (unsigned)(iy1 * a.w + ix1) * 4u : kWgOob;
            soffy = (unsigned)img_lo * ystride;
            soffx = (unsigned)img_lo * xstride;
            left = n_img;
        }
    };
    open_slot();
    const int my_slots = wave < npair ? (npair - wave + WG_WAVES - 1) / WG_WAVES : 0;
    const int steps = my_slots * n_img;
    auto issue = [&](WgradStage& st) {
        const bool live = left > 0;
        const unsigned vy = live ? voffy : kWgOob;
        st.a0 = wg_load(ry, vy, soffy);
        st.a1 = wg_load(ry, vy, soffy + 128u);
        st.b0 = wg_load(rx, live ? voffx0 : kWgOob, soffx);
        st.b1 = wg_load(rx, live ? voffx1 : kWgOob, soffx);
        soffy += ystride;
        soffx += xstride;
        if (--left <= 0) { slot += WG_WAVES; open_slot(); }
    };
    auto mma = [&](const WgradStage& st) {
        acc[0][0] = WG_MFMA(st.a0, st.b0, acc[0][0]);
        acc[0][1] = WG_MFMA(st.a0, st.b1, acc[0][1]);
        acc[1][0] = WG_MFMA(st.a1, st.b0, acc[1][0]);
        acc[1][1] = WG_MFMA(st.a1, st.b1, acc[1][1]);
    };
    WgradStage s0, s1, s2, s3;
    issue(s0);
    issue(s1);
    issue(s2);
#pragma unroll 1
    for (int t = 0; t < steps; t += 4) {
        issue(s3); mma(s0);
        issue(s0); mma(s1);
        issue(s1); mma(s2);
        issue(s2); mma(s3);
    }
    float* mine = wg_lds + wave * 4096;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mine[(32 * u + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + 32 * v + i] = acc[u][v][r];
    __syncthreads();
    float* out = a.partial + (size_t)split * a.c_out * a.K;
    for (int e = threadIdx.x; e < 4096; e += WG_WAVES * 64) {
        const int row = e >> 6, col = e & 63;
        if (col >= taps) continue;
        float s = wg_lds[e];
#pragma unroll
        for (int wv = 1; wv < WG_WAVES; ++wv) s = s + wg_lds[wv * 4096 + e];
        out[(size_t)(co0 + row) * a.K + col] = s;
    }
}

__global__ void conv_wgrad_reduce_kernel(const float* __restrict__ partial, int splits, size_t total, float* __restrict__ dw) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    float s = partial[e];
    int k = 1;
    for (; k + 8 <= splits; k += 8) {              // split order kept; eight loads in flight
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = partial[(size_t)(k + j) * total + e];
#pragma unroll
        for (int j = 0; j < 8; ++j) s = s + v[j];
    }
    for (; k < splits; ++k) s = s + partial[(size_t)k * total + e];
    dw[e] = s;
}

static int wgrad_splits(int n, int groups) {
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int splits = std::max(1, std::min(n, cus / std::max(1, groups)));
    const int ips = (n + splits - 1) / splits;
    return (n + ips - 1) / ips;
}

}  // namespace ipsx

using namespace ipsx;

IPSX_API int ipsx_conv2d_wgrad_nhwc_supported(int c_in, int c_out, int kh, int kw, int stride, int pad) {
    if (!(c_out > 0 && c_out % 64 == 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0)) return 0;
    if (c_in == 1) return kh * kw <= 64 ? 1 : 0;                 // one input channel: the taps are the B operand's columns
    return (c_in > 0 && c_in % 64 == 0 && kh * kw <= 49) ? 1 : 0;
}

IPSX_API size_t ipsx_conv2d_wgrad_nhwc_workspace_bytes(int64_t n, int c_in, int c_out, int kh, int kw) {
    if (n <= 0 || c_in <= 0 || c_out <= 0) return 0;
    const bool by_tap = kh * kw == 9 && c_in > 1;
    const int groups = c_in == 1 ? c_out / 64 : (by_tap ? 1 : kh * kw) * (c_in / 64) * (c_out / 64);
    const int splits = wgrad_splits((int)std::min<int64_t>(n, 1 << 30), groups);
    return (size_t)splits * c_out * kh * kw * c_in * sizeof(float);
}

IPSX_API int ipsx_conv2d_wgrad_nhwc(const float* x, const float* dy, int64_t n, int h, int w, int c_in, int c_out, int kh, int kw,
                                    int stride, int pad, float* dw, void* workspace, size_t workspace_bytes, void* stream) {
    IPSX_REQUIRE(x && dy && dw && n > 0 && h > 0 && w > 0, "conv2d_wgrad_nhwc: bad arguments");
    IPSX_REQUIRE(ipsx_conv2d_wgrad_nhwc_supported(c_in, c_out, kh, kw, stride, pad),
                 "conv2d_wgrad_nhwc: C_in = %d (1, or a multiple of 64) and C_out = %d (a multiple of 64), kernel %dx%d", c_in, c_out, kh, kw);
    const int ho = conv_out(h, kh, stride, pad), wo = conv_out(w, kw, stride, pad);
    IPSX_REQUIRE(ho > 0 && wo > 0, "conv2d_wgrad_nhwc: empty output");
    IPSX_REQUIRE((int64_t)n * h * w * c_in * 4 < ((int64_t)1 << 31) && (int64_t)n * ho * wo * c_out * 4 < ((int64_t)1 << 31),
                 "conv2d_wgrad_nhwc: activations of %lld images exceed one 2 GiB buffer - call per slice and add", (long long)n);
    const size_t need = ipsx_conv2d_wgrad_nhwc_workspace_bytes(n, c_in, c_out, kh, kw);
    if (!workspace || workspace_bytes < need)
        return fail(IPSX_EWORKSPACE, "conv2d_wgrad_nhwc: workspace %zu B < %zu B", workspace_bytes, need);
    WgradArgs a;
    a.x = x; a.dy = dy; a.partial = static_cast<float*>(workspace);
    a.x_bytes = (unsigned)((int64_t)n * h * w * c_in * 4);
    a.dy_bytes = (unsigned)((int64_t)n * ho * wo * c_out * 4);
    a.n = (int)n; a.h = h; a.w = w; a.c_in = c_in; a.ho = ho; a.wo = wo; a.c_out = c_out;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad;
    a.ci_blocks = c_in / 64; a.co_blocks = c_out / 64; a.groups = kh * kw * a.ci_blocks * a.co_blocks;
    a.K = kh * kw * c_in;
    const bool by_tap = kh * kw == 9 && c_in > 1;                // 3x3: one wavefront per tap (conv_wgrad_taps_kernel)
    if (by_tap) a.groups = a.ci_blocks * a.co_blocks;
    if (c_in == 1) a.groups = a.co_blocks;
    const int splits = wgrad_splits((int)n, a.groups);
    a.imgs_per_split = ((int)n + splits - 1) / splits;
    hipStream_t s = as_stream(stream);
    const size_t total = (size_t)c_out * a.K;
    if (by_tap) {
        static bool attr_t = false;
        const size_t lds_t = (size_t)WG_WAVES * 4096 * sizeof(float);
        if (!attr_t) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_taps_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t);
            attr_t = true;
        }
        conv_wgrad_taps_kernel<<<dim3((unsigned)(a.groups * splits)), dim3(WG_WAVES * 64), lds_t, s>>>(a);
        IPSX_TRY(launched("conv2d_wgrad_nhwc (taps)"));
        conv_wgrad_reduce_kernel<<<dim3((unsigned)cdiv((int64_t)total, 256)), dim3(256), 0, s>>>(a.partial, splits, total, dw);
        return launched("conv2d_wgrad_nhwc reduce");
    }
    const size_t lds = (size_t)WG_WAVES * 4096 * sizeof(float);
    if (c_in == 1) {
        static bool attr_s = false;
        if (!attr_s) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_stem_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_s = true;
        }
        conv_wgrad_stem_kernel<<<dim3((unsigned)(a.groups * splits)), dim3(WG_WAVES * 64), lds, s>>>(a);
        IPSX_TRY(launched("conv2d_wgrad_nhwc (stem)"));
        conv_wgrad_reduce_kernel<<<dim3((unsigned)cdiv((int64_t)total, 256)), dim3(256), 0, s>>>(a.partial, splits, total, dw);
        return launched("conv2d_wgrad_nhwc reduce");
    }
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    conv_wgrad_kernel<<<dim3((unsigned)(a.groups * splits)), dim3(WG_WAVES * 64), lds, s>>>(a);
    IPSX_TRY(launched("conv2d_wgrad_nhwc"));
    conv_wgrad_reduce_kernel<<<dim3((unsigned)cdiv((int64_t)total, 256)), dim3(256), 0, s>>>(a.partial, splits, total, dw);
    return launched("conv2d_wgrad_nhwc reduce");
}
