// patchify.hip - the step in front of ips(): image on disk -> (B, N, C, ph, pw) patch tensor in HBM
// (SURVEY.md section 8 f, N-c).
//
//   ipsx_patchify         dense (B, C, H, W) image -> patches; what
//                         img.unfold(1, ph, sh).unfold(2, pw, sw).permute(1, 2, 0, 3, 4).reshape(-1, C, ph, pw)
//                         produces (reference data/megapixel_mnist/mnist_dataset.py:44-51,
//                         data/traffic/traffic_dataset.py:336-343), for a whole batch in one launch.
//   ipsx_patchify_sparse  Megapixel-MNIST's on-disk form - per image the flat indices and values of the
//                         non-zero pixels of an (H, W, C) canvas (mnist_dataset.py:34-42) - straight to patches:
//                         zero the patch tensor, scatter every non-zero into the patch(es) covering it, and
//                         flag the patches that received one.  A 1500x1500 image has ~30 k non-zeros
//                         (0.4 MB) against 9 MB dense, so the host and PCIe move 20x less, and the
//                         blank-patch flags ipsx_trunk_encode_dedup needs come for free.
//
// Both are HBM-bound copies: algorithmic bytes = 4 B read + 4 B written per output element (dense),
// 4 B written per output element + 20 B per non-zero (sparse).

#include "ipsx_common.h"

namespace ipsx {

struct PatchGeom {
    int c, h, w, ph, pw, sh, sw, ny, nx;
};

// one thread per V (float or float4) of the output; x runs fastest so reads and writes are both contiguous
template <typename V, int VW>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, PatchGeom g, long long total,
                                                       V* __restrict__ out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int rowv = g.pw / VW;
    long long r = e;
    const int xv = (int)(r % rowv); r /= rowv;
    const int y = (int)(r % g.ph); r /= g.ph;
    const int c = (int)(r % g.c); r /= g.c;
    const int px = (int)(r % g.nx); r /= g.nx;
    const int py = (int)(r % g.ny); r /= g.ny;      // r = image
    const size_t src = (((size_t)r * g.c + c) * g.h + (size_t)py * g.sh + y) * g.w + (size_t)px * g.sw + (size_t)xv * VW;
    out[e] = *reinterpret_cast<const V*>(img + src);
}

// one thread per non-zero; the image of a non-zero is found by bisection of the (B+1) offsets
__global__ __launch_bounds__(256) void patchify_sparse_kernel(const long long* __restrict__ index,
                                                              const float* __restrict__ value,
                                                              const long long* __restrict__ offsets, int b, PatchGeom g,
                                                              float* __restrict__ out, int* __restrict__ nonblank) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= offsets[b]) return;
    int lo = 0, hi = b;                               // offsets[lo] <= e < offsets[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= e) lo = mid; else hi = mid;
    }
    const long long flat = index[e];
    if (flat < 0 || flat >= (long long)g.h * g.w * g.c) return;      // the host checks the range before upload
    const float v = value[e];
    const int c = (int)(flat % g.c);
    const long long pix = flat / g.c;
    const int y = (int)(pix / g.w), x = (int)(pix % g.w);
    // patches covering row y: py*sh <= y < py*sh + ph
    const int py1 = min(y / g.sh, g.ny - 1), px1 = min(x / g.sw, g.nx - 1);
    const int py0 = max(0, (y - g.ph + g.sh) / g.sh), px0 = max(0, (x - g.pw + g.sw) / g.sw);
    const bool nz = (__float_as_uint(v) & 0x7FFFFFFFu) != 0u;
    for (int py = py0; py <= py1; ++py) {
        const int yy = y - py * g.sh;
        if (yy < 0 || yy >= g.ph) continue;
        for (int px = px0; px <= px1; ++px) {
            const int xx = x - px * g.sw;
            if (xx < 0 || xx >= g.pw) continue;
            const size_t n = ((size_t)lo * g.ny + py) * g.nx + px;
            out[((n * g.c + c) * g.ph + yy) * g.pw + xx] = v;
            if (nz && nonblank) nonblank[n] = 1;
        }
    }
}

static int geom(PatchGeom& g, int c, int h, int w, int ph, int pw, int sh, int sw) {
    if (c <= 0 || h <= 0 || w <= 0 || ph <= 0 || pw <= 0 || sh <= 0 || sw <= 0 || ph > h || pw > w)
        return fail(IPSX_EINVAL, "patchify: image %dx%dx%d, patch %dx%d, stride %dx%d", c, h, w, ph, pw, sh, sw);
    g = PatchGeom{c, h, w, ph, pw, sh, sw, (h - ph) / sh + 1, (w - pw) / sw + 1};
    return IPSX_OK;
}

}  // namespace ipsx

using namespace ipsx;

IPSX_API int64_t ipsx_patchify_count(int h, int w, int ph, int pw, int sh, int sw) {
    if (h <= 0 || w <= 0 || ph <= 0 || pw <= 0 || sh <= 0 || sw <= 0 || ph > h || pw > w) return 0;
    return (int64_t)((h - ph) / sh + 1) * ((w - pw) / sw + 1);
}

IPSX_API int ipsx_patchify(const float* img, int b, int c, int h, int w, int ph, int pw, int sh, int sw,
                           float* patches, void* stream) {
    IPSX_REQUIRE(img && patches && b > 0, "patchify: bad arguments");
    PatchGeom g;
    IPSX_TRY(geom(g, c, h, w, ph, pw, sh, sw));
    const long long elems = (long long)b * g.ny * g.nx * c * ph * pw;
    const bool v4 = pw % 4 == 0 && w % 4 == 0 && sw % 4 == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(patches) & 15) == 0;
    if (v4) {
        const long long total = elems / 4;
        patchify_kernel<float4, 4><<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream)>>>(
            img, g, total, reinterpret_cast<float4*>(patches));
    } else {
        patchify_kernel<float, 1><<<dim3((unsigned)cdiv(elems, 256)), dim3(256), 0, as_stream(stream)>>>(
            img, g, elems, patches);
    }
    return launched("patchify");
}

IPSX_API int ipsx_patchify_sparse(const int64_t* index, const float* value, const int64_t* offsets, int64_t nnz, int b,
                                  int c, int h, int w, int ph, int pw, int sh, int sw, float* patches,
                                  int32_t* nonblank, void* stream) {
    IPSX_REQUIRE(offsets && patches && b > 0 && nnz >= 0 && (nnz == 0 || (index && value)), "patchify_sparse: bad arguments");
    PatchGeom g;
    IPSX_TRY(geom(g, c, h, w, ph, pw, sh, sw));
    hipStream_t s = as_stream(stream);
    const size_t n_patch = (size_t)b * g.ny * g.nx;
    if (hipMemsetAsync(patches, 0, n_patch * c * ph * pw * sizeof(float), s) != hipSuccess)
        return fail(IPSX_EHIP, "patchify_sparse: memset failed");
    if (nonblank && hipMemsetAsync(nonblank, 0, n_patch * sizeof(int32_t), s) != hipSuccess)
        return fail(IPSX_EHIP, "patchify_sparse: memset failed");
    if (nnz == 0) return IPSX_OK;
    patchify_sparse_kernel<<<dim3((unsigned)cdiv(nnz, 256)), dim3(256), 0, s>>>(
        reinterpret_cast<const long long*>(index), value, reinterpret_cast<const long long*>(offsets), b, g, patches,
        reinterpret_cast<int*>(nonblank));
    return launched("patchify_sparse");
}
