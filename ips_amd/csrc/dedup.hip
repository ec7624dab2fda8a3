// dedup.hip - exact blank-patch deduplication in front of the encoder (SURVEY.md section 8 f, N-c).
//
// About 93 % of the 32-px patches of a Megapixel-MNIST image are all-zero
// (reference data/megapixel_mnist/make_mnist.py).  In eval / no-grad mode the encoder is a pure
// function of the patch (reference architecture/ips_net.py:191-193), so all blank patches share ONE
// embedding: encode the non-blank patches and a single blank one, then copy.  The result is
// bit-identical to encoding every patch (same kernel, same inputs) - this is not an approximation.
// Opt-in (ipsx_trunk_encode_dedup); everything stays on the device, no host synchronisation:
//   blank_flags_kernel   one wavefront per patch: is any element non-zero?        (HBM-bound, reads all patches once)
//   compact_kernel       one workgroup: ordered list of the patches to encode (non-blank ones + the first
//                        blank), its length, and for every patch the slot of its representative
//   fused trunk          launched for the worst case, workgroups beyond the device-side count exit at once
//   scatter_rows_kernel  emb[j] = emb_unique[slot[j]]

#include "ipsx_common.h"

namespace ipsx {

bool fused_trunk_supported(const ipsx_trunk* t);
int fused_trunk_encode_indexed(const ipsx_trunk* t, const float* patches, int64_t n_max, const int* index,
                               const int* count, float* emb, hipStream_t s);

__global__ __launch_bounds__(256) void blank_flags_kernel(const float* __restrict__ x, long long n, int elems4,
                                                          int* __restrict__ nonblank) {
    const int lane = threadIdx.x & 63;
    const long long j = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    const uint4* p = reinterpret_cast<const uint4*>(x) + (size_t)j * elems4;
    unsigned any = 0;
    for (int i = lane; i < elems4; i += 64) {
        const uint4 v = p[i];
        any |= (v.x | v.y | v.z | v.w) & 0x7FFFFFFFu;        // -0.0 is zero too
    }
    const unsigned long long b = __ballot(any != 0);
    if (lane == 0) nonblank[j] = b != 0ull;
}

// single workgroup, ordered compaction (patch order is preserved, so results do not depend on timing)
__global__ __launch_bounds__(1024) void compact_kernel(const int* __restrict__ nonblank, int n, int* __restrict__ index,
                                                       int* __restrict__ slot, int* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int base, first_blank;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { base = 0; first_blank = 0x7FFFFFFF; }
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int j = c0 + tid;
        const int f = (j < n) ? nonblank[j] : 0;
        const unsigned long long m = __ballot(f != 0);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        if (j < n && !f) atomicMin(&first_blank, j);   // value = smallest blank index, whatever the arrival order
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (f) { index[off + before] = j; slot[j] = off + before; }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; base += t; }
        __syncthreads();
    }
    // one representative for all blank patches, after the non-blank ones
    const int nb = base, fb = first_blank == 0x7FFFFFFF ? -1 : first_blank;
    if (tid == 0) {
        if (fb >= 0) index[nb] = fb;
        *count = nb + (fb >= 0 ? 1 : 0);
    }
    for (int j = tid; j < n; j += 1024)
        if (!nonblank[j]) slot[j] = nb;
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const float4* __restrict__ src, const int* __restrict__ slot,
                                                           float4* __restrict__ dst, long long n, int row4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * row4) return;
    const long long j = i / row4;
    const int c = (int)(i - j * row4);
    dst[i] = src[(size_t)slot[j] * row4 + c];
}

}  // namespace ipsx

using namespace ipsx;

IPSX_API int ipsx_trunk_encode_indexed(const ipsx_trunk* t, const float* patches, const int32_t* index,
                                       int64_t n_index, float* emb, void* stream) {
    IPSX_REQUIRE(t && patches && index && emb && n_index >= 0, "trunk_encode_indexed: bad arguments");
    IPSX_REQUIRE(fused_trunk_supported(t), "trunk_encode_indexed: only the fused 1x32x32 trunk is supported");
    if (n_index == 0) return IPSX_OK;
    return fused_trunk_encode_indexed(t, patches, n_index, index, nullptr, emb, as_stream(stream));
}

IPSX_API size_t ipsx_trunk_dedup_workspace_bytes(const ipsx_trunk* t, int64_t n_patch) {
    if (!t || n_patch <= 0) return 0;
    // nonblank, index, slot (n ints each) + count + the compacted embeddings (n+1 rows of 128 floats)
    return (size_t)n_patch * 12 + 256 + (size_t)(n_patch + 1) * 128 * sizeof(float) + 256;
}

static int encode_dedup(const ipsx_trunk* t, const float* patches, int64_t n_patch, const int32_t* flags, float* emb,
                        void* workspace, size_t workspace_bytes, int32_t* n_encoded, void* stream) {
    IPSX_REQUIRE(t && patches && emb && n_patch >= 0, "trunk_encode_dedup: bad arguments");
    IPSX_REQUIRE(fused_trunk_supported(t), "trunk_encode_dedup: only the fused 1x32x32 trunk is supported");
    IPSX_REQUIRE(n_patch < ((int64_t)1 << 31), "trunk_encode_dedup: too many patches");
    if (n_patch == 0) return IPSX_OK;
    const size_t need = ipsx_trunk_dedup_workspace_bytes(t, n_patch);
    if (!workspace || workspace_bytes < need)
        return fail(IPSX_EWORKSPACE, "trunk_encode_dedup: workspace %zu B < %zu B", workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    int* nonblank = static_cast<int*>(workspace);
    int* index = nonblank + n_patch;
    int* slot = index + n_patch;
    int* count = slot + n_patch;
    float* emb_u = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) +
                                            (((size_t)n_patch * 12 + 4 + 255) & ~(size_t)255));
    const int elems4 = t->c_in * t->h * t->w / 4;
    if (!flags) {
        blank_flags_kernel<<<dim3((unsigned)cdiv(n_patch, 4)), dim3(256), 0, s>>>(patches, n_patch, elems4, nonblank);
        IPSX_TRY(launched("blank_flags"));
    }
    compact_kernel<<<dim3(1), dim3(1024), 0, s>>>(flags ? reinterpret_cast<const int*>(flags) : nonblank, (int)n_patch,
                                                  index, slot, count);
    IPSX_TRY(launched("compact"));
    IPSX_TRY(fused_trunk_encode_indexed(t, patches, n_patch, index, count, emb_u, s));
    scatter_rows_kernel<<<dim3((unsigned)cdiv(n_patch * 32, 256)), dim3(256), 0, s>>>(
        reinterpret_cast<const float4*>(emb_u), slot, reinterpret_cast<float4*>(emb), n_patch, 32);
    IPSX_TRY(launched("scatter_rows"));
    if (n_encoded) {
        if (hipMemcpyAsync(n_encoded, count, sizeof(int), hipMemcpyDeviceToDevice, s) != hipSuccess)
            return fail(IPSX_EHIP, "trunk_encode_dedup: count copy failed");
    }
    return IPSX_OK;
}

static int dedup_needs_f32(const ipsx_trunk* t) {
    IPSX_REQUIRE(t && t->patch_dtype == 0, "trunk_encode_dedup: blank-patch detection reads float32 patches");
    return IPSX_OK;
}

IPSX_API int ipsx_trunk_encode_dedup(const ipsx_trunk* t, const float* patches, int64_t n_patch, float* emb,
                                     void* workspace, size_t workspace_bytes, int32_t* n_encoded, void* stream) {
    IPSX_TRY(dedup_needs_f32(t));
    return encode_dedup(t, patches, n_patch, nullptr, emb, workspace, workspace_bytes, n_encoded, stream);
}

IPSX_API int ipsx_trunk_encode_dedup_flagged(const ipsx_trunk* t, const float* patches, int64_t n_patch,
                                             const int32_t* nonblank, float* emb, void* workspace,
                                             size_t workspace_bytes, int32_t* n_encoded, void* stream) {
    IPSX_REQUIRE(nonblank, "trunk_encode_dedup_flagged: no flags");
    IPSX_TRY(dedup_needs_f32(t));
    return encode_dedup(t, patches, n_patch, nonblank, emb, workspace, workspace_bytes, n_encoded, stream);
}
