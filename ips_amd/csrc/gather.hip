// gather.hip - row gathers that assemble the output of IPSNet.ips:
// mem_patch = gather(patches, mem_idx), mem_pos = gather(pos_enc, mem_idx)
// (reference architecture/ips_net.py:245-250).  Pure HBM-bound copy: one workgroup
// per selected row, 16 bytes per lane when the row allows it.

#include "ipsx_common.h"

namespace ipsx {

template <typename V>
__global__ __launch_bounds__(256) void gather_rows_kernel(const V* __restrict__ src,
                                                          const long long* __restrict__ idx,
                                                          V* __restrict__ dst, long long n_rows, int m,
                                                          long long row_units, long long src_bstride_rows) {
    const int j = blockIdx.x, b = blockIdx.y;
    long long r = idx[(size_t)b * m + j];
    r = r < 0 ? 0 : (r >= n_rows ? n_rows - 1 : r);          // never read out of bounds
    const V* s = src + ((size_t)b * src_bstride_rows + (size_t)r) * row_units;
    V* d = dst + ((size_t)b * m + j) * row_units;
    for (long long i = threadIdx.x; i < row_units; i += 256) d[i] = s[i];
}

// The end of an ips() call whose selection ran as a resident loop, in ONE launch (round 5; it was four: a copy of the
// loop's index buffer, the gather of the patches, the gather of the positional encodings and the copy of the loop's
// status word to the host): workgroup (j, b) copies patch row and positional row of selection j of image b (16 bytes per
// lane) and the index itself; the first thread of the launch also hands the loop's status word to its pinned host mirror.
struct FinishArgs {
    const uint4* patches; long long patch_units, patch_bstride_rows, n_rows;
    const uint4* pos; long long pos_units, pos_bstride_rows;
    const long long* idx; long long* idx_out;
    uint4* out_patch; uint4* out_pos;
    const int* status; int* status_host;
    int m;
};

__global__ __launch_bounds__(256) void ips_finish_kernel(FinishArgs a) {
    const int j = blockIdx.x, b = blockIdx.y;
    const long long raw = a.idx[(size_t)b * a.m + j];
    const long long r = raw < 0 ? 0 : (raw >= a.n_rows ? a.n_rows - 1 : raw);         // never read out of bounds
    const uint4* s = a.patches + ((size_t)b * a.patch_bstride_rows + (size_t)r) * a.patch_units;
    uint4* d = a.out_patch + ((size_t)b * a.m + j) * a.patch_units;
    for (long long i = threadIdx.x; i < a.patch_units; i += 256) d[i] = s[i];
    if (a.pos) {
        const uint4* sp = a.pos + ((size_t)b * a.pos_bstride_rows + (size_t)r) * a.pos_units;
        uint4* dp = a.out_pos + ((size_t)b * a.m + j) * a.pos_units;
        for (long long i = threadIdx.x; i < a.pos_units; i += 256) dp[i] = sp[i];
    }
    if (threadIdx.x == 0) {
        a.idx_out[(size_t)b * a.m + j] = raw;
        if (a.status_host && j == 0 && b == 0)
            __hip_atomic_store(a.status_host, __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace ipsx

using namespace ipsx;

IPSX_API int ipsx_ips_finish(const void* patches, int64_t patch_row_bytes, int64_t patch_bstride_rows, int64_t n_rows,
                             const void* pos, int64_t pos_row_bytes, int64_t pos_bstride_rows, const int64_t* mem_idx, int b, int m,
                             void* mem_patch, void* mem_pos, int64_t* mem_idx_out, const int32_t* status, int32_t* status_host,
                             void* stream) {
    IPSX_REQUIRE(patches && mem_idx && mem_patch && mem_idx_out && b > 0 && m > 0 && n_rows > 0, "ips_finish: bad arguments");
    IPSX_REQUIRE(patch_row_bytes > 0 && patch_row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(patches) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(mem_patch) & 15) == 0, "ips_finish: patch rows of 16-byte units at 16-byte addresses");
    IPSX_REQUIRE(!pos || (mem_pos && pos_row_bytes > 0 && pos_row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(pos) & 15) == 0 &&
                          (reinterpret_cast<uintptr_t>(mem_pos) & 15) == 0), "ips_finish: positional rows of 16-byte units");
    IPSX_REQUIRE(!status_host || status, "ips_finish: a host mirror needs the status word");
    FinishArgs a;
    a.patches = static_cast<const uint4*>(patches); a.patch_units = patch_row_bytes / 16; a.patch_bstride_rows = patch_bstride_rows;
    a.n_rows = n_rows;
    a.pos = static_cast<const uint4*>(pos); a.pos_units = pos ? pos_row_bytes / 16 : 0; a.pos_bstride_rows = pos_bstride_rows;
    a.idx = reinterpret_cast<const long long*>(mem_idx); a.idx_out = reinterpret_cast<long long*>(mem_idx_out);
    a.out_patch = static_cast<uint4*>(mem_patch); a.out_pos = static_cast<uint4*>(mem_pos);
    a.status = status; a.status_host = status_host; a.m = m;
    ips_finish_kernel<<<dim3((unsigned)m, (unsigned)b), dim3(256), 0, as_stream(stream)>>>(a);
    return launched("ips_finish");
}

IPSX_API int ipsx_gather_rows(const void* src, const int64_t* idx, void* dst, int b, int64_t n_rows, int m,
                              int64_t row_bytes, int64_t src_bstride_rows, void* stream) {
    IPSX_REQUIRE(src && idx && dst && b > 0 && n_rows > 0 && m > 0, "gather_rows: bad arguments");
    IPSX_REQUIRE(row_bytes > 0 && row_bytes % 4 == 0, "gather_rows: row of %lld bytes", (long long)row_bytes);
    dim3 grid((unsigned)m, (unsigned)b);
    const bool a16 = row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    if (a16)
        gather_rows_kernel<uint4><<<grid, dim3(256), 0, as_stream(stream)>>>(
            static_cast<const uint4*>(src), reinterpret_cast<const long long*>(idx), static_cast<uint4*>(dst),
            n_rows, m, row_bytes / 16, src_bstride_rows);
    else
        gather_rows_kernel<uint32_t><<<grid, dim3(256), 0, as_stream(stream)>>>(
            static_cast<const uint32_t*>(src), reinterpret_cast<const long long*>(idx),
            static_cast<uint32_t*>(dst), n_rows, m, row_bytes / 4, src_bstride_rows);
    return launched("gather_rows");
}
