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

}  // namespace ipsx

using namespace ipsx;

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
