// scorer.hip - the selection loop behind the C ABI: ipsx_scan* / ipsx_scan_persistent* / ipsx_scan_gate / ipsx_publish_rows
// pick the kernel a shape takes (scan_fast.hip, scan_cam.hip, scan_large.hip) and own the process-wide switches.
// (Round 6: the logits live in logits.hip, scores / top-M in topm.hip, shared device code in scan_common.h.)
//
// Reference: MultiHeadCrossAttention.get_attn (architecture/transformer.py:71-83),
// ScaledDotProductAttention.compute_attn (:29-34), Transformer.get_scores (:143-148),
// IPSNet.score_and_select (architecture/ips_net.py:136-155) and the chunk loop of
// IPSNet.ips (:213-241).
//
// A patch's logits  l[h,t] = (q_w q)[t,h,:]/sqrt(Dk) . (k_w (emb+pos))[h,:]  depend on
// that patch alone; only the softmax denominator depends on the candidate set.  So
//   logits_kernel  computes them once per patch (K projection on the fp32 matrix
//                  cores, q.k reduction from LDS), and
//   scan_fast_kernel / scan_large_kernel  replay the reference loop on those cached
//                  logits with ONE workgroup per image: per-(h,t) softmax statistics by
//                  wavefront reductions, mean over heads then tokens, rank, keep the
//                  top M - no host round trips.
// Arithmetic order is the oracle's (oracle/ips_oracle.cpp orc_logits,
// orc_scores_from_logits, orc_topm).
//
// Algorithmic bytes of the scan: R*4 bytes of logits per patch (R = H*n_token), read
// once per iteration it takes part in.

#include <algorithm>
#include <cstdlib>

#include "scan_common.h"

namespace ipsx {

int g_tie_order = 1;     // 0 canonical; 1 (default) torch.topk's order where bit-identical candidates tie; 2 wherever scores tie
int g_persist_wait_ms = 50;         // ipsx_set_persistent_wait_ms: longest wait of a persistent loop / its gate without progress
bool g_scan_generic = false;        // diagnostic (ipsx_dbg_scan_generic): every shape through scan_large_kernel
bool g_replay_stamps_on = false;    // diagnostic (ipsx_dbg_replay_stamps): the replay's phases are stamped from the first read on
bool g_scan_team_trunc = true;      // diagnostic (ipsx_dbg_scan_team_trunc)
int g_scan_team = -1;               // diagnostic (ipsx_dbg_scan_team): workgroups per image of scan_large_team_kernel (-1: default)
bool g_scan_direct = true;          // diagnostic (ipsx_dbg_scan_direct): 0 = scan_large_kernel's five generic passes for every shape
bool g_scan_r8 = true;              // diagnostic (ipsx_dbg_scan_r8): 0 sends the shape of scan_cam_kernel through scan_fast_kernel
unsigned long long* g_scan_stamps = nullptr;   // diagnostic only (ipsx_dbg_scan_stamps)

// Diagnostic (ipsx_dbg_persist_log; tools/soak.py): what the gate and the resident loops saw, on the 100 MHz clock -
// [0] gate launches, [1] longest gate wait (ticks), [2] gate waits that ran into their bound, [3] start of the last gate,
// [4] the moment the last loop became resident, [5] loops that gave up waiting for rows, [6] start of the slowest gate,
// [7] the moment the loop it waited for became resident (0: not before the gate gave up)
__device__ unsigned long long g_persist_log[8];


unsigned long long* persist_log() {
    static unsigned long long* p[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!p[dev]) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_persist_log)) == hipSuccess) p[dev] = static_cast<unsigned long long*>(q);
    }
    return p[dev];
}

}  // namespace ipsx

using namespace ipsx;

IPSX_API int ipsx_scan(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
                       int64_t* mem_idx, float* mem_score, int32_t* tie_flag, void* workspace, size_t workspace_bytes,
                       void* stream) {
    IPSX_REQUIRE(n > m && i > 0, "scan: needs more patches (%lld) than memory slots (%d)", (long long)n, m);
    if (tie_flag && hipMemsetAsync(tie_flag, 0, sizeof(int32_t) * (size_t)std::max(b, 0), as_stream(stream)) != hipSuccess)
        return fail(IPSX_EHIP, "scan: memset failed");
    return ipsx_scan_range(logits, b, n, m, i, h, n_token, 0, (n - m + i - 1) / i, mem_idx, mem_score, tie_flag,
                           workspace, workspace_bytes, stream);
}

static int scan_range_impl(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
                           int64_t it_begin, int64_t it_end, int64_t* mem_idx, float* mem_score,
                           int32_t* tie_flag, const int32_t* ready, int32_t* status, void* workspace,
                           size_t workspace_bytes, void* stream, const int32_t* cond = nullptr, int32_t cond_mask = 0,
                           int ready_stride = 0, int workgroups = 0);

IPSX_API int ipsx_scan_range(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
                             int64_t it_begin, int64_t it_end, int64_t* mem_idx, float* mem_score,
                             int32_t* tie_flag, void* workspace, size_t workspace_bytes, void* stream) {
    return scan_range_impl(logits, b, n, m, i, h, n_token, it_begin, it_end, mem_idx, mem_score, tie_flag, nullptr, nullptr,
                           workspace, workspace_bytes, stream);
}

IPSX_API size_t ipsx_scan_workspace_bytes(int b, int m, int i, int h, int n_token) {
    if (b <= 0 || m <= 0 || i <= 0 || h <= 0 || n_token <= 0) return 0;
    if (scan_fast_plan(m, i, h, n_token).ok && !g_scan_generic) return 0;
    return (size_t)b * scan_large_ws_per_image(m, i, h, n_token);
}

IPSX_API int ipsx_scan_workgroups_per_image(int b, int m, int i, int h, int n_token) {
    if (b <= 0 || m <= 0 || i <= 0 || h <= 0 || n_token <= 0) return 0;
    if (scan_fast_plan(m, i, h, n_token).ok && !g_scan_generic) return 1;
    return scan_large_team(b, m, i, h, n_token);
}

__global__ void publish_rows_kernel(int* ready, int value) {
    __hip_atomic_store(ready, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

IPSX_API int ipsx_scan_persistent_supported(int m, int i, int h, int n_token) {
    if (m <= 0 || i <= 0 || h <= 0 || n_token <= 0) return 0;
    return scan_fast_plan(m, i, h, n_token).ok ? 1 : 0;
}

IPSX_API int ipsx_scan_persistent(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
                                  int64_t* mem_idx, float* mem_score, int32_t* tie_flag, const int32_t* ready,
                                  int32_t ready_per_image, int32_t* status, void* stream) {
    IPSX_REQUIRE(ready && status, "scan_persistent: needs the progress word(s) and the status word");
    IPSX_REQUIRE(n > m && i > 0, "scan: needs more patches (%lld) than memory slots (%d)", (long long)n, m);
    IPSX_REQUIRE(ipsx_scan_persistent_supported(m, i, h, n_token), "scan_persistent: shape not covered (use ipsx_scan_range)");
    return scan_range_impl(logits, b, n, m, i, h, n_token, 0, (n - m + i - 1) / i, mem_idx, mem_score, tie_flag, ready, status,
                           nullptr, 0, stream, nullptr, 0, ready_per_image ? 1 : 0);
}

IPSX_API int ipsx_scan_persistent_groupable(int m, int i, int h, int n_token) {
    return g_scan_r8 && scan_cam_shape(m, i, h, n_token) ? 1 : 0;
}

IPSX_API int ipsx_scan_persistent_on(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
                                     int64_t* mem_idx, float* mem_score, int32_t* tie_flag, const int32_t* ready,
                                     int32_t ready_per_image, int32_t* status, int workgroups, void* stream) {
    IPSX_REQUIRE(ready && status, "scan_persistent: needs the progress word(s) and the status word");
    IPSX_REQUIRE(n > m && i > 0, "scan: needs more patches (%lld) than memory slots (%d)", (long long)n, m);
    IPSX_REQUIRE(ipsx_scan_persistent_supported(m, i, h, n_token), "scan_persistent: shape not covered (use ipsx_scan_range)");
    return scan_range_impl(logits, b, n, m, i, h, n_token, 0, (n - m + i - 1) / i, mem_idx, mem_score, tie_flag, ready, status,
                           nullptr, 0, stream, nullptr, 0, ready_per_image ? 1 : 0, workgroups);
}

// The persistent loop for EVERY shape ipsx_scan covers - candidate sets beyond the LDS take the workspace of ipsx_scan
// (ipsx_scan_workspace_bytes; scan_large_kernel waits for its rows like the LDS-resident loops do) - and the conditional
// recovery launch with a workspace.
IPSX_API int ipsx_scan_persistent_ws(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
                                     int64_t* mem_idx, float* mem_score, int32_t* tie_flag, const int32_t* ready,
                                     int32_t ready_per_image, int32_t* status, int workgroups, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    IPSX_REQUIRE(ready && status, "scan_persistent: needs the progress word(s) and the status word");
    IPSX_REQUIRE(n > m && i > 0, "scan: needs more patches (%lld) than memory slots (%d)", (long long)n, m);
    return scan_range_impl(logits, b, n, m, i, h, n_token, 0, (n - m + i - 1) / i, mem_idx, mem_score, tie_flag, ready, status,
                           workspace, workspace_bytes, stream, nullptr, 0, ready_per_image ? 1 : 0, workgroups);
}

IPSX_API int ipsx_scan_range_if_ws(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
                                   int64_t it_begin, int64_t it_end, int64_t* mem_idx, float* mem_score,
                                   int32_t* tie_flag, const int32_t* cond, int32_t cond_mask, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    IPSX_REQUIRE(cond && cond_mask, "scan_range_if: needs the condition word and a mask");
    return scan_range_impl(logits, b, n, m, i, h, n_token, it_begin, it_end, mem_idx, mem_score, tie_flag, nullptr, nullptr,
                           workspace, workspace_bytes, stream, cond, cond_mask);
}

// one thread that holds its stream until every workgroup of the persistent scan is resident (bounded: ~0.5 s)
__global__ void scan_gate_kernel(const int* status, unsigned long long wait_ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool gave_up = false;
    while ((__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) == 0) {
        __builtin_amdgcn_s_sleep(8);
        if (__builtin_amdgcn_s_memrealtime() - t0 > wait_ticks) { gave_up = true; break; }
    }
    const unsigned long long waited = __builtin_amdgcn_s_memrealtime() - t0;
    g_persist_log[0] += 1;
    g_persist_log[3] = t0;
    if (gave_up) g_persist_log[2] += 1;
    if (waited > g_persist_log[1]) {
        g_persist_log[1] = waited;
        g_persist_log[6] = t0;
        g_persist_log[7] = gave_up ? 0ull : __hip_atomic_load(&g_persist_log[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

IPSX_API int ipsx_scan_gate(const int32_t* status, void* stream) {
    IPSX_REQUIRE(status, "scan_gate: null pointer");
    scan_gate_kernel<<<dim3(1), dim3(1), 0, as_stream(stream)>>>(status, (unsigned long long)g_persist_wait_ms * 100000ull);
    return launched("scan_gate");
}

IPSX_API int ipsx_publish_rows(int32_t* ready, int32_t value, void* stream) {
    IPSX_REQUIRE(ready, "publish_rows: null pointer");
    publish_rows_kernel<<<dim3(1), dim3(1), 0, as_stream(stream)>>>(ready, value);
    return launched("publish_rows");
}

IPSX_API int ipsx_scan_range_if(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
                                int64_t it_begin, int64_t it_end, int64_t* mem_idx, float* mem_score,
                                int32_t* tie_flag, const int32_t* cond, int32_t cond_mask, void* stream) {
    IPSX_REQUIRE(cond && cond_mask, "scan_range_if: needs the condition word and a mask");
    IPSX_REQUIRE(scan_fast_plan(m, i, h, n_token).ok, "scan_range_if: shapes of the LDS-resident loop only");
    return scan_range_impl(logits, b, n, m, i, h, n_token, it_begin, it_end, mem_idx, mem_score, tie_flag, nullptr, nullptr,
                           nullptr, 0, stream, cond, cond_mask);
}

static int scan_range_impl(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
                           int64_t it_begin, int64_t it_end, int64_t* mem_idx, float* mem_score,
                           int32_t* tie_flag, const int32_t* ready, int32_t* status, void* workspace,
                           size_t workspace_bytes, void* stream, const int32_t* cond, int32_t cond_mask, int ready_stride,
                           int workgroups) {
    IPSX_REQUIRE(logits && mem_idx, "scan: null pointer");
    IPSX_REQUIRE(b > 0 && m > 0 && i > 0 && h > 0 && n_token > 0, "scan: bad sizes");
    IPSX_REQUIRE(n > m, "scan: needs more patches (%lld) than memory slots (%d)", (long long)n, m);
    IPSX_REQUIRE(n < ((int64_t)1 << 31), "scan: too many patches");
    IPSX_REQUIRE(it_begin >= 0 && it_begin <= it_end && it_end <= (n - m + i - 1) / i,
                 "scan: iteration range [%lld, %lld) outside the loop", (long long)it_begin, (long long)it_end);
    if (it_begin == it_end) return IPSX_OK;
    const FastPlan fp = scan_fast_plan(m, i, h, n_token);
    ScanCall c = {logits, b, n, m, i, h, n_token, it_begin, it_end, mem_idx, mem_score, tie_flag, ready, status, workspace,
                  workspace_bytes, stream, cond, cond_mask, ready_stride, workgroups};
    // every shape the LDS-resident loops do not cover - other head / token counts, candidate sets beyond the LDS (the
    // reference's shipped CAMELYON configuration: M = I = 5000): scan_large_kernel (forced generic: plain launches only)
    if (!fp.ok || (g_scan_generic && !ready && !cond)) return launch_scan_large(c);
    // BASELINE configs[3] (8 heads, one token, M = I = 256): the specialised loop
    if (g_scan_r8 && scan_cam_shape(m, i, h, n_token)) return launch_scan_cam(c);
    return launch_scan_fast(c, fp);
}

IPSX_API int ipsx_set_persistent_wait_ms(int ms) {
    const int prev = g_persist_wait_ms;
    if (ms > 0) g_persist_wait_ms = ms > 20000 ? 20000 : ms;
    return prev;
}

IPSX_API int ipsx_set_tie_order(int mode) {
    const int prev = ipsx::g_tie_order;
    if (mode >= 0 && mode <= 2) ipsx::g_tie_order = mode;
    return prev;
}

// Diagnostic entry point (not part of include/ipsx.h): when set to a device buffer of b*8 uint64, the next
// resident scans accumulate per-phase s_memtime cycles there (tools/scan_stamps.py); NULL switches it off.
// Diagnostic (not part of include/ipsx.h): copies g_persist_log to out8 (host memory) and clears it; synchronises the device
extern "C" __attribute__((visibility("default"))) int ipsx_dbg_persist_log(unsigned long long* out8) {
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(ipsx::g_persist_log), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(ipsx::g_persist_log), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}

extern "C" __attribute__((visibility("default"))) void ipsx_dbg_scan_stamps(unsigned long long* buf) {
    ipsx::g_scan_stamps = buf;
}

// Diagnostic entry point (not part of include/ipsx.h): nonzero sends every shape through the generic loop kernel
// (scan_large_kernel) - tools/scan_compare.py holds the two loop kernels against each other this way.
extern "C" __attribute__((visibility("default"))) void ipsx_dbg_scan_generic(int on) { g_scan_generic = on != 0; }

extern "C" __attribute__((visibility("default"))) void ipsx_dbg_scan_direct(int on) { g_scan_direct = on != 0; }
// Diagnostic: workgroups per image for candidate sets beyond the LDS (-1 the default, 0 / 1 one workgroup, 2 / 4 / 8)
extern "C" __attribute__((visibility("default"))) void ipsx_dbg_scan_team(int w) { g_scan_team = w; }
// Diagnostic: 0 = the team's ranking always from whole runs (the path taken when the runs' top halves are not provably enough)
extern "C" __attribute__((visibility("default"))) void ipsx_dbg_scan_team_trunc(int on) { g_scan_team_trunc = on != 0; }
extern "C" __attribute__((visibility("default"))) void ipsx_dbg_scan_r8(int on) { g_scan_r8 = on != 0; }
