// topm.hip - Transformer.get_scores on arbitrary embeddings (ipsx_scores; reference architecture/transformer.py:143-148)
// and torch.topk(scores, M)[1] with torch's CPU tie order (ipsx_topm; reference architecture/ips_net.py:148): the pieces of
// IPSNet.score_and_select as stand-alone entry points.

#include <algorithm>
#include <cstdlib>

#include "scan_common.h"

namespace ipsx {

// Transformer.get_scores on the logits (b, L, R) of arbitrary embeddings
struct ScoresArgs {
    const float* lg;
    int L, h, T, use_lds;
    float* scores;    // (b, L)
    float* attn;      // (b, h, T, L) or nullptr
};

__global__ __launch_bounds__(256) void scores_kernel(ScoresArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int R = a.h * a.T;
    float* rmax = reinterpret_cast<float*>(smem);
    float* rden = rmax + R;
    float* cl = a.use_lds ? rden + R : nullptr;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* lg = a.lg + (size_t)b * a.L * R;
    CandView v;
    v.cl = cl; v.lg = lg; v.cand = nullptr; v.R = R;
    if (cl) {
        for (int e = tid; e < a.L * R; e += 256) {
            const int l = e / R, r = e - l * R;
            cl[l * (R + 1) + r] = lg[e];
        }
        __syncthreads();
    }
    row_stats(v, a.L, rmax, rden);
    __syncthreads();
    float* attn = a.attn ? a.attn + (size_t)b * R * a.L : nullptr;
    for (int l = tid; l < a.L; l += 256)
        a.scores[(size_t)b * a.L + l] = cand_score(v, l, a.h, a.T, rmax, rden, attn, a.L);
}



__global__ __launch_bounds__(256) void topm_kernel(TopmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keyA = reinterpret_cast<uint64_t*>(smem);
    uint64_t* keyB = keyA + a.n2;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int l = tid; l < a.n2; l += 256)
        keyA[l] = l < a.L ? rank_key(a.scores[(size_t)b * a.L + l], (uint32_t)l) : 0ull;
    uint64_t* sorted = sort_desc(keyA, keyB, a.L, a.n2);
    if (a.tie && tid == 0)
        a.tie[b] = (a.L > a.m && (sorted[a.m - 1] >> 32) == (sorted[a.m] >> 32)) ? 1 : 0;
    if (a.tie_order != 0 && ranked_ties(sorted, a.L, a.m, tid & 63))
        torch_tie_order<256>(sorted, sorted == keyA ? keyB : keyA, a.L, a.m, reinterpret_cast<int*>(smem + a.stk_off), tid);
    for (int j = tid; j < a.m; j += 256) a.top[(size_t)b * a.m + j] = key_pos(sorted[j]);
}

}  // namespace ipsx

using namespace ipsx;

IPSX_API size_t ipsx_scores_workspace_bytes(int b, int l, int d, int h, int n_token) {
    return (((size_t)b * l * h * n_token * sizeof(float) + 255) & ~(size_t)255) + ipsx_folded_query_elems(h, n_token, d) * sizeof(float);
}

IPSX_API int ipsx_scores(const float* x, const float* wk_packed, const float* qs, int b, int l, int d, int h,
                         int dk, int n_token, float* scores, float* attn, void* workspace,
                         size_t workspace_bytes, void* stream) {
    IPSX_REQUIRE(x && wk_packed && qs && scores, "scores: null pointer");
    IPSX_REQUIRE(b > 0 && l > 0 && d > 0 && h > 0 && dk > 0 && n_token > 0, "scores: bad sizes");
    const size_t need = ipsx_scores_workspace_bytes(b, l, d, h, n_token);
    if (!workspace || workspace_bytes < need)
        return fail(IPSX_EWORKSPACE, "scores: workspace %zu B < %zu B", workspace_bytes, need);
    const int R = h * n_token;
    float* lg = static_cast<float*>(workspace);
    float* vp = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + (((size_t)b * l * R * sizeof(float) + 255) & ~(size_t)255));
    IPSX_TRY(ipsx_fold_query(qs, wk_packed, h, dk, n_token, d, vp, stream));
    IPSX_TRY(ipsx_logits(x, (int64_t)l * d, nullptr, 0, vp, b, l, d, R, lg, (int64_t)l * R, stream));
    ScoresArgs a;
    const size_t base = (size_t)R * 8, stage = (size_t)l * (R + 1) * 4;
    a.lg = lg; a.L = l; a.h = h; a.T = n_token; a.use_lds = base + stage <= kLdsLimit;
    a.scores = scores; a.attn = attn;
    const size_t lds = base + (a.use_lds ? stage : 0);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scores_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    scores_kernel<<<dim3((unsigned)b), dim3(256), lds, as_stream(stream)>>>(a);
    return launched("scores");
}

IPSX_API size_t ipsx_topm_workspace_bytes(int b, int l, int m) {
    if (b <= 0 || l <= 0 || m <= 0) return 0;
    if ((size_t)next_pow2(l) * 16 + STK_BYTES <= kLdsLimit) return 0;
    return (size_t)b * topm_large_ws_per_row(l);
}

IPSX_API int ipsx_topm(const float* scores, int b, int l, int m, int64_t* top_idx, int32_t* tie_flag,
                       void* workspace, size_t workspace_bytes, void* stream) {
    IPSX_REQUIRE(scores && top_idx && b > 0 && l > 0 && m > 0 && m <= l, "topm: bad arguments (l=%d m=%d)", l, m);
    TopmArgs a;
    a.scores = scores; a.L = l; a.m = m; a.n2 = next_pow2(l);
    a.top = reinterpret_cast<long long*>(top_idx); a.tie = tie_flag;
    const size_t lds = (size_t)a.n2 * 16 + STK_BYTES;
    a.tie_order = g_tie_order; a.stk_off = (int)(lds - STK_BYTES);
    if (lds > kLdsLimit) {                                             // one key array in LDS, tie lists in the workspace
        IPSX_REQUIRE(l <= scan_large_max_l(), "topm: %d candidates - at most %d are supported", l, scan_large_max_l());
        return launch_topm_large(a, b, workspace, workspace_bytes, stream);
    }
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    topm_kernel<<<dim3((unsigned)b), dim3(256), lds, as_stream(stream)>>>(a);
    return launched("topm");
}
