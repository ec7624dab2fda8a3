// aggregate.hip - feature projector and the no-grad/eval aggregation path.
//
//  * ipsx_projector: IPSNet.get_projector (reference architecture/ips_net.py:54-60),
//    LayerNorm(no affine) -> Linear -> BatchNorm1d(eval) -> ReLU.  The LayerNorm is a
//    wavefront-per-row kernel (moments in the wave reduction order), the Linear runs on
//    the fp32 matrix cores as a 1x1 convolution (conv.hip) with the BatchNorm affine,
//    the Linear bias and the ReLU in its epilogue.
//  * ipsx_aggregate: Transformer.forward (architecture/transformer.py:85-109, 122-132):
//    cross-attention of the learned queries over the M selected embeddings, fc, residual
//    on the queries, LayerNorm, MLP, residual, LayerNorm.
//  * ipsx_head: Linear -> softmax | sigmoid (ips_net.py:72-81).
// Arithmetic order: oracle/ips_oracle.cpp orc_projector / orc_aggregate / orc_head.

#include "ipsx_common.h"
#include "ipsx_math.h"
#include "ipsx_rowstats.h"

namespace ipsx {

// ------------------------------------------------------------------ LayerNorm rows
// one wavefront per row; g/b may be null (no affine)
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* __restrict__ x, long long n, int d,
                                                             float eps, const float* __restrict__ g,
                                                             const float* __restrict__ b, float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* xr = x + (size_t)row * d;
    float s = 0.0f;
    for (int i = lane; i < d; i += 64) s = s + xr[i];
    const float mean = wave_butterfly_sum(s) / (float)d;
    float q = 0.0f;
    for (int i = lane; i < d; i += 64) { const float c = xr[i] - mean; const float c2 = c * c; q = q + c2; }
    const float var = wave_butterfly_sum(q) / (float)d;
    const float rstd = 1.0f / __builtin_sqrtf(var + eps);
    float* yr = y + (size_t)row * d;
    for (int i = lane; i < d; i += 64) {
        float v = (xr[i] - mean) * rstd;
        if (g) v = __builtin_fmaf(v, g[i], b[i]);
        yr[i] = v;
    }
}

// (mean, rstd) of every feature row for the projector's GEMM, which folds the LayerNorm into its epilogue
// (conv_nhwc_kernel<.., NORM>): the moments in the order the GEMM's own operand stream would give them
// (ipsx_rowstats.h row_moments_wave32 - projector_stream_kernel takes them off its operand registers instead).
// One wavefront per 32 rows.
__global__ __launch_bounds__(256) void row_moments_kernel(const float* __restrict__ x, long long n, int d, float eps,
                                                          float2* __restrict__ stats, int* ready, int value) {
    // (ipsx_projector_stats_publish: everything enqueued before this launch has completed and is visible - that is what
    //  the stream order of two kernels means - so the first thread can say so on behalf of a launch of its own)
    if (ready && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(ready, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const int lane = threadIdx.x & 63;
    const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
    if (row0 >= n) return;
    const float2 st = row_moments_wave32(x, row0, n, d, eps, lane);
    if (lane < 32 && row0 + lane < n) stats[row0 + lane] = st;
}

// cs[o] = sum_c w[o][c] of a Linear's (c_out, c_in) weights: ascending in float64, rounded once (oracle orc_weight_colsum)
__global__ void weight_colsum_kernel(const float* __restrict__ w, int c_out, int c_in, float* __restrict__ cs) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= c_out) return;
    const float* wr = w + (size_t)o * c_in;
    double s = 0.0;
    for (int c = 0; c < c_in; ++c) s = s + (double)wr[c];
    cs[o] = (float)s;
}

// LayerNorm of a row held in LDS by ONE wavefront (all 64 lanes call it)
__device__ __forceinline__ void layernorm_lds_row(const float* xr, int d, float eps, const float* g,
                                                  const float* b, float* yr, int lane) {
    float s = 0.0f;
    for (int i = lane; i < d; i += 64) s = s + xr[i];
    const float mean = wave_butterfly_sum(s) / (float)d;
    float q = 0.0f;
    for (int i = lane; i < d; i += 64) { const float c = xr[i] - mean; const float c2 = c * c; q = q + c2; }
    const float var = wave_butterfly_sum(q) / (float)d;
    const float rstd = 1.0f / __builtin_sqrtf(var + eps);
    for (int i = lane; i < d; i += 64) {
        float v = (xr[i] - mean) * rstd;
        v = __builtin_fmaf(v, g[i], b[i]);
        yr[i] = v;
    }
}

// ------------------------------------------------------------------ attention context
// ctx[b][t][hh*dv + j] = sum_l softmax_l(logits[b][l][hh,t]) * v[b][l][hh*dv + j]
struct CtxArgs {
    const float* lg;    // (b, m, R)
    const float* v;     // (b, m, h*dv)
    float* ctx;         // (b, T, h*dv)
    int m, h, T, dv;
};

__global__ __launch_bounds__(256) void attn_ctx_kernel(CtxArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int R = a.h * a.T, hdv = a.h * a.dv;
    float* rmax = sm;
    float* rden = sm + R;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lg = a.lg + (size_t)b * a.m * R;
    for (int r = wave; r < R; r += 4) {
        float mx = -__builtin_huge_valf();
        for (int i = lane; i < a.m; i += 64) mx = nanmax(mx, lg[(size_t)i * R + r]);
        mx = wave_max(mx);
        float s = 0.0f;
        for (int i = lane; i < a.m; i += 64) s = s + det_expf(lg[(size_t)i * R + r] - mx);
        s = wave_butterfly_sum(s);
        if (lane == 0) { rmax[r] = mx; rden[r] = 1.0f / s; }         // (the reciprocal: oracle orc_scores_from_logits)
    }
    __syncthreads();
    const float* v = a.v + (size_t)b * a.m * hdv;
    for (int o = tid; o < a.T * hdv; o += 256) {
        const int t = o / hdv, col = o - t * hdv, hh = col / a.dv;
        const int r = hh * a.T + t;
        const float mx = rmax[r], den = rden[r];
        float acc = 0.0f;
        for (int l = 0; l < a.m; ++l) {
            const float w = det_expf(lg[(size_t)l * R + r] - mx) * den;
            acc = __builtin_fmaf(w, v[(size_t)l * hdv + col], acc);
        }
        a.ctx[((size_t)b * a.T + t) * hdv + col] = acc;
    }
}

// ------------------------------------------------------------------ fc + LN + MLP + LN
struct TailArgs {
    const float* ctx;   // (b, T, hdv)
    ipsx_transf t;
    float* out;         // (b, T, d)
};

__global__ __launch_bounds__(256) void transf_tail_kernel(TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const ipsx_transf& p = a.t;
    const int T = p.n_token, d = p.d, hdv = p.h * p.dv, di = p.d_inner;
    float* ctx = sm;                 // T*hdv
    float* y = ctx + T * hdv;        // T*d
    float* z = y + T * d;            // T*d
    float* hid = z + T * d;          // T*di
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < T * hdv; i += 256) ctx[i] = a.ctx[(size_t)b * T * hdv + i];
    __syncthreads();
    // fc (no bias) + residual on the learned queries
    for (int o = tid; o < T * d; o += 256) {
        const int t = o / d, c = o - t * d;
        const float* w = p.fc + (size_t)c * hdv;
        float acc = 0.0f;
        for (int j = 0; j < hdv; ++j) acc = __builtin_fmaf(ctx[t * hdv + j], w[j], acc);
        y[o] = acc + p.q[o];
    }
    __syncthreads();
    for (int t = wave; t < T; t += 4) layernorm_lds_row(y + t * d, d, p.ln_eps, p.ln1_g, p.ln1_b, z + t * d, lane);
    __syncthreads();
    for (int o = tid; o < T * di; o += 256) {
        const int t = o / di, u = o - t * di;
        const float* w = p.w1 + (size_t)u * d;
        float acc = 0.0f;
        for (int c = 0; c < d; ++c) acc = __builtin_fmaf(z[t * d + c], w[c], acc);
        acc = acc + p.b1[u];
        hid[o] = acc > 0.0f ? acc : 0.0f;
    }
    __syncthreads();
    for (int o = tid; o < T * d; o += 256) {
        const int t = o / d, c = o - t * d;
        const float* w = p.w2 + (size_t)c * di;
        float acc = 0.0f;
        for (int u = 0; u < di; ++u) acc = __builtin_fmaf(hid[t * di + u], w[u], acc);
        acc = acc + p.b2[c];
        y[o] = acc + z[o];
    }
    __syncthreads();
    for (int t = wave; t < T; t += 4)
        layernorm_lds_row(y + t * d, d, p.ln_eps, p.ln2_g, p.ln2_b, a.out + ((size_t)b * T + t) * d, lane);
}

// ------------------------------------------------------------------ task head
__global__ __launch_bounds__(64) void head_kernel(const float* __restrict__ emb, int n_token, int d, int token,
                                                  const float* __restrict__ w, const float* __restrict__ bias,
                                                  int n_class, int act, float* __restrict__ out) {
    // one wavefront per image; lane c computes class c, c+64, ...
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* e = emb + ((size_t)b * n_token + token) * d;
    float mx = -__builtin_huge_valf();
    for (int c = lane; c < n_class; c += 64) {
        float acc = 0.0f;
        for (int j = 0; j < d; ++j) acc = __builtin_fmaf(e[j], w[(size_t)c * d + j], acc);
        acc = acc + bias[c];
        out[(size_t)b * n_class + c] = acc;          // logits, overwritten below
        mx = acc > mx ? acc : mx;
    }
    if (act == 0) {
        mx = wave_max(mx);
        float s = 0.0f;
        for (int c = lane; c < n_class; c += 64) {
            const float ex = det_expf(out[(size_t)b * n_class + c] - mx);
            out[(size_t)b * n_class + c] = ex;
            s = s + ex;
        }
        s = wave_butterfly_sum(s);
        for (int c = lane; c < n_class; c += 64) out[(size_t)b * n_class + c] = out[(size_t)b * n_class + c] / s;
    } else {
        for (int c = lane; c < n_class; c += 64) {
            const float zc = out[(size_t)b * n_class + c];
            out[(size_t)b * n_class + c] = 1.0f / (1.0f + det_expf(-zc));
        }
    }
}

// conv_nhwc.hip
int conv_nhwc_impl(const ipsx_conv* cv, const float* x, const float* residual, const float* row_stats, float* y,
                   int64_t n, int h, int w, int relu, void* stream, int* ready = nullptr, int ready_value = 0);

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct AggLayout {
    size_t qs, wkp, vq, wvp, lg, v, ctx, total;
};

static AggLayout agg_layout(const ipsx_transf* t, int b, int m) {
    AggLayout L;
    const int hdk = t->h * t->dk, hdv = t->h * t->dv, R = t->h * t->n_token;
    size_t off = 0;
    L.qs = off;  off += align256((size_t)t->n_token * hdk * 4);
    L.wkp = off; off += align256(ipsx_packed_conv_weight_elems(hdk, t->d, 1, 1) * 4);
    L.vq = off;  off += align256(ipsx_folded_query_elems(t->h, t->n_token, t->d) * 4);
    L.wvp = off; off += align256(ipsx_packed_conv_weight_elems(hdv, t->d, 1, 1) * 4);
    L.lg = off;  off += align256((size_t)b * m * R * 4);
    L.v = off;   off += align256((size_t)b * m * hdv * 4);
    L.ctx = off; off += align256((size_t)b * t->n_token * hdv * 4);
    L.total = off;
    return L;
}

}  // namespace ipsx

using namespace ipsx;

IPSX_API int ipsx_projector_stats(const float* x, int64_t n, int f, float ln_eps, float* stats, void* stream) {
    IPSX_REQUIRE(x && stats && n >= 0 && f > 0 && f % 8 == 0, "projector_stats: bad arguments (the row length is a multiple of 8)");
    if (n == 0) return IPSX_OK;
    row_moments_kernel<<<dim3((unsigned)cdiv(n, 128)), dim3(256), 0, as_stream(stream)>>>(x, n, f, ln_eps, reinterpret_cast<float2*>(stats),
                                                                                        nullptr, 0);
    return launched("projector row moments");
}

IPSX_API int ipsx_weight_colsum(const float* w, int c_out, int c_in, float* colsum, void* stream) {
    IPSX_REQUIRE(w && colsum && c_out > 0 && c_in > 0, "weight_colsum: bad arguments");
    weight_colsum_kernel<<<dim3((unsigned)cdiv(c_out, 64)), dim3(64), 0, as_stream(stream)>>>(w, c_out, c_in, colsum);
    return launched("weight_colsum");
}

IPSX_API int ipsx_projector_apply(const ipsx_conv* lin, const float* x, int64_t n, const float* stats, float* out,
                                  void* stream) {
    IPSX_REQUIRE(lin && x && out && stats && n >= 0, "projector_apply: bad arguments");
    IPSX_REQUIRE(lin->colsum, "projector_apply: lin->colsum (ipsx_weight_colsum) is needed: the LayerNorm is folded into the epilogue");
    IPSX_REQUIRE(lin->kh == 1 && lin->kw == 1 && lin->stride == 1 && lin->pad == 0, "projector: lin must be 1x1");
    if (n == 0) return IPSX_OK;
    return conv_nhwc_impl(lin, x, nullptr, stats, out, n, 1, 1, 1, stream);
}

IPSX_API int ipsx_projector_apply_publish(const ipsx_conv* lin, const float* x, int64_t n, const float* stats, float* out,
                                          int32_t* ready, int32_t value, void* stream) {
    IPSX_REQUIRE(lin && x && out && stats && ready && n > 0 && lin->colsum, "projector_apply_publish: bad arguments");
    IPSX_REQUIRE(lin->kh == 1 && lin->kw == 1 && lin->stride == 1 && lin->pad == 0, "projector: lin must be 1x1");
    return conv_nhwc_impl(lin, x, nullptr, stats, out, n, 1, 1, 1, stream, ready, value);
}

IPSX_API int ipsx_projector(const ipsx_conv* lin, const float* x, int64_t n, float ln_eps, float* out,
                            void* workspace, size_t workspace_bytes, void* stream) {
    IPSX_REQUIRE(lin && x && out && n >= 0, "projector: bad arguments");
    if (n == 0) return IPSX_OK;
    // LayerNorm is folded into the Linear: a statistics pass leaves (mean, rstd) per row (8 B per row - the only
    // workspace) and the GEMM applies them in its epilogue, rstd * (x W^T - mean * colsum(W)); the normalised rows
    // never exist
    const size_t need = (size_t)n * 2 * sizeof(float);
    if (!workspace || workspace_bytes < need)
        return fail(IPSX_EWORKSPACE, "projector: workspace %zu B < %zu B", workspace_bytes, need);
    IPSX_TRY(ipsx_projector_stats(x, n, lin->c_in, ln_eps, static_cast<float*>(workspace), stream));
    return ipsx_projector_apply(lin, x, n, static_cast<const float*>(workspace), out, stream);
}

IPSX_API size_t ipsx_projector_workspace_bytes(int64_t n) { return n > 0 ? (size_t)n * 2 * sizeof(float) : 0; }

IPSX_API size_t ipsx_aggregate_workspace_bytes(const ipsx_transf* t, int b, int m) {
    if (!t || b <= 0 || m <= 0) return 0;
    return agg_layout(t, b, m).total;
}

IPSX_API int ipsx_aggregate(const ipsx_transf* t, const float* x, int b, int m, float* out, void* workspace,
                            size_t workspace_bytes, void* stream) {
    return ipsx_aggregate_packed(t, nullptr, nullptr, x, b, m, out, workspace, workspace_bytes, stream);
}

IPSX_API int ipsx_aggregate_packed(const ipsx_transf* t, const float* vq_packed, const float* wv_packed, const float* x, int b,
                                   int m, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    IPSX_REQUIRE(t && x && out && b > 0 && m > 0, "aggregate: bad arguments");
    IPSX_REQUIRE(t->q && t->wq && t->wk && t->wv && t->fc && t->ln1_g && t->ln1_b && t->w1 && t->b1 && t->w2 &&
                     t->b2 && t->ln2_g && t->ln2_b, "aggregate: missing weights");
    const AggLayout L = agg_layout(t, b, m);
    if (!workspace || workspace_bytes < L.total)
        return fail(IPSX_EWORKSPACE, "aggregate: workspace %zu B < %zu B", workspace_bytes, L.total);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    float* qs = reinterpret_cast<float*>(ws + L.qs);
    float* wkp = reinterpret_cast<float*>(ws + L.wkp);
    float* vq = reinterpret_cast<float*>(ws + L.vq);
    float* wvp = reinterpret_cast<float*>(ws + L.wvp);
    float* lg = reinterpret_cast<float*>(ws + L.lg);
    float* v = reinterpret_cast<float*>(ws + L.v);
    float* ctx = reinterpret_cast<float*>(ws + L.ctx);
    const int hdk = t->h * t->dk, hdv = t->h * t->dv, R = t->h * t->n_token;
    hipStream_t s = as_stream(stream);

    // the folded query and the packed V weights depend on the parameters only: a caller that keeps them between calls
    // (ipsx_fold_query / ipsx_pack_conv_weight, re-made when a parameter moves) saves four small launches per call
    if (!vq_packed) {
        IPSX_TRY(ipsx_query_proj(t->q, t->wq, t->temperature, t->n_token, t->d, hdk, qs, stream));
        IPSX_TRY(ipsx_pack_conv_weight(t->wk, hdk, t->d, 1, 1, wkp, stream));
        IPSX_TRY(ipsx_fold_query(qs, wkp, t->h, t->dk, t->n_token, t->d, vq, stream));
        vq_packed = vq;
    }
    if (!wv_packed) {
        IPSX_TRY(ipsx_pack_conv_weight(t->wv, hdv, t->d, 1, 1, wvp, stream));
        wv_packed = wvp;
    }
    IPSX_TRY(ipsx_logits(x, (int64_t)m * t->d, nullptr, 0, vq_packed, b, m, t->d, R, lg, (int64_t)m * R, stream));
    ipsx_conv vproj;
    vproj.c_in = t->d; vproj.c_out = hdv; vproj.kh = vproj.kw = 1; vproj.stride = 1; vproj.pad = 0;
    vproj.w_packed = wv_packed; vproj.alpha = nullptr; vproj.shift = nullptr;
    IPSX_TRY(ipsx_conv2d_affine_nhwc(&vproj, x, nullptr, v, (int64_t)b * m, 1, 1, 0, stream));

    CtxArgs c;
    c.lg = lg; c.v = v; c.ctx = ctx; c.m = m; c.h = t->h; c.T = t->n_token; c.dv = t->dv;
    attn_ctx_kernel<<<dim3((unsigned)b), dim3(256), (size_t)R * 8, s>>>(c);
    IPSX_TRY(launched("attn_ctx"));

    TailArgs ta;
    ta.ctx = ctx; ta.t = *t; ta.out = out;
    const size_t lds = ((size_t)t->n_token * (hdv + 2 * t->d + t->d_inner)) * sizeof(float);
    IPSX_REQUIRE(lds <= 160 * 1024, "aggregate: token state (%zu B) exceeds LDS", lds);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(transf_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    transf_tail_kernel<<<dim3((unsigned)b), dim3(256), lds, s>>>(ta);
    return launched("transf_tail");
}

IPSX_API int ipsx_head(const float* emb, int b, int n_token, int d, int token, const float* w, const float* bias,
                       int n_class, int act, float* out, void* stream) {
    IPSX_REQUIRE(emb && w && bias && out && b > 0 && d > 0 && n_class > 0, "head: bad arguments");
    IPSX_REQUIRE(token >= 0 && token < n_token && (act == 0 || act == 1), "head: token %d / act %d", token, act);
    head_kernel<<<dim3((unsigned)b), dim3(64), 0, as_stream(stream)>>>(emb, n_token, d, token, w, bias, n_class,
                                                                       act, out);
    return launched("head");
}
