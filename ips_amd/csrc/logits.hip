// logits.hip - the cross-attention scorer's per-patch part: query projection, the query folded into the key weights, the
// attention logits of every patch (fp32 matrix cores; bf16 for BASELINE configs[4]).
//
// Reference: MultiHeadCrossAttention.get_attn (architecture/transformer.py:71-83), ScaledDotProductAttention.compute_attn
// (:29-34).  A patch's logits  l[h,t] = (q_w q)[t,h,:]/sqrt(Dk) . (k_w (emb+pos))[h,:]  depend on that patch alone; only
// the softmax denominator depends on the candidate set - so they are computed ONCE per patch here and the selection loops
// (scan_*.hip) replay the reference loop on them.  Arithmetic order: oracle/ips_oracle.cpp orc_fold_query, orc_logits.

#include <algorithm>
#include <cstdlib>

#include "scan_common.h"

namespace ipsx {

// ------------------------------------------------------------------ query projection
__global__ void query_proj_kernel(const float* __restrict__ q, const float* __restrict__ wq, float temperature,
                                  int n_token, int d, int hdk, float* __restrict__ qs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_token * hdk) return;
    const int t = i / hdk, o = i - t * hdk;
    float a = 0.0f;
    for (int c = 0; c < d; ++c) a = __builtin_fmaf(q[t * d + c], wq[(size_t)o * d + c], a);
    qs[i] = a / temperature;
}

// ------------------------------------------------------------------ logits
// logit[n][h*T + t] = sum_j qs[t][h,j] * sum_c W_k[h*dk + j][c] * x[n][c],  x = emb (+ pos).  The reference evaluates
// the inner sum first (a D x H*Dk projection per patch, transformer.py:77); the two sums commute, and here the query
// is folded into the key weights once per call:  V[h*T + t][c] = sum_j qs[t][h,j] W_k[h*dk + j][c]  (fold_query_kernel,
// j ascending), after which a patch costs H*T*D multiply-adds instead of H*Dk*D (+ H*T*Dk) and the kernel is bound by
// reading the embeddings.  The oracle restates exactly this order (orc_fold_query + orc_logits).
//
// wkp: k_w.weight packed as a 1x1 conv ([C_out/32][K/8][64 lanes][4]); vp: V in the same packing (C_out = H*T).
__global__ void fold_query_kernel(const float* __restrict__ qs, const float* __restrict__ wkp, int h, int dk, int T, int d,
                                  int kgs, int r_pad, float* __restrict__ vp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // over r_pad x (kgs * 8)
    const int dpad = kgs * 8;
    if (idx >= r_pad * dpad) return;
    const int r = idx / dpad, c = idx - r * dpad;
    const int R = h * T, hdk = h * dk;
    float acc = 0.0f;
    if (r < R && c < d) {
        const int hh = r / T, t = r - hh * T;
        const int kg = c >> 3, sub = c & 7;
        for (int j = 0; j < dk; ++j) {
            const int o = hh * dk + j;
            const float w = wkp[(((size_t)(o >> 5) * kgs + kg) * 64 + (o & 31) + 32 * (sub >> 2)) * 4 + (sub & 3)];
            acc = __builtin_fmaf(qs[(size_t)t * hdk + o], w, acc);
        }
    }
    vp[(((size_t)(r >> 5) * kgs + (c >> 3)) * 64 + (r & 31) + 32 * ((c & 7) >> 2)) * 4 + (c & 3)] = acc;
}

struct LogitsArgs {
    const float* emb; long long emb_bs;
    const float* pos; long long pos_bs;
    const float* vp;             // folded query, packed (fold_query_kernel)
    long long n;
    int d, R, kgs;
    float* out; long long out_bs;
};

// One wavefront = 32 patches x all H*T logits (NT tiles of 32 columns); 4 wavefronts per workgroup.
template <int NT>
__device__ __forceinline__ void logits_wave(const LogitsArgs& a, unsigned bx, int bi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    const long long r0 = ((long long)bx * 4 + wave) * 32;
    if (r0 >= a.n) return;                                           // wave-uniform
    const long long row = r0 + (lane & 31);
    const bool rv = row < a.n;
    const float* e = a.emb + (size_t)bi * a.emb_bs + (size_t)(rv ? row : 0) * a.d + 4 * half;
    const float* p = a.pos ? a.pos + (size_t)bi * a.pos_bs + (size_t)(rv ? row : 0) * a.d + 4 * half : nullptr;

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const float4* vq = reinterpret_cast<const float4*>(a.vp) + lane;
    const bool vec = (a.d & 7) == 0;                 // rows are 16-byte aligned and every k-group is complete
    int kg0 = 0;
    if (vec) {
        // eight k-groups per trip, every operand load of the trip in flight before its first MFMA: with one wave per SIMD
        // (a part of a slide is ~100 workgroups) nothing else hides the load latency - 64 dependent trips of ~0.5 us
        // were the whole 35 us of the kernel.  The MFMA sequence of every accumulator is unchanged.
        // (two register sets: the loads of the next trip are issued before this trip's MFMAs)
        float4 ev[2][8], bv[2][NT][8];
        auto fetch = [&](int k0, int set) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ev[set][u] = *reinterpret_cast<const float4*>(e + (k0 + u) * 8);
                if (p) {
                    const float4 pv = *reinterpret_cast<const float4*>(p + (k0 + u) * 8);
                    ev[set][u].x = ev[set][u].x + pv.x; ev[set][u].y = ev[set][u].y + pv.y;
                    ev[set][u].z = ev[set][u].z + pv.z; ev[set][u].w = ev[set][u].w + pv.w;
                }
#pragma unroll
                for (int i = 0; i < NT; ++i) bv[set][i][u] = vq[((size_t)i * a.kgs + k0 + u) * 64];
            }
        };
        auto mma = [&](int set) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float a0 = rv ? ev[set][u].x : 0.0f, a1 = rv ? ev[set][u].y : 0.0f;
                const float a2 = rv ? ev[set][u].z : 0.0f, a3 = rv ? ev[set][u].w : 0.0f;
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[set][i][u].x, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[set][i][u].y, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, bv[set][i][u].z, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, bv[set][i][u].w, acc[i], 0, 0, 0);
                }
            }
        };
        if (a.kgs >= 8) {
            fetch(0, 0);
            for (; kg0 + 16 <= a.kgs; kg0 += 16) {
                fetch(kg0 + 8, 1);
                mma(0);
                if (kg0 + 24 <= a.kgs) fetch(kg0 + 16, 0);
                mma(1);
            }
            if (kg0 + 8 <= a.kgs) { mma(0); kg0 += 8; }
        }
    }
    for (int kg = kg0; kg < a.kgs; ++kg) {
        float av[4];
        if (vec) {                                   // one 16-byte load per operand row per k-group
            const float4 ev = *reinterpret_cast<const float4*>(e + kg * 8);
            av[0] = ev.x; av[1] = ev.y; av[2] = ev.z; av[3] = ev.w;
            if (p) {
                const float4 pv = *reinterpret_cast<const float4*>(p + kg * 8);
                av[0] = av[0] + pv.x; av[1] = av[1] + pv.y; av[2] = av[2] + pv.z; av[3] = av[3] + pv.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = rv ? av[j] : 0.0f;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = kg * 8 + j;              // + 4*half is in the base pointers
                float v = (c + 4 * half < a.d) ? e[c] : 0.0f;
                if (p) v = v + ((c + 4 * half < a.d) ? p[c] : 0.0f);
                av[j] = rv ? v : 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const float4 b = vq[((size_t)i * a.kgs + kg) * 64];
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], b.x, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], b.y, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], b.z, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], b.w, acc[i], 0, 0, 0);
        }
    }
    // C layout: lane = column (logit index), registers = rows (patches)
    float* out = a.out + (size_t)bi * a.out_bs;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int o = i * 32 + (lane & 31);
        if (o < a.R) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long rr = r0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (rr < a.n) out[(size_t)rr * a.R + o] = acc[i][r];
            }
        }
    }
}

template <int NT>
__global__ __launch_bounds__(256) void logits_kernel(LogitsArgs a) { logits_wave<NT>(a, blockIdx.x, (int)blockIdx.y); }

// The same logits and, in the same launch (workgroups beyond the logits'), the LayerNorm row moments of the NEXT slab of
// feature rows: two short, latency-bound kernels of the CAMELYON pipeline that sat one after the other between two GEMM
// parts.  No publication in here (a device-wide release per workgroup costs more than the launch it would save: the next
// launch in the stream publishes, ipsx_projector_apply_publish).
template <int NT>
__global__ __launch_bounds__(256) void logits_stats_kernel(LogitsArgs a, unsigned n_logits_x, const float* __restrict__ sx,
                                                           long long sn, int sf, float eps, float2* __restrict__ sout) {
    if (blockIdx.x < n_logits_x) { logits_wave<NT>(a, blockIdx.x, (int)blockIdx.y); return; }
    if (blockIdx.y != 0) return;
    const int lane = threadIdx.x & 63;                               // one wavefront per 32 rows (row_moments_kernel's arithmetic)
    const long long row0 = ((long long)(blockIdx.x - n_logits_x) * 4 + (threadIdx.x >> 6)) * 32;
    if (row0 >= sn) return;
    const float2 st = row_moments_wave32(sx, row0, sn, sf, eps, lane);
    if (lane < 32 && row0 + lane < sn) sout[row0 + lane] = st;
}

// ---- the same logits on the bf16 matrix pipe (BASELINE configs[4]: "MFMA bf16/fp16 QK^T path").  x = emb (+ pos)
// rounded to bfloat16 in the A-operand load, the folded query rounded to bfloat16 once per call, fp32 accumulation
// (v_mfma_f32_32x32x16_bf16: lane l holds row / column l & 31 and the 8 consecutive k of half l >> 5).  No reference
// behaviour exists for reduced precision; checked against logits_kernel with a tolerance.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// vq16: [R/32 tiles][D/16 k-steps][64 lanes][8 bf16]; element (r, c) at tile r>>5, step c>>4, lane (r&31) + 32*((c&15)>>3), j = c&7
__global__ void fold_query_bf16_kernel(const float* __restrict__ qs, const float* __restrict__ wkp, int h, int dk, int T, int d,
                                       int kgs, int ksteps, int r_pad, unsigned short* __restrict__ vq16) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // over r_pad x (ksteps * 16)
    const int dpad = ksteps * 16;
    if (idx >= r_pad * dpad) return;
    const int r = idx / dpad, c = idx - r * dpad;
    const int R = h * T, hdk = h * dk;
    float acc = 0.0f;
    if (r < R && c < d) {                                           // the fp32 fold of fold_query_kernel (j ascending)
        const int hh = r / T, t = r - hh * T;
        const int kg = c >> 3, sub = c & 7;
        for (int j = 0; j < dk; ++j) {
            const int o = hh * dk + j;
            const float w = wkp[(((size_t)(o >> 5) * kgs + kg) * 64 + (o & 31) + 32 * (sub >> 2)) * 4 + (sub & 3)];
            acc = __builtin_fmaf(qs[(size_t)t * hdk + o], w, acc);
        }
    }
    const __bf16 hv = (__bf16)acc;
    vq16[(((size_t)(r >> 5) * ksteps + (c >> 4)) * 64 + (r & 31) + 32 * ((c & 15) >> 3)) * 8 + (c & 7)] =
        __builtin_bit_cast(unsigned short, hv);
}

template <int NT>
__global__ __launch_bounds__(256) void logits_bf16_kernel(LogitsArgs a, const uint4* __restrict__ vq16, int ksteps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    const long long r0 = ((long long)blockIdx.x * 4 + wave) * 32;
    if (r0 >= a.n) return;                                           // wave-uniform
    const int bi = blockIdx.y;
    const long long row = r0 + (lane & 31);
    const bool rv = row < a.n;
    const float* e = a.emb + (size_t)bi * a.emb_bs + (size_t)(rv ? row : 0) * a.d + 8 * half;
    const float* p = a.pos ? a.pos + (size_t)bi * a.pos_bs + (size_t)(rv ? row : 0) * a.d + 8 * half : nullptr;
    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int ks = 0; ks < ksteps; ++ks) {
        float xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = ks * 16 + 8 * half + j;
            float v = (rv && c < a.d) ? e[ks * 16 + j] : 0.0f;
            if (p) v = v + ((rv && c < a.d) ? p[ks * 16 + j] : 0.0f);
            xv[j] = v;
        }
        bf16x8_t av;
#pragma unroll
        for (int j = 0; j < 8; ++j) av[j] = (__bf16)xv[j];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const uint4 b = vq16[((size_t)i * ksteps + ks) * 64 + lane];
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8_t, b), acc[i], 0, 0, 0);
        }
    }
    float* out = a.out + (size_t)bi * a.out_bs;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int o = i * 32 + (lane & 31);
        if (o < a.R) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long rr = r0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (rr < a.n) out[(size_t)rr * a.R + o] = acc[i][r];
            }
        }
    }
}

static int launch_logits(const LogitsArgs& a, int b, hipStream_t s) {
    const int nt = (a.R + 31) / 32;
    IPSX_REQUIRE(nt <= 8, "logits: H * n_token = %d > 256 not supported", a.R);
    dim3 grid((unsigned)cdiv(a.n, 128), (unsigned)b);
    if (nt == 1) logits_kernel<1><<<grid, dim3(256), 0, s>>>(a);
    else if (nt == 2) logits_kernel<2><<<grid, dim3(256), 0, s>>>(a);
    else if (nt <= 4) logits_kernel<4><<<grid, dim3(256), 0, s>>>(a);
    else logits_kernel<8><<<grid, dim3(256), 0, s>>>(a);
    return launched("logits");
}

}  // namespace ipsx

using namespace ipsx;

IPSX_API int ipsx_query_proj(const float* q, const float* wq, float temperature, int n_token, int d, int hdk,
                             float* qs, void* stream) {
    IPSX_REQUIRE(q && wq && qs && n_token > 0 && d > 0 && hdk > 0, "query_proj: bad arguments");
    query_proj_kernel<<<dim3((unsigned)cdiv(n_token * hdk, 256)), dim3(256), 0, as_stream(stream)>>>(
        q, wq, temperature, n_token, d, hdk, qs);
    return launched("query_proj");
}

IPSX_API size_t ipsx_folded_query_elems(int h, int n_token, int d) {
    if (h <= 0 || n_token <= 0 || d <= 0) return 0;
    const int nt = (h * n_token + 31) / 32;
    return ipsx_packed_conv_weight_elems((nt <= 2 ? nt : (nt <= 4 ? 4 : 8)) * 32, d, 1, 1);     // whole tiles of the kernel variant
}

IPSX_API int ipsx_fold_query(const float* qs, const float* wk_packed, int h, int dk, int n_token, int d,
                             float* v_packed, void* stream) {
    IPSX_REQUIRE(qs && wk_packed && v_packed && h > 0 && dk > 0 && n_token > 0 && d > 0, "fold_query: bad arguments");
    const int kgs = (int)cdiv(d, 8);
    const int r_pad = (int)(ipsx_folded_query_elems(h, n_token, d) / ((size_t)kgs * 8));
    fold_query_kernel<<<dim3((unsigned)cdiv((int64_t)r_pad * kgs * 8, 256)), dim3(256), 0, as_stream(stream)>>>(
        qs, wk_packed, h, dk, n_token, d, kgs, r_pad, v_packed);
    return launched("fold_query");
}

IPSX_API int ipsx_logits(const float* emb, int64_t emb_bstride, const float* pos, int64_t pos_bstride,
                         const float* v_packed, int b, int64_t n, int d, int r, float* logits,
                         int64_t logits_bstride, void* stream) {
    IPSX_REQUIRE(emb && v_packed && logits, "logits: null pointer");
    IPSX_REQUIRE(b > 0 && n >= 0 && d > 0 && r > 0, "logits: bad sizes");
    if (n == 0) return IPSX_OK;
    LogitsArgs a;
    a.emb = emb; a.emb_bs = emb_bstride; a.pos = pos; a.pos_bs = pos_bstride;
    a.vp = v_packed; a.n = n; a.d = d; a.R = r;
    a.kgs = (int)cdiv(d, 8);
    a.out = logits; a.out_bs = logits_bstride;
    return launch_logits(a, b, as_stream(stream));
}

IPSX_API int ipsx_logits_stats(const float* emb, int64_t emb_bstride, const float* pos, int64_t pos_bstride,
                               const float* v_packed, int b, int64_t n, int d, int r, float* logits, int64_t logits_bstride,
                               const float* stats_x, int64_t stats_n, int stats_f, float ln_eps, float* stats_out,
                               void* stream) {
    IPSX_REQUIRE(emb && v_packed && logits && stats_x && stats_out, "logits_stats: null pointer");
    IPSX_REQUIRE(b > 0 && n > 0 && d > 0 && r > 0 && stats_n > 0 && stats_f > 0, "logits_stats: bad sizes");
    LogitsArgs a;
    a.emb = emb; a.emb_bs = emb_bstride; a.pos = pos; a.pos_bs = pos_bstride;
    a.vp = v_packed; a.n = n; a.d = d; a.R = r;
    a.kgs = (int)cdiv(d, 8);
    a.out = logits; a.out_bs = logits_bstride;
    const unsigned nlx = (unsigned)cdiv(n, 128);
    const int nt = (r + 31) / 32;
    IPSX_REQUIRE(nt <= 8, "logits_stats: H * n_token = %d > 256 not supported", r);
    IPSX_REQUIRE(stats_f % 8 == 0, "logits_stats: the feature rows' length is a multiple of 8");
    dim3 grid(nlx + (unsigned)cdiv(stats_n, 128), (unsigned)b);
    hipStream_t s = as_stream(stream);
    float2* so = reinterpret_cast<float2*>(stats_out);
    if (nt == 1) logits_stats_kernel<1><<<grid, dim3(256), 0, s>>>(a, nlx, stats_x, stats_n, stats_f, ln_eps, so);
    else if (nt == 2) logits_stats_kernel<2><<<grid, dim3(256), 0, s>>>(a, nlx, stats_x, stats_n, stats_f, ln_eps, so);
    else if (nt <= 4) logits_stats_kernel<4><<<grid, dim3(256), 0, s>>>(a, nlx, stats_x, stats_n, stats_f, ln_eps, so);
    else logits_stats_kernel<8><<<grid, dim3(256), 0, s>>>(a, nlx, stats_x, stats_n, stats_f, ln_eps, so);
    return launched("logits_stats");
}

IPSX_API size_t ipsx_folded_query_bf16_bytes(int h, int n_token, int d) {
    if (h <= 0 || n_token <= 0 || d <= 0) return 0;
    return (size_t)cdiv(h * n_token, 32) * (size_t)cdiv(d, 16) * 64 * 16;
}

IPSX_API int ipsx_fold_query_bf16(const float* qs, const float* wk_packed, int h, int dk, int n_token, int d,
                                  void* v_packed_bf16, void* stream) {
    IPSX_REQUIRE(qs && wk_packed && v_packed_bf16 && h > 0 && dk > 0 && n_token > 0 && d > 0, "fold_query_bf16: bad arguments");
    const int R = h * n_token, r_pad = (int)cdiv(R, 32) * 32, ksteps = (int)cdiv(d, 16), kgs = (int)cdiv(d, 8);
    const int total = r_pad * ksteps * 16;
    fold_query_bf16_kernel<<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream)>>>(
        qs, wk_packed, h, dk, n_token, d, kgs, ksteps, r_pad, static_cast<unsigned short*>(v_packed_bf16));
    return launched("fold_query_bf16");
}

IPSX_API int ipsx_logits_bf16(const float* emb, int64_t emb_bstride, const float* pos, int64_t pos_bstride,
                              const void* v_packed_bf16, int b, int64_t n, int d, int r, float* logits,
                              int64_t logits_bstride, void* stream) {
    IPSX_REQUIRE(emb && v_packed_bf16 && logits && b > 0 && n >= 0 && d > 0 && r > 0, "logits_bf16: bad arguments");
    IPSX_REQUIRE(r <= 128, "logits_bf16: at most 128 logits per patch (got %d)", r);
    if (n == 0) return IPSX_OK;
    LogitsArgs a;
    a.emb = emb; a.emb_bs = emb_bstride; a.pos = pos; a.pos_bs = pos_bstride; a.vp = nullptr; a.n = n; a.d = d; a.R = r;
    a.kgs = 0; a.out = logits; a.out_bs = logits_bstride;
    const int ksteps = (int)cdiv(d, 16), nt = (int)cdiv(r, 32);
    const dim3 grid((unsigned)cdiv(n, 128), (unsigned)b), block(256);
    const uint4* vq = static_cast<const uint4*>(v_packed_bf16);
    hipStream_t s = as_stream(stream);
    if (nt == 1) logits_bf16_kernel<1><<<grid, block, 0, s>>>(a, vq, ksteps);
    else if (nt == 2) logits_bf16_kernel<2><<<grid, block, 0, s>>>(a, vq, ksteps);
    else logits_bf16_kernel<4><<<grid, block, 0, s>>>(a, vq, ksteps);
    return launched("logits_bf16");
}
