// scorer.hip - the cross-attention patch scorer and the IPS selection loop.
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

#include "ipsx_common.h"
#include "ipsx_math.h"
#include "ipsx_rowstats.h"
#include "ipsx_stdorder.h"

namespace ipsx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// ------------------------------------------------------------------ scoring of a candidate set
// Candidate logits either staged in LDS (cl, row stride R+1) or read through `cand`
// from the global (n, R) table of one image.
struct CandView {
    const float* cl;       // LDS staging or nullptr
    const float* lg;       // global logits of this image, (n, R)
    const int* cand;       // LDS: candidate -> patch index (nullptr = identity)
    int R;
    __device__ __forceinline__ float get(int l, int r) const {
        if (cl) return cl[l * (R + 1) + r];
        const size_t row = cand ? (size_t)cand[l] : (size_t)l;
        return lg[row * R + r];
    }
};

// per-(h,t) row maximum and softmax denominator over L candidates (wave per row)
__device__ __forceinline__ void row_stats(const CandView& v, int L, float* rmax, float* rden) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int r = wave; r < v.R; r += nw) {
        float m = -__builtin_huge_valf();
        for (int i = lane; i < L; i += 64) m = nanmax(m, v.get(i, r));
        m = wave_max(m);
        float s = 0.0f;
        for (int i = lane; i < L; i += 64) s = s + det_expf(v.get(i, r) - m);
        s = wave_butterfly_sum(s);
        if (lane == 0) { rmax[r] = m; rden[r] = 1.0f / s; }        // (the RECIPROCAL: weights are e * (1 / den), oracle orc_scores_from_logits)
    }
}

// score of candidate l: mean over heads, then over tokens, of its attention weights
__device__ __forceinline__ float cand_score(const CandView& v, int l, int h, int T, const float* rmax,
                                            const float* rden, float* attn, int L) {
    float st = 0.0f;
    for (int t = 0; t < T; ++t) {
        float sh = 0.0f;
        for (int hh = 0; hh < h; ++hh) {
            const int r = hh * T + t;
            const float a = det_expf(v.get(l, r) - rmax[r]) * rden[r];
            if (attn) attn[((size_t)hh * T + t) * L + l] = a;
            sh = sh + a;
        }
        st = st + sh / (float)h;
    }
    return st / (float)T;
}

// Sort `n2` (power of two) keys descending.  src holds the keys; the sorted keys end
// up in the returned buffer (src or tmp).  Keys are unique.  Must be called by all
// threads of the workgroup; contains barriers.
__device__ __forceinline__ uint64_t* sort_desc(uint64_t* src, uint64_t* tmp, int L, int n2) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (L <= 512) {
        // rank by counting: rank = number of larger keys (LDS broadcast reads, no barriers inside)
        __syncthreads();
        for (int l = tid; l < L; l += nt) {
            const uint64_t k = src[l];
            int rank = 0;
            for (int j = 0; j < L; ++j) rank += (src[j] > k) ? 1 : 0;
            tmp[rank] = k;
        }
        __syncthreads();
        return tmp;
    }
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = tid; i < n2; i += nt) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t x = src[i], y = src[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) { src[i] = y; src[ixj] = x; }
                }
            }
        }
    }
    __syncthreads();
    return src;
}

// ---- exact score ties (ipsx_stdorder.h).  `sorted` = keys in canonical order (score desc, earlier
// position first).  When two of the first m+1 ranked scores are equal the reference returns whatever libstdc++'s
// nth_element / sort / partial_sort leave behind, and that order feeds the next iteration; with tie order 1
// ("torch", the default) one lane replays those routines on the candidate array and rewrites sorted[0..m).
static int g_tie_order = 1;     // 0 canonical; 1 (default) torch.topk's order where bit-identical candidates tie; 2 wherever scores tie
constexpr int STK_BYTES = 3 * stdorder::STACK_RANGES * 4;

// every wave evaluates this on the same data: the result is uniform over the workgroup without a barrier
__device__ __forceinline__ bool ranked_ties(const uint64_t* sorted, int L, int m, int lane) {
    const int n = m < L - 1 ? m : L - 1;
    bool any = false;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        const bool e = j < n && (sorted[j] >> 32) == (sorted[j + 1] >> 32);
        any = any || (__ballot(e) != 0ull);
    }
    return any;
}

// the same test on the padded key array of the large kernels (key j at slot j + (j >> 4))
// where the logit rows of a loop's candidates live (scan_large_kernel): candidate p < m is memory slot p = patch mem[p],
// candidate p >= m is patch lo + (p - m); R floats per patch.  lg == nullptr: no rows (ipsx_topm: scores only)
struct TieRows { const float* lg; const long long* mem; long long lo; int m, R; };

// ties among the first m + 1 canonical ranks that call for torch.topk's order: tie_order 2 (and callers without rows) any
// two neighbours of equal score; tie_order 1, the loop's rule (oracle orc_topm_loop): neighbours of equal score whose logit
// rows are bit-identical
// (the whole workgroup calls this: the pairs are dealt out to the wavefronts, the verdict is a barrier's OR - every
//  wavefront going through all 5,000 pairs of the shipped CAMELYON sizes on its own was 13 us of a 94 us iteration)
__device__ __forceinline__ bool ranked_ties_padded(const uint64_t* sorted, int L, int m, int lane, int tie_order = 2,
                                                   const TieRows* rows = nullptr) {
    const int n = m < L - 1 ? m : L - 1;
    const bool by_rows = tie_order == 1 && rows != nullptr && rows->lg != nullptr;
    bool any = false;
    for (int j = (int)threadIdx.x; j < n; j += (int)blockDim.x) {
        const uint64_t ka = sorted[j + (j >> 4)], kb = sorted[j + 1 + ((j + 1) >> 4)];
        bool e = (ka >> 32) == (kb >> 32);
        if (e && by_rows) {
            const long long pa = key_pos(ka), pb = key_pos(kb);
            const float* ra = rows->lg + (size_t)(pa < rows->m ? rows->mem[pa] : rows->lo + (pa - rows->m)) * rows->R;
            const float* rb = rows->lg + (size_t)(pb < rows->m ? rows->mem[pb] : rows->lo + (pb - rows->m)) * rows->R;
            for (int r = 0; r < rows->R; ++r) e = e && as_u32(ra[r]) == as_u32(rb[r]);
        }
        any = any || e;
    }
    (void)lane;
    return __syncthreads_or(any ? 1 : 0) != 0;
}

// ---- the replay on ONE WAVEFRONT instead of one lane.  libstdc++'s routines are sequential, but what they compute is
// not: (1) the unguarded Hoare partition pairs the t-th element from the left that stops the upward scan (not greater than
// the pivot) with the t-th from the right that stops the downward scan (not smaller), for as long as the left one lies
// before the right one, swaps each pair, and returns where the upward scan stops next - both scans only ever see
// elements no swap has touched, so the pairs can be read off the ORIGINAL array with two ballots per 64 elements;
// (2) the final insertion pass of std::sort never moves an element across a partition cut and is a stable sort, so every
// element of a leaf of the introsort loop (<= 16 elements) finds its place by counting, a lane per element.  Everything else (median of three, the
// <= 3-element tail of nth_element, the heap fallbacks at depth 0) stays the sequential restatement on lane 0.
// oracle/check_stdorder.cpp holds this formulation against std:: itself; tests/test_hip_kernels.py holds the device code
// against the one-lane replay.  One lane took ~600 k cycles for 512 candidates (a quarter of a millisecond - 10 % of a
// CAMELYON slide for ONE tie among 255 iterations); the wavefront takes ~130 k.
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// GL = the index lists la / lb live in GLOBAL memory (candidate sets beyond the LDS, scan_large_kernel): what one lane
// wrote there must be visible to the lane that reads it next - a workgroup-scope fence (the lists never leave the
// compute unit's L1 / its write-through path) on top of the LDS wait.  q stays in LDS either way.
template <bool GL>
__device__ __forceinline__ void wave_fence() {
    if (GL) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
        wave_lds_fence();
    }
}

template <bool GL>
__device__ __forceinline__ int wave_partition(stdorder::E* q, int first, int last, const stdorder::E P, int* la, int* lb, int lane) {
    const unsigned long long below = (1ull << lane) - 1ull;
    int na = 0, nb = 0;
    for (int base = first; base < last; base += 64) {                  // ascending: indices that stop the upward scan
        const int x = base + lane;
        const bool in = x < last;
        const stdorder::E e = q[in ? x : first];
        const bool stop = in && !stdorder::gt(e, P);
        const unsigned long long mask = __ballot(stop);
        if (stop) la[na + __popcll(mask & below)] = x;
        na += __popcll(mask);
    }
    for (int top = last - 1; top >= first; top -= 64) {                // descending: indices that stop the downward scan
        const int x = top - lane;
        const bool in = x >= first;
        const stdorder::E e = q[in ? x : first];
        const bool stop = in && !stdorder::gt(P, e);
        const unsigned long long mask = __ballot(stop);
        if (stop) lb[nb + __popcll(mask & below)] = x;
        nb += __popcll(mask);
    }
    wave_fence<GL>();
    const int np = na < nb ? na : nb;
    int t = 0;                                                         // pairs that are swapped: la[u] < lb[u], a prefix
    for (int base = 0; base < np; base += 64) {                        // (the pairs are disjoint: swapped as they are found)
        const int u = base + lane;
        const int i = la[u < np ? u : 0], j = lb[u < np ? u : 0];
        const bool ok = u < np && i < j;
        const unsigned long long mask = __ballot(ok);
        if (ok) {
            const stdorder::E ei = q[i], ej = q[j];
            q[i] = ej;
            q[j] = ei;
        }
        const int valid = np - base < 64 ? np - base : 64;
        const unsigned long long full = valid == 64 ? ~0ull : ((1ull << valid) - 1ull);
        if ((mask & full) == full) { t += valid; continue; }
        t += __ffsll((long long)(~mask)) - 1;
        break;
    }
    const int cut = (t < na && (t == 0 || la[t] < lb[t - 1])) ? la[t] : lb[t - 1];
    wave_fence<GL>();
    return cut;
}

template <bool GL>
__device__ __forceinline__ int wave_partition_pivot(stdorder::E* q, int first, int last, int* la, int* lb, int lane) {
    // std::__move_median_to_first(first, first + 1, mid, last - 1): the four elements are read at once (one round trip,
    // every lane the same addresses), the decision is the restatement's, lane 0 does the swap
    const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
    const stdorder::E er = q[first], ea = q[ia], eb = q[ib], ec = q[ic];
    int sel;
    if (stdorder::gt(ea, eb)) sel = stdorder::gt(eb, ec) ? ib : (stdorder::gt(ea, ec) ? ic : ia);
    else sel = stdorder::gt(ea, ec) ? ia : (stdorder::gt(eb, ec) ? ic : ib);
    const stdorder::E P = sel == ia ? ea : (sel == ib ? eb : ec);
    if (lane == 0) { q[first] = P; q[sel] = er; }
    wave_fence<GL>();
    return wave_partition<GL>(q, first + 1, last, P, la, lb, lane);
}

// q[0..n) = (score, position) in candidate order on entry; q[0..k) = torch.topk's answer on return.  Called by the 64
// lanes of ONE wavefront.  la / lb: n ints each (lb = la + n: the two together are the n-element scratch of the last
// pass); stk: 2 * STACK_RANGES ints of pending ranges + the leaf bitmap `leaf` of `leaf_words` 64-bit words (n <= 64 *
// leaf_words, else one lane runs the sequential restatement).
template <bool GL>
__device__ __forceinline__ void torch_topk_wave(stdorder::E* q, int n, int k, int* la, int* lb, int* stk,
                                                unsigned long long* leaf, int leaf_words, int lane) {
    using namespace stdorder;
    if (k <= 0 || n <= 0) return;
    if ((long long)k * 64 <= (long long)n || n > 64 * leaf_words) {    // heap select / sort, or beyond the leaf bitmap
        if (lane == 0) torch_topk(q, n, k, stk);
        wave_fence<GL>();
        return;
    }
    {   // std::nth_element(q, q + k - 1, q + n)
        int first = 0, last = n;
        const int nth = k - 1;
        bool done = nth == last;
        int depth = lg2(last - first) * 2;
        while (!done && last - first > 3) {
            if (depth == 0) {
                if (lane == 0) { heap_select(q, first, nth + 1, last); swp(q, first, nth); }
                done = true;
                break;
            }
            --depth;
            const int cut = wave_partition_pivot<GL>(q, first, last, la, lb, lane);
            if (cut <= nth) first = cut;
            else last = cut;
        }
        if (!done && lane == 0) insertion_sort(q, first, last);
        wave_fence<GL>();
    }
    const int last = k - 1;                                            // std::sort(q, q + k - 1)
    if (last <= 0) return;
    for (int w = lane; w < leaf_words; w += 64) leaf[w] = 0ull;
    wave_fence<GL>();
    int sp = 1;
    if (lane == 0) { stk[0] = 0; stk[1] = last; stk[2] = lg2(last) * 2; }
    wave_fence<GL>();
    while (sp > 0) {
        --sp;
        int rf = stk[3 * sp], rl = stk[3 * sp + 1], depth = stk[3 * sp + 2];
        while (rl - rf > 16) {
            if (depth == 0) {
                if (lane == 0) { make_heap(q, rf, rl); sort_heap(q, rf, rl); }
                wave_fence<GL>();
                break;
            }
            --depth;
            const int cut = wave_partition_pivot<GL>(q, rf, rl, la, lb, lane);
            if (lane == 0) { stk[3 * sp] = cut; stk[3 * sp + 1] = rl; stk[3 * sp + 2] = depth; }
            wave_fence<GL>();
            ++sp;
            rl = cut;
        }
        if (lane == 0) {                                               // [rf, rl) is a leaf (a heap-sorted range is one too)
            leaf[rf >> 6] |= 1ull << (rf & 63);
            if (rl < last) leaf[rl >> 6] |= 1ull << (rl & 63);
        }
        wave_fence<GL>();
    }
    // The final insertion pass, leaf by leaf.  Linear insertion is a STABLE sort (an element moves left past strictly smaller
    // ones only), so an element's place in its leaf is the number of leaf elements that are greater plus the number of
    // equivalent ones in front of it: one lane per element counts over its leaf (<= 16 independent reads) instead of one
    // lane per leaf shifting elements one dependent LDS round trip at a time.  Results go to the list scratch and back.
    // (A range that ended in the heap sort is a "leaf" of more than 16 elements and already in order: it stays.)
    E* tmp = reinterpret_cast<E*>(la);                                  // la / lb: 2 n ints = n elements
    for (int base = 0; base < last; base += 64) {
        const int x = base + lane;
        if (x < last) {
            const E own = q[x];
            int w = x >> 6;
            unsigned long long m = leaf[w] & (~0ull >> (63 - (x & 63)));
            while (m == 0ull && w > 0) m = leaf[--w];
            const int sfirst = m ? w * 64 + 63 - __clzll((long long)m) : 0;
            int e = last;
            w = x >> 6;
            m = (x & 63) == 63 ? 0ull : (leaf[w] >> ((x & 63) + 1)) << ((x & 63) + 1);
            while (m == 0ull && w < leaf_words - 1) m = leaf[++w];
            if (m) e = w * 64 + __ffsll((long long)m) - 1;
            if (e > last) e = last;
            int dst = x;
            if (e - sfirst <= 16) {
                int rank = 0;
                for (int j = sfirst; j < e; ++j) {
                    const E o = q[j];
                    const bool greater = gt(o, own);
                    const bool equiv = !greater && !gt(own, o);
                    rank += (greater || (equiv && j < x)) ? 1 : 0;
                }
                dst = sfirst + rank;
            }
            tmp[dst] = own;
        }
    }
    wave_fence<GL>();
    for (int base = 0; base < last; base += 64)
        if (base + lane < last) q[base + lane] = tmp[base + lane];
    wave_fence<GL>();
}

// ---- the replay on the whole WORKGROUP (candidate sets beyond the LDS: scan_large_kernel ranks 10,000 candidates whose
// scores lie within a binade or two, so SOME pair of the first m + 1 is bit-equal in practically every iteration and the
// replay is part of every iteration there).  std::sort's introsort loop only ever splits a range into two disjoint
// ranges that never interact again, so the order in which pending ranges are processed is immaterial: instead of one
// wavefront working through a stack, the ranges of a level are dealt to the 16 wavefronts (two range lists in LDS,
// breadth first, a barrier per level), each range partitioned by the unchanged wave_partition_pivot with its own stretch
// of the index lists (la + first, lb + first: ranges are disjoint, so are the stretches).  std::nth_element in front of it is
// ONE chain of partitions and stays on one wavefront; the final insertion pass runs on all threads.
// The unguarded partition of q[first + 1, last) around the median of three, by ALL NT threads (16 wavefronts): what
// wave_partition does with one - the t-th element from the left that stops the upward scan paired with the t-th from the
// right that stops the downward scan, swapped while the left one lies before the right one - with the stretch cut into
// one slice per wavefront: ballots per 64 elements (kept in registers), the slices' counts through LDS, an exclusive sum,
// the two index lists written in one go, the pair swaps an element per thread.  Lists in GLOBAL memory (they are as long
// as the range; every access is a coalesced bulk access behind a barrier).  For the long stretches of std::nth_element's
// chain (10,000, 5,000, 2,500 candidates: 80 k cycles on one wavefront with its lists in global memory, ~10 k here).
// `sc`: 2 * (NT / 64) + 2 ints of LDS scratch.  Returns the cut (workgroup-uniform).  Contains barriers.
constexpr int BLOCK_PART_CHUNKS = 16;       // 64-element chunks per wavefront slice: stretches up to 16 * 16 * 64 = 16,384

template <int NT>
__device__ __forceinline__ int block_partition_pivot(stdorder::E* q, int first, int last, int* la, int* lb, int* sc) {
    using namespace stdorder;
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {   // std::__move_median_to_first(first, first + 1, mid, last - 1)
        const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
        const E er = q[first], ea = q[ia], eb = q[ib], ec = q[ic];
        int sel;
        if (gt(ea, eb)) sel = gt(eb, ec) ? ib : (gt(ea, ec) ? ic : ia);
        else sel = gt(ea, ec) ? ia : (gt(eb, ec) ? ic : ib);
        const E P = sel == ia ? ea : (sel == ib ? eb : ec);
        q[first] = P; q[sel] = er;
        sc[2 * NW] = 0x7fffffff;                                       // t: first pair that is not swapped
    }
    __syncthreads();
    const E P = q[first];
    const int f = first + 1, len = last - f;
    const int per = ((len + NW - 1) / NW + 63) & ~63;                  // slice length, whole chunks
    const int s0 = f + wave * per, s1 = min(last, s0 + per);           // this wavefront's slice
    int ca = 0, cb = 0;                                                // (two passes over the slice: the ballots are taken again
    for (int c = 0; c < BLOCK_PART_CHUNKS; ++c) {                      //  in the second instead of living in 64 registers)
        const int x = s0 + c * 64 + lane;
        if (s0 + c * 64 >= s1) break;
        const bool in = x < s1;
        const E e = q[in ? x : first];
        ca += __popcll(__ballot(in && !gt(e, P)));                     // stops the upward scan
        cb += __popcll(__ballot(in && !gt(P, e)));                     // stops the downward scan
    }
    if (lane == 0) { sc[wave] = ca; sc[NW + wave] = cb; }
    __syncthreads();
    int base_a = 0, base_b = 0, na = 0, nb = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int a_ = sc[w], b_ = sc[NW + w];
        base_a += w < wave ? a_ : 0;                                   // la ascends: lower slices first
        base_b += w > wave ? b_ : 0;                                   // lb descends: higher slices first
        na += a_; nb += b_;
    }
    {
        const unsigned long long below = (1ull << lane) - 1ull, above = lane == 63 ? 0ull : (~0ull << (lane + 1));
        int pa = base_a, pb = base_b + cb;                             // pb: end of this slice's stretch of lb
        for (int c = 0; c < BLOCK_PART_CHUNKS; ++c) {
            const int x = s0 + c * 64 + lane;
            if (s0 + c * 64 >= s1) break;
            const bool in = x < s1;
            const E e = q[in ? x : first];
            const unsigned long long ma = __ballot(in && !gt(e, P)), mb = __ballot(in && !gt(P, e));
            if ((ma >> lane) & 1ull) la[pa + __popcll(ma & below)] = x;
            pa += __popcll(ma);
            pb -= __popcll(mb);                                        // the chunk's stops, highest index first, start here
            if ((mb >> lane) & 1ull) lb[pb + __popcll(mb & above)] = x;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    const int np = na < nb ? na : nb;
    int tmin = 0x7fffffff;
    for (int u = tid; u < np; u += NT) {
        const int i = la[u], j = lb[u];
        if (i < j) {
            const E ei = q[i], ej = q[j];
            q[i] = ej;
            q[j] = ei;
        } else {
            tmin = u < tmin ? u : tmin;
        }
    }
    for (int off = 32; off >= 1; off >>= 1) tmin = min(tmin, __shfl_xor(tmin, off, 64));
    if (lane == 0 && tmin != 0x7fffffff) atomicMin(&sc[2 * NW], tmin);
    __syncthreads();
    int t = sc[2 * NW];
    t = t < np ? t : np;
    const int cut = (t < na && (t == 0 || la[t] < lb[t - 1])) ? la[t] : lb[t - 1];
    __syncthreads();                                                   // sc and the lists are reused by the next call
    return cut;
}

__device__ __forceinline__ uint64_t wave_sort_desc(uint64_t key, int lane);        // (defined with the ranking helpers below)

constexpr int BLOCK_QCAP = 1024;            // ranges of more than 16 elements pending at one level: <= 16,384 / 17
constexpr int BLOCK_RANGE = 2048;           // ranges of std::sort longer than this are partitioned by the whole workgroup

// Diagnostic (ipsx_dbg_persist_log; tools/soak.py): what the gate and the resident loops saw, on the 100 MHz clock -
// [0] gate launches, [1] longest gate wait (ticks), [2] gate waits that ran into their bound, [3] start of the last gate,
// [4] the moment the last loop became resident, [5] loops that gave up waiting for rows, [6] start of the slowest gate,
// [7] the moment the loop it waited for became resident (0: not before the gate gave up)
__device__ unsigned long long g_persist_log[8];

// Diagnostic (ipsx_dbg_replay_stamps): shader cycles of the replay's phases, summed by thread 0 of every workgroup -
// [0] nth_element by the workgroup, [1] its chain on one wavefront, [2 .. 5] the first four levels of std::sort's
// partitions, [6] the deeper levels, [7] the final insertion pass; [8] = replays counted.
__device__ unsigned long long g_replay_t[10];
#define RSTAMP(k)                                                          \
    do {                                                                   \
        if (rst && tid == 0) {                                             \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();    \
            g_replay_t[k] += t_ - rlast;                                   \
            rlast = t_;                                                    \
        }                                                                  \
    } while (0)

// la_n / lb_n: the lists of std::nth_element's chain (n ints each, GLOBAL memory); la / lb: the lists of std::sort's
// ranges and, together, the scratch of the final pass (k - 1 ints each; GL = in global memory, else in LDS).
template <int NT, bool GL>
__device__ __forceinline__ void torch_topk_block(stdorder::E* q, int n, int k, int* la_n, int* lb_n, int* la, int* lb, int* stk,
                                                 unsigned long long* leaf, int leaf_words, int* queue, int* qcount,
                                                 const unsigned long long* tiebits, const uint64_t* canon = nullptr,
                                                 bool rst = false) {
    using namespace stdorder;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (k <= 0 || n <= 0) return;
    unsigned long long rlast = rst && tid == 0 ? __builtin_amdgcn_s_memtime() : 0ull;       // (rst: diagnostic stamps on)
    if (rst && tid == 0) g_replay_t[8] += 1;
    if ((long long)k * 64 <= (long long)n || n > 64 * leaf_words || (k - 1) / 17 + 1 > BLOCK_QCAP) {
        if (tid == 0) torch_topk(q, n, k, stk);                       // heap select / sort (partial_sort), or beyond the tables
        __syncthreads();
        return;
    }
    {   // std::nth_element(q, q + k - 1, q + n): long stretches by the whole workgroup, the rest of the chain by wave 0
        int first = 0, last = n;
        const int nth = k - 1;
        bool done = nth == last;
        int depth = lg2(last - first) * 2;
        while (!done && last - first > 2048 && depth > 0) {            // (workgroup-uniform)
            --depth;
            const int cut = block_partition_pivot<NT>(q, first, last, la_n, lb_n, queue);
            if (cut <= nth) first = cut;
            else last = cut;
        }
        RSTAMP(0);
        if (wave == 0) {
            while (!done && last - first > 3) {
                if (depth == 0) {
                    if (lane == 0) { heap_select(q, first, nth + 1, last); swp(q, first, nth); }
                    done = true;
                    break;
                }
                --depth;
                // (the sort phase's lists are free until then: in LDS - not GL - a partition of this chain is a few LDS
                //  round trips instead of a few L2 round trips)
                const int cut = (!GL && last - first <= k - 1) ? wave_partition_pivot<false>(q, first, last, la, lb, lane)
                                                               : wave_partition_pivot<true>(q, first, last, la_n, lb_n, lane);
                if (cut <= nth) first = cut;
                else last = cut;
            }
            if (!done && lane == 0) insertion_sort(q, first, last);
            wave_fence<true>();
        }
    }
    __syncthreads();
    RSTAMP(1);
    const int last = k - 1;                                            // std::sort(q, q + k - 1)
    for (int w = tid; w < leaf_words; w += NT) leaf[w] = 0ull;
    // range lists: one word per range, first | last << 16 (both < 2^15); the depth budget of std::sort's introsort loop
    // falls by one per partition, i.e. it is the same for every range of a level
    if (tid == 0) {
        qcount[0] = qcount[1] = 0;
        if (last > 16) {
            queue[0] = 0 | (last << 16);
            qcount[0] = 1;
        }
    }
    __syncthreads();
    if (last <= 0) return;
    int depth = lg2(last) * 2;
    int level = 0;
    for (int cur = 0;; cur ^= 1, --depth, ++level) {
        const int ncur = qcount[cur];
        if (ncur == 0) break;
        const int* qc = queue + cur * BLOCK_QCAP;
        int* qn = queue + (cur ^ 1) * BLOCK_QCAP;
        // tie bits j in [b0, b1] of word w (bit j: the canonical ranks j and j + 1 have equal scores)
        auto tie_word = [&](int w, int b0, int b1) -> unsigned long long {
            unsigned long long m = tiebits[w];
            if (w == (b0 >> 6)) m &= ~0ull << (b0 & 63);
            if (w == (b1 >> 6)) m &= ~0ull >> (63 - (b1 & 63));
            return m;
        };
        // A range of introsort holds the elements of final ranks [rf, rl).  When no two neighbouring ranks in there AND
        // across its two ends have equal scores, these are exactly the canonical ranks [rf, rl) and std::sort can only
        // leave them in the one strictly descending order: they are copied from the canonical ranking (`canon`, kept
        // in global memory by the caller) instead of being partitioned level by level - with a handful of equal pairs
        // among thousands of candidates, only the ranges on the way to those pairs are still replayed.
        // (1) ranges of more than BLOCK_RANGE elements: the whole workgroup, one range after the other
        for (int r = 0; r < ncur; ++r) {
            const int rf = qc[r] & 0xFFFF, rl = (int)((unsigned)qc[r] >> 16);
            if (rl - rf <= BLOCK_RANGE || depth <= 0) continue;           // (workgroup-uniform)
            if (tiebits && canon) {
                const int b0 = rf > 0 ? rf - 1 : 0, b1 = rl - 1;
                bool anyb = false;
                for (int w = (b0 >> 6) + tid; w <= (b1 >> 6); w += NT) anyb |= tie_word(w, b0, b1) != 0ull;
                if (!__syncthreads_or(anyb ? 1 : 0)) {
                    for (int x = rf + tid; x < rl; x += NT) {
                        const uint64_t key = canon[x];
                        E o; o.v = key_score(key); o.i = (int)key_pos(key);
                        q[x] = o;
                    }
                    continue;
                }
            }
            const int cut = block_partition_pivot<NT>(q, rf, rl, la_n, lb_n, stk);
            if (tid == 0) {
                if (cut < last) atomicOr(&leaf[cut >> 6], 1ull << (cut & 63));
                if (cut - rf > 16) qn[atomicAdd(&qcount[cur ^ 1], 1)] = rf | (cut << 16);
                if (rl - cut > 16) qn[atomicAdd(&qcount[cur ^ 1], 1)] = cut | (rl << 16);
            }
        }
        // (2) the others: a range per wavefront
        for (int r = wave; r < ncur; r += NT / 64) {
            const int rf = qc[r] & 0xFFFF, rl = (int)((unsigned)qc[r] >> 16);
            if (depth <= 0) {                                          // heap sort of the range: stays as it is afterwards
                if (lane == 0) { make_heap(q, rf, rl); sort_heap(q, rf, rl); }
                continue;
            }
            if (rl - rf > BLOCK_RANGE) continue;                       // (done above)
            if (tiebits && rl - rf <= 64) {
                // No equal neighbours INSIDE a range of at most 64: one in-register wave sort of the elements that are there
                // instead of replaying two more levels of partitions and the leaves.  (With ties inside, the replay goes on.)
                const int lo_w = rf >> 6, hi_w = (rl - 2) >> 6;                              // pairs (j, j + 1), j in [rf, rl - 2]
                unsigned long long any = 0ull;
                for (int w = lo_w; w <= hi_w; ++w) any |= tie_word(w, rf, rl - 2);
                if (any == 0ull) {
                    const int x = rf + lane;
                    const E e = q[x < rl ? x : rf];
                    const uint64_t key = wave_sort_desc(x < rl ? rank_key(e.v, (uint32_t)e.i) : 0ull, lane);
                    if (x < rl) { E o; o.v = key_score(key); o.i = (int)key_pos(key); q[x] = o; }
                    continue;                                          // (a "leaf" of more than 16 elements: the last pass leaves it)
                }
            } else if (tiebits && canon) {
                const int b0 = rf > 0 ? rf - 1 : 0, b1 = rl - 1;
                bool anyb = false;
                for (int w = (b0 >> 6) + lane; w <= (b1 >> 6); w += 64) anyb |= tie_word(w, b0, b1) != 0ull;
                if (__ballot(anyb) == 0ull) {
                    for (int x = rf + lane; x < rl; x += 64) {
                        const uint64_t key = canon[x];
                        E o; o.v = key_score(key); o.i = (int)key_pos(key);
                        q[x] = o;
                    }
                    continue;
                }
            }
            const int cut = wave_partition_pivot<GL>(q, rf, rl, la + rf, lb + rf, lane);
            if (lane == 0) {
                if (cut < last) atomicOr(&leaf[cut >> 6], 1ull << (cut & 63));      // every leaf starts at 0 or at a cut
                if (cut - rf > 16) qn[atomicAdd(&qcount[cur ^ 1], 1)] = rf | (cut << 16);
                if (rl - cut > 16) qn[atomicAdd(&qcount[cur ^ 1], 1)] = cut | (rl << 16);
            }
        }
        __syncthreads();
        if (tid == 0) qcount[cur] = 0;
        __syncthreads();
        RSTAMP(level < 4 ? 2 + level : 6);
    }
    // the final insertion pass, leaf by leaf, an element per thread (see torch_topk_wave)
    E* tmp = reinterpret_cast<E*>(la);
    for (int base = 0; base < last; base += NT) {
        const int x = base + tid;
        if (x < last) {
            const E own = q[x];
            // the leaf of x: [last cut <= x (or 0), first cut > x (or last)).  Only leaves of at most 16 elements are
            // touched, so the cuts that matter lie in [x - 15, x + 16]: 32 bits of the bitmap, two words at most (with
            // whole ranges copied from the canonical ranking the cuts are sparse, and a scan for the nearest one was long)
            const int wb = x < 15 ? 0 : x - 15, wi = wb >> 6, sh = wb & 63, tx = x - wb;
            unsigned long long bits = leaf[wi] >> sh;
            if (sh && wi + 1 < leaf_words) bits |= leaf[wi + 1] << (64 - sh);
            const unsigned long long back = bits & ((2ull << tx) - 1ull);          // cuts at wb .. x
            const unsigned long long fwd = (bits >> (tx + 1)) & 0xFFFFull;           // cuts at x + 1 .. x + 16
            const int sfirst = back ? wb + 63 - __clzll((long long)back) : (wb == 0 ? 0 : -64);   // (-64: further away than 15)
            int e = fwd ? x + 1 + (__ffsll((long long)fwd) - 1) : last;
            if (e > last) e = last;
            int dst = x;
            if (e - sfirst <= 16) {
                int rank = 0;
                for (int j = sfirst; j < e; ++j) {
                    const E o = q[j];
                    const bool greater = gt(o, own);
                    const bool equiv = !greater && !gt(own, o);
                    rank += (greater || (equiv && j < x)) ? 1 : 0;
                }
                dst = sfirst + rank;
            }
            tmp[dst] = own;
        }
    }
    __syncthreads();
    for (int x = tid; x < last; x += NT) q[x] = tmp[x];
    __syncthreads();
    RSTAMP(7);
}
#undef RSTAMP

// sorted and other are the two key arrays (>= L entries each); all threads of the workgroup call this together.
// NOTE: `sorted` serves as scratch meanwhile - on return only sorted[0, m) is defined.
template <int NT>
__device__ __forceinline__ void torch_tie_order(uint64_t* sorted, uint64_t* other, int L, int m, int* stk, int tid) {
    stdorder::E* q = reinterpret_cast<stdorder::E*>(other);
    for (int j = tid; j < L; j += NT) {
        const uint64_t k = sorted[j];
        const int p = (int)key_pos(k);
        q[p].v = key_score(k);
        q[p].i = p;
    }
    __syncthreads();
    if (tid < 64) {
        int* la = reinterpret_cast<int*>(sorted);
        torch_topk_wave<false>(q, L, m, la, la + L, stk, reinterpret_cast<unsigned long long*>(stk + 2 * stdorder::STACK_RANGES), 16, tid);
    }
    __syncthreads();
    for (int j = tid; j < m; j += NT) sorted[j] = rank_key(q[j].v, (uint32_t)q[j].i);
    __syncthreads();
}

struct ScanArgs {
    int tie_order, stk_off;
    const float* lg;       // (b, n, R)
    long long n;
    long long it0, it1;    // iterations [it0, it1) of the loop; it0 > 0 resumes from mem_idx
    int m, i, h, T, n2, use_lds;
    long long* mem_idx;
    float* mem_score;
    int* tie;
    const int* ready;      // persistent launch: number of patches whose logits are in memory (grows while we run);
    int ready_stride;      //   image b polls ready[b * ready_stride] (0: one word for all images, 1: a word per image)
    int ready_words;       //   progress words of the call (1, or b): any of them moving restarts the wait's clock
    unsigned long long wait_ticks;     // persistent launch: longest wait WITHOUT any progress, in 100 MHz ticks (ipsx_set_persistent_wait_ms)
    int* status;           // persistent launch: set to 1 when the wait for `ready` timed out
    const int* cond;       // conditional launch (ipsx_scan_range_if): run only when (*cond & cond_mask) != 0, or nullptr
    int cond_mask;
    int slides;            // images of the call; a launch of fewer workgroups (scan_cam_kernel) gives workgroup w the
                           // images w, w + gridDim.x, ... one after the other
};

// ipsx_scan_range_if: the recovery launch behind a persistent loop - every workgroup looks at the word the loop sets when
// it gave up waiting and leaves at once when it is clear (workgroup-uniform).
__device__ __forceinline__ bool scan_skipped(const int* cond, int mask) {
    return cond != nullptr && (__hip_atomic_load(cond, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & mask) == 0;
}

// The selection loop kernels: 1024 threads (16 wavefronts) per image - the loop is a chain of short VALU-bound phases
// (two exp + one division per candidate x (head, token)), and 4 waves per SIMD give them 4x the issue slots of a
// 256-thread block.  scan_fast_kernel keeps the candidates' logits and exponentials in LDS (every shape the reference
// ships); scan_large_kernel is the generic one (any head / token count, up to 16,384 candidates, staging through a
// caller workspace).  Same arithmetic order in both and in the oracle: wave-order row sums, ascending sums over heads
// then tokens.
constexpr int SCAN_NT = 1024;
constexpr int SCAN_PF = 4;     // prefetch registers per thread: chunk <= 1024 * 4 floats
__host__ __device__ constexpr int scan_pf(int R, int lch) { return (R == 32 && lch == 8) ? 5 : SCAN_PF; }

// Small candidate sets: rank by counting with P lanes per candidate (P = power of two <= 64,
// P * L <= blockDim): lane `part` counts the keys j = part, part+P, ... that are larger; the partial
// counts are added by an xor butterfly over the P lanes (integer adds: order-free).  O(L^2 / P).
__device__ __forceinline__ void rank_scatter(const uint64_t* src, uint64_t* dst, int L, int P) {
    const int tid = threadIdx.x;
    const int l = tid / P, part = tid & (P - 1);
    int cnt = 0;
    uint64_t k = 0ull;
    if (l < L) {
        k = src[l];
        for (int j = part; j < L; j += P) cnt += (src[j] > k) ? 1 : 0;
    }
    for (int off = P >> 1; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (l < L && part == 0) dst[cnt] = k;
}

// Ranking of L <= 64 * (waves per block) unique keys, descending, without an O(L^2) pass:
//   1. wave w bitonic-sorts keys [64w, 64w+64) in registers (lane shuffles, no barrier) and
//      publishes the sorted run;
//   2. every key's rank = its position in its own run + for each other run the number of
//      larger keys there, found by a branch-free binary search (7 LDS reads per run, the
//      searches of all runs in flight together).
// src/dst hold L keys (dst gets them sorted); runs is scratch for 64 * ceil(L/64) keys.
// partner's key for the exchange with lane ^ J: register to register for every stride (lane_xor_*, ipsx_math.h)
template <int J>
__device__ __forceinline__ uint64_t xor_partner(uint64_t key, int lane) { return lane_xor_u64<J>(key, lane); }

template <int J>
__device__ __forceinline__ float xor_partner_f32(float v, int lane) { return lane_xor_f32<J>(v, lane); }

// the wavefront reductions of the contract (xor butterfly, offsets 32 ... 1) on two values at once
__device__ __forceinline__ void wave_max2(float& a, float& b, int lane) {
    a = nanmax(a, xor_partner_f32<32>(a, lane)); b = nanmax(b, xor_partner_f32<32>(b, lane));
    a = nanmax(a, xor_partner_f32<16>(a, lane)); b = nanmax(b, xor_partner_f32<16>(b, lane));
    a = nanmax(a, xor_partner_f32<8>(a, lane)); b = nanmax(b, xor_partner_f32<8>(b, lane));
    a = nanmax(a, xor_partner_f32<4>(a, lane)); b = nanmax(b, xor_partner_f32<4>(b, lane));
    a = nanmax(a, xor_partner_f32<2>(a, lane)); b = nanmax(b, xor_partner_f32<2>(b, lane));
    a = nanmax(a, xor_partner_f32<1>(a, lane)); b = nanmax(b, xor_partner_f32<1>(b, lane));
}

__device__ __forceinline__ void wave_sum2(float& a, float& b, int lane) {
    a = a + xor_partner_f32<32>(a, lane); b = b + xor_partner_f32<32>(b, lane);
    a = a + xor_partner_f32<16>(a, lane); b = b + xor_partner_f32<16>(b, lane);
    a = a + xor_partner_f32<8>(a, lane); b = b + xor_partner_f32<8>(b, lane);
    a = a + xor_partner_f32<4>(a, lane); b = b + xor_partner_f32<4>(b, lane);
    a = a + xor_partner_f32<2>(a, lane); b = b + xor_partner_f32<2>(b, lane);
    a = a + xor_partner_f32<1>(a, lane); b = b + xor_partner_f32<1>(b, lane);
}

template <int K, int J>
__device__ __forceinline__ uint64_t cmpx(uint64_t key, int lane) {
    const uint64_t other = xor_partner<J>(key, lane);
    const bool take_max = ((lane & K) == 0) == ((lane & J) == 0);
    const bool gt = key > other;
    return (take_max == gt) ? key : other;
}

__device__ __forceinline__ uint64_t wave_sort_desc(uint64_t key, int lane) {
    key = cmpx<2, 1>(key, lane);
    key = cmpx<4, 2>(key, lane); key = cmpx<4, 1>(key, lane);
    key = cmpx<8, 4>(key, lane); key = cmpx<8, 2>(key, lane); key = cmpx<8, 1>(key, lane);
    key = cmpx<16, 8>(key, lane); key = cmpx<16, 4>(key, lane); key = cmpx<16, 2>(key, lane); key = cmpx<16, 1>(key, lane);
    key = cmpx<32, 16>(key, lane); key = cmpx<32, 8>(key, lane); key = cmpx<32, 4>(key, lane); key = cmpx<32, 2>(key, lane);
    key = cmpx<32, 1>(key, lane);
    key = cmpx<64, 32>(key, lane); key = cmpx<64, 16>(key, lane); key = cmpx<64, 8>(key, lane); key = cmpx<64, 4>(key, lane);
    key = cmpx<64, 2>(key, lane); key = cmpx<64, 1>(key, lane);
    return key;
}

__device__ __forceinline__ void rank_runs(const uint64_t* src, uint64_t* dst, uint64_t* runs, int L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nruns = (L + 63) >> 6;
    uint64_t mine = 0ull;
    if (wave < nruns) {
        const int idx = wave * 64 + lane;
        mine = wave_sort_desc(idx < L ? src[idx] : 0ull, lane);     // padding keys (0) sort last
        runs[idx] = mine;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // LDS traffic only
    if (wave < nruns && mine != 0ull) {
        // rank = number of larger keys over ALL runs (in the own run that is the lane index: keys are unique), found
        // by branch-free binary searches, 8 runs at a time with their LDS reads in flight together - no per-run
        // control flow, which would serialise the 7 dependent reads of every search
        int rank = 0;
        for (int r0 = 0; r0 < nruns; r0 += 8) {
            int lo[8];
            const uint64_t* base[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lo[j] = 0;
                base[j] = runs + (r0 + j < nruns ? r0 + j : nruns - 1) * 64;      // clamped: a duplicate search, not counted
            }
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                uint64_t probe[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) probe[j] = base[j][lo[j] + step - 1];
#pragma unroll
                for (int j = 0; j < 8; ++j) lo[j] += (probe[j] > mine) ? step : 0;
            }
            uint64_t last[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) last[j] = base[j][lo[j]];
#pragma unroll
            for (int j = 0; j < 8; ++j) rank += (r0 + j < nruns) ? lo[j] + ((last[j] > mine) ? 1 : 0) : 0;
        }
        dst[rank] = mine;
    }
}

// rank_runs for the fast scan: the same ranking (wave-sorted runs of 64, rank = larger keys over all runs), with the
// searches organised for LATENCY - the loop runs at two waves per SIMD, so a dependent LDS round trip costs more than
// the instructions around it: a 4-ary search (three probes per round, three rounds + one final probe for 64 keys) of
// every run, all runs' probes of a round in flight together, and no probes for runs that do not exist.
template <int NRUN>
__device__ __forceinline__ int rank_in_runs(const uint64_t* runs, uint64_t mine) {
    int lo[NRUN];                                    // number of keys of run j known to be larger than `mine`
#pragma unroll
    for (int j = 0; j < NRUN; ++j) lo[j] = 0;
#pragma unroll
    for (int step = 16; step >= 1; step >>= 2) {     // 64 = 4 * 16 -> 4 * 4 -> 4 * 1
        uint64_t p1[NRUN], p2[NRUN], p3[NRUN];
#pragma unroll
        for (int j = 0; j < NRUN; ++j) {
            const uint64_t* q = runs + j * 64 + lo[j];
            p1[j] = q[step - 1]; p2[j] = q[2 * step - 1]; p3[j] = q[3 * step - 1];
        }
#pragma unroll
        for (int j = 0; j < NRUN; ++j)               // descending run: the probes that are larger form a prefix
            lo[j] += ((p1[j] > mine) ? step : 0) + ((p2[j] > mine) ? step : 0) + ((p3[j] > mine) ? step : 0);
    }
    uint64_t last[NRUN];
#pragma unroll
    for (int j = 0; j < NRUN; ++j) last[j] = runs[j * 64 + lo[j]];
    int rank = 0;
#pragma unroll
    for (int j = 0; j < NRUN; ++j) rank += lo[j] + ((last[j] > mine) ? 1 : 0);
    return rank;
}

__device__ __forceinline__ void rank_runs4(const uint64_t* src, uint64_t* dst, uint64_t* runs, int L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nruns = (L + 63) >> 6;
    uint64_t mine = 0ull;
    if (wave < nruns) {
        const int idx = wave * 64 + lane;
        mine = wave_sort_desc(idx < L ? src[idx] : 0ull, lane);     // padding keys (0) sort last
        runs[idx] = mine;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (wave < nruns && mine != 0ull) {
        // every run is searched, the own one too (there the result is the lane index: keys are unique)
        int rank = 0;
        int r0 = 0;
        for (; r0 + 8 <= nruns; r0 += 8) rank += rank_in_runs<8>(runs + r0 * 64, mine);
        const int left = nruns - r0;                 // workgroup-uniform
        if (left >= 4) { rank += rank_in_runs<4>(runs + r0 * 64, mine); r0 += 4; }
        if (nruns - r0 == 3) rank += rank_in_runs<3>(runs + r0 * 64, mine);
        else if (nruns - r0 == 2) rank += rank_in_runs<2>(runs + r0 * 64, mine);
        else if (nruns - r0 == 1) rank += rank_in_runs<1>(runs + r0 * 64, mine);
        dst[rank] = mine;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The LDS-resident loop (R = H*T a power of two <= 64, (M+I)*R <= 1024*EPT, M+I <= 64*LCH).  One image runs on ONE compute unit, 16 waves on 4 SIMDs: an instruction every thread executes costs
// 16 issue slots, so the loop is bound by instructions per thread and by dependent LDS round trips
// (tools/scan_stamps.py).  What this organisation does about it:
//   * one thread per ELEMENT (candidate l, row r) with r fixed per thread (1024 % R == 0), EPT elements per thread: the
//     row maximum is a v_max_f32 reduction (lane steps R .. 32, then one LDS exchange between the 16 waves) instead of
//     one wave walking a whole row.  NaNs (a NaN must win, the contract's nanmax) are looked for on the side; an
//     iteration that sees one takes the exact key-based reduction instead (workgroup-uniform branch);
//   * exp(x - max) is a function of (x, max) alone, and the maximum of a row rarely moves from one iteration to the
//     next (it belongs to a patch that stays in the memory): the exponentials of the M memory rows travel with the
//     winners and only the I new rows are evaluated - unless the row's maximum changed (bitwise), then that row is
//     recomputed.  det_expf_np leaves out the overflow tests a non-positive argument cannot trigger;
//   * the contract's row sums (lane j adds elements j, j+64, ... in ascending order, then the xor butterfly) read the
//     exponentials back row-wise with all LCH reads of a lane in flight together: R short wave jobs;
//   * the attention weights e / den are formed by all 1024 threads and transposed through LDS for the per-candidate
//     ascending head / token sums;
//   * barriers wait for LDS traffic only (lds_barrier): the prefetch of the next chunk stays in flight across them;
//   * the kernel claims 128 registers per lane: 16 waves x 128 = the whole register file of the compute unit, so no
//     workgroup of the encoder running beside the loop can be placed on it and compete for its issue slots.
// LDS: two logit buffers + two exp buffers of (M+I) x (R+1) floats (the spare exp buffer doubles as the weight
// buffer and as the run scratch of the ranking).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define FAST_STAMP(k)                                                              \
    do {                                                                           \
        if (STAMP) {                                                               \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();            \
            if (tid == 0) { tacc[k] += t_ - tlast; }                               \
            tlast = t_;                                                            \
        }                                                                          \
    } while (0)

// the replay of torch.topk's tie order (rare) lives outside the loop body: inlined, its registers would be the loop's
__device__ __attribute__((noinline)) void tie_order_slow(uint64_t* sorted, uint64_t* other, int L, int m, int* stk) {
    torch_tie_order<SCAN_NT>(sorted, other, L, m, stk, threadIdx.x);
}

// PERSIST: ONE launch for the whole loop, started BEFORE the encoder has produced anything: the kernel waits (bounded)
// until `*a.ready` says the logits of the rows it is about to read exist and reads them past the vector L1 (agent-scope
// loads: the producer is another kernel that finished meanwhile).  Launched onto an idle GPU (ipsx_scan_gate holds the
// producers back until it is resident) it never has to wait for a compute unit to drain, which a workgroup of 16 waves
// does for a long time beside an encoder grid - and it claims 128 registers per lane, i.e. with 16 waves the whole
// register file of its compute unit, so no producer workgroup is placed beside it: a producer launch must then be sized
// for the OTHER compute units (one sized for all 256 runs two workgroups on one of them and takes twice as long; a
// producer workgroup sharing the loop's compute unit is a straggler that costs about as much).
// The row maxima are kept as order-preserving keys (max_key: a NaN wins, as in the contract's nanmax) in LDS, one word per
// row for the memory rows and one for the chunk rows: whoever has the values in registers anyway - the gather of the new
// memory, the prep of the next chunk, the prologue - folds them in with one ds_max_u32 per (wave, row).
template <int R>
__device__ __forceinline__ void fold_row_max(uint32_t key, uint32_t* dst, int lane) {
    if (R <= 8) key = max(key, lane_xor_u32<8>(key, lane));
    if (R <= 16) key = max(key, lane_xor_u32<16>(key, lane));
    key = max(key, lane_xor_u32<32>(key, lane));
    if (lane < R && key != 0u) atomicMax(dst + lane, key);
}

// exp(x - max) down one column of the candidate buffers (row stride ld), an element per thread; not inlined: it runs in
// the minority of iterations, and inlined its registers are the loop's (the same lesson as tie_order_slow)
__device__ __attribute__((noinline)) void exp_column(const float* xcol, float* ecol, int L, int ld, float mx) {
    for (int l = threadIdx.x; l < L; l += blockDim.x) ecol[l * ld] = det_expf_np(xcol[l * ld] - mx);
}

// ... by the threads t0, t0 + nt, ... of a part of the workgroup
__device__ __attribute__((noinline)) void exp_column_part(const float* xcol, float* ecol, int L, int ld, float mx, int t0, int nt) {
    for (int l = t0; l < L; l += nt) ecol[l * ld] = det_expf_np(xcol[l * ld] - mx);
}

template <bool PERSIST>
__device__ __forceinline__ float scan_load(const float* p) {
    if (PERSIST) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

template <int R, int T, int EPT, int LCH, bool STAMP, bool PERSIST>
__global__ __launch_bounds__(SCAN_NT) void scan_fast_kernel(ScanArgs a, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // prefetch registers per (prefetching) thread: 4, 5 for 32 rows x up to 512 candidates - the reference's shipped
    // Megapixel-MNIST sizes (M = I = 100, 4 tokens: a chunk is 3,200 logits for 704 prefetching threads)
    constexpr int PF = scan_pf(R, LCH);
    if (!PERSIST && scan_skipped(a.cond, a.cond_mask)) return;
    // encoder workgroups share this compute unit (their matrix-pipe work coexists with this VALU-bound loop); where the
    // two compete for issue slots the loop - the serial part of the job - goes first
    __builtin_amdgcn_s_setprio(3);
    if (PERSIST) {
        // 128 registers per lane x 16 waves = the whole register file of the compute unit: it is ours alone (callers
        // size the producers' launches for the remaining compute units)
        asm volatile("v_mov_b32 v127, 0" ::: "v127");
        // resident: tell the gate on the producing stream (ipsx_scan_gate) that the encoder may start
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_or(a.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&g_persist_log[4], (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = STAMP ? __builtin_amdgcn_s_memtime() : 0;
    constexpr int H = R / T, ld = R + 1;
    constexpr int log2R = R == 8 ? 3 : (R == 16 ? 4 : (R == 32 ? 5 : 6));
    constexpr int log2T = T == 1 ? 0 : (T == 2 ? 1 : (T == 4 ? 2 : 3));
    static_assert((1 << log2R) == R && (1 << log2T) == T && H * T == R, "scan_fast_kernel: R, T powers of two");
    const int Lmax = a.m + a.i;
    uint64_t* keyA = reinterpret_cast<uint64_t*>(smem);
    uint64_t* keyB = keyA + a.n2;
    int* candA = reinterpret_cast<int*>(keyB + a.n2);
    int* candB = candA + Lmax;
    uint32_t* pmax = reinterpret_cast<uint32_t*>(candB + Lmax + ((4 - ((2 * Lmax) & 3)) & 3));   // [R][16], 16-byte aligned
    uint32_t* wmin = pmax + 16 * R;               // [16] (16-byte aligned): per wave, the lowest score key of its memory rows
    int* ccount = reinterpret_cast<int*>(wmin + 16);          // [0]: chunk candidates that can still reach the top M;
                                                              // [1]: lowest memory score key; [2], [3]: tie flag (by parity)
    int* nanflag = ccount + 4;                    // [2]: a NaN among this iteration's logits (by parity)
    uint32_t* prevk = reinterpret_cast<uint32_t*>(ccount + 8);     // [2][R]: bits of the row maxima of the previous iteration
    float* rden = reinterpret_cast<float*>(prevk + 2 * R);
    float* xA = rden + R;
    float* xB = xA + (size_t)Lmax * ld;
    float* eA = xB + (size_t)Lmax * ld;
    float* eB = eA + (size_t)Lmax * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float* lg = a.lg + (size_t)b * a.n * R;
    const int r = tid & (R - 1), lrow0 = tid >> log2R;
    constexpr int lstep = SCAN_NT >> log2R;

    int* cand = candA;
    int* cnew = candB;
    float* xc = xA;
    float* xn = xB;
    float* ec = eA;
    float* en = eB;
    // persistent launch: rows below ready_known exist.  Wave 0 polls, everybody learns the result through LDS; a negative
    // value (cancelled / timed out) ends the kernel.  The wait is bounded by a.wait_ticks of the 100 MHz clock WITHOUT
    // PROGRESS: the clock restarts whenever any progress word of the call has moved (lane k watches word k), so a slide
    // whose turn at the projector comes late waits as long as the slides in front of it are being worked on - and a call
    // whose producers cannot run at all (serialised kernels) gives up after wait_ticks (default 50 ms) and is redone by
    // the conditional launch behind it.
    long long ready_known = 0;
#define SCAN_WAIT_ROWS(need)                                                                                   \
    do {                                                                                                       \
        if (PERSIST && (long long)(need) > ready_known) {                                                      \
            if (wave == 0) {                                                                                   \
                unsigned long long t0_ = __builtin_amdgcn_s_memrealtime();                                     \
                int v_ = __hip_atomic_load(a.ready + b * a.ready_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                int seen_ = -1;                                                                                \
                while (v_ >= 0 && v_ < (need)) {                                                               \
                    __builtin_amdgcn_s_sleep(16);                                                              \
                    int w_ = lane < a.ready_words ? __hip_atomic_load(a.ready + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0; \
                    for (int o_ = 32; o_ >= 1; o_ >>= 1) w_ += __shfl_xor(w_, o_, 64);                         \
                    const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();                          \
                    if (w_ != seen_) { seen_ = w_; t0_ = now_; }                                               \
                    if (now_ - t0_ > a.wait_ticks) { v_ = -1; break; }                                         \
                    v_ = __hip_atomic_load(a.ready + b * a.ready_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                }                                                                                              \
                if (lane == 0) ccount[6] = v_;                                                     \
            }                                                                                                  \
            lds_barrier();                                                                                     \
            /* the rows the producers published: every wave's loads of them are ordered after the poll that saw the */ \
            /* progress word (one agent-scope acquire per wait - an LDS barrier alone orders nothing in global memory) */ \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                                                 \
            ready_known = ccount[6];                                                               \
            if (ready_known < 0) {                                                                             \
                if (tid == 0) { __hip_atomic_fetch_or(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); g_persist_log[5] += 1; } \
                return;                                                                                        \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
    uint32_t* const mkey = pmax;                   // [R] max key of the memory rows, [R] of the chunk rows (see fold_row_max)
    uint32_t* const ckey = pmax + R;
    if (tid < 2 * R) pmax[tid] = 0u;
    lds_barrier();
    SCAN_WAIT_ROWS(std::min<long long>(a.n, a.it0 * a.i + a.m + a.i));
    {
        uint32_t km = 0u;
        for (int k = 0; k < EPT; ++k) {
            const int l = lrow0 + k * lstep;
            if (l < a.m) {
                const size_t row = a.it0 == 0 ? (size_t)l : (size_t)a.mem_idx[(size_t)b * a.m + l];
                const float v = scan_load<PERSIST>(lg + row * R + r);
                xc[l * ld + r] = v;
                km = max(km, max_key(v));
            }
        }
        fold_row_max<R>(km, mkey, lane);
    }
    for (int j = tid; j < a.m; j += SCAN_NT) cand[j] = a.it0 == 0 ? j : (int)a.mem_idx[(size_t)b * a.m + j];
    if (tid < 2) { nanflag[tid] = 0; ccount[2 + tid] = 0; }
    const long long n_iter = a.it1 - a.it0;
    // The chunk of iteration it + 1 is fetched into registers during iteration it - 1 .. it and moved into the spare
    // buffers, together with its exponentials under the CURRENT row maxima, while the ranking of iteration it runs
    // (below: "prep").  With more than 128 candidates the ranking occupies waves 0-4 only, so the prefetch registers live
    // in the threads of waves 5-15 (PF0 = 320 and the number of prefetching threads are multiples of R: a prefetching
    // thread's elements belong to its own row r).
    constexpr int PF0 = LCH > 2 ? 320 : 0, PFT = SCAN_NT - PF0;
    const int pt = tid - PF0;                                  // < 0: this thread prefetches nothing
    float pf[PF];
    {
        const long long lo = a.it0 * a.i + a.m;
        const int cnt = n_iter > 0 ? (int)std::min<long long>(a.i, a.n - lo) : 0;
        uint32_t kc = 0u;
#pragma unroll
        for (int k = 0; k < PF; ++k) {                    // first chunk: straight into its rows
            const int e = tid + SCAN_NT * k;
            if (e < cnt * R) {
                const float v = scan_load<PERSIST>(lg + (size_t)lo * R + e);
                xc[(a.m + (e >> log2R)) * ld + r] = v;
                kc = max(kc, max_key(v));
            }
        }
        fold_row_max<R>(kc, ckey, lane);
        for (int j = tid; j < cnt; j += SCAN_NT) cand[a.m + j] = (int)(lo + j);
        const long long lo1 = lo + a.i;
        const int cnt1 = n_iter > 1 ? (int)std::max<long long>(0, std::min<long long>(a.i, a.n - lo1)) : 0;
        if (cnt1 > 0) SCAN_WAIT_ROWS(lo1 + cnt1);
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int e = pt + PFT * k;
            pf[k] = (pt >= 0 && e < cnt1 * R) ? scan_load<PERSIST>(lg + (size_t)lo1 * R + e) : 0.0f;
        }
    }
    int tie = 0;
    uint64_t* const sorted = keyB;
    for (long long it = a.it0; it < a.it1; ++it) {
        const long long lo = it * a.i + a.m;
        const int cnt = (int)std::min<long long>(a.i, a.n - lo);
        const int L = a.m + cnt;
        const int par = (int)((it - a.it0) & 1);
        // P0: nothing to stage - this iteration's chunk rows (and, speculatively, their exponentials) were written by the
        // previous iteration's prep (or by the prologue)
        lds_barrier();
        FAST_STAMP(0);
        // P1: row maxima = the larger of the two key words of the row (memory rows: folded in by the previous iteration's
        // gather; chunk rows: by its prep) - two LDS reads instead of a pass over the rows, a cross-lane and a cross-wave
        // reduction and a barrier.  The words are cleared after the next barrier, when everybody has read them.
        if (tid == 0) {                                        // (all last read several barriers ago)
            ccount[0] = 0; ccount[1] = -1; ccount[2 + (par ^ 1)] = 0;
        }
        const uint32_t mk = max(mkey[r], ckey[r]);
        const uint32_t mbits = as_u32(max_key_value(mk));
        const float rowmax = as_float(mbits);
        // the exponentials of the memory rows are those of the previous iteration while the row's maximum is the same
        const bool changed = it == a.it0 || prevk[par * R + r] != mbits;
        if (tid < R) prevk[(par ^ 1) * R + r] = mbits;
        FAST_STAMP(1);
        // P2: exp(x - max) where it is new: the whole column of every row r whose maximum moved (that includes the chunk
        // rows, whose speculative exponentials were taken under the old maximum; at the first iteration of a launch every
        // column).  A column is L elements: one per thread of the first L threads, all lanes busy - done by the thread
        // that owns the element instead, the lanes of the unchanged rows idle through every exp (7 of 8, on all 16 waves).
        {
            unsigned long long moved = __ballot(changed) & (R == 64 ? ~0ull : ((1ull << R) - 1ull));   // lane r < R holds row r
            while (moved) {
                const int rr = __ffsll((long long)moved) - 1;
                moved &= moved - 1ull;
                exp_column(xc + rr, ec + rr, L, ld, __shfl(rowmax, rr, 64));
            }
        }
        lds_barrier();
        if (tid < 2 * R) pmax[tid] = 0u;                       // the maxima have been read by everybody: clear for the next folds
        FAST_STAMP(2);
        // P3: softmax denominators in the contract's order: lane j adds rows j, j + 64, ... ascending, xor butterfly
        for (int r0 = wave; r0 < R; r0 += 32) {
            const int r1 = r0 + 16;
            const bool has1 = r1 < R;
            float v0[LCH], v1[LCH];
#pragma unroll
            for (int u = 0; u < LCH; ++u) {                    // every read in flight before the first add; slots beyond L
                const int i = lane + 64 * u;                   // add an exact + 0.0
                v0[u] = i < L ? ec[i * ld + r0] : 0.0f;
                v1[u] = (i < L && has1) ? ec[i * ld + r1] : 0.0f;
            }
            float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
            for (int u = 0; u < LCH; ++u) { s0 = s0 + v0[u]; s1 = s1 + v1[u]; }
            wave_sum2(s0, s1, lane);
            if (lane == 0) { rden[r0] = 1.0f / s0; if (has1) rden[r1] = 1.0f / s1; }      // (reciprocals: one division per row)
        }
        lds_barrier();
        FAST_STAMP(3);
        // P4: attention weights e * (1 / den) by every thread, transposed through the spare buffer; then one lane per
        // (candidate, token) adds its H weights in ascending head order and the T lanes of a candidate their tokens
        {
            const float den = rden[r];
            float ev[EPT];
#pragma unroll
            for (int k = 0; k < EPT; ++k) {                      // (all reads in flight before the first product)
                const int l = lrow0 + k * lstep;
                ev[k] = l < L ? ec[l * ld + r] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const int l = lrow0 + k * lstep;
                if (l < L) en[l * ld + r] = ev[k] * den;
            }
        }
        lds_barrier();
        // Keys: the M memory keys go to keyA[0, M); a chunk candidate keeps its key in a register until it is known
        // whether it can still reach the top M (below)
        constexpr int KT = (LCH * 64 * T + SCAN_NT - 1) / SCAN_NT;
        uint64_t mykey[KT];
        uint32_t lowest = 0xFFFFFFFFu;                           // lowest score key among this lane's memory candidates
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int e = kt * SCAN_NT + tid, l = e >> log2T, t = e & (T - 1);
            mykey[kt] = 0ull;
            if (e < a.n2 * T) {                                  // (workgroup-uniform up to the last trip)
                float q = 0.0f;
                if (l < L) {
                    const float* wrow = en + l * ld + t;
                    float wh[H];                                 // all reads in flight together
#pragma unroll
                    for (int hh = 0; hh < H; ++hh) wh[hh] = wrow[hh * T];
                    float sh = 0.0f;
#pragma unroll
                    for (int hh = 0; hh < H; ++hh) sh = sh + wh[hh];
                    q = sh / (float)H;
                }
                float st = q;
                if (T > 1) {
                    st = 0.0f;
#pragma unroll
                    for (int tt = 0; tt < T; ++tt) st = st + __shfl(q, (lane & ~(T - 1)) + tt, 64);
                }
                if (t == 0 && l < L) {
                    mykey[kt] = rank_key(st / (float)T, (uint32_t)l);
                    if (l < a.m) {
                        keyA[l] = mykey[kt];
                        lowest = min(lowest, (uint32_t)(mykey[kt] >> 32));
                    }
                }
            }
        }
        // (64 lanes on one LDS address serialise: the wave reduces first - DPP inside a row of 16, two cross-row steps)
        lowest = min(lowest, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)lowest, 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
        lowest = min(lowest, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)lowest, 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
        lowest = min(lowest, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)lowest, 0x124, 0xF, 0xF, false));    // row_ror:4
        lowest = min(lowest, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)lowest, 0x128, 0xF, 0xF, false));    // row_ror:8
        lowest = min(lowest, lane_xor_u32<16>(lowest, lane));
        lowest = min(lowest, lane_xor_u32<32>(lowest, lane));
        if (lane == 0 && lowest != 0xFFFFFFFFu) atomicMin(reinterpret_cast<unsigned int*>(ccount + 1), lowest);
        lds_barrier();
        // A chunk candidate whose score is below the lowest memory score cannot be among the M best of memory + chunk
        // (the M memory candidates alone beat it), and in a long scan that is almost every one of them: only the others
        // - equal scores included, so exact ties are all still there - are appended behind the memory keys and ranked.
        {
            const uint32_t tau = (uint32_t)ccount[1];             // lowest memory score key (ds_min_u32 above)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int e = kt * SCAN_NT + tid, l = e >> log2T;
                const bool in = mykey[kt] != 0ull && l >= a.m && (uint32_t)(mykey[kt] >> 32) >= tau;
                const unsigned long long mask = __ballot(in);
                if (mask != 0ull) {                              // wave-uniform
                    int base = 0;
                    if (lane == 0) base = atomicAdd(ccount, __popcll(mask));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (in) keyA[a.m + base + __popcll(mask & ((1ull << lane) - 1ull))] = mykey[kt];
                }
            }
        }
        lds_barrier();
        const int Lr = a.m + ccount[0];                          // candidates that take part in the ranking
        // prep of iteration it + 1 (fills the issue slots the ranking leaves idle): its chunk into rows m.. of the SPARE
        // buffers (they become the current ones at the end of this iteration; the weights that lived in `en` are dead, the
        // ranking's scratch sits in its first rows only), with exponentials under this iteration's maxima - right when
        // the maxima do not move, which the next iteration checks bitwise; then the loads of the chunk after that
        {
            const long long lo1 = lo + a.i;
            const int cnt1 = it + 1 < a.it1 ? (int)std::max<long long>(0, std::min<long long>(a.i, a.n - lo1)) : 0;
            uint32_t kc = 0u;
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int e = pt + PFT * k;
                if (pt >= 0 && e < cnt1 * R) {
                    const int row = a.m + (e >> log2R);
                    xn[row * ld + r] = pf[k];
                    en[row * ld + r] = det_expf_np(pf[k] - rowmax);
                    kc = max(kc, max_key(pf[k]));
                }
            }
            if (pt >= 0) fold_row_max<R>(kc, ckey, lane);          // (whole waves: PF0 is a multiple of 64)
            for (int j = tid; j < cnt1; j += SCAN_NT) cnew[a.m + j] = (int)(lo1 + j);
        }
        if (STAMP && tid == 0) tacc[7] += (unsigned long long)(Lr - a.m);
        if (STAMP && PERSIST && tid == 0 && b == 0 && it < 512) stamps[8 * gridDim.x + 4 * it] = __builtin_amdgcn_s_memtime() - tlast;
        FAST_STAMP(4);
        if (Lr <= 192) {                     // counting rank below the crossover of the two rankings (~200 keys)
            int P = 1;
            while (P < 64 && 2 * P * Lr <= SCAN_NT) P <<= 1;
            rank_scatter(keyA, keyB, Lr, P);
        } else {
            rank_runs4(keyA, keyB, reinterpret_cast<uint64_t*>(en), Lr);
        }
        lds_barrier();
        // exact ties among the first M + 1 ranked scores that call for torch.topk's order?  The loop's rule (oracle
        // orc_topm_loop, round 5): two NEIGHBOURS of equal score whose logit rows are bit-identical (tie_order 2: any two of
        // equal score).  One pair per thread, any hit raises the flag.
        if (a.tie_order != 0) {
            const int npair = a.m < Lr - 1 ? a.m : Lr - 1;
            bool hit = false;
            for (int j = tid; j < npair; j += SCAN_NT) {
                if ((sorted[j] >> 32) != (sorted[j + 1] >> 32)) continue;
                bool same = true;
                if (a.tie_order == 1) {
                    const float* ra = xc + key_pos(sorted[j]) * ld;
                    const float* rb = xc + key_pos(sorted[j + 1]) * ld;
                    for (int rr = 0; rr < R; ++rr) same = same && as_u32(ra[rr]) == as_u32(rb[rr]);
                }
                hit = hit || same;
            }
            if (__ballot(hit) != 0ull && lane == 0) ccount[2 + par] = 1;
        }
        lds_barrier();
        bool boundary_tie = Lr > a.m && (sorted[a.m - 1] >> 32) == (sorted[a.m] >> 32);   // bit-equal score keys (the oracle's rule: two NaNs tie)
        if (a.tie_order != 0 && ccount[2 + par] != 0) {
            // torch.topk's order under ties depends on the WHOLE candidate array, so every chunk key goes back to its
            // place, all L candidates are ranked and the replay runs on them (rare)
            const unsigned long long ts0 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int l = (kt * SCAN_NT + tid) >> log2T;
                if (mykey[kt] != 0ull && l >= a.m) keyA[l] = mykey[kt];
            }
            lds_barrier();
            if (L <= 192) {
                int P = 1;
                while (P < 64 && 2 * P * L <= SCAN_NT) P <<= 1;
                rank_scatter(keyA, keyB, L, P);
            } else {
                rank_runs(keyA, keyB, reinterpret_cast<uint64_t*>(en), L);
            }
            lds_barrier();
            boundary_tie = L > a.m && (sorted[a.m - 1] >> 32) == (sorted[a.m] >> 32);
            const unsigned long long ts1 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
            tie_order_slow(keyB, keyA, L, a.m, reinterpret_cast<int*>(smem + a.stk_off));
            if (STAMP && PERSIST && tid == 0 && b == 0) {
                stamps[8 * gridDim.x + 2044] = ts1 - ts0;
                stamps[8 * gridDim.x + 2045] = __builtin_amdgcn_s_memtime() - ts1;
            }
        }
        if (STAMP && PERSIST && tid == 0 && b == 0 && it < 512) { stamps[8 * gridDim.x + 4 * it + 1] = __builtin_amdgcn_s_memtime() - tlast; stamps[8 * gridDim.x + 4 * it + 2] = __builtin_amdgcn_s_memrealtime(); }
        FAST_STAMP(5);
        // the loads of the chunk after the next one, issued HERE - behind the ranking, not in the prep in front of it: the
        // ranking's searches reload a spilled register pair, the compiler therefore drains the vector-memory counter in
        // front of them, and loads issued before that point were waited for on the spot (1.2 k cycles per iteration by
        // every wave, in-kernel stamps); from here they fly through the gather and the first phases of the next
        // iteration, which touch the LDS only
        {
            const long long lo2 = lo + 2 * a.i;
            const int cnt2 = it + 2 < a.it1 ? (int)std::max<long long>(0, std::min<long long>(a.i, a.n - lo2)) : 0;
            if (cnt2 > 0) SCAN_WAIT_ROWS(lo2 + cnt2);
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int e = pt + PFT * k;
                pf[k] = (pt >= 0 && e < cnt2 * R) ? scan_load<PERSIST>(lg + (size_t)lo2 * R + e) : 0.0f;
            }
        }
        // P6: new memory: indices, logit rows and exponentials of the winners, into the other buffers
        for (int j = tid; j < a.m; j += SCAN_NT) cnew[j] = cand[key_pos(sorted[j])];
        {
            int src[EPT];
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const int j = lrow0 + k * lstep;
                src[k] = j < a.m ? (int)key_pos(sorted[j]) * ld + r : 0;
            }
            float gx[EPT], ge[EPT];
#pragma unroll
            for (int k = 0; k < EPT; ++k) { gx[k] = xc[src[k]]; ge[k] = ec[src[k]]; }
            uint32_t km = 0u;
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const int j = lrow0 + k * lstep;
                if (j < a.m) { xn[j * ld + r] = gx[k]; en[j * ld + r] = ge[k]; km = max(km, max_key(gx[k])); }
            }
            fold_row_max<R>(km, mkey, lane);                       // maxima of the NEW memory rows, for the next iteration
        }
        if (tid == 0 && boundary_tie) tie = 1;
        { int* t = cand; cand = cnew; cnew = t; }
        { float* t = xc; xc = xn; xn = t; }
        { float* t = ec; ec = en; en = t; }
        FAST_STAMP(6);
        // no barrier here: the next iteration's first phase writes rows m.. of the new buffers only, and its barrier
        // orders everything before the maxima are read
    }
    lds_barrier();
    for (int j = tid; j < a.m; j += SCAN_NT) {
        a.mem_idx[(size_t)b * a.m + j] = cand[j];
        if (a.mem_score) a.mem_score[(size_t)b * a.m + j] = n_iter > 0 ? key_score(sorted[j]) : 0.0f;
    }
    if (a.tie && tid == 0 && tie) a.tie[b] = 1;
    if (STAMP && tid == 0)
        for (int k = 0; k < 8; ++k) stamps[(size_t)b * 8 + k] = tacc[k];
}

// ---------------------------------------------------------------------------------------------------------------
// scan_cam_kernel (round 4): the LDS-resident loop SPECIALISED for BASELINE configs[3] - 8 logits per candidate (8 heads, one
// token), M = I = 256: 512 candidates.  The same arithmetic as scan_fast_kernel, every sum in the contract's order:
// bit-identical indices, scores and tie flags (tools/scan_compare.py holds the two against each other and against the
// generic kernel).
//
// What shapes it (rocprofv3 counters of the loop alone and every wave's clock at every barrier, tools/scan_stamps.py
// camwaves / tools/pmc_scan.sh; profiles/r04_scan_*.txt): on its ONE compute unit scan_fast_kernel is bound by instruction
// ISSUE - 11.1 k wave-instructions (6.6 k vector, 3.7 k scalar, 0.8 k LDS) in the 11.1 k cycles of an iteration, i.e. one
// instruction per SIMD every four cycles whatever its kind; with half the waves (8) the same work is 8.3 k instructions in
// 11.6 k cycles - then a wave's own dependent-issue and LDS latency binds.  So the loop keeps 16 waves and sheds
// INSTRUCTIONS:
//   * every size is a compile-time constant: LDS addresses are immediates, no loop or address arithmetic on runtime M / I,
//     no scalar registers spilled to vector lanes (scan_fast_kernel: 135 spills, ~500 v_readlane / v_writelane);
//   * one THREAD per candidate on waves 0..7: its 8 exponentials with two 16-byte reads (rows of 12 words: conflict-free),
//     8 divisions, the head sum and the key - no transposition of the weights through LDS; wave w also sums row w
//     (contract order) right in front; waves 8..15 prepare the next chunk (logits, speculative exponentials, maxima);
//   * ranking on the 32-bit SCORE keys: without exact ties the scores alone order the candidates; a tie anywhere among the
//     ranked candidates sends the iteration through the 64-bit ranking and the replay of torch.topk's order (a superset of
//     "ties among the first M + 1 ranks": the replay reproduces torch's result either way);
//   * a memory wave sorts its 64 scores WITHOUT payload - v_med3_u32 against all-ones / zero picks max or min: 3
//     instructions and no lane-mask registers per stage - into a run that is only a search structure;
//   * chunk candidates at or above the lowest memory score (typically 4-10 of 256) are compacted into an unsorted list S;
//   * every (memory key, run) pair is ONE 4-ary search of 11 reads on its own thread (4 x 256 = all 1,024 threads);
//     survivors' pairs are a second pass on the first waves; counts against S ride on the run-0 threads; the partial
//     counts meet in LDS, the key's owner adds them and places its 64-bit key.  More than CAM_SMAX survivors (the first
//     iterations of a scan): scan_fast_kernel's ranking of all keys;
//   * the gather moves the exponentials 16 bytes at a time (waves 8..15) beside the logits (waves 0..7, which fold the new
//     memory's row maxima).
namespace cam {
constexpr int M = 256, I = 256, L = 512, R = 8, H = 8, LD = 12, NT = 1024, SMAX = 32, PRW = M + SMAX;
constexpr int OFF_SORTED = 0;                        // u64[L]: the ranked keys
constexpr int OFF_KEYA = OFF_SORTED + L * 8;         // u64[L]: survivors' keys at [M..], every key by position on the tie path
constexpr int OFF_CAND = OFF_KEYA + L * 8;           // int[2][L]: patch index of every candidate (two sets)
constexpr int OFF_PMAX = OFF_CAND + 2 * L * 4;       // u32[2][2 R]: row-maximum keys, [memory R | chunk R], by parity
constexpr int OFF_CNT = OFF_PMAX + 4 * R * 4;        // int[8]: see ccount below
constexpr int OFF_DEN = OFF_CNT + 32;                // float[R], 16-byte aligned
constexpr int OFF_PREV = OFF_DEN + R * 4;            // u32[2][R]: bits of the previous row maxima, by parity
constexpr int OFF_RUNS = OFF_PREV + 2 * R * 4;       // u32[4][64]: the memory waves' sorted scores
constexpr int OFF_SC = OFF_RUNS + M * 4;             // u32[M]: score key of memory candidate l
constexpr int OFF_PR = OFF_SC + M * 4;               // int[4][PRW]: partial counts
constexpr int OFF_X = (OFF_PR + 4 * PRW * 4 + 15) & ~15;     // float[2][L][LD]: logits
constexpr int OFF_E = OFF_X + 2 * L * LD * 4;        // float[2][L][LD]: exponentials
constexpr int OFF_STK = OFF_E + 2 * L * LD * 4;      // scratch of the tie replay
constexpr int LDS_BYTES = OFF_STK + STK_BYTES;
static_assert(OFF_DEN % 16 == 0 && OFF_X % 16 == 0 && (LD * 4) % 16 == 0, "16-byte rows");
static_assert(M * 4 + M * 4 + 4 * PRW * 4 >= L * 8, "the ranking scratch doubles as rank_runs' run buffer");
}  // namespace cam

// descending bitonic sort of one 32-bit key per lane, the direction of every stage from one bit of `dir` (bit n set: this
// lane keeps the LARGER key in stage n): median(s, partner, all-ones | 0) = max | min.  Duplicates are kept.
template <int N, int J>
__device__ __forceinline__ uint32_t cmpx32(uint32_t s, uint32_t dir, int lane) {
    const uint32_t o = lane_xor_u32<J>(s, lane);
    const uint32_t c = (uint32_t)__builtin_amdgcn_sbfe((int)dir, N, 1);       // v_bfe_i32: 0 or 0xFFFFFFFF
    uint32_t d;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(s), "v"(o), "v"(c));
    return d;
}

// bit n of the result: lane keeps the larger key in stage n of wave_sort_desc_u32 (stage (K, J): ((lane & K) == 0) == ((lane & J) == 0))
__device__ __forceinline__ uint32_t sort_directions(int lane) {
    constexpr int KJ[21][2] = {{2, 1}, {4, 2}, {4, 1}, {8, 4}, {8, 2}, {8, 1}, {16, 8}, {16, 4}, {16, 2}, {16, 1}, {32, 16}, {32, 8},
                               {32, 4}, {32, 2}, {32, 1}, {64, 32}, {64, 16}, {64, 8}, {64, 4}, {64, 2}, {64, 1}};
    uint32_t d = 0u;
#pragma unroll
    for (int n = 0; n < 21; ++n) d |= ((((lane & KJ[n][0]) == 0) == ((lane & KJ[n][1]) == 0)) ? 1u : 0u) << n;
    return d;
}

__device__ __forceinline__ uint32_t wave_sort_desc_u32(uint32_t s, uint32_t dir, int lane) {
    s = cmpx32<0, 1>(s, dir, lane);
    s = cmpx32<1, 2>(s, dir, lane); s = cmpx32<2, 1>(s, dir, lane);
    s = cmpx32<3, 4>(s, dir, lane); s = cmpx32<4, 2>(s, dir, lane); s = cmpx32<5, 1>(s, dir, lane);
    s = cmpx32<6, 8>(s, dir, lane); s = cmpx32<7, 4>(s, dir, lane); s = cmpx32<8, 2>(s, dir, lane); s = cmpx32<9, 1>(s, dir, lane);
    s = cmpx32<10, 16>(s, dir, lane); s = cmpx32<11, 8>(s, dir, lane); s = cmpx32<12, 4>(s, dir, lane); s = cmpx32<13, 2>(s, dir, lane);
    s = cmpx32<14, 1>(s, dir, lane);
    s = cmpx32<15, 32>(s, dir, lane); s = cmpx32<16, 16>(s, dir, lane); s = cmpx32<17, 8>(s, dir, lane); s = cmpx32<18, 4>(s, dir, lane);
    s = cmpx32<19, 2>(s, dir, lane); s = cmpx32<20, 1>(s, dir, lane);
    return s;
}

// Score keys of this loop are images of non-negative floats or NaN (means of softmax weights): bit 31 is always set, two
// keys differ by less than 2^31, so "p > m" is the sign bit of m - p - comparisons without the condition-code register
// (on gfx950 a VALU write of VCC costs the next VALU reader two wait states) and without selects.
__device__ __forceinline__ uint32_t key_gt(uint32_t p, uint32_t m) { return (m - p) >> 31; }

// number of keys of a descending run of 64 that are larger than m (4-ary search: 3 + 3 + 3 + 2 reads).  `eq` collects, as a
// running minimum of xors, whether a key EQUAL to m sits at the insertion point (0 = yes): an exact tie when the run is not
// the key's own; `own` (all-ones for the key's own run, else 0) masks that test.
__device__ __forceinline__ int search_run_u32(const uint32_t* run, uint32_t m, uint32_t own, uint32_t& eq) {
    uint32_t lo = 0;
#pragma unroll
    for (int step = 16; step >= 1; step >>= 2) {
        const uint32_t p1 = run[lo + step - 1], p2 = run[lo + 2 * step - 1], p3 = run[lo + 3 * step - 1];
        lo += (key_gt(p1, m) + key_gt(p2, m) + key_gt(p3, m)) * step;
    }
    const uint32_t last = run[lo], nxt = run[lo < 63 ? lo + 1 : 63];      // lo <= 63
    // an equal key in the run is `last` (then it is not larger) or the one behind a larger `last`
    eq = min(eq, min((last ^ m) | own, (nxt ^ m) | own));
    return (int)(lo + key_gt(last, m));
}

// two searches at once, their dependent reads interleaved: (run A, key mA, own-run mask) and (run B, key mB) - what the first
// 128 threads do in the ranking (a memory key's pair AND a survivor's pair: done one after the other the second search's 11
// dependent LDS round trips kept waves 0 and 1 at the barrier 1,000-1,500 cycles after everybody else, round 5)
__device__ __forceinline__ void search_run_u32_x2(const uint32_t* runA, uint32_t mA, uint32_t ownA, const uint32_t* runB, uint32_t mB,
                                                  uint32_t& eqA, uint32_t& eqB, int& cA, int& cB) {
    uint32_t loA = 0, loB = 0;
#pragma unroll
    for (int step = 16; step >= 1; step >>= 2) {
        const uint32_t a1 = runA[loA + step - 1], a2 = runA[loA + 2 * step - 1], a3 = runA[loA + 3 * step - 1];
        const uint32_t b1 = runB[loB + step - 1], b2 = runB[loB + 2 * step - 1], b3 = runB[loB + 3 * step - 1];
        loA += (key_gt(a1, mA) + key_gt(a2, mA) + key_gt(a3, mA)) * step;
        loB += (key_gt(b1, mB) + key_gt(b2, mB) + key_gt(b3, mB)) * step;
    }
    const uint32_t lastA = runA[loA], nxtA = runA[loA < 63 ? loA + 1 : 63];
    const uint32_t lastB = runB[loB], nxtB = runB[loB < 63 ? loB + 1 : 63];
    eqA = min(eqA, min((lastA ^ mA) | ownA, (nxtA ^ mA) | ownA));
    eqB = min(eqB, min(lastB ^ mB, nxtB ^ mB));
    cA = (int)(loA + key_gt(lastA, mA));
    cB = (int)(loB + key_gt(lastB, mB));
}

template <bool STAMP, bool PERSIST>
__global__ __launch_bounds__(cam::NT) void scan_cam_kernel(ScanArgs a, unsigned long long* stamps) {
    using namespace cam;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (!PERSIST && scan_skipped(a.cond, a.cond_mask)) return;
    __builtin_amdgcn_s_setprio(3);
    if (PERSIST) {
        asm volatile("v_mov_b32 v127, 0" ::: "v127");               // (the whole register file of the compute unit: see scan_fast_kernel)
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_or(a.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&g_persist_log[4], (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    unsigned long long tacc[8], tlast = 0;
    uint64_t* const sorted = reinterpret_cast<uint64_t*>(smem + OFF_SORTED);
    uint64_t* const keyA = reinterpret_cast<uint64_t*>(smem + OFF_KEYA);
    uint32_t* const pmax = reinterpret_cast<uint32_t*>(smem + OFF_PMAX);
    int* const ccount = reinterpret_cast<int*>(smem + OFF_CNT);   // [0] survivors, [1] lowest memory score key, [2], [3] tie flag (by parity), [6] rows known, [7] replay
    float* const rden = reinterpret_cast<float*>(smem + OFF_DEN);
    uint32_t* const prevk = reinterpret_cast<uint32_t*>(smem + OFF_PREV);
    uint32_t* const runs32 = reinterpret_cast<uint32_t*>(smem + OFF_RUNS);
    uint32_t* const sc32 = reinterpret_cast<uint32_t*>(smem + OFF_SC);
    int* const pr = reinterpret_cast<int*>(smem + OFF_PR);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = tid & (R - 1);
    const int hid = tid - 512;                                  // helper index (waves 8..15), < 0 on the candidate waves
    const uint32_t dir = sort_directions(lane);
    // Fewer workgroups than slides (ipsx_scan_persistent_on): this one takes the slides blockIdx.x, + gridDim.x, ... one
    // after the other - the producer works through the slides in that order, and a loop is faster than its slide's
    // projector, so two resident loops follow 16 slides and the projector keeps the other 14 compute units.
    for (int b = blockIdx.x; b < a.slides; b += (int)gridDim.x) {
    const float* lg = a.lg + (size_t)b * a.n * R;
#pragma unroll
    for (int k = 0; k < 8; ++k) tacc[k] = 0;
    if (STAMP) tlast = __builtin_amdgcn_s_memtime();
    int ready_known = 0;
    if (tid < 4 * R) pmax[tid] = 0u;
    lds_barrier();
    SCAN_WAIT_ROWS(std::min<long long>(a.n, a.it0 * (long long)I + M + I));
    {
        float* const x0 = reinterpret_cast<float*>(smem + OFF_X);
        int* const cand0 = reinterpret_cast<int*>(smem + OFF_CAND);
        uint32_t km = 0u;
#pragma unroll
        for (int k = 0; k < 2; ++k) {                            // memory rows: 2,048 logits, two per thread
            const int l = (tid >> 3) + 128 * k;
            const size_t row = a.it0 == 0 ? (size_t)l : (size_t)a.mem_idx[(size_t)b * M + l];
            const float v = scan_load<PERSIST>(lg + row * R + r);
            x0[l * LD + r] = v;
            km = max(km, max_key(v));
        }
        fold_row_max<R>(km, pmax, lane);                        // set 0: read by the first iteration
        if (tid < M) cand0[tid] = a.it0 == 0 ? tid : (int)a.mem_idx[(size_t)b * M + tid];
    }
    if (tid < 2) ccount[2 + tid] = 0;
    const long long n_iter = a.it1 - a.it0;
    float pf[4];
    {
        float* const x0 = reinterpret_cast<float*>(smem + OFF_X);
        int* const cand0 = reinterpret_cast<int*>(smem + OFF_CAND);
        const long long lo = a.it0 * (long long)I + M;
        const int cnt = n_iter > 0 ? (int)std::min<long long>(I, a.n - lo) : 0;
        uint32_t kc = 0u;
#pragma unroll
        for (int k = 0; k < 2; ++k) {                            // first chunk: straight into its rows
            const int e = tid + NT * k;
            if (e < cnt * R) {
                const float v = scan_load<PERSIST>(lg + (size_t)lo * R + e);
                x0[(M + (e >> 3)) * LD + r] = v;
                kc = max(kc, max_key(v));
            }
        }
        fold_row_max<R>(kc, pmax + R, lane);
        if (tid < cnt) cand0[M + tid] = (int)(lo + tid);
        const long long lo1 = lo + I;
        const int cnt1 = n_iter > 1 ? (int)std::max<long long>(0, std::min<long long>(I, a.n - lo1)) : 0;
        if (cnt1 > 0) SCAN_WAIT_ROWS(lo1 + cnt1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = hid + 512 * k;
            pf[k] = (hid >= 0 && e < cnt1 * R) ? scan_load<PERSIST>(lg + (size_t)lo1 * R + e) : 0.0f;
        }
    }
    int tie = 0;
    const int n_it = (int)n_iter, n_rows = (int)a.n;             // (n < 2^31: 32-bit row arithmetic inside the loop)
    for (int k_it = 0; k_it < n_it; ++k_it) {
        const int lo = ((int)a.it0 + k_it) * I + M;
        const int cnt = min(I, n_rows - lo);
        const int Lc = M + cnt;                                  // candidates of this iteration (512 but for a ragged last chunk)
        const int par = k_it & 1;
        // current / spare buffers by parity; everything else sits at a fixed address
        float* const xc = reinterpret_cast<float*>(smem + OFF_X) + par * (L * LD);
        float* const xn = reinterpret_cast<float*>(smem + OFF_X) + (par ^ 1) * (L * LD);
        float* const ec = reinterpret_cast<float*>(smem + OFF_E) + par * (L * LD);
        float* const en = reinterpret_cast<float*>(smem + OFF_E) + (par ^ 1) * (L * LD);
        int* const cand = reinterpret_cast<int*>(smem + OFF_CAND) + par * L;
        int* const cnew = reinterpret_cast<int*>(smem + OFF_CAND) + (par ^ 1) * L;
        uint32_t* const mkey = pmax + 2 * R * par;              // row-maximum keys read by this iteration: [memory | chunk]
        uint32_t* const mkey_nx = pmax + 2 * R * (par ^ 1);     // ... and folded into by this iteration, for the next one
        // diagnostic (STAMP build): every wave's clock at 15 points of iterations 100..103 of image 0 (tools/scan_stamps.py camwaves)
        unsigned long long* const wlog = (STAMP && b == 0 && k_it >= 100 && k_it < 104)
                                             ? stamps + 8 * gridDim.x + 2048 + (k_it - 100) * 256 + wave * 16 : nullptr;
#define WSTAMP(k_) do { if (STAMP && wlog != nullptr && lane == 0) wlog[k_] = __builtin_amdgcn_s_memtime(); } while (0)
        const int lo1 = lo + I;
        const int cnt1 = k_it + 1 < n_it ? max(0, min(I, n_rows - lo1)) : 0;
        uint32_t kc = 0u;
        // one element of the next chunk into the SPARE buffers: its logit, its exponential under this iteration's maxima
        // (right unless a maximum moves - checked bitwise by the next iteration), its share of the chunk rows' maxima
#define CAM_PREP(k_)                                                                        \
        do {                                                                                \
            const int row_ = (hid >> 3) + 64 * (k_);                                        \
            if (row_ < cnt1) {                                                              \
                xn[(M + row_) * LD + r] = pf[k_];                                           \
                en[(M + row_) * LD + r] = det_expf_np(pf[k_] - rowmax);                     \
                kc = max(kc, max_key(pf[k_]));                                              \
            }                                                                               \
        } while (0)
        WSTAMP(0);
        lds_barrier();                                          // B0
        WSTAMP(1);
        FAST_STAMP(0);
        // P1 (helper waves - the candidate waves need no maxima): row maxima from the two key words of the row
        if (tid == 0) { ccount[0] = 0; ccount[1] = -1; ccount[2 + (par ^ 1)] = 0; ccount[7] = 0; }
        float rowmax = 0.0f;
        if (wave >= 8) {
            const uint32_t mk = max(mkey[r], mkey[R + r]);
            const uint32_t mbits = as_u32(max_key_value(mk));
            rowmax = as_float(mbits);
            const bool changed = k_it == 0 || prevk[par * R + r] != mbits;
            if (hid < R) prevk[(par ^ 1) * R + r] = mbits;
            // P2: exp(x - max) of every row whose maximum moved (a column of Lc elements, one per helper thread)
            unsigned long long moved = __ballot(changed) & ((1ull << R) - 1ull);
            while (moved) {
                const int rr = __ffsll((long long)moved) - 1;
                moved &= moved - 1ull;
                exp_column_part(xc + rr, ec + rr, Lc, LD, __shfl(rowmax, rr, 64), hid, 512);
            }
        }
        FAST_STAMP(1);
        WSTAMP(2);
        lds_barrier();                                          // B1
        WSTAMP(3);
        if (tid < 2 * R) mkey[tid] = 0u;                        // read by everybody: cleared for the folds of the NEXT iteration
        FAST_STAMP(2);
        // P3: the candidate waves fetch their exponentials and sum one row each (contract order: lane j adds candidates
        // j, j + 64, ... ascending, then the xor butterfly); the helper waves start on the next chunk
        const bool is_cand = tid < Lc;
        float4 ev0, ev1;
        if (wave < 8) {
            const float* const row = ec + (is_cand ? tid : 0) * LD;
            ev0 = *reinterpret_cast<const float4*>(row);
            ev1 = *reinterpret_cast<const float4*>(row + 4);
            float v0[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v0[u] = ec[(lane + 64 * u) * LD + wave];          // (rows beyond Lc: stale, masked below)
            float s0 = 0.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u) s0 = s0 + ((lane + 64 * u < Lc) ? v0[u] : 0.0f);
            s0 = wave_butterfly_sum(s0);
            if (lane == 0) rden[wave] = 1.0f / s0;              // (the reciprocal: one division per row, eight products per candidate)
        } else {
            CAM_PREP(0);
            CAM_PREP(1);
        }
        WSTAMP(4);
        lds_barrier();                                          // B2
        WSTAMP(5);
        FAST_STAMP(3);
        uint64_t key = 0ull;
        if (wave < 8) {
            // weights e * (1 / den), heads added in ascending order, mean over the 8 heads (one token: the mean over tokens is the
            // identity) - the operations of scan_fast_kernel's weight and score phases on this candidate
            const float4 d0 = *reinterpret_cast<const float4*>(rden), d1 = *reinterpret_cast<const float4*>(rden + 4);
            const float w0 = ev0.x * d0.x, w1 = ev0.y * d0.y, w2 = ev0.z * d0.z, w3 = ev0.w * d0.w;
            const float w4 = ev1.x * d1.x, w5 = ev1.y * d1.y, w6 = ev1.z * d1.z, w7 = ev1.w * d1.w;
            float sh = 0.0f;
            sh = sh + w0; sh = sh + w1; sh = sh + w2; sh = sh + w3; sh = sh + w4; sh = sh + w5; sh = sh + w6; sh = sh + w7;
            const float q = sh / (float)H;
            if (is_cand) key = rank_key(q / 1.0f, (uint32_t)tid);
            if (wave < 4) {                                      // lowest memory score of this wave -> the threshold
                uint32_t lowest = (uint32_t)(key >> 32);
                sc32[tid] = lowest;
                lowest = min(lowest, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)lowest, 0xB1, 0xF, 0xF, false));
                lowest = min(lowest, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)lowest, 0x4E, 0xF, 0xF, false));
                lowest = min(lowest, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)lowest, 0x124, 0xF, 0xF, false));
                lowest = min(lowest, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)lowest, 0x128, 0xF, 0xF, false));
                lowest = min(lowest, lane_xor_u32<16>(lowest, lane));
                lowest = min(lowest, lane_xor_u32<32>(lowest, lane));
                if (lane == 0) atomicMin(reinterpret_cast<unsigned int*>(ccount + 1), lowest);
            }
        } else {
            CAM_PREP(2);
            CAM_PREP(3);
            fold_row_max<R>(kc, mkey_nx + R, lane);
            if (hid < cnt1) cnew[M + hid] = (int)(lo1 + hid);
        }
        WSTAMP(6);
        lds_barrier();                                          // B4: the threshold is known
        WSTAMP(7);
        if (wave < 4) {
            // this wave's memory scores as one sorted run (a search structure: no payload); equal neighbours = an exact tie
            const uint32_t s = wave_sort_desc_u32((uint32_t)(key >> 32), dir, lane);
            runs32[tid] = s;
            const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp((int)~s, (int)s, 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane 0: ~s)
            if (__ballot(up == s) != 0ull && lane == 0) ccount[2 + par] = 1;
        } else if (wave < 8) {
            const uint32_t tau = (uint32_t)ccount[1];
            const bool in = is_cand && (uint32_t)(key >> 32) >= tau;
            const unsigned long long mask = __ballot(in);
            if (mask != 0ull) {                                  // wave-uniform
                int base = 0;
                if (lane == 0) base = atomicAdd(ccount, __popcll(mask));
                base = __builtin_amdgcn_readfirstlane(base);
                // (v_mbcnt: survivors on the lanes below this one - no per-lane mask constant, which the compiler kept in two
                //  registers across the loop and SPILLED: a scratch reload in the hot loop, round 5)
                if (in) keyA[M + base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = key;
            }
        }
        WSTAMP(8);
        lds_barrier();                                          // B5: runs and survivors are in place
        WSTAMP(9);
        const int ks = ccount[0];
        const int Lr = M + ks;
        if (STAMP && tid == 0) tacc[7] += (unsigned long long)ks;
        FAST_STAMP(4);
        if (ks <= SMAX) {
            const uint32_t* const skeys = reinterpret_cast<const uint32_t*>(keyA + M);       // survivor i: words 2 i (position), 2 i + 1 (score)
            uint32_t eq = 0xFFFFFFFFu;                           // becomes 0 when this thread sees two equal scores
            {   // pass A: memory key kk against run rb (wave-uniform: is it the key's own run?); pass B, on the first 4 SMAX
                // threads (waves 0 and 1, wave-uniform): survivor i against run rb2 - the same threads' two searches run
                // interleaved (search_run_u32_x2)
                const int kk = tid & (M - 1), rb = tid >> 8;
                const uint32_t m = sc32[kk];
                const uint32_t own = (wave & 3) == rb ? 0xFFFFFFFFu : 0u;
                int c, c2 = 0;
                const int rb2 = tid >> 5, i2 = tid & (SMAX - 1);
                const bool passb = ks > 0 && wave < 2;               // (tid < 4 SMAX = 128)
                uint32_t m2 = 0u, eq2 = 0xFFFFFFFFu;
                if (passb) {
                    m2 = skeys[2 * (i2 < ks ? i2 : 0) + 1];
                    search_run_u32_x2(runs32 + 64 * rb, m, own, runs32 + 64 * rb2, m2, eq, eq2, c, c2);
                } else {
                    c = search_run_u32(runs32 + 64 * rb, m, own, eq);
                }
                // the counts against the (unsorted) survivors are dealt out over the four partial counts of a key - survivors
                // rb, rb + 4, ... ride on the threads of run rb (it was all of them on run 0's: waves 0-3 late at the barrier)
                for (int i = rb; i < ks; i += 4) {               // (wave-uniform trip count; broadcast reads)
                    const uint32_t sv = skeys[2 * i + 1];
                    c += (int)key_gt(sv, m);
                    eq = min(eq, sv ^ m);
                }
                pr[rb * PRW + kk] = c;
                if (passb && i2 < ks) {
                    for (int j = rb2; j < ks; j += 4) {
                        const uint32_t sv = skeys[2 * j + 1];
                        c2 += (int)key_gt(sv, m2);
                        eq2 = min(eq2, (sv ^ m2) | (j == i2 ? 0xFFFFFFFFu : 0u));
                    }
                    pr[rb2 * PRW + M + i2] = c2;
                    eq = min(eq, eq2);
                }
            }
            if (__ballot(eq == 0u) != 0ull && lane == 0) ccount[2 + par] = 1;
            WSTAMP(10);
            lds_barrier();                                      // B6: the partial counts are in place
            if (ccount[2 + par] == 0) {                          // (with a tie the 64-bit ranking below replaces all of this)
                if (tid < M) {
                    sorted[pr[tid] + pr[PRW + tid] + pr[2 * PRW + tid] + pr[3 * PRW + tid]] = key;
                } else if (wave == 8 && lane < ks) {
                    // (addresses from a copy of the lane index the compiler cannot see through: hoisted out of the loop they
                    //  were four registers it spilled - four scratch reloads per iteration, in front of a barrier everybody
                    //  waits at, that went to HBM whenever a producer beside the loop streamed through the L2; round 5)
                    int ll = lane;
                    asm volatile("" : "+v"(ll));
                    const int* const prs = pr + M + ll;
                    sorted[prs[0] + prs[PRW] + prs[2 * PRW] + prs[3 * PRW]] = keyA[M + ll];
                }
            }
            WSTAMP(11);
            lds_barrier();                                      // B7
        } else {
            // many survivors (the first iterations of a scan): the ranking of scan_fast_kernel over memory keys + survivors
            if (wave < 4) keyA[tid] = key;
            lds_barrier();
            rank_runs4(keyA, sorted, reinterpret_cast<uint64_t*>(smem + OFF_RUNS), Lr);
            lds_barrier();
            {
                const int npair = M < Lr - 1 ? M : Lr - 1;
                bool hit = false;
                for (int j = tid; j < npair; j += NT) hit = hit || (sorted[j] >> 32) == (sorted[j + 1] >> 32);
                if (__ballot(hit) != 0ull && lane == 0) ccount[2 + par] = 1;
            }
            lds_barrier();
        }
        const bool tied = ccount[2 + par] != 0;
        if (tied) {
            // an exact tie somewhere among the ranked candidates: the 64-bit ranking (score, then earlier position) of ALL
            // candidates and, for the reference's order, the replay of torch.topk on them (its order depends on the whole array)
            if (is_cand) keyA[tid] = key;
            lds_barrier();
            rank_runs(keyA, sorted, reinterpret_cast<uint64_t*>(smem + OFF_RUNS), Lc);
            lds_barrier();
            if (a.tie_order != 0) {
                // (ties among the first M + 1 ranks only: without one torch.topk's result is the canonical order - and, the
                //  loop's rule, oracle orc_topm_loop: only between NEIGHBOURS whose logit rows are bit-identical; two
                //  different rows whose scores collide in the last bit keep the canonical order.  tie_order 2: any tie)
                const int npair = M < Lc - 1 ? M : Lc - 1;
                bool hit = false;
                for (int j = tid; j < npair; j += NT) {
                    if ((sorted[j] >> 32) != (sorted[j + 1] >> 32)) continue;
                    bool same = true;
                    if (a.tie_order == 1) {
                        const uint4* ra = reinterpret_cast<const uint4*>(xc + key_pos(sorted[j]) * LD);
                        const uint4* rb = reinterpret_cast<const uint4*>(xc + key_pos(sorted[j + 1]) * LD);
                        const uint4 a0 = ra[0], a1 = ra[1], b0 = rb[0], b1 = rb[1];
                        same = a0.x == b0.x && a0.y == b0.y && a0.z == b0.z && a0.w == b0.w &&
                               a1.x == b1.x && a1.y == b1.y && a1.z == b1.z && a1.w == b1.w;
                    }
                    hit = hit || same;
                }
                if (__ballot(hit) != 0ull && lane == 0) ccount[7] = 1;
                lds_barrier();
            }
        }
        if (tid == 0) {
            const int Lk = tied ? Lc : Lr;                       // candidates in `sorted`
            if (Lk > M && (sorted[M - 1] >> 32) == (sorted[M] >> 32)) tie = 1;      // (before the replay reorders the first M)
        }
        if (tied && a.tie_order != 0 && ccount[7] != 0)
            tie_order_slow(sorted, keyA, Lc, M, reinterpret_cast<int*>(smem + OFF_STK));
        FAST_STAMP(5);
        {
            // (round 5: a SECOND chunk in flight - requested three iterations ahead - changed nothing, 5.18 against 5.16 us per
            //  iteration beside the projector stream: the loop is not waiting for these loads)
            const int lo2 = lo + 2 * I;
            const int cnt2 = k_it + 2 < n_it ? max(0, min(I, n_rows - lo2)) : 0;
            if (cnt2 > 0) SCAN_WAIT_ROWS(lo2 + cnt2);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = hid + 512 * k;
                pf[k] = (hid >= 0 && e < cnt2 * R) ? scan_load<PERSIST>(lg + (size_t)lo2 * R + e) : 0.0f;
            }
        }
        WSTAMP(12);
        // P6: new memory into the other buffers.  Waves 0..7: the logit rows, four elements per thread (rows (tid >> 3) + 64 k,
        // column tid & 7), and the new rows' maxima; waves 8..15: the exponentials, half a row (16 bytes) per thread, and the
        // patch indices.  Every read of a thread is in flight before its first write.
        {
            const uint32_t* const spos = reinterpret_cast<const uint32_t*>(sorted);         // word 2 j: ~position of rank j
            if (wave < 8) {
                // (addresses from a copy of the thread index the compiler cannot see through: hoisted out of the loop they
                //  were four registers it spilled - four scratch reloads per iteration that went to HBM whenever a producer
                //  beside the loop streamed through the L2, round 5)
                int tl = tid;
                asm volatile("" : "+v"(tl));
                const uint32_t* const sp = spos + 2 * (tl >> 3);
                uint32_t p[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) p[k] = ~sp[128 * k] & (L - 1);
                float gx[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) gx[k] = xc[p[k] * LD + r];
                uint32_t km = 0u;
#pragma unroll
                for (int k = 0; k < 4; ++k) { xn[((tid >> 3) + 64 * k) * LD + r] = gx[k]; km = max(km, max_key(gx[k])); }
                fold_row_max<R>(km, mkey_nx, lane);                // maxima of the NEW memory rows, for the next iteration
            } else {
                const int j = hid >> 1, half = (hid & 1) * 4;
                const uint32_t p = ~spos[2 * j] & (L - 1);
                const uint32_t pc = ~spos[2 * (hid & (M - 1))] & (L - 1);
                const float4 ge = *reinterpret_cast<const float4*>(ec + p * LD + half);
                const int ci = cand[pc];
                *reinterpret_cast<float4*>(en + j * LD + half) = ge;
                if (hid < M) cnew[hid] = ci;
            }
        }
        WSTAMP(13);
        WSTAMP(14);
        FAST_STAMP(6);
        // diagnostic (STAMP build; tools/scan_stamps.py campipe): when this iteration ended (100 MHz clock) and how many rows
        // the loop knew to be published then - the timeline of a call, loop against producer
        if (STAMP && b == 0 && tid == 0 && k_it < 512) {
            stamps[8 * gridDim.x + 4 * k_it + 2] = __builtin_amdgcn_s_memrealtime();
            stamps[8 * gridDim.x + 4 * k_it + 3] = (unsigned long long)ready_known;
        }
#undef WSTAMP
#undef CAM_PREP
    }
    lds_barrier();
    {
        const int parn = (int)(n_iter & 1);                       // the set the last iteration wrote
        const int* const cand = reinterpret_cast<const int*>(smem + OFF_CAND) + parn * L;
        if (tid < M) {
            a.mem_idx[(size_t)b * M + tid] = cand[tid];
            if (a.mem_score) a.mem_score[(size_t)b * M + tid] = n_iter > 0 ? key_score(sorted[tid]) : 0.0f;
        }
    }
    if (a.tie && tid == 0 && tie) a.tie[b] = 1;
    if (STAMP && tid == 0)
        for (int k = 0; k < 8; ++k) stamps[(size_t)b * 8 + k] = tacc[k];
    lds_barrier();                                               // (the next slide starts on the same LDS)
    }
}

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

struct TopmArgs {
    int tie_order, stk_off;
    const float* scores;
    int L, m, n2;
    long long* top;
    int* tie;
};

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

// ------------------------------------------------------------------ candidate sets beyond one compute unit's LDS
// The reference's shipped CAMELYON configuration keeps M = 5000 patches and scores them against I = 5000 new ones
// (config/camelyon_config.yml:35-36): torch.topk ranks L = 10,000 candidates per iteration (ips_net.py:148), a few
// iterations per slide.  Here that is one 1024-thread workgroup per image again, but only the RANKING lives in LDS
// (one array of next_pow2(L) 64-bit keys, L <= 16,384: 128 KiB); everything else goes through a caller-owned workspace
// in global memory that stays in the L2: the candidates' logits staged TRANSPOSED ([row][candidate], so that the
// row-wise passes of the contract - maximum, exponentials, the wave-ordered sum - are coalesced) and the index lists
// of the tie replay.  Same arithmetic, same order of every sum as scan_fast_kernel and the oracle.
constexpr int LARGE_NT = 1024;
constexpr int LARGE_MAX_L = 16384;
constexpr int LARGE_KPT = LARGE_MAX_L / LARGE_NT;              // keys / memory slots a thread may hold in registers
constexpr int LARGE_LEAF_WORDS = LARGE_MAX_L / 64;

// The ranking keys live in LDS with one 8-byte pad per 16 keys: key i at slot i + (i >> 4).  A thread of the sort owns 16
// consecutive keys = 136 consecutive bytes, and 16 lanes at a stride of 136 B cover all 32 banks once - unpadded (128 B)
// every lane of a wavefront would hit the same bank.
__device__ __forceinline__ int large_slot(int i) { return i + (i >> 4); }
__device__ __forceinline__ int next_pow2_dev(int v) { return v <= 1 ? 1 : 1 << (32 - __clz(v - 1)); }
static size_t large_key_bytes(int n2) { return (size_t)(n2 + (n2 >> 4)) * 8; }

// compare-exchange so that x >= y afterwards (descending)
#define IPSX_CE_DESC(x, y)                                     \
    do {                                                       \
        const uint64_t x_ = (x), y_ = (y);                     \
        const bool sw_ = x_ < y_;                              \
        (x) = sw_ ? y_ : x_;                                   \
        (y) = sw_ ? x_ : y_;                                   \
    } while (0)

// 16 keys in registers, descending: bitonic network (80 compare-exchanges, static indices)
__device__ __forceinline__ void sort16_desc(uint64_t (&k)[16]) {
#pragma unroll
    for (int kk = 2; kk <= 16; kk <<= 1)
#pragma unroll
        for (int j = kk >> 1; j >= 1; j >>= 1)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if ((c ^ j) <= c) continue;
                if ((c & kk) == 0 || kk == 16) IPSX_CE_DESC(k[c], k[c ^ j]);
                else IPSX_CE_DESC(k[c ^ j], k[c]);
            }
}

// keys[0, n2) (n2 a power of two >= 64, padded slots, padding keys 0) sorted descending in place by the workgroup: a MERGE
// sort.  Every thread sorts its 16 keys in registers, then log2(n2 / 16) rounds merge neighbouring runs: a thread produces
// the 16 outputs [16 t, 16 t + 16) of its pair of runs - where they start in the two runs is a binary search along the
// merge path (two LDS reads per step), the next 16 keys of either run are read at once (32 independent reads, no dependent
// chain) and the 16 largest of the 32 fall out of half a bitonic merge in registers (max(a[c], b[15 - c]), then four
// compare-exchange stages).  n log n comparisons instead of the bitonic network's n log^2 n: the network's cross-lane
// stages alone were ~10 k VALU instructions per wavefront (200 k cycles for 16,384 keys; this: 3 k).  Keys are unique
// (equal padding zeros aside).  Not inlined (see large_tie_replay); the key array is the start of the dynamic LDS.
// The real keys are keys[0, L): a thread whose 16 outputs lie behind the real keys of its pair of runs (padding zeros:
// 6,384 of 16,384 slots at 10,000 candidates) writes zeros without searching, reading or merging.
__device__ __attribute__((noinline)) void sort_desc_large(int n2, int L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    const int tid = threadIdx.x;
    const bool act = tid < (n2 >> 4);                                 // threads that own a run of 16
    const int o = tid * 16;                                           // first output position of this thread, every round
    uint64_t k[16];
    if (act && o < L) {
#pragma unroll
        for (int c = 0; c < 16; ++c) k[c] = keys[17 * tid + c];
        sort16_desc(k);
#pragma unroll
        for (int c = 0; c < 16; ++c) keys[17 * tid + c] = k[c];
    }
    int steps = 5;                                                    // binary-search steps of a round: log2(len) + 1
    for (int len = 16; len < n2; len <<= 1, ++steps) {
        // runs of up to 512 keys: a pair of runs lies inside ONE wavefront's 1,024 keys, whose lanes run in lockstep and
        // whose LDS operations complete in order - no barrier of the workgroup (twelve of them at 16,384 slots)
        const bool local = 2 * len <= 1024;
        if (local) { wave_lds_fence(); __builtin_amdgcn_wave_barrier(); } else __syncthreads();
        const int base = o & ~(2 * len - 1), diag = o - base;
        const bool pad = diag >= min(2 * len, max(0, L - base));      // all 16 outputs are padding zeros
        if (act && pad) {
#pragma unroll
            for (int c = 0; c < 16; ++c) k[c] = 0ull;
        }
        if (act && !pad) {
            const int bA = base, bB = base + len;
            int lo = diag > len ? diag - len : 0, hi = diag < len ? diag : len;
            for (int it = 0; it < steps; ++it) {                      // (uniform trip count; finished lanes idle)
                const int mid = (lo + hi) >> 1;
                const bool go = lo < hi;
                const uint64_t av = keys[large_slot(bA + (go ? mid : 0))];
                const uint64_t bv = keys[large_slot(bB + (go ? diag - 1 - mid : 0))];
                if (go) { if (av > bv) lo = mid + 1; else hi = mid; }
            }
            const int ai = lo, bi = diag - lo;
            uint64_t av[16], bv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                av[c] = keys[large_slot(bA + (ai + c < len ? ai + c : len - 1))];
                bv[c] = keys[large_slot(bB + (bi + c < len ? bi + c : len - 1))];
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (ai + c >= len) av[c] = 0ull;
                if (bi + c >= len) bv[c] = 0ull;
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) k[c] = av[c] > bv[15 - c] ? av[c] : bv[15 - c];   // the 16 largest, a bitonic sequence
#pragma unroll
            for (int j = 8; j >= 1; j >>= 1)
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if ((c & j) == 0) IPSX_CE_DESC(k[c], k[c | j]);
        }
        if (local) { wave_lds_fence(); __builtin_amdgcn_wave_barrier(); } else __syncthreads();   // every read of this round is done
        if (act) {
#pragma unroll
            for (int c = 0; c < 16; ++c) keys[17 * tid + c] = k[c];
        }
    }
    __syncthreads();
}
#undef IPSX_CE_DESC

// Before the sort: only the first m + 1 ranks are ever used in order (the new memory, the tie test, the replay's copies
// and tie bits) - the other candidates only have to EXIST for the replay to put them back into candidate order.  A
// threshold score T is taken from a sample (every wavefront sorts 64 of its keys in registers and reports the one at the
// target quantile; T = the median of the 16 reports), the keys at or above it are COUNTED exactly (S) and, when S lies
// between need and half the slots, moved to the front ([0, S)), zero padding up to the power of two n2s behind them, the
// rest behind that - and the merge sort then runs on n2s slots with S real keys instead of n2 slots with L (10,000
// candidates, M = 5000: 8,192 slots with ~6,600 keys instead of 16,384 with 10,000).  Returns S and n2s, or false when
// the sample missed or nothing is gained (the sort then takes everything as before: the result is the same either way).
// All threads; contains barriers; not inlined.
__device__ __attribute__((noinline)) bool select_top_large(int n2, int L, int need, int tail, int* S_out, int* n2s_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    int* sel = reinterpret_cast<int*>(smem + tail);                                        // (the replay's stack: 192 ints)
    int* cnt = reinterpret_cast<int*>(smem + tail) + 3 * stdorder::STACK_RANGES;          // (its leaf bitmap: 512 ints)
    constexpr int NW = LARGE_NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = n2 >> 1;
    if (need >= half - (half >> 3)) return false;                         // (workgroup-uniform: not enough to gain)
    uint32_t sk[LARGE_KPT];
#pragma unroll
    for (int c = 0; c < LARGE_KPT; ++c) {
        const int l = tid + c * LARGE_NT;
        sk[c] = l < L ? (uint32_t)(keys[large_slot(l)] >> 32) : 0u;
    }
    // the sample: lane's key of slot (lane + wave) mod its valid slots - spread over memory and chunk candidates alike
    const int nvalid = (L - tid + LARGE_NT - 1) / LARGE_NT;               // >= 1 for tid < L (L >= 2048 > tid)
    const int pick = (lane + 5 * wave) % nvalid;
    uint32_t smp = 0u;
#pragma unroll
    for (int c = 0; c < LARGE_KPT; ++c) smp = c == pick ? sk[c] : smp;
    const uint32_t dir = sort_directions(lane);
    smp = wave_sort_desc_u32(smp, dir, lane);                             // lane j: the wavefront's j-th largest sample
    // aim a little above what is needed: the median of 16 quantiles of 64 samples is off by ~2 % of L (one sigma); a miss
    // on the low side falls back to the whole sort
    const int target = min(need + L / 12, (need + half) >> 1);
    const int qi = min(63, max(0, (int)(((long long)target * 64) / L)));
    const uint32_t rep = (uint32_t)__shfl((int)smp, qi, 64);
    if (lane == 0) sel[8 + wave] = (int)rep;
    __syncthreads();
    uint32_t T;
    {
        uint32_t v = lane < NW ? (uint32_t)sel[8 + lane] : 0u;           // 16 reports, the rest 0: sorted descending they lead
        v = wave_sort_desc_u32(v, dir, lane);
        T = (uint32_t)__shfl((int)v, NW / 2, 64);                         // the median report
    }
    // ---- exact count of the keys at or above T
    int mine = 0;
#pragma unroll
    for (int c = 0; c < LARGE_KPT; ++c) mine += (tid + c * LARGE_NT < L && sk[c] >= T) ? 1 : 0;
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
    __syncthreads();                                                      // (sel[8 ..] read by every wavefront above)
    if (lane == 0) sel[8 + wave] = mine;
    __syncthreads();
    int S = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) S += sel[8 + w];
    __syncthreads();
    const int n2s = max(64, next_pow2_dev(S));
    if (S < need || n2s >= n2 || n2s + (L - S) > n2) return false;
    // ---- compaction through registers: counts per (slot c, wavefront), an exclusive scan of the 256 + 256 counts by the
    // first wavefront, then every key to its place (the order inside the two groups is immaterial)
    uint64_t hold[LARGE_KPT];
    unsigned long long selm[LARGE_KPT];
#pragma unroll
    for (int c = 0; c < LARGE_KPT; ++c) {
        const int l = tid + c * LARGE_NT;
        hold[c] = l < L ? keys[large_slot(l)] : 0ull;
        const bool is = l < L && sk[c] >= T;
        selm[c] = __ballot(is);
        const unsigned long long nonm = __ballot(l < L && !is);
        if (lane == 0) { cnt[c * NW + wave] = __popcll(selm[c]); cnt[256 + c * NW + wave] = __popcll(nonm); }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            int v[4], tot = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = cnt[half * 256 + 4 * lane + k]; tot += v[k]; }
            int incl = tot;
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_up(incl, off, 64);
                if (lane >= off) incl += u;
            }
            int run = incl - tot;
#pragma unroll
            for (int k = 0; k < 4; ++k) { cnt[half * 256 + 4 * lane + k] = run; run += v[k]; }
        }
    }
    for (int l = tid; l < n2; l += LARGE_NT) keys[large_slot(l)] = 0ull;   // (every key is in `hold` by now: barrier above)
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int c = 0; c < LARGE_KPT; ++c) {
        const int l = tid + c * LARGE_NT;
        if (l < L) {
            const bool is = (selm[c] >> lane) & 1ull;
            const unsigned long long valid = l - lane + 63 < L ? ~0ull : ((1ull << (L - (l - lane))) - 1ull);
            const unsigned long long nonm = ~selm[c] & valid;
            const int dst = is ? cnt[c * NW + wave] + __popcll(selm[c] & below)
                               : n2s + cnt[256 + c * NW + wave] + __popcll(nonm & below);
            keys[large_slot(dst)] = hold[c];
        }
    }
    __syncthreads();
    *S_out = S;
    *n2s_out = n2s;
    return true;
}

// keys = the L ranked keys (canonical order) in LDS.  When two of the first m + 1 ranked scores are equal and the tie
// order is the reference's, the key array is turned - through registers, in place - into the (score, position) pairs in
// CANDIDATE order and torch.topk's routines are replayed on them (one wavefront; the index lists in the workspace); q[0, m)
// is then the answer.  Returns whether that happened (workgroup-uniform).  All threads; contains barriers.
// (Not inlined, like the sort; `tail` = LDS offset of the stack / leaf bitmap / range lists behind the keys.)
__device__ __attribute__((noinline)) bool large_tie_replay(int L, int m, int n2, int tie_order, int* lists, int tail,
                                                           uint64_t* canon, bool rst, const TieRows* rows = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    int* stk = reinterpret_cast<int*>(smem + tail);
    unsigned long long* leaf = reinterpret_cast<unsigned long long*>(stk + 3 * stdorder::STACK_RANGES);
    int* queue = reinterpret_cast<int*>(leaf + LARGE_LEAF_WORDS);
    const int tid = threadIdx.x;
    if (tie_order == 0 || !ranked_ties_padded(keys, L, m, tid & 63, tie_order, rows)) return false;
    unsigned long long* tiebits = reinterpret_cast<unsigned long long*>(queue + 2 + 2 * BLOCK_QCAP);
    uint64_t hold[LARGE_KPT];
#pragma unroll
    for (int s = 0; s < LARGE_KPT; ++s) {
        const int j = tid + s * LARGE_NT;
        hold[s] = j < n2 ? keys[large_slot(j)] : 0ull;                      // (after select_top_large the candidates below
        if (canon && j < L) canon[j] = hold[s];                             //  the first m + 1 ranks sit behind a stretch of zeros)
        const uint64_t next = j + 1 < L ? keys[large_slot(j + 1)] : 0ull;   // (a wavefront's 64 ranks are one word of the bitmap)
        const unsigned long long word = __ballot(j + 1 < L && (hold[s] >> 32) == (next >> 32));
        if ((tid & 63) == 0) tiebits[(tid >> 6) + s * (LARGE_NT / 64)] = word;
    }
    __syncthreads();
    stdorder::E* q = reinterpret_cast<stdorder::E*>(keys);
#pragma unroll
    for (int s = 0; s < LARGE_KPT; ++s) {
        if (hold[s] != 0ull) {                                              // (a real key is never 0: padding is)
            const int p = (int)key_pos(hold[s]);
            q[p].v = key_score(hold[s]);
            q[p].i = p;
        }
    }
    __syncthreads();
    // the lists of the sort phase (2 (m - 1) ints) live in LDS behind the L pairs when the power-of-two key array has that
    // much room (10,000 candidates in 16,384 slots: yes) - a partition of a short range is then a few LDS round trips instead
    // of a few L2 round trips, and there are hundreds of them
    if (n2 - L >= m) {
        int* ls = reinterpret_cast<int*>(keys + L);
        torch_topk_block<LARGE_NT, false>(q, L, m, lists, lists + L, ls, ls + (m - 1), stk, leaf, LARGE_LEAF_WORDS, queue + 2, queue, tiebits, canon, rst);
    } else {
        torch_topk_block<LARGE_NT, true>(q, L, m, lists, lists + L, lists, lists + L, stk, leaf, LARGE_LEAF_WORDS, queue + 2, queue, tiebits, canon, rst);
    }
    return true;
}

struct LargeArgs {
    int tie_order;
    int rstamp;            // 1: the replay's phase stamps are collected (ipsx_dbg_replay_stamps)
    int direct;            // 1: the register-resident passes for 8 heads x one token (diagnostic ipsx_dbg_scan_direct(0): off)
    const float* lg;       // (b, n, R)
    long long n;
    long long it0, it1;
    int m, i, h, T, n2, Lp;
    long long* mem_idx;
    float* mem_score;
    int* tie;
    unsigned char* ws;     // per image: R * Lp floats (staged logits / exponentials) + 2 * Lp ints (tie replay lists)
    size_t ws_per_image;
    const int* ready;      // persistent launch (ipsx_scan_persistent_ws): rows whose logits are in memory, per image or one word
    int ready_stride, ready_words;
    unsigned long long wait_ticks;
    int* status;           //   bit 0: gave up waiting, bit 1: resident
    const int* cond;       // conditional launch (ipsx_scan_range_if_ws): run only when (*cond & cond_mask) != 0
    int cond_mask;
};

// keys (padded) | row maxima, denominators | stack of the sequential fallbacks | leaf bitmap | two range lists + counters
constexpr size_t LARGE_TAIL_BYTES = (size_t)3 * stdorder::STACK_RANGES * 4 + (size_t)LARGE_LEAF_WORDS * 8 +
                                    (size_t)(2 * BLOCK_QCAP + 2) * 4 + (size_t)LARGE_LEAF_WORDS * 8;         // ... | tie bitmap
static size_t large_lds_bytes(int n2, int R) { return large_key_bytes(n2) + (size_t)((R + 1) & ~1) * 8 + LARGE_TAIL_BYTES; }

// A pass over global memory at 16 waves per compute unit is bound by round trips, not by bandwidth: every loop below
// keeps LARGE_U independent loads of a thread in flight before it uses the first.
constexpr int LARGE_U = 8;

#define LARGE_STAMP(k)                                                             \
    do {                                                                           \
        if (STAMP) {                                                               \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();            \
            if (tid == 0) tacc[k] += t_ - tlast;                                   \
            tlast = t_;                                                            \
        }                                                                          \
    } while (0)

template <bool STAMP>
__global__ __launch_bounds__(LARGE_NT) void scan_large_kernel(LargeArgs a, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int R = a.h * a.T, Lp = a.Lp, m = a.m;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    uint32_t* rmaxkey = reinterpret_cast<uint32_t*>(keys + a.n2 + (a.n2 >> 4));   // row maxima as order-preserving keys (max_key)
    float* rden = reinterpret_cast<float*>(rmaxkey + ((R + 1) & ~1));
    const int tail = (a.n2 + (a.n2 >> 4)) * 8 + ((R + 1) & ~1) * 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    constexpr int NW = LARGE_NT / 64;
    const float* lg = a.lg + (size_t)b * a.n * R;
    long long* mem = a.mem_idx + (size_t)b * m;
    float* xT = reinterpret_cast<float*>(a.ws + (size_t)b * a.ws_per_image);
    int* lists = reinterpret_cast<int*>(xT + (size_t)R * Lp);
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;
    if (scan_skipped(a.cond, a.cond_mask)) return;                     // (the recovery launch behind a persistent loop)
    if (a.ready && tid == 0) {
        __hip_atomic_fetch_or(a.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&g_persist_log[4], (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int* const wword = reinterpret_cast<int*>(smem + tail);           // (the replay's stack: free outside the replay)
    long long ready_known = 0;
    if (a.it0 == 0)
        for (int j = tid; j < m; j += LARGE_NT) mem[j] = j;
    __syncthreads();
    int tie = 0;
    // row maxima in `seg` stretches per row so that every wavefront has one (max is order-free; a NaN wins: max_key)
    const int seg = R >= NW ? 1 : NW / R;
    for (long long it = a.it0; it < a.it1; ++it) {
        const long long lo = it * a.i + m;
        const int cnt = (int)std::min<long long>(a.i, a.n - lo);
        const int L = m + cnt;
        if (a.ready && lo + cnt > ready_known) {
            // persistent: the rows of this iteration's chunk must have been published (the wait of scan_fast_kernel:
            // bounded, any progress word moving restarts the clock; then ONE acquire, and plain loads are good)
            if (wave == 0) {
                unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                int v = __hip_atomic_load(a.ready + b * a.ready_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int seen = -1;
                while (v >= 0 && v < lo + cnt) {
                    __builtin_amdgcn_s_sleep(16);
                    int w = lane < a.ready_words ? __hip_atomic_load(a.ready + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                    for (int o = 32; o >= 1; o >>= 1) w += __shfl_xor(w, o, 64);
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if (w != seen) { seen = w; t0 = now; }
                    if (now - t0 > a.wait_ticks) { v = -1; break; }
                    v = __hip_atomic_load(a.ready + b * a.ready_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (lane == 0) wword[0] = v;
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            ready_known = wword[0];
            __syncthreads();
            if (ready_known < 0) {
                if (tid == 0) __hip_atomic_fetch_or(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
        // 8 heads, one token (the reference's shipped CAMELYON configuration): a thread gathers ITS candidates' 8 logits -
        // 32 contiguous bytes each, five candidates in flight - for the row maxima, and again (from L2) for the
        // exponentials, which it writes transposed for the row sums and the scores.  The generic path below stages the
        // logits transposed first: three passes over 320 KB and two writes of it where this has two gathers and one
        // write.  Same values, same order of every sum.
        constexpr int DG = 5;
        const bool direct = a.direct && R == 8 && a.T == 1;        // (uniform)
        if (direct) {
            uint32_t km[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) km[r] = 0u;
            for (int r = tid; r < R; r += LARGE_NT) rmaxkey[r] = 0u;
            for (int l0 = tid; l0 < L; l0 += DG * LARGE_NT) {
                float4 v[DG][2];
#pragma unroll
                for (int c = 0; c < DG; ++c) {
                    const int l = l0 + c * LARGE_NT;
                    const size_t row = l >= L ? (size_t)0 : (l < m ? (size_t)mem[l] : (size_t)(lo + (l - m)));
                    const float4* src = reinterpret_cast<const float4*>(lg + row * 8);
                    v[c][0] = src[0];
                    v[c][1] = src[1];
                }
#pragma unroll
                for (int c = 0; c < DG; ++c)
                    if (l0 + c * LARGE_NT < L) {
                        km[0] = max(km[0], max_key(v[c][0].x)); km[1] = max(km[1], max_key(v[c][0].y));
                        km[2] = max(km[2], max_key(v[c][0].z)); km[3] = max(km[3], max_key(v[c][0].w));
                        km[4] = max(km[4], max_key(v[c][1].x)); km[5] = max(km[5], max_key(v[c][1].y));
                        km[6] = max(km[6], max_key(v[c][1].z)); km[7] = max(km[7], max_key(v[c][1].w));
                    }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                uint32_t best = km[r];
                best = max(best, lane_xor_u32<32>(best, lane)); best = max(best, lane_xor_u32<16>(best, lane));
                best = max(best, lane_xor_u32<8>(best, lane)); best = max(best, lane_xor_u32<4>(best, lane));
                best = max(best, lane_xor_u32<2>(best, lane)); best = max(best, lane_xor_u32<1>(best, lane));
                km[r] = best;
            }
            __syncthreads();                                        // (rmaxkey zeroed)
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < 8; ++r) atomicMax(&rmaxkey[r], km[r]);
            }
            __syncthreads();
            LARGE_STAMP(0);
            LARGE_STAMP(1);
            {
                float mx[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) mx[r] = max_key_value(rmaxkey[r]);
                for (int l0 = tid; l0 < L; l0 += DG * LARGE_NT) {
                    float4 v[DG][2];
#pragma unroll
                    for (int c = 0; c < DG; ++c) {
                        const int l = l0 + c * LARGE_NT;
                        const size_t row = l >= L ? (size_t)0 : (l < m ? (size_t)mem[l] : (size_t)(lo + (l - m)));
                        const float4* src = reinterpret_cast<const float4*>(lg + row * 8);
                        v[c][0] = src[0];
                        v[c][1] = src[1];
                    }
#pragma unroll
                    for (int c = 0; c < DG; ++c) {
                        const int l = l0 + c * LARGE_NT;
                        if (l < L) {
                            float* dst = xT + l;
                            dst[0] = det_expf(v[c][0].x - mx[0]);
                            dst[Lp] = det_expf(v[c][0].y - mx[1]);
                            dst[2 * (size_t)Lp] = det_expf(v[c][0].z - mx[2]);
                            dst[3 * (size_t)Lp] = det_expf(v[c][0].w - mx[3]);
                            dst[4 * (size_t)Lp] = det_expf(v[c][1].x - mx[4]);
                            dst[5 * (size_t)Lp] = det_expf(v[c][1].y - mx[5]);
                            dst[6 * (size_t)Lp] = det_expf(v[c][1].z - mx[6]);
                            dst[7 * (size_t)Lp] = det_expf(v[c][1].w - mx[7]);
                        }
                    }
                }
            }
            __syncthreads();
            LARGE_STAMP(2);
            // ---- denominators in the wavefront order of the contract: lane j adds elements j, j + 64, ... ascending
            for (int r = wave; r < R; r += NW) {
                const float* x = xT + (size_t)r * Lp;
                float sum = 0.0f;
                for (int l0 = lane; l0 < L; l0 += 64 * 2 * LARGE_U) {
                    float v[2 * LARGE_U];
#pragma unroll
                    for (int u = 0; u < 2 * LARGE_U; ++u) {
                        const int l = l0 + 64 * u;
                        v[u] = x[l < L ? l : l0];
                    }
#pragma unroll
                    for (int u = 0; u < 2 * LARGE_U; ++u)
                        if (l0 + 64 * u < L) sum = sum + v[u];
                }
                sum = wave_butterfly_sum(sum);
                if (lane == 0) rden[r] = 1.0f / sum;                // (the reciprocal: weights are e * (1 / den))
            }
            __syncthreads();
            LARGE_STAMP(3);
        } else {
            // ---- candidates' logits, memory first, transposed into the workspace: a thread takes 4 candidates at a time
            for (int l0 = tid; l0 < L; l0 += LARGE_NT * 4) {
                size_t row[4];
                bool ok[4];
    #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int l = l0 + u * LARGE_NT;
                    ok[u] = l < L;
                    row[u] = !ok[u] ? (size_t)0 : (l < m ? (size_t)mem[l] : (size_t)(lo + (l - m)));
                }
                if ((R & 3) == 0) {
                    for (int r = 0; r < R; r += 8) {
                        const bool two = r + 4 < R;
                        float4 v[4][2];
    #pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float4* src = reinterpret_cast<const float4*>(lg + row[u] * R + r);
                            v[u][0] = src[0];
                            v[u][1] = src[two ? 1 : 0];
                        }
    #pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (!ok[u]) continue;
                            float* dst = xT + (size_t)r * Lp + (l0 + u * LARGE_NT);
                            dst[0] = v[u][0].x; dst[Lp] = v[u][0].y; dst[2 * (size_t)Lp] = v[u][0].z; dst[3 * (size_t)Lp] = v[u][0].w;
                            if (two) {
                                dst += 4 * (size_t)Lp;
                                dst[0] = v[u][1].x; dst[Lp] = v[u][1].y; dst[2 * (size_t)Lp] = v[u][1].z; dst[3 * (size_t)Lp] = v[u][1].w;
                            }
                        }
                    }
                } else {
                    for (int r = 0; r < R; ++r) {
                        float v[4];
    #pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = lg[row[u] * R + r];
    #pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (ok[u]) xT[(size_t)r * Lp + (l0 + u * LARGE_NT)] = v[u];
                    }
                }
            }
            for (int r = tid; r < R; r += LARGE_NT) rmaxkey[r] = 0u;
            __syncthreads();
            LARGE_STAMP(0);
            // ---- row maxima
            {
                const int seg_len = ((L + seg - 1) / seg + 63) & ~63;
                for (int unit = wave; unit < R * seg; unit += NW) {
                    const int r = unit / seg, sg = unit - r * seg;
                    const float* x = xT + (size_t)r * Lp;
                    const int l_end = std::min(L, (sg + 1) * seg_len);
                    uint32_t best = 0u;
                    for (int l0 = sg * seg_len + lane; l0 < l_end; l0 += 64 * LARGE_U) {
                        float v[LARGE_U];
    #pragma unroll
                        for (int u = 0; u < LARGE_U; ++u) {
                            const int l = l0 + 64 * u;
                            v[u] = x[l < l_end ? l : l0];
                        }
    #pragma unroll
                        for (int u = 0; u < LARGE_U; ++u) {
                            const uint32_t k = max_key(v[u]);
                            best = k > best ? k : best;
                        }
                    }
                    best = max(best, lane_xor_u32<32>(best, lane)); best = max(best, lane_xor_u32<16>(best, lane));
                    best = max(best, lane_xor_u32<8>(best, lane)); best = max(best, lane_xor_u32<4>(best, lane));
                    best = max(best, lane_xor_u32<2>(best, lane)); best = max(best, lane_xor_u32<1>(best, lane));
                    if (lane == 0) atomicMax(&rmaxkey[r], best);
                }
            }
            __syncthreads();
            LARGE_STAMP(1);
            // ---- exponentials, in place: blocks of 64 candidates of one row, LARGE_U blocks of a wavefront in flight
            {
                const int bpr = Lp >> 6;                                   // blocks per row
                const int nblk = R * bpr;
                for (int b0 = wave; b0 < nblk; b0 += NW * LARGE_U) {
                    float v[LARGE_U], mx[LARGE_U];
                    float* px[LARGE_U];
                    bool ok[LARGE_U];
    #pragma unroll
                    for (int u = 0; u < LARGE_U; ++u) {
                        const int blk = b0 + u * NW;
                        const int r = blk < nblk ? blk / bpr : 0;
                        const int l = (blk - r * bpr) * 64 + lane;
                        ok[u] = blk < nblk && l < L;
                        px[u] = xT + (size_t)r * Lp + (ok[u] ? l : 0);
                        mx[u] = max_key_value(rmaxkey[r]);
                        v[u] = *px[u];
                    }
    #pragma unroll
                    for (int u = 0; u < LARGE_U; ++u)
                        if (ok[u]) *px[u] = det_expf(v[u] - mx[u]);
                }
            }
            __syncthreads();
            LARGE_STAMP(2);
            // ---- denominators in the wavefront order of the contract: lane j adds elements j, j + 64, ... ascending
            for (int r = wave; r < R; r += NW) {
                const float* x = xT + (size_t)r * Lp;
                float sum = 0.0f;
                for (int l0 = lane; l0 < L; l0 += 64 * 2 * LARGE_U) {
                    float v[2 * LARGE_U];
    #pragma unroll
                    for (int u = 0; u < 2 * LARGE_U; ++u) {
                        const int l = l0 + 64 * u;
                        v[u] = x[l < L ? l : l0];
                    }
    #pragma unroll
                    for (int u = 0; u < 2 * LARGE_U; ++u)
                        if (l0 + 64 * u < L) sum = sum + v[u];
                }
                sum = wave_butterfly_sum(sum);
                if (lane == 0) rden[r] = 1.0f / sum;                // (the reciprocal: weights are e * (1 / den))
            }
            __syncthreads();
            LARGE_STAMP(3);
        }
        // ---- scores: mean over heads, then over tokens; ranking keys
        for (int l = tid; l < a.n2; l += LARGE_NT) {
            uint64_t key = 0ull;
            if (l < L) {
                float st = 0.0f;
                for (int t = 0; t < a.T; ++t) {
                    float sh = 0.0f;
                    for (int hh = 0; hh < a.h; ++hh) {
                        const int r = hh * a.T + t;
                        sh = sh + xT[(size_t)r * Lp + l] * rden[r];
                    }
                    st = st + sh / (float)a.h;
                }
                key = rank_key(st / (float)a.T, (uint32_t)l);
            }
            keys[large_slot(l)] = key;
        }
        __syncthreads();
        LARGE_STAMP(4);
        {
            int S = L, n2s = a.n2;
            if (a.direct && L > m + 1 && L >= 2048) (void)select_top_large(a.n2, L, m + 1, tail, &S, &n2s);
            sort_desc_large(n2s, S);
        }
        LARGE_STAMP(5);
        if (tid == 0 && L > m && (keys[large_slot(m - 1)] >> 32) == (keys[large_slot(m)] >> 32)) tie = 1;
        // (the exponentials' workspace is free by now: the canonical ranking goes there when it fits - 8 B per candidate)
        const TieRows rows = {lg, mem, lo, m, R};
        const bool replayed = large_tie_replay(L, m, a.n2, a.tie_order, lists, tail,
                                               R >= 2 ? reinterpret_cast<uint64_t*>(xT) : nullptr, a.rstamp != 0, &rows);
        LARGE_STAMP(6);
        const stdorder::E* q = reinterpret_cast<const stdorder::E*>(keys);
        const bool want_score = a.mem_score != nullptr && it + 1 == a.it1;
        int nw[LARGE_KPT];
#pragma unroll
        for (int s = 0; s < LARGE_KPT; ++s) {
            const int j = tid + s * LARGE_NT;
            nw[s] = 0;
            if (j < m) {
                int pos;
                float sc;
                if (replayed) {
                    pos = q[j].i;
                    sc = key_score(rank_key(q[j].v, 0u));
                } else {
                    pos = (int)key_pos(keys[large_slot(j)]);
                    sc = key_score(keys[large_slot(j)]);
                }
                if (want_score) a.mem_score[(size_t)b * m + j] = sc;
                nw[s] = pos < m ? (int)mem[pos] : (int)(lo + (pos - m));
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < LARGE_KPT; ++s) {
            const int j = tid + s * LARGE_NT;
            if (j < m) mem[j] = nw[s];
        }
        __syncthreads();
        LARGE_STAMP(7);
    }
    if (a.tie && tid == 0 && tie) a.tie[b] = 1;
    if (STAMP && tid == 0)
        for (int k = 0; k < 8; ++k) stamps[(size_t)b * 8 + k] += tacc[k];
}
#undef LARGE_STAMP

// torch.topk(scores, m)[1] for l <= 16,384 candidates per row: the ranking of scan_large_kernel alone
__global__ __launch_bounds__(LARGE_NT) void topm_large_kernel(TopmArgs a, unsigned char* ws, size_t ws_per_row) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int l = tid; l < a.n2; l += LARGE_NT)
        keys[large_slot(l)] = l < a.L ? rank_key(a.scores[(size_t)b * a.L + l], (uint32_t)l) : 0ull;
    __syncthreads();
    sort_desc_large(a.n2, a.L);
    if (a.tie && tid == 0)
        a.tie[b] = (a.L > a.m && (keys[large_slot(a.m - 1)] >> 32) == (keys[large_slot(a.m)] >> 32)) ? 1 : 0;
    int* lists = reinterpret_cast<int*>(ws + (size_t)b * ws_per_row);
    const bool replayed = large_tie_replay(a.L, a.m, a.n2, a.tie_order, lists, (a.n2 + (a.n2 >> 4)) * 8, nullptr, false);
    const stdorder::E* q = reinterpret_cast<const stdorder::E*>(keys);
    for (int j = tid; j < a.m; j += LARGE_NT)
        a.top[(size_t)b * a.m + j] = replayed ? (long long)q[j].i : (long long)key_pos(keys[large_slot(j)]);
}

static unsigned long long* g_scan_stamps = nullptr;   // diagnostic only (ipsx_dbg_scan_stamps)

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static const size_t kLdsLimit = 160 * 1024;

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

// Is this shape the LDS-resident loop's (scan_fast_kernel)?  Otherwise scan_large_kernel takes it (and needs a workspace).
struct FastPlan {
    bool ok;
    int ept, lch;
    size_t lds;
};

static int g_persist_wait_ms = 50;         // ipsx_set_persistent_wait_ms: longest wait of a persistent loop / its gate without progress
static bool g_scan_generic = false;        // diagnostic (ipsx_dbg_scan_generic): every shape through scan_large_kernel
static bool g_replay_stamps_on = false;    // diagnostic (ipsx_dbg_replay_stamps): the replay's phases are stamped from the first read on
static bool g_scan_direct = true;          // diagnostic (ipsx_dbg_scan_direct): 0 = scan_large_kernel's five generic passes for every shape
static bool g_scan_r8 = true;              // diagnostic (ipsx_dbg_scan_r8): 0 sends the shape of scan_cam_kernel through scan_fast_kernel

static FastPlan scan_fast_plan(int m, int i, int h, int n_token) {
    FastPlan p = {false, 1, 2, 0};
    const int R = h * n_token, Lmax = m + i, n2 = next_pow2(Lmax);
    if (!((R == 8 && n_token == 1) || (R == 32 && n_token == 4))) return p;      // the instantiated (R, T) pairs
    while ((size_t)p.ept * SCAN_NT < (size_t)Lmax * R) p.ept <<= 1;
    p.lch = Lmax <= 128 ? 2 : (Lmax <= 512 ? 8 : 16);
    const size_t stage = (size_t)Lmax * (R + 1) * 4;
    const int pad = (4 - ((2 * Lmax) & 3)) & 3;
    const size_t fixed = (size_t)n2 * 16 + (size_t)(2 * Lmax + pad) * 4 + (size_t)R * 19 * 4 + 96;
    p.lds = ((fixed + 4 * stage + 15) & ~(size_t)15) + STK_BYTES;
    const bool scratch_fits = (size_t)((Lmax + 63) / 64) * 64 * 8 <= (size_t)m * (R + 1) * 4;     // inside the memory rows
    const bool pf_fits = (size_t)i * R <= (size_t)(Lmax > 128 ? SCAN_NT - 320 : SCAN_NT) * scan_pf(R, p.lch);
    p.ok = p.ept <= 8 && Lmax <= SCAN_NT && pf_fits && scratch_fits && p.lds <= kLdsLimit;
    return p;
}

static size_t scan_large_ws_per_image(int m, int i, int h, int n_token) {
    const size_t Lp = ((size_t)(m + i) + 63) & ~(size_t)63;
    return ((size_t)h * n_token * Lp * 4 + 2 * Lp * 4 + 255) & ~(size_t)255;
}

IPSX_API size_t ipsx_scan_workspace_bytes(int b, int m, int i, int h, int n_token) {
    if (b <= 0 || m <= 0 || i <= 0 || h <= 0 || n_token <= 0) return 0;
    if (scan_fast_plan(m, i, h, n_token).ok && !g_scan_generic) return 0;
    return (size_t)b * scan_large_ws_per_image(m, i, h, n_token);
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
    return g_scan_r8 && h * n_token == cam::R && n_token == 1 && m == cam::M && i == cam::I ? 1 : 0;
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
    const int R = h * n_token, Lmax = m + i, n2 = std::max(64, next_pow2(Lmax));
    const FastPlan fp = scan_fast_plan(m, i, h, n_token);
    if (!fp.ok || (g_scan_generic && !ready && !cond)) {       // (forced generic: plain launches only)
        // every shape the LDS-resident loop does not cover - other head / token counts, candidate sets beyond the LDS (the
        // reference's shipped CAMELYON configuration: M = I = 5000): ranking in LDS, everything else through the
        // caller's workspace (scan_large_kernel)
        IPSX_REQUIRE(!(ready && cond), "scan: a persistent launch is not conditional");
        IPSX_REQUIRE(!ready || status, "scan_persistent: needs the status word");
        IPSX_REQUIRE(Lmax <= LARGE_MAX_L, "scan: M+I = %d candidates - at most %d are supported", Lmax, LARGE_MAX_L);
        IPSX_REQUIRE(R <= 256, "scan: H * n_token = %d > 256 not supported", R);
        const size_t need = (size_t)b * scan_large_ws_per_image(m, i, h, n_token);
        if (!workspace || workspace_bytes < need)
            return fail(IPSX_EWORKSPACE, "scan: M=%d I=%d H=%d n_token=%d needs a workspace of %zu B (ipsx_scan_workspace_bytes), got %zu",
                        m, i, h, n_token, need, workspace_bytes);
        LargeArgs la;
        la.tie_order = g_tie_order;
        la.direct = g_scan_direct ? 1 : 0;
        la.rstamp = g_replay_stamps_on ? 1 : 0;
        la.lg = logits; la.n = n; la.it0 = it_begin; la.it1 = it_end;
        la.m = m; la.i = i; la.h = h; la.T = n_token; la.n2 = n2; la.Lp = (Lmax + 63) & ~63;
        la.mem_idx = reinterpret_cast<long long*>(mem_idx); la.mem_score = mem_score; la.tie = tie_flag;
        la.ws = static_cast<unsigned char*>(workspace); la.ws_per_image = scan_large_ws_per_image(m, i, h, n_token);
        la.ready = ready; la.status = status; la.ready_stride = ready_stride;
        la.ready_words = ready ? (ready_stride ? std::min(b, 64) : 1) : 0;
        la.wait_ticks = (unsigned long long)g_persist_wait_ms * 100000ull;
        la.cond = cond; la.cond_mask = cond_mask;
        const size_t lds = large_lds_bytes(n2, R);
        IPSX_REQUIRE(lds <= kLdsLimit, "scan: internal - %zu B of LDS", lds);
        if (g_scan_stamps) {                                           // diagnostic build (tools/scan_stamps.py large)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_large_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            scan_large_kernel<true><<<dim3((unsigned)b), dim3(LARGE_NT), lds, as_stream(stream)>>>(la, g_scan_stamps);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_large_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            scan_large_kernel<false><<<dim3((unsigned)b), dim3(LARGE_NT), lds, as_stream(stream)>>>(la, nullptr);
        }
        return launched("scan");
    }
    ScanArgs a;
    a.lg = logits; a.n = n; a.m = m; a.i = i; a.h = h; a.T = n_token; a.n2 = next_pow2(Lmax);
    a.it0 = it_begin; a.it1 = it_end;
    a.mem_idx = reinterpret_cast<long long*>(mem_idx); a.mem_score = mem_score; a.tie = tie_flag;
    a.ready = ready; a.status = status; a.ready_stride = ready_stride;
    a.ready_words = ready ? (ready_stride ? std::min(b, 64) : 1) : 0;
    a.wait_ticks = (unsigned long long)g_persist_wait_ms * 100000ull;
    a.cond = cond; a.cond_mask = cond_mask;
    a.slides = b;
    a.tie_order = g_tie_order;
    a.use_lds = 1;
    a.stk_off = (int)(fp.lds - STK_BYTES);
    const size_t fast = fp.lds;
    const int ept = fp.ept, lch = fp.lch;
    unsigned long long* st = g_scan_stamps;
#define IPSX_LAUNCH_FAST(RR, TT, E, C, S)                                                                           \
    do {                                                                                                            \
        if (a.ready) {                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_fast_kernel<RR, TT, E, C, false, true>),   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast);                       \
            scan_fast_kernel<RR, TT, E, C, false, true><<<dim3((unsigned)b), dim3(SCAN_NT), fast, as_stream(stream)>>>(a, nullptr); \
        } else {                                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_fast_kernel<RR, TT, E, C, S, false>),      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast);                       \
            scan_fast_kernel<RR, TT, E, C, S, false><<<dim3((unsigned)b), dim3(SCAN_NT), fast, as_stream(stream)>>>(a, st); \
        }                                                                                                           \
        return launched("scan");                                                                                    \
    } while (0)
#define IPSX_LAUNCH_FAST_C(RR, TT, E)                                                                               \
    do {                                                                                                            \
        if (lch == 2) IPSX_LAUNCH_FAST(RR, TT, E, 2, false);                                                        \
        else if (lch == 8) IPSX_LAUNCH_FAST(RR, TT, E, 8, false);                                                   \
        else IPSX_LAUNCH_FAST(RR, TT, E, 16, false);                                                                \
    } while (0)
#define IPSX_LAUNCH_FAST_E(RR, TT)                                                                                  \
    do {                                                                                                            \
        if (ept == 1) IPSX_LAUNCH_FAST_C(RR, TT, 1);                                                                \
        else if (ept == 2) IPSX_LAUNCH_FAST_C(RR, TT, 2);                                                           \
        else if (ept == 4) IPSX_LAUNCH_FAST_C(RR, TT, 4);                                                           \
        else IPSX_LAUNCH_FAST_C(RR, TT, 8);                                                                         \
    } while (0)
    if (g_scan_r8 && R == 8 && n_token == 1 && m == cam::M && i == cam::I) {
        // BASELINE configs[3] (8 heads, one token, M = I = 256): the specialised loop (scan_cam_kernel)
        static_assert(cam::LDS_BYTES <= 160 * 1024, "scan_cam_kernel: LDS");
        a.stk_off = cam::OFF_STK;
        const int grid = workgroups > 0 && workgroups < b ? workgroups : b;
#define IPSX_LAUNCH_CAM(S, P)                                                                                       \
    do {                                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_cam_kernel<S, P>),                             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)cam::LDS_BYTES);                 \
        scan_cam_kernel<S, P><<<dim3((unsigned)grid), dim3(cam::NT), cam::LDS_BYTES, as_stream(stream)>>>(a, st);   \
        return launched("scan");                                                                                    \
    } while (0)
        if (st && a.ready) IPSX_LAUNCH_CAM(true, true);
        if (st) IPSX_LAUNCH_CAM(true, false);
        if (a.ready) IPSX_LAUNCH_CAM(false, true);
        IPSX_LAUNCH_CAM(false, false);
#undef IPSX_LAUNCH_CAM
    }
    IPSX_REQUIRE(workgroups <= 0 || workgroups >= b, "scan_persistent_on: fewer workgroups than images only for the shapes of "
                 "ipsx_scan_persistent_groupable");
    // the diagnostic (stamped) build exists for the two benchmark shapes
    if (st && a.ready && R == 8 && n_token == 1 && ept == 4 && lch == 8) {      // stamped persistent loop (diagnostic)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_fast_kernel<8, 1, 4, 8, true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast);
        scan_fast_kernel<8, 1, 4, 8, true, true><<<dim3((unsigned)b), dim3(SCAN_NT), fast, as_stream(stream)>>>(a, st);
        return launched("scan");
    }
    if (st && R == 8 && n_token == 1 && ept == 4 && lch == 8) IPSX_LAUNCH_FAST(8, 1, 4, 8, true);
    if (st && R == 32 && n_token == 4 && ept == 4 && lch == 2) IPSX_LAUNCH_FAST(32, 4, 4, 2, true);
    if (R == 8 && n_token == 1) IPSX_LAUNCH_FAST_E(8, 1);
    IPSX_LAUNCH_FAST_E(32, 4);
#undef IPSX_LAUNCH_FAST_C
#undef IPSX_LAUNCH_FAST_E
#undef IPSX_LAUNCH_FAST
}

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

static size_t topm_large_ws_per_row(int l) { return ((size_t)2 * l * 4 + 255) & ~(size_t)255; }

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
        IPSX_REQUIRE(l <= LARGE_MAX_L, "topm: %d candidates - at most %d are supported", l, LARGE_MAX_L);
        const size_t need = ipsx_topm_workspace_bytes(b, l, m);
        if (!workspace || workspace_bytes < need)
            return fail(IPSX_EWORKSPACE, "topm: %d candidates need a workspace of %zu B (ipsx_topm_workspace_bytes), got %zu",
                        l, need, workspace_bytes);
        const size_t big = large_key_bytes(a.n2) + LARGE_TAIL_BYTES;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topm_large_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)big);
        topm_large_kernel<<<dim3((unsigned)b), dim3(LARGE_NT), big, as_stream(stream)>>>(
            a, static_cast<unsigned char*>(workspace), topm_large_ws_per_row(l));
        return launched("topm");
    }
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    topm_kernel<<<dim3((unsigned)b), dim3(256), lds, as_stream(stream)>>>(a);
    return launched("topm");
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

// Diagnostic entry point (not part of include/ipsx.h): 0 sends the shape of scan_cam_kernel (8 logits per candidate,
// M = I = 256) through scan_fast_kernel instead - tools/scan_compare.py and tools/scan_stamps.py use it.
// Diagnostic: read (and clear) the replay's phase stamps (g_replay_t) into out[10]
extern "C" __attribute__((visibility("default"))) int ipsx_dbg_replay_stamps(unsigned long long* out) {
    unsigned long long zero[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    g_replay_stamps_on = out != nullptr;                                   // (null: off again)
    if (!out) return 0;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ipsx::g_replay_t), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(ipsx::g_replay_t), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("default"))) void ipsx_dbg_scan_direct(int on) { g_scan_direct = on != 0; }
extern "C" __attribute__((visibility("default"))) void ipsx_dbg_scan_r8(int on) { g_scan_r8 = on != 0; }
