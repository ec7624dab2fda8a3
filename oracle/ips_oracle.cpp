// ips_oracle.cpp - CPU restatement of the IPS no-grad hot path of benbergner/ips.
//
// TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the cpu_baseline
// leg of bench.py may load this library; nothing under ips_amd/ does.
//
// It restates, function by function, what the reference computes on this path
// (file:line cited at each function, relative to the reference repo), with the
// arithmetic ORDER pinned down so that the gfx950 kernels can reproduce the same
// bits (see include/ipsx.h "Arithmetic contract"):
//   * contractions are single fp32 fma chains; conv K index is tap-major k = (ky*KW+kx)*C_in + c;
//   * contractions the device runs on the matrix cores (convolutions, the projector Linear,
//     the K and V projections) visit each aligned group of 8 consecutive k in the order
//     0,4,1,5,2,6,3,7 (mfma_order() below): v_mfma_f32_32x32x2_f32 consumes k in pairs
//     (lane half 0, lane half 1) and a lane half fetches 4 CONSECUTIVE k with one 16-byte read;
//     all other chains (q.k, attn.v, fc, MLP, heads) are in ascending index order,
//   * exp is det_expf() below, row sums are wave_sum64() below.
// The reference's own bits (oneDNN / MKL / Sleef on CPU) are not reproducible by
// any GPU kernel; this oracle is pinned against the reference through the golden
// fixtures under tests/golden/ (tools/gen_golden.py imports the reference and
// records its selected indices and outputs; tests/test_oracle_golden.py checks
// index equality and value agreement to 1e-5).
//
// Build: oracle/Makefile (g++ -O3 -mavx2 -mfma -fopenmp -ffp-contract=off).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

#define ORC_API extern "C" __attribute__((visibility("default")))

// ------------------------------------------------------------------ primitives
static inline float as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t as_u32(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// exp() as a fixed sequence of IEEE fp32 operations (Cody-Waite reduction by
// ln2 = C1 + C2, degree-5 Horner polynomial, exponent reconstruction by bits).
// Same source text in ips_amd/csrc/ipsx_math.h; every operation is exactly
// rounded on both sides, so host and device produce identical bits.
static inline float det_expf(float x) {
    if (x != x) return x;
    if (x > 88.72f) return std::numeric_limits<float>::infinity();
    if (x < -104.0f) return 0.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r);
    y = y + 1.0f;
    int ni = (int)n;
    if (ni < -126) {
        y = y * as_float((uint32_t)(ni + 127 + 64) << 23);
        return y * 5.42101086242752217e-20f;  // 2^-64
    }
    if (ni > 127) {
        y = y * as_float((uint32_t)(ni + 126) << 23);
        return y * 2.0f;
    }
    return y * as_float((uint32_t)(ni + 127) << 23);
}

// Sum of x[0..n) in the wavefront order: lane j accumulates x[j], x[j+64], ...
// (ascending, starting from +0), then an xor butterfly with offsets 32,16,...,1.
static inline float wave_sum64(const float* x, int64_t n, int64_t stride = 1) {
    float part[64], nxt[64];
    for (int j = 0; j < 64; ++j) {
        float s = 0.0f;
        for (int64_t i = j; i < n; i += 64) s = s + x[i * stride];
        part[j] = s;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        for (int j = 0; j < 64; ++j) nxt[j] = part[j] + part[j ^ off];
        std::memcpy(part, nxt, sizeof(part));
    }
    return part[0];
}

// chain order of a K-long contraction done on the matrix cores: per group of 8: 0,4,1,5,2,6,3,7
static std::vector<int> mfma_order(int K) {
    std::vector<int> ord;
    ord.reserve(K);
    for (int g = 0; g * 8 < K; ++g)
        for (int j = 0; j < 4; ++j)
            for (int h = 0; h < 2; ++h) {
                const int k = g * 8 + 4 * h + j;
                if (k < K) ord.push_back(k);
            }
    return ord;
}

ORC_API float orc_expf(float x) { return det_expf(x); }
ORC_API float orc_wave_sum64(const float* x, int64_t n) { return wave_sum64(x, n); }
ORC_API int orc_version(void) { return 103; }

// ------------------------------------------------------------------ encoder ops
// nn.BatchNorm2d / BatchNorm1d in eval mode (architecture/ips_net.py:37,58 via
// torchvision BasicBlock; eval forced at ips_net.py:191-193).  ATen's CPU kernel
// evaluates y = x*alpha + beta' with alpha = gamma*invstd, beta' = beta - mean*alpha.
ORC_API void orc_bn_affine(const float* gamma, const float* beta, const float* mean,
                           const float* var, float eps, int c, float* alpha, float* shift) {
    for (int i = 0; i < c; ++i) {
        float invstd = 1.0f / sqrtf(var[i] + eps);
        float a = gamma[i] * invstd;
        float m = mean[i] * a;
        alpha[i] = a;
        shift[i] = beta[i] - m;
    }
}

struct orc_conv {
    int c_in, c_out, kh, kw, stride, pad;
    const float* w;       // OIHW
    const float* alpha;   // c_out or NULL
    const float* shift;   // c_out or NULL
};

static inline int out_dim(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }

// nn.Conv2d(bias=False) -> BatchNorm(eval) [-> += identity] [-> ReLU]
// (torchvision BasicBlock.forward as composed by ips_net.py:35-50).
// acc = fma chain over k = (ky,kx,c) in mfma_order(), zero for padded taps.
ORC_API void orc_conv2d_affine(const orc_conv* cv, const float* x, const float* residual,
                               float* y, int64_t n, int h, int w, int relu) {
    const int ci = cv->c_in, co = cv->c_out, kh = cv->kh, kw = cv->kw;
    const int ho = out_dim(h, kh, cv->stride, cv->pad), wo = out_dim(w, kw, cv->stride, cv->pad);
    const int K = kh * kw * ci;
    const std::vector<int> ord = mfma_order(K);
    // weights transposed to [k][o] so the chain vectorises over output channels
    std::vector<float> wt((size_t)K * co);
    for (int o = 0; o < co; ++o)
        for (int c = 0; c < ci; ++c)
            for (int t = 0; t < kh * kw; ++t)
                wt[(size_t)(t * ci + c) * co + o] = cv->w[((size_t)o * ci + c) * kh * kw + t];
#pragma omp parallel
    {
        std::vector<float> acc(co);
#pragma omp for collapse(2) schedule(static)
        for (int64_t p = 0; p < n; ++p)
            for (int oy = 0; oy < ho; ++oy)
                for (int ox = 0; ox < wo; ++ox) {
                    std::fill(acc.begin(), acc.end(), 0.0f);
                    float* a = acc.data();
                    for (int kk = 0; kk < K; ++kk) {
                        const int k = ord[kk];
                        const int tap = k / ci, c = k - tap * ci;
                        const int ky = tap / kw, kx = tap - ky * kw;
                        const int iy = oy * cv->stride + ky - cv->pad;
                        const int ix = ox * cv->stride + kx - cv->pad;
                        const bool in = iy >= 0 && iy < h && ix >= 0 && ix < w;
                        const float v = in ? x[(((size_t)p * ci + c) * h + iy) * w + ix] : 0.0f;
                        const float* wr = &wt[(size_t)k * co];
                        for (int o = 0; o < co; ++o) a[o] = __builtin_fmaf(v, wr[o], a[o]);
                    }
                    for (int o = 0; o < co; ++o) {
                        float v = a[o];
                        if (cv->alpha) v = __builtin_fmaf(v, cv->alpha[o], cv->shift ? cv->shift[o] : 0.0f);
                        else if (cv->shift) v = v + cv->shift[o];
                        const size_t oi = (((size_t)p * co + o) * ho + oy) * wo + ox;
                        if (residual) v = v + residual[oi];
                        if (relu) v = v > 0.0f ? v : 0.0f;
                        y[oi] = v;
                    }
                }
    }
}

// nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (ips_net.py:39)
ORC_API void orc_maxpool_3x3s2(const float* x, float* y, int64_t n, int c, int h, int w) {
    const int ho = out_dim(h, 3, 2, 1), wo = out_dim(w, 3, 2, 1);
#pragma omp parallel for schedule(static)
    for (int64_t pc = 0; pc < n * c; ++pc)
        for (int oy = 0; oy < ho; ++oy)
            for (int ox = 0; ox < wo; ++ox) {
                float m = -std::numeric_limits<float>::infinity();
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
                        if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
                        const float v = x[((size_t)pc * h + iy) * w + ix];
                        m = (v > m || v != v) ? v : m;
                    }
                y[((size_t)pc * ho + oy) * wo + ox] = m;
            }
}

// nn.AdaptiveAvgPool2d((1,1)) (ips_net.py:50): sequential sum over (y,x), then / (h*w)
ORC_API void orc_avgpool(const float* x, float* y, int64_t n, int c, int hw) {
#pragma omp parallel for schedule(static)
    for (int64_t pc = 0; pc < n * c; ++pc) {
        float s = 0.0f;
        for (int i = 0; i < hw; ++i) s = s + x[(size_t)pc * hw + i];
        y[pc] = s / (float)hw;
    }
}

struct orc_block {
    int n_conv;
    orc_conv conv[3];
    int has_down;
    orc_conv down;
};

struct orc_trunk {
    int c_in, h, w;
    orc_conv stem;
    int n_block;
    const orc_block* blocks;
};

// IPSNet.encoder for images (ips_net.py:17-52): stem, max-pool, residual blocks,
// average pool.  patches (n,c,h,w) -> emb (n, D)
ORC_API void orc_trunk_encode(const orc_trunk* t, const float* patches, int64_t n, float* emb) {
    int h = out_dim(t->h, 7, 2, 3), w = out_dim(t->w, 7, 2, 3);
    int c = t->stem.c_out;
    std::vector<float> a((size_t)n * c * h * w), b, s;
    orc_conv2d_affine(&t->stem, patches, nullptr, a.data(), n, t->h, t->w, 1);
    int hp = out_dim(h, 3, 2, 1), wp = out_dim(w, 3, 2, 1);
    b.resize((size_t)n * c * hp * wp);
    orc_maxpool_3x3s2(a.data(), b.data(), n, c, h, w);
    a.swap(b); h = hp; w = wp;                      // a = current activation (n,c,h,w)
    for (int bi = 0; bi < t->n_block; ++bi) {
        const orc_block& B = t->blocks[bi];
        const float* in = a.data();
        int ih = h, iw = w;
        std::vector<float> cur, nxt;
        const float* src = in;
        int ch = h, cw = w;
        for (int j = 0; j < B.n_conv - 1; ++j) {     // all but the last conv: BN + ReLU
            const orc_conv& cv = B.conv[j];
            int oh = out_dim(ch, cv.kh, cv.stride, cv.pad), ow = out_dim(cw, cv.kw, cv.stride, cv.pad);
            nxt.assign((size_t)n * cv.c_out * oh * ow, 0.0f);
            orc_conv2d_affine(&cv, src, nullptr, nxt.data(), n, ch, cw, 1);
            cur.swap(nxt); src = cur.data(); ch = oh; cw = ow;
        }
        const orc_conv& last = B.conv[B.n_conv - 1];
        int oh = out_dim(ch, last.kh, last.stride, last.pad), ow = out_dim(cw, last.kw, last.stride, last.pad);
        const float* shortcut = in;
        if (B.has_down) {
            s.assign((size_t)n * B.down.c_out * oh * ow, 0.0f);
            orc_conv2d_affine(&B.down, in, nullptr, s.data(), n, ih, iw, 0);
            shortcut = s.data();
        }
        b.assign((size_t)n * last.c_out * oh * ow, 0.0f);
        orc_conv2d_affine(&last, src, shortcut, b.data(), n, ch, cw, 1);  // bn -> += identity -> relu
        a.swap(b); h = oh; w = ow; c = last.c_out;
    }
    orc_avgpool(a.data(), emb, n, c, h * w);
}

// nn.LayerNorm over the last dim (ips_net.py:56 eps 1e-5 no affine;
// transformer.py:69,119 eps 1e-6 affine).  Moments in wave_sum64 order.
static void layernorm_row(const float* x, int d, float eps, const float* g, const float* b, float* y) {
    const float mean = wave_sum64(x, d) / (float)d;
    std::vector<float> sq(d);
    for (int i = 0; i < d; ++i) { float c = x[i] - mean; sq[i] = c * c; }
    const float var = wave_sum64(sq.data(), d) / (float)d;
    const float rstd = 1.0f / sqrtf(var + eps);
    for (int i = 0; i < d; ++i) {
        float v = (x[i] - mean) * rstd;
        if (g) v = __builtin_fmaf(v, g[i], b[i]);
        y[i] = v;
    }
}

ORC_API void orc_layernorm(const float* x, int64_t n, int d, float eps, const float* g,
                           const float* b, float* y) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) layernorm_row(x + r * d, d, eps, g, b, y + r * d);
}

// y[r][o] = (fma chain over c of x[r][c]*w[o][c]) [+ bias[o]]   (nn.Linear)
// mfma != 0: chain in mfma_order() (the device runs this Linear on the matrix cores)
static void linear_impl(const float* x, const float* w, const float* bias, int64_t n, int d_in,
                        int d_out, float* y, int mfma) {
    std::vector<float> wt((size_t)d_in * d_out);
    const std::vector<int> ord = mfma_order(d_in);
    for (int o = 0; o < d_out; ++o)
        for (int c = 0; c < d_in; ++c) wt[(size_t)c * d_out + o] = w[(size_t)o * d_in + c];
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        float* a = y + r * d_out;
        for (int o = 0; o < d_out; ++o) a[o] = 0.0f;
        for (int cc = 0; cc < d_in; ++cc) {
            const int c = mfma ? ord[cc] : cc;
            const float v = x[r * d_in + c];
            const float* wr = &wt[(size_t)c * d_out];
            for (int o = 0; o < d_out; ++o) a[o] = __builtin_fmaf(v, wr[o], a[o]);
        }
        if (bias) for (int o = 0; o < d_out; ++o) a[o] = a[o] + bias[o];
    }
}

ORC_API void orc_linear(const float* x, const float* w, const float* bias, int64_t n, int d_in,
                        int d_out, float* y) {
    linear_impl(x, w, bias, n, d_in, d_out, y, 0);
}

// The LayerNorm moments of a feature row as the device takes them OFF THE OPERAND STREAM of the projector's GEMM
// (round 5; ips_amd/csrc/ipsx_rowstats.h row_moments_*): on the matrix cores lane (row, half h) holds the four
// consecutive k = 8g + 4h + j (j = 0..3) of k-group g, so the sums run as EIGHT chains per row - chain (h, j) adds
// x[8g + 4h + j] for g ascending (sum: fp32 add from +0; sum of squares: one fma per element) - folded as
// ((c0 + c1) + (c2 + c3)) per half and half 0 + half 1.  mean = sum / F, var = E[x^2] - mean^2 (one fma, clamped at 0),
// rstd = 1 / sqrt(var + eps).  F is a multiple of 8.
// Round 6 (advisor, round 5): E[x^2] - mean^2 loses (mean / std)^2 * 2^-24 of the variance, and the folded form
// acc - mean * colsum cancels the same way.  A row whose one-pass variance is below 1/17 of its E[x^2] (mean^2 / var > 16:
// |mean| > 4 std) is CENTRED, as nn.LayerNorm itself does (ips_net.py:56): over d = x - mean the same eight chains sum d
// and d * d (sd = sd + d; q = fma(d, d, q)), folded the same way; mean' = mean + E[d] (a constant row's residual is exact:
// mean' is the constant), var = fma(-E[d], E[d], E[d^2]) clamped at 0; the Linear then runs on x - mean' (orc_projector).
// The statistics say so in the sign: (mean', -rstd).  Rows of well-conditioned features (every fixture) never take that
// path: their bits are unchanged.  Device: ipsx_rowstats.h rm_finish / rm_recentred.
#define ORC_RM_RECENTRE 17.0f
static void projector_moments(const float* x, int f, float eps, float* mean_out, float* rstd_out) {
    float t[2], u[2];
    for (int h = 0; h < 2; ++h) {
        float s[4] = {0.0f, 0.0f, 0.0f, 0.0f}, q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int g = 0; g * 8 < f; ++g)
            for (int j = 0; j < 4; ++j) {
                const float v = x[g * 8 + 4 * h + j];
                s[j] = s[j] + v;
                q[j] = __builtin_fmaf(v, v, q[j]);
            }
        t[h] = (s[0] + s[1]) + (s[2] + s[3]);
        u[h] = (q[0] + q[1]) + (q[2] + q[3]);
    }
    const float sum = t[0] + t[1], sumsq = u[0] + u[1];
    float mean = sum / (float)f;
    const float ex2 = sumsq / (float)f;
    float var = __builtin_fmaf(-mean, mean, ex2);
    var = var > 0.0f ? var : 0.0f;
    const bool centred = var * ORC_RM_RECENTRE < ex2;
    if (centred) {
        for (int h = 0; h < 2; ++h) {
            float sd[4] = {0.0f, 0.0f, 0.0f, 0.0f}, q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int g = 0; g * 8 < f; ++g)
                for (int j = 0; j < 4; ++j) {
                    const float d = x[g * 8 + 4 * h + j] - mean;
                    sd[j] = sd[j] + d;
                    q[j] = __builtin_fmaf(d, d, q[j]);
                }
            t[h] = (sd[0] + sd[1]) + (sd[2] + sd[3]);
            u[h] = (q[0] + q[1]) + (q[2] + q[3]);
        }
        const float dm = (t[0] + t[1]) / (float)f;
        var = __builtin_fmaf(-dm, dm, (u[0] + u[1]) / (float)f);
        var = var > 0.0f ? var : 0.0f;
        mean = mean + dm;
    }
    *mean_out = mean;
    *rstd_out = 1.0f / sqrtf(var + eps);
    if (centred) *rstd_out = -*rstd_out;             // (mean', -rstd): a CENTRED row - its Linear runs on x - mean' (orc_projector)
}

ORC_API void orc_projector_moments(const float* x, int64_t n, int f, float eps, float* stats /* (n, 2): mean, rstd */) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) projector_moments(x + r * f, f, eps, stats + 2 * r, stats + 2 * r + 1);
}

// column sums of a Linear's weights, cs[o] = sum_c w[o][c]: ascending in float64, rounded to fp32 once
// (device: weight_colsum_kernel, ips_amd/csrc/aggregate.hip - IEEE double adds in the same order)
ORC_API void orc_weight_colsum(const float* w, int d_out, int d_in, float* cs) {
    for (int o = 0; o < d_out; ++o) {
        double s = 0.0;
        for (int c = 0; c < d_in; ++c) s = s + (double)w[(size_t)o * d_in + c];
        cs[o] = (float)s;
    }
}

// IPSNet.encoder for features (ips_net.py:54-60): LN(eps 1e-5, no affine) -> Linear(F,D)+bias -> BatchNorm1d eval ->
// ReLU.  Round 5: the LayerNorm is FOLDED into the Linear's epilogue - Linear(LN(x)) = rstd * (x W^T - mean * colsum(W))
// + b, exact algebra - so that the device reads every feature row ONCE (the moments come off the GEMM's own operand
// registers: no separate pass over x, and the normalised row never exists):
//     acc[o] = fma chain over c (mfma_order) of x[c] * w[o][c]          the RAW row on the matrix cores
//     t      = fma(-mean, cs[o], acc[o]);   u = t * rstd
//     y[o]   = relu(fma(u, alpha[o], fma(bias[o], alpha[o], shift[o])))  (bias through the BatchNorm affine, as before)
// This ordering IS the arithmetic contract of the projector (device: projector_stream_kernel, conv_nhwc_kernel<NORM>);
// against the reference it is a rounding-level difference, pinned like everything else by the fixtures (indices
// identical, values within 1e-4).
ORC_API void orc_projector(const float* x, int64_t n, int f, int d, float ln_eps, const float* w,
                           const float* bias, const float* alpha, const float* shift, float* out) {
    std::vector<float> cs(d), st((size_t)n * 2);
    orc_weight_colsum(w, d, f, cs.data());
    orc_projector_moments(x, n, f, ln_eps, st.data());
    // rows marked centred (mean^2 / var > 16) go through the matrix cores' chain as x - mean, t = acc; all others raw
    std::vector<float> xc;
    const float* xin = x;
    for (int64_t r = 0; r < n; ++r)
        if (st[2 * r + 1] < 0.0f) {
            if (xc.empty()) { xc.assign(x, x + (size_t)n * f); xin = xc.data(); }
            for (int c = 0; c < f; ++c) xc[(size_t)r * f + c] = x[(size_t)r * f + c] - st[2 * r];
        }
    linear_impl(xin, w, nullptr, n, f, d, out, 1);                // on the matrix cores
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        const bool centred = st[2 * r + 1] < 0.0f;
        const float nm = -st[2 * r], rstd = fabsf(st[2 * r + 1]);
        for (int o = 0; o < d; ++o) {
            const float sh = __builtin_fmaf(bias[o], alpha[o], shift[o]);
            const float t = centred ? out[r * d + o] : __builtin_fmaf(nm, cs[o], out[r * d + o]);
            const float u = t * rstd;
            const float v = __builtin_fmaf(u, alpha[o], sh);
            out[r * d + o] = v > 0.0f ? v : 0.0f;
        }
    }
}

// ------------------------------------------------------------------ scorer
// q_w(self.q) and the "/ temperature" of compute_attn (transformer.py:76, :31)
ORC_API void orc_query_proj(const float* q, const float* wq, float temperature, int n_token, int d,
                            int hdk, float* qs) {
    orc_linear(q, wq, nullptr, n_token, d, hdk, qs);
    for (int i = 0; i < n_token * hdk; ++i) qs[i] = qs[i] / temperature;
}

// Attention logits of one patch (transformer.py:77,31: matmul(q / temperature, k_w(x)^T), before the softmax):
//     logit[h*T + t] = sum_j qs[t][h,j] * sum_c W_k[h*dk + j][c] * x[c]          x = emb (+ pos)
// The reference evaluates the inner sum first (a D x H*Dk projection per patch).  The two sums commute, and the hot
// path evaluates them the other way round - exact algebra, 1/Dk of the work:
//     V[h*T + t][c] = sum_j qs[t][h,j] * W_k[h*dk + j][c]      (orc_fold_query: once per call, j ascending)
//     logit[r]      = sum_c x[c] * V[r][c]                      (on the matrix cores: the contract's k order)
// This ordering IS the arithmetic contract of the logits (device: fold_query_kernel + logits_kernel); against the
// reference it is a rounding-level difference, pinned like everything else by the fixtures (indices identical,
// scores within 1e-4).
ORC_API void orc_fold_query(const float* qs, const float* wk, int h, int dk, int T, int d, float* v /* (h*T, d) */) {
    const int hdk = h * dk;
    for (int hh = 0; hh < h; ++hh)
        for (int t = 0; t < T; ++t)
            for (int c = 0; c < d; ++c) {
                float a = 0.0f;
                for (int j = 0; j < dk; ++j)
                    a = __builtin_fmaf(qs[(size_t)t * hdk + hh * dk + j], wk[(size_t)(hh * dk + j) * d + c], a);
                v[(size_t)(hh * T + t) * d + c] = a;
            }
}

ORC_API void orc_logits(const float* emb, const float* pos, const float* wk, const float* qs,
                        int64_t n, int d, int h, int dk, int T, float* logits) {
    const int R = h * T;
    std::vector<float> v((size_t)R * d);
    orc_fold_query(qs, wk, h, dk, T, d, v.data());
    if (!pos) {
        linear_impl(emb, v.data(), nullptr, n, d, R, logits, 1);
        return;
    }
    std::vector<float> x((size_t)n * d);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n * (int64_t)d; ++i) x[i] = emb[i] + pos[i];
    linear_impl(x.data(), v.data(), nullptr, n, d, R, logits, 1);
}

// softmax over candidates per (h,t) row, mean over heads, mean over tokens
// (transformer.py:32 softmax(dim=-1); :148 attn.mean(dim=1).transpose(1,2).mean(-1)).
// lg: (L, h*T) logits of the candidates; scores (L); attn (h,T,L) optional.
ORC_API void orc_scores_from_logits(const float* lg, int L, int h, int T, float* scores, float* attn) {
    const int R = h * T;
    std::vector<float> mx(R), den(R), e(L);
    for (int r = 0; r < R; ++r) {
        float m = -std::numeric_limits<float>::infinity();
        for (int l = 0; l < L; ++l) { float v = lg[(size_t)l * R + r]; m = (v > m || v != v) ? v : m; }
        for (int l = 0; l < L; ++l) e[l] = det_expf(lg[(size_t)l * R + r] - m);
        mx[r] = m;
        // the weights are e * (1 / den): ONE IEEE division per (head, token) row and a multiplication per candidate
        // (round 5; it was a division per candidate - eight per candidate and iteration at CAMELYON, a tenth of the
        // device loop's instructions).  Against a true quotient the product is off by at most an ulp: a rounding-level
        // decision like the summation orders, pinned by the fixtures.
        den[r] = 1.0f / wave_sum64(e.data(), L);
    }
    for (int l = 0; l < L; ++l) {
        float st = 0.0f;
        for (int t = 0; t < T; ++t) {
            float sh = 0.0f;
            for (int hh = 0; hh < h; ++hh) {
                const int r = hh * T + t;
                const float a = det_expf(lg[(size_t)l * R + r] - mx[r]) * den[r];
                if (attn) attn[((size_t)hh * T + t) * L + l] = a;
                sh = sh + a;
            }
            st = st + sh / (float)h;
        }
        scores[l] = st / (float)T;
    }
}

// Transformer.get_scores(x) for x (L, d) (transformer.py:143-148)
ORC_API void orc_scores(const float* x, const float* wk, const float* qs, int L, int d, int h, int dk,
                        int T, float* scores, float* attn) {
    std::vector<float> lg((size_t)L * h * T);
    orc_logits(x, nullptr, wk, qs, L, d, h, dk, T, lg.data());
    orc_scores_from_logits(lg.data(), L, h, T, scores, attn);
}

// key used for ranking: score descending, NaN first (ATen TopKImpl comparator),
// ties broken by the earlier candidate position
static inline uint64_t rank_key(float s, uint32_t pos) {
    uint32_t u;
    if (s != s) u = 0xFFFFFFFFu;
    else {
        s = s + 0.0f;                       // -0 -> +0
        u = as_u32(s);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    return ((uint64_t)u << 32) | (uint64_t)(0xFFFFFFFFu - pos);
}

// torch.topk(scores, M)[1] for one row (ips_net.py:148), canonical tie rule.
// tie = 1 when the M-th and (M+1)-th ranked scores are bit-equal.
ORC_API void orc_topm(const float* scores, int L, int M, int64_t* top, int32_t* tie) {
    std::vector<uint64_t> key(L);
    for (int l = 0; l < L; ++l) key[l] = rank_key(scores[l], (uint32_t)l);
    std::sort(key.begin(), key.end(), std::greater<uint64_t>());
    for (int j = 0; j < M; ++j) top[j] = (int64_t)(0xFFFFFFFFu - (uint32_t)(key[j] & 0xFFFFFFFFu));
    if (tie) *tie = (M < L && (key[M - 1] >> 32) == (key[M] >> 32)) ? 1 : 0;
}

// torch.topk on CPU as ATen implements it (ATen/native/TopKImpl.h topk_impl_loop,
// largest=true, sorted=true): partial_sort when k*64 <= n, else nth_element + sort of
// the first k-1, on (value,index) pairs with comparator (nan first) x > y.  With
// libstdc++ this reproduces torch's order under exact ties.
ORC_API void orc_topm_aten(const float* scores, int L, int M, int64_t* top) {
    using P = std::pair<float, int64_t>;
    std::vector<P> q(L);
    for (int l = 0; l < L; ++l) q[l] = P(scores[l], l);
    auto gt = [](const P& x, const P& y) {
        return ((x.first != x.first) && !(y.first != y.first)) || (x.first > y.first);
    };
    if ((int64_t)M * 64 <= L) {
        std::partial_sort(q.begin(), q.begin() + M, q.end(), gt);
    } else {
        std::nth_element(q.begin(), q.begin() + M - 1, q.end(), gt);
        std::sort(q.begin(), q.begin() + M - 1, gt);
    }
    for (int j = 0; j < M; ++j) top[j] = q[j].second;
}

// The tie rule of the selection LOOP (round 5).  torch.topk's order differs from the canonical one only where scores are
// bit-equal, and the device replays libstdc++'s routines to follow it - 100+ us per iteration at 10,000 candidates.  Two
// kinds of equal scores exist: candidates whose attention logits are bit-IDENTICAL rows (duplicated patches - blank
// Megapixel-MNIST patches without positional encoding: they tie in the reference's own arithmetic too, and its order is
// torch's), and DIFFERENT rows whose scores happen to collide in the last bit of THIS arithmetic (in the reference's
// oneDNN / Sleef arithmetic those two are an ulp apart and other pairs collide: on the CAMELYON bench slide 18 of 255
// iterations here, none there - replaying torch on them reproduces nothing of the reference).  So: the canonical order,
// unless two candidates of ONE RUN of equal scores that reaches into the first M + 1 canonical ranks have bit-identical logit rows - then
// torch.topk's order on the whole array, as before.  lg: (L, R) logits of the candidates, row-major.
ORC_API void orc_topm_loop(const float* scores, const float* lg, int L, int R, int M, int64_t* top, int32_t* tie) {
    std::vector<uint64_t> key(L);
    for (int l = 0; l < L; ++l) key[l] = rank_key(scores[l], (uint32_t)l);
    std::sort(key.begin(), key.end(), std::greater<uint64_t>());
    if (tie) *tie = (M < L && (key[M - 1] >> 32) == (key[M] >> 32)) ? 1 : 0;
    // (round 6, advisor: not only NEIGHBOURS - in a run A, B, C of equal scores with A and C duplicates and B a different row
    //  that collides with them, no neighbouring pair is identical: every member j of the first M ranks (the selected ones) is held against
    //  EVERY later member of its run of equal scores, also beyond rank M - a run across the boundary decides who stays)
    bool structural = false;
    const int n = M < L - 1 ? M : L - 1;
    for (int j = 0; j < n && !structural; ++j)
        for (int k = j + 1; k < L && !structural && (key[k] >> 32) == (key[j] >> 32); ++k) {
            const uint32_t a = 0xFFFFFFFFu - (uint32_t)(key[j] & 0xFFFFFFFFu), b = 0xFFFFFFFFu - (uint32_t)(key[k] & 0xFFFFFFFFu);
            structural = std::memcmp(lg + (size_t)a * R, lg + (size_t)b * R, (size_t)R * sizeof(float)) == 0;
        }
    if (structural) orc_topm_aten(scores, L, M, top);
    else for (int j = 0; j < M; ++j) top[j] = (int64_t)(0xFFFFFFFFu - (uint32_t)(key[j] & 0xFFFFFFFFu));
}

// The loop of IPSNet.ips (ips_net.py:206-241) for ONE image, on embeddings:
//   memory = first M patches; for each chunk of I: candidates = memory ++ chunk
//   (:231-232), emb_pos = emb + gather(pos, idx) (:235-236), scores (:145), top-M
//   (:148), gather memory (:152-153).
// The K projection is recomputed for every candidate in every iteration exactly
// as the reference does.  emb (N,d); pos (N,d) or NULL; out: mem_idx (M) final,
// trace_idx (n_iter,M) per-iteration memory or NULL, trace_score (n_iter,M) or
// NULL, tie (n_iter) or NULL.  aten_ties: 0 the canonical order, 1 the loop's rule (orc_topm_loop: torch.topk's order
// where bit-identical candidates tie), 2 torch.topk's order wherever scores tie (orc_topm_aten).
ORC_API void orc_ips_scan(const float* emb, const float* pos, const float* wk, const float* qs,
                          int64_t N, int d, int h, int dk, int T, int M, int I, int aten_ties,
                          int64_t* mem_idx, int64_t* trace_idx, float* trace_score, int32_t* tie) {
    std::vector<int64_t> mem(M), cand, top(M);
    for (int j = 0; j < M; ++j) mem[j] = j;
    const int n_iter = (int)((N - M + I - 1) / I);
    std::vector<float> x, sc, lgv;
    for (int it = 0; it < n_iter; ++it) {
        const int64_t lo = (int64_t)it * I + M, hi = std::min<int64_t>(lo + I, N);
        cand.assign(mem.begin(), mem.end());
        for (int64_t p = lo; p < hi; ++p) cand.push_back(p);
        const int L = (int)cand.size();
        x.resize((size_t)L * d);
        for (int l = 0; l < L; ++l)
            for (int c = 0; c < d; ++c) {
                const float e = emb[cand[l] * d + c];
                x[(size_t)l * d + c] = pos ? e + pos[cand[l] * d + c] : e;
            }
        sc.resize(L);
        lgv.resize((size_t)L * h * T);
        orc_logits(x.data(), nullptr, wk, qs, L, d, h, dk, T, lgv.data());          // (= orc_scores, with the logits kept)
        orc_scores_from_logits(lgv.data(), L, h, T, sc.data(), nullptr);
        int32_t tf = 0;
        if (aten_ties == 1) orc_topm_loop(sc.data(), lgv.data(), L, h * T, M, top.data(), nullptr);
        else if (aten_ties) orc_topm_aten(sc.data(), L, M, top.data());
        else orc_topm(sc.data(), L, M, top.data(), &tf);
        if (tie) tie[it] = tf;
        for (int j = 0; j < M; ++j) {
            mem[j] = cand[top[j]];
            if (trace_idx) trace_idx[(size_t)it * M + j] = mem[j];
            if (trace_score) trace_score[(size_t)it * M + j] = sc[top[j]];
        }
    }
    for (int j = 0; j < M; ++j) mem_idx[j] = mem[j];
}

// ------------------------------------------------------------------ aggregation
struct orc_transf {
    int n_token, h, d, dk, dv, d_inner;
    const float *q, *wq, *wk, *wv, *fc, *ln1_g, *ln1_b, *w1, *b1, *w2, *b2, *ln2_g, *ln2_b;
    float temperature, ln_eps;
};

// Transformer.forward for one image: x (M,d) -> out (T,d)
// (transformer.py:85-109 cross-attention with residual on the learned queries and
// LayerNorm; :122-132 MLP with residual and LayerNorm; dropout = identity in eval)
ORC_API void orc_aggregate(const orc_transf* t, const float* x, int M, float* out) {
    const int T = t->n_token, h = t->h, d = t->d, dk = t->dk, dv = t->dv;
    const int hdk = h * dk, hdv = h * dv;
    std::vector<float> qs((size_t)T * hdk), lg((size_t)M * h * T), sc(M), attn((size_t)h * T * M);
    orc_query_proj(t->q, t->wq, t->temperature, T, d, hdk, qs.data());
    orc_logits(x, nullptr, t->wk, qs.data(), M, d, h, dk, T, lg.data());
    orc_scores_from_logits(lg.data(), M, h, T, sc.data(), attn.data());
    std::vector<float> v((size_t)M * hdv), ctx((size_t)T * hdv), y((size_t)T * d), z((size_t)T * d);
    linear_impl(x, t->wv, nullptr, M, d, hdv, v.data(), 1);       // V projection: on the matrix cores
    for (int tt = 0; tt < T; ++tt)
        for (int hh = 0; hh < h; ++hh)
            for (int j = 0; j < dv; ++j) {
                float a = 0.0f;
                for (int l = 0; l < M; ++l)
                    a = __builtin_fmaf(attn[((size_t)hh * T + tt) * M + l], v[(size_t)l * hdv + hh * dv + j], a);
                ctx[(size_t)tt * hdv + hh * dv + j] = a;
            }
    orc_linear(ctx.data(), t->fc, nullptr, T, hdv, d, y.data());
    for (int i = 0; i < T * d; ++i) y[i] = y[i] + t->q[i];
    orc_layernorm(y.data(), T, d, t->ln_eps, t->ln1_g, t->ln1_b, z.data());
    std::vector<float> hid((size_t)T * t->d_inner);
    orc_linear(z.data(), t->w1, t->b1, T, d, t->d_inner, hid.data());
    for (auto& u : hid) u = u > 0.0f ? u : 0.0f;
    orc_linear(hid.data(), t->w2, t->b2, T, t->d_inner, d, y.data());
    for (int i = 0; i < T * d; ++i) y[i] = y[i] + z[i];
    orc_layernorm(y.data(), T, d, t->ln_eps, t->ln2_g, t->ln2_b, out);
}

// one task head (ips_net.py:72-81): Linear(D, n_class) -> softmax | sigmoid
ORC_API void orc_head(const float* emb_token, int d, const float* w, const float* bias, int n_class,
                      int act, float* out) {
    std::vector<float> z(n_class);
    orc_linear(emb_token, w, bias, 1, d, n_class, z.data());
    if (act == 0) {
        float m = -std::numeric_limits<float>::infinity();
        for (int c = 0; c < n_class; ++c) m = z[c] > m ? z[c] : m;
        for (int c = 0; c < n_class; ++c) z[c] = det_expf(z[c] - m);
        const float den = wave_sum64(z.data(), n_class);
        for (int c = 0; c < n_class; ++c) out[c] = z[c] / den;
    } else {
        for (int c = 0; c < n_class; ++c) out[c] = 1.0f / (1.0f + det_expf(-z[c]));
    }
}
