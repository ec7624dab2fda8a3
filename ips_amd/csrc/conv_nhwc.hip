// conv_nhwc.hip - implicit-GEMM convolution on channels-last activations: the layer-by-layer
// encoder path for every trunk the fused kernel does not cover (other patch sizes, ResNet-18 with
// four stages, ResNet-50 bottlenecks) and the Linear layers that run on the matrix cores (projector,
// V projection).
//
// Same arithmetic as conv.hip / the oracle (one fp32 fma chain per output in the contract's k
// order).  What differs is the data path, shaped by what tools/ubench measured for the fp32 MFMA
// (every non-MFMA instruction costs matrix-pipe time):
//   * activations are pixel-major, x[pixel][C_in]: the 4 consecutive k a lane half needs per
//     k-group are ONE 16-byte load;
//   * operands are fetched with raw buffer loads: the per-lane pixel offset lives in a VGPR that
//     changes once per tap, the channel / k-group offset is a scalar register, so a stage issues
//     4 loads + 16 MFMAs and no address arithmetic on the vector pipe;
//   * halo (padding) lanes carry an out-of-range offset: the buffer's hardware bounds check
//     returns zeros - no select, no zero page;
//   * 4-slot register ring, operands requested three stages (3 x 1024 matrix-pipe cycles) ahead.
// Wave tile 64(M) x 64(N) = 2x2 accumulators; the 4 waves of a workgroup are arranged WM x WN
// (along N for wide layers so the activation rows are fetched once per workgroup).
//
// Algorithmic cost: 2*K*C_out flop per output pixel; bytes: activations in + out once.

#include <algorithm>

#include "ipsx_common.h"
#include "ipsx_math.h"
#include "ipsx_rowstats.h"

namespace ipsx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)

constexpr unsigned kOob = 0x80000000u;      // voffset of a padding lane: beyond any buffer we bind

struct NhwcArgs {
    const float* x;
    const float* wp;
    const float* alpha;
    const float* shift;
    const float* res;
    float* y;
    unsigned m_total;      // n * ho * wo
    unsigned x_bytes, w_bytes;
    int c_in, h, w, c_out, ho, wo, kh, kw, stride, pad, relu;
    int spt;               // stages (k-groups) per tap = c_in / 8
    int kgs;               // packed k-groups per n-tile = kh*kw*c_in / 8
    const float* stats;    // NORM kernels: (mean, rstd) per input row (1x1 convolutions = Linear layers only)
    const float* colsum;   // NORM kernels: column sums of the weights, one per output channel (ipsx_weight_colsum)
    int* ready;            // NORM kernels, optional: *ready = ready_value by the first thread (see ipsx_projector_apply_publish)
    int ready_value;
};

template <int NTW>
struct NhwcStage {
    f32x4 a0, a1, b[NTW];
};

__device__ __forceinline__ f32x4 bufload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

template <int NTW>
__device__ __forceinline__ void nhwc_mma(const NhwcStage<NTW>& st, f32x16 (&acc)[2][NTW]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[0][t] = MFMA(st.a0[j], st.b[t][j], acc[0][t]);
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[1][t] = MFMA(st.a1[j], st.b[t][j], acc[1][t]);
    }
}

struct NhwcPixel {
    int iy0, ix0;          // top-left input coordinate of the receptive field
    unsigned img_pix;      // img * h * w
    bool valid;
};

__device__ __forceinline__ NhwcPixel nhwc_pixel(const NhwcArgs& a, unsigned m) {
    NhwcPixel p;
    p.valid = m < a.m_total;
    const unsigned howo = (unsigned)(a.ho * a.wo);
    const unsigned mm = p.valid ? m : 0u;
    const unsigned img = mm / howo, pix = mm - img * howo;
    const unsigned oy = pix / (unsigned)a.wo, ox = pix - oy * (unsigned)a.wo;
    p.iy0 = (int)oy * a.stride - a.pad;
    p.ix0 = (int)ox * a.stride - a.pad;
    p.img_pix = img * (unsigned)(a.h * a.w);
    return p;
}

// byte offset of this lane's source pixel row (+ its half's 16 bytes) for a tap, or kOob
__device__ __forceinline__ unsigned nhwc_voff(const NhwcArgs& a, const NhwcPixel& p, int tap, int half) {
    const int ky = tap / a.kw, kx = tap - ky * a.kw;
    const int iy = p.iy0 + ky, ix = p.ix0 + kx;
    const bool ok = p.valid && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
    return ok ? ((p.img_pix + (unsigned)(iy * a.w + ix)) * (unsigned)a.c_in + 4u * half) * 4u : kOob;
}

// NORM: a Linear behind a LayerNorm without affine (the projector, reference architecture/ips_net.py:54-60) with the
// LayerNorm FOLDED into the epilogue - Linear(LN(x)) = rstd * (x W^T - mean * colsum(W)) + b, exact algebra (round 5; the
// contract: oracle/ips_oracle.cpp orc_projector): the RAW rows run through the matrix pipe, the epilogue applies
// t = fma(-mean, cs[n], acc), u = t * rstd with the row moments of row_moments_kernel (aggregate.hip) - no arithmetic on
// the operands between load and MFMA, and the normalised row never exists.  1x1 convolutions only.
// NTW: n-tiles (32 output channels each) per wave - 2 (wave tile 64 x 64) or 4 (64 x 128: half the activation loads
// and half the NORM arithmetic per MFMA, and with 4 waves along N a workgroup covers 512 output channels, so very
// wide layers read every activation row once); 8 * NTW MFMAs and 2 + NTW loads per stage.
template <int WM, int WN, bool NORM, int NTW>
__global__ __launch_bounds__(256, 2) void conv_nhwc_kernel(NhwcArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    const int wm = __builtin_amdgcn_readfirstlane(wave % WM), wn = __builtin_amdgcn_readfirstlane(wave / WM);
    if (NORM && a.ready && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        __hip_atomic_store(a.ready, a.ready_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned m_base = (blockIdx.x * WM + wm) * 64u;
    const int nt0 = (blockIdx.y * WN + wn) * NTW;                   // first of this wave's n-tiles
    if (m_base >= a.m_total || nt0 * 32 >= a.c_out) return;         // wave-uniform
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, (int)a.w_bytes, 0x00020000);
    const NhwcPixel p0 = nhwc_pixel(a, m_base + (lane & 31));
    const NhwcPixel p1 = nhwc_pixel(a, m_base + 32 + (lane & 31));
    const unsigned lb = lane * 16u;
    unsigned wb[NTW];                                               // byte offset of every n-tile's weight stream
#pragma unroll                                                      // (a tile beyond C_out re-reads the last real one;
    for (int t = 0; t < NTW; ++t)                                   //  its accumulators are never stored)
        wb[t] = (unsigned)min(nt0 + t, (a.c_out + 31) / 32 - 1) * (unsigned)a.kgs * 1024u;
    const int taps = a.kh * a.kw, total = taps * a.spt;

    f32x16 acc[2][NTW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // prefetch stream state: stage gp = (tap pt, k-group pc); pv0/pv1 = pixel offsets of tap pt
    int gp = 0, pt = 0, pc = 0;
    unsigned pv0 = nhwc_voff(a, p0, 0, half), pv1 = nhwc_voff(a, p1, 0, half);
    NhwcStage<NTW> s0, s1, s2, s3;       // (s3: ring of 4 only)
#define NHWC_ISSUE(S)                                                       \
    do {                                                                    \
        const unsigned ca = (unsigned)pc * 32u, cb = (unsigned)gp * 1024u;  \
        S.a0 = bufload(rx, pv0, ca);                                        \
        S.a1 = bufload(rx, pv1, ca);                                        \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t) S.b[t] = bufload(rw, lb, wb[t] + cb); \
    } while (0)
#define NHWC_ADVANCE()                                                      \
    do {                                                                    \
        if (gp + 1 < total) {                                               \
            ++gp;                                                           \
            if (++pc == a.spt) {                                            \
                pc = 0; ++pt;                                               \
                pv0 = nhwc_voff(a, p0, pt, half);                           \
                pv1 = nhwc_voff(a, p1, pt, half);                           \
            }                                                               \
        }                                                                   \
    } while (0)
// the loads of the stage three ahead are spread between the MFMAs of this one (same finding as in fused_trunk.hip:
// a few MFMAs between two loads); the stream state advances after the stage so that loads and MFMAs share a basic block
#define SGB_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define SGB_LOAD() __builtin_amdgcn_sched_group_barrier(0x020, 1, 0)
#define NHWC_STAGE(SL, SM)                                                                  \
    NHWC_ISSUE(SL);                                                                         \
    nhwc_mma<NTW>(SM, acc);                                                                 \
    if (NTW == 2) {                                                                         \
        SGB_MFMA(3); SGB_LOAD(); SGB_MFMA(3); SGB_LOAD(); SGB_MFMA(3); SGB_LOAD(); SGB_MFMA(3); SGB_LOAD(); SGB_MFMA(4); \
    } else {                                                                                \
        SGB_MFMA(5); SGB_LOAD(); SGB_MFMA(5); SGB_LOAD(); SGB_MFMA(5); SGB_LOAD();          \
        SGB_MFMA(5); SGB_LOAD(); SGB_MFMA(5); SGB_LOAD(); SGB_MFMA(5); SGB_LOAD(); SGB_MFMA(2); \
    }                                                                                       \
    SB();                                                                                   \
    NHWC_ADVANCE();
    if (NTW == 2) {
        // operands are requested THREE stages ahead (ring of 4: the slot being refilled is the one consumed a stage ago)
        NHWC_ISSUE(s0); NHWC_ADVANCE();
        NHWC_ISSUE(s1); NHWC_ADVANCE();
        NHWC_ISSUE(s2); NHWC_ADVANCE();
#pragma unroll 1
        for (int g = 0; g < total; g += 4) {       // total is a multiple of 4 (C_in % 32 == 0)
            NHWC_STAGE(s3, s0)
            NHWC_STAGE(s0, s1)
            NHWC_STAGE(s1, s2)
            NHWC_STAGE(s2, s3)
        }
    } else {
        // 64 x 128 wave tile: a stage is 32 MFMAs (2048 matrix-pipe cycles), so TWO stages ahead cover more time than
        // three did above - and a ring of 3 keeps the kernel at 2 waves per SIMD (128 accumulator registers + 3 x 24
        // operand registers).  total need not be a multiple of 3: the stages past the end are skipped (uniform branch).
        NHWC_ISSUE(s0); NHWC_ADVANCE();
        NHWC_ISSUE(s1); NHWC_ADVANCE();
#pragma unroll 1
        for (int g = 0; g < total; g += 3) {
            NHWC_STAGE(s2, s0)
            if (g + 1 < total) { NHWC_STAGE(s0, s1) }
            if (g + 2 < total) { NHWC_STAGE(s1, s2) }
        }
    }
#undef NHWC_STAGE
#undef SGB_MFMA
#undef SGB_LOAD
#undef NHWC_ADVANCE
#undef NHWC_ISSUE

    // epilogue: (the folded LayerNorm,) BatchNorm affine, residual, ReLU; lanes of a store are 32 consecutive channels
    const int i = lane & 31;
    float nmean[NORM ? 2 : 1][NORM ? 16 : 1], rstd[NORM ? 2 : 1][NORM ? 16 : 1];
    if (NORM) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned m = min(m_base + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.m_total - 1);
                const float2 st = reinterpret_cast<const float2*>(a.stats)[m];
                nmean[NORM ? mt : 0][NORM ? r : 0] = st.y < 0.0f ? 0.0f : -st.x;      // (mean, -rstd): a centred row, below
                rstd[NORM ? mt : 0][NORM ? r : 0] = __builtin_fabsf(st.y);
            }
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int n = (nt0 + nt) * 32 + i;
        if (n >= a.c_out) continue;
        const float al = a.alpha ? a.alpha[n] : 1.0f;
        const float sh = a.shift ? a.shift[n] : 0.0f;
        const float cs = NORM ? a.colsum[n] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned m = m_base + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= a.m_total) continue;
                const size_t idx = (size_t)m * a.c_out + n;
                float v = acc[mt][nt][r];
                if (NORM) {
                    v = __builtin_fmaf(nmean[NORM ? mt : 0][NORM ? r : 0], cs, v);
                    v = v * rstd[NORM ? mt : 0][NORM ? r : 0];
                }
                if (a.alpha) v = __builtin_fmaf(v, al, sh);
                else if (a.shift) v = v + sh;
                if (a.res) v = v + a.res[idx];
                if (a.relu) v = v > 0.0f ? v : 0.0f;
                a.y[idx] = v;
            }
    }
    if (NORM) {
        // rows the statistics mark as centred (ipsx_rowstats.h: mean^2 / var > 64 - never on well-conditioned features): this
        // wave's columns of them again, as the chain over (x - mean) * w
        const unsigned mrow = m_base + lane;
        unsigned long long todo = __builtin_amdgcn_ballot_w64(mrow < a.m_total && reinterpret_cast<const float2*>(a.stats)[min(mrow, a.m_total - 1)].y < 0.0f);
        while (todo) {
            const unsigned m = m_base + (unsigned)__builtin_ctzll(todo);
            todo &= todo - 1;
            const float2 st = reinterpret_cast<const float2*>(a.stats)[m];
            for (int c = lane; c < NTW * 32; c += 64) {
                const int n = nt0 * 32 + c;
                if (n >= a.c_out) continue;
                const size_t idx = (size_t)m * a.c_out + n;
                float v = centred_row_dot(a.x + (size_t)m * a.c_in, st.x, a.wp, a.kgs, n) * -st.y;
                if (a.alpha) v = __builtin_fmaf(v, a.alpha[n], a.shift ? a.shift[n] : 0.0f);
                else if (a.shift) v = v + a.shift[n];
                if (a.res) v = v + a.res[idx];
                if (a.relu) v = v > 0.0f ? v : 0.0f;
                a.y[idx] = v;
            }
        }
    }
}

// ------------------------------------------------------------------ the projector of a whole slide as ONE launch
// IPSNet.ips on pre-extracted features (reference architecture/ips_net.py:213-241 with the LayerNorm + Linear + BatchNorm
// + ReLU projector of :60-66): the selection loop consumes rows in order, 256 per iteration, and is as long as the
// projector itself - so what matters is WHEN rows become available, not only how fast the GEMM runs.  Launch by launch
// (ipsx_projector_apply_publish) a part's rows are published when the NEXT launch starts, behind a statistics kernel
// and a logits kernel, and the loop waits for a whole part at the start and works off a whole part at the end.  Here
// resident workgroups pull row tiles off a counter and do everything a tile needs - the Linear on the fp32 matrix cores
// on the RAW rows (the stage loop of conv_nhwc_kernel<1, 4, true, 4>), the LayerNorm moments of the tile's rows taken off
// the very operand registers that feed the MFMAs (round 5: every feature row is read ONCE - rounds 3-4 read it in a
// moments pass first, 10 % of a tile with the matrix pipe idle and twice the HBM traffic), the folded LayerNorm +
// BatchNorm + ReLU in the epilogue, the tile's logits against the folded query (the MFMA sequence of logits_kernel, A
// operand from the LDS copy of the tile) - and publish it: a flag per 32-row unit, and whoever completes the lowest
// unpublished unit advances the loop's progress word past every completed unit behind it.
// Same arithmetic as the three kernels it replaces: embeddings, logits and with them the selection are bit-identical
// (tests/test_hip_kernels.py::test_projector_stream_equals_the_launch_by_launch_projector).
// A tile is 64 rows (2 units); the FIRST pull of `short_first` workgroups is one unit, so that completions - which would
// otherwise come in bursts of one tile per workgroup - are spread over two phases and the first rows arrive after half
// a tile time.  short_first = -2: EVERY tile is one unit - 12 % less throughput (1.46 against 1.31 ms per 65,536 rows on
// 248 units) for a supply without bursts: what ONE slide wants, whose loop consumes rows as fast as they are made
// (35.7 -> 38 M patches/s; the first 1 / 2 / 4 pulls short: 35.7 / 36.7 / 37.1).
struct StreamArgs {
    const float* x;
    const float* wp; unsigned w_bytes;
    const float* alpha; const float* shift; int relu;
    const float* colsum;           // column sums of the weights (the folded LayerNorm's mean term)
    int c_in, c_out, kgs;          // kgs = c_in / 8 packed k-groups per n-tile
    float eps;
    float* emb;                    // (n, c_out)
    const float* vp; int R, vkgs;  // folded query packed for one 32-column tile (R <= 32), vkgs = c_out / 8
    float* logits;                 // (n, R)
    unsigned n, n_units;           // rows (of all slides, one after the other); units of 32 rows
    unsigned slide_rows;           // rows per slide: ready[s] = rows of slide s published
    unsigned short_first;
    int short_pulls;               // every workgroup's first short_pulls pulls are one unit
    unsigned tail_start;           // units from here on are handed out one at a time (the launch ends evenly)
    unsigned split_start, split_units;     // the LAST split_units units are handed out as SPLIT_P column quarters each
    unsigned split_base;           // ctl[split_base]: next quarter; then the quarters' hand-over flags and accumulators
    unsigned exit_word;            // ctl[exit_word]: workgroups that have left (the last one out publishes every row)
    unsigned head_units, head_wgs; // the FIRST head_units units go out as column quarters too, to the first head_wgs workgroups
                                   // (ctl[exit_word + 1]: next head quarter): a lone slide's loop starts on rows that exist after
                                   // a quarter's ~50 us instead of a 32-row tile's ~135 us
    int* ctl;                      // [0] next unit to hand out, [1] first unpublished unit; [2 ...] one flag per unit
    int* ready;                    // rows published, per slide (ipsx_scan_persistent's progress words)
    unsigned long long* stamps;    // diagnostic (ipsx_dbg_projector_stream_stamps): cycles per phase, summed by workgroup 0
};

// The last few units of a launch - what is left when every workgroup has had its whole share: 2,048 units on 255 compute
// units are 8 each and 8 over - would be a round of their own that most of the chip sits out (a 32-row tile: 0.166 ms of
// a 1.38 ms launch).  They are handed out as SPLIT_P column quarters to as many workgroups: every quarter computes the
// rows' moments (off its own operand stream), its 128 columns of the Linear (a wave: 32) and its 16 k-groups of the logits' MFMA chain, whose
// accumulators pass from quarter to quarter through memory (ctl: a flag and 1,024 floats per hand-over) - the same chain
// in the same order, so embeddings and logits stay bit-identical.  The last quarter publishes the unit.
constexpr int SPLIT_P = 4, SPLIT_TAIL = 16, SPLIT_HEAD = 48, SPLIT_MAX = SPLIT_TAIL + SPLIT_HEAD;
// control words of the quarters: [next quarter | a hand-over flag per (unit, part)] - ZEROED by the caller like the words in
// front of them and the two exit words behind them - then the hand-over accumulators, which are written before they are
// read and need no zeroing (786 KB of the 790 KB the caller used to fill in front of every call: advisor, round 5)
constexpr int SPLIT_ZERO_WORDS = 1 + SPLIT_MAX * (SPLIT_P - 1);
constexpr int SPLIT_WORDS = SPLIT_ZERO_WORDS + SPLIT_MAX * (SPLIT_P - 1) * 1024;
constexpr int ST_EP = 516;                         // floats per row of the LDS copy of a tile (512 channels + 4 pad)
constexpr int ST_STATS = 64 * 2;                    // floats: (mean, rstd) of the tile's rows
constexpr size_t ST_LDS = (size_t)64 * ST_EP * 4 + ST_STATS * 4 + 16;

#define ST_STAMP(k)                                                                         \
    do {                                                                                    \
        if (STAMP && blockIdx.x == 0 && threadIdx.x == 0) {                                 \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                     \
            a.stamps[k] += t_ - a.stamps[7];                                                \
            a.stamps[7] = t_;                                                               \
        }                                                                                   \
    } while (0)

__device__ __forceinline__ void bufstore(__amdgpu_buffer_rsrc_t r, f32x4 v, unsigned voff) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, 0, 0);
}

// The rows of a tile that the statistics mark as centred (ipsx_rowstats.h), again, on the centred row: columns [c0, c0 +
// ncol) into the LDS copy of the tile.  Not inlined: a rare path that must not cost the tile's GEMM registers.
__device__ __attribute__((noinline)) void stream_centred_rows(const float* __restrict__ x, const float* __restrict__ wp,
                                                              const float* __restrict__ alpha, const float* __restrict__ shift,
                                                              int relu, int c_in, int kgs, unsigned n, unsigned row0, int c0,
                                                              int ncol, float* tile, const float2* s_stats,
                                                              unsigned long long todo) {
    while (todo) {
        const int lr = __builtin_ctzll(todo);
        todo &= todo - 1;
        const float2 st = s_stats[lr];
        const float* xrow = x + (size_t)min(row0 + (unsigned)lr, n - 1) * c_in;
        for (int c = threadIdx.x; c < ncol; c += 256) {
            const int nn = c0 + c;
            float v = centred_row_dot(xrow, st.x, wp, kgs, nn) * -st.y;
            if (alpha) v = __builtin_fmaf(v, alpha[nn], shift ? shift[nn] : 0.0f);
            else if (shift) v = v + shift[nn];
            if (relu) v = v > 0.0f ? v : 0.0f;
            tile[lr * ST_EP + nn] = v;
        }
    }
}

// MB: the row block (32 rows) whose moments THIS wavefront sums - every wavefront of a workgroup streams the same rows
// through its MFMAs, so the 64-row tile's two blocks are shared out by wavefront parity (4 instead of 8 packed VALU
// instructions per stage: each costs matrix-pipe time) and meet in the LDS; every row's sums are still ONE wavefront's
// chains in the contract's order.
template <int MT, int NTW, int MB, bool STAMP>
__device__ __forceinline__ void stream_tile(const StreamArgs& a, unsigned row0, int part, unsigned split_unit, float* tile,
                                            float2* s_stats) {
    static_assert(NTW == 4 || MT == 1, "column parts are 32-row tiles");
    static_assert(MB < MT, "row block");
    constexpr int P = 4 / NTW;                                     // workgroups that share the tile's columns (NTW = 4: one)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5, i = lane & 31;
    ST_STAMP(0);                                                   // (pull + publication of the tile before)
    ST_STAMP(1);                                                   // (rounds 3-4: the row-moment pass; gone)

    // ---- Linear: this wave's 128 output channels of the tile's 32 MT rows
    // (the buffer is the TILE's rows: any number of slides, one after the other, stays addressable)
    const unsigned rows_here = min(32u * MT, a.n - row0);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (size_t)row0 * a.c_in), 0,
                                                                        (int)(rows_here * (unsigned)a.c_in * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, (int)a.w_bytes, 0x00020000);
    const int nt0 = part * (16 / P) + wave * NTW;
    unsigned pv[MT];
    RowMoments mom;                                                // of row 32 MB + i, this lane's half of every k-group
    rm_zero(mom);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const unsigned lr = mt * 32 + i, row = row0 + lr;
        pv[mt] = row < a.n ? (lr * (unsigned)a.c_in + 4u * half) * 4u : kOob;      // (no such row: zeros, never stored)
    }
    const unsigned lb = lane * 16u;
    unsigned wb[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) wb[t] = (unsigned)(nt0 + t) * (unsigned)a.kgs * 1024u;
    f32x16 acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][t][r] = 0.0f;
    struct Stage { f32x4 a[MT], b[NTW]; };
    Stage s0, s1, s2, s3;          // ring of 4, operands requested THREE stages (3 x 2048 matrix-pipe cycles, ~2.6 us) ahead -
                                   // a ring of 3 (two ahead): 1.246 ms per 65,536 rows on 256 units, of 4: 1.217, of 8: 1.210 -
    const int total = a.kgs;       // with one wavefront per SIMD nothing else hides the latency of rows that come from HBM
    int gp = 0;
#define ST_ISSUE(S)                                                                        \
    do {                                                                                   \
        const unsigned ca = (unsigned)gp * 32u, cb = (unsigned)gp * 1024u;                 \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) S.a[mt] = bufload(rx, pv[mt], ca); \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t) S.b[t] = bufload(rw, lb, wb[t] + cb); \
        if (gp + 1 < total) ++gp;                                                          \
    } while (0)
#define ST_STAGE(SL, SM)                                                                   \
    ST_ISSUE(SL);                                                                          \
    rm_add(mom, SM.a[MB]);                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                          \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                  \
            _Pragma("unroll") for (int t = 0; t < NTW; ++t) acc[mt][t] = MFMA(SM.a[mt][j], SM.b[t][j], acc[mt][t]); \
    if (MT == 2) {                                                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                 \
    } else if (NTW == 4) {                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 \
    } else {                                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
    }                                                                                      \
    SB();
    ST_ISSUE(s0);
    ST_ISSUE(s1);
    ST_ISSUE(s2);
#pragma unroll 1
    for (int g = 0; g < total; g += 4) {           // total is a multiple of 4 (C_in % 32 == 0)
        ST_STAGE(s3, s0)
        ST_STAGE(s0, s1)
        ST_STAGE(s1, s2)
        ST_STAGE(s2, s3)
    }
#undef ST_STAGE
#undef ST_ISSUE
    ST_STAMP(2);

    // ---- the rows' moments meet in the LDS (two wavefronts hold each block's: the same bits)
    // (a row whose mean dwarfs its spread comes back as (mean, -rstd): its moments were recentred, its Linear is redone
    //  on the centred row below - ipsx_rowstats.h; never on well-conditioned features)
    {
        const unsigned mrow = min(row0 + MB * 32u + (unsigned)i, a.n - 1);
        const float2 st = rm_finish(mom, a.c_in, a.eps, lane, a.x + (size_t)mrow * a.c_in + 4 * half);
        if (lane < 32) s_stats[MB * 32 + i] = st;
    }
    __syncthreads();
    f32x2 nmean[MT][8], rstd[MT][8];                               // of accumulator registers (r, r + 1): rows lr, lr + 1
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float4 st = *reinterpret_cast<const float4*>(s_stats + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half);
            nmean[mt][r >> 1] = f32x2{st.y < 0.0f ? 0.0f : -st.x, st.w < 0.0f ? 0.0f : -st.z};
            rstd[mt][r >> 1] = f32x2{__builtin_fabsf(st.y), __builtin_fabsf(st.w)};
        }
    const unsigned long long centred_rows = __builtin_amdgcn_ballot_w64(lane < 32 * MT && s_stats[lane < 32 * MT ? lane : 0].y < 0.0f);

    // ---- folded LayerNorm, BatchNorm affine, ReLU: into the LDS copy of the tile (the logits read it; the wavefronts
    // that have no logits to do carry it to HBM meanwhile, 16 bytes per lane - rounds 3-4 stored 4 bytes per lane from here)
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int nn = (nt0 + t) * 32 + i;
        const float al = a.alpha ? a.alpha[nn] : 1.0f;
        const float sh = a.shift ? a.shift[nn] : 0.0f;
        const float cs = a.colsum[nn];
        const f32x2 al2 = {al, al}, sh2 = {sh, sh}, cs2 = {cs, cs};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const unsigned lr = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                f32x2 v = {acc[mt][t][r], acc[mt][t][r + 1]};
                v = __builtin_elementwise_fma(nmean[mt][r >> 1], cs2, v);
                v = v * rstd[mt][r >> 1];
                if (a.alpha) v = __builtin_elementwise_fma(v, al2, sh2);
                else if (a.shift) v = v + sh2;
                if (a.relu) { v[0] = v[0] > 0.0f ? v[0] : 0.0f; v[1] = v[1] > 0.0f ? v[1] : 0.0f; }
                tile[lr * ST_EP + nn] = v[0];
                tile[(lr + 1) * ST_EP + nn] = v[1];
            }
    }
    __syncthreads();
    if (centred_rows != 0ull) {                                    // workgroup-uniform (every wave read the same statistics); rare
        stream_centred_rows(a.x, a.wp, a.alpha, a.shift, a.relu, a.c_in, a.kgs, a.n, row0, part * (16 / P) * 32, 4 * NTW * 32,
                            tile, s_stats, centred_rows);
        __syncthreads();
    }
    ST_STAMP(3);

    // ---- the tile's embeddings to HBM, by the wavefronts the logits leave idle (rows beyond the end: the buffer's bounds
    // check drops the store)
    if (wave >= MT) {
        constexpr int COLS4 = 32 * NTW;                            // float4 per row of this workgroup's columns
        constexpr int NCOPY = 4 - MT;
        const unsigned col0 = (unsigned)part * (16 / P) * 32u;
        const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(a.emb + (size_t)row0 * a.c_out, 0,
                                                                            (int)(rows_here * (unsigned)a.c_out * 4u), 0x00020000);
#pragma unroll 8
        for (int e = (wave - MT) * 64 + lane; e < 32 * MT * COLS4; e += NCOPY * 64) {
            const unsigned r = (unsigned)e / COLS4, c = col0 + 4u * ((unsigned)e % COLS4);
            bufstore(re, *reinterpret_cast<const f32x4*>(tile + r * ST_EP + c), (r * (unsigned)a.c_out + c) * 4u);
        }
    }

    // ---- logits of the tile's rows: wave mt takes rows 32 mt .. 32 mt + 31, all R logits (one 32-column tile).
    // A column part (P > 1) runs ITS k-groups of the chain: the accumulators arrive from the part before and go on to the
    // part behind through memory (written through, one release fence before the flag; the reader: an acquire after it).
    if (wave < MT) {
        const unsigned lr = wave * 32 + i, row = row0 + lr;
        const bool rv = row < a.n;
        const float* e = tile + lr * ST_EP + 4 * half;
        const float4* vq = reinterpret_cast<const float4*>(a.vp) + lane;
        const int kg0 = part * (64 / P), kg1 = P > 1 ? kg0 + 64 / P : a.vkgs;
        int* const sflag = a.ctl + a.split_base + 1 + split_unit * (SPLIT_P - 1);
        float* const sacc = reinterpret_cast<float*>(a.ctl + a.split_base + SPLIT_ZERO_WORDS + 2) +     // (behind the two exit words)
                            (size_t)split_unit * (SPLIT_P - 1) * 1024;
        f32x16 lacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) lacc[r] = 0.0f;
        if (P > 1 && part > 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool ok = true;
            while (__hip_atomic_load(sflag + part - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) { ok = false; break; }     // 50 ms: never (no hang on a bug)
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
            for (int r = 0; r < 16; ++r)
                lacc[r] = ok ? __hip_atomic_load(sacc + (part - 1) * 1024 + r * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : __builtin_nanf("");
        }
#pragma unroll 1
        for (int k0 = kg0; k0 < kg1; k0 += 8) {
            float4 ev[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ev[u] = *reinterpret_cast<const float4*>(e + (k0 + u) * 8);
                bv[u] = vq[(size_t)(k0 + u) * 64];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                lacc = MFMA(rv ? ev[u].x : 0.0f, bv[u].x, lacc);
                lacc = MFMA(rv ? ev[u].y : 0.0f, bv[u].y, lacc);
                lacc = MFMA(rv ? ev[u].z : 0.0f, bv[u].z, lacc);
                lacc = MFMA(rv ? ev[u].w : 0.0f, bv[u].w, lacc);
            }
        }
        if (P > 1 && part < P - 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __hip_atomic_store(sacc + part * 1024 + r * 64 + lane, lacc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (lane == 0) __hip_atomic_store(sflag + part, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (i < a.R) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned rr = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                // written THROUGH to the level the loop (another XCD) reads from: the publication below then needs no
                // L2 write-back of its own (a buffer_wbl2 per tile costs ~30 us and serialises the XCD: 1.6 -> 5.4 ms)
                if (rr < a.n) __hip_atomic_store(a.logits + (size_t)rr * a.R + i, lacc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // ... and have arrived there
    }
    ST_STAMP(4);
}

template <bool STAMP>
__global__ __launch_bounds__(256, 1) void projector_stream_kernel(StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float st_lds[];
    float* tile = st_lds;
    float2* s_stats = reinterpret_cast<float2*>(st_lds + 64 * ST_EP);
    int* s_u0 = reinterpret_cast<int*>(st_lds + 64 * ST_EP + ST_STATS);    // [0] unit, [1] units taken, [2] column part or -1
    bool first = true, head_left = a.head_units > 0 && blockIdx.x < a.head_wgs;      // (thread 0's)
    int pulls = 0;
    for (;;) {
        if (threadIdx.x == 0) {
            // (the look at the counter may be a pull or two behind: it only decides the SIZE of this pull)
            unsigned u = a.n_units;
            int part = -1, take = 1;
            if (head_left) {                                        // a column quarter of one of the FIRST units
                const unsigned j = (unsigned)atomicAdd(&a.ctl[a.exit_word + 1], 1);
                if (j < a.head_units * SPLIT_P) { u = j / SPLIT_P; part = (int)(j % SPLIT_P); }
                else head_left = false;
            }
            if (part < 0) {
            const bool tail = a.head_units + (unsigned)__hip_atomic_load(&a.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.tail_start;
            take = ((first && blockIdx.x >= a.head_wgs && blockIdx.x - a.head_wgs < (a.short_first & 0x7fffffffu)) || pulls < a.short_pulls || tail) ? 1 : 2;
            u = a.head_units + (unsigned)atomicAdd(&a.ctl[0], take);
            first = false;
            if (u >= a.split_start) {                               // the whole units are gone: a column quarter of one of the last
                const unsigned j = (unsigned)atomicAdd(&a.ctl[a.split_base], 1);
                if (j < a.split_units * SPLIT_P) { u = a.split_start + j / SPLIT_P; part = (int)(j % SPLIT_P); take = 1; }
                else u = a.n_units;
            } else if (u + take > a.split_start) {
                take = 1;
            }
            }
            s_u0[0] = (int)u;
            s_u0[1] = take;
            s_u0[2] = part;
        }
        ++pulls;
        __syncthreads();
        const unsigned u0 = (unsigned)__builtin_amdgcn_readfirstlane(s_u0[0]);
        const int take = __builtin_amdgcn_readfirstlane(s_u0[1]);
        const int part = __builtin_amdgcn_readfirstlane(s_u0[2]);
        if (u0 >= a.n_units) break;                                 // workgroup-uniform
        const int units = (take == 2 && u0 + 1 < a.n_units) ? 2 : 1;
        if (part >= 0)      // (hand-over slots: the last units' [0, SPLIT_TAIL), the first units' behind them)
            stream_tile<1, 4 / SPLIT_P, 0, STAMP>(a, u0 * 32u, part, u0 >= a.split_start ? u0 - a.split_start : SPLIT_TAIL + u0, tile, s_stats);
        else if (units == 2) {
            if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) stream_tile<2, 4, 1, STAMP>(a, u0 * 32u, 0, 0u, tile, s_stats);
            else stream_tile<2, 4, 0, STAMP>(a, u0 * 32u, 0, 0u, tile, s_stats);
        } else stream_tile<1, 4, 0, STAMP>(a, u0 * 32u, 0, 0u, tile, s_stats);
        // ---- publish: the tile's logits have been written through (stream_tile); one release fence, then relaxed atomics.
        // (The embeddings are ordinary stores: nothing reads them before the launch is over.)  The first wavefront sets
        // the flags of this tile's units, reads the cursor - the first unpublished unit - and the 64 flags from there on
        // in ONE load, and raises cursor and progress word past the completed units it finds (atomic maxima: both only
        // ever grow).  Measured on the way: an agent-scope release or acquire per tile (L2 write-back / invalidate: every
        // workgroup of the XCD re-fetches the 4 MB of weights it streams) 1.6 -> 5.4 ms; a compare-and-swap per unit
        // (2,048 round trips to one cache line, one after the other) the same.  Without a full fence between flag and
        // cursor, two workgroups that finish neighbouring units at the same moment may both miss the other's flag; the
        // unit then waits for the next workgroup that finishes anything (about every microsecond) or, at the very end,
        // for the caller's ipsx_publish_rows behind this launch.
        __syncthreads();
        if (threadIdx.x < 64 && (part < 0 || part == SPLIT_P - 1)) {           // (a column quarter: the last one publishes)
            const int lane = threadIdx.x;
            // ONE agent-scope release per tile, in the publishing wavefront only (round 4; the advisor's form): the
            // workgroup barrier above has collected every wave's stores, so flag, cursor and progress word are ordered
            // behind the tile's logits by the memory model and not only by the write-through stores having arrived.
            // Measured: 1.329 -> 1.344 ms per 65,536 rows (a release by EVERY wave, or an acquire on top: 5.4 ms).
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (lane < units) __hip_atomic_store(&a.ctl[2 + u0 + lane], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (;;) {
                const int p = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&a.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if ((unsigned)p >= a.n_units) break;
                const unsigned u = (unsigned)p + lane;
                const int f = u < a.n_units ? __hip_atomic_load(&a.ctl[2 + u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                const unsigned long long done = __ballot(f != 0);
                const int c = done == ~0ull ? 64 : __builtin_ctzll(~done);        // completed units in a row from the cursor
                if (c == 0) break;
                if (lane == 0) {
                    __hip_atomic_fetch_max(&a.ctl[1], p + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // rows [32 p, 32 (p + c)) of the flat row space: every slide they touch learns how far it has got
                    const unsigned done = min((unsigned)(p + c) * 32u, a.n);
                    for (unsigned sl = (unsigned)p * 32u / a.slide_rows; sl * a.slide_rows < done; ++sl)
                        __hip_atomic_fetch_max(a.ready + sl, (int)min(done - sl * a.slide_rows, a.slide_rows), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                }
                if (c < 64) break;
            }
        }
    }
    // The last workgroup out publishes whatever two simultaneous finishers left to each other (round 5; the caller's
    // ipsx_publish_rows launches behind this one did that): every workgroup has fenced its tiles before it counts itself out.
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (__hip_atomic_fetch_add(&a.ctl[a.exit_word], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            __hip_atomic_fetch_max(&a.ctl[1], (int)a.n_units, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (unsigned sl = 0; sl * a.slide_rows < a.n; ++sl)
                __hip_atomic_fetch_max(a.ready + sl, (int)a.slide_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// nn.MaxPool2d(3, 2, 1) on channels-last activations: one thread per (output pixel, 4 channels)
__global__ void maxpool_3x3s2_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total4, int c4,
                                          int h, int w, int ho, int wo) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int cq = (int)(i % c4);
    size_t t = i / c4;
    const int ox = (int)(t % wo); t /= wo;
    const int oy = (int)(t % ho);
    const size_t img = t / ho;
    const float4* src = reinterpret_cast<const float4*>(x) + img * (size_t)h * w * c4 + cq;
    float4 m = make_float4(-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf());
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
            if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
            const float4 v = src[(size_t)(iy * w + ix) * c4];
            m.x = nanmax(m.x, v.x); m.y = nanmax(m.y, v.y); m.z = nanmax(m.z, v.z); m.w = nanmax(m.w, v.w);
        }
    reinterpret_cast<float4*>(y)[i] = m;
}

// nn.AdaptiveAvgPool2d(1) on channels-last activations: (n, hw, c) -> (n, c), sequential sum over pixels
__global__ void avgpool_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int c, int hw) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t img = i / c;
    const int ch = (int)(i - img * c);
    const float* src = x + img * (size_t)hw * c + ch;
    float s = 0.0f;
    for (int j = 0; j < hw; ++j) s = s + src[(size_t)j * c];
    y[i] = s / (float)hw;
}

}  // namespace ipsx

using namespace ipsx;

namespace ipsx {
int conv_nhwc_impl(const ipsx_conv* cv, const float* x, const float* residual, const float* row_stats, float* y,
                   int64_t n, int h, int w, int relu, void* stream, int* ready = nullptr, int ready_value = 0);
}

IPSX_API int ipsx_conv2d_affine_nhwc(const ipsx_conv* cv, const float* x, const float* residual, float* y,
                                     int64_t n, int h, int w, int relu, void* stream) {
    return conv_nhwc_impl(cv, x, residual, nullptr, y, n, h, w, relu, stream);
}

// row_stats != null: the LayerNorm of the x rows is folded into the epilogue ((mean, rstd) per row; 1x1 convolution on 1x1 maps)
int ipsx::conv_nhwc_impl(const ipsx_conv* cv, const float* x, const float* residual, const float* row_stats, float* y,
                         int64_t n, int h, int w, int relu, void* stream, int* ready, int ready_value) {
    IPSX_REQUIRE(cv && cv->w_packed && x && y && n >= 0 && h > 0 && w > 0, "conv2d_affine_nhwc: bad arguments");
    IPSX_REQUIRE(!row_stats || (cv->kh == 1 && cv->kw == 1 && cv->pad == 0 && cv->stride == 1 && h == 1 && w == 1 && cv->colsum),
                 "conv2d_affine_nhwc: row statistics go with a Linear layer (1x1 convolution on rows) that carries its column sums");
    IPSX_REQUIRE(cv->c_in % 32 == 0, "conv2d_affine_nhwc: C_in = %d is not a multiple of 32", cv->c_in);
    if (n == 0) return IPSX_OK;
    const int ho = conv_out(h, cv->kh, cv->stride, cv->pad), wo = conv_out(w, cv->kw, cv->stride, cv->pad);
    IPSX_REQUIRE(ho > 0 && wo > 0, "conv2d_affine_nhwc: empty output");
    const int64_t howo = (int64_t)ho * wo;
    const int kgs = cv->kh * cv->kw * cv->c_in / 8;
    const int64_t w_bytes = (int64_t)cdiv(cv->c_out, 32) * kgs * 1024;
    IPSX_REQUIRE(w_bytes < ((int64_t)1 << 31), "conv2d_affine_nhwc: weights too large for one buffer");
    // per launch: input below 2 GiB (buffer range + the out-of-range marker), output pixels below 2^31
    const int64_t in_img = (int64_t)h * w * cv->c_in * 4;
    int64_t per = std::min<int64_t>(n, std::max<int64_t>(1, (((int64_t)1 << 31) - 65536) / in_img));
    per = std::min<int64_t>(per, std::max<int64_t>(1, ((int64_t)1 << 30) / howo));
    for (int64_t i0 = 0; i0 < n; i0 += per) {
        const int64_t cnt = std::min(per, n - i0);
        NhwcArgs a;
        a.x = x + (size_t)i0 * h * w * cv->c_in;
        a.y = y + (size_t)i0 * howo * cv->c_out;
        a.res = residual ? residual + (size_t)i0 * howo * cv->c_out : nullptr;
        a.wp = cv->w_packed; a.alpha = cv->alpha; a.shift = cv->shift;
        a.m_total = (unsigned)(cnt * howo);
        a.x_bytes = (unsigned)(cnt * in_img);
        a.w_bytes = (unsigned)w_bytes;
        a.c_in = cv->c_in; a.h = h; a.w = w; a.c_out = cv->c_out; a.ho = ho; a.wo = wo;
        a.kh = cv->kh; a.kw = cv->kw; a.stride = cv->stride; a.pad = cv->pad; a.relu = relu;
        a.spt = cv->c_in / 8; a.kgs = kgs;
        a.stats = row_stats ? row_stats + (size_t)i0 * 2 : nullptr;
        a.colsum = cv->colsum;
        a.ready = (row_stats && i0 == 0) ? ready : nullptr;      // (the first launch of the call follows what was enqueued before)
        a.ready_value = ready_value;
        const unsigned mt64 = (unsigned)cdiv(a.m_total, 64), nt64 = (unsigned)cdiv(cv->c_out, 64);
        hipStream_t s = as_stream(stream);
        const unsigned nt128 = (unsigned)cdiv(cv->c_out, 128);
        if (row_stats) {
            // Linear layers with LayerNorm in the operand load (projector): wave tile 64 x 128, a workgroup = 64 rows x 512
            // columns.  A launch is as long as ONE workgroup takes, however few there are: one that would leave half the
            // compute units idle anyway (<= 127 row tiles) runs as twice as many workgroups of 64 rows x 256 columns (wave
            // tile 64 x 64) and is over in half the time - what a slab-by-slab producer wants for the slab a consumer is
            // waiting for (the first and the last rows of a slide, IPSNet._select_hip_overlapped).  Same fma chain per output.
            // A launch of at most one workgroup per compute unit asks for more than half of the LDS it does not use: the
            // dispatcher then CANNOT put two of them on one unit (it otherwise does, now and then, long before the units run
            // out - a part of 232 workgroups beside the resident loop took twice as long) nor one beside a
            // persistent selection loop, whose buffers fill most of its unit's LDS.
            const size_t lds_pad = 96 * 1024;
            const bool big = cv->c_out >= 512 && mt64 > 127;
            const unsigned wgs = big ? mt64 * (unsigned)cdiv(nt128, 4) : mt64 * (unsigned)cdiv(nt64, 4);
            const size_t lds = (cv->c_out >= 256 && wgs <= 256) ? lds_pad : 0;
            if (big) {
                if (lds) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_nhwc_kernel<1, 4, true, 4>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pad);
                conv_nhwc_kernel<1, 4, true, 4><<<dim3(mt64, (unsigned)cdiv(nt128, 4)), dim3(256), lds, s>>>(a);
            } else if (cv->c_out >= 256) {
                if (lds) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_nhwc_kernel<1, 4, true, 2>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pad);
                conv_nhwc_kernel<1, 4, true, 2><<<dim3(mt64, (unsigned)cdiv(nt64, 4)), dim3(256), lds, s>>>(a);
            }
            else
                conv_nhwc_kernel<2, 2, true, 2><<<dim3((unsigned)cdiv(mt64, 2), (unsigned)cdiv(nt64, 2)), dim3(256), 0, s>>>(a);
        } else if (cv->c_out >= 256)        // wide layer: the 4 waves share the activation rows
            conv_nhwc_kernel<1, 4, false, 2><<<dim3(mt64, (unsigned)cdiv(nt64, 4)), dim3(256), 0, s>>>(a);
        else if (cv->c_out > 64)
            conv_nhwc_kernel<2, 2, false, 2><<<dim3((unsigned)cdiv(mt64, 2), (unsigned)cdiv(nt64, 2)), dim3(256), 0, s>>>(a);
        else
            conv_nhwc_kernel<4, 1, false, 2><<<dim3((unsigned)cdiv(mt64, 4), nt64), dim3(256), 0, s>>>(a);
        IPSX_TRY(launched("conv2d_affine_nhwc"));
    }
    return IPSX_OK;
}

IPSX_API int ipsx_maxpool_3x3s2_nhwc(const float* x, float* y, int64_t n, int c, int h, int w, void* stream) {
    IPSX_REQUIRE(x && y && n >= 0 && c > 0 && c % 4 == 0 && h > 0 && w > 0, "maxpool_nhwc: bad arguments");
    const int ho = conv_out(h, 3, 2, 1), wo = conv_out(w, 3, 2, 1);
    const size_t total4 = (size_t)n * ho * wo * (c / 4);
    if (!total4) return IPSX_OK;
    maxpool_3x3s2_nhwc_kernel<<<dim3((unsigned)cdiv(total4, 256)), dim3(256), 0, as_stream(stream)>>>(
        x, y, total4, c / 4, h, w, ho, wo);
    return launched("maxpool_nhwc");
}

IPSX_API int ipsx_avgpool_nhwc(const float* x, float* y, int64_t n, int c, int hw, void* stream) {
    IPSX_REQUIRE(x && y && n >= 0 && c > 0 && hw > 0, "avgpool_nhwc: bad arguments");
    const size_t total = (size_t)n * c;
    if (!total) return IPSX_OK;
    avgpool_nhwc_kernel<<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream)>>>(x, y, total, c, hw);
    return launched("avgpool_nhwc");
}

// The whole projector + logits of one slide as ONE persistent launch that publishes rows to a resident selection loop
// (ipsx_scan_persistent) as it goes; see projector_stream_kernel.  ctl: ipsx_projector_stream_ctl_words(n) int32 words,
// ZEROED by the caller before every call (the work counter, the publication cursor, a flag per 32 rows).
static unsigned long long* g_stream_stamps = nullptr;
// Diagnostic (not part of include/ipsx.h; tools/projector_stream_bench.py): 8 zeroed uint64 on the device; workgroup 0 adds
// the shader cycles of each phase of its tiles - [0] pull + publication, [1] moments, [2] GEMM, [3] epilogue, [4] logits.
extern "C" __attribute__((visibility("default"))) void ipsx_dbg_projector_stream_stamps(unsigned long long* p) { g_stream_stamps = p; }

// Units a lone slide's stream hands out as column quarters FIRST (0: none; at most SPLIT_HEAD; IPSX_CAM_HEAD).  Measured,
// one CAMELYON slide per call (M patches/s of the call | the stream's share of the fp32 MFMA peak inside the call):
// 0: 46.6 | 0.736, 16: 47.0 | 0.724, 32: 47.3 | 0.709, 48: 47.6 | 0.707 - the loop starts ~60 us earlier and the call gains
// 2 %, the stream itself loses 4 % (quarters run at ~2/3 of a whole tile's rate).  Off by default: the call is bound by
// the loop's 255 iterations either way, and the kernel's own rate is the figure its roofline is judged on.
static int g_stream_head = 0;
extern "C" __attribute__((visibility("default"))) void ipsx_dbg_stream_head(int units) { g_stream_head = units; }

static int g_stream_split = 1;
// Diagnostic: 0 = the last units of a guided launch as whole 32-row tiles (no column quarters)
extern "C" __attribute__((visibility("default"))) void ipsx_dbg_stream_split(int on) { g_stream_split = on; }

IPSX_API size_t ipsx_projector_stream_ctl_words(int64_t n) {
    return n > 0 ? (size_t)ipsx::cdiv(n, 32) + 2 + ipsx::SPLIT_WORDS + 2 : 0;
}
// ... of which only the FIRST this many must be zero when a call starts (the rest are hand-over accumulators)
IPSX_API size_t ipsx_projector_stream_ctl_zero_words(int64_t n) {
    return n > 0 ? (size_t)ipsx::cdiv(n, 32) + 2 + ipsx::SPLIT_ZERO_WORDS + 2 : 0;
}

IPSX_API int ipsx_projector_stream_supported(const ipsx_conv* lin, int64_t n, int r) {
    if (!lin || !lin->w_packed || lin->kh != 1 || lin->kw != 1 || lin->stride != 1 || lin->pad != 0) return 0;
    if (lin->c_out != 512 || lin->c_in % 32 != 0 || !lin->colsum) return 0;
    if (r < 1 || r > 32 || n < 64) return 0;
    return n < ((int64_t)1 << 31) - 64 ? 1 : 0;
}

IPSX_API int ipsx_projector_stream(const ipsx_conv* lin, const float* x, int64_t n, int64_t slide_rows, float ln_eps, float* emb,
                                   const float* v_packed, int r, float* logits, int32_t* ctl, int32_t* ready,
                                   int workgroups, int short_first, void* stream) {
    IPSX_REQUIRE(lin && x && emb && v_packed && logits && ctl && ready, "projector_stream: bad arguments");
    IPSX_REQUIRE(slide_rows > 0 && n % slide_rows == 0 && (n == slide_rows || slide_rows % 32 == 0),
                 "projector_stream: %lld rows are not whole slides of %lld rows (a multiple of 32 when there are several)",
                 (long long)n, (long long)slide_rows);
    IPSX_REQUIRE(ipsx_projector_stream_supported(lin, n, r), "projector_stream: needs a 1x1 Linear with 512 outputs, C_in %% 32 == 0, "
                 "its column sums (ipsx_conv.colsum) and at most 32 logits per row");
    ipsx::StreamArgs a;
    a.x = x;
    a.wp = lin->w_packed; a.kgs = lin->c_in / 8;
    a.w_bytes = (unsigned)((int64_t)(lin->c_out / 32) * a.kgs * 1024);
    a.alpha = lin->alpha; a.shift = lin->shift; a.relu = 1; a.colsum = lin->colsum;
    a.c_in = lin->c_in; a.c_out = lin->c_out; a.eps = ln_eps;
    a.emb = emb; a.vp = v_packed; a.R = r; a.vkgs = lin->c_out / 8; a.logits = logits;
    a.n = (unsigned)n; a.n_units = (unsigned)ipsx::cdiv(n, 32); a.slide_rows = (unsigned)slide_rows;
    a.ctl = ctl; a.ready = ready; a.stamps = g_stream_stamps;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // 7 of the 8 compute units of every shader engine: the resident loop's unit is then free wherever the dispatcher
    // turns, and the LDS request keeps these workgroups one to a unit and off the loop's
    const int wgs = workgroups > 0 ? workgroups : cus / 8 * 7;
    a.short_first = (unsigned)(short_first >= 0 ? short_first : wgs / 2);
    a.short_pulls = short_first == -2 ? 0x7fffffff : 0;                            // (-2: every tile 32 rows)
    a.tail_start = a.n_units;
    a.split_start = a.n_units; a.split_units = 0; a.split_base = a.n_units + 2;
    a.exit_word = a.n_units + 2 + (unsigned)ipsx::SPLIT_ZERO_WORDS;
    a.head_units = a.head_wgs = 0;
    if (short_first <= -3) {
        // -3 - (head + 8 tail): every workgroup's first `head` pulls and the last `tail` x workgroups units are 32-row tiles -
        // early first rows, full-rate 64-row tiles in the middle, and an end without a last round that most units sit out
        const int k = -short_first - 3, head = k & 7, tail = k >> 3;
        a.short_pulls = head;
        if (head > 0) a.short_first = 0;
        // What is left over when every WORKING workgroup has had its whole share of units goes out in column quarters.
        // Working: a launch that leaves compute units to resident loops is dealt to the XCDs round-robin whatever is
        // free there, so each loop may keep one workgroup of its XCD waiting until the others leave - the
        // share is computed for 2 wgs - cus workers (255 of 256: 254; too few assumed just starts the quarters early).
        const unsigned eff = (unsigned)std::max(1, wgs < cus ? 2 * wgs - cus : wgs);
        // ONE slide, long enough: its loop consumes rows about as fast as they are made, so when the FIRST rows exist
        // decides when it can start - the first units go out as column quarters to 3/8 of the workgroups
        // (two rounds of ~50 us each), beside the others' first 32-row (~135 us) and 64-row (~257 us) tiles
        if (g_stream_head > 0 && n == slide_rows && a.n_units >= 16u * (unsigned)wgs / 4 && wgs >= 64) {
            a.head_units = (unsigned)std::min(g_stream_head, ipsx::SPLIT_HEAD);
            a.head_wgs = (unsigned)wgs * 3 / 8;
            a.short_first = (a.short_first != 0) ? ((unsigned)wgs - a.head_wgs) / 2 : 0;
        }
        const unsigned main_units = a.n_units - a.head_units;
        const unsigned per = main_units / eff, left = main_units - per * eff;
        if (g_stream_split && per >= 2 && left > 0 && left <= (unsigned)ipsx::SPLIT_TAIL && left * ipsx::SPLIT_P <= eff) {
            a.split_units = left;
            a.split_start = a.n_units - left;
        }
        const long long ts = (long long)a.split_start - (long long)tail * wgs;
        a.tail_start = (unsigned)(ts > 0 ? ts : 0);
    }

    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ipsx::projector_stream_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)ipsx::ST_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ipsx::projector_stream_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)ipsx::ST_LDS);
        attr = true;
    }
    if (a.stamps)
        ipsx::projector_stream_kernel<true><<<dim3((unsigned)wgs), dim3(256), ipsx::ST_LDS, ipsx::as_stream(stream)>>>(a);
    else
        ipsx::projector_stream_kernel<false><<<dim3((unsigned)wgs), dim3(256), ipsx::ST_LDS, ipsx::as_stream(stream)>>>(a);
    return ipsx::launched("projector_stream");
}
