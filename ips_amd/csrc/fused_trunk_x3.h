// fused_trunk_x3.h - the fused trunk with every fp32 operand of the residual stages carried as THREE bf16
// terms (included by fused_trunk.hip; precision 2, "fp32x3").
//
// Why: the fp32 matrix pipe tops out at 157 TFLOP/s; the bf16 pipe at 2.5 PFLOP/s.  A float splits EXACTLY
// into hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (3 x 8 significand bits,
// same exponent range as fp32), so  x*w = sum of 9 bf16 products; the three smallest (mid*lo, lo*mid, lo*lo)
// are below 2^-24 |x*w| - under the rounding of the fp32 accumulator itself - and are dropped.  The remaining
// SIX products run on v_mfma_f32_32x32x16_bf16 (exact bf16 products, fp32 accumulation): 6 x 32 cycles per 16 k
// against 8 x 64 for v_mfma_f32_32x32x2_f32, i.e. 2.67x less matrix-pipe time for the same contraction at the
// accuracy of an fp32 chain (oracle comparison: tests/test_hip_kernels.py::test_fp32x3_*; it is NOT bit-identical
// to the fma chain of the contract, so this path is tolerance-tested and opt-in).
//
// Structure = the fp32 kernel's (wave = patch in the 8x8 stage, 4 waves x 4 patches in the 4x4 stage, stem in
// exact fp32), with the contraction TRANSPOSED: weights are the A operand (rows = output channels), activations
// the B operand (columns = pixels).  A lane then owns one pixel and 4 CONSECUTIVE channels per register quad, so
// the epilogue (BatchNorm, residual, ReLU, 3-way split) packs 4 bf16 per plane and stores 8 bytes at a time.
//   LDS image   [pixel][plane][C bf16 + 8 pad]   8x8: 3 x 144 B per pixel row, 4x4: 3 x 272 B; one zero row (halo)
//   weights     [row tile][K/16][plane][64 lanes][8 bf16]  (ipsx_pack_conv_weight_x3), 4.0 MB for the trunk
//   stage       one K-step of 16 in the 8x8 stage: 6 ds_read_b128 + 6 global_load_dwordx4 + 24 MFMA, the loads of
//               the next stage spread between the MFMAs with sched_group_barrier (tools/ubench/x3_stage.hip:
//               833-843 cycles per stage against 768 of pure matrix-pipe time depending on the pattern; 901 with the loads in front)
// 112 KB of LDS per workgroup: one workgroup (one wave per SIMD) per CU.

constexpr int XP1 = 144, XR1 = 3 * XP1, XZ1 = 64;     // 8x8 stage: plane bytes, row bytes, zero row
constexpr int XP2 = 272, XR2 = 3 * XP2, XZ2 = 16;     // 4x4 stage
constexpr int SLABX = (XZ1 + 1) * XR1;                // 28,080 B per patch (>= 38*38*4, >= 17*816, >= 16*132*4)

struct XOp1 { uint4 p[2][3]; };                       // [tile][plane] operand registers of one K-step

__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xFFFF0000u); }
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    return (unsigned)bf16_bits(a) | ((unsigned)bf16_bits(b) << 16);
}

// 4 floats -> hi / mid / lo planes, 4 bf16 (8 bytes) each; hi + mid + lo == x exactly
__device__ __forceinline__ void split4(const float (&x)[4], uint2& hi, uint2& mid, uint2& lo) {
    hi.x = pk_bf16(x[0], x[1]); hi.y = pk_bf16(x[2], x[3]);
    const float r0 = x[0] - bf_lo(hi.x), r1 = x[1] - bf_hi(hi.x), r2 = x[2] - bf_lo(hi.y), r3 = x[3] - bf_hi(hi.y);
    mid.x = pk_bf16(r0, r1); mid.y = pk_bf16(r2, r3);
    const float q0 = r0 - bf_lo(mid.x), q1 = r1 - bf_hi(mid.x), q2 = r2 - bf_lo(mid.y), q3 = r3 - bf_hi(mid.y);
    lo.x = pk_bf16(q0, q1); lo.y = pk_bf16(q2, q3);
}

__device__ __forceinline__ void join4(const uint2& hi, const uint2& mid, const uint2& lo, float (&x)[4]) {
    x[0] = (bf_lo(hi.x) + bf_lo(mid.x)) + bf_lo(lo.x);
    x[1] = (bf_hi(hi.x) + bf_hi(mid.x)) + bf_hi(lo.x);
    x[2] = (bf_lo(hi.y) + bf_lo(mid.y)) + bf_lo(lo.y);
    x[3] = (bf_hi(hi.y) + bf_hi(mid.y)) + bf_hi(lo.y);
}

// stem output (standard C layout: lane = channel, registers = pixels) -> the three planes of the 8x8 image
__device__ __forceinline__ void store_stem_x(char* S, const f32x16 (&v)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float x = v[mt][nt][r];
                const unsigned short h = bf16_bits(x);
                const float r1 = x - __uint_as_float((unsigned)h << 16);
                const unsigned short m = bf16_bits(r1);
                const unsigned short l = bf16_bits(r1 - __uint_as_float((unsigned)m << 16));
                char* d = S + pix * XR1 + 2 * (nt * 32 + i);
                *reinterpret_cast<unsigned short*>(d) = h;
                *reinterpret_cast<unsigned short*>(d + XP1) = m;
                *reinterpret_cast<unsigned short*>(d + 2 * XP1) = l;
            }
}

// transposed C layout of the 8x8 stage: t[rt][ct][r] = channel 32rt + (r&3) + 8(r>>2) + 4half of pixel 32ct + i
__device__ __forceinline__ void load_l1x(const char* S, f32x16 (&t)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const char* s = S + (ct * 32 + i) * XR1 + 2 * (rt * 32 + 8 * g + 4 * half);
                float x[4];
                join4(*reinterpret_cast<const uint2*>(s), *reinterpret_cast<const uint2*>(s + XP1),
                      *reinterpret_cast<const uint2*>(s + 2 * XP1), x);
#pragma unroll
                for (int j = 0; j < 4; ++j) t[rt][ct][4 * g + j] = x[j];
            }
}

// BatchNorm (+ identity) + ReLU on the transposed tile, split and store; RES: v += idn, idn = result
template <bool RES>
__device__ __forceinline__ void epilogue_l1x(char* S, const float* __restrict__ al, const float* __restrict__ sh,
                                             const f32x16 (&acc)[2][2], f32x16 (&idn)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = rt * 32 + 8 * g + 4 * half;
            const float4 A = *reinterpret_cast<const float4*>(al + ch), B = *reinterpret_cast<const float4*>(sh + ch);
            const float Aa[4] = {A.x, A.y, A.z, A.w}, Bb[4] = {B.x, B.y, B.z, B.w};
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = __builtin_fmaf(acc[rt][ct][4 * g + j], Aa[j], Bb[j]);
                    if (RES) x = x + idn[rt][ct][4 * g + j];
                    x = x > 0.0f ? x : 0.0f;
                    if (RES) idn[rt][ct][4 * g + j] = x;
                    v[j] = x;
                }
                uint2 hi, mid, lo;
                split4(v, hi, mid, lo);
                char* d = S + (ct * 32 + i) * XR1 + 2 * ch;
                *reinterpret_cast<uint2*>(d) = hi;
                *reinterpret_cast<uint2*>(d + XP1) = mid;
                *reinterpret_cast<uint2*>(d + 2 * XP1) = lo;
            }
        }
}

// ------------------------------------------------------------------ 8x8 stage, wave = patch
struct XTap { const char* s0; const char* s1; };

__device__ __forceinline__ XTap x1_tap(int tap, const char* S, int i, int half) {
    const int t3 = tap / 3;
    const int dy = t3 - 1, dx = tap - 3 * t3 - 1;
    const int x = i & 7, y0 = i >> 3;
    const bool okx = (unsigned)(x + dx) < 8u;
    const bool ok0 = okx && (unsigned)(y0 + dy) < 8u;
    const bool ok1 = okx && (unsigned)(y0 + 4 + dy) < 8u;
    const int p0 = i + dy * 8 + dx;
    XTap d;
    d.s0 = S + (ok0 ? p0 : XZ1) * XR1 + 16 * half;
    d.s1 = S + (ok1 ? p0 + 32 : XZ1) * XR1 + 16 * half;
    return d;
}

template <int KS>
__device__ __forceinline__ void x1_load(XOp1& st, const XTap& d) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        st.p[0][pl] = *reinterpret_cast<const uint4*>(d.s0 + pl * XP1 + KS * 32);
        st.p[1][pl] = *reinterpret_cast<const uint4*>(d.s1 + pl * XP1 + KS * 32);
    }
}

__device__ __forceinline__ void x1_loadw(XOp1& w, const char* wb, unsigned loff, int g) {
    g = g < 36 ? g : 35;
    const char* p = wb + (size_t)g * 3072;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        w.p[0][pl] = *reinterpret_cast<const uint4*>(p + pl * 1024 + loff);
        w.p[1][pl] = *reinterpret_cast<const uint4*>(p + (size_t)36 * 3072 + pl * 1024 + loff);
    }
}

// the six significant plane pairs, small terms first: (w.lo, x.hi) (w.hi, x.lo) (w.mid, x.mid) (w.mid, x.hi) (w.hi, x.mid) (w.hi, x.hi)
__device__ __forceinline__ void x1_mma(const XOp1& w, const XOp1& x, f32x16 (&acc)[2][2]) {
    constexpr int PW_[6] = {2, 0, 1, 1, 0, 0}, PX_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = MFMA16(w.p[rt][PW_[q]], x.p[ct][PX_[q]], acc[rt][ct]);
}

// 24 MFMAs with the 12 loads of the following stage spread between them
#define X1_GROUPS()                                                      \
    _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) {                   \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);               \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);               \
    }                                                                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
#define X1_STAGE(XC, LOADN, WC, WF) \
    LOADN; x1_loadw(WF, w, lo, g + 2); x1_mma(WC, XC, acc); X1_GROUPS(); SB(); ++g;

__device__ __forceinline__ void conv_l1x(const void* __restrict__ wp, const char* S, f32x16 (&acc)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
    const char* w = reinterpret_cast<const char*>(wp);
    const unsigned lo = lane * 16;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) zero(acc[rt][ct]);
    XTap cur = x1_tap(0, S, i, half);
    XOp1 xa, xb, w0, w1, w2;                       // weight ring of 3: slot = stage % 3, refilled 2 stages ahead
    x1_loadw(w0, w, lo, 0);
    x1_loadw(w1, w, lo, 1);
    x1_load<0>(xa, cur);
    int g = 0;
#pragma unroll 1
    for (int tap = 0; tap < 9; tap += 3) {         // 3 taps = 12 stages per trip: the ring closes
        XTap nxt = x1_tap(tap + 1, S, i, half);
        X1_STAGE(xa, x1_load<1>(xb, cur), w0, w2) X1_STAGE(xb, x1_load<2>(xa, cur), w1, w0)
        X1_STAGE(xa, x1_load<3>(xb, cur), w2, w1) X1_STAGE(xb, x1_load<0>(xa, nxt), w0, w2)
        cur = nxt; nxt = x1_tap(tap + 2, S, i, half);
        X1_STAGE(xa, x1_load<1>(xb, cur), w1, w0) X1_STAGE(xb, x1_load<2>(xa, cur), w2, w1)
        X1_STAGE(xa, x1_load<3>(xb, cur), w0, w2) X1_STAGE(xb, x1_load<0>(xa, nxt), w1, w0)
        cur = nxt; nxt = x1_tap(tap + 3 < 9 ? tap + 3 : 8, S, i, half);
        X1_STAGE(xa, x1_load<1>(xb, cur), w2, w1) X1_STAGE(xb, x1_load<2>(xa, cur), w0, w2)
        X1_STAGE(xa, x1_load<3>(xb, cur), w1, w0) X1_STAGE(xb, x1_load<0>(xa, nxt), w2, w1)
        cur = nxt;
    }
}

// ------------------------------------------------------------------ 4x4 stage, 4 waves x 4 patches
// Column tile ct = patches 2ct, 2ct+1 (column i -> patch 2ct + (i>>4), pixel i & 15); wave `wave` owns the
// output channels 32*wave .. 32*wave+31 (row tile) for both column tiles, so each weight is fetched once per
// workgroup.  SK K-steps per stage so that a tap is always 4 stages.
template <int SK> struct XOp2 { uint4 x[2][SK][3]; };     // activations [ct][k-step][plane]
template <int SK> struct XW2 { uint4 w[SK][3]; };         // weights [k-step][plane]

template <int WIN, int RB, int ZR, int STRIDE, int KS>
__device__ __forceinline__ XTap x2_tap(int tap, const char* S0, int oy, int ox) {
    constexpr int PAD = KS / 2;
    const int ky = tap / KS, kx = tap - ky * KS;
    const int iy = oy * STRIDE + ky - PAD, ix = ox * STRIDE + kx - PAD;
    const bool ok = (unsigned)iy < (unsigned)WIN && (unsigned)ix < (unsigned)WIN;
    XTap d;
    d.s0 = S0 + (ok ? iy * WIN + ix : ZR) * RB;
    d.s1 = d.s0 + 2 * SLABX;
    return d;
}

template <int SK, int PB, int CS>
__device__ __forceinline__ void x2_load(XOp2<SK>& st, const XTap& d) {
#pragma unroll
    for (int q = 0; q < SK; ++q)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            st.x[0][q][pl] = *reinterpret_cast<const uint4*>(d.s0 + pl * PB + (CS * SK + q) * 32);
            st.x[1][q][pl] = *reinterpret_cast<const uint4*>(d.s1 + pl * PB + (CS * SK + q) * 32);
        }
}

template <int SK, int G>
__device__ __forceinline__ void x2_loadw(XW2<SK>& b, const char* wb, unsigned loff, int g) {
    g = g < G ? g : G - 1;
    const char* p = wb + (size_t)g * SK * 3072;
#pragma unroll
    for (int q = 0; q < SK; ++q)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) b.w[q][pl] = *reinterpret_cast<const uint4*>(p + q * 3072 + pl * 1024 + loff);
}

template <int SK>
__device__ __forceinline__ void x2_mma(const XW2<SK>& b, const XOp2<SK>& st, f32x16 (&acc)[2]) {
    constexpr int PW_[6] = {2, 0, 1, 1, 0, 0}, PX_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int q = 0; q < SK; ++q)
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            acc[0] = MFMA16(b.w[q][PW_[t]], st.x[0][q][PX_[t]], acc[0]);
            acc[1] = MFMA16(b.w[q][PW_[t]], st.x[1][q][PX_[t]], acc[1]);
        }
}

// per stage: 12*SK MFMAs, 6*SK ds_read_b128, 3*SK global loads
#define X2_GROUPS()                                                      \
    _Pragma("unroll") for (int q_ = 0; q_ < 3 * SK; ++q_) {              \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);               \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);               \
    }                                                                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 3 * SK, 0);
#define X2_STAGE(XC, LOADN, WC, WF) \
    LOADN; x2_loadw<SK, G>(WF, w, lo, g + 2); x2_mma<SK>(WC, XC, acc); X2_GROUPS(); SB(); ++g;

template <int CIN, int WIN, int RB, int PB, int ZR, int STRIDE, int KS>
__device__ __forceinline__ void conv_l2x(const void* __restrict__ wp, const char* lds, f32x16 (&acc)[2], int lane,
                                         int wave) {
    constexpr int SK = CIN / 64, TAPS = KS * KS, KSTEPS = TAPS * CIN / 16, G = KSTEPS / SK;   // G = 4 * TAPS stages
    const int i = lane & 31, half = lane >> 5;
    const int pix = i & 15, oy = pix >> 2, ox = pix & 3;
    const char* S0 = lds + (i >> 4) * SLABX + 16 * half;
    const char* w = reinterpret_cast<const char*>(wp) + (size_t)__builtin_amdgcn_readfirstlane(wave) * KSTEPS * 3072;
    const unsigned lo = lane * 16;
    zero(acc[0]); zero(acc[1]);
    XTap cur = x2_tap<WIN, RB, ZR, STRIDE, KS>(0, S0, oy, ox);
    XOp2<SK> xa, xb;
    XW2<SK> w0, w1, w2;
    x2_loadw<SK, G>(w0, w, lo, 0);
    x2_loadw<SK, G>(w1, w, lo, 1);
    x2_load<SK, PB, 0>(xa, cur);
    int g = 0;
    if (TAPS == 1) {                               // 1x1 projection: 4 stages, ring of 3 walked once
        X2_STAGE(xa, (x2_load<SK, PB, 1>(xb, cur)), w0, w2) X2_STAGE(xb, (x2_load<SK, PB, 2>(xa, cur)), w1, w0)
        X2_STAGE(xa, (x2_load<SK, PB, 3>(xb, cur)), w2, w1) X2_STAGE(xb, (x2_load<SK, PB, 3>(xa, cur)), w0, w2)
        return;
    }
#pragma unroll 1
    for (int tap = 0; tap < TAPS; tap += 3) {      // 3 taps = 12 stages per trip: the ring of 3 closes
        XTap nxt = x2_tap<WIN, RB, ZR, STRIDE, KS>(tap + 1, S0, oy, ox);
        X2_STAGE(xa, (x2_load<SK, PB, 1>(xb, cur)), w0, w2) X2_STAGE(xb, (x2_load<SK, PB, 2>(xa, cur)), w1, w0)
        X2_STAGE(xa, (x2_load<SK, PB, 3>(xb, cur)), w2, w1) X2_STAGE(xb, (x2_load<SK, PB, 0>(xa, nxt)), w0, w2)
        cur = nxt; nxt = x2_tap<WIN, RB, ZR, STRIDE, KS>(tap + 2, S0, oy, ox);
        X2_STAGE(xa, (x2_load<SK, PB, 1>(xb, cur)), w1, w0) X2_STAGE(xb, (x2_load<SK, PB, 2>(xa, cur)), w2, w1)
        X2_STAGE(xa, (x2_load<SK, PB, 3>(xb, cur)), w0, w2) X2_STAGE(xb, (x2_load<SK, PB, 0>(xa, nxt)), w1, w0)
        cur = nxt; nxt = x2_tap<WIN, RB, ZR, STRIDE, KS>(tap + 3 < TAPS ? tap + 3 : TAPS - 1, S0, oy, ox);
        X2_STAGE(xa, (x2_load<SK, PB, 1>(xb, cur)), w2, w1) X2_STAGE(xb, (x2_load<SK, PB, 2>(xa, cur)), w0, w2)
        X2_STAGE(xa, (x2_load<SK, PB, 3>(xb, cur)), w1, w0) X2_STAGE(xb, (x2_load<SK, PB, 0>(xa, nxt)), w2, w1)
        cur = nxt;
    }
}

// transposed tile of the 4x4 stage: v[ct][r] = channel 32wave + (r&3) + 8(r>>2) + 4half of patch 2ct + (i>>4), pixel i&15
// MODE 0: BN + ReLU -> planes;  1: BN + identity + ReLU -> planes, identity updated;  2: like 1 but the result is
// stored as fp32 [pix][PS2] for the average pool
template <int MODE>
__device__ __forceinline__ void epilogue_l2x(char* lds, const float* __restrict__ al, const float* __restrict__ sh,
                                             const f32x16 (&acc)[2], f32x16 (&id2)[2], int lane, int wave) {
    const int i = lane & 31, half = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ch = 32 * wave + 8 * g + 4 * half;
        const float4 A = *reinterpret_cast<const float4*>(al + ch), B = *reinterpret_cast<const float4*>(sh + ch);
        const float Aa[4] = {A.x, A.y, A.z, A.w}, Bb[4] = {B.x, B.y, B.z, B.w};
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = __builtin_fmaf(acc[ct][4 * g + j], Aa[j], Bb[j]);
                if (MODE != 0) x = x + id2[ct][4 * g + j];
                x = x > 0.0f ? x : 0.0f;
                if (MODE != 0) id2[ct][4 * g + j] = x;
                v[j] = x;
            }
            char* slab = lds + (2 * ct + (i >> 4)) * SLABX;
            if (MODE == 2) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(slab) + (i & 15) * PS2 + ch) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                uint2 hi, mid, lo;
                split4(v, hi, mid, lo);
                char* d = slab + (i & 15) * XR2 + 2 * ch;
                *reinterpret_cast<uint2*>(d) = hi;
                *reinterpret_cast<uint2*>(d + XP2) = mid;
                *reinterpret_cast<uint2*>(d + 2 * XP2) = lo;
            }
        }
    }
}

template <bool STAMP>
__global__ __launch_bounds__(256, 1) void fused_trunk_x3_kernel(FusedArgs a, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char ldsx[];           // 4 slabs of SLABX bytes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long p_first = (long long)blockIdx.x * 4;
    const long long n_valid = a.count ? (long long)*a.count : a.n;
    if (p_first >= n_valid) return;
    long long pi = p_first + wave;
    if (pi >= n_valid) pi = n_valid - 1;
    if (a.index) pi = a.index[pi];
    char* Sb = ldsx + wave * SLABX;
    float* S = reinterpret_cast<float*>(Sb);
    IPSX_STAMP(0);

    // ---- fp32 input -> zero-padded 38x38 image; zero pixel row of the 8x8 stage
    {
        const float4* src = reinterpret_cast<const float4*>(a.patches + (size_t)pi * 1024);
        float4 px[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) px[k] = src[k * 64 + lane];
        for (int z = lane; z < (PW * PW + 3) / 4; z += 64) reinterpret_cast<float4*>(S)[z] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z = lane; z < XR1 / 4; z += 64) reinterpret_cast<unsigned*>(Sb + XZ1 * XR1)[z] = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = (k * 64 + lane) * 4, y = e >> 5, x = e & 31;
            float* d = S + (y + 3) * PW + x + 3;
            d[0] = px[k].x; d[1] = px[k].y; d[2] = px[k].z; d[3] = px[k].w;
        }
    }
    wave_fence();

    f32x16 idn[2][2], acc[2][2];
    IPSX_STAMP(1);
    stem_pool(a, S, idn, lane);                      // exact fp32 stem + pool (standard layout)
    wave_fence();
    store_stem_x(Sb, idn, lane);
    wave_fence();
    load_l1x(Sb, idn, lane);                         // the identity in the transposed layout (hi + mid + lo is exact)
    IPSX_STAMP(2);

#pragma unroll 1
    for (int blk = 0; blk < 2; ++blk) {
        conv_l1x(a.wh[2 * blk], Sb, acc, lane);
        wave_fence();
        IPSX_STAMP(3 + 4 * blk);
        epilogue_l1x<false>(Sb, a.al[2 * blk], a.sh[2 * blk], acc, idn, lane);
        wave_fence();
        IPSX_STAMP(4 + 4 * blk);
        conv_l1x(a.wh[2 * blk + 1], Sb, acc, lane);
        wave_fence();
        IPSX_STAMP(5 + 4 * blk);
        epilogue_l1x<true>(Sb, a.al[2 * blk + 1], a.sh[2 * blk + 1], acc, idn, lane);
        __syncthreads();
        IPSX_STAMP(6 + 4 * blk);
    }

    f32x16 t2[2], id2[2];
    conv_l2x<64, 8, XR1, XP1, XZ1, 2, 3>(a.wh[4], ldsx, t2, lane, wave);
    conv_l2x<64, 8, XR1, XP1, XZ1, 2, 1>(a.wh_down, ldsx, id2, lane, wave);
    {   // projection shortcut: BatchNorm only, kept in fp32 registers
        const int half = lane >> 5;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = 32 * wave + 8 * g + 4 * half;
            const float4 A = *reinterpret_cast<const float4*>(a.a_down + ch), B = *reinterpret_cast<const float4*>(a.s_down + ch);
            const float Aa[4] = {A.x, A.y, A.z, A.w}, Bb[4] = {B.x, B.y, B.z, B.w};
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int j = 0; j < 4; ++j) id2[ct][4 * g + j] = __builtin_fmaf(id2[ct][4 * g + j], Aa[j], Bb[j]);
        }
    }
    IPSX_STAMP(11);
    __syncthreads();                                  // every wave is done with the 8x8 images
    epilogue_l2x<0>(ldsx, a.al[4], a.sh[4], t2, id2, lane, wave);
    for (int z = lane; z < XR2 / 4; z += 64) reinterpret_cast<unsigned*>(Sb + XZ2 * XR2)[z] = 0u;
    __syncthreads();
#pragma unroll 1
    for (int cv = 5; cv < 8; ++cv) {
        conv_l2x<128, 4, XR2, XP2, XZ2, 1, 3>(a.wh[cv], ldsx, t2, lane, wave);
        __syncthreads();
        if (cv == 5) epilogue_l2x<1>(ldsx, a.al[cv], a.sh[cv], t2, id2, lane, wave);
        else if (cv == 6) epilogue_l2x<0>(ldsx, a.al[cv], a.sh[cv], t2, id2, lane, wave);
        else epilogue_l2x<2>(ldsx, a.al[cv], a.sh[cv], t2, id2, lane, wave);
        __syncthreads();
        IPSX_STAMP(7 + cv);
    }
    for (int o = threadIdx.x; o < 4 * 128; o += 256) {
        const int pl = o >> 7, n = o & 127;
        const float* sp = reinterpret_cast<const float*>(ldsx + pl * SLABX) + n;
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum = sum + sp[k * PS2];
        if (p_first + pl < n_valid) a.emb[(size_t)(p_first + pl) * 128 + n] = sum / 16.0f;
    }
    IPSX_STAMP(15);
}

// OIHW fp32 -> three-plane bf16 A-operand stream [C_out/32][K/16][plane][64 lanes][8]: element j of lane l holds
// term `plane` (0 hi, 1 mid, 2 lo) of the weight at k = 16*step + 8*(l>>5) + j (tap-major k), channel 32*tile + (l&31).
__global__ void pack_conv_weight_x3_kernel(const float* __restrict__ w, int c_out, int c_in, int kh, int kw,
                                           int ksteps, size_t total, unsigned short* __restrict__ packed) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx & 7);
    const int lane = (int)((idx >> 3) & 63);
    const size_t gp = idx >> 9;
    const int pl = (int)(gp % 3);
    const size_t g = gp / 3;
    const int ks = (int)(g % ksteps);
    const int nt = (int)(g / ksteps);
    const int n = nt * 32 + (lane & 31);
    const int k = ks * 16 + 8 * (lane >> 5) + j;
    const int K = kh * kw * c_in;
    float v = 0.0f;
    if (n < c_out && k < K) {
        const int tap = k / c_in, c = k - tap * c_in;
        v = w[((size_t)n * c_in + c) * kh * kw + tap];
    }
    const unsigned short h = bf16_bits(v);
    const float r1 = v - __uint_as_float((unsigned)h << 16);
    const unsigned short m = bf16_bits(r1);
    const unsigned short l = bf16_bits(r1 - __uint_as_float((unsigned)m << 16));
    packed[idx] = pl == 0 ? h : (pl == 1 ? m : l);
}
