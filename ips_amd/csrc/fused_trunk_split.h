// fused_trunk_split.h - the fused trunk on the bf16 matrix pipe (included by fused_trunk.hip), in two flavours of one
// template:
//
//   PL = 3, precision 2 "fp32x3": every fp32 operand of the residual stages is carried as THREE bf16 terms.
//     The fp32 matrix pipe tops out at 157 TFLOP/s, the bf16 pipe at 2.5 PFLOP/s.  A float splits EXACTLY into
//     hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (3 x 8 significand bits, fp32's
//     exponent range), so x*w is a sum of 9 bf16 products; the three smallest (mid*lo, lo*mid, lo*lo) are below
//     2^-24 |x*w| - under the rounding of the fp32 accumulator itself - and are dropped.  The other SIX run on
//     v_mfma_f32_32x32x16_bf16 (exact products, fp32 accumulation): 6 x 32 cycles per 16 k against 8 x 64 for
//     v_mfma_f32_32x32x2_f32, 2.67x less matrix-pipe time at the accuracy of an fp32 chain (tests/test_hip_kernels.py::
//     test_fp32x3_trunk_has_fp32_accuracy).  Not bit-identical to the contract's fma chains: tolerance-tested, opt-in.
//   PL = 1, precision 1 "bf16" (BASELINE configs[4]): operands rounded to bf16, one product, fp32 accumulation.  The
//     reference has no reduced-precision path; this one is tolerance-tested against the fp32 kernel and a float64
//     emulation that rounds at the same places.
//
// Structure = the fp32 kernel's (wave = patch in the 8x8 stage, 4 waves x 4 patches in the 4x4 stage; the stem runs on
// the same pipe with a row-padded K, see stem_pool; identity and BatchNorm / residual / ReLU in fp32 registers), with
// the contraction of the residual stages TRANSPOSED: weights are the
// A operand (rows = output channels), activations the B operand (columns = pixels).  A lane then owns one pixel and 4
// CONSECUTIVE channels per register quad, so the epilogue packs 4 bf16 per plane and stores 8 bytes at a time.
//   LDS image   [pixel][plane][C bf16 + 8 pad]   8x8: PL x 144 B per pixel row, 4x4: PL x 272 B; one zero row (halo)
//   weights     [row tile][K/16][plane][64 lanes][8 bf16]  (ipsx_pack_conv_weight_x3 / _bf16), 4.0 / 1.35 MB
//   stage       one K-step of 16 in the 8x8 stage: 2 PL ds_read_b128 + 2 PL global_load_dwordx4 + 4 NP MFMA (NP = 6 or
//               1 products), the loads of the following stage spread between the MFMAs with sched_group_barrier
//               (tools/ubench/x3_stage.hip: 833-843 cycles per stage against 768 of pure pipe time; 901 in front)
// LDS per workgroup: 112 KB (PL = 3: one workgroup = one wave per SIMD per CU) / 37 KB (PL = 1).

constexpr int XP1 = 144, XZ1 = 64;                    // 8x8 stage: bytes per plane per pixel row (64 bf16 + 8 pad); zero row
constexpr int XP2 = 272, XZ2 = 16;                    // 4x4 stage

template <int PL> struct XL {
    static constexpr int R1 = PL * XP1, R2 = PL * XP2;            // bytes per pixel row
    static constexpr int SLAB = (XZ1 + 1) * R1;                   // 28,080 B (PL = 3) / 9,360 B (PL = 1) per patch
    static constexpr int NP = PL == 3 ? 6 : 1;                    // bf16 products per fp32 product
    static_assert(SLAB >= PL * SPLANE && SLAB >= (XZ2 + 1) * R2 && SLAB >= 16 * PS2 * 4 && SLAB >= 32 * PS1 * 4 &&
                  SLAB % 16 == 0, "slab");
};

__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xFFFF0000u); }
// (a, b) -> two bfloat16 in one register, a in the low half: ONE v_cvt_pk_bf16_f32 (written as two scalar conversions and an
// or, the compiler paired the wrong elements and repaired the order with four more instructions per pair)
typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const pk_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pk_bf16x2));
}

// 4 consecutive channels of one pixel -> PL planes, 4 bf16 (8 bytes) per plane; PL = 3: hi + mid + lo == x exactly
template <int PL>
__device__ __forceinline__ void store_planes4(char* d, int plane_bytes, const float (&x)[4]) {
    uint2 hi;
    hi.x = pk_bf16(x[0], x[1]); hi.y = pk_bf16(x[2], x[3]);
    *reinterpret_cast<uint2*>(d) = hi;
    if (PL == 3) {
        const float r0 = x[0] - bf_lo(hi.x), r1 = x[1] - bf_hi(hi.x), r2 = x[2] - bf_lo(hi.y), r3 = x[3] - bf_hi(hi.y);
        uint2 mid, lo;
        mid.x = pk_bf16(r0, r1); mid.y = pk_bf16(r2, r3);
        lo.x = pk_bf16(r0 - bf_lo(mid.x), r1 - bf_hi(mid.x)); lo.y = pk_bf16(r2 - bf_lo(mid.y), r3 - bf_hi(mid.y));
        *reinterpret_cast<uint2*>(d + plane_bytes) = mid;
        *reinterpret_cast<uint2*>(d + 2 * plane_bytes) = lo;
    }
}

// stem output (standard C layout: in[mt][nt][r] = channel 32nt + i of pixel 32mt + (r&3) + 8(r>>2) + 4half) -> transposed
// layout of the 8x8 stage (out[rt][ct][r] = channel 32rt + (r&3) + 8(r>>2) + 4half of pixel 32ct + i), through an fp32
// [32 pixels][PS1] image in the slab, one pixel half at a time (8.7 KB: fits the one-plane slab too): 32 conflict-free
// ds_write_b32 and 8 ds_read_b128 per half.  Exact: the identity stays fp32.
__device__ __forceinline__ void transpose_stem(float* S, const f32x16 (&in)[2][2], f32x16 (&out)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * half) * PS1 + nt * 32 + i] = in[ct][nt][r];
        wave_fence();
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = *reinterpret_cast<const float4*>(S + i * PS1 + rt * 32 + 8 * g + 4 * half);
                out[rt][ct][4 * g] = v.x; out[rt][ct][4 * g + 1] = v.y; out[rt][ct][4 * g + 2] = v.z; out[rt][ct][4 * g + 3] = v.w;
            }
        wave_fence();
    }
}

// 8x8 stage: BatchNorm (+ identity) + ReLU on the transposed tile, then planes.  MODE 0: BN + ReLU; 1: BN + identity +
// ReLU, identity updated; 2: no arithmetic (the stem's output), just the planes
template <int PL, int MODE>
__device__ __forceinline__ void epilogue_l1s(char* S, const float* __restrict__ al, const float* __restrict__ sh,
                                             const f32x16 (&acc)[2][2], f32x16 (&idn)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = rt * 32 + 8 * g + 4 * half;
            float Aa[4] = {1.f, 1.f, 1.f, 1.f}, Bb[4] = {0.f, 0.f, 0.f, 0.f};
            if (MODE != 2) {
                const float4 A = *reinterpret_cast<const float4*>(al + ch), B = *reinterpret_cast<const float4*>(sh + ch);
                Aa[0] = A.x; Aa[1] = A.y; Aa[2] = A.z; Aa[3] = A.w; Bb[0] = B.x; Bb[1] = B.y; Bb[2] = B.z; Bb[3] = B.w;
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = idn[rt][ct][4 * g + j];
                    if (MODE != 2) {
                        x = __builtin_fmaf(acc[rt][ct][4 * g + j], Aa[j], Bb[j]);
                        if (MODE == 1) x = x + idn[rt][ct][4 * g + j];
                        x = x > 0.0f ? x : 0.0f;
                        if (MODE == 1) idn[rt][ct][4 * g + j] = x;
                    }
                    v[j] = x;
                }
                store_planes4<PL>(S + (ct * 32 + i) * XL<PL>::R1 + 2 * ch, XP1, v);
            }
        }
}

// ------------------------------------------------------------------ 8x8 stage, wave = patch
struct XTap { const char* s0; const char* s1; };
template <int PL> struct XOp1 { uint4 p[2][PL]; };    // [tile][plane] operand registers of one K-step

template <int PL>
__device__ __forceinline__ XTap x1_tap(int tap, const char* S, int i, int half) {
    const int t3 = tap / 3;
    const int dy = t3 - 1, dx = tap - 3 * t3 - 1;
    const int x = i & 7, y0 = i >> 3;
    const bool okx = (unsigned)(x + dx) < 8u;
    const bool ok0 = okx && (unsigned)(y0 + dy) < 8u;
    const bool ok1 = okx && (unsigned)(y0 + 4 + dy) < 8u;
    const int p0 = i + dy * 8 + dx;
    XTap d;
    d.s0 = S + (ok0 ? p0 : XZ1) * XL<PL>::R1 + 16 * half;
    d.s1 = S + (ok1 ? p0 + 32 : XZ1) * XL<PL>::R1 + 16 * half;
    return d;
}

template <int PL, int KS>
__device__ __forceinline__ void x1_load(XOp1<PL>& st, const XTap& d) {
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) {
        st.p[0][pl] = *reinterpret_cast<const uint4*>(d.s0 + pl * XP1 + KS * 32);
        st.p[1][pl] = *reinterpret_cast<const uint4*>(d.s1 + pl * XP1 + KS * 32);
    }
}

template <int PL>
__device__ __forceinline__ void x1_loadw(XOp1<PL>& w, const char* wb, unsigned loff, int g) {
    g = g < 36 ? g : 35;
    const char* p = wb + (size_t)g * (PL * 1024);
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) {
        w.p[0][pl] = *reinterpret_cast<const uint4*>(p + pl * 1024 + loff);
        w.p[1][pl] = *reinterpret_cast<const uint4*>(p + (size_t)36 * (PL * 1024) + pl * 1024 + loff);
    }
}

template <int PL>
__device__ __forceinline__ void x1_mma(const XOp1<PL>& w, const XOp1<PL>& x, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int q = 0; q < XL<PL>::NP; ++q)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = MFMA16(w.p[rt][pair_w(PL, q)], x.p[ct][pair_x(PL, q)], acc[rt][ct]);
}

#define SG_MFMA_(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define SG_LDS_(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define SG_VMEM_(n) __builtin_amdgcn_sched_group_barrier(0x020, n, 0)
// PL = 3: 24 MFMAs, 6 + 6 loads - a load every 1.5 MFMAs, 6 MFMAs at the end;  PL = 1: 4 MFMAs, 2 + 2 loads
#define X1_GROUPS()                                                                                   \
    if (PL == 3) {                                                                                    \
        _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) { SG_MFMA_(2); SG_LDS_(1); SG_MFMA_(1); SG_VMEM_(1); } \
        SG_MFMA_(6);                                                                                  \
    } else {                                                                                          \
        SG_MFMA_(1); SG_LDS_(1); SG_VMEM_(1); SG_MFMA_(1); SG_LDS_(1); SG_VMEM_(1); SG_MFMA_(2);        \
    }
#define X1_STAGE(XC, LOADN, WC, WF) \
    LOADN; x1_loadw<PL>(WF, w, lo, g + 2); x1_mma<PL>(WC, XC, acc); X1_GROUPS(); SB(); ++g;

template <int PL>
__device__ __forceinline__ void conv_l1s(const void* __restrict__ wp, const char* S, f32x16 (&acc)[2][2], int lane) {
    const int i = lane & 31, half = lane >> 5;
    const char* w = reinterpret_cast<const char*>(wp);
    const unsigned lo = lane * 16;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) zero(acc[rt][ct]);
    XTap cur = x1_tap<PL>(0, S, i, half);
    XOp1<PL> xa, xb, w0, w1, w2;                   // weight ring of 3: slot = stage % 3, refilled 2 stages ahead
    x1_loadw<PL>(w0, w, lo, 0);
    x1_loadw<PL>(w1, w, lo, 1);
    x1_load<PL, 0>(xa, cur);
    int g = 0;
#pragma unroll 1
    for (int tap = 0; tap < 9; tap += 3) {         // 3 taps = 12 stages per trip: the ring closes
        XTap nxt = x1_tap<PL>(tap + 1, S, i, half);
        X1_STAGE(xa, (x1_load<PL, 1>(xb, cur)), w0, w2) X1_STAGE(xb, (x1_load<PL, 2>(xa, cur)), w1, w0)
        X1_STAGE(xa, (x1_load<PL, 3>(xb, cur)), w2, w1) X1_STAGE(xb, (x1_load<PL, 0>(xa, nxt)), w0, w2)
        cur = nxt; nxt = x1_tap<PL>(tap + 2, S, i, half);
        X1_STAGE(xa, (x1_load<PL, 1>(xb, cur)), w1, w0) X1_STAGE(xb, (x1_load<PL, 2>(xa, cur)), w2, w1)
        X1_STAGE(xa, (x1_load<PL, 3>(xb, cur)), w0, w2) X1_STAGE(xb, (x1_load<PL, 0>(xa, nxt)), w1, w0)
        cur = nxt; nxt = x1_tap<PL>(tap + 3 < 9 ? tap + 3 : 8, S, i, half);
        X1_STAGE(xa, (x1_load<PL, 1>(xb, cur)), w2, w1) X1_STAGE(xb, (x1_load<PL, 2>(xa, cur)), w0, w2)
        X1_STAGE(xa, (x1_load<PL, 3>(xb, cur)), w1, w0) X1_STAGE(xb, (x1_load<PL, 0>(xa, nxt)), w2, w1)
        cur = nxt;
    }
}

// ------------------------------------------------------------------ 4x4 stage, 4 waves x 4 patches
// Column tile ct = patches 2ct, 2ct+1 (column i -> patch 2ct + (i>>4), pixel i & 15); wave `wave` owns the output
// channels 32*wave .. 32*wave+31 (row tile) for both column tiles, so each weight is fetched once per workgroup.
// SK K-steps per stage so that a tap is always 4 stages.
template <int PL, int SK> struct XOp2 { uint4 x[2][SK][PL]; };     // activations [ct][k-step][plane]
template <int PL, int SK> struct XW2 { uint4 w[SK][PL]; };         // weights [k-step][plane]

template <int PL, int WIN, int RB, int ZR, int STRIDE, int KS>
__device__ __forceinline__ XTap x2_tap(int tap, const char* S0, int oy, int ox) {
    constexpr int PAD = KS / 2;
    const int ky = tap / KS, kx = tap - ky * KS;
    const int iy = oy * STRIDE + ky - PAD, ix = ox * STRIDE + kx - PAD;
    const bool ok = (unsigned)iy < (unsigned)WIN && (unsigned)ix < (unsigned)WIN;
    XTap d;
    d.s0 = S0 + (ok ? iy * WIN + ix : ZR) * RB;
    d.s1 = d.s0 + 2 * XL<PL>::SLAB;
    return d;
}

template <int PL, int SK, int PB, int CS>
__device__ __forceinline__ void x2_load(XOp2<PL, SK>& st, const XTap& d) {
#pragma unroll
    for (int q = 0; q < SK; ++q)
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) {
            st.x[0][q][pl] = *reinterpret_cast<const uint4*>(d.s0 + pl * PB + (CS * SK + q) * 32);
            st.x[1][q][pl] = *reinterpret_cast<const uint4*>(d.s1 + pl * PB + (CS * SK + q) * 32);
        }
}

template <int PL, int SK, int G>
__device__ __forceinline__ void x2_loadw(XW2<PL, SK>& b, const char* wb, unsigned loff, int g) {
    g = g < G ? g : G - 1;
    const char* p = wb + (size_t)g * SK * (PL * 1024);
#pragma unroll
    for (int q = 0; q < SK; ++q)
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) b.w[q][pl] = *reinterpret_cast<const uint4*>(p + q * (PL * 1024) + pl * 1024 + loff);
}

template <int PL, int SK>
__device__ __forceinline__ void x2_mma(const XW2<PL, SK>& b, const XOp2<PL, SK>& st, f32x16 (&acc)[2]) {
#pragma unroll
    for (int q = 0; q < SK; ++q)
#pragma unroll
        for (int t = 0; t < XL<PL>::NP; ++t) {
            acc[0] = MFMA16(b.w[q][pair_w(PL, t)], st.x[0][q][pair_x(PL, t)], acc[0]);
            acc[1] = MFMA16(b.w[q][pair_w(PL, t)], st.x[1][q][pair_x(PL, t)], acc[1]);
        }
}

// per stage: 2 NP SK MFMAs, 2 PL SK ds_read_b128, PL SK global loads
#define X2_GROUPS()                                                                                          \
    if (PL == 3) {                                                                                           \
        _Pragma("unroll") for (int q_ = 0; q_ < 3 * SK; ++q_) { SG_MFMA_(2); SG_LDS_(2); SG_MFMA_(1); SG_VMEM_(1); } \
        SG_MFMA_(3 * SK);                                                                                    \
    } else {                                                                                                 \
        _Pragma("unroll") for (int q_ = 0; q_ < SK; ++q_) { SG_MFMA_(1); SG_LDS_(2); SG_VMEM_(1); SG_MFMA_(1); }   \
    }
#define X2_STAGE(XC, LOADN, WC, WF) \
    LOADN; x2_loadw<PL, SK, G>(WF, w, lo, g + 2); x2_mma<PL, SK>(WC, XC, acc); X2_GROUPS(); SB(); ++g;

template <int PL, int CIN, int WIN, int RB, int PB, int ZR, int STRIDE, int KS>
__device__ __forceinline__ void conv_l2s(const void* __restrict__ wp, const char* lds, f32x16 (&acc)[2], int lane,
                                         int wave) {
    constexpr int SK = CIN / 64, TAPS = KS * KS, KSTEPS = TAPS * CIN / 16, G = KSTEPS / SK;   // G = 4 * TAPS stages
    const int i = lane & 31, half = lane >> 5;
    const int pix = i & 15, oy = pix >> 2, ox = pix & 3;
    const char* S0 = lds + (i >> 4) * XL<PL>::SLAB + 16 * half;
    const char* w = reinterpret_cast<const char*>(wp) + (size_t)__builtin_amdgcn_readfirstlane(wave) * KSTEPS * (PL * 1024);
    const unsigned lo = lane * 16;
    zero(acc[0]); zero(acc[1]);
    XTap cur = x2_tap<PL, WIN, RB, ZR, STRIDE, KS>(0, S0, oy, ox);
    XOp2<PL, SK> xa, xb;
    XW2<PL, SK> w0, w1, w2;
    x2_loadw<PL, SK, G>(w0, w, lo, 0);
    x2_loadw<PL, SK, G>(w1, w, lo, 1);
    x2_load<PL, SK, PB, 0>(xa, cur);
    int g = 0;
    if (TAPS == 1) {                               // 1x1 projection: 4 stages, ring of 3 walked once
        X2_STAGE(xa, (x2_load<PL, SK, PB, 1>(xb, cur)), w0, w2) X2_STAGE(xb, (x2_load<PL, SK, PB, 2>(xa, cur)), w1, w0)
        X2_STAGE(xa, (x2_load<PL, SK, PB, 3>(xb, cur)), w2, w1) X2_STAGE(xb, (x2_load<PL, SK, PB, 3>(xa, cur)), w0, w2)
        return;
    }
#pragma unroll 1
    for (int tap = 0; tap < TAPS; tap += 3) {      // 3 taps = 12 stages per trip: the ring of 3 closes
        XTap nxt = x2_tap<PL, WIN, RB, ZR, STRIDE, KS>(tap + 1, S0, oy, ox);
        X2_STAGE(xa, (x2_load<PL, SK, PB, 1>(xb, cur)), w0, w2) X2_STAGE(xb, (x2_load<PL, SK, PB, 2>(xa, cur)), w1, w0)
        X2_STAGE(xa, (x2_load<PL, SK, PB, 3>(xb, cur)), w2, w1) X2_STAGE(xb, (x2_load<PL, SK, PB, 0>(xa, nxt)), w0, w2)
        cur = nxt; nxt = x2_tap<PL, WIN, RB, ZR, STRIDE, KS>(tap + 2, S0, oy, ox);
        X2_STAGE(xa, (x2_load<PL, SK, PB, 1>(xb, cur)), w1, w0) X2_STAGE(xb, (x2_load<PL, SK, PB, 2>(xa, cur)), w2, w1)
        X2_STAGE(xa, (x2_load<PL, SK, PB, 3>(xb, cur)), w0, w2) X2_STAGE(xb, (x2_load<PL, SK, PB, 0>(xa, nxt)), w1, w0)
        cur = nxt; nxt = x2_tap<PL, WIN, RB, ZR, STRIDE, KS>(tap + 3 < TAPS ? tap + 3 : TAPS - 1, S0, oy, ox);
        X2_STAGE(xa, (x2_load<PL, SK, PB, 1>(xb, cur)), w2, w1) X2_STAGE(xb, (x2_load<PL, SK, PB, 2>(xa, cur)), w0, w2)
        X2_STAGE(xa, (x2_load<PL, SK, PB, 3>(xb, cur)), w1, w0) X2_STAGE(xb, (x2_load<PL, SK, PB, 0>(xa, nxt)), w2, w1)
        cur = nxt;
    }
}

// transposed tile of the 4x4 stage: v[ct][r] = channel 32wave + (r&3) + 8(r>>2) + 4half of patch 2ct + (i>>4), pixel i&15
// MODE 0: BN + ReLU -> planes;  1: BN + identity + ReLU -> planes, identity updated;  2: like 1 but the result is
// stored as fp32 [pix][PS2] for the average pool
template <int PL, int MODE>
__device__ __forceinline__ void epilogue_l2s(char* lds, const float* __restrict__ al, const float* __restrict__ sh,
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
            char* slab = lds + (2 * ct + (i >> 4)) * XL<PL>::SLAB;
            if (MODE == 2)
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(slab) + (i & 15) * PS2 + ch) = make_float4(v[0], v[1], v[2], v[3]);
            else
                store_planes4<PL>(slab + (i & 15) * XL<PL>::R2 + 2 * ch, XP2, v);
        }
    }
}

template <int PL, bool STAMP>
__device__ __forceinline__ void fused_trunk_split_body(const FusedArgs& a, unsigned long long* stamps, char* ldsx) {
    constexpr int SLABX = XL<PL>::SLAB, R1 = XL<PL>::R1, R2 = XL<PL>::R2;
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

    // ---- fp32 input -> PL bf16 planes of the zero-padded 38 x 38 image (row pitch SPW)
    {
        float4 px[4];
        if (a.in_dtype == 0) {
            const float4* src = reinterpret_cast<const float4*>(a.patches + (size_t)pi * 1024);
#pragma unroll
            for (int k = 0; k < 4; ++k) px[k] = src[k * 64 + lane];
        } else {                                     // half-precision storage: 2 KiB per patch, 8 bytes per lane and load
            const uint2* src = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.patches) + (size_t)pi * 1024);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint2 h = src[k * 64 + lane];
                const unsigned short hs[4] = {(unsigned short)(h.x & 0xFFFFu), (unsigned short)(h.x >> 16),
                                              (unsigned short)(h.y & 0xFFFFu), (unsigned short)(h.y >> 16)};
                float f[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    f[j] = a.in_dtype == 1 ? __uint_as_float((unsigned)hs[j] << 16)
                                           : (float)__builtin_bit_cast(_Float16, hs[j]);
                px[k] = make_float4(f[0], f[1], f[2], f[3]);
            }
        }
        for (int z = lane; z < PL * SPLANE / 16; z += 64) reinterpret_cast<uint4*>(Sb)[z] = make_uint4(0u, 0u, 0u, 0u);
        wave_fence();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = (k * 64 + lane) * 4, y = e >> 5, x = e & 31;     // 4 pixels of row y starting at x (x % 4 == 0)
            const float v[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
            unsigned short pl_bits[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned short h = bf16_bits(v[j]);
                const float r1 = v[j] - __uint_as_float((unsigned)h << 16);
                const unsigned short m = bf16_bits(r1);
                pl_bits[0][j] = h; pl_bits[1][j] = m; pl_bits[2][j] = bf16_bits(r1 - __uint_as_float((unsigned)m << 16));
            }
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) {                             // columns x+3 (odd), x+4..x+5 (aligned pair), x+6
                char* d = Sb + pl * SPLANE + ((y + 3) * SPW + x + 3) * 2;
                *reinterpret_cast<unsigned short*>(d) = pl_bits[pl][0];
                *reinterpret_cast<unsigned*>(d + 2) = (unsigned)pl_bits[pl][1] | ((unsigned)pl_bits[pl][2] << 16);
                *reinterpret_cast<unsigned short*>(d + 6) = pl_bits[pl][3];
            }
        }
    }
    wave_fence();

    f32x16 idn[2][2], acc[2][2];
    IPSX_STAMP(1);
    stem_pool<PL>(a, Sb, idn, lane);                 // stem on the bf16 pipe + pool on the accumulators (standard layout)
    wave_fence();                                    // the input image is dead
    transpose_stem(S, idn, acc, lane);               // the identity: one pixel and 4 consecutive channels per lane quad
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) idn[rt][ct] = acc[rt][ct];
    epilogue_l1s<PL, 2>(Sb, nullptr, nullptr, acc, idn, lane);
    for (int z = lane; z < R1 / 4; z += 64) reinterpret_cast<unsigned*>(Sb + XZ1 * R1)[z] = 0u;    // zero pixel row
    wave_fence();
    IPSX_STAMP(2);

#pragma unroll 1
    for (int blk = 0; blk < 2; ++blk) {
        conv_l1s<PL>(a.wh[2 * blk], Sb, acc, lane);
        wave_fence();
        IPSX_STAMP(3 + 4 * blk);
        epilogue_l1s<PL, 0>(Sb, a.al[2 * blk], a.sh[2 * blk], acc, idn, lane);
        wave_fence();
        IPSX_STAMP(4 + 4 * blk);
        conv_l1s<PL>(a.wh[2 * blk + 1], Sb, acc, lane);
        wave_fence();
        IPSX_STAMP(5 + 4 * blk);
        epilogue_l1s<PL, 1>(Sb, a.al[2 * blk + 1], a.sh[2 * blk + 1], acc, idn, lane);
        __syncthreads();
        IPSX_STAMP(6 + 4 * blk);
    }

    f32x16 t2[2], id2[2];
    conv_l2s<PL, 64, 8, R1, XP1, XZ1, 2, 3>(a.wh[4], ldsx, t2, lane, wave);
    conv_l2s<PL, 64, 8, R1, XP1, XZ1, 2, 1>(a.wh_down, ldsx, id2, lane, wave);
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
    epilogue_l2s<PL, 0>(ldsx, a.al[4], a.sh[4], t2, id2, lane, wave);
    for (int z = lane; z < R2 / 4; z += 64) reinterpret_cast<unsigned*>(Sb + XZ2 * R2)[z] = 0u;
    __syncthreads();
#pragma unroll 1
    for (int cv = 5; cv < 8; ++cv) {
        conv_l2s<PL, 128, 4, R2, XP2, XZ2, 1, 3>(a.wh[cv], ldsx, t2, lane, wave);
        __syncthreads();
        if (cv == 5) epilogue_l2s<PL, 1>(ldsx, a.al[cv], a.sh[cv], t2, id2, lane, wave);
        else if (cv == 6) epilogue_l2s<PL, 0>(ldsx, a.al[cv], a.sh[cv], t2, id2, lane, wave);
        else epilogue_l2s<PL, 2>(ldsx, a.al[cv], a.sh[cv], t2, id2, lane, wave);
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

template <bool STAMP>
__global__ __launch_bounds__(256, 1) void fused_trunk_x3_kernel(FusedArgs a, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char ldsx[];           // 4 slabs of XL<3>::SLAB bytes
    fused_trunk_split_body<3, STAMP>(a, stamps, ldsx);
}

template <bool STAMP>
__global__ __launch_bounds__(256, 2) void fused_trunk_bf16_kernel(FusedArgs a, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char ldsx[];           // 4 slabs of XL<1>::SLAB bytes
    fused_trunk_split_body<1, STAMP>(a, stamps, ldsx);
}

// OIHW fp32 -> bf16 A-operand stream [C_out/32][K/16][plane][64 lanes][8] with PL planes: element j of lane l holds term
// `plane` (PL = 3: 0 hi, 1 mid, 2 lo; PL = 1: the rounded value) of the weight at k = 16*step + 8*(l>>5) + j (tap-major k),
// channel 32*tile + (l&31); round to nearest even.
template <int PL>
__global__ void pack_conv_weight_split_kernel(const float* __restrict__ w, int c_out, int c_in, int kh, int kw,
                                              int ksteps, size_t total, unsigned short* __restrict__ packed) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx & 7);
    const int lane = (int)((idx >> 3) & 63);
    const size_t gp = idx >> 9;
    const int pl = (int)(gp % PL);
    const size_t g = gp / PL;
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

// Stem weights (C_out, 1, 7, 7) fp32 -> B-operand stream of the split stem [C_out/32][4 K-steps][plane][64 lanes][8]:
// element j of lane l holds term `plane` of w[32 tile + (l&31)][ky = 2 step + (l>>5)][kx = j], zero for ky = 7 or kx = 7.
template <int PL>
__global__ void pack_stem_weight_split_kernel(const float* __restrict__ w, int c_out, size_t total,
                                              unsigned short* __restrict__ packed) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx & 7);
    const int lane = (int)((idx >> 3) & 63);
    const size_t gp = idx >> 9;
    const int pl = (int)(gp % PL);
    const size_t g = gp / PL;
    const int ks = (int)(g & 3), nt = (int)(g >> 2);
    const int n = nt * 32 + (lane & 31), ky = 2 * ks + (lane >> 5), kx = j;
    const float v = (n < c_out && ky < 7 && kx < 7) ? w[(size_t)n * 49 + ky * 7 + kx] : 0.0f;
    const unsigned short h = bf16_bits(v);
    const float r1 = v - __uint_as_float((unsigned)h << 16);
    const unsigned short m = bf16_bits(r1);
    const unsigned short l = bf16_bits(r1 - __uint_as_float((unsigned)m << 16));
    packed[idx] = pl == 0 ? h : (pl == 1 ? m : l);
}
