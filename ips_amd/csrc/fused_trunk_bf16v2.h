// fused_trunk_bf16v2.h - the bf16 trunk of BASELINE configs[4], second build (round 6; included by fused_trunk.hip behind
// fused_trunk_split.h, whose stem, packed-weight layout, LDS image layout and 4x4 stage it shares).
//
// What round 5's stamps said about the first build (profiles/r05_fused_stamps_bf16.txt): a wave lives
// 137 k cycles per patch for 36.9 k cycles of its own MFMAs, two waves per SIMD.  A 64 -> 64 convolution of the 8x8 stage
// takes 9.6-12.3 k cycles against 4.6 k of matrix-pipe time and NEVER less than ~8 k - also while the SIMD's other wave is in
// a phase without MFMAs: with wave = patch a K-step is 4 MFMAs (128 pipe cycles) on 2 KB of weights, requested two
// steps ahead - 256 cycles of cover for an L2 round trip of 500+ -, and eight waves asking for 2 KB per 128 cycles each
// is the L1's whole 64 B / clk.  So the pipe idles whenever the two waves of a SIMD are not both in a convolution.
//
// This build, 8x8 stage: wave = (ROW TILE of 32 output channels, PAIR of patches).  A K-step is still 4 MFMAs - the
// pair's 128 pixels are 4 column tiles - but on ONE KB of weights (half the L1 traffic per MFMA), and a weight operand is
// one 16-byte register quad, so the ring holds EIGHT K-steps (requested seven ahead: ~900 pipe cycles of cover even when
// the wave has the pipe to itself) in the registers the first build spent on three.  Activations: 4 ds_read_b128 per
// K-step, two steps ahead.  The two waves of a pair share their patches' images, so the stage ping-pongs between two sets
// of LDS images (convolution reads one, its epilogue writes the other: ONE workgroup barrier per layer) - 75 KB per
// workgroup, two workgroups per unit as before.  The stem stays wave = patch (stem_pool<1> as it is); its fp32 result
// reaches the wave that carries it as the residual identity through the dead input slab.
// The 4x4 stage is the first build's (weights already fetched once per workgroup) with the same deep weight ring.
// Arithmetic: exactly the first build's (same operand rounding, same products; fp32 identity) - the embeddings of the two
// builds are bit-identical (tests/test_hip_kernels.py::test_bf16_trunk_builds_agree).

// A patch's slab: its 8x8 image (65 pixel rows of 144 B = 9,360 B) rounded up to a multiple of 256 B.  The 4x4 stage reads
// TWO patches per ds_read_b128 (lanes 0-15 | 16-31); the LDS serves such a read in groups of 16 lanes that mix the two
// patches, conflict-free only when the second patch's rows fall on the bank slots the first one leaves free - which they do
// when the patches lie a multiple of 256 B apart.  (The first build's 9,360 B put 7 of 16 lanes of every group on a taken
// slot: the 4x4 stage ran at 1.8-2.1 x its matrix-pipe time even with the unit to itself, 2.8 beside a second wave -
// it was LDS-bound.)  The 4x4 images (17 rows of 272 B) get a stride of their own, V2_S2, for the same reason.
constexpr int V2_SLAB = (XL<1>::SLAB + 255) & ~255;      // 9,472 B
constexpr int V2_S2 = ((XZ2 + 1) * XP2 + 255) & ~255;   // 4,864 B: a patch's 4x4 image (16 pixel rows + the zero row)
constexpr int V2_BUF = 4 * V2_SLAB;           // one set of images: the four patches' slabs, one after the other
constexpr int V2_LDS = 2 * V2_BUF;            // 75,776 B per workgroup
static_assert(V2_LDS <= 80 * 1024 && 4 * V2_S2 <= V2_BUF && V2_SLAB >= 16 * PS2 * 4, "two workgroups per unit");

#define V2_SG_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define V2_SG_LDS(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define V2_SG_VMEM(n) __builtin_amdgcn_sched_group_barrier(0x020, n, 0)

// byte offset (within a patch slab) of the pixel row a lane reads for tap `tap` of output pixel 32 h + i: the source pixel,
// or the zero row for a halo tap; + 16 * half: this lane half's 8 channels of a K-step
__device__ __forceinline__ unsigned v2_tap_off(int tap, int h, int i, int half) {
    const int t3 = tap / 3;
    const int dy = t3 - 1, dx = tap - 3 * t3 - 1;
    const int x = i & 7, y = (i >> 3) + 4 * h;
    const bool ok = (unsigned)(x + dx) < 8u && (unsigned)(y + dy) < 8u;
    return (unsigned)((ok ? 32 * h + i + dy * 8 + dx : XZ1) * XP1 + 16 * half);
}

// acc[2 q + h] = output channels 32 rt .. 32 rt + 31 (rows) x pixels 32 h .. 32 h + 31 of patch q of the pair (columns) of
// conv3x3(images at P0, P0 + V2_SLAB) over K = 9 * 64.  wp: the layer's packed weights, wave-uniform; rt: wave-uniform.
__device__ __forceinline__ void conv_l1v2(const void* __restrict__ wp, const char* P0, int rt, f32x16 (&acc)[4], int lane) {
    constexpr int WR = 8, XR = 3, G = 36;          // weight ring (7 K-steps ahead), activation ring (2 ahead), K-steps
    const int i = lane & 31, half = lane >> 5;
    const char* wb = reinterpret_cast<const char*>(wp) + (size_t)rt * G * 1024;      // wave-uniform: scalar base ...
    const unsigned lo = lane * 16;                                                  // ... + the only vector part
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) zero(acc[ct]);
    uint4 wr[WR], xr[XR][4];
#define V2_LOADW(g) wr[(g) % WR] = *reinterpret_cast<const uint4*>(wb + (size_t)((g) < G ? (g) : G - 1) * 1024 + lo)
#define V2_LOADX(g)                                                                                              \
    do {                                                                                                         \
        const int g_ = (g) < G ? (g) : G - 1;                                                                    \
        const char* p0_ = P0 + v2_tap_off(g_ >> 2, 0, i, half) + (g_ & 3) * 32;                                  \
        const char* p1_ = P0 + v2_tap_off(g_ >> 2, 1, i, half) + (g_ & 3) * 32;                                  \
        xr[(g) % XR][0] = *reinterpret_cast<const uint4*>(p0_);                                                  \
        xr[(g) % XR][1] = *reinterpret_cast<const uint4*>(p1_);                                                  \
        xr[(g) % XR][2] = *reinterpret_cast<const uint4*>(p0_ + V2_SLAB);                                        \
        xr[(g) % XR][3] = *reinterpret_cast<const uint4*>(p1_ + V2_SLAB);                                        \
    } while (0)
#pragma unroll
    for (int g = 0; g < WR - 1; ++g) V2_LOADW(g);
    V2_LOADX(0);
    V2_LOADX(1);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        V2_LOADX(g + 2);
        V2_LOADW(g + WR - 1);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = MFMA16(wr[g % WR], xr[g % XR][ct], acc[ct]);
        V2_SG_MFMA(1); V2_SG_LDS(2); V2_SG_MFMA(1); V2_SG_VMEM(1); V2_SG_MFMA(1); V2_SG_LDS(2); V2_SG_MFMA(1);
        SB();
    }
#undef V2_LOADW
#undef V2_LOADX
}

// BatchNorm (+ identity) + ReLU on the wave's tiles, then the bf16 image the next layer reads: channels 32 rt .. of both
// patches of the pair at Q0, Q0 + V2_SLAB.  MODE 0: BN + ReLU; 1: BN + identity + ReLU, identity updated
template <int MODE>
__device__ __forceinline__ void epilogue_l1v2(char* Q0, const float* __restrict__ al, const float* __restrict__ sh, int rt,
                                              const f32x16 (&acc)[4], f32x16 (&idn)[4], int lane) {
    const int i = lane & 31, half = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ch = rt * 32 + 8 * g + 4 * half;
        const float4 A = *reinterpret_cast<const float4*>(al + ch), B = *reinterpret_cast<const float4*>(sh + ch);
        const float Aa[4] = {A.x, A.y, A.z, A.w}, Bb[4] = {B.x, B.y, B.z, B.w};
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = __builtin_fmaf(acc[ct][4 * g + j], Aa[j], Bb[j]);
                if (MODE == 1) x = x + idn[ct][4 * g + j];
                x = x > 0.0f ? x : 0.0f;
                if (MODE == 1) idn[ct][4 * g + j] = x;
                v[j] = x;
            }
            store_planes4<1>(Q0 + (ct >> 1) * V2_SLAB + ((ct & 1) * 32 + i) * XP1 + 2 * ch, XP1, v);
        }
    }
}

// ------------------------------------------------------------------ 4x4 stage: conv_l2s with the deep weight ring
// (same tiles, same products as fused_trunk_split.h conv_l2s<1, ...>: wave = 32 output channels x the four patches' 64
//  pixels; a K-step = 2 MFMAs - 64 pipe cycles - on 1 KB of weights, so the weight ring holds TWELVE K-steps, requested
//  eleven ahead; activations three ahead.  Passes of 12 K-steps: both rings close.)
template <int CIN, int WIN, int RB, int ZR, int STRIDE, int KS, int PSTR>
__device__ __forceinline__ void conv_l2v2(const void* __restrict__ wp, const char* lds, f32x16 (&acc)[2], int lane, int wave) {
    constexpr int TAPS = KS * KS, SPT = CIN / 16, G = TAPS * SPT;          // K-steps per tap, K-steps
    constexpr int TPP = TAPS < 3 ? TAPS : 3, PASS = TPP * SPT;             // a pass = 3 taps (12 or 24 K-steps; 1x1: all 4)
    constexpr int WR = 12, XR = 4, XA = 3;
    constexpr int PAD = KS / 2;
    static_assert(G % PASS == 0 && (G == PASS || (PASS % XR == 0 && PASS % WR == 0)) && XA < SPT, "passes close the rings");
    const int i = lane & 31, half = lane >> 5;
    const int pix = i & 15, oy = pix >> 2, ox = pix & 3;
    const char* wb = reinterpret_cast<const char*>(wp) + (size_t)wave * G * 1024;    // wave-uniform
    const unsigned lo = lane * 16;
    zero(acc[0]); zero(acc[1]);
    uint4 wr[WR], xr[XR][2];
    // this lane's source pixel row for a tap (the zero row for a halo tap): computed once per tap, the K-steps of a tap are
    // immediate offsets from it.  (Byte offsets from `lds`, not pointers: a pointer carried around the pass loop loses its
    // address space and the reads become flat loads - which count in vmcnt too and serialise against the weight ring.)
    auto tap_row = [&](int tap) -> unsigned {
        tap = tap < TAPS ? tap : TAPS - 1;
        const int ky = tap / KS, kx = tap - ky * KS;
        const int iy = oy * STRIDE + ky - PAD, ix = ox * STRIDE + kx - PAD;
        const bool ok = (unsigned)iy < (unsigned)WIN && (unsigned)ix < (unsigned)WIN;
        return (unsigned)((i >> 4) * PSTR + 16 * half + (ok ? iy * WIN + ix : ZR) * RB);
    };
#pragma unroll
    for (int g = 0; g < WR - 1; ++g) wr[g] = *reinterpret_cast<const uint4*>(wb + (size_t)(g < G ? g : G - 1) * 1024 + lo);
    unsigned rows[TPP + 1];
#pragma unroll
    for (int t = 0; t <= TPP; ++t) rows[t] = tap_row(t);
#pragma unroll
    for (int g = 0; g < XA; ++g) {
        xr[g][0] = *reinterpret_cast<const uint4*>(lds + rows[0] + g * 32);
        xr[g][1] = *reinterpret_cast<const uint4*>(lds + rows[0] + g * 32 + 2 * PSTR);
    }
#pragma unroll 1
    for (int g0 = 0; g0 < G; g0 += PASS) {
#pragma unroll
        for (int u = 0; u < PASS; ++u) {
            const int g = g0 + u;
            {   // activations of K-step g + XA: tap (u + XA) / SPT of this pass (the last: the first tap of the next pass)
                const unsigned p = rows[(u + XA) / SPT] + ((u + XA) % SPT) * 32;
                xr[(u + XA) % XR][0] = *reinterpret_cast<const uint4*>(lds + p);
                xr[(u + XA) % XR][1] = *reinterpret_cast<const uint4*>(lds + p + 2 * PSTR);
            }
            wr[(u + WR - 1) % WR] = *reinterpret_cast<const uint4*>(wb + (size_t)(g + WR - 1 < G ? g + WR - 1 : G - 1) * 1024 + lo);
            acc[0] = MFMA16(wr[u % WR], xr[u % XR][0], acc[0]);
            acc[1] = MFMA16(wr[u % WR], xr[u % XR][1], acc[1]);
            V2_SG_MFMA(1); V2_SG_LDS(2); V2_SG_VMEM(1); V2_SG_MFMA(1);
            SB();
        }
        if (G > PASS) {
            const int t0 = (g0 + PASS) / SPT;                               // first tap of the next pass
#pragma unroll
            for (int t = 0; t <= TPP; ++t) rows[t] = tap_row(t0 + t);
        }
    }
}

// input pixels of one patch, 16 per lane: float32 or half-precision storage (2 KiB per patch, 8 bytes per lane and load)
__device__ __forceinline__ void v2_fetch(const FusedArgs& a, long long pi, int lane, float4 (&px)[4]) {
    if (a.in_dtype == 0) {
        const float4* src = reinterpret_cast<const float4*>(a.patches + (size_t)pi * 1024);
#pragma unroll
        for (int k = 0; k < 4; ++k) px[k] = src[k * 64 + lane];
    } else {
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
}

// PERSISTENT: the workgroup takes quads blockIdx.x, blockIdx.x + gridDim.x, ... of the launch (the grid is two workgroups
// per unit): no dispatch gap between one quad and the next, and the next quad's pixels are requested while the 4x4 stage
// of the current one runs.
// epilogue of the 4x4 stage (fused_trunk_split.h epilogue_l2s<1, MODE>'s arithmetic) with this build's patch strides:
// v[ct][r] = channel 32 wave + (r&3) + 8(r>>2) + 4 half of patch 2 ct + (i>>4), pixel i & 15.  MODE 0: BN + ReLU -> bf16 image
// (patches V2_S2 apart);  1: BN + identity + ReLU -> image, identity updated;  2: like 1, stored as fp32 [pix][PS2] (patches
// V2_SLAB apart) for the average pool
template <int MODE>
__device__ __forceinline__ void epilogue_l2v2(char* lds, const float* __restrict__ al, const float* __restrict__ sh,
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
            if (MODE == 2)
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(lds + (2 * ct + (i >> 4)) * V2_SLAB) + (i & 15) * PS2 + ch) =
                    make_float4(v[0], v[1], v[2], v[3]);
            else
                store_planes4<1>(lds + (2 * ct + (i >> 4)) * V2_S2 + (i & 15) * XP2 + 2 * ch, XP2, v);
        }
    }
}

template <bool STAMP>
__device__ __forceinline__ void fused_trunk_bf16v2_body(const FusedArgs& a, unsigned long long* stamps, char* ldsx) {
    constexpr int R1 = XL<1>::R1, R2 = XL<1>::R2;
    const int lane0 = threadIdx.x & 63, wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long n_valid = a.count ? (long long)*a.count : a.n;
    const long long n_quads = (n_valid + 3) / 4;
    char* const buf0 = ldsx;
    char* const buf1 = ldsx + V2_BUF;
    auto patch_of = [&](long long quad) -> long long {
        long long pi = quad * 4 + wave0;
        if (pi >= n_valid) pi = n_valid - 1;
        return a.index ? (long long)a.index[pi] : pi;
    };
    float4 px[4];
    if ((long long)blockIdx.x >= n_quads) return;         // workgroup-uniform
    v2_fetch(a, patch_of(blockIdx.x), lane0, px);
#undef IPSX_STAMP
#define IPSX_STAMP(k)                                                                      \
    do {                                                                                   \
        if (STAMP) {                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                             \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                    \
            if (lane == 0) stamps[((size_t)quad * 4 + wave) * 16 + (k)] = t_;              \
            __builtin_amdgcn_sched_barrier(0);                                             \
        }                                                                                  \
    } while (0)
#pragma unroll 1
    for (long long quad = blockIdx.x; quad < n_quads; quad += gridDim.x) {
    // (lane and wave through an opaque copy per quad: everything derived from them - tap offsets, slab addresses - is
    //  then formed where it is used instead of being carried around the loop in registers the 8x8 stage does not have)
    int lane = lane0, wave = wave0;
    asm volatile("" : "+v"(lane));
    asm volatile("" : "+s"(wave));
    const int rt = wave & 1, pp = wave >> 1;
    char* Sb = buf0 + wave * V2_SLAB;                 // this wave's own patch: input image, transposition scratch, hand-over
    float* S = reinterpret_cast<float*>(Sb);
    const long long p_first = quad * 4;
    IPSX_STAMP(0);

    // ---- input -> bf16 plane of the zero-padded 38 x 38 image (row pitch SPW), in the patch's slab of set 0
    {
        for (int z = lane; z < SPLANE / 16; z += 64) reinterpret_cast<uint4*>(Sb)[z] = make_uint4(0u, 0u, 0u, 0u);
        // the zero (halo) pixel rows of this patch's 8x8 images, both sets
        for (int z = lane; z < R1 / 4; z += 64) {
            reinterpret_cast<unsigned*>(Sb + XZ1 * R1)[z] = 0u;
            reinterpret_cast<unsigned*>(Sb + V2_BUF + XZ1 * R1)[z] = 0u;
        }
        wave_fence();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = (k * 64 + lane) * 4, y = e >> 5, x = e & 31;     // 4 pixels of row y starting at x (x % 4 == 0)
            const unsigned short b0 = bf16_bits(px[k].x), b1 = bf16_bits(px[k].y), b2 = bf16_bits(px[k].z), b3 = bf16_bits(px[k].w);
            char* d = Sb + ((y + 3) * SPW + x + 3) * 2;                    // columns x+3 (odd), x+4..x+5 (aligned pair), x+6
            *reinterpret_cast<unsigned short*>(d) = b0;
            *reinterpret_cast<unsigned*>(d + 2) = (unsigned)b1 | ((unsigned)b2 << 16);
            *reinterpret_cast<unsigned short*>(d + 6) = b3;
        }
    }
    wave_fence();

    f32x16 idn[4], acc[4];
    IPSX_STAMP(1);
    {
        f32x16 st[2][2], tr[2][2];
        stem_pool<1>(a, Sb, st, lane);               // wave = patch: stem on the bf16 pipe + pool on the accumulators
        wave_fence();                                // the input image is dead
        transpose_stem(S, st, tr, lane);             // tr[channel tile][pixel half]: one pixel, 4 consecutive channels per quad
        epilogue_l1s<1, 2>(buf1 + wave * V2_SLAB, nullptr, nullptr, tr, tr, lane);      // the patch's image, all 64 channels
        // the fp32 identity: this wave carries channel tile rt of BOTH patches of its pair - its own patch's (patch rt of
        // the pair) stays in registers, the other channel tile goes to the pair's other wave through this (now dead) slab
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                S[(h * 16 + r) * 64 + lane] = rt ? tr[0][h][r] : tr[1][h][r];
                acc[h][r] = rt ? tr[1][h][r] : tr[0][h][r];
            }
    }
    __syncthreads();
    {
        const float* O = reinterpret_cast<const float*>(buf0 + (wave ^ 1) * V2_SLAB);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float o = O[(h * 16 + r) * 64 + lane];
                idn[h][r] = rt ? o : acc[h][r];
                idn[2 + h][r] = rt ? acc[h][r] : o;
            }
    }
    IPSX_STAMP(2);

    // ---- layer1: two BasicBlocks at 8x8, wave = (channel tile, patch pair); images ping-pong set 1 -> 0 -> 1 -> 0 -> 1
    char* const P1 = buf1 + 2 * pp * V2_SLAB;
    char* const P0 = buf0 + 2 * pp * V2_SLAB;
#pragma unroll 1
    for (int blk = 0; blk < 2; ++blk) {
        conv_l1v2(a.wh[2 * blk], P1, rt, acc, lane);
        IPSX_STAMP(3 + 4 * blk);
        if (blk == 0) __syncthreads();               // set 0 still holds the hand-over of the identities: every wave has read
        epilogue_l1v2<0>(P0, a.al[2 * blk], a.sh[2 * blk], rt, acc, idn, lane);
        __syncthreads();
        IPSX_STAMP(4 + 4 * blk);
        conv_l1v2(a.wh[2 * blk + 1], P0, rt, acc, lane);
        IPSX_STAMP(5 + 4 * blk);
        epilogue_l1v2<1>(P1, a.al[2 * blk + 1], a.sh[2 * blk + 1], rt, acc, idn, lane);
        __syncthreads();
        IPSX_STAMP(6 + 4 * blk);
    }

    // the NEXT quad's pixels: requested here, used at the top of the loop (the 4x4 stage has registers to spare)
    // (unconditionally - a quad past the end re-reads the last patch - so that the registers are dead between their use at
    //  the top of the loop and here)
    v2_fetch(a, patch_of(quad + gridDim.x < n_quads ? quad + gridDim.x : n_quads - 1), lane, px);

    // ---- layer2 (the first build's tiles: wave = 32 output channels x the four patches): 8x8 images in set 1
    f32x16 t2[2], id2[2];
    conv_l2v2<64, 8, R1, XZ1, 2, 3, V2_SLAB>(a.wh[4], buf1, t2, lane, wave);
    conv_l2v2<64, 8, R1, XZ1, 2, 1, V2_SLAB>(a.wh_down, buf1, id2, lane, wave);
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
    epilogue_l2v2<0>(buf0, a.al[4], a.sh[4], t2, id2, lane, wave);          // 4x4 images into set 0 (set 1 is still read)
    for (int z = lane; z < R2 / 4; z += 64) reinterpret_cast<unsigned*>(buf0 + wave * V2_S2 + XZ2 * R2)[z] = 0u;
    __syncthreads();                                  // every wave is done with the 8x8 images
    for (int z = lane; z < R2 / 4; z += 64) reinterpret_cast<unsigned*>(buf1 + wave * V2_S2 + XZ2 * R2)[z] = 0u;
    // l2.0.c2: set 0 -> 1;  l2.1.c1: 1 -> 0;  l2.1.c2: 0 -> fp32 [pix][PS2] in set 1 (the average pool reads it)
    conv_l2v2<128, 4, R2, XZ2, 1, 3, V2_S2>(a.wh[5], buf0, t2, lane, wave);
    epilogue_l2v2<1>(buf1, a.al[5], a.sh[5], t2, id2, lane, wave);
    __syncthreads();
    IPSX_STAMP(12);
    conv_l2v2<128, 4, R2, XZ2, 1, 3, V2_S2>(a.wh[6], buf1, t2, lane, wave);
    epilogue_l2v2<0>(buf0, a.al[6], a.sh[6], t2, id2, lane, wave);
    __syncthreads();
    IPSX_STAMP(13);
    conv_l2v2<128, 4, R2, XZ2, 1, 3, V2_S2>(a.wh[7], buf0, t2, lane, wave);
    epilogue_l2v2<2>(buf1, a.al[7], a.sh[7], t2, id2, lane, wave);
    __syncthreads();
    IPSX_STAMP(14);
    for (int o = threadIdx.x; o < 4 * 128; o += 256) {
        const int pl = o >> 7, n = o & 127;
        const float* sp = reinterpret_cast<const float*>(buf1 + pl * V2_SLAB) + n;
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum = sum + sp[k * PS2];
        if (p_first + pl < n_valid) a.emb[(size_t)(p_first + pl) * 128 + n] = sum / 16.0f;
    }
    IPSX_STAMP(15);
    // (the next quad writes set 0 first - last read by l2.1.c2, behind a barrier - and set 1's zero rows, which the
    //  average pool does not touch; its images in set 1 are written behind the next quad's first barrier... by waves that
    //  may still be summing here: one barrier)
    __syncthreads();
    }
}

template <bool STAMP>
__global__ __launch_bounds__(256, 2) void fused_trunk_bf16v2_kernel(FusedArgs a, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char ldsx[];           // two sets of four slabs
    fused_trunk_bf16v2_body<STAMP>(a, stamps, ldsx);
}
