// fused_trunk_bf16v3.h - the bf16 trunk, third build (round 6; included by fused_trunk.hip behind fused_trunk_bf16v2.h, whose
// 8x8 stage - conv_l1v2 / epilogue_l1v2: wave = (channel tile, patch pair), deep weight ring - and stem hand-over it keeps).
//
// What the second build's stamps left open: its 4x4 stage streams 1 KB of weights per 2 MFMAs and wave - four
// waves fill the unit's 64 B / clk L1 path - because a workgroup's four patches are only 64 pixels = TWO column tiles
// per weight operand; 44 % of a patch's matrix work ran at 1.8 x its pipe time alone and 2.9 x beside a second workgroup.
// Here a workgroup takes EIGHT patches: two quads go through stem + layer1 one after the other (the 8x8 images in place
// this time - convolution reads, barrier, epilogue overwrites, barrier: 75 KB hold all eight patches' images, the first
// quad's final images resting while the second quad works), and the 4x4 stage then runs ONCE over all eight: 128 pixels =
// FOUR column tiles per weight operand - a K-step is 4 MFMAs on the same 1 KB: half the weight traffic per patch, and the
// 8x8 stage's loads-per-MFMA.  The images of the 4x4 stage ping-pong between two sets of eight (2 x 38 KB).
// Arithmetic: the first and second builds' (same operand rounding, same products in the same order, fp32 identity) -
// bit-identical embeddings (tests/test_hip_kernels.py::test_bf16_trunk_builds_agree).

constexpr int V3_LDS = 16 * V2_S2;                  // 77,824 B: two sets of eight 4x4 images; the eight 8x8 slabs need 75,776
static_assert(8 * V2_SLAB <= V3_LDS && V3_LDS <= 80 * 1024, "two workgroups per unit");

// 4x4 stage over EIGHT patches: wave = 32 output channels x 128 pixels - column tile ct = patches 2 ct, 2 ct + 1 (column
// i -> patch 2 ct + (i >> 4), pixel i & 15); the second build's conv_l2v2 with four column tiles per weight operand.
template <int CIN, int WIN, int RB, int ZR, int STRIDE, int KS, int PSTR>
__device__ __forceinline__ void conv_l2v3(const void* __restrict__ wp, const char* lds, f32x16 (&acc)[4], int lane, int wave) {
    constexpr int TAPS = KS * KS, SPT = CIN / 16, G = TAPS * SPT;          // K-steps per tap, K-steps
    constexpr int TPP = TAPS < 3 ? TAPS : 3, PASS = TPP * SPT;             // a pass = 3 taps (12 or 24 K-steps; 1x1: all 4)
    constexpr int WR = 6, XR = 3, XA = 2;                                  // a K-step is 128 pipe cycles here: 5 ahead = 640
    constexpr int PAD = KS / 2;
    static_assert(G % PASS == 0 && (G == PASS || (PASS % XR == 0 && PASS % WR == 0)) && XA < SPT, "passes close the rings");
    const int i = lane & 31, half = lane >> 5;
    const int pix = i & 15, oy = pix >> 2, ox = pix & 3;
    const char* wb = reinterpret_cast<const char*>(wp) + (size_t)wave * G * 1024;    // wave-uniform
    const unsigned lo = lane * 16;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) zero(acc[ct]);
    uint4 wr[WR], xr[XR][4];
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
    for (int g = 0; g < XA; ++g)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) xr[g][ct] = *reinterpret_cast<const uint4*>(lds + rows[0] + g * 32 + ct * 2 * PSTR);
#pragma unroll 1
    for (int g0 = 0; g0 < G; g0 += PASS) {
#pragma unroll
        for (int u = 0; u < PASS; ++u) {
            const int g = g0 + u;
            {
                const unsigned p = rows[(u + XA) / SPT] + ((u + XA) % SPT) * 32;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) xr[(u + XA) % XR][ct] = *reinterpret_cast<const uint4*>(lds + p + ct * 2 * PSTR);
            }
            wr[(u + WR - 1) % WR] = *reinterpret_cast<const uint4*>(wb + (size_t)(g + WR - 1 < G ? g + WR - 1 : G - 1) * 1024 + lo);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = MFMA16(wr[u % WR], xr[u % XR][ct], acc[ct]);
            V2_SG_MFMA(1); V2_SG_LDS(2); V2_SG_MFMA(1); V2_SG_VMEM(1); V2_SG_MFMA(1); V2_SG_LDS(2); V2_SG_MFMA(1);
            SB();
        }
        if (G > PASS) {
            const int t0 = (g0 + PASS) / SPT;
#pragma unroll
            for (int t = 0; t <= TPP; ++t) rows[t] = tap_row(t0 + t);
        }
    }
}

// epilogue of the 4x4 stage over eight patches (epilogue_l2v2's arithmetic on four column tiles)
template <int MODE>
__device__ __forceinline__ void epilogue_l2v3(char* lds, const float* __restrict__ al, const float* __restrict__ sh,
                                              const f32x16 (&acc)[4], f32x16 (&id2)[4], int lane, int wave) {
    const int i = lane & 31, half = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ch = 32 * wave + 8 * g + 4 * half;
        const float4 A = *reinterpret_cast<const float4*>(al + ch), B = *reinterpret_cast<const float4*>(sh + ch);
        const float Aa[4] = {A.x, A.y, A.z, A.w}, Bb[4] = {B.x, B.y, B.z, B.w};
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
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

// STAMP: the diagnostic build (tools/fused_stamps.py): the phase boundaries of quad q on row 8 * workgroup + 4 q + wave; the
// 4x4 stage's (stamps 11 .. 15) on both quads' rows - the SECOND quad's rows (4 .. 7 of every 8) read like the other builds'
template <bool STAMP>
__global__ __launch_bounds__(256, 2) void fused_trunk_bf16v3_kernel(FusedArgs a, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char ldsx[];
#undef IPSX_STAMP
#define IPSX_STAMP(k)                                                                      \
    do {                                                                                   \
        if (STAMP) {                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                             \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                    \
            if (lane == 0) stamps[((size_t)blockIdx.x * 8 + 4 * q + wave) * 16 + (k)] = t_; \
            __builtin_amdgcn_sched_barrier(0);                                             \
        }                                                                                  \
    } while (0)
#define IPSX_STAMP2(k)                                                                     \
    do {                                                                                   \
        if (STAMP) {                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                             \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                    \
            if (lane == 0) {                                                               \
                stamps[((size_t)blockIdx.x * 8 + wave) * 16 + (k)] = t_;                   \
                stamps[((size_t)blockIdx.x * 8 + 4 + wave) * 16 + (k)] = t_;               \
            }                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                             \
        }                                                                                  \
    } while (0)
    constexpr int R1 = XL<1>::R1, R2 = XL<1>::R2;
    const int lane0 = threadIdx.x & 63, wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long n_valid = a.count ? (long long)*a.count : a.n;
    const long long p_first = (long long)blockIdx.x * 8;
    if (p_first >= n_valid) return;                      // workgroup-uniform

    // ---- two quads through stem + layer1, one after the other; the 8x8 images in place
#pragma unroll 1
    for (int q = 0; q < 2; ++q) {
        // (lane and wave through an opaque copy per quad: what is derived from them - tap offsets, slab addresses - is formed
        //  where it is used instead of being carried round the loop in registers the 8x8 stage does not have)
        int lane = lane0, wave = wave0;
        asm volatile("" : "+v"(lane));
        asm volatile("" : "+s"(wave));
        const int rt = wave & 1, pp = wave >> 1;
        char* const base = ldsx + 4 * q * V2_SLAB;       // this quad's four slabs
        char* const Sb = base + wave * V2_SLAB;          // this wave's own patch: input image, transposition scratch, hand-over
        float* const S = reinterpret_cast<float*>(Sb);
        long long pi = p_first + 4 * q + wave;
        if (pi >= n_valid) pi = n_valid - 1;             // tail: recompute a valid patch, store nothing
        if (a.index) pi = a.index[pi];
        IPSX_STAMP(0);
        {
            float4 px[4];
            v2_fetch(a, pi, lane, px);
            for (int z = lane; z < SPLANE / 16; z += 64) reinterpret_cast<uint4*>(Sb)[z] = make_uint4(0u, 0u, 0u, 0u);
            wave_fence();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = (k * 64 + lane) * 4, y = e >> 5, x = e & 31;
                const unsigned short b0 = bf16_bits(px[k].x), b1 = bf16_bits(px[k].y), b2 = bf16_bits(px[k].z), b3 = bf16_bits(px[k].w);
                char* d = Sb + ((y + 3) * SPW + x + 3) * 2;
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
            stem_pool<1>(a, Sb, st, lane);
            wave_fence();                                // the input image is dead
            transpose_stem(S, st, tr, lane);
            // the other channel tile of this patch's fp32 stem output: to the pair's other wave through this slab
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[(h * 16 + r) * 64 + lane] = rt ? tr[0][h][r] : tr[1][h][r];
            __syncthreads();
            {
                const float* O = reinterpret_cast<const float*>(base + (wave ^ 1) * V2_SLAB);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float o = O[(h * 16 + r) * 64 + lane];
                        const float own = rt ? tr[1][h][r] : tr[0][h][r];
                        idn[h][r] = rt ? o : own;
                        idn[2 + h][r] = rt ? own : o;
                    }
            }
            __syncthreads();                             // both hand-overs are read: the slabs become images
            epilogue_l1s<1, 2>(Sb, nullptr, nullptr, tr, tr, lane);      // the patch's image, all 64 channels, in its own slab
            for (int z = lane; z < R1 / 4; z += 64) reinterpret_cast<unsigned*>(Sb + XZ1 * R1)[z] = 0u;
        }
        __syncthreads();
        IPSX_STAMP(2);
        char* const P = base + 2 * pp * V2_SLAB;         // the pair's two slabs
#pragma unroll 1
        for (int blk = 0; blk < 2; ++blk) {
            conv_l1v2(a.wh[2 * blk], P, rt, acc, lane);
            IPSX_STAMP(3 + 4 * blk);
            __syncthreads();                             // every wave has read the images: overwrite them
            epilogue_l1v2<0>(P, a.al[2 * blk], a.sh[2 * blk], rt, acc, idn, lane);
            __syncthreads();
            IPSX_STAMP(4 + 4 * blk);
            conv_l1v2(a.wh[2 * blk + 1], P, rt, acc, lane);
            IPSX_STAMP(5 + 4 * blk);
            __syncthreads();
            epilogue_l1v2<1>(P, a.al[2 * blk + 1], a.sh[2 * blk + 1], rt, acc, idn, lane);
            __syncthreads();
            IPSX_STAMP(6 + 4 * blk);
        }
    }

    // ---- layer2 over the eight patches: wave = 32 output channels x 128 pixels
    int lane = lane0, wave = wave0;
    asm volatile("" : "+v"(lane));
    asm volatile("" : "+s"(wave));
    char* const setX = ldsx;
    char* const setY = ldsx + 8 * V2_S2;
    f32x16 t2[4], id2[4];
    conv_l2v3<64, 8, R1, XZ1, 2, 3, V2_SLAB>(a.wh[4], ldsx, t2, lane, wave);
    conv_l2v3<64, 8, R1, XZ1, 2, 1, V2_SLAB>(a.wh_down, ldsx, id2, lane, wave);
    {   // projection shortcut: BatchNorm only, kept in fp32 registers
        const int half = lane >> 5;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = 32 * wave + 8 * g + 4 * half;
            const float4 A = *reinterpret_cast<const float4*>(a.a_down + ch), B = *reinterpret_cast<const float4*>(a.s_down + ch);
            const float Aa[4] = {A.x, A.y, A.z, A.w}, Bb[4] = {B.x, B.y, B.z, B.w};
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int j = 0; j < 4; ++j) id2[ct][4 * g + j] = __builtin_fmaf(id2[ct][4 * g + j], Aa[j], Bb[j]);
        }
    }
    IPSX_STAMP2(11);
    __syncthreads();                                     // every wave is done with the 8x8 images: their space is the 4x4 stage's
    epilogue_l2v3<0>(setX, a.al[4], a.sh[4], t2, id2, lane, wave);
    for (int p = wave; p < 8; p += 4)                    // the zero (halo) rows of both sets' 4x4 images
        for (int z = lane; z < R2 / 4; z += 64) {
            reinterpret_cast<unsigned*>(setX + p * V2_S2 + XZ2 * R2)[z] = 0u;
            reinterpret_cast<unsigned*>(setY + p * V2_S2 + XZ2 * R2)[z] = 0u;
        }
    __syncthreads();
    conv_l2v3<128, 4, R2, XZ2, 1, 3, V2_S2>(a.wh[5], setX, t2, lane, wave);
    epilogue_l2v3<1>(setY, a.al[5], a.sh[5], t2, id2, lane, wave);
    __syncthreads();
    IPSX_STAMP2(12);
    conv_l2v3<128, 4, R2, XZ2, 1, 3, V2_S2>(a.wh[6], setY, t2, lane, wave);
    epilogue_l2v3<0>(setX, a.al[6], a.sh[6], t2, id2, lane, wave);
    __syncthreads();
    IPSX_STAMP2(13);
    conv_l2v3<128, 4, R2, XZ2, 1, 3, V2_S2>(a.wh[7], setX, t2, lane, wave);
    __syncthreads();                                     // the fp32 images of the average pool take the whole space
    epilogue_l2v3<2>(ldsx, a.al[7], a.sh[7], t2, id2, lane, wave);
    __syncthreads();
    IPSX_STAMP2(14);
    for (int o = threadIdx.x; o < 8 * 128; o += 256) {
        const int pl = o >> 7, n = o & 127;
        const float* sp = reinterpret_cast<const float*>(ldsx + pl * V2_SLAB) + n;
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum = sum + sp[k * PS2];
        if (p_first + pl < n_valid) a.emb[(size_t)(p_first + pl) * 128 + n] = sum / 16.0f;
    }
    IPSX_STAMP2(15);
}
