// fused_trunk_pair.h - the fused 1x32x32 trunk (fused_trunk.hip) for the patches that do not fill the GPU: TWO
// wavefronts per patch.  Included by fused_trunk.hip.
//
// The fused trunk's unit is one patch per wavefront, two wavefronts per SIMD: a launch of up to 2,048 patches (256 CUs)
// takes one patch-pair time whatever its size.  One image of the headline workload is 2,500 patches = 2,048 + 452: the
// 452 would occupy 452 of 1,024 SIMDs for a whole patch time (0.31 ms beside the 0.56 ms of the full round).  Here a
// patch is split over the two wavefronts of a pair BY OUTPUT CHANNEL - wave `nh` owns channels 32 nh .. 32 nh + 31 of the
// stem and of every 64-channel convolution (two m-tiles x one n-tile of accumulators instead of 2 x 2), the slab is
// shared by the pair - and the 4x4 stage gives each of the workgroup's four wavefronts 32 of the 128 channels of its TWO
// patches (one 32-row tile).  Every output's fma chain runs over k in the contract's order exactly as in
// fused_trunk_kernel, so the embeddings are bit-identical to it (tests/test_hip_kernels.py); only the operand traffic per
// MFMA differs (6 loads per 16 MFMAs at 8x8 instead of 4, 8 per 16 at 4x4 instead of 6), which is why this is the
// kernel for a remainder of at most a quarter round (one workgroup per CU, one wavefront per SIMD) and not the kernel.
#pragma once

// ------------------------------------------------------------------ stem + max-pool, one n-tile per wavefront
// stem_pool<0> of fused_trunk.hip with the n-tile fixed to `nh`; idn[mt] = pooled 8x8 activation, channels 32 nh + lane%32.
__device__ __forceinline__ void stem_pool_pair(const FusedArgs& a, const float* S, f32x16 (&idn)[2], int lane, int nh) {
    const int i = lane & 31, half = lane >> 5;
    const int ox = i & 15;
    float bw[28];
    const float4* wp = reinterpret_cast<const float4*>(a.w_stem) + nh * 7 * 64 + lane;
#pragma unroll
    for (int kg = 0; kg < 7; ++kg) {
        const float4 v = wp[kg * 64];
        bw[4 * kg] = v.x; bw[4 * kg + 1] = v.y; bw[4 * kg + 2] = v.z; bw[4 * kg + 3] = v.w;
    }
    const float al = a.a_stem[32 * nh + i], sh = a.s_stem[32 * nh + i];
    float prev[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) prev[x] = -__builtin_huge_valf();

#pragma unroll 1
    for (int t = 0; t < 8; ++t) {
        const int oy = 2 * t + (i >> 4);
        const float* base = S + (2 * oy) * PW + 2 * ox;
        const float* baseN = base + half * 4;
        const float* baseW = base + half * (PW - 3);
        float av[25];
#pragma unroll
        for (int st = 0; st < 25; ++st) {
            const int k0 = 8 * (st >> 2) + (st & 3), ky0 = k0 / 7, kx0 = k0 % 7;
            av[st] = (kx0 <= 2) ? baseN[ky0 * PW + kx0] : baseW[ky0 * PW + kx0];
        }
        av[24] = half ? 0.0f : av[24];
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc;
        zero(acc);
#pragma unroll
        for (int st = 0; st < 25; ++st) acc = MFMA(av[st], bw[st], acc);
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = __builtin_fmaf(acc[r], al, sh);
            v[r] = x > 0.0f ? x : 0.0f;
        }
        float own[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            own[j] = max3(prev[j], v[j], v[8 + j]);
            own[4 + j] = max3(prev[4 + j], v[4 + j], v[12 + j]);
            prev[j] = v[8 + j];
            prev[4 + j] = v[12 + j];
        }
        float L[9];
        L[0] = half ? own[3] : -__builtin_huge_valf();
#pragma unroll
        for (int j = 0; j < 4; ++j) half_swap(own[j], own[4 + j], L[1 + j], L[5 + j]);
        float pooled[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pooled[q] = max3(L[2 * q], L[2 * q + 1], L[2 * q + 2]);
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            if (tt == t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) idn[tt >> 2][4 * (tt & 3) + q] = pooled[q];
            }
        }
    }
}

// the wave's 64 px x 32 ch (C layout) into the pair's slab as [pix][c]
__device__ __forceinline__ void store_l1_pair(float* S, const f32x16 (&v)[2], int lane, int nh) {
    const int i = lane & 31, half = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pix = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            S[pix * PS1 + nh * 32 + i] = v[mt][r];
        }
}

// ------------------------------------------------------------------ 8x8 stage, wave = (patch, n-tile)
// conv3x3 over K = 9 * 64 for the wave's 32 channels and both m-tiles: the stage of conv_l2 (two packed k-groups, 2 tiles)
// with the 8x8 stage's tap addresses.
__device__ __forceinline__ L2Tap l1_tap_pair(int tap, const float* S, int i, int half) {
    const L1Tap t = l1_tap(tap, S, i, half);
    L2Tap d;
    d.s0 = t.s0;
    d.s1 = t.s1;
    return d;
}

__device__ __forceinline__ void conv_l1_pair(const float* __restrict__ wp, const float* S, f32x16 (&acc)[2], int lane, int nh) {
    constexpr int G2 = 36;
    const int i = lane & 31, half = lane >> 5;
    const char* w = reinterpret_cast<const char*>(wp) + (size_t)__builtin_amdgcn_readfirstlane(nh) * 72 * 1024;
    const unsigned lo = lane * 16;
    zero(acc[0]); zero(acc[1]);
    L2Tap cur = l1_tap_pair(0, S, i, half);
    L2Stage sa, sb;
    float4 b0[2], b1[2], b2[2], b3[2];
    l2_loadb<G2>(b0, w, lo, 0);
    l2_loadb<G2>(b1, w, lo, 1);
    l2_load<0>(sa, cur);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const L2Tap nxt = l1_tap_pair(tap < 8 ? tap + 1 : 8, S, i, half);
        const int g = tap * 4;
        l2_load<1>(sb, cur); l2_loadb<G2>(b2, w, lo, g + 2); L2_PRE(); l2_mma(sa, b0, acc); L2_POST();
        l2_load<2>(sa, cur); l2_loadb<G2>(b3, w, lo, g + 3); L2_PRE(); l2_mma(sb, b1, acc); L2_POST();
        l2_load<3>(sb, cur); l2_loadb<G2>(b0, w, lo, g + 4); L2_PRE(); l2_mma(sa, b2, acc); L2_POST();
        l2_load<0>(sa, nxt); l2_loadb<G2>(b1, w, lo, g + 5); L2_PRE(); l2_mma(sb, b3, acc); L2_POST();
        cur = nxt;
    }
}

// ------------------------------------------------------------------ 4x4 stage, 4 waves x 2 patches
// One tile: row i -> patch i >> 4, pixel i & 15; wave `wave` accumulates output channels 32 wave .. 32 wave + 31.
// Stage = KPS packed k-groups: 4 (16 MFMAs) for 128 input channels, 2 (8 MFMAs) for 64 - four stages per tap either way.
template <int KPS>
struct P2Stage {
    float4 a[KPS];
};

template <int KPS, int C>
__device__ __forceinline__ void p2_load(P2Stage<KPS>& st, const float* d) {
#pragma unroll
    for (int q = 0; q < KPS; ++q) st.a[q] = *reinterpret_cast<const float4*>(d + (C * KPS + q) * 8);
}

template <int KPS, int NST>
__device__ __forceinline__ void p2_loadb(float4 (&b)[KPS], const char* wb, unsigned loff, int s) {
    s = s < NST ? s : NST - 1;
    const char* p = wb + (size_t)s * (KPS * 1024);
#pragma unroll
    for (int q = 0; q < KPS; ++q) b[q] = *reinterpret_cast<const float4*>(p + q * 1024 + loff);
}

template <int KPS>
__device__ __forceinline__ void p2_mma(const P2Stage<KPS>& st, const float4 (&b)[KPS], f32x16& acc) {
#pragma unroll
    for (int q = 0; q < KPS; ++q) {
        const float av[4] = {st.a[q].x, st.a[q].y, st.a[q].z, st.a[q].w};
        const float bv[4] = {b[q].x, b[q].y, b[q].z, b[q].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = MFMA(av[j], bv[j], acc);
    }
}

#if IPSX_SPREAD
#define P2_POST4()                                                                                          \
    SG_MFMA(2); SG_LDS(1); SG_MFMA(2); SG_VMEM(1); SG_MFMA(2); SG_LDS(1); SG_MFMA(2); SG_VMEM(1); SG_MFMA(2); \
    SG_LDS(1); SG_MFMA(2); SG_VMEM(1); SG_MFMA(1); SG_LDS(1); SG_MFMA(1); SG_VMEM(1); SG_MFMA(2); SB();
#define P2_POST2() SG_MFMA(2); SG_LDS(1); SG_MFMA(2); SG_VMEM(1); SG_MFMA(1); SG_LDS(1); SG_MFMA(1); SG_VMEM(1); SG_MFMA(2); SB();
#else
#define P2_POST4() SB()
#define P2_POST2() SB()
#endif

template <int CIN, int WIN, int PS, int ZP, int STRIDE, int KS>
__device__ __forceinline__ void conv_l2_pair(const float* __restrict__ wp, const float* lds, f32x16& acc, int lane, int wave) {
    constexpr int KPS = CIN / 32, KGS = KS * KS * CIN / 8, NST = KGS / KPS, TAPS = KS * KS, PAD = KS / 2;
    static_assert(KPS == 2 || KPS == 4, "stage schedule is written for 64 or 128 input channels");
    const int i = lane & 31, half = lane >> 5;
    const int pix = i & 15, oy = pix >> 2, ox = pix & 3;
    const float* S0 = lds + (i >> 4) * SLAB + 4 * half;
    const char* w = reinterpret_cast<const char*>(wp) + (size_t)__builtin_amdgcn_readfirstlane(wave) * KGS * 1024;
    const unsigned lo = lane * 16;
    auto tap_src = [&](int tap) {
        const int ky = tap / KS, kx = tap - ky * KS;
        const int iy = oy * STRIDE + ky - PAD, ix = ox * STRIDE + kx - PAD;
        const bool ok = (unsigned)iy < (unsigned)WIN && (unsigned)ix < (unsigned)WIN;
        return S0 + (ok ? iy * WIN + ix : ZP) * PS;
    };
    zero(acc);
    const float* cur = tap_src(0);
    P2Stage<KPS> sa, sb;
    float4 b0[KPS], b1[KPS], b2[KPS], b3[KPS];
    p2_loadb<KPS, NST>(b0, w, lo, 0);
    p2_loadb<KPS, NST>(b1, w, lo, 1);
    p2_load<KPS, 0>(sa, cur);
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
        const float* nxt = tap_src(tap < TAPS - 1 ? tap + 1 : TAPS - 1);
        const int g = tap * 4;
        if (KPS == 4) {
            p2_load<KPS, 1>(sb, cur); p2_loadb<KPS, NST>(b2, w, lo, g + 2); p2_mma<KPS>(sa, b0, acc); P2_POST4();
            p2_load<KPS, 2>(sa, cur); p2_loadb<KPS, NST>(b3, w, lo, g + 3); p2_mma<KPS>(sb, b1, acc); P2_POST4();
            p2_load<KPS, 3>(sb, cur); p2_loadb<KPS, NST>(b0, w, lo, g + 4); p2_mma<KPS>(sa, b2, acc); P2_POST4();
            p2_load<KPS, 0>(sa, nxt); p2_loadb<KPS, NST>(b1, w, lo, g + 5); p2_mma<KPS>(sb, b3, acc); P2_POST4();
        } else {
            p2_load<KPS, 1>(sb, cur); p2_loadb<KPS, NST>(b2, w, lo, g + 2); p2_mma<KPS>(sa, b0, acc); P2_POST2();
            p2_load<KPS, 2>(sa, cur); p2_loadb<KPS, NST>(b3, w, lo, g + 3); p2_mma<KPS>(sb, b1, acc); P2_POST2();
            p2_load<KPS, 3>(sb, cur); p2_loadb<KPS, NST>(b0, w, lo, g + 4); p2_mma<KPS>(sa, b2, acc); P2_POST2();
            p2_load<KPS, 0>(sa, nxt); p2_loadb<KPS, NST>(b1, w, lo, g + 5); p2_mma<KPS>(sb, b3, acc); P2_POST2();
        }
        cur = nxt;
    }
}

// the wave's tile (C layout) into the two slabs in 4x4-stage layout [pix][c] (row stride PS2)
__device__ __forceinline__ void store_l2_pair(float* lds, const f32x16& v, int lane, int wave) {
    const int i = lane & 31, half = lane >> 5;
    const int n = 32 * wave + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int pl = r >> 3, pix = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;
        lds[pl * SLAB + pix * PS2 + n] = v[r];
    }
}

// two patches p_first, p_first + 1 by the workgroup's four wavefronts; KEEP: the two embeddings are also left in
// lds[0 .. 255] (behind a barrier) for a caller that goes on with them (fused_trunk_stream_kernel: the logits)
template <bool KEEP>
__device__ __forceinline__ void trunk_pair_tile(const FusedArgs& a, long long p_first, float* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ps = wave >> 1, nh = wave & 1;                              // patch slot of the workgroup, n-tile of the pair
    const int i = lane & 31;
    long long pi = p_first + ps;
    if (pi >= a.n) pi = a.n - 1;                                          // odd tail: recompute a valid patch, store nothing
    if (a.index) pi = a.index[pi];
    float* S = lds + ps * SLAB;

    // ---- input patch -> slab as a zero-padded 38x38 image, half of it per wavefront
    {
        const float4* src = reinterpret_cast<const float4*>(a.patches + (size_t)pi * 1024);
        float4 px[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) px[k] = src[(2 * nh + k) * 64 + lane];
        for (int z = lane + 64 * nh; z < (PW * PW + 3) / 4; z += 128) reinterpret_cast<float4*>(S)[z] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z = lane + 64 * nh; z < PS1; z += 128) S[ZP1 * PS1 + z] = 0.0f;
        __syncthreads();                                                  // the padding is laid by both waves of the pair
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = ((2 * nh + k) * 64 + lane) * 4, y = e >> 5, x = e & 31;
            float* d = S + (y + 3) * PW + x + 3;
            d[0] = px[k].x; d[1] = px[k].y; d[2] = px[k].z; d[3] = px[k].w;
        }
    }
    __syncthreads();

    f32x16 idn[2], acc[2];
    stem_pool_pair(a, S, idn, lane, nh);
    __syncthreads();                                                      // the input is dead for both waves
    store_l1_pair(S, idn, lane, nh);
    __syncthreads();

    // ---- layer1: two BasicBlocks at 8x8
#pragma unroll 1
    for (int blk = 0; blk < 2; ++blk) {
        conv_l1_pair(a.w[2 * blk], S, acc, lane, nh);
        {
            const float A = a.al[2 * blk][nh * 32 + i], B = a.sh[2 * blk][nh * 32 + i];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = __builtin_fmaf(acc[mt][r], A, B);
                    acc[mt][r] = v > 0.0f ? v : 0.0f;
                }
        }
        __syncthreads();
        store_l1_pair(S, acc, lane, nh);
        __syncthreads();
        conv_l1_pair(a.w[2 * blk + 1], S, acc, lane, nh);
        {
            const float A = a.al[2 * blk + 1][nh * 32 + i], B = a.sh[2 * blk + 1][nh * 32 + i];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = __builtin_fmaf(acc[mt][r], A, B);
                    v = v + idn[mt][r];
                    idn[mt][r] = v > 0.0f ? v : 0.0f;
                }
        }
        __syncthreads();
        store_l1_pair(S, idn, lane, nh);
        __syncthreads();
    }

    // ---- layer2 block 0: conv3x3/2 (64->128) and the 1x1/2 projection read the 8x8 stage of both patches
    f32x16 t2, id2;
    const int n2 = 32 * wave + i;
    conv_l2_pair<64, 8, PS1, ZP1, 2, 3>(a.w[4], lds, t2, lane, wave);
    conv_l2_pair<64, 8, PS1, ZP1, 2, 1>(a.w_down, lds, id2, lane, wave);
    {
        const float A = a.al[4][n2], B = a.sh[4][n2], Ad = a.a_down[n2], Bd = a.s_down[n2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = __builtin_fmaf(t2[r], A, B);
            t2[r] = v > 0.0f ? v : 0.0f;
            id2[r] = __builtin_fmaf(id2[r], Ad, Bd);
        }
    }
    __syncthreads();
    store_l2_pair(lds, t2, lane, wave);
    if (wave < 2)
        for (int z = lane; z < PS2; z += 64) lds[wave * SLAB + ZP2 * PS2 + z] = 0.0f;   // zero pixel row of the 4x4 stage
    __syncthreads();
#pragma unroll 1
    for (int cv = 5; cv < 8; ++cv) {
        conv_l2_pair<128, 4, PS2, ZP2, 1, 3>(a.w[cv], lds, t2, lane, wave);
        const float A = a.al[cv][n2], B = a.sh[cv][n2];
        const bool plain = (cv == 6);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = __builtin_fmaf(t2[r], A, B);
            if (!plain) v = v + id2[r];
            v = v > 0.0f ? v : 0.0f;
            t2[r] = v;
            if (!plain) id2[r] = v;
        }
        __syncthreads();
        store_l2_pair(lds, t2, lane, wave);
        __syncthreads();
    }

    // ---- global average pool over the 16 pixels, sequential order: one output per thread
    {
        const int pl = threadIdx.x >> 7, n = threadIdx.x & 127;
        const float* s = lds + pl * SLAB + n;
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum = sum + s[k * PS2];
        const float e = sum / 16.0f;
        if (p_first + pl < a.n) a.emb[(size_t)(p_first + pl) * 128 + n] = e;
        if (KEEP) {
            __syncthreads();                                              // every sum has been read
            lds[threadIdx.x] = e;
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256, 2) void fused_trunk_pair_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // 2 slabs
    trunk_pair_tile<false>(a, (long long)blockIdx.x * 2, lds);
}

// ------------------------------------------------------------------ one image as ONE persistent launch
// A single image is not worth cutting into parts for the selection loop to run beside: the loop's
// workgroup has to find a compute unit the trunk's workgroups leave free, and the last part's iterations are exposed.
// Here the loop is resident from the start (ipsx_scan_persistent) and the trunk's workgroups - one per remaining compute
// unit, two patches at a time (trunk_pair_tile: one wavefront per SIMD) - pull patch pairs off a counter, encode them,
// compute their logits (the MFMA sequence of logits_kernel on emb + pos, 2 valid rows of a 32-row tile) and publish them
// the way projector_stream_kernel does: logits written through, a flag per pair, the first wavefront raises the cursor and
// the loop's progress word past every completed pair behind the first unpublished one.
struct TrunkStreamArgs {
    FusedArgs f;
    const float* pos;          // (n, 128) positional encodings added to the embeddings for the logits, or nullptr
    const float* vp;           // folded query, packed for one 32-column tile
    int R;                     // logits per patch (<= 32)
    float* logits;             // (n, R)
    int* ctl;                  // [0] next pair, [1] first unpublished pair, [2 ...] a flag per pair; zeroed by the caller
    int* ready;                // patches published
    unsigned n_pairs;
    int quad_pulls;            // a workgroup's first quad_pulls pulls are four patches, the rest two
};

__global__ __launch_bounds__(256, 1) void fused_trunk_stream_kernel(TrunkStreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // 4 slabs + the pull word behind them
    int* s_pull = reinterpret_cast<int*>(lds + 4 * SLAB);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, i = lane & 31;
    const unsigned n = (unsigned)a.f.n;
    int pulls = 0;
    for (;;) {
        // a workgroup's first quad_pulls pulls are FOUR patches (one wavefront per patch: the trunk's full rate), the
        // rest two (trunk_pair_tile: half the time per pull - what is left of the image is dealt out finely)
        const int take = pulls < a.quad_pulls ? 2 : 1;
        ++pulls;
        if (threadIdx.x == 0) *s_pull = atomicAdd(&a.ctl[0], take);
        __syncthreads();
        const unsigned g = (unsigned)__builtin_amdgcn_readfirstlane(*s_pull);
        if (g >= a.n_pairs) break;                                        // workgroup-uniform
        const unsigned p0 = 2u * g;
        const int pairs = (take == 2 && g + 1 < a.n_pairs) ? 2 : 1;
        if (pairs == 2) trunk_quad_tile<false, true>(a.f, (long long)p0, (long long)n, lds, nullptr);   // lds[0 .. 511] = 4 embeddings
        else trunk_pair_tile<true>(a.f, (long long)p0, lds);                                             // lds[0 .. 255] = 2
        // ---- logits: the first 2 * pairs rows of a 32-row tile (the others carry zeros), K = 128 in the contract's order
        if (wave == 0) {
            const unsigned row = p0 + (unsigned)i;
            const bool rv = i < 2 * pairs && row < n;
            const float* e = lds + (rv ? i : 0) * 128 + 4 * half;
            const float* pp = a.pos ? a.pos + (size_t)(rv ? row : 0) * 128 + 4 * half : nullptr;
            const float4* vq = reinterpret_cast<const float4*>(a.vp) + lane;
            f32x16 lacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) lacc[r] = 0.0f;
            float4 ev[16], bv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                ev[u] = *reinterpret_cast<const float4*>(e + u * 8);
                if (pp) {
                    const float4 pv = *reinterpret_cast<const float4*>(pp + u * 8);
                    ev[u].x = ev[u].x + pv.x; ev[u].y = ev[u].y + pv.y; ev[u].z = ev[u].z + pv.z; ev[u].w = ev[u].w + pv.w;
                }
                bv[u] = vq[(size_t)u * 64];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                lacc = MFMA(rv ? ev[u].x : 0.0f, bv[u].x, lacc);
                lacc = MFMA(rv ? ev[u].y : 0.0f, bv[u].y, lacc);
                lacc = MFMA(rv ? ev[u].z : 0.0f, bv[u].z, lacc);
                lacc = MFMA(rv ? ev[u].w : 0.0f, bv[u].w, lacc);
            }
            // C layout: lane = logit, registers = rows; rows 0 .. 3 are registers 0 .. 3 of lane half 0
            if (half == 0 && i < a.R) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < 2 * pairs && p0 + r < n)
                        __hip_atomic_store(a.logits + (size_t)(p0 + r) * a.R + i, lacc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ---- publish (see projector_stream_kernel): the flags of these pairs, then cursor and progress word past every
            // completed pair behind the first unpublished one
            // (one agent-scope release in the publishing wavefront: the hand-over then rests on the memory model, not only on
            //  the write-through logits having arrived - measured at under 1 % of the stream, projector_stream_kernel)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (lane < pairs) __hip_atomic_store(&a.ctl[2 + g + lane], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (;;) {
                const int p = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&a.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if ((unsigned)p >= a.n_pairs) break;
                const unsigned u = (unsigned)p + lane;
                const int f = u < a.n_pairs ? __hip_atomic_load(&a.ctl[2 + u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                const unsigned long long done = __ballot(f != 0);
                const int c = done == ~0ull ? 64 : __builtin_ctzll(~done);
                if (c == 0) break;
                if (lane == 0) {
                    __hip_atomic_fetch_max(&a.ctl[1], p + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_max(a.ready, (int)min(2u * (unsigned)(p + c), n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (c < 64) break;
            }
        }
        __syncthreads();                                                  // lds[0 .. 511] and the pull word are free again
    }
    // The last workgroup out publishes whatever two simultaneous finishers left to each other (round 5; the caller's
    // ipsx_publish_rows launch behind this one did that): every workgroup has fenced its tiles before it counts itself out.
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (__hip_atomic_fetch_add(&a.ctl[2 + a.n_pairs], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            __hip_atomic_fetch_max(&a.ctl[1], (int)a.n_pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(a.ready, (int)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
