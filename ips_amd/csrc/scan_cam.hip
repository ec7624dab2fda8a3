// scan_cam.hip - scan_cam_kernel: the LDS-resident selection loop specialised for BASELINE configs[3] (CAMELYON features:
// 8 heads, one token, M = I = 256; reference loop architecture/ips_net.py:213-241, scores transformer.py:143-148).

#include <algorithm>
#include <cstdlib>

#include "scan_common.h"

namespace ipsx {

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
            __hip_atomic_store(&a.plog[4], (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                // (round 6: every later member of the run, not only the neighbour - see scan_fast_kernel)
                const int npair = M < Lc - 1 ? M : Lc - 1;
                bool hit = false;
                for (int j = tid; j < npair && !hit; j += NT)
                    for (int k = j + 1; k < Lc && !hit && (sorted[k] >> 32) == (sorted[j] >> 32); ++k) {
                        bool same = true;
                        if (a.tie_order == 1) {
                            const uint4* ra = reinterpret_cast<const uint4*>(xc + key_pos(sorted[j]) * LD);
                            const uint4* rb = reinterpret_cast<const uint4*>(xc + key_pos(sorted[k]) * LD);
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

bool scan_cam_shape(int m, int i, int h, int n_token) { return h * n_token == cam::R && n_token == 1 && m == cam::M && i == cam::I; }

int launch_scan_cam(const ScanCall& c) {
    const int b = c.b;
    void* const stream = c.stream;
    ScanArgs a;
    a.plog = persist_log();
    a.lg = c.logits; a.n = c.n; a.m = c.m; a.i = c.i; a.h = c.h; a.T = c.n_token; a.n2 = next_pow2(c.m + c.i);
    a.it0 = c.it_begin; a.it1 = c.it_end;
    a.mem_idx = reinterpret_cast<long long*>(c.mem_idx); a.mem_score = c.mem_score; a.tie = c.tie_flag;
    a.ready = c.ready; a.status = c.status; a.ready_stride = c.ready_stride;
    a.ready_words = c.ready ? (c.ready_stride ? std::min(c.b, 64) : 1) : 0;
    a.wait_ticks = (unsigned long long)g_persist_wait_ms * 100000ull;
    a.cond = c.cond; a.cond_mask = c.cond_mask;
    a.slides = c.b;
    a.tie_order = g_tie_order;
    a.use_lds = 1;
    unsigned long long* st = g_scan_stamps;
    {
        // BASELINE configs[3] (8 heads, one token, M = I = 256): the specialised loop (scan_cam_kernel)
        static_assert(cam::LDS_BYTES <= 160 * 1024, "scan_cam_kernel: LDS");
        a.stk_off = cam::OFF_STK;
        const int grid = c.workgroups > 0 && c.workgroups < b ? c.workgroups : b;
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
}

}  // namespace ipsx
