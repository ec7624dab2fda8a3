// scan_large.hip - scan_large_kernel / topm_large_kernel: the selection loop and torch.topk for candidate sets beyond one
// compute unit's LDS (the reference's shipped CAMELYON configuration M = I = 5000, config/camelyon_config.yml:35-36: 10,000
// candidates per iteration of architecture/ips_net.py:213-241).

#include <algorithm>
#include <cstdlib>

#include "scan_common.h"

namespace ipsx {

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
    unsigned long long* plog;   // the resident loops' log (scorer.hip)
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
    size_t team_off;       // scan_large_team_kernel: where an image's team block (counters, then the sorted runs) starts in its workspace
    unsigned long long team_ticks;     // ... and the longest wait of one workgroup of a team for another
    int team_trunc;        // ... 1: the ranking from the top halves of the runs when that is provably enough (diagnostic ipsx_dbg_scan_team_trunc(0): never)
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
    const int tid = threadIdx.x, b = blockIdx.x;
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
        __hip_atomic_store(&a.plog[4], (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int* const wword = reinterpret_cast<int*>(smem + tail);           // (the replay's stack: free outside the replay)
    long long ready_known = 0;
    if (a.it0 == 0)
        for (int j = tid; j < m; j += LARGE_NT) mem[j] = j;
    __syncthreads();
    int tie = 0;
    // row maxima in `seg` stretches per row so that every wavefront has one (max is order-free; a NaN wins: max_key)
    const int seg = R >= NW ? 1 : NW / R;
    const int tid_outer = tid;
    for (long long it = a.it0; it < a.it1; ++it) {
        // (nothing that depends on the thread's number is carried round the loop - see scan_large_team_kernel)
        int tid = tid_outer;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
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

#include "scan_large_team.h"

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

static int device_cus() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    return (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
}

// staged logits / exponentials + the tie replay's lists [+ the team's block: counters and sorted runs]
static size_t large_ws_base(int m, int i, int h, int n_token) {
    const size_t Lp = ((size_t)(m + i) + 63) & ~(size_t)63;
    return ((size_t)h * n_token * Lp * 4 + 2 * Lp * 4 + 255) & ~(size_t)255;
}
static bool team_shape(int m, int i, int h, int n_token) {
    return h == 8 && n_token == 1 && m + i > 4096 && m + i <= LARGE_MAX_L;
}
size_t scan_large_ws_per_image(int m, int i, int h, int n_token) {
    const size_t base = large_ws_base(m, i, h, n_token);
    if (!team_shape(m, i, h, n_token)) return base;
    return base + ((team_bytes(std::max(64, next_pow2(m + i)), m) + 255) & ~(size_t)255);
}

// Workgroups scan_large_team_kernel gives ONE image of this shape (1: scan_large_kernel, one workgroup).  g_scan_team:
// -1 the default (8), 0 off, else 2 / 4 / 8 (IPSX_LARGE_TEAM, ipsx_dbg_scan_team).
int scan_large_team(int b, int m, int i, int h, int n_token) {
    static const int env = [] { const char* e = std::getenv("IPSX_LARGE_TEAM"); return e && *e ? std::atoi(e) : -1; }();
    const int want = g_scan_team >= 0 ? g_scan_team : (env >= 0 ? env : 8);
    if (want < 2 || !g_scan_direct || !team_shape(m, i, h, n_token)) return 1;
    const int n2 = std::max(64, next_pow2(m + i));
    int W = want >= 8 ? 8 : (want >= 4 ? 4 : 2);
    while (W > 1 && n2 / (LARGE_NT * W) < 1) W >>= 1;            // (a run is at least 1,024 slots)
    // The teams of a call keep b x W compute units, beside a producer for the whole call - and only the LAST image's loop
    // is on the critical path of a call of several (the others run beside the producer of the images behind them): 16
    // units at most (1 / 16 of an MI355X: two images get teams of 8, four of 4, eight of 2).
    const int cap = std::max(8, device_cus() / 16);
    while (W > 1 && (long long)b * W > cap) W >>= 1;
    return W < 2 ? 1 : W;
}

int scan_large_max_l() { return LARGE_MAX_L; }
size_t topm_large_ws_per_row(int l) { return ((size_t)2 * l * 4 + 255) & ~(size_t)255; }

int launch_scan_large(const ScanCall& c) {
    const float* const logits = c.logits;
    const int b = c.b, m = c.m, i = c.i, h = c.h, n_token = c.n_token, ready_stride = c.ready_stride;
    const int64_t n = c.n, it_begin = c.it_begin, it_end = c.it_end;
    int64_t* const mem_idx = c.mem_idx;
    float* const mem_score = c.mem_score;
    int32_t* const tie_flag = c.tie_flag;
    const int32_t* const ready = c.ready;
    int32_t* const status = c.status;
    void* const workspace = c.workspace;
    const size_t workspace_bytes = c.workspace_bytes;
    void* const stream = c.stream;
    const int32_t* const cond = c.cond;
    const int32_t cond_mask = c.cond_mask;
    const int R = h * n_token, Lmax = m + i, n2 = std::max(64, next_pow2(Lmax));
        IPSX_REQUIRE(!(ready && cond), "scan: a persistent launch is not conditional");
        IPSX_REQUIRE(!ready || status, "scan_persistent: needs the status word");
        IPSX_REQUIRE(Lmax <= LARGE_MAX_L, "scan: M+I = %d candidates - at most %d are supported", Lmax, LARGE_MAX_L);
        IPSX_REQUIRE(R <= 256, "scan: H * n_token = %d > 256 not supported", R);
        const size_t need = (size_t)b * scan_large_ws_per_image(m, i, h, n_token);
        if (!workspace || workspace_bytes < need)
            return fail(IPSX_EWORKSPACE, "scan: M=%d I=%d H=%d n_token=%d needs a workspace of %zu B (ipsx_scan_workspace_bytes), got %zu",
                        m, i, h, n_token, need, workspace_bytes);
        LargeArgs la;
        la.plog = persist_log();
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
        // a team of workgroups per image (scan_large_team.h) - not the conditional recovery launch, which must not wait on anyone
        const int W = cond ? 1 : scan_large_team(b, m, i, h, n_token);
        if (W > 1 && !(g_scan_stamps && n2 / (LARGE_NT * W) != 2)) {
            la.team_off = large_ws_base(m, i, h, n_token);
            la.team_ticks = la.wait_ticks * (ready ? 1ull : 20ull);
            la.team_trunc = g_scan_team_trunc ? 1 : 0;
            unsigned char* ctl0 = la.ws + la.team_off;
            // (the counters, and the runs' words behind them: a key on its way carries its iteration's number - team_key_out)
            if (hipMemset2DAsync(ctl0, la.ws_per_image, 0, (size_t)TEAM_CTL_INTS * 4 + (size_t)n2 * 8, (size_t)b, as_stream(stream)) != hipSuccess)
                return fail(IPSX_EHIP, "scan: clearing the team counters: %s", hipGetErrorString(hipGetLastError()));
            const dim3 grid((unsigned)(b * W)), block(LARGE_NT);
#define IPSX_TEAM_LAUNCH(CPT, ST)                                                                                              \
    do {                                                                                                                        \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_large_team_kernel<CPT, ST>),                               \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                        \
        scan_large_team_kernel<CPT, ST><<<grid, block, lds, as_stream(stream)>>>(la, ST ? g_scan_stamps : nullptr);             \
    } while (0)
            switch (n2 / (LARGE_NT * W)) {
                case 1: IPSX_TEAM_LAUNCH(1, false); break;
                case 2: if (g_scan_stamps) IPSX_TEAM_LAUNCH(2, true); else IPSX_TEAM_LAUNCH(2, false); break;
                case 4: IPSX_TEAM_LAUNCH(4, false); break;
                case 8: IPSX_TEAM_LAUNCH(8, false); break;
                default: return fail(IPSX_EINVAL, "scan: internal - team of %d workgroups for %d slots", W, n2);
            }
#undef IPSX_TEAM_LAUNCH
            return launched("scan");
        }
        if (g_scan_stamps) {                                           // diagnostic build (tools/scan_stamps.py large)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_large_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            scan_large_kernel<true><<<dim3((unsigned)b), dim3(LARGE_NT), lds, as_stream(stream)>>>(la, g_scan_stamps);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_large_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            scan_large_kernel<false><<<dim3((unsigned)b), dim3(LARGE_NT), lds, as_stream(stream)>>>(la, nullptr);
        }
        return launched("scan");
}

int launch_topm_large(const TopmArgs& a, int b, void* workspace, size_t workspace_bytes, void* stream) {
    const size_t need = (size_t)b * topm_large_ws_per_row(a.L);
    if (!workspace || workspace_bytes < need)
        return fail(IPSX_EWORKSPACE, "topm: %d candidates need a workspace of %zu B (ipsx_topm_workspace_bytes), got %zu",
                    a.L, need, workspace_bytes);
    const size_t big = large_key_bytes(a.n2) + LARGE_TAIL_BYTES;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topm_large_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)big);
    topm_large_kernel<<<dim3((unsigned)b), dim3(LARGE_NT), big, as_stream(stream)>>>(
        a, static_cast<unsigned char*>(workspace), topm_large_ws_per_row(a.L));
    return launched("topm");
}

}  // namespace ipsx

using namespace ipsx;

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
