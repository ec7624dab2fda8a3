// scan_large_team.h - scan_large_team_kernel: the selection loop for candidate sets beyond one compute unit's LDS with a TEAM
// of workgroups per image (round 6; included by scan_large.hip, which holds the sort, the tie replay and LargeArgs).
//
// One workgroup took 92 us per iteration at the reference's shipped CAMELYON sizes (M = I = 5000, config/camelyon_config.yml:
// 35-36; the loop of architecture/ips_net.py:213-241): 47 us of element-wise passes over 10,000 candidates x 8 logits - bound
// by ONE compute unit's gather rate, not by anything the chip lacks - and 41 us of merge sort over 16,384 slots.  Here W
// workgroups (2, 4 or 8; on W compute units) share an image:
//
//   every workgroup   owns the candidates of the 64-blocks  w, w + W, w + 2 W, ...  (memory and chunk candidates alike):
//                     gathers their 8 logits ONCE (they stay in registers), takes its row maxima [hop A: all to all]
//                     exponentials -> workspace (transposed, for the row sums) [hop B: all to all]
//   workgroup w       the denominators of the rows [w * 8 / W, (w + 1) * 8 / W) in the pinned order of the contract - lane j
//                     of a wavefront adds elements j, j + 64, ... ascending, then the butterfly: 512 sequential chains that
//                     cannot be cut, but whose LOADS can (all wavefronts load, the sum is handed on) [hop C: all to all]
//   every workgroup   scores and ranking keys of its candidates from the registers, SORTS them (its run: n2 / W slots, in
//                     its own LDS) and publishes the sorted run [hop D: all to all]
//   every workgroup   RANKS its own run among all candidates - a key's rank is its place in its run plus, for every other
//                     run, the number of keys there that are larger (a binary search per run) - and writes the patches of
//                     its ranks below m into the new memory (a second buffer: the old one is still being read)
//                     [hop E: all to all - the next iteration's memory rows]
//   main (w = 0)      only when an iteration is flagged - two of the first m + 1 ranks tie in a way that calls for
//                     torch.topk's order, or the runs' top halves turned out not to be enough - merges the runs and
//                     replays the order alone, the way scan_large_kernel ends an iteration (team_redo) [hop E2]
//
// Same values, same order of every sum, same keys as scan_large_kernel (keys are unique, so ranks are ranks whoever counts
// them): tools/scan_team_check.py and tests/test_hip_scan_team.py hold the two against each other.
//
// A hop = release + a counter in the workspace (monotonic within a launch: target = arrivals per iteration x the iteration's
// number; zeroed by the launch) + acquire (hops B, E) - ONE L2 write-back and ONE invalidate per workgroup, not one per
// wavefront: team_arrive, TEAM_WAIT; where the data can carry the iteration's number
// itself - hops A and C: 8-64 words, epoch << 32 | value; hop D: the sorted runs, team_key_out - it is ONE relaxed store
// per word, polled by its readers, and there is no fence and no counter.  Every wait is bounded: a persistent launch reports
// through its status word like every resident loop (the recovery launch, always one workgroup per image, redoes the
// work), a plain launch traps.  Workgroups of a team are neighbours in the grid, so the in-order dispatcher never holds a
// team's first members on units its last members need.

constexpr int TEAM_CTL_INTS = 512;                 // per image, at LargeArgs::team_off of its workspace; the runs follow
constexpr int TC_RESIDENT = 0, TC_A = 32, TC_B = 64, TC_C = 96, TC_D = 128, TC_E = 160;     // (a 128-byte line each)
constexpr int TC_PMAX = 192;                       // 8 workgroups x 8 row maxima: 64-bit words, epoch << 32 | order-preserving key
constexpr int TC_RDEN = 320;                       // 8 reciprocal denominators: epoch << 32 | float bits
constexpr int TC_FLAG = 384;                       // the last iteration (its number + 1) whose ranking the main workgroup must redo alone
constexpr int TC_TIE = 448;                        // the last iteration (its number + 1) with equal scores at ranks m - 1 and m
constexpr int TC_E2 = 416;                         // ... how many of those it has redone
// counters | the W sorted runs (n2 keys) | the memory's second buffer (m patch numbers: an iteration reads one, writes the other)
static size_t team_bytes(int n2, int m) { return (size_t)TEAM_CTL_INTS * 4 + (size_t)n2 * 8 + (((size_t)m * 8 + 255) & ~(size_t)255); }

// candidates of workgroup w of W among the first L: the 64-blocks w, w + W, ...
__device__ __forceinline__ int team_count(int L, int w, int W) {
    const int full = L >> 6, rem = L & 63;
    return (full > w ? (full - w + W - 1) / W * 64 : 0) + ((rem != 0 && full % W == w) ? rem : 0);
}

// wave 0: wait until *word >= target (relaxed agent-scope polls); false after `ticks` of the 100 MHz clock
__device__ __forceinline__ bool team_poll(const int* word, int target, unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > ticks) return false;
    }
    return true;
}

// every thread: what it stored before is visible to whoever sees the counter move (all threads call; one barrier)
// (ONE L2 write-back per workgroup, not sixteen: every wavefront waits until its own stores have reached the L2 - the compute
//  unit's waves share it -, the barrier collects them, and the first thread's release covers what the barrier collected)
__device__ __forceinline__ void team_arrive(int* word) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(word, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// keys[0, n2) (padded slots from slot `first_slot` of the dynamic LDS, a 17-slot boundary; `pre` at int `pre_word` of it) = n2 / run sorted runs (descending, real keys
// first, zeros behind; run w holds pre[w + 1] - pre[w] real keys): merged in place by the workgroup, log2(n2 / run) rounds
// of the merge step of sort_desc_large.
__device__ __attribute__((noinline)) void merge_sorted_runs(int first_slot, int n2, int run, int pre_word) {
    // (offsets, not pointers: a pointer handed to a function that is not inlined is a generic one - flat loads instead of LDS loads)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem) + first_slot;
    const int* pre = reinterpret_cast<const int*>(smem) + pre_word;
    const int tid = threadIdx.x;
    const bool act = tid < (n2 >> 4);
    const int o = tid * 16;
    const int rsh = 31 - __clz(run);
    uint64_t k[16];
    int steps = rsh + 1;
    for (int len = run; len < n2; len <<= 1, ++steps) {
        __syncthreads();
        const int base = o & ~(2 * len - 1), diag = o - base;
        const int p0 = pre[base >> rsh], p1 = pre[(base + len) >> rsh], p2 = pre[(base + 2 * len) >> rsh];
        const int cA = p1 - p0, cB = p2 - p1;
        const bool pad = diag >= cA + cB;
        if (act && pad) {
#pragma unroll
            for (int c = 0; c < 16; ++c) k[c] = 0ull;
        }
        if (act && !pad) {
            const int bA = base, bB = base + len;
            int lo = diag > cB ? diag - cB : 0, hi = diag < cA ? diag : cA;
            for (int it = 0; it < steps; ++it) {
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
                if (ai + c >= cA) av[c] = 0ull;
                if (bi + c >= cB) bv[c] = 0ull;
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) k[c] = av[c] > bv[15 - c] ? av[c] : bv[15 - c];
#pragma unroll
            for (int j = 8; j >= 1; j >>= 1)
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if ((c & j) == 0) {
                        const uint64_t x_ = k[c], y_ = k[c | j];
                        const bool sw_ = x_ < y_;
                        k[c] = sw_ ? y_ : x_;
                        k[c | j] = sw_ ? x_ : y_;
                    }
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int c = 0; c < 16; ++c) keys[17 * tid + c] = k[c];
        }
    }
    __syncthreads();
}

// The workgroup's run of 1024 * CPT keys - key c of thread tid is element tid + 1024 c, in registers - sorted descending
// into keys[0, RUN) (padded slots; zeros = no candidate, last): every wavefront sorts its 64-key groups in registers
// (wave_sort_desc), then log2(RUN / 64) rounds merge neighbouring runs BY RANK - an element's place in the merged pair is
// its place in its own run plus the number of elements of the other run that go before it (the A side counts greater, the
// B side greater-or-equal, so equal zeros get places of their own).  Every thread of the workgroup has its CPT searches
// in flight; a search is the branch-free lower bound over a power-of-two run (log2(len) + 1 reads, five instructions a
// step - the loop is bound by instruction issue, not by LDS); the rounds ping-pong between two UNPADDED buffers (X at slot
// 0, Y behind the padded result), the last round writes the padded layout the merge of the runs expects.
// sort_desc_large on the same run has two wavefronts at work (16 outputs per thread): 40 k cycles for 2,048 slots.
template <int CPT>
__device__ __forceinline__ void team_sort_run(uint64_t (&key)[CPT], uint64_t* keys, int tid, int lane) {
    constexpr int RUN = LARGE_NT * CPT;
    constexpr int ROUNDS = CPT == 1 ? 4 : (CPT == 2 ? 5 : (CPT == 4 ? 6 : 7));
    constexpr int YOFF = RUN + (RUN >> 4);
    int src = (ROUNDS & 1) ? YOFF : 0;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        key[c] = wave_sort_desc(key[c], lane);
        keys[src + tid + c * LARGE_NT] = key[c];
    }
    int round = 0;
    for (int len = 64; len < RUN; len <<= 1, ++round) {
        __syncthreads();
        int pos[CPT];
        const uint64_t* other[CPT];
        uint64_t bias[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int p = tid + c * LARGE_NT;
            if (len > 64) key[c] = keys[src + p];
            const bool inA = (p & len) == 0;
            other[c] = keys + src + (p & ~(2 * len - 1)) + (inA ? len : 0);
            // "o goes before key": A side o > key, B side o >= key = o > key - 1 (a real key is never 0; for a zero of the B
            // side key - 1 wraps to the largest value and nothing goes before it - but every A element must: see below)
            bias[c] = inA ? key[c] : key[c] - 1ull;
            pos[c] = 0;
        }
        for (int half = len >> 1; half >= 1; half >>= 1) {
#pragma unroll
            for (int c = 0; c < CPT; ++c) pos[c] += other[c][pos[c] + half - 1] > bias[c] ? half : 0;
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c) pos[c] += other[c][pos[c]] > bias[c] ? 1 : 0;
        const bool last = 2 * len == RUN;
        const int dst = last ? 0 : (src ? 0 : YOFF);
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int p = tid + c * LARGE_NT;
            const bool inA = (p & len) == 0;
            // (a zero of the B side goes behind all of A: its place is its own index + len)
            const int rank = (!inA && key[c] == 0ull) ? len : pos[c];
            const int o = (p & ~(2 * len - 1)) + (p & (len - 1)) + rank;
            keys[dst + (last ? large_slot(o) : o)] = key[c];
        }
        src = dst;
    }
    __syncthreads();
}

// A ranking key on its way through the workspace carries the iteration's number in the 18 bits of its low word that are
// all ones in every key (the low word is 0xFFFFFFFF - position, positions below 16,384): data and flag in ONE 64-bit word,
// one relaxed agent-scope store, polled by whoever needs it - hop D has no fence and no counter.  The launch clears the
// runs' words, a key's score word is never 0, and a slot is rewritten every iteration: a word that carries this
// iteration's number is this iteration's key.
__device__ __forceinline__ uint64_t team_key_out(uint64_t key, uint32_t tag) { return key ^ ((uint64_t)(0x3FFFFu ^ (tag & 0x3FFFFu)) << 14); }
__device__ __forceinline__ bool team_key_in(uint64_t word, uint32_t tag, uint64_t* key) {
    *key = word | (0x3FFFFull << 14);
    return (uint32_t)((word >> 14) & 0x3FFFFull) == (tag & 0x3FFFFu) && (word >> 32) != 0ull;
}

// wave 0: wait until the W * 8 (or 8) words at `words` all carry epoch e in their upper halves; the lower halves of the
// lanes' words come back in `low`.  Data and flag are ONE 64-bit word, written with one relaxed agent-scope store: no fence.
__device__ __forceinline__ bool team_poll_words(const unsigned long long* words, int n, int e, unsigned long long ticks,
                                                int lane, uint32_t* low) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned long long v = lane < n ? __hip_atomic_load(words + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        if (__ballot(lane < n && (int)(v >> 32) != e) == 0ull) { *low = (uint32_t)v; return true; }
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > ticks) return false;
    }
}

// The main workgroup alone: the ranking of iteration `itf` again, from the W sorted runs in the workspace, the way
// scan_large_kernel ends an iteration - all runs merged, the tie test, torch.topk's order replayed where the loops' rule asks
// for it, the new memory (all m slots: what the team wrote from its own ranks is overwritten).  -> the boundary tie bit.
template <int CPT>
__device__ __attribute__((noinline)) int team_redo(const LargeArgs& a, long long itf, int b, int W, const float* lg, long long* mem_out,
                                                   long long* mem_alt, float* xT, int* lists, const uint64_t* gkeys, int tail) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RUN = LARGE_NT * CPT;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    int* const pre = reinterpret_cast<int*>(smem + tail) + 8 + (LARGE_NT / 64) * 8;
    const int tid = threadIdx.x, m = a.m;
    const long long lo = itf * a.i + m;
    const int cnt = (int)std::min<long long>(a.i, a.n - lo);
    const int L = m + cnt;
    const long long* const src = ((itf - a.it0) & 1) ? mem_alt : mem_out;
    long long* const dst = ((itf - a.it0) & 1) ? mem_out : mem_alt;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int u = 0; u < W; ++u) { pre[u] = run; run += team_count(L, u, W); }
        pre[W] = run;
    }
    __syncthreads();
    for (int j0 = tid; j0 < a.n2; j0 += 8 * LARGE_NT) {
        uint64_t t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * LARGE_NT;
            const int u_ = j / RUN;
            t[u] = (j < a.n2 && j - u_ * RUN < pre[u_ + 1] - pre[u_]) ? (gkeys[j] | (0x3FFFFull << 14)) : 0ull;     // (tagged: team_key_out)
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j0 + u * LARGE_NT < a.n2) keys[large_slot(j0 + u * LARGE_NT)] = t[u];
    }
    merge_sorted_runs(0, a.n2, RUN, (int)(reinterpret_cast<unsigned char*>(pre) - smem) >> 2);
    const int tie = (L > m && (keys[large_slot(m - 1)] >> 32) == (keys[large_slot(m)] >> 32)) ? 1 : 0;
    const TieRows rows = {lg, src, lo, m, 8};
    const bool replayed = large_tie_replay(L, m, a.n2, a.tie_order, lists, tail, reinterpret_cast<uint64_t*>(xT), a.rstamp != 0, &rows);
    const stdorder::E* q = reinterpret_cast<const stdorder::E*>(keys);
    const bool want_score = a.mem_score != nullptr && itf + 1 == a.it1;
    for (int j = tid; j < m; j += LARGE_NT) {
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
        dst[j] = pos < m ? src[pos] : lo + (pos - m);
    }
    __syncthreads();
    return tie;
}

#define TEAM_STAMP(k)                                                              \
    do {                                                                           \
        if (STAMP) {                                                               \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();            \
            if (tid == 0 && w == 0) tacc[k] += t_ - tlast;                         \
            tlast = t_;                                                            \
        }                                                                          \
    } while (0)

// one wait of the team: wave 0 polls, the verdict goes through LDS word `site`, every thread acquires
#define TEAM_WAIT(site, word, target)                                                                  \
    do {                                                                                               \
        if (wave == 0) {                                                                               \
            const bool ok_ = team_poll((word), (target), a.team_ticks);                               \
            if (lane == 0) wword[site] = ok_ ? 1 : 0;                                                  \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     /* (one invalidate per compute unit, before the barrier) */ \
        }                                                                                              \
        __syncthreads();                                                                               \
        if (!wword[site]) {                                                                            \
            if (!a.status) __builtin_trap();                                                           \
            if (tid == 0) __hip_atomic_fetch_or(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            return;                                                                                    \
        }                                                                                              \
    } while (0)

// CPT = candidates per thread: a workgroup's run is 1024 * CPT slots, the team W = n2 / (1024 * CPT) workgroups.
// 8 heads, one token (R = 8) only - the shape scan_large_kernel's register-resident passes are written for.
template <int CPT, bool STAMP>
__global__ __launch_bounds__(LARGE_NT) void scan_large_team_kernel(LargeArgs a, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RUN = LARGE_NT * CPT;
    constexpr int NW = LARGE_NT / 64;
    const int W = a.n2 / RUN, Lp = a.Lp, m = a.m;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    uint32_t* rmaxkey = reinterpret_cast<uint32_t*>(keys + a.n2 + (a.n2 >> 4));
    const int tail = (a.n2 + (a.n2 >> 4)) * 8 + 8 * 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / W, w = blockIdx.x - b * W;
    const float* lg = a.lg + (size_t)b * a.n * 8;
    long long* const mem_out = a.mem_idx + (size_t)b * m;
    unsigned char* wsb = a.ws + (size_t)b * a.ws_per_image;
    float* xT = reinterpret_cast<float*>(wsb);
    int* lists = reinterpret_cast<int*>(xT + (size_t)8 * Lp);
    int* ctl = reinterpret_cast<int*>(wsb + a.team_off);
    uint64_t* gkeys = reinterpret_cast<uint64_t*>(ctl + TEAM_CTL_INTS);
    long long* const mem_alt = reinterpret_cast<long long*>(gkeys + a.n2);
    // (the replay's stack, 192 ints, is free outside the replay: wait verdicts, the wavefronts' row maxima, the runs' counts)
    int* const wword = reinterpret_cast<int*>(smem + tail);
    uint32_t* const wmax = reinterpret_cast<uint32_t*>(wword + 8);          // 16 wavefronts x 8 rows
    int* const pre = wword + 8 + NW * 8;                                     // W + 1 <= 9
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;
    if (a.ready) {
        // resident only as a whole team: the producers' gate (ipsx_scan_gate) opens when the main workgroup says so
        if (tid == 0) __hip_atomic_fetch_add(&ctl[TC_RESIDENT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == 0) {
            TEAM_WAIT(0, &ctl[TC_RESIDENT], W);
            if (tid == 0) {
                __hip_atomic_fetch_or(a.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&a.plog[4], (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (STAMP && w == 0 && tid == 0) stamps[(size_t)gridDim.x / W * 8 + 127] = __builtin_amdgcn_s_memrealtime();
    long long ready_known = 0;
    if (w == 0 && a.it0 == 0)
        for (int j = tid; j < m; j += LARGE_NT) mem_out[j] = j;      // (only the main workgroup's own redo reads it in iteration 0)
    __syncthreads();
    int redone = 0;                                                  // iterations the main workgroup has redone alone
    int tie = 0;
    const int tid_outer = tid;
    for (long long it = a.it0; it < a.it1; ++it) {
        // (nothing that depends on the thread's number is carried round the loop: hoisted address arithmetic - 8 rows x CPT
        //  candidates of workspace pointers, the gather's, the sort's - was 400 spilled registers in a 128-register kernel)
        int tid = tid_outer;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
        const int e = (int)(it - a.it0) + 1;
        const long long lo = it * a.i + m;
        const int cnt = (int)std::min<long long>(a.i, a.n - lo);
        const int L = m + cnt;
        // the memory's patch numbers: an iteration reads one buffer and every workgroup writes its ranks into the other
        const long long* const mem = ((it - a.it0) & 1) ? mem_alt : mem_out;
        long long* const mem_new = ((it - a.it0) & 1) ? mem_out : mem_alt;
        // ---- what this iteration reads: its chunk's rows (persistent: published by the producer) and the memory the team
        // wrote at the end of the iteration before [hop E: all to all]
        {
            const bool need_rows = a.ready && lo + cnt > ready_known;
            const bool need_mem = it > a.it0;
            if (wave == 0) {
                int v = 1;
                if (need_rows) {
                    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    v = __hip_atomic_load(a.ready + b * a.ready_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    int seen = -1;
                    while (v >= 0 && v < lo + cnt) {
                        __builtin_amdgcn_s_sleep(16);
                        int wd = lane < a.ready_words ? __hip_atomic_load(a.ready + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                        for (int o = 32; o >= 1; o >>= 1) wd += __shfl_xor(wd, o, 64);
                        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                        if (wd != seen) { seen = wd; t0 = now; }
                        if (now - t0 > a.wait_ticks) { v = -1; break; }
                        v = __hip_atomic_load(a.ready + b * a.ready_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                // (the main workgroup may still be sorting the iteration before: its own progress is the producer's clock,
                //  so the bound here is the team's)
                if (v >= 0 && need_mem && !team_poll(&ctl[TC_E], W * (e - 1), a.team_ticks)) v = -1;
                if (lane == 0) wword[1] = v;
                if (need_rows || need_mem) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            const int v = wword[1];
            if (v < 0) {
                if (!a.status) __builtin_trap();
                if (tid == 0) __hip_atomic_fetch_or(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            if (need_rows) ready_known = v;
            // the iteration before asked for the one-workgroup ranking (torch.topk's order to replay, or the runs' top halves
            // were not enough): the main workgroup redoes it from the runs, the others wait for its memory [hop E2]
            if (need_mem && __hip_atomic_load(&ctl[TC_FLAG], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == e - 1) {
                ++redone;
                if (w == 0) {
                    tie |= team_redo<CPT>(a, it - 1, b, W, lg, mem_out, mem_alt, xT, lists, gkeys, tail);
                    team_arrive(&ctl[TC_E2]);
                } else {
                    TEAM_WAIT(6, &ctl[TC_E2], redone);
                }
            } else if (need_mem && w == 0 && __hip_atomic_load(&ctl[TC_TIE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == e - 1) {
                tie = 1;
            }
        }
        if (tid == 0) {
            int run = 0;
            for (int u = 0; u < W; ++u) { pre[u] = run; run += team_count(L, u, W); }
            pre[W] = run;
            // (diagnostic build: when this iteration's rows were there, on the 100 MHz clock - tools/scan_stamps.py largepipe)
            if (STAMP && w == 0 && it - a.it0 < 64) stamps[(size_t)gridDim.x / W * 8 + 2 * (it - a.it0)] = __builtin_amdgcn_s_memrealtime();
        }
        // ---- my candidates' 8 logits, once; row maxima of what I hold
        float4 v[CPT][2];
        uint32_t km[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) km[r] = 0u;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int l = ((wave + NW * c) * W + w) * 64 + lane;
            const size_t row = l >= L ? (size_t)0 : (l < m ? (it == 0 ? (size_t)l : (size_t)mem[l]) : (size_t)(lo + (l - m)));
            const float4* src = reinterpret_cast<const float4*>(lg + row * 8);
            v[c][0] = src[0];
            v[c][1] = src[1];
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c)
            if (((wave + NW * c) * W + w) * 64 + lane < L) {
                km[0] = max(km[0], max_key(v[c][0].x)); km[1] = max(km[1], max_key(v[c][0].y));
                km[2] = max(km[2], max_key(v[c][0].z)); km[3] = max(km[3], max_key(v[c][0].w));
                km[4] = max(km[4], max_key(v[c][1].x)); km[5] = max(km[5], max_key(v[c][1].y));
                km[6] = max(km[6], max_key(v[c][1].z)); km[7] = max(km[7], max_key(v[c][1].w));
            }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            uint32_t best = km[r];
            best = max(best, lane_xor_u32<32>(best, lane)); best = max(best, lane_xor_u32<16>(best, lane));
            best = max(best, lane_xor_u32<8>(best, lane)); best = max(best, lane_xor_u32<4>(best, lane));
            best = max(best, lane_xor_u32<2>(best, lane)); best = max(best, lane_xor_u32<1>(best, lane));
            if (lane == r) wmax[wave * 8 + r] = best;
        }
        __syncthreads();
        if (tid < 8) {
            uint32_t best = 0u;
#pragma unroll
            for (int q = 0; q < NW; ++q) best = max(best, wmax[q * 8 + tid]);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(ctl + TC_PMAX) + w * 8 + tid,
                               ((unsigned long long)(unsigned)e << 32) | best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        TEAM_STAMP(0);
        // ---- hop A: every workgroup's maxima, flag and value in one word each
        if (wave == 0) {
            uint32_t pk = 0u;
            const bool ok = team_poll_words(reinterpret_cast<const unsigned long long*>(ctl + TC_PMAX), 8 * W, e, a.team_ticks, lane, &pk);
            if (lane >= 8 * W) pk = 0u;
            pk = max(pk, lane_xor_u32<32>(pk, lane)); pk = max(pk, lane_xor_u32<16>(pk, lane)); pk = max(pk, lane_xor_u32<8>(pk, lane));
            if (lane < 8) rmaxkey[lane] = pk;
            if (lane == 0) wword[2] = ok ? 1 : 0;
        }
        __syncthreads();
        if (!wword[2]) {
            if (!a.status) __builtin_trap();
            if (tid == 0) __hip_atomic_fetch_or(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        float mx[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) mx[r] = max_key_value(rmaxkey[r]);
        TEAM_STAMP(1);
        // ---- exponentials: kept, and written transposed for the row sums
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            v[c][0].x = det_expf(v[c][0].x - mx[0]); v[c][0].y = det_expf(v[c][0].y - mx[1]);
            v[c][0].z = det_expf(v[c][0].z - mx[2]); v[c][0].w = det_expf(v[c][0].w - mx[3]);
            v[c][1].x = det_expf(v[c][1].x - mx[4]); v[c][1].y = det_expf(v[c][1].y - mx[5]);
            v[c][1].z = det_expf(v[c][1].z - mx[6]); v[c][1].w = det_expf(v[c][1].w - mx[7]);
            const int l = ((wave + NW * c) * W + w) * 64 + lane;
            if (l < L) {
                float* dst = xT + l;
                dst[0] = v[c][0].x; dst[Lp] = v[c][0].y; dst[2 * (size_t)Lp] = v[c][0].z; dst[3 * (size_t)Lp] = v[c][0].w;
                dst[4 * (size_t)Lp] = v[c][1].x; dst[5 * (size_t)Lp] = v[c][1].y; dst[6 * (size_t)Lp] = v[c][1].z; dst[7 * (size_t)Lp] = v[c][1].w;
            }
        }
        team_arrive(&ctl[TC_B]);
        // ---- hop B (all to all): every workgroup's exponentials are in the workspace
        TEAM_WAIT(3, &ctl[TC_B], W * e);
        TEAM_STAMP(2);
        // ---- denominators in the wavefront order of the contract: lane j adds elements j, j + 64, ... ascending - 64 chains
        // per row that cannot be cut.  Their LOADS can: workgroup w takes the rows [w * 8 / W, (w + 1) * 8 / W), 2 W
        // wavefronts per row; every wavefront loads its stretch of the chain at once (one round trip for the whole row, the
        // other workgroups' exponentials come from their XCDs' side of the memory), then the sum goes from wavefront to
        // wavefront through LDS in the chain's order.  (One workgroup, one wavefront per row, 16 elements in flight: ten
        // round trips one after the other - 28 k cycles.)
        {
            constexpr int KB = 8 * CPT;                          // blocks of 64 a wavefront holds: K <= 16 * CPT * W blocks over 2 W wavefronts
            const int wpr = 2 * W;                               // wavefronts per row
            const int rloc = wave / wpr, q = wave - rloc * wpr, r = w * (8 / W) + rloc;
            const int K = (L + 63) >> 6, kb = (K + wpr - 1) / wpr;
            const float* x = xT + (size_t)r * Lp + lane;
            float* part = reinterpret_cast<float*>(keys);        // (the ranking's LDS is free here)
            float t[KB];
#pragma unroll
            for (int u = 0; u < KB; ++u) {
                const int k = q * kb + u;
                t[u] = x[(size_t)64 * ((u < kb && k * 64 + lane < L) ? k : 0)];
            }
            for (int step = 0; step < wpr; ++step) {
                if (q == step) {
                    float sum = step ? part[rloc * 64 + lane] : 0.0f;
#pragma unroll
                    for (int u = 0; u < KB; ++u)
                        if (u < kb && (q * kb + u) * 64 + lane < L) sum = sum + t[u];
                    if (step + 1 < wpr) {
                        part[rloc * 64 + lane] = sum;
                    } else {
                        sum = wave_butterfly_sum(sum);
                        if (lane == 0)                               // (the reciprocal: weights are e * (1 / den))
                            __hip_atomic_store(reinterpret_cast<unsigned long long*>(ctl + TC_RDEN) + r,
                                               ((unsigned long long)(unsigned)e << 32) | as_u32(1.0f / sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                __syncthreads();
            }
        }
        // ---- hop C (all to all): the 8 reciprocal denominators, flag and value in one word each
        float rd[8];
        if (wave == 0) {
            uint32_t bits = 0u;
            const bool ok = team_poll_words(reinterpret_cast<const unsigned long long*>(ctl + TC_RDEN), 8, e, a.team_ticks, lane, &bits);
            if (lane < 8) rmaxkey[lane] = bits;
            if (lane == 0) wword[5] = ok ? 1 : 0;
        }
        __syncthreads();
        if (!wword[5]) {
            if (!a.status) __builtin_trap();
            if (tid == 0) __hip_atomic_fetch_or(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) rd[r] = __uint_as_float(rmaxkey[r]);
        TEAM_STAMP(3);
        // ---- scores (mean over the heads; one token) and ranking keys of my candidates, my run sorted
        uint64_t key[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int l = ((wave + NW * c) * W + w) * 64 + lane;
            key[c] = 0ull;
            if (l < L) {
                float st = 0.0f;
                float sh = 0.0f;
                sh = sh + v[c][0].x * rd[0]; sh = sh + v[c][0].y * rd[1]; sh = sh + v[c][0].z * rd[2]; sh = sh + v[c][0].w * rd[3];
                sh = sh + v[c][1].x * rd[4]; sh = sh + v[c][1].y * rd[5]; sh = sh + v[c][1].z * rd[6]; sh = sh + v[c][1].w * rd[7];
                st = st + sh / (float)a.h;
                key[c] = rank_key(st / (float)a.T, (uint32_t)l);
            }
        }
        team_sort_run<CPT>(key, keys, tid, lane);
        TEAM_STAMP(4);
        // ---- hop D (all to all): every workgroup's sorted run through the workspace - the real keys only, each with the
        // iteration's number in it (team_key_out): no fence, no counter; the readers below poll the words they need
        {
            const int mine_all = pre[w + 1] - pre[w];
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const int j = tid + c * LARGE_NT;
                if (j < mine_all)
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(gkeys) + (size_t)w * RUN + j,
                                       (unsigned long long)team_key_out(keys[large_slot(j)], (uint32_t)e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ---- the ranking, by every workgroup for its OWN run: a key's rank among all L candidates is its place in its run
        // plus, for every other run, the number of keys there that are larger (keys are unique) - a branch-free lower bound
        // per run, in LDS.  Only the first m + 1 ranks are ever used, and with candidates dealt by 64-blocks every run holds
        // about its share of them: the runs' TOP HALVES are loaded and searched when (m + 1) x 1.25 fits them; the ranks are
        // then the true ones down to rank m if the score at rank m is above every score left out (the largest is its run's
        // key KT) - the thread that holds rank m checks.  Otherwise, or when two of the first m + 1 ranks tie in a way that
        // calls for torch.topk's order (a replay wants all L candidates in one place), the iteration is flagged and the
        // main workgroup redoes its ranking alone from the runs (team_redo: the ranking of scan_large_kernel).
        int trunc_total = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) trunc_total += u < W ? min(RUN / 2, pre[u + 1] - pre[u]) : 0;
        const bool trunc = a.team_trunc && L > m + 1 && (long long)(m + 1) * 5 <= (long long)(a.n2 >> 1) * 4 && trunc_total > m;
        const int KT = trunc ? RUN / 2 : RUN;
        const int ksh = 31 - __clz(KT);
        uint64_t* const T = keys + (RUN + (RUN >> 4));             // the other runs, unpadded, behind this workgroup's own
        unsigned long long* const exclw = reinterpret_cast<unsigned long long*>(wword + 164);
        bool late = false;                                         // a word that never came (bounded like every wait)
        if (wave == 0) {                                           // the largest key left out
            unsigned long long ex = 0ull;
            if (trunc && lane < W && pre[lane + 1] - pre[lane] > KT) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                uint64_t k_ = 0ull;
                while (!team_key_in(__hip_atomic_load(reinterpret_cast<const unsigned long long*>(gkeys) + (size_t)lane * RUN + KT,
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), (uint32_t)e, &k_)) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > a.team_ticks) { late = true; break; }
                }
                ex = k_;
            }
            for (int o = 32; o >= 1; o >>= 1) {
                const unsigned long long other = __shfl_xor(ex, o, 64);
                ex = other > ex ? other : ex;
            }
            if (lane == 0) *exclw = ex;
        }
        for (int i0 = tid; i0 < (W - 1) * KT; i0 += 8 * LARGE_NT) {
            uint64_t t[8];
            const unsigned long long* src[8];
            uint32_t need = 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * LARGE_NT;
                const int tv = i >> ksh, v_ = tv < w ? tv : tv + 1, il = i - (tv << ksh);
                const bool real = i < (W - 1) * KT && il < pre[v_ + 1] - pre[v_];
                src[u] = reinterpret_cast<const unsigned long long*>(gkeys) + (real ? (size_t)v_ * RUN + il : (size_t)0);
                t[u] = 0ull;
                need |= real ? 1u << u : 0u;
            }
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (need != 0u) {                                      // (all eight in flight; the ones that have come drop out)
                unsigned long long wd[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wd[u] = (need >> u) & 1u ? __hip_atomic_load(src[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if ((need >> u) & 1u) {
                        uint64_t k_;
                        if (team_key_in(wd[u], (uint32_t)e, &k_)) { t[u] = k_; need &= ~(1u << u); }
                    }
                if (need != 0u) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > a.team_ticks) { late = true; break; }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * LARGE_NT < (W - 1) * KT) T[i0 + u * LARGE_NT] = t[u];
        }
        if (__syncthreads_or(late ? 1 : 0)) {
            if (!a.status) __builtin_trap();
            if (tid == 0) __hip_atomic_fetch_or(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        TEAM_STAMP(5);
        {
            const int mine = min(KT, pre[w + 1] - pre[w]), mine_all = pre[w + 1] - pre[w];
            const int nmax = m < L - 1 ? m : L - 1;                 // ranks whose ties matter (ranked_ties_padded)
            const bool want_score = a.mem_score != nullptr && it + 1 == a.it1;
            bool flag = false;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const int j = tid + c * LARGE_NT;
                if (j < mine) {                                      // (wave-uniform up to the run's last block)
                    const uint64_t ka = keys[large_slot(j)];
                    int pos[7];
#pragma unroll
                    for (int tv = 0; tv < 7; ++tv) pos[tv] = 0;
                    for (int half = KT >> 1; half >= 1; half >>= 1) {
#pragma unroll
                        for (int tv = 0; tv < 7; ++tv)
                            if (tv < W - 1) pos[tv] += T[(tv << ksh) + pos[tv] + half - 1] > ka ? half : 0;
                    }
                    int g = j;
                    // the next smaller key of every run (0: none): an equal score there is a tie of this rank
                    uint64_t nxt = j + 1 < mine_all ? keys[large_slot(j + 1)] : 0ull;
                    bool eq = nxt != 0ull && (nxt >> 32) == (ka >> 32);
#pragma unroll
                    for (int tv = 0; tv < 7; ++tv)
                        if (tv < W - 1) {
                            const uint64_t last = T[(tv << ksh) + pos[tv]];
                            pos[tv] += last > ka ? 1 : 0;
                            const uint64_t s_ = last > ka ? (pos[tv] < KT ? T[(tv << ksh) + pos[tv]] : 0ull) : last;
                            eq = eq || (s_ != 0ull && (s_ >> 32) == (ka >> 32));
                            g += pos[tv];
                        }
                    if (trunc && g == m && !((ka >> 32) > (*exclw >> 32))) flag = true;
                    if (g < m) {
                        const int p_ = (int)key_pos(ka);
                        mem_new[g] = p_ < m ? (it == 0 ? (long long)p_ : mem[p_]) : lo + (p_ - m);
                        if (want_score) a.mem_score[(size_t)b * m + g] = key_score(ka);
                    }
                    // (the score at rank m equals this one; counted by the main workgroup unless the iteration is redone)
                    if (eq && g == m - 1 && L > m) __hip_atomic_fetch_max(&ctl[TC_TIE], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (eq && g < nmax && a.tie_order != 0) {
                        // equal scores among the first m + 1 ranks (rare: a few pairs among 10,000 candidates).  The loops' rule
                        // (oracle orc_topm_loop): torch.topk's order where two members of the run of equal scores have
                        // bit-identical logit rows; tie_order 2: wherever scores tie
                        bool rep = a.tie_order != 1;
                        if (!rep) {
                            const long long pa = key_pos(ka);
                            const float* ra = lg + (size_t)(pa < m ? (it == 0 ? pa : mem[pa]) : lo + (pa - m)) * 8;
                            const float4 a0 = *reinterpret_cast<const float4*>(ra), a1 = *reinterpret_cast<const float4*>(ra + 4);
                            for (int tv = -1; tv < W - 1 && !rep; ++tv) {
                                for (int q = tv < 0 ? j + 1 : pos[tv]; q < (tv < 0 ? mine_all : KT) && !rep; ++q) {
                                    const uint64_t kb = tv < 0 ? keys[large_slot(q)] : T[(tv << ksh) + q];
                                    if (kb == 0ull || (kb >> 32) != (ka >> 32)) break;
                                    const long long pb = key_pos(kb);
                                    const float* rb = lg + (size_t)(pb < m ? (it == 0 ? pb : mem[pb]) : lo + (pb - m)) * 8;
                                    const float4 b0 = *reinterpret_cast<const float4*>(rb), b1 = *reinterpret_cast<const float4*>(rb + 4);
                                    const uint32_t diff = (as_u32(a0.x) ^ as_u32(b0.x)) | (as_u32(a0.y) ^ as_u32(b0.y)) | (as_u32(a0.z) ^ as_u32(b0.z)) |
                                                          (as_u32(a0.w) ^ as_u32(b0.w)) | (as_u32(a1.x) ^ as_u32(b1.x)) | (as_u32(a1.y) ^ as_u32(b1.y)) |
                                                          (as_u32(a1.z) ^ as_u32(b1.z)) | (as_u32(a1.w) ^ as_u32(b1.w));
                                    rep = diff == 0u;
                                }
                            }
                        }
                        flag = flag || rep;
                    }
                }
            }
            if (flag) __hip_atomic_fetch_max(&ctl[TC_FLAG], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        TEAM_STAMP(6);
        team_arrive(&ctl[TC_E]);
        TEAM_STAMP(7);
        if (STAMP && w == 0 && tid == 0 && it - a.it0 < 64) stamps[(size_t)gridDim.x / W * 8 + 2 * (it - a.it0) + 1] = __builtin_amdgcn_s_memrealtime();
    }
    // ---- the end: the main workgroup sees the last iteration through (everybody's ranks written; redone if it was flagged)
    // and leaves the memory in the caller's buffer
    if (w != 0 || a.it1 <= a.it0) return;
    {
        const int e = (int)(a.it1 - a.it0);
        TEAM_WAIT(7, &ctl[TC_E], W * e);
        if (__hip_atomic_load(&ctl[TC_FLAG], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == e)
            tie |= team_redo<CPT>(a, a.it1 - 1, b, W, lg, mem_out, mem_alt, xT, lists, gkeys, tail);
        else if (__hip_atomic_load(&ctl[TC_TIE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == e)
            tie = 1;
        if (a.tie && tid == 0 && tie) a.tie[b] = 1;
        if (e & 1) {                                               // an odd number of iterations: the last one wrote the second buffer
            __syncthreads();
            for (int j = tid; j < m; j += LARGE_NT) mem_out[j] = mem_alt[j];
        }
    }
    if (STAMP && tid == 0)
        for (int k = 0; k < 8; ++k) stamps[(size_t)b * 8 + k] += tacc[k];
}
#undef TEAM_STAMP
#undef TEAM_WAIT
