// scan_common.h - what the selection-loop translation units share (round 6: scorer.hip, 3,390 lines, became logits.hip,
// scan_fast.hip, scan_cam.hip, scan_large.hip, topm.hip and scorer.hip = the C ABI's dispatcher): scoring of a candidate
// set, the replay of torch.topk's tie order (ipsx_stdorder.h on a wavefront / a workgroup), the argument block and the
// ranking helpers of the LDS-resident loops, the waiting of a persistent loop, and the host-side declarations that tie the
// units together.  Reference: IPSNet.score_and_select (architecture/ips_net.py:136-155), Transformer.get_scores
// (architecture/transformer.py:143-148), the chunk loop of IPSNet.ips (:213-241).
#pragma once

#include <algorithm>
#include <cstdlib>

#include "ipsx_common.h"
#include "ipsx_math.h"
#include "ipsx_rowstats.h"
#include "ipsx_stdorder.h"

namespace ipsx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------ scoring of a candidate set
// Candidate logits either staged in LDS (cl, row stride R+1) or read through `cand`
// from the global (n, R) table of one image.
struct CandView {
    const float* cl;       // LDS staging or nullptr
    const float* lg;       // global logits of this image, (n, R)
    const int* cand;       // LDS: candidate -> patch index (nullptr = identity)
    int R;
    __device__ __forceinline__ float get(int l, int r) const {
        if (cl) return cl[l * (R + 1) + r];
        const size_t row = cand ? (size_t)cand[l] : (size_t)l;
        return lg[row * R + r];
    }
};

// per-(h,t) row maximum and softmax denominator over L candidates (wave per row)
__device__ __forceinline__ void row_stats(const CandView& v, int L, float* rmax, float* rden) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int r = wave; r < v.R; r += nw) {
        float m = -__builtin_huge_valf();
        for (int i = lane; i < L; i += 64) m = nanmax(m, v.get(i, r));
        m = wave_max(m);
        float s = 0.0f;
        for (int i = lane; i < L; i += 64) s = s + det_expf(v.get(i, r) - m);
        s = wave_butterfly_sum(s);
        if (lane == 0) { rmax[r] = m; rden[r] = 1.0f / s; }        // (the RECIPROCAL: weights are e * (1 / den), oracle orc_scores_from_logits)
    }
}

// score of candidate l: mean over heads, then over tokens, of its attention weights
__device__ __forceinline__ float cand_score(const CandView& v, int l, int h, int T, const float* rmax,
                                            const float* rden, float* attn, int L) {
    float st = 0.0f;
    for (int t = 0; t < T; ++t) {
        float sh = 0.0f;
        for (int hh = 0; hh < h; ++hh) {
            const int r = hh * T + t;
            const float a = det_expf(v.get(l, r) - rmax[r]) * rden[r];
            if (attn) attn[((size_t)hh * T + t) * L + l] = a;
            sh = sh + a;
        }
        st = st + sh / (float)h;
    }
    return st / (float)T;
}

// Sort `n2` (power of two) keys descending.  src holds the keys; the sorted keys end
// up in the returned buffer (src or tmp).  Keys are unique.  Must be called by all
// threads of the workgroup; contains barriers.
__device__ __forceinline__ uint64_t* sort_desc(uint64_t* src, uint64_t* tmp, int L, int n2) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (L <= 512) {
        // rank by counting: rank = number of larger keys (LDS broadcast reads, no barriers inside)
        __syncthreads();
        for (int l = tid; l < L; l += nt) {
            const uint64_t k = src[l];
            int rank = 0;
            for (int j = 0; j < L; ++j) rank += (src[j] > k) ? 1 : 0;
            tmp[rank] = k;
        }
        __syncthreads();
        return tmp;
    }
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = tid; i < n2; i += nt) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t x = src[i], y = src[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) { src[i] = y; src[ixj] = x; }
                }
            }
        }
    }
    __syncthreads();
    return src;
}

// ---- exact score ties (ipsx_stdorder.h).  `sorted` = keys in canonical order (score desc, earlier
// position first).  When two of the first m+1 ranked scores are equal the reference returns whatever libstdc++'s
// nth_element / sort / partial_sort leave behind, and that order feeds the next iteration; with tie order 1
// ("torch", the default) one lane replays those routines on the candidate array and rewrites sorted[0..m).
extern int g_tie_order;         // (scorer.hip) 0 canonical; 1 (default) torch.topk's order where bit-identical candidates tie; 2 wherever scores tie
constexpr int STK_BYTES = 3 * stdorder::STACK_RANGES * 4;

// every wave evaluates this on the same data: the result is uniform over the workgroup without a barrier
__device__ __forceinline__ bool ranked_ties(const uint64_t* sorted, int L, int m, int lane) {
    const int n = m < L - 1 ? m : L - 1;
    bool any = false;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        const bool e = j < n && (sorted[j] >> 32) == (sorted[j + 1] >> 32);
        any = any || (__ballot(e) != 0ull);
    }
    return any;
}

// the same test on the padded key array of the large kernels (key j at slot j + (j >> 4))
// where the logit rows of a loop's candidates live (scan_large_kernel): candidate p < m is memory slot p = patch mem[p],
// candidate p >= m is patch lo + (p - m); R floats per patch.  lg == nullptr: no rows (ipsx_topm: scores only)
struct TieRows { const float* lg; const long long* mem; long long lo; int m, R; };

// ties among the first m + 1 canonical ranks that call for torch.topk's order: tie_order 2 (and callers without rows) any
// two neighbours of equal score; tie_order 1, the loop's rule (oracle orc_topm_loop): two members of one run of equal
// scores (reaching into the first m + 1 ranks) whose logit rows are bit-identical
// (the whole workgroup calls this: the pairs are dealt out to the wavefronts, the verdict is a barrier's OR - every
//  wavefront going through all 5,000 pairs of the shipped CAMELYON sizes on its own was 13 us of a 94 us iteration)
__device__ __forceinline__ bool ranked_ties_padded(const uint64_t* sorted, int L, int m, int lane, int tie_order = 2,
                                                   const TieRows* rows = nullptr) {
    const int n = m < L - 1 ? m : L - 1;
    const bool by_rows = tie_order == 1 && rows != nullptr && rows->lg != nullptr;
    bool any = false;
    // (round 6: member j against EVERY later member of its run of equal scores, not only its neighbour - oracle orc_topm_loop)
    for (int j = (int)threadIdx.x; j < n && !any; j += (int)blockDim.x) {
        const uint64_t ka = sorted[j + (j >> 4)];
        for (int k = j + 1; k < L && !any; ++k) {
            const uint64_t kb = sorted[k + (k >> 4)];
            if ((ka >> 32) != (kb >> 32)) break;
            bool e = true;
            if (by_rows) {
                const long long pa = key_pos(ka), pb = key_pos(kb);
                const float* ra = rows->lg + (size_t)(pa < rows->m ? rows->mem[pa] : rows->lo + (pa - rows->m)) * rows->R;
                const float* rb = rows->lg + (size_t)(pb < rows->m ? rows->mem[pb] : rows->lo + (pb - rows->m)) * rows->R;
                // (every load issued before the first compare: `e && ...` made them R round trips one after the other -
                //  with a bit-equal pair of scores in nearly every iteration of 10,000 candidates that was 4 us of each)
                if ((rows->R & 3) == 0) {
                    uint32_t diff = 0u;
                    for (int r = 0; r < rows->R; r += 4) {
                        const float4 va = *reinterpret_cast<const float4*>(ra + r), vb = *reinterpret_cast<const float4*>(rb + r);
                        diff |= (as_u32(va.x) ^ as_u32(vb.x)) | (as_u32(va.y) ^ as_u32(vb.y)) | (as_u32(va.z) ^ as_u32(vb.z)) | (as_u32(va.w) ^ as_u32(vb.w));
                    }
                    e = diff == 0u;
                } else {
                    uint32_t diff = 0u;
                    for (int r = 0; r < rows->R; ++r) diff |= as_u32(ra[r]) ^ as_u32(rb[r]);
                    e = diff == 0u;
                }
            }
            any = any || e;
            if (!by_rows) break;                       // (any tie counts: the neighbour settles it)
        }
    }
    (void)lane;
    return __syncthreads_or(any ? 1 : 0) != 0;
}

// ---- the replay on ONE WAVEFRONT instead of one lane.  libstdc++'s routines are sequential, but what they compute is
// not: (1) the unguarded Hoare partition pairs the t-th element from the left that stops the upward scan (not greater than
// the pivot) with the t-th from the right that stops the downward scan (not smaller), for as long as the left one lies
// before the right one, swaps each pair, and returns where the upward scan stops next - both scans only ever see
// elements no swap has touched, so the pairs can be read off the ORIGINAL array with two ballots per 64 elements;
// (2) the final insertion pass of std::sort never moves an element across a partition cut and is a stable sort, so every
// element of a leaf of the introsort loop (<= 16 elements) finds its place by counting, a lane per element.  Everything else (median of three, the
// <= 3-element tail of nth_element, the heap fallbacks at depth 0) stays the sequential restatement on lane 0.
// oracle/check_stdorder.cpp holds this formulation against std:: itself; tests/test_hip_kernels.py holds the device code
// against the one-lane replay.  One lane took ~600 k cycles for 512 candidates (a quarter of a millisecond - 10 % of a
// CAMELYON slide for ONE tie among 255 iterations); the wavefront takes ~130 k.
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// GL = the index lists la / lb live in GLOBAL memory (candidate sets beyond the LDS, scan_large_kernel): what one lane
// wrote there must be visible to the lane that reads it next - a workgroup-scope fence (the lists never leave the
// compute unit's L1 / its write-through path) on top of the LDS wait.  q stays in LDS either way.
template <bool GL>
__device__ __forceinline__ void wave_fence() {
    if (GL) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
        wave_lds_fence();
    }
}

template <bool GL>
__device__ __forceinline__ int wave_partition(stdorder::E* q, int first, int last, const stdorder::E P, int* la, int* lb, int lane) {
    const unsigned long long below = (1ull << lane) - 1ull;
    int na = 0, nb = 0;
    for (int base = first; base < last; base += 64) {                  // ascending: indices that stop the upward scan
        const int x = base + lane;
        const bool in = x < last;
        const stdorder::E e = q[in ? x : first];
        const bool stop = in && !stdorder::gt(e, P);
        const unsigned long long mask = __ballot(stop);
        if (stop) la[na + __popcll(mask & below)] = x;
        na += __popcll(mask);
    }
    for (int top = last - 1; top >= first; top -= 64) {                // descending: indices that stop the downward scan
        const int x = top - lane;
        const bool in = x >= first;
        const stdorder::E e = q[in ? x : first];
        const bool stop = in && !stdorder::gt(P, e);
        const unsigned long long mask = __ballot(stop);
        if (stop) lb[nb + __popcll(mask & below)] = x;
        nb += __popcll(mask);
    }
    wave_fence<GL>();
    const int np = na < nb ? na : nb;
    int t = 0;                                                         // pairs that are swapped: la[u] < lb[u], a prefix
    for (int base = 0; base < np; base += 64) {                        // (the pairs are disjoint: swapped as they are found)
        const int u = base + lane;
        const int i = la[u < np ? u : 0], j = lb[u < np ? u : 0];
        const bool ok = u < np && i < j;
        const unsigned long long mask = __ballot(ok);
        if (ok) {
            const stdorder::E ei = q[i], ej = q[j];
            q[i] = ej;
            q[j] = ei;
        }
        const int valid = np - base < 64 ? np - base : 64;
        const unsigned long long full = valid == 64 ? ~0ull : ((1ull << valid) - 1ull);
        if ((mask & full) == full) { t += valid; continue; }
        t += __ffsll((long long)(~mask)) - 1;
        break;
    }
    const int cut = (t < na && (t == 0 || la[t] < lb[t - 1])) ? la[t] : lb[t - 1];
    wave_fence<GL>();
    return cut;
}

template <bool GL>
__device__ __forceinline__ int wave_partition_pivot(stdorder::E* q, int first, int last, int* la, int* lb, int lane) {
    // std::__move_median_to_first(first, first + 1, mid, last - 1): the four elements are read at once (one round trip,
    // every lane the same addresses), the decision is the restatement's, lane 0 does the swap
    const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
    const stdorder::E er = q[first], ea = q[ia], eb = q[ib], ec = q[ic];
    int sel;
    if (stdorder::gt(ea, eb)) sel = stdorder::gt(eb, ec) ? ib : (stdorder::gt(ea, ec) ? ic : ia);
    else sel = stdorder::gt(ea, ec) ? ia : (stdorder::gt(eb, ec) ? ic : ib);
    const stdorder::E P = sel == ia ? ea : (sel == ib ? eb : ec);
    if (lane == 0) { q[first] = P; q[sel] = er; }
    wave_fence<GL>();
    return wave_partition<GL>(q, first + 1, last, P, la, lb, lane);
}

// q[0..n) = (score, position) in candidate order on entry; q[0..k) = torch.topk's answer on return.  Called by the 64
// lanes of ONE wavefront.  la / lb: n ints each (lb = la + n: the two together are the n-element scratch of the last
// pass); stk: 2 * STACK_RANGES ints of pending ranges + the leaf bitmap `leaf` of `leaf_words` 64-bit words (n <= 64 *
// leaf_words, else one lane runs the sequential restatement).
template <bool GL>
__device__ __forceinline__ void torch_topk_wave(stdorder::E* q, int n, int k, int* la, int* lb, int* stk,
                                                unsigned long long* leaf, int leaf_words, int lane) {
    using namespace stdorder;
    if (k <= 0 || n <= 0) return;
    if ((long long)k * 64 <= (long long)n || n > 64 * leaf_words) {    // heap select / sort, or beyond the leaf bitmap
        if (lane == 0) torch_topk(q, n, k, stk);
        wave_fence<GL>();
        return;
    }
    {   // std::nth_element(q, q + k - 1, q + n)
        int first = 0, last = n;
        const int nth = k - 1;
        bool done = nth == last;
        int depth = lg2(last - first) * 2;
        while (!done && last - first > 3) {
            if (depth == 0) {
                if (lane == 0) { heap_select(q, first, nth + 1, last); swp(q, first, nth); }
                done = true;
                break;
            }
            --depth;
            const int cut = wave_partition_pivot<GL>(q, first, last, la, lb, lane);
            if (cut <= nth) first = cut;
            else last = cut;
        }
        if (!done && lane == 0) insertion_sort(q, first, last);
        wave_fence<GL>();
    }
    const int last = k - 1;                                            // std::sort(q, q + k - 1)
    if (last <= 0) return;
    for (int w = lane; w < leaf_words; w += 64) leaf[w] = 0ull;
    wave_fence<GL>();
    int sp = 1;
    if (lane == 0) { stk[0] = 0; stk[1] = last; stk[2] = lg2(last) * 2; }
    wave_fence<GL>();
    while (sp > 0) {
        --sp;
        int rf = stk[3 * sp], rl = stk[3 * sp + 1], depth = stk[3 * sp + 2];
        while (rl - rf > 16) {
            if (depth == 0) {
                if (lane == 0) { make_heap(q, rf, rl); sort_heap(q, rf, rl); }
                wave_fence<GL>();
                break;
            }
            --depth;
            const int cut = wave_partition_pivot<GL>(q, rf, rl, la, lb, lane);
            if (lane == 0) { stk[3 * sp] = cut; stk[3 * sp + 1] = rl; stk[3 * sp + 2] = depth; }
            wave_fence<GL>();
            ++sp;
            rl = cut;
        }
        if (lane == 0) {                                               // [rf, rl) is a leaf (a heap-sorted range is one too)
            leaf[rf >> 6] |= 1ull << (rf & 63);
            if (rl < last) leaf[rl >> 6] |= 1ull << (rl & 63);
        }
        wave_fence<GL>();
    }
    // The final insertion pass, leaf by leaf.  Linear insertion is a STABLE sort (an element moves left past strictly smaller
    // ones only), so an element's place in its leaf is the number of leaf elements that are greater plus the number of
    // equivalent ones in front of it: one lane per element counts over its leaf (<= 16 independent reads) instead of one
    // lane per leaf shifting elements one dependent LDS round trip at a time.  Results go to the list scratch and back.
    // (A range that ended in the heap sort is a "leaf" of more than 16 elements and already in order: it stays.)
    E* tmp = reinterpret_cast<E*>(la);                                  // la / lb: 2 n ints = n elements
    for (int base = 0; base < last; base += 64) {
        const int x = base + lane;
        if (x < last) {
            const E own = q[x];
            int w = x >> 6;
            unsigned long long m = leaf[w] & (~0ull >> (63 - (x & 63)));
            while (m == 0ull && w > 0) m = leaf[--w];
            const int sfirst = m ? w * 64 + 63 - __clzll((long long)m) : 0;
            int e = last;
            w = x >> 6;
            m = (x & 63) == 63 ? 0ull : (leaf[w] >> ((x & 63) + 1)) << ((x & 63) + 1);
            while (m == 0ull && w < leaf_words - 1) m = leaf[++w];
            if (m) e = w * 64 + __ffsll((long long)m) - 1;
            if (e > last) e = last;
            int dst = x;
            if (e - sfirst <= 16) {
                int rank = 0;
                for (int j = sfirst; j < e; ++j) {
                    const E o = q[j];
                    const bool greater = gt(o, own);
                    const bool equiv = !greater && !gt(own, o);
                    rank += (greater || (equiv && j < x)) ? 1 : 0;
                }
                dst = sfirst + rank;
            }
            tmp[dst] = own;
        }
    }
    wave_fence<GL>();
    for (int base = 0; base < last; base += 64)
        if (base + lane < last) q[base + lane] = tmp[base + lane];
    wave_fence<GL>();
}

// ---- the replay on the whole WORKGROUP (candidate sets beyond the LDS: scan_large_kernel ranks 10,000 candidates whose
// scores lie within a binade or two, so SOME pair of the first m + 1 is bit-equal in practically every iteration and the
// replay is part of every iteration there).  std::sort's introsort loop only ever splits a range into two disjoint
// ranges that never interact again, so the order in which pending ranges are processed is immaterial: instead of one
// wavefront working through a stack, the ranges of a level are dealt to the 16 wavefronts (two range lists in LDS,
// breadth first, a barrier per level), each range partitioned by the unchanged wave_partition_pivot with its own stretch
// of the index lists (la + first, lb + first: ranges are disjoint, so are the stretches).  std::nth_element in front of it is
// ONE chain of partitions and stays on one wavefront; the final insertion pass runs on all threads.
// The unguarded partition of q[first + 1, last) around the median of three, by ALL NT threads (16 wavefronts): what
// wave_partition does with one - the t-th element from the left that stops the upward scan paired with the t-th from the
// right that stops the downward scan, swapped while the left one lies before the right one - with the stretch cut into
// one slice per wavefront: ballots per 64 elements (kept in registers), the slices' counts through LDS, an exclusive sum,
// the two index lists written in one go, the pair swaps an element per thread.  Lists in GLOBAL memory (they are as long
// as the range; every access is a coalesced bulk access behind a barrier).  For the long stretches of std::nth_element's
// chain (10,000, 5,000, 2,500 candidates: 80 k cycles on one wavefront with its lists in global memory, ~10 k here).
// `sc`: 2 * (NT / 64) + 2 ints of LDS scratch.  Returns the cut (workgroup-uniform).  Contains barriers.
constexpr int BLOCK_PART_CHUNKS = 16;       // 64-element chunks per wavefront slice: stretches up to 16 * 16 * 64 = 16,384

template <int NT>
__device__ __forceinline__ int block_partition_pivot(stdorder::E* q, int first, int last, int* la, int* lb, int* sc) {
    using namespace stdorder;
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {   // std::__move_median_to_first(first, first + 1, mid, last - 1)
        const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
        const E er = q[first], ea = q[ia], eb = q[ib], ec = q[ic];
        int sel;
        if (gt(ea, eb)) sel = gt(eb, ec) ? ib : (gt(ea, ec) ? ic : ia);
        else sel = gt(ea, ec) ? ia : (gt(eb, ec) ? ic : ib);
        const E P = sel == ia ? ea : (sel == ib ? eb : ec);
        q[first] = P; q[sel] = er;
        sc[2 * NW] = 0x7fffffff;                                       // t: first pair that is not swapped
    }
    __syncthreads();
    const E P = q[first];
    const int f = first + 1, len = last - f;
    const int per = ((len + NW - 1) / NW + 63) & ~63;                  // slice length, whole chunks
    const int s0 = f + wave * per, s1 = min(last, s0 + per);           // this wavefront's slice
    int ca = 0, cb = 0;                                                // (two passes over the slice: the ballots are taken again
    for (int c = 0; c < BLOCK_PART_CHUNKS; ++c) {                      //  in the second instead of living in 64 registers)
        const int x = s0 + c * 64 + lane;
        if (s0 + c * 64 >= s1) break;
        const bool in = x < s1;
        const E e = q[in ? x : first];
        ca += __popcll(__ballot(in && !gt(e, P)));                     // stops the upward scan
        cb += __popcll(__ballot(in && !gt(P, e)));                     // stops the downward scan
    }
    if (lane == 0) { sc[wave] = ca; sc[NW + wave] = cb; }
    __syncthreads();
    int base_a = 0, base_b = 0, na = 0, nb = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int a_ = sc[w], b_ = sc[NW + w];
        base_a += w < wave ? a_ : 0;                                   // la ascends: lower slices first
        base_b += w > wave ? b_ : 0;                                   // lb descends: higher slices first
        na += a_; nb += b_;
    }
    {
        const unsigned long long below = (1ull << lane) - 1ull, above = lane == 63 ? 0ull : (~0ull << (lane + 1));
        int pa = base_a, pb = base_b + cb;                             // pb: end of this slice's stretch of lb
        for (int c = 0; c < BLOCK_PART_CHUNKS; ++c) {
            const int x = s0 + c * 64 + lane;
            if (s0 + c * 64 >= s1) break;
            const bool in = x < s1;
            const E e = q[in ? x : first];
            const unsigned long long ma = __ballot(in && !gt(e, P)), mb = __ballot(in && !gt(P, e));
            if ((ma >> lane) & 1ull) la[pa + __popcll(ma & below)] = x;
            pa += __popcll(ma);
            pb -= __popcll(mb);                                        // the chunk's stops, highest index first, start here
            if ((mb >> lane) & 1ull) lb[pb + __popcll(mb & above)] = x;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    const int np = na < nb ? na : nb;
    int tmin = 0x7fffffff;
    for (int u = tid; u < np; u += NT) {
        const int i = la[u], j = lb[u];
        if (i < j) {
            const E ei = q[i], ej = q[j];
            q[i] = ej;
            q[j] = ei;
        } else {
            tmin = u < tmin ? u : tmin;
        }
    }
    for (int off = 32; off >= 1; off >>= 1) tmin = min(tmin, __shfl_xor(tmin, off, 64));
    if (lane == 0 && tmin != 0x7fffffff) atomicMin(&sc[2 * NW], tmin);
    __syncthreads();
    int t = sc[2 * NW];
    t = t < np ? t : np;
    const int cut = (t < na && (t == 0 || la[t] < lb[t - 1])) ? la[t] : lb[t - 1];
    __syncthreads();                                                   // sc and the lists are reused by the next call
    return cut;
}

__device__ __forceinline__ uint64_t wave_sort_desc(uint64_t key, int lane);        // (defined with the ranking helpers below)

constexpr int BLOCK_QCAP = 1024;            // ranges of more than 16 elements pending at one level: <= 16,384 / 17
constexpr int BLOCK_RANGE = 2048;           // ranges of std::sort longer than this are partitioned by the whole workgroup

// Diagnostic (ipsx_dbg_replay_stamps): shader cycles of the replay's phases, summed by thread 0 of every workgroup -
// [0] nth_element by the workgroup, [1] its chain on one wavefront, [2 .. 5] the first four levels of std::sort's
// partitions, [6] the deeper levels, [7] the final insertion pass; [8] = replays counted.
static __device__ unsigned long long g_replay_t[10];      // (one copy per translation unit: only scan_large.hip stamps)
#define RSTAMP(k)                                                          \
    do {                                                                   \
        if (rst && tid == 0) {                                             \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();    \
            g_replay_t[k] += t_ - rlast;                                   \
            rlast = t_;                                                    \
        }                                                                  \
    } while (0)

// la_n / lb_n: the lists of std::nth_element's chain (n ints each, GLOBAL memory); la / lb: the lists of std::sort's
// ranges and, together, the scratch of the final pass (k - 1 ints each; GL = in global memory, else in LDS).
template <int NT, bool GL>
__device__ __forceinline__ void torch_topk_block(stdorder::E* q, int n, int k, int* la_n, int* lb_n, int* la, int* lb, int* stk,
                                                 unsigned long long* leaf, int leaf_words, int* queue, int* qcount,
                                                 const unsigned long long* tiebits, const uint64_t* canon = nullptr,
                                                 bool rst = false) {
    using namespace stdorder;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (k <= 0 || n <= 0) return;
    unsigned long long rlast = rst && tid == 0 ? __builtin_amdgcn_s_memtime() : 0ull;       // (rst: diagnostic stamps on)
    if (rst && tid == 0) g_replay_t[8] += 1;
    if ((long long)k * 64 <= (long long)n || n > 64 * leaf_words || (k - 1) / 17 + 1 > BLOCK_QCAP) {
        if (tid == 0) torch_topk(q, n, k, stk);                       // heap select / sort (partial_sort), or beyond the tables
        __syncthreads();
        return;
    }
    {   // std::nth_element(q, q + k - 1, q + n): long stretches by the whole workgroup, the rest of the chain by wave 0
        int first = 0, last = n;
        const int nth = k - 1;
        bool done = nth == last;
        int depth = lg2(last - first) * 2;
        while (!done && last - first > 2048 && depth > 0) {            // (workgroup-uniform)
            --depth;
            const int cut = block_partition_pivot<NT>(q, first, last, la_n, lb_n, queue);
            if (cut <= nth) first = cut;
            else last = cut;
        }
        RSTAMP(0);
        if (wave == 0) {
            while (!done && last - first > 3) {
                if (depth == 0) {
                    if (lane == 0) { heap_select(q, first, nth + 1, last); swp(q, first, nth); }
                    done = true;
                    break;
                }
                --depth;
                // (the sort phase's lists are free until then: in LDS - not GL - a partition of this chain is a few LDS
                //  round trips instead of a few L2 round trips)
                const int cut = (!GL && last - first <= k - 1) ? wave_partition_pivot<false>(q, first, last, la, lb, lane)
                                                               : wave_partition_pivot<true>(q, first, last, la_n, lb_n, lane);
                if (cut <= nth) first = cut;
                else last = cut;
            }
            if (!done && lane == 0) insertion_sort(q, first, last);
            wave_fence<true>();
        }
    }
    __syncthreads();
    RSTAMP(1);
    const int last = k - 1;                                            // std::sort(q, q + k - 1)
    for (int w = tid; w < leaf_words; w += NT) leaf[w] = 0ull;
    // range lists: one word per range, first | last << 16 (both < 2^15); the depth budget of std::sort's introsort loop
    // falls by one per partition, i.e. it is the same for every range of a level
    if (tid == 0) {
        qcount[0] = qcount[1] = 0;
        if (last > 16) {
            queue[0] = 0 | (last << 16);
            qcount[0] = 1;
        }
    }
    __syncthreads();
    if (last <= 0) return;
    int depth = lg2(last) * 2;
    int level = 0;
    for (int cur = 0;; cur ^= 1, --depth, ++level) {
        const int ncur = qcount[cur];
        if (ncur == 0) break;
        const int* qc = queue + cur * BLOCK_QCAP;
        int* qn = queue + (cur ^ 1) * BLOCK_QCAP;
        // tie bits j in [b0, b1] of word w (bit j: the canonical ranks j and j + 1 have equal scores)
        auto tie_word = [&](int w, int b0, int b1) -> unsigned long long {
            unsigned long long m = tiebits[w];
            if (w == (b0 >> 6)) m &= ~0ull << (b0 & 63);
            if (w == (b1 >> 6)) m &= ~0ull >> (63 - (b1 & 63));
            return m;
        };
        // A range of introsort holds the elements of final ranks [rf, rl).  When no two neighbouring ranks in there AND
        // across its two ends have equal scores, these are exactly the canonical ranks [rf, rl) and std::sort can only
        // leave them in the one strictly descending order: they are copied from the canonical ranking (`canon`, kept
        // in global memory by the caller) instead of being partitioned level by level - with a handful of equal pairs
        // among thousands of candidates, only the ranges on the way to those pairs are still replayed.
        // (1) ranges of more than BLOCK_RANGE elements: the whole workgroup, one range after the other
        for (int r = 0; r < ncur; ++r) {
            const int rf = qc[r] & 0xFFFF, rl = (int)((unsigned)qc[r] >> 16);
            if (rl - rf <= BLOCK_RANGE || depth <= 0) continue;           // (workgroup-uniform)
            if (tiebits && canon) {
                const int b0 = rf > 0 ? rf - 1 : 0, b1 = rl - 1;
                bool anyb = false;
                for (int w = (b0 >> 6) + tid; w <= (b1 >> 6); w += NT) anyb |= tie_word(w, b0, b1) != 0ull;
                if (!__syncthreads_or(anyb ? 1 : 0)) {
                    for (int x = rf + tid; x < rl; x += NT) {
                        const uint64_t key = canon[x];
                        E o; o.v = key_score(key); o.i = (int)key_pos(key);
                        q[x] = o;
                    }
                    continue;
                }
            }
            const int cut = block_partition_pivot<NT>(q, rf, rl, la_n, lb_n, stk);
            if (tid == 0) {
                if (cut < last) atomicOr(&leaf[cut >> 6], 1ull << (cut & 63));
                if (cut - rf > 16) qn[atomicAdd(&qcount[cur ^ 1], 1)] = rf | (cut << 16);
                if (rl - cut > 16) qn[atomicAdd(&qcount[cur ^ 1], 1)] = cut | (rl << 16);
            }
        }
        // (2) the others: a range per wavefront
        for (int r = wave; r < ncur; r += NT / 64) {
            const int rf = qc[r] & 0xFFFF, rl = (int)((unsigned)qc[r] >> 16);
            if (depth <= 0) {                                          // heap sort of the range: stays as it is afterwards
                if (lane == 0) { make_heap(q, rf, rl); sort_heap(q, rf, rl); }
                continue;
            }
            if (rl - rf > BLOCK_RANGE) continue;                       // (done above)
            if (tiebits && rl - rf <= 64) {
                // No equal neighbours INSIDE a range of at most 64: one in-register wave sort of the elements that are there
                // instead of replaying two more levels of partitions and the leaves.  (With ties inside, the replay goes on.)
                const int lo_w = rf >> 6, hi_w = (rl - 2) >> 6;                              // pairs (j, j + 1), j in [rf, rl - 2]
                unsigned long long any = 0ull;
                for (int w = lo_w; w <= hi_w; ++w) any |= tie_word(w, rf, rl - 2);
                if (any == 0ull) {
                    const int x = rf + lane;
                    const E e = q[x < rl ? x : rf];
                    const uint64_t key = wave_sort_desc(x < rl ? rank_key(e.v, (uint32_t)e.i) : 0ull, lane);
                    if (x < rl) { E o; o.v = key_score(key); o.i = (int)key_pos(key); q[x] = o; }
                    continue;                                          // (a "leaf" of more than 16 elements: the last pass leaves it)
                }
            } else if (tiebits && canon) {
                const int b0 = rf > 0 ? rf - 1 : 0, b1 = rl - 1;
                bool anyb = false;
                for (int w = (b0 >> 6) + lane; w <= (b1 >> 6); w += 64) anyb |= tie_word(w, b0, b1) != 0ull;
                if (__ballot(anyb) == 0ull) {
                    for (int x = rf + lane; x < rl; x += 64) {
                        const uint64_t key = canon[x];
                        E o; o.v = key_score(key); o.i = (int)key_pos(key);
                        q[x] = o;
                    }
                    continue;
                }
            }
            const int cut = wave_partition_pivot<GL>(q, rf, rl, la + rf, lb + rf, lane);
            if (lane == 0) {
                if (cut < last) atomicOr(&leaf[cut >> 6], 1ull << (cut & 63));      // every leaf starts at 0 or at a cut
                if (cut - rf > 16) qn[atomicAdd(&qcount[cur ^ 1], 1)] = rf | (cut << 16);
                if (rl - cut > 16) qn[atomicAdd(&qcount[cur ^ 1], 1)] = cut | (rl << 16);
            }
        }
        __syncthreads();
        if (tid == 0) qcount[cur] = 0;
        __syncthreads();
        RSTAMP(level < 4 ? 2 + level : 6);
    }
    // the final insertion pass, leaf by leaf, an element per thread (see torch_topk_wave)
    E* tmp = reinterpret_cast<E*>(la);
    for (int base = 0; base < last; base += NT) {
        const int x = base + tid;
        if (x < last) {
            const E own = q[x];
            // the leaf of x: [last cut <= x (or 0), first cut > x (or last)).  Only leaves of at most 16 elements are
            // touched, so the cuts that matter lie in [x - 15, x + 16]: 32 bits of the bitmap, two words at most (with
            // whole ranges copied from the canonical ranking the cuts are sparse, and a scan for the nearest one was long)
            const int wb = x < 15 ? 0 : x - 15, wi = wb >> 6, sh = wb & 63, tx = x - wb;
            unsigned long long bits = leaf[wi] >> sh;
            if (sh && wi + 1 < leaf_words) bits |= leaf[wi + 1] << (64 - sh);
            const unsigned long long back = bits & ((2ull << tx) - 1ull);          // cuts at wb .. x
            const unsigned long long fwd = (bits >> (tx + 1)) & 0xFFFFull;           // cuts at x + 1 .. x + 16
            const int sfirst = back ? wb + 63 - __clzll((long long)back) : (wb == 0 ? 0 : -64);   // (-64: further away than 15)
            int e = fwd ? x + 1 + (__ffsll((long long)fwd) - 1) : last;
            if (e > last) e = last;
            int dst = x;
            if (e - sfirst <= 16) {
                int rank = 0;
                for (int j = sfirst; j < e; ++j) {
                    const E o = q[j];
                    const bool greater = gt(o, own);
                    const bool equiv = !greater && !gt(own, o);
                    rank += (greater || (equiv && j < x)) ? 1 : 0;
                }
                dst = sfirst + rank;
            }
            tmp[dst] = own;
        }
    }
    __syncthreads();
    for (int x = tid; x < last; x += NT) q[x] = tmp[x];
    __syncthreads();
    RSTAMP(7);
}
#undef RSTAMP

// sorted and other are the two key arrays (>= L entries each); all threads of the workgroup call this together.
// NOTE: `sorted` serves as scratch meanwhile - on return only sorted[0, m) is defined.
template <int NT>
__device__ __forceinline__ void torch_tie_order(uint64_t* sorted, uint64_t* other, int L, int m, int* stk, int tid) {
    stdorder::E* q = reinterpret_cast<stdorder::E*>(other);
    for (int j = tid; j < L; j += NT) {
        const uint64_t k = sorted[j];
        const int p = (int)key_pos(k);
        q[p].v = key_score(k);
        q[p].i = p;
    }
    __syncthreads();
    if (tid < 64) {
        int* la = reinterpret_cast<int*>(sorted);
        torch_topk_wave<false>(q, L, m, la, la + L, stk, reinterpret_cast<unsigned long long*>(stk + 2 * stdorder::STACK_RANGES), 16, tid);
    }
    __syncthreads();
    for (int j = tid; j < m; j += NT) sorted[j] = rank_key(q[j].v, (uint32_t)q[j].i);
    __syncthreads();
}

struct ScanArgs {
    int tie_order, stk_off;
    const float* lg;       // (b, n, R)
    long long n;
    long long it0, it1;    // iterations [it0, it1) of the loop; it0 > 0 resumes from mem_idx
    int m, i, h, T, n2, use_lds;
    long long* mem_idx;
    float* mem_score;
    int* tie;
    const int* ready;      // persistent launch: number of patches whose logits are in memory (grows while we run);
    int ready_stride;      //   image b polls ready[b * ready_stride] (0: one word for all images, 1: a word per image)
    int ready_words;       //   progress words of the call (1, or b): any of them moving restarts the wait's clock
    unsigned long long wait_ticks;     // persistent launch: longest wait WITHOUT any progress, in 100 MHz ticks (ipsx_set_persistent_wait_ms)
    int* status;           // persistent launch: set to 1 when the wait for `ready` timed out
    const int* cond;       // conditional launch (ipsx_scan_range_if): run only when (*cond & cond_mask) != 0, or nullptr
    int cond_mask;
    int slides;            // images of the call; a launch of fewer workgroups (scan_cam_kernel) gives workgroup w the
                           // images w, w + gridDim.x, ... one after the other
    unsigned long long* plog;   // the resident loops' log (scorer.hip g_persist_log; diagnostic: ipsx_dbg_persist_log)
};

// ipsx_scan_range_if: the recovery launch behind a persistent loop - every workgroup looks at the word the loop sets when
// it gave up waiting and leaves at once when it is clear (workgroup-uniform).
__device__ __forceinline__ bool scan_skipped(const int* cond, int mask) {
    return cond != nullptr && (__hip_atomic_load(cond, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & mask) == 0;
}

// The selection loop kernels: 1024 threads (16 wavefronts) per image - the loop is a chain of short VALU-bound phases
// (two exp + one division per candidate x (head, token)), and 4 waves per SIMD give them 4x the issue slots of a
// 256-thread block.  scan_fast_kernel keeps the candidates' logits and exponentials in LDS (every shape the reference
// ships); scan_large_kernel is the generic one (any head / token count, up to 16,384 candidates, staging through a
// caller workspace).  Same arithmetic order in both and in the oracle: wave-order row sums, ascending sums over heads
// then tokens.
constexpr int SCAN_NT = 1024;
constexpr int SCAN_PF = 4;     // prefetch registers per thread: chunk <= 1024 * 4 floats
__host__ __device__ constexpr int scan_pf(int R, int lch) { return (R == 32 && lch == 8) ? 5 : SCAN_PF; }

// Small candidate sets: rank by counting with P lanes per candidate (P = power of two <= 64,
// P * L <= blockDim): lane `part` counts the keys j = part, part+P, ... that are larger; the partial
// counts are added by an xor butterfly over the P lanes (integer adds: order-free).  O(L^2 / P).
__device__ __forceinline__ void rank_scatter(const uint64_t* src, uint64_t* dst, int L, int P) {
    const int tid = threadIdx.x;
    const int l = tid / P, part = tid & (P - 1);
    int cnt = 0;
    uint64_t k = 0ull;
    if (l < L) {
        k = src[l];
        for (int j = part; j < L; j += P) cnt += (src[j] > k) ? 1 : 0;
    }
    for (int off = P >> 1; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (l < L && part == 0) dst[cnt] = k;
}

// Ranking of L <= 64 * (waves per block) unique keys, descending, without an O(L^2) pass:
//   1. wave w bitonic-sorts keys [64w, 64w+64) in registers (lane shuffles, no barrier) and
//      publishes the sorted run;
//   2. every key's rank = its position in its own run + for each other run the number of
//      larger keys there, found by a branch-free binary search (7 LDS reads per run, the
//      searches of all runs in flight together).
// src/dst hold L keys (dst gets them sorted); runs is scratch for 64 * ceil(L/64) keys.
// partner's key for the exchange with lane ^ J: register to register for every stride (lane_xor_*, ipsx_math.h)
template <int J>
__device__ __forceinline__ uint64_t xor_partner(uint64_t key, int lane) { return lane_xor_u64<J>(key, lane); }

template <int J>
__device__ __forceinline__ float xor_partner_f32(float v, int lane) { return lane_xor_f32<J>(v, lane); }

// the wavefront reductions of the contract (xor butterfly, offsets 32 ... 1) on two values at once
__device__ __forceinline__ void wave_max2(float& a, float& b, int lane) {
    a = nanmax(a, xor_partner_f32<32>(a, lane)); b = nanmax(b, xor_partner_f32<32>(b, lane));
    a = nanmax(a, xor_partner_f32<16>(a, lane)); b = nanmax(b, xor_partner_f32<16>(b, lane));
    a = nanmax(a, xor_partner_f32<8>(a, lane)); b = nanmax(b, xor_partner_f32<8>(b, lane));
    a = nanmax(a, xor_partner_f32<4>(a, lane)); b = nanmax(b, xor_partner_f32<4>(b, lane));
    a = nanmax(a, xor_partner_f32<2>(a, lane)); b = nanmax(b, xor_partner_f32<2>(b, lane));
    a = nanmax(a, xor_partner_f32<1>(a, lane)); b = nanmax(b, xor_partner_f32<1>(b, lane));
}

__device__ __forceinline__ void wave_sum2(float& a, float& b, int lane) {
    a = a + xor_partner_f32<32>(a, lane); b = b + xor_partner_f32<32>(b, lane);
    a = a + xor_partner_f32<16>(a, lane); b = b + xor_partner_f32<16>(b, lane);
    a = a + xor_partner_f32<8>(a, lane); b = b + xor_partner_f32<8>(b, lane);
    a = a + xor_partner_f32<4>(a, lane); b = b + xor_partner_f32<4>(b, lane);
    a = a + xor_partner_f32<2>(a, lane); b = b + xor_partner_f32<2>(b, lane);
    a = a + xor_partner_f32<1>(a, lane); b = b + xor_partner_f32<1>(b, lane);
}

template <int K, int J>
__device__ __forceinline__ uint64_t cmpx(uint64_t key, int lane) {
    const uint64_t other = xor_partner<J>(key, lane);
    const bool take_max = ((lane & K) == 0) == ((lane & J) == 0);
    const bool gt = key > other;
    return (take_max == gt) ? key : other;
}

__device__ __forceinline__ uint64_t wave_sort_desc(uint64_t key, int lane) {
    key = cmpx<2, 1>(key, lane);
    key = cmpx<4, 2>(key, lane); key = cmpx<4, 1>(key, lane);
    key = cmpx<8, 4>(key, lane); key = cmpx<8, 2>(key, lane); key = cmpx<8, 1>(key, lane);
    key = cmpx<16, 8>(key, lane); key = cmpx<16, 4>(key, lane); key = cmpx<16, 2>(key, lane); key = cmpx<16, 1>(key, lane);
    key = cmpx<32, 16>(key, lane); key = cmpx<32, 8>(key, lane); key = cmpx<32, 4>(key, lane); key = cmpx<32, 2>(key, lane);
    key = cmpx<32, 1>(key, lane);
    key = cmpx<64, 32>(key, lane); key = cmpx<64, 16>(key, lane); key = cmpx<64, 8>(key, lane); key = cmpx<64, 4>(key, lane);
    key = cmpx<64, 2>(key, lane); key = cmpx<64, 1>(key, lane);
    return key;
}

__device__ __forceinline__ void rank_runs(const uint64_t* src, uint64_t* dst, uint64_t* runs, int L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nruns = (L + 63) >> 6;
    uint64_t mine = 0ull;
    if (wave < nruns) {
        const int idx = wave * 64 + lane;
        mine = wave_sort_desc(idx < L ? src[idx] : 0ull, lane);     // padding keys (0) sort last
        runs[idx] = mine;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // LDS traffic only
    if (wave < nruns && mine != 0ull) {
        // rank = number of larger keys over ALL runs (in the own run that is the lane index: keys are unique), found
        // by branch-free binary searches, 8 runs at a time with their LDS reads in flight together - no per-run
        // control flow, which would serialise the 7 dependent reads of every search
        int rank = 0;
        for (int r0 = 0; r0 < nruns; r0 += 8) {
            int lo[8];
            const uint64_t* base[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lo[j] = 0;
                base[j] = runs + (r0 + j < nruns ? r0 + j : nruns - 1) * 64;      // clamped: a duplicate search, not counted
            }
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                uint64_t probe[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) probe[j] = base[j][lo[j] + step - 1];
#pragma unroll
                for (int j = 0; j < 8; ++j) lo[j] += (probe[j] > mine) ? step : 0;
            }
            uint64_t last[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) last[j] = base[j][lo[j]];
#pragma unroll
            for (int j = 0; j < 8; ++j) rank += (r0 + j < nruns) ? lo[j] + ((last[j] > mine) ? 1 : 0) : 0;
        }
        dst[rank] = mine;
    }
}

// rank_runs for the fast scan: the same ranking (wave-sorted runs of 64, rank = larger keys over all runs), with the
// searches organised for LATENCY - the loop runs at two waves per SIMD, so a dependent LDS round trip costs more than
// the instructions around it: a 4-ary search (three probes per round, three rounds + one final probe for 64 keys) of
// every run, all runs' probes of a round in flight together, and no probes for runs that do not exist.
template <int NRUN>
__device__ __forceinline__ int rank_in_runs(const uint64_t* runs, uint64_t mine) {
    int lo[NRUN];                                    // number of keys of run j known to be larger than `mine`
#pragma unroll
    for (int j = 0; j < NRUN; ++j) lo[j] = 0;
#pragma unroll
    for (int step = 16; step >= 1; step >>= 2) {     // 64 = 4 * 16 -> 4 * 4 -> 4 * 1
        uint64_t p1[NRUN], p2[NRUN], p3[NRUN];
#pragma unroll
        for (int j = 0; j < NRUN; ++j) {
            const uint64_t* q = runs + j * 64 + lo[j];
            p1[j] = q[step - 1]; p2[j] = q[2 * step - 1]; p3[j] = q[3 * step - 1];
        }
#pragma unroll
        for (int j = 0; j < NRUN; ++j)               // descending run: the probes that are larger form a prefix
            lo[j] += ((p1[j] > mine) ? step : 0) + ((p2[j] > mine) ? step : 0) + ((p3[j] > mine) ? step : 0);
    }
    uint64_t last[NRUN];
#pragma unroll
    for (int j = 0; j < NRUN; ++j) last[j] = runs[j * 64 + lo[j]];
    int rank = 0;
#pragma unroll
    for (int j = 0; j < NRUN; ++j) rank += lo[j] + ((last[j] > mine) ? 1 : 0);
    return rank;
}

__device__ __forceinline__ void rank_runs4(const uint64_t* src, uint64_t* dst, uint64_t* runs, int L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nruns = (L + 63) >> 6;
    uint64_t mine = 0ull;
    if (wave < nruns) {
        const int idx = wave * 64 + lane;
        mine = wave_sort_desc(idx < L ? src[idx] : 0ull, lane);     // padding keys (0) sort last
        runs[idx] = mine;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (wave < nruns && mine != 0ull) {
        // every run is searched, the own one too (there the result is the lane index: keys are unique)
        int rank = 0;
        int r0 = 0;
        for (; r0 + 8 <= nruns; r0 += 8) rank += rank_in_runs<8>(runs + r0 * 64, mine);
        const int left = nruns - r0;                 // workgroup-uniform
        if (left >= 4) { rank += rank_in_runs<4>(runs + r0 * 64, mine); r0 += 4; }
        if (nruns - r0 == 3) rank += rank_in_runs<3>(runs + r0 * 64, mine);
        else if (nruns - r0 == 2) rank += rank_in_runs<2>(runs + r0 * 64, mine);
        else if (nruns - r0 == 1) rank += rank_in_runs<1>(runs + r0 * 64, mine);
        dst[rank] = mine;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The LDS-resident loop (R = H*T a power of two <= 64, (M+I)*R <= 1024*EPT, M+I <= 64*LCH).  One image runs on ONE compute unit, 16 waves on 4 SIMDs: an instruction every thread executes costs
// 16 issue slots, so the loop is bound by instructions per thread and by dependent LDS round trips
// (tools/scan_stamps.py).  What this organisation does about it:
//   * one thread per ELEMENT (candidate l, row r) with r fixed per thread (1024 % R == 0), EPT elements per thread: the
//     row maximum is a v_max_f32 reduction (lane steps R .. 32, then one LDS exchange between the 16 waves) instead of
//     one wave walking a whole row.  NaNs (a NaN must win, the contract's nanmax) are looked for on the side; an
//     iteration that sees one takes the exact key-based reduction instead (workgroup-uniform branch);
//   * exp(x - max) is a function of (x, max) alone, and the maximum of a row rarely moves from one iteration to the
//     next (it belongs to a patch that stays in the memory): the exponentials of the M memory rows travel with the
//     winners and only the I new rows are evaluated - unless the row's maximum changed (bitwise), then that row is
//     recomputed.  det_expf_np leaves out the overflow tests a non-positive argument cannot trigger;
//   * the contract's row sums (lane j adds elements j, j+64, ... in ascending order, then the xor butterfly) read the
//     exponentials back row-wise with all LCH reads of a lane in flight together: R short wave jobs;
//   * the attention weights e / den are formed by all 1024 threads and transposed through LDS for the per-candidate
//     ascending head / token sums;
//   * barriers wait for LDS traffic only (lds_barrier): the prefetch of the next chunk stays in flight across them;
//   * the kernel claims 128 registers per lane: 16 waves x 128 = the whole register file of the compute unit, so no
//     workgroup of the encoder running beside the loop can be placed on it and compete for its issue slots.
// LDS: two logit buffers + two exp buffers of (M+I) x (R+1) floats (the spare exp buffer doubles as the weight
// buffer and as the run scratch of the ranking).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define FAST_STAMP(k)                                                              \
    do {                                                                           \
        if (STAMP) {                                                               \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();            \
            if (tid == 0) { tacc[k] += t_ - tlast; }                               \
            tlast = t_;                                                            \
        }                                                                          \
    } while (0)

// the replay of torch.topk's tie order (rare) lives outside the loop body: inlined, its registers would be the loop's
__device__ __attribute__((noinline)) void tie_order_slow(uint64_t* sorted, uint64_t* other, int L, int m, int* stk) {
    torch_tie_order<SCAN_NT>(sorted, other, L, m, stk, threadIdx.x);
}

// PERSIST: ONE launch for the whole loop, started BEFORE the encoder has produced anything: the kernel waits (bounded)
// until `*a.ready` says the logits of the rows it is about to read exist and reads them past the vector L1 (agent-scope
// loads: the producer is another kernel that finished meanwhile).  Launched onto an idle GPU (ipsx_scan_gate holds the
// producers back until it is resident) it never has to wait for a compute unit to drain, which a workgroup of 16 waves
// does for a long time beside an encoder grid - and it claims 128 registers per lane, i.e. with 16 waves the whole
// register file of its compute unit, so no producer workgroup is placed beside it: a producer launch must then be sized
// for the OTHER compute units (one sized for all 256 runs two workgroups on one of them and takes twice as long; a
// producer workgroup sharing the loop's compute unit is a straggler that costs about as much).
// The row maxima are kept as order-preserving keys (max_key: a NaN wins, as in the contract's nanmax) in LDS, one word per
// row for the memory rows and one for the chunk rows: whoever has the values in registers anyway - the gather of the new
// memory, the prep of the next chunk, the prologue - folds them in with one ds_max_u32 per (wave, row).
template <int R>
__device__ __forceinline__ void fold_row_max(uint32_t key, uint32_t* dst, int lane) {
    if (R <= 8) key = max(key, lane_xor_u32<8>(key, lane));
    if (R <= 16) key = max(key, lane_xor_u32<16>(key, lane));
    key = max(key, lane_xor_u32<32>(key, lane));
    if (lane < R && key != 0u) atomicMax(dst + lane, key);
}

// exp(x - max) down one column of the candidate buffers (row stride ld), an element per thread; not inlined: it runs in
// the minority of iterations, and inlined its registers are the loop's (the same lesson as tie_order_slow)
__device__ __attribute__((noinline)) void exp_column(const float* xcol, float* ecol, int L, int ld, float mx) {
    for (int l = threadIdx.x; l < L; l += blockDim.x) ecol[l * ld] = det_expf_np(xcol[l * ld] - mx);
}

// ... by the threads t0, t0 + nt, ... of a part of the workgroup
__device__ __attribute__((noinline)) void exp_column_part(const float* xcol, float* ecol, int L, int ld, float mx, int t0, int nt) {
    for (int l = t0; l < L; l += nt) ecol[l * ld] = det_expf_np(xcol[l * ld] - mx);
}

template <bool PERSIST>
__device__ __forceinline__ float scan_load(const float* p) {
    if (PERSIST) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// A persistent loop waits here for the rows it is about to read (wave 0 polls; see scan_fast_kernel for the variables it
// expects: a, b, tid, wave, lane, ccount, ready_known).
#define SCAN_WAIT_ROWS(need)                                                                                   \
    do {                                                                                                       \
        if (PERSIST && (long long)(need) > ready_known) {                                                      \
            if (wave == 0) {                                                                                   \
                unsigned long long t0_ = __builtin_amdgcn_s_memrealtime();                                     \
                int v_ = __hip_atomic_load(a.ready + b * a.ready_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                int seen_ = -1;                                                                                \
                while (v_ >= 0 && v_ < (need)) {                                                               \
                    __builtin_amdgcn_s_sleep(16);                                                              \
                    int w_ = lane < a.ready_words ? __hip_atomic_load(a.ready + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0; \
                    for (int o_ = 32; o_ >= 1; o_ >>= 1) w_ += __shfl_xor(w_, o_, 64);                         \
                    const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();                          \
                    if (w_ != seen_) { seen_ = w_; t0_ = now_; }                                               \
                    if (now_ - t0_ > a.wait_ticks) { v_ = -1; break; }                                         \
                    v_ = __hip_atomic_load(a.ready + b * a.ready_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                }                                                                                              \
                if (lane == 0) ccount[6] = v_;                                                     \
            }                                                                                                  \
            lds_barrier();                                                                                     \
            /* the rows the producers published: every wave's loads of them are ordered after the poll that saw the */ \
            /* progress word (one agent-scope acquire per wait - an LDS barrier alone orders nothing in global memory) */ \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                                                 \
            ready_known = ccount[6];                                                               \
            if (ready_known < 0) {                                                                             \
                if (tid == 0) { __hip_atomic_fetch_or(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); a.plog[5] += 1; } \
                return;                                                                                        \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)

// descending bitonic sort of one 32-bit key per lane, the direction of every stage from one bit of `dir` (bit n set: this
// lane keeps the LARGER key in stage n): median(s, partner, all-ones | 0) = max | min.  Duplicates are kept.
template <int N, int J>
__device__ __forceinline__ uint32_t cmpx32(uint32_t s, uint32_t dir, int lane) {
    const uint32_t o = lane_xor_u32<J>(s, lane);
    const uint32_t c = (uint32_t)__builtin_amdgcn_sbfe((int)dir, N, 1);       // v_bfe_i32: 0 or 0xFFFFFFFF
    uint32_t d;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(s), "v"(o), "v"(c));
    return d;
}

// bit n of the result: lane keeps the larger key in stage n of wave_sort_desc_u32 (stage (K, J): ((lane & K) == 0) == ((lane & J) == 0))
__device__ __forceinline__ uint32_t sort_directions(int lane) {
    constexpr int KJ[21][2] = {{2, 1}, {4, 2}, {4, 1}, {8, 4}, {8, 2}, {8, 1}, {16, 8}, {16, 4}, {16, 2}, {16, 1}, {32, 16}, {32, 8},
                               {32, 4}, {32, 2}, {32, 1}, {64, 32}, {64, 16}, {64, 8}, {64, 4}, {64, 2}, {64, 1}};
    uint32_t d = 0u;
#pragma unroll
    for (int n = 0; n < 21; ++n) d |= ((((lane & KJ[n][0]) == 0) == ((lane & KJ[n][1]) == 0)) ? 1u : 0u) << n;
    return d;
}

__device__ __forceinline__ uint32_t wave_sort_desc_u32(uint32_t s, uint32_t dir, int lane) {
    s = cmpx32<0, 1>(s, dir, lane);
    s = cmpx32<1, 2>(s, dir, lane); s = cmpx32<2, 1>(s, dir, lane);
    s = cmpx32<3, 4>(s, dir, lane); s = cmpx32<4, 2>(s, dir, lane); s = cmpx32<5, 1>(s, dir, lane);
    s = cmpx32<6, 8>(s, dir, lane); s = cmpx32<7, 4>(s, dir, lane); s = cmpx32<8, 2>(s, dir, lane); s = cmpx32<9, 1>(s, dir, lane);
    s = cmpx32<10, 16>(s, dir, lane); s = cmpx32<11, 8>(s, dir, lane); s = cmpx32<12, 4>(s, dir, lane); s = cmpx32<13, 2>(s, dir, lane);
    s = cmpx32<14, 1>(s, dir, lane);
    s = cmpx32<15, 32>(s, dir, lane); s = cmpx32<16, 16>(s, dir, lane); s = cmpx32<17, 8>(s, dir, lane); s = cmpx32<18, 4>(s, dir, lane);
    s = cmpx32<19, 2>(s, dir, lane); s = cmpx32<20, 1>(s, dir, lane);
    return s;
}

// ------------------------------------------------------------------ host side: what the units export to each other
struct TopmArgs {
    int tie_order, stk_off;
    const float* scores;
    int L, m, n2;
    long long* top;
    int* tie;
};

// one call of the selection loop as the C ABI hands it over (scorer.hip: scan_range_impl) to the unit whose kernel takes it
struct ScanCall {
    const float* logits;
    int b;
    int64_t n;
    int m, i, h, n_token;
    int64_t it_begin, it_end;
    int64_t* mem_idx;
    float* mem_score;
    int32_t* tie_flag;
    const int32_t* ready;
    int32_t* status;
    void* workspace;
    size_t workspace_bytes;
    void* stream;
    const int32_t* cond;
    int32_t cond_mask;
    int ready_stride, workgroups;
};

// Is this shape the LDS-resident loop's (scan_fast_kernel)?  Otherwise scan_large_kernel takes it (and needs a workspace).
struct FastPlan {
    bool ok;
    int ept, lch;
    size_t lds;
};
FastPlan scan_fast_plan(int m, int i, int h, int n_token);                 // scan_fast.hip
int launch_scan_fast(const ScanCall& c, const FastPlan& fp);               // scan_fast.hip
bool scan_cam_shape(int m, int i, int h, int n_token);                     // scan_cam.hip: BASELINE configs[3]'s shape
int launch_scan_cam(const ScanCall& c);                                    // scan_cam.hip
size_t scan_large_ws_per_image(int m, int i, int h, int n_token);          // scan_large.hip
int launch_scan_large(const ScanCall& c);                                  // scan_large.hip
int scan_large_team(int b, int m, int i, int h, int n_token);              // scan_large.hip: workgroups per image (1: no team)
int launch_topm_large(const TopmArgs& a, int b, void* workspace, size_t workspace_bytes, void* stream);   // scan_large.hip
size_t topm_large_ws_per_row(int l);                                       // scan_large.hip
int scan_large_max_l();                                                    // scan_large.hip

// process-wide switches (defined in scorer.hip)
extern int g_persist_wait_ms;          // ipsx_set_persistent_wait_ms
extern bool g_scan_generic;            // diagnostic (ipsx_dbg_scan_generic): every shape through scan_large_kernel
extern bool g_replay_stamps_on;        // diagnostic (ipsx_dbg_replay_stamps)
extern bool g_scan_direct;             // diagnostic (ipsx_dbg_scan_direct)
extern bool g_scan_team_trunc;         // diagnostic (ipsx_dbg_scan_team_trunc): 0 = the team's main workgroup always merges whole runs
extern int g_scan_team;                // diagnostic (ipsx_dbg_scan_team): -1 the default, 0 one workgroup per image, 2 / 4 / 8
extern bool g_scan_r8;                 // diagnostic (ipsx_dbg_scan_r8): 0 sends scan_cam_kernel's shape through scan_fast_kernel
extern unsigned long long* g_scan_stamps;      // diagnostic (ipsx_dbg_scan_stamps)
unsigned long long* persist_log();     // device address of the resident loops' / the gate's log (scorer.hip: g_persist_log)

inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
constexpr size_t kLdsLimit = 160 * 1024;

}  // namespace ipsx
