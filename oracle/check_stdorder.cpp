// check_stdorder.cpp - TEST INFRASTRUCTURE: ips_amd/csrc/ipsx_stdorder.h (the restatement the kernels run under score
// ties) against libstdc++'s own std::partial_sort / std::nth_element / std::sort, called exactly as ATen's CPU
// top-k calls them (ATen/native/TopKImpl.h:45-68), on tie-heavy random inputs.  Exit code 0 = identical everywhere.
//   make -C oracle check_stdorder && oracle/check_stdorder [cases]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>
#include <utility>
#include <vector>

#include "../ips_amd/csrc/ipsx_stdorder.h"

using P = std::pair<float, int64_t>;

static void aten_topk(std::vector<P>& q, int k) {
    const int n = (int)q.size();
    auto gt = [](const P& x, const P& y) -> bool {
        return ((std::isnan(x.first) && !std::isnan(y.first)) || (x.first > y.first));
    };
    if ((int64_t)k * 64 <= n) {
        std::partial_sort(q.begin(), q.begin() + k, q.end(), gt);
    } else {
        std::nth_element(q.begin(), q.begin() + k - 1, q.end(), gt);
        std::sort(q.begin(), q.begin() + k - 1, gt);
    }
}

// ---- the formulation the device runs on a whole wavefront (csrc/scorer.hip torch_topk_wave): the Hoare partition as
// "pair the t-th element from the left that stops the upward scan with the t-th from the right that stops the downward
// scan, while the left one lies before the right one", the introsort leaves sorted independently.  Written here as
// plain loops - the specification - and held against the move-for-move restatement above.
namespace lists {
using ipsx::stdorder::E;
using ipsx::stdorder::gt;

static int partition(E* q, int first, int last, int pivot, std::vector<int>& A, std::vector<int>& B) {
    const E P = q[pivot];
    A.clear(); B.clear();
    for (int x = first; x < last; ++x) if (!gt(q[x], P)) A.push_back(x);          // stops the upward scan
    for (int x = last - 1; x >= first; --x) if (!gt(P, q[x])) B.push_back(x);     // stops the downward scan
    size_t t = 0;
    while (t < A.size() && t < B.size() && A[t] < B[t]) ++t;
    for (size_t u = 0; u < t; ++u) std::swap(q[A[u]], q[B[u]]);
    if (t < A.size() && (t == 0 || A[t] < B[t - 1])) return A[t];
    return B[t - 1];
}

static int partition_pivot(E* q, int first, int last, std::vector<int>& A, std::vector<int>& B) {
    const int mid = first + (last - first) / 2;
    ipsx::stdorder::move_median_to_first(q, first, first + 1, mid, last - 1);
    return partition(q, first + 1, last, first, A, B);
}

static void leaf_sort(E* q, int s, int e) {          // linear insertion that stops at the leaf's first element
    for (int i = s + 1; i < e; ++i) {
        const E val = q[i];
        int j = i;
        while (j > s && gt(val, q[j - 1])) { q[j] = q[j - 1]; --j; }
        q[j] = val;
    }
}

static void topk(E* q, int n, int k, int* stk) {
    using namespace ipsx::stdorder;
    if (k <= 0 || n <= 0) return;
    if ((long long)k * 64 <= (long long)n) { partial_sort(q, 0, k, n); return; }
    std::vector<int> A, B;
    {   // nth_element(q, 0, k - 1, n)
        int first = 0, last = n;
        const int nth = k - 1;
        bool done = (first == last || nth == last);
        int depth = done ? 0 : lg2(last - first) * 2;
        while (!done && last - first > 3) {
            if (depth == 0) { heap_select(q, first, nth + 1, last); swp(q, first, nth); done = true; break; }
            --depth;
            const int cut = partition_pivot(q, first, last, A, B);
            if (cut <= nth) first = cut; else last = cut;
        }
        if (!done) insertion_sort(q, first, last);
    }
    {   // sort(q, 0, k - 1): leaves of the introsort loop, each insertion-sorted on its own
        const int first = 0, last = k - 1;
        if (first == last) return;
        std::vector<char> leaf(last + 1, 0);
        int sp = 0;
        stk[0] = first; stk[1] = last; stk[2] = lg2(last - first) * 2; sp = 1;
        while (sp > 0) {
            --sp;
            int rf = stk[3 * sp], rl = stk[3 * sp + 1], depth = stk[3 * sp + 2];
            bool heap = false;
            while (rl - rf > 16) {
                if (depth == 0) { make_heap(q, rf, rl); sort_heap(q, rf, rl); heap = true; break; }
                --depth;
                const int cut = partition_pivot(q, rf, rl, A, B);
                stk[3 * sp] = cut; stk[3 * sp + 1] = rl; stk[3 * sp + 2] = depth; ++sp;
                rl = cut;
            }
            (void)heap;
            leaf[rf] = 1;                  // [rf, rl) is a leaf (or a heap-sorted range: sorted already, one more boundary)
            if (rl <= last) leaf[rl] = 1;
        }
        int s = first;
        for (int x = first + 1; x <= last; ++x)
            if (x == last || leaf[x]) { leaf_sort(q, s, x); s = x; }
    }
}
}  // namespace lists

int main(int argc, char** argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 200000;
    std::mt19937_64 rng(12345);
    long bad = 0, bad_lists = 0, checked = 0;
    std::vector<int> stk(3 * ipsx::stdorder::STACK_RANGES);
    for (long c = 0; c < cases; ++c) {
        const int n = 1 + (int)(rng() % (c % 50 == 0 ? 2100 : 600));
        const int k = 1 + (int)(rng() % n);
        const int kinds = 1 + (int)(rng() % 6);
        const int distinct = kinds == 1 ? 1 : (kinds == 2 ? 2 : (kinds == 3 ? 5 : (kinds == 4 ? 37 : 100000)));
        const bool nans = rng() % 9 == 0;
        const int pattern = (int)(rng() % 4);      // random / ascending / descending / organ pipe: depth-limit paths
        std::vector<P> a(n);
        std::vector<ipsx::stdorder::E> b(n);
        for (int i = 0; i < n; ++i) {
            float v;
            if (pattern == 0) v = (float)(rng() % distinct) * 0.25f;
            else if (pattern == 1) v = (float)((i * (int64_t)distinct) / n);
            else if (pattern == 2) v = (float)(((n - 1 - i) * (int64_t)distinct) / n);
            else v = (float)(std::min(i, n - 1 - i) % distinct);
            if (nans && rng() % 7 == 0) v = std::numeric_limits<float>::quiet_NaN();
            if (rng() % 31 == 0) v = -v;
            a[i] = P(v, i);
            b[i].v = v; b[i].i = i;
        }
        std::vector<ipsx::stdorder::E> w(b);
        aten_topk(a, k);
        ipsx::stdorder::torch_topk(b.data(), n, k, stk.data());
        lists::topk(w.data(), n, k, stk.data());
        for (int j = 0; j < k; ++j) {
            ++checked;
            if (a[j].second != b[j].i) { ++bad; break; }
            if (a[j].second != w[j].i) { ++bad_lists; break; }
        }
    }
    // the median-of-3 killer sequence drives introsort / introselect to their heap fallbacks
    for (int n : {64, 257, 1024, 4096}) {
        for (int k : {n / 2, n - 1, n}) {
            std::vector<float> v(n);
            const int half = n / 2;
            for (int i = 0; i < half; ++i) { v[2 * i] = (float)(i + 1); v[2 * i + 1] = (float)(half + i + 1); }
            if (n & 1) v[n - 1] = (float)n;
            std::vector<P> a(n);
            std::vector<ipsx::stdorder::E> b(n);
            for (int i = 0; i < n; ++i) { a[i] = P(-v[i], i); b[i].v = -v[i]; b[i].i = i; }
            std::vector<ipsx::stdorder::E> w(b);
            aten_topk(a, k);
            ipsx::stdorder::torch_topk(b.data(), n, k, stk.data());
            lists::topk(w.data(), n, k, stk.data());
            for (int j = 0; j < k; ++j) if (a[j].second != b[j].i) { ++bad; break; }
            for (int j = 0; j < k; ++j) if (a[j].second != w[j].i) { ++bad_lists; break; }
        }
    }
    printf("cases %ld  compared %ld  mismatching cases %ld  (wavefront formulation: %ld)\n", cases, checked, bad, bad_lists);
    return (bad || bad_lists) ? 1 : 0;
}
