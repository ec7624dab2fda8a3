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

int main(int argc, char** argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 200000;
    std::mt19937_64 rng(12345);
    long bad = 0, checked = 0;
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
        aten_topk(a, k);
        ipsx::stdorder::torch_topk(b.data(), n, k, stk.data());
        for (int j = 0; j < k; ++j) {
            ++checked;
            if (a[j].second != b[j].i) { ++bad; break; }
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
            aten_topk(a, k);
            ipsx::stdorder::torch_topk(b.data(), n, k, stk.data());
            for (int j = 0; j < k; ++j) if (a[j].second != b[j].i) { ++bad; break; }
        }
    }
    printf("cases %ld  compared %ld  mismatching cases %ld\n", cases, checked, bad);
    return bad ? 1 : 0;
}
