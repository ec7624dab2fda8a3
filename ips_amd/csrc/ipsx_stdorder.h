// ipsx_stdorder.h - the order torch.topk(largest=True, sorted=True) returns on CPU, restated so that it can run
// on the device (and, for the check against the real thing, on the host).
//
// ATen's CPU top-k (reference call site architecture/ips_net.py:148 -> ATen/native/TopKImpl.h:16-96) fills a
// vector of (value, index) pairs and calls, with the comparator  gt(x, y) = (isnan(x) && !isnan(y)) || x > y :
//     k * 64 <= n :  std::partial_sort(q, q + k, q + n, gt)
//     otherwise   :  std::nth_element(q, q + k - 1, q + n, gt);  std::sort(q, q + k - 1, gt)
// Neither is stable, so with EQUAL scores the returned order is whatever libstdc++'s introselect / introsort /
// heap routines leave behind.  The selection loop feeds its output order back into the next
// iteration, so reproducing the reference under ties means reproducing these routines move for move.  What
// follows is a restatement of the published libstdc++ algorithms (bits/stl_algo.h, bits/stl_heap.h: median-of-
// three to the front, unguarded Hoare partition, insertion sort below 16 / 3 elements, heap select / sort with
// the hole-sifting __adjust_heap), on a plain array of pairs, without recursion (explicit stack) - checked
// against std:: itself on millions of tie-heavy inputs by oracle/check_stdorder.cpp.
//
// Sequential as written; the kernels run it only in iterations whose first k+1 ranked scores contain equal neighbours
// (otherwise the answer is the strict descending order every algorithm agrees on), and then on a whole wavefront:
// scorer.hip torch_topk_wave evaluates the Hoare partitions with ballots and the final insertion pass leaf by leaf,
// and calls the routines below on one lane for what is left (median of three, heap fallbacks, partial_sort).
#pragma once

#if defined(__HIPCC__)
#define IPSX_HD __host__ __device__ __forceinline__
#else
#define IPSX_HD inline
#endif

namespace ipsx {
namespace stdorder {

struct E {
    float v;
    int i;
};

IPSX_HD bool gt(const E& x, const E& y) {
    const bool xn = x.v != x.v, yn = y.v != y.v;
    return (xn && !yn) || (x.v > y.v);
}

IPSX_HD void swp(E* q, int a, int b) {
    const E t = q[a];
    q[a] = q[b];
    q[b] = t;
}

IPSX_HD int lg2(int n) {          // std::__lg: floor(log2(n)), n > 0
    int k = 0;
    while (n > 1) { n >>= 1; ++k; }
    return k;
}

// ---- heap routines on q[first .. first+len)
IPSX_HD void push_heap(E* q, int first, int hole, int top, E value) {
    int parent = (hole - 1) / 2;
    while (hole > top && gt(q[first + parent], value)) {
        q[first + hole] = q[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    q[first + hole] = value;
}

IPSX_HD void adjust_heap(E* q, int first, int hole, int len, E value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (gt(q[first + child], q[first + child - 1])) --child;
        q[first + hole] = q[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        q[first + hole] = q[first + child - 1];
        hole = child - 1;
    }
    push_heap(q, first, hole, top, value);
}

IPSX_HD void make_heap(E* q, int first, int last) {
    const int len = last - first;
    if (len < 2) return;
    int parent = (len - 2) / 2;
    while (true) {
        const E value = q[first + parent];
        adjust_heap(q, first, parent, len, value);
        if (parent == 0) return;
        --parent;
    }
}

IPSX_HD void pop_heap(E* q, int first, int last, int result) {      // heap is [first, last); result receives the top
    const E value = q[result];
    q[result] = q[first];
    adjust_heap(q, first, 0, last - first, value);
}

IPSX_HD void heap_select(E* q, int first, int middle, int last) {
    make_heap(q, first, middle);
    for (int i = middle; i < last; ++i)
        if (gt(q[i], q[first])) pop_heap(q, first, middle, i);
}

IPSX_HD void sort_heap(E* q, int first, int last) {
    while (last - first > 1) {
        --last;
        pop_heap(q, first, last, last);
    }
}

// ---- insertion sorts
IPSX_HD void unguarded_linear_insert(E* q, int last) {
    const E val = q[last];
    int next = last - 1;
    while (gt(val, q[next])) {
        q[last] = q[next];
        last = next;
        --next;
    }
    q[last] = val;
}

IPSX_HD void insertion_sort(E* q, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (gt(q[i], q[first])) {
            const E val = q[i];
            for (int j = i; j > first; --j) q[j] = q[j - 1];      // move_backward(first, i, i + 1)
            q[first] = val;
        } else {
            unguarded_linear_insert(q, i);
        }
    }
}

// ---- partition around the median of three, pivot parked at `first`
IPSX_HD void move_median_to_first(E* q, int result, int a, int b, int c) {
    if (gt(q[a], q[b])) {
        if (gt(q[b], q[c])) swp(q, result, b);
        else if (gt(q[a], q[c])) swp(q, result, c);
        else swp(q, result, a);
    } else if (gt(q[a], q[c])) {
        swp(q, result, a);
    } else if (gt(q[b], q[c])) {
        swp(q, result, c);
    } else {
        swp(q, result, b);
    }
}

IPSX_HD int unguarded_partition(E* q, int first, int last, int pivot) {
    while (true) {
        while (gt(q[first], q[pivot])) ++first;
        --last;
        while (gt(q[pivot], q[last])) --last;
        if (!(first < last)) return first;
        swp(q, first, last);
        ++first;
    }
}

IPSX_HD int unguarded_partition_pivot(E* q, int first, int last) {
    const int mid = first + (last - first) / 2;
    move_median_to_first(q, first, first + 1, mid, last - 1);
    return unguarded_partition(q, first + 1, last, first);
}

// ---- std::nth_element
IPSX_HD void nth_element(E* q, int first, int nth, int last) {
    if (first == last || nth == last) return;
    int depth = lg2(last - first) * 2;
    while (last - first > 3) {
        if (depth == 0) {
            heap_select(q, first, nth + 1, last);
            swp(q, first, nth);
            return;
        }
        --depth;
        const int cut = unguarded_partition_pivot(q, first, last);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    insertion_sort(q, first, last);
}

// ---- std::sort: introsort loop (right part "recursed" through an explicit stack, left part iterated - the
// order of the two is immaterial, they touch disjoint ranges), then the final insertion sort
// `stk`: 3 * STACK_RANGES ints of scratch (at most 2 * lg(n) ranges are ever pending).
constexpr int STACK_RANGES = 64;

IPSX_HD void sort(E* q, int first, int last, int* stk) {
    if (first == last) return;
    int sp = 0;
    stk[0] = first; stk[1] = last; stk[2] = lg2(last - first) * 2;
    sp = 1;
    while (sp > 0) {
        --sp;
        int rf = stk[3 * sp], rl = stk[3 * sp + 1], depth = stk[3 * sp + 2];
        while (rl - rf > 16) {
            if (depth == 0) {                         // heap sort of the range
                make_heap(q, rf, rl);
                sort_heap(q, rf, rl);
                break;
            }
            --depth;
            const int cut = unguarded_partition_pivot(q, rf, rl);
            stk[3 * sp] = cut; stk[3 * sp + 1] = rl; stk[3 * sp + 2] = depth;
            ++sp;
            rl = cut;
        }
    }
    if (last - first > 16) {
        insertion_sort(q, first, first + 16);
        for (int i = first + 16; i != last; ++i) unguarded_linear_insert(q, i);
    } else {
        insertion_sort(q, first, last);
    }
}

// ---- std::partial_sort
IPSX_HD void partial_sort(E* q, int first, int middle, int last) {
    heap_select(q, first, middle, last);
    sort_heap(q, first, middle);
}

// q[0..n) = (score, position) in candidate order on entry; q[0..k) = torch.topk's answer on return
IPSX_HD void torch_topk(E* q, int n, int k, int* stk) {
    if (k <= 0 || n <= 0) return;
    if ((long long)k * 64 <= (long long)n) {
        partial_sort(q, 0, k, n);
    } else {
        nth_element(q, 0, k - 1, n);
        sort(q, 0, k - 1, stk);
    }
}

}  // namespace stdorder
}  // namespace ipsx
