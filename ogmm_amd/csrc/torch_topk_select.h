// Which k candidates does torch.topk(dist, k, largest=False) keep when distances TIE exactly at rank k?
//
// lib/utils.py:43 calls torch.topk on CPU; PyTorch's CPU kernel (ATen TopKImpl.h) fills a vector of
// (value, index) pairs in index order and runs, per row,
//     k * 64 <= n :  std::partial_sort(begin, begin + k, end, value-less)      -> heap select
//     otherwise   :  std::nth_element(begin, begin + k - 1, end, value-less)   -> introselect
// (then sorts the kept prefix).  Neither is stable, so among candidates whose fp32 distance equals the k-th
// smallest, the survivors are decided by the element moves of GNU libstdc++ (PyTorch 2.10 is built with
// GCC 11).  The expanded distance formula quantises neighbour distances to ~3e-8, so ~6e-5 of all rows hold
// such a tie, and a single swapped neighbour moves the final rotation by ~2e-5 rad: to stay within 1e-5 of the
// reference the selection below re-states those two library algorithms move for move (index-based instead of
// iterator-based; comparisons look at the value only).  It runs only for rows flagged by the fast kernel.
//
// Host + device: the same code is compiled by g++ for the CPU test that checks it against torch.topk.
#pragma once

#if defined(__HIPCC__)
#define OGMM_HD __host__ __device__
#else
#define OGMM_HD
#endif

namespace ogmm_select {

struct Cand { float v; int i; };

OGMM_HD inline bool less_v(const Cand& a, const Cand& b) { return a.v < b.v; }
OGMM_HD inline void swap_c(Cand& a, Cand& b) { const Cand t = a; a = b; b = t; }

// ---- binary max-heap on q[0..len) (value order), hole-based sift as in libstdc++'s adjust/push pair
OGMM_HD inline void sift(Cand* q, int hole, int len, Cand value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (less_v(q[child], q[child - 1])) --child;
        q[hole] = q[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        q[hole] = q[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && less_v(q[parent], value)) {
        q[hole] = q[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    q[hole] = value;
}

// keeps the `mid` smallest of q[0..n) in q[0..mid) (as a heap): make_heap, then every later element that is
// strictly smaller than the current maximum replaces it
OGMM_HD inline void heap_select(Cand* q, int mid, int n) {
    if (mid >= 2) {
        for (int parent = (mid - 2) / 2;; --parent) {
            sift(q, parent, mid, q[parent]);
            if (parent == 0) break;
        }
    }
    for (int i = mid; i < n; ++i) {
        if (less_v(q[i], q[0])) {
            const Cand value = q[i];
            q[i] = q[0];
            sift(q, 0, mid, value);
        }
    }
}

OGMM_HD inline void median_to_first(Cand* q, int result, int a, int b, int c) {
    if (less_v(q[a], q[b])) {
        if (less_v(q[b], q[c])) swap_c(q[result], q[b]);
        else if (less_v(q[a], q[c])) swap_c(q[result], q[c]);
        else swap_c(q[result], q[a]);
    } else if (less_v(q[a], q[c])) swap_c(q[result], q[a]);
    else if (less_v(q[b], q[c])) swap_c(q[result], q[c]);
    else swap_c(q[result], q[b]);
}

OGMM_HD inline int partition_around(Cand* q, int first, int last, int pivot) {
    for (;;) {
        while (less_v(q[first], q[pivot])) ++first;
        --last;
        while (less_v(q[pivot], q[last])) --last;
        if (!(first < last)) return first;
        swap_c(q[first], q[last]);
        ++first;
    }
}

OGMM_HD inline void insertion_sort(Cand* q, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        const Cand val = q[i];
        if (less_v(val, q[first])) {
            for (int j = i; j > first; --j) q[j] = q[j - 1];
            q[first] = val;
        } else {
            int j = i;
            while (less_v(val, q[j - 1])) { q[j] = q[j - 1]; --j; }
            q[j] = val;
        }
    }
}

// std::nth_element's loop from a given state (range [first, last), remaining depth budget)
OGMM_HD inline void introselect_range(Cand* q, int nth, int first, int last, int depth) {
    while (last - first > 3) {
        if (depth == 0) {
            heap_select(q + first, nth + 1 - first, last - first);
            swap_c(q[first], q[nth]);
            return;
        }
        --depth;
        const int mid = first + (last - first) / 2;
        median_to_first(q, first, first + 1, mid, last - 1);
        const int cut = partition_around(q, first + 1, last, first);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    insertion_sort(q, first, last);
}

// std::nth_element(q, q + nth, q + n)
OGMM_HD inline void introselect(Cand* q, int nth, int n) {
    int lg = 0;
    for (int m = n; m > 1; m >>= 1) ++lg;
    introselect_range(q, nth, 0, n, 2 * lg);
}

// After the call q[0..k) is the set torch.topk(largest=False) keeps for a row of n candidates given in index order.
OGMM_HD inline void torch_topk_smallest_set(Cand* q, int n, int k) {
    if (k <= 0 || n <= 0) return;
    if ((long long)k * 64 <= n) heap_select(q, k, n);
    else introselect(q, k - 1, n);
}

}  // namespace ogmm_select
