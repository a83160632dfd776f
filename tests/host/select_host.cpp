// Host build of the tie-resolution code (ogmm_amd/csrc/torch_topk_select.h) for the CPU test that checks it
// against torch.topk itself.  g++ -O2 -shared -fPIC select_host.cpp -o libselect_host.so
#include <algorithm>
#include <vector>
#include "../../ogmm_amd/csrc/torch_topk_select.h"

extern "C" void ogmm_test_topk_set(const float* values, int rows, int n, int k, int* out /*[rows][k], sorted by (value, index)*/) {
    std::vector<ogmm_select::Cand> q(n);
    for (int r = 0; r < rows; ++r) {
        for (int j = 0; j < n; ++j) { q[j].v = values[(long long)r * n + j]; q[j].i = j; }
        ogmm_select::torch_topk_smallest_set(q.data(), n, k);
        std::sort(q.begin(), q.begin() + k, [](const ogmm_select::Cand& a, const ogmm_select::Cand& b) { return a.v < b.v || (a.v == b.v && a.i < b.i); });
        for (int j = 0; j < k; ++j) out[(long long)r * k + j] = q[j].i;
    }
}
