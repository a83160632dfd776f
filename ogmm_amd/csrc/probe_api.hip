// libogmm_probe.so (tools only): entry to the retired first LDS-DMA GEMM engine (gemm_f16x3_v6.hip) and its ablation builds, selected by
// ogmm_gemm.precision codes 60..89 (tools/gemm_v6_check.py).  Links against libogmm_hip.so for the shared error / launch helpers.
#include "gemm_common.h"

namespace ogmm {
bool gemm_f16x3_v6_applicable(const ogmm_gemm& g);
int gemm_nt_f16x3_v6(const ogmm_gemm& g, hipStream_t s);
}

extern "C" int ogmm_probe_gemm_v6(const ogmm_gemm* d, void* stream) {
    OGMM_REQUIRE(d != nullptr && d->A && d->B_hi && d->B_lo && d->C, "ogmm_probe_gemm_v6: A, the fragment-major B image and C required");
    OGMM_REQUIRE(ogmm::gemm_f16x3_v6_applicable(*d), "ogmm_probe_gemm_v6: shape not taken by the v6 engine");
    return ogmm::gemm_nt_f16x3_v6(*d, ogmm::as_stream(stream));
}
