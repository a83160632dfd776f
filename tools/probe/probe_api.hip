// libogmm_probe.so (tools only; never loaded by the product path): the GEMM engines compiled WITH their ablation / clock-probe instantiations
// (-DOGMM_ABLATIONS: precision codes 12..40, 60..89, 100..121 of tools/gemm_bench.py and tools/gemm_v6_check.py), the retired first LDS-DMA engine
// (gemm_f16x3_v6.hip) and the first fp16x3 engine on row-major split planes (gemm_f16x3.hip, OGMM_PREC_F16X3).  Its objects are a second build of the
// engine sources, bound to themselves (-Bsymbolic); it links against libogmm_hip.so only for the shared error / launch helpers.
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

// the same entry as the product's ogmm_gemm_nt (this library's own build of gemm.hip, with every ablation code alive)
extern "C" int ogmm_gemm_nt(const ogmm_gemm* d, void* stream);
extern "C" int ogmm_probe_gemm_nt(const ogmm_gemm* d, void* stream) { return ogmm_gemm_nt(d, stream); }
