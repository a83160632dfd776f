// Shared helpers for the gfx950 kernels of libogmm_hip.so (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/ogmm_hip.h"

namespace ogmm {

int fail(const char* fmt, ...);          // records the message, returns 1
int check_launch(const char* what);      // hipGetLastError() -> 0 / fail()

#define OGMM_REQUIRE(cond, ...) do { if (!(cond)) return ::ogmm::fail(__VA_ARGS__); } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// One-time per-DEVICE setup (hipFuncSetAttribute for > 64 KiB of dynamic LDS, cached device properties): function attributes are per device,
// so a per-process `static bool` would leave a second GPU used by the same process without them.  first() is true once per device ordinal.
struct PerDeviceOnce {
    unsigned char seen[64] = {};
    int device() const { int d = 0; (void)hipGetDevice(&d); return d; }
    bool first() {
        const int d = device();
        if (d < 0 || d >= 64) return true;          // unknown ordinal: do the (cheap, idempotent) setup every time
        if (seen[d]) return false;
        seen[d] = 1;
        return true;
    }
};
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- wave-level reductions (64 lanes), result in every lane
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// exact-rounding fp32 primitives: the compiler must not contract these (discrete choices depend on them)
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }

// |p|^2 as torch's sum(p**2, -1): ((x*x + y*y) + z*z), every product and sum rounded (oracle header)
__device__ __forceinline__ float sqnorm3(float x, float y, float z) {
    return add_rn(add_rn(mul_rn(x, x), mul_rn(y, y)), mul_rn(z, z));
}
// sum((p - c)**2, -1) in the direct form of lib/utils.py:194
__device__ __forceinline__ float sqdist3_direct(float x, float y, float z, float cx, float cy, float cz) {
    const float dx = sub_rn(x, cx), dy = sub_rn(y, cy), dz = sub_rn(z, cz);
    return add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
}

}  // namespace ogmm
