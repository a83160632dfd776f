// tools-only: instruction-rate probe for the binary16 split sequences (round 5).  The paired form of the thin weight gradient's in-register split
// (v_cvt_pk_f16_f32 + v_cvt_f32_f16_sdwa src0_sel:WORD_1) ran 4.7x slower than the per-element form; this measures the candidates in isolation.
// build + run (GPU box): hipcc --offload-arch=gfx950 -O2 tools/sdwa_rate.hip -o /tmp/sdwa_rate && /tmp/sdwa_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed) {
    const float step = seed * 1e-3f;
    float t0 = seed + threadIdx.x * step, t1 = seed * 0.5f + threadIdx.x * 2.0f * step, acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            uint32_t pk; float r0, r1;
            if (KIND == 0) {          // per element: v_cvt_f16_f32 x2, v_cvt_f32_f16 x2
                uint32_t h0, h1;
                asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h0) : "v"(t0));
                asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h1) : "v"(t1));
                asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r0) : "v"(h0));
                asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r1) : "v"(h1));
            } else if (KIND == 1) {   // paired: v_cvt_pk_f16_f32, v_cvt_f32_f16 (low half), v_cvt_f32_f16_sdwa WORD_1
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(t0), "v"(t1));
                asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r0) : "v"(pk));
                asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r1) : "v"(pk));
            } else if (KIND == 2) {   // paired, high half through a shift
                uint32_t hi;
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(t0), "v"(t1));
                asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r0) : "v"(pk));
                asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(hi) : "v"(pk));
                asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r1) : "v"(hi));
            } else {                  // only v_cvt_pk_f16_f32
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(t0), "v"(t1));
                r0 = __uint_as_float(pk); r1 = 0.0f;
            }
            acc += r0 + r1;
            t0 += step; t1 -= step * 0.001f;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    float* out; hipMalloc(&out, 4);
    const char* names[] = {"per element (4 conversions)", "paired + sdwa WORD_1", "paired + shift", "v_cvt_pk_f16_f32 only"};
    for (int pass = 0; pass < 3; ++pass)
        for (int kind = 0; kind < 4; ++kind) {
            const float seed = pass == 2 ? 1.0e-5f : 1.0f;          // pass 2: every result is a binary16 SUBNORMAL
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int iters = 20000;
            hipEventRecord(e0, 0);
            if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(2048), dim3(256), 0, 0, out, iters, seed);
            if (kind == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(2048), dim3(256), 0, 0, out, iters, seed);
            if (kind == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(2048), dim3(256), 0, 0, out, iters, seed);
            if (kind == 3) hipLaunchKernelGGL(rate_kernel<3>, dim3(2048), dim3(256), 0, 0, out, iters, seed);
            hipEventRecord(e1, 0); hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (pass) printf("%s %-32s %8.2f ms  (%.1f ns per 16-pair group and wave)\n", pass == 2 ? "subnormal results:" : "normal results:   ", names[kind], ms, ms * 1e6 / (2048.0 * 4 / (256 * 4 * 2.0) * iters));
        }
    return 0;
}
