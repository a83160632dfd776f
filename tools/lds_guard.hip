// tools-only: a guard kernel for the co-run investigation (HISTORY.md section 4, round 5).  Every workgroup fills its LDS allocation and a set of
// registers with a pattern that encodes (workgroup, address), then spins: re-reading both, recording every word that is not what it wrote
// (workgroup, address, value seen, iteration, hardware id) and restoring it.  Run on one stream beside a suspect kernel on another, it says
// whether the suspect writes into a neighbour's LDS or registers, and what it wrote.
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fno-slp-vectorize -shared -fPIC tools/lds_guard.hip -o tools/lds_guard.so
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
__device__ __forceinline__ uint32_t pat(uint32_t wg, uint32_t i) { return (wg * 2654435761u) ^ (i * 40503u + 0x9e3779b9u); }

__global__ __launch_bounds__(256) void lds_guard_kernel(uint32_t* __restrict__ report, uint32_t* __restrict__ count, int lds_words, int spins,
                                                        int max_reports, int sleep) {
    extern __shared__ uint32_t s[];
    const uint32_t wg = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < lds_words; i += 256) s[i] = pat(wg, i);
    uint32_t r[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) { r[k] = pat(wg ^ 0x5a5a0000u, tid * 32 + k); asm volatile("" : "+v"(r[k])); }
    __syncthreads();
    const uint32_t hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));          // HW_REG_HW_ID, all 32 bits
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));        // HW_REG_XCC_ID
    for (int it = 0; it < spins; ++it) {
        for (int i = tid; i < lds_words; i += 256) {
            const uint32_t v = s[i], want = pat(wg, i);
            if (v != want) {
                const uint32_t slot = atomicAdd(count, 1u);
                if ((int)slot < max_reports) {
                    uint32_t* o = report + (size_t)slot * 8;
                    o[0] = 0; o[1] = wg; o[2] = (uint32_t)i; o[3] = v; o[4] = want; o[5] = (uint32_t)it; o[6] = hw; o[7] = xcc;
                }
                s[i] = want;
            }
        }
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            asm volatile("" : "+v"(r[k]));
            const uint32_t want = pat(wg ^ 0x5a5a0000u, tid * 32 + k);
            if (r[k] != want) {
                const uint32_t slot = atomicAdd(count, 1u);
                if ((int)slot < max_reports) {
                    uint32_t* o = report + (size_t)slot * 8;
                    o[0] = 1; o[1] = wg; o[2] = tid * 32 + k; o[3] = r[k]; o[4] = want; o[5] = (uint32_t)it; o[6] = hw; o[7] = xcc;
                }
                r[k] = want;
            }
        }
        for (int z = 0; z < sleep; ++z) __builtin_amdgcn_s_sleep(8);
    }
}

// A global-memory reader: every thread reads the same read-only words again and again and records a value that differs from the first read
// (the fetch path -- TA / TCP / L2 -- rather than LDS or registers).
__global__ __launch_bounds__(256) void load_guard_kernel(const uint32_t* __restrict__ src, int words, uint32_t* __restrict__ report,
                                                         uint32_t* __restrict__ count, int spins, int max_reports) {
    const uint32_t wg = blockIdx.x, tid = threadIdx.x;
    for (int it = 0; it < spins; ++it)
        for (int i = tid; i < words; i += 256) {
            const uint32_t v = __builtin_nontemporal_load(src + ((size_t)wg * words + i));
            const uint32_t want = pat(wg, i);
            if (v != want) {
                const uint32_t slot = atomicAdd(count, 1u);
                if ((int)slot < max_reports) {
                    uint32_t* o = report + (size_t)slot * 8;
                    o[0] = 2; o[1] = wg; o[2] = (uint32_t)i; o[3] = v; o[4] = want; o[5] = (uint32_t)it; o[6] = 0; o[7] = 0;
                }
            }
        }
}

// The packed-fp32 guard: the inner loop of gmm_feat_mean16_kernel in miniature -- a row of 16 weights read from LDS as four ds_read_b128, each
// multiplied into a pair of accumulators with one v_pk_fma_f32 (the two op_sel forms the compiler makes), back to back -- with the same sums kept
// by scalar fmas, compared every 64 rows.  (This file is compiled with -fno-slp-vectorize so that the scalar shadow stays scalar.)
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void pk_guard_kernel(uint32_t* __restrict__ report, uint32_t* __restrict__ count, int iters, int max_reports,
                                                       int scalar_only) {
    __shared__ __attribute__((aligned(16))) float gs[128][16];
    const uint32_t tid = threadIdx.x, wg = blockIdx.x;
    for (int i = tid; i < 128 * 16; i += 256) gs[i >> 4][i & 15] = (float)((int)((i * 2654435761u + wg) >> 20 & 255) - 128) * (1.0f / 16384.0f);
    __syncthreads();
    f2 f; f.x = 1.0f + 0.001f * (float)(tid & 63); f.y = -1.0f + 0.002f * (float)(tid & 63);
    f2 acc[16]; float lo[16], hi[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc[j].x = lo[j] = 0.0f; acc[j].y = hi[j] = 0.0f; }
    for (int it = 0; it < iters; ++it) {
        const float4* g4 = reinterpret_cast<const float4*>(gs[(it * 4 + (tid >> 6)) & 127]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 g = g4[q];
            f2 g01, g23; g01.x = g.x; g01.y = g.y; g23.x = g.z; g23.y = g.w;
            if (!scalar_only) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[4 * q]) : "v"(f), "v"(g01));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[4 * q + 1]) : "v"(f), "v"(g01));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[4 * q + 2]) : "v"(f), "v"(g23));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[4 * q + 3]) : "v"(f), "v"(g23));
            } else {
                acc[4 * q].x = fmaf(f.x, g.x, acc[4 * q].x);         acc[4 * q].y = fmaf(f.y, g.x, acc[4 * q].y);
                acc[4 * q + 1].x = fmaf(f.x, g.y, acc[4 * q + 1].x); acc[4 * q + 1].y = fmaf(f.y, g.y, acc[4 * q + 1].y);
                acc[4 * q + 2].x = fmaf(f.x, g.z, acc[4 * q + 2].x); acc[4 * q + 2].y = fmaf(f.y, g.z, acc[4 * q + 2].y);
                acc[4 * q + 3].x = fmaf(f.x, g.w, acc[4 * q + 3].x); acc[4 * q + 3].y = fmaf(f.y, g.w, acc[4 * q + 3].y);
            }
            lo[4 * q] = fmaf(f.x, g.x, lo[4 * q]);         hi[4 * q] = fmaf(f.y, g.x, hi[4 * q]);
            lo[4 * q + 1] = fmaf(f.x, g.y, lo[4 * q + 1]); hi[4 * q + 1] = fmaf(f.y, g.y, hi[4 * q + 1]);
            lo[4 * q + 2] = fmaf(f.x, g.z, lo[4 * q + 2]); hi[4 * q + 2] = fmaf(f.y, g.z, hi[4 * q + 2]);
            lo[4 * q + 3] = fmaf(f.x, g.w, lo[4 * q + 3]); hi[4 * q + 3] = fmaf(f.y, g.w, hi[4 * q + 3]);
        }
        if ((it & 63) == 63) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const bool b0 = __float_as_uint(acc[j].x) != __float_as_uint(lo[j]), b1 = __float_as_uint(acc[j].y) != __float_as_uint(hi[j]);
                if (b0 || b1) {
                    const uint32_t slot = atomicAdd(count, 1u);
                    if ((int)slot < max_reports) {
                        uint32_t* o = report + (size_t)slot * 8;
                        o[0] = 3; o[1] = wg; o[2] = tid; o[3] = __float_as_uint(b1 ? acc[j].y : acc[j].x); o[4] = __float_as_uint(b1 ? hi[j] : lo[j]);
                        o[5] = (uint32_t)it; o[6] = (uint32_t)j; o[7] = (b0 ? 1u : 0u) | (b1 ? 2u : 0u);
                    }
                }
                acc[j].x = lo[j] = 0.0f; acc[j].y = hi[j] = 0.0f;
            }
        }
    }
}

__global__ void fill_kernel(uint32_t* dst, int words) {
    const uint32_t wg = blockIdx.x;
    for (int i = threadIdx.x; i < words; i += 256) dst[(size_t)wg * words + i] = pat(wg, i);
}
}  // namespace

extern "C" int lds_guard(uint32_t* report, uint32_t* count, int workgroups, int lds_bytes, int spins, int max_reports, int sleep, void* stream) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lds_guard_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(lds_guard_kernel, dim3(workgroups), dim3(256), lds_bytes, (hipStream_t)stream, report, count, lds_bytes / 4, spins, max_reports,
                       sleep);
    return (int)hipGetLastError();
}
extern "C" int pk_guard(uint32_t* report, uint32_t* count, int workgroups, int iters, int max_reports, int scalar_only, void* stream) {
    hipLaunchKernelGGL(pk_guard_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, report, count, iters, max_reports, scalar_only);
    return (int)hipGetLastError();
}
extern "C" int load_guard_fill(uint32_t* dst, int workgroups, int words, void* stream) {
    hipLaunchKernelGGL(fill_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, dst, words);
    return (int)hipGetLastError();
}
extern "C" int load_guard(const uint32_t* src, int workgroups, int words, uint32_t* report, uint32_t* count, int spins, int max_reports, void* stream) {
    hipLaunchKernelGGL(load_guard_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, src, words, report, count, spins, max_reports);
    return (int)hipGetLastError();
}
