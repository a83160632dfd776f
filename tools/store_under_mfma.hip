// tools-only micro-benchmark (round 6): the premise of the one GEMM-engine form that was never built -- two waves per SIMD half a
// tile apart, one wave's output stores under its partner's matrix instructions.  Does a wave's MFMA stream keep its rate while the OTHER wave of its SIMD issues
// the engine's store burst (dword stores, two 128-byte row segments per instruction), on the whole chip, on real data?
//
//   hipcc --offload-arch=gfx950 -O3 tools/store_under_mfma.hip -o /tmp/store_under_mfma && /tmp/store_under_mfma
//
// 256 x n workgroups of 8 waves (waves w and w + 4 share a SIMD).  Role A = waves 0-3: `iters` rounds of 48 v_mfma_f32_32x32x16_f16 on eight accumulators (the
// engine's per-step stream of one wave).  Role B = waves 4-7, by mode:
//   0  idle                  A alone: one wave per SIMD
//   1  the same MFMA loop    both compute: the matrix pipe is shared, A should take twice as long
//   2  store bursts          128 accumulator registers per burst as dword stores into a streaming region (real HBM writes), bursts back to back
//   3  store bursts, A idle  the store rate alone
//   4  store bursts at s_setprio 3 (the storing wave outranks its computing partner in the issue arbitration)
// Reported per mode: mean shader cycles of the A waves and of the B waves (s_memtime), wall time, the A waves' matrix rate, the B waves' store rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ unsigned long long g_cyc[4];          // A cycles, A waves, B cycles, B waves

template <int MODE>
__global__ __launch_bounds__(512) void kern(const f16x8* __restrict__ frag, float* __restrict__ sink, float* __restrict__ region, long long region_floats, int iters,
                                            int bursts) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const bool role_a = wave < 4;
    const long long t0 = clock64();
    if (role_a ? (MODE != 3) : (MODE == 1)) {
        // ---- the MFMA stream: 48 instructions per round on 8 accumulators, operands from registers (random binary16 data)
        f16x8 a[2], b[8];
        a[0] = frag[lane];
        a[1] = frag[64 + lane];
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = frag[128 + j * 64 + lane];
        f32x16 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b[j], acc[j], 0, 0, 0);
        }
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[j][r];
        if (sum == 1.2345f) sink[0] = sum;
    } else if (!role_a && (MODE == 2 || MODE == 3 || MODE == 4)) {
        if (MODE == 4) __builtin_amdgcn_s_setprio(3);
        // ---- the engine's store burst: 128 registers of a 32-row x 256-column slab, lane = column, register = row: two 128-byte row segments per instruction
        float v[8][16];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[j][r] = (float)(lane + j * 16 + r);
        const long long slab = 32LL * 1024;                                        // floats per burst and wave: 32 rows of a 1024-column map (row pitch 1024)
        const long long waves_total = (long long)gridDim.x * 4;
        const long long my = (long long)blockIdx.x * 4 + (wave - 4);
        for (int bu = 0; bu < bursts; ++bu) {
            const long long base = ((my + (long long)bu * waves_total) * slab) % (region_floats - slab);
            float* __restrict__ cp = region + base + 4 * lh * 1024 + lr;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) cp[(long long)((r & 3) + 8 * (r >> 2)) * 1024 + j * 32] = v[j][r] + (float)bu;
        }
    }
    const long long dt = clock64() - t0;
    if (lane == 0) {
        const bool active_a = role_a && MODE != 3, active_b = !role_a && MODE != 0;
        if (active_a) { atomicAdd(&g_cyc[0], (unsigned long long)dt); atomicAdd(&g_cyc[1], 1ull); }
        if (active_b) { atomicAdd(&g_cyc[2], (unsigned long long)dt); atomicAdd(&g_cyc[3], 1ull); }
    }
}

template <int MODE>
static void run(const char* what, const f16x8* frag, float* sink, float* region, long long region_floats, int wgs, int iters, int bursts) {
    unsigned long long z[4] = {0, 0, 0, 0};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern<MODE>, dim3(wgs), dim3(512), 0, 0, frag, sink, region, region_floats, iters, bursts);          // warm-up
    hipDeviceSynchronize();
    hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), z, sizeof(z));
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern<MODE>, dim3(wgs), dim3(512), 0, 0, frag, sink, region, region_floats, iters, bursts);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(z, HIP_SYMBOL(g_cyc), sizeof(z));
    const double a_cyc = z[1] ? (double)z[0] / z[1] : 0.0, b_cyc = z[3] ? (double)z[2] / z[3] : 0.0;
    const double flops = z[1] ? (double)z[1] * iters * 48.0 * 2.0 * 32 * 32 * 16 : 0.0;
    const double bytes = z[3] && MODE >= 2 ? (double)z[3] * bursts * 32.0 * 256 * 4 : 0.0;
    printf("mode %d  %-34s  wall %8.1f us | A waves: %9.0f cycles, %6.1f cycles per MFMA, %7.1f TFLOP/s issued | B waves: %9.0f cycles, stores %6.2f TB/s = %5.1f B/clk/CU at the A clock\n",
           MODE, what, ms * 1e3, a_cyc, a_cyc / (iters * 48.0), flops / (ms * 1e-3) / 1e12, b_cyc, bytes / (ms * 1e-3) / 1e12,
           (b_cyc > 0 && MODE >= 2) ? (double)bursts * 4 * 32 * 256 * 4 / b_cyc : 0.0);
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 400;
    std::vector<_Float16> h(10 * 64 * 8);
    srand(1);
    for (auto& x : h) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
    f16x8* frag;
    float *sink, *region;
    const long long region_floats = 1LL << 30;          // 4 GiB streaming region: the stores are real HBM writes
    hipMalloc(&frag, h.size() * sizeof(_Float16));
    hipMalloc(&sink, 64);
    hipMalloc(&region, region_floats * sizeof(float));
    hipMemcpy(frag, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
    // bursts sized so that the B waves store for about as long as the A waves compute (48 MFMAs x 32 cycles per round against ~8200 cycles per 128-register burst)
    const int bursts = iters * 48 * 32 / 8200 + 1;
    printf("# %d workgroups of 8 waves, %d rounds of 48 MFMAs per A wave, %d store bursts of 128 registers per B wave\n", wgs, iters, bursts);
    run<0>("A: MFMA, B: idle", frag, sink, region, region_floats, wgs, iters, bursts);
    run<1>("A: MFMA, B: MFMA", frag, sink, region, region_floats, wgs, iters, bursts);
    run<2>("A: MFMA, B: store bursts", frag, sink, region, region_floats, wgs, iters, bursts);
    run<3>("A: idle, B: store bursts", frag, sink, region, region_floats, wgs, iters, bursts);
    run<4>("A: MFMA, B: store bursts, prio 3", frag, sink, region, region_floats, wgs, iters, bursts);
    run<0>("A: MFMA, B: idle (again)", frag, sink, region, region_floats, wgs, iters, bursts);
    return 0;
}
