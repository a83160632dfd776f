// tools-only: minimal reproducer for the co-run finding of round 5 (HISTORY.md section 4).  A "victim" kernel keeps computing the same fused
// multiply-add two ways -- one packed-fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) and two scalar v_fma_f32 -- and records
// every iteration where they disagree (iteration, lane, which half, both values).  A "neighbour" kernel streams matrix instructions
// (v_mfma_f32_32x32x16_f16, v_mfma_f32_32x32x2_f32 or none: a VALU-only loop) on a second stream.  Both are sized so that their waves share SIMDs.
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fno-slp-vectorize tools/pk_mfma_hazard.hip -o /tmp/pk_mfma_hazard && /tmp/pk_mfma_hazard
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Rec { uint32_t wg, lane, it, half, got, want, form, pad; };

// FORM 0: v_pk_fma_f32 op_sel_hi:[1,0,1] (hi half takes the LOW word of src1: the broadcast form the compiler makes of  a0 += g*f.x; a1 += g*f.y)
// FORM 1: v_pk_fma_f32 (plain)      FORM 2: v_pk_mul_f32 + v_pk_add_f32       FORM 3: control, two scalar fmas against two scalar fmas
template <int FORM>
__global__ __launch_bounds__(256) void victim_kernel(Rec* __restrict__ rec, uint32_t* __restrict__ count, int iters, int max_rec) {
    const uint32_t tid = threadIdx.x, wg = blockIdx.x;
    f2 acc; acc.x = 1.0f + 0.001f * (float)(tid & 63); acc.y = 2.0f + 0.002f * (float)(tid & 63);
    float lo = acc.x, hi = acc.y;
    f2 a; a.x = 0.99990f; a.y = 0.99985f;
    f2 b; b.x = 1.00003f + 1e-6f * (float)wg; b.y = 0.99997f;
    f2 c; c.x = 1e-3f; c.y = -1e-3f;
    uint32_t bad = 0;
    for (int it = 0; it < iters; ++it) {
        if (FORM == 0) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(a.x), "v"(b.x));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(a.y), "v"(b.x));
        } else if (FORM == 1) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(a.x), "v"(b.x));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(a.y), "v"(b.y));
        } else if (FORM == 2) {
            asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(acc) : "v"(a), "v"(c));
            asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(lo) : "v"(a.x), "v"(c.x));
            asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(hi) : "v"(a.y), "v"(c.y));
        } else {
            float x0 = acc.x, x1 = acc.y;
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x0) : "v"(a.x), "v"(b.x));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x1) : "v"(a.y), "v"(b.x));
            acc.x = x0; acc.y = x1;
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(a.x), "v"(b.x));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(a.y), "v"(b.x));
        }
        const bool b0 = __float_as_uint(acc.x) != __float_as_uint(lo), b1 = __float_as_uint(acc.y) != __float_as_uint(hi);
        if (b0 || b1) {
            ++bad;
            const uint32_t slot = atomicAdd(count, 1u);
            if ((int)slot < max_rec) {
                Rec r; r.wg = wg; r.lane = tid; r.it = (uint32_t)it; r.half = (b0 ? 1u : 0u) | (b1 ? 2u : 0u);
                r.got = __float_as_uint(b1 ? acc.y : acc.x); r.want = __float_as_uint(b1 ? hi : lo); r.form = FORM; r.pad = 0;
                rec[slot] = r;
            }
            acc.x = lo; acc.y = hi;
        }
        if ((it & 255) == 255) { acc.x = lo = 1.0f + 0.001f * (float)(tid & 63); acc.y = hi = 2.0f + 0.002f * (float)(tid & 63); }
    }
    if (bad == 0xffffffffu) rec[0].pad = bad;
}

// FORM 4: the form that reproduces beside the real GEMM (tools/lds_guard.hip pk_guard_kernel): weights from LDS (ds_read_b128), sixteen
// v_pk_fma_f32 back to back with the two crossed op_sel forms, scalar shadow, compared every 64 rows.
template <bool CROSSED>
__global__ __launch_bounds__(256) void victim_dense_kernel(Rec* __restrict__ rec, uint32_t* __restrict__ count, int iters, int max_rec, int check_mask = 63,
                                                           uint32_t* __restrict__ alt_count = nullptr) {
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
            if (CROSSED) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[4 * q]) : "v"(f), "v"(g01));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[4 * q + 1]) : "v"(f), "v"(g01));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[4 * q + 2]) : "v"(f), "v"(g23));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[4 * q + 3]) : "v"(f), "v"(g23));
            } else {          // the same sums with the weight duplicated into a register pair: no operand select modifiers
                f2 gx, gy, gz, gw; gx.x = gx.y = g.x; gy.x = gy.y = g.y; gz.x = gz.y = g.z; gw.x = gw.y = g.w;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[4 * q]) : "v"(f), "v"(gx));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[4 * q + 1]) : "v"(f), "v"(gy));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[4 * q + 2]) : "v"(f), "v"(gz));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[4 * q + 3]) : "v"(f), "v"(gw));
            }
            lo[4 * q] = fmaf(f.x, g.x, lo[4 * q]);         hi[4 * q] = fmaf(f.y, g.x, hi[4 * q]);
            lo[4 * q + 1] = fmaf(f.x, g.y, lo[4 * q + 1]); hi[4 * q + 1] = fmaf(f.y, g.y, hi[4 * q + 1]);
            lo[4 * q + 2] = fmaf(f.x, g.z, lo[4 * q + 2]); hi[4 * q + 2] = fmaf(f.y, g.z, hi[4 * q + 2]);
            lo[4 * q + 3] = fmaf(f.x, g.w, lo[4 * q + 3]); hi[4 * q + 3] = fmaf(f.y, g.w, hi[4 * q + 3]);
        }
        if ((it & check_mask) == check_mask) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const bool b0 = __float_as_uint(acc[j].x) != __float_as_uint(lo[j]), b1 = __float_as_uint(acc[j].y) != __float_as_uint(hi[j]);
                if (b0 || b1) {
                    if (alt_count && check_mask == 0) {
                        // one row per check: what would the DEFAULT operand selects have given?  (low half: f.x * the pair's low word; high half: f.y * its high word)
                        const float glo = gs[(it * 4 + (tid >> 6)) & 127][(j & ~1)], ghi = gs[(it * 4 + (tid >> 6)) & 127][(j | 1)];
                        const float alt = b1 ? fmaf(f.y, ghi, 0.0f) : fmaf(f.x, glo, 0.0f);
                        if (__float_as_uint(alt) == __float_as_uint(b1 ? acc[j].y : acc[j].x)) atomicAdd(alt_count, 1u);
                    }
                    const uint32_t slot = atomicAdd(count, 1u);
                    if ((int)slot < max_rec) {
                        Rec r; r.wg = wg; r.lane = tid; r.it = (uint32_t)it; r.half = (b0 ? 1u : 0u) | (b1 ? 2u : 0u);
                        r.got = __float_as_uint(b1 ? acc[j].y : acc[j].x); r.want = __float_as_uint(b1 ? hi[j] : lo[j]); r.form = 4; r.pad = (uint32_t)j;
                        rec[slot] = r;
                    }
                }
                acc[j].x = lo[j] = 0.0f; acc[j].y = hi[j] = 0.0f;
            }
        }
    }
}

// The same question for v_fma_mix_f32 (the engines' in-register binary16 split uses it with op_sel:[1,0,0] op_sel_hi:[1,0,0]: the HIGH half of a packed
// register as the binary16 multiplicand), which shares the VOP3P encoding and its operand-select bits with the packed-fp32 instructions.
__global__ __launch_bounds__(256) void victim_mix_kernel(Rec* __restrict__ rec, uint32_t* __restrict__ count, int iters, int max_rec) {
    const uint32_t tid = threadIdx.x, wg = blockIdx.x;
    uint32_t pk[8];
    float acc[16], ref[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 lo = (_Float16)(0.5f + 0.01f * (float)((tid + j) & 31)), hi = (_Float16)(-0.25f - 0.02f * (float)((tid * 3 + j) & 31));
        pk[j] = (uint32_t)__builtin_bit_cast(unsigned short, lo) | ((uint32_t)__builtin_bit_cast(unsigned short, hi) << 16);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = ref[j] = 0.0f;
    float c = 1.0f + 1e-3f * (float)(wg & 7);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[2 * j]) : "v"(pk[j]), "v"(c));                       // low half
            asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * j + 1]) : "v"(pk[j]), "v"(c));    // high half
            float lo, hi; uint32_t sh;
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(lo) : "v"(pk[j]));
            asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(sh) : "v"(pk[j]));
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(hi) : "v"(sh));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref[2 * j]) : "v"(lo), "v"(c));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref[2 * j + 1]) : "v"(hi), "v"(c));
        }
        if ((it & 63) == 63) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (__float_as_uint(acc[j]) != __float_as_uint(ref[j])) {
                    const uint32_t slot = atomicAdd(count, 1u);
                    if ((int)slot < max_rec) {
                        Rec r; r.wg = wg; r.lane = tid; r.it = (uint32_t)it; r.half = (j & 1) ? 2u : 1u; r.got = __float_as_uint(acc[j]); r.want = __float_as_uint(ref[j]);
                        r.form = 5; r.pad = (uint32_t)j;
                        rec[slot] = r;
                    }
                }
                acc[j] = ref[j] = 0.0f;
            }
        }
    }
}

// ... and for the pair the engines actually use for the split residuals: v_fma_mixlo_f16 d, h, -1.0, t op_sel_hi:[1,0,0] (low half of h) and
// v_fma_mixhi_f16 d, h, -1.0, t op_sel:[1,0,0] op_sel_hi:[1,0,0] (HIGH half of h: a non-default select), 2252 of each in the library.
__global__ __launch_bounds__(256) void victim_mixhalf_kernel(Rec* __restrict__ rec, uint32_t* __restrict__ count, int iters, int max_rec) {
    const uint32_t tid = threadIdx.x, wg = blockIdx.x;
    float t0[8], t1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { t0[j] = 0.37f + 0.013f * (float)((tid + 5 * j) & 63) + 1e-4f * (float)(wg & 15); t1[j] = -1.91f + 0.021f * (float)((tid * 3 + j) & 63); }
    for (int it = 0; it < iters; ++it) {
        uint32_t pk[8], d[8], ref[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[j]) : "v"(t0[j]), "v"(t1[j]));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            d[j] = 0;
            asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(d[j]) : "v"(pk[j]), "v"(t0[j]));
            asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d[j]) : "v"(pk[j]), "v"(t1[j]));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {          // the same residuals through scalar instructions
            float h0, h1; uint32_t sh, r0, r1;
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(h0) : "v"(pk[j]));
            asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(sh) : "v"(pk[j]));
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(h1) : "v"(sh));
            float e0 = t0[j] - h0, e1 = t1[j] - h1;
            asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(r0) : "v"(e0));
            asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(r1) : "v"(e1));
            ref[j] = (r0 & 0xffffu) | (r1 << 16);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (d[j] != ref[j]) {
                const uint32_t slot = atomicAdd(count, 1u);
                if ((int)slot < max_rec) {
                    Rec r; r.wg = wg; r.lane = tid; r.it = (uint32_t)it; r.half = ((d[j] ^ ref[j]) & 0xffffu ? 1u : 0u) | ((d[j] ^ ref[j]) >> 16 ? 2u : 0u); r.got = d[j]; r.want = ref[j];
                    r.form = 6; r.pad = (uint32_t)j;
                    rec[slot] = r;
                }
            }
            t0[j] += 1e-3f; t1[j] -= 7e-4f;
        }
        if ((it & 1023) == 1023) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { t0[j] = 0.37f + 0.013f * (float)((tid + 5 * j) & 63); t1[j] = -1.91f + 0.021f * (float)((tid * 3 + j) & 63); }
        }
    }
}

// KIND 3: v_pk_fma_f32 (plain)  KIND 4: v_pk_mul_f32 / v_pk_add_f32  KIND 5: LDS traffic (ds_write_b64 / ds_read_b128)  KIND 6: global loads
// KIND 7: matrix + packed + LDS together   KIND 8: v_pk_fma_f32 with crossed op_sel (as the victim)
// KIND 9: matrix (f16) + scalar VALU   KIND 10: matrix (fp32 32x32x2) + packed + LDS   KIND 11: matrix (f16 16x16x32) + scalar VALU
template <int KIND>
__global__ __launch_bounds__(256) void neighbour2_kernel(float* __restrict__ sink, const float* __restrict__ src, int iters) {
    __shared__ __attribute__((aligned(16))) float buf[4096];
    f2 x; x.x = 1.0f + 1e-3f * (float)threadIdx.x; x.y = 0.5f;
    f2 m; m.x = 0.9999f; m.y = 1.0001f;
    f2 c; c.x = 1e-4f; c.y = -1e-4f;
    f16v c0 = {};
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (float)((threadIdx.x + i) & 7)); b[i] = (_Float16)(0.02f * (float)((threadIdx.x * 3 + i) & 7)); }
    float s = 0.0f;
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = (float)i;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 3 || KIND == 7) {
            for (int u = 0; u < 8; ++u) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
        }
        if (KIND == 8) {
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(x) : "v"(m), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0]" : "+v"(x) : "v"(m), "v"(c));
            }
        }
        if (KIND == 4) {
            for (int u = 0; u < 4; ++u) asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(x) : "v"(m), "v"(c));
        }
        if (KIND == 5 || KIND == 7) {
            const float4 v = *reinterpret_cast<const float4*>(&buf[((threadIdx.x + it) & 1023) * 4]);
            s += v.x + v.w;
            *reinterpret_cast<f2*>(&buf[((threadIdx.x * 2 + it) & 2047) * 2]) = x;
        }
        if (KIND == 6) {
            s += __builtin_nontemporal_load(src + (((size_t)blockIdx.x * 256 + threadIdx.x + (size_t)it * 65536) & ((1u << 24) - 1)));
        }
        if (KIND == 3 + 100 || KIND == 10) {
            for (int u = 0; u < 8; ++u) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
            const float4 v = *reinterpret_cast<const float4*>(&buf[((threadIdx.x + it) & 1023) * 4]);
            s += v.x + v.w;
            *reinterpret_cast<f2*>(&buf[((threadIdx.x * 2 + it) & 2047) * 2]) = x;
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, x.y, c0, 0, 0, 0);
        }
        if (KIND == 9 || KIND == 11) {
            float v = x.x;
            for (int u = 0; u < 8; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(m.x), "v"(c.x));
            x.x = v;
        }
        if (KIND == 11) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            f4v d; d[0] = c0[0]; d[1] = c0[1]; d[2] = c0[2]; d[3] = c0[3];
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
            c0[0] = d[0]; c0[1] = d[1]; c0[2] = d[2]; c0[3] = d[3];
        }
        if (KIND == 7 || KIND == 9) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    }
    s += x.x + x.y;
    for (int i = 0; i < 16; ++i) s += c0[i];
    if (s == 123.456f) sink[0] = s;
}

// KIND 0: v_mfma_f32_32x32x16_f16 (the fp16x3 engines' instruction)   KIND 1: v_mfma_f32_32x32x2_f32 (the exact-fp32 engine's)   KIND 2: VALU only
template <int KIND>
__global__ __launch_bounds__(256) void neighbour_kernel(float* __restrict__ sink, int iters) {
    f16v c0 = {}, c1 = {};
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (float)((threadIdx.x + i) & 7)); b[i] = (_Float16)(0.02f * (float)((threadIdx.x * 3 + i) & 7)); }
    float fa = 0.5f + 1e-3f * (float)threadIdx.x, fb = 0.25f;
    float v = fa;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
        } else if (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, c1, 0, 0, 0);
        } else {
            for (int u = 0; u < 16; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(fb), "v"(fa));
        }
    }
    float s = v;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    if (s == 123.456f) sink[0] = s;
}

template <int FORM>
static void launch_victim(Rec* rec, uint32_t* count, int wgs, int iters, int max_rec, hipStream_t s) {
    hipLaunchKernelGGL(victim_kernel<FORM>, dim3(wgs), dim3(256), 0, s, rec, count, iters, max_rec);
}

int main(int argc, char** argv) {
    const int max_rec = 1 << 16;
    const int v_iters = argc > 1 ? atoi(argv[1]) : 200000, n_iters = argc > 2 ? atoi(argv[2]) : 400000, wgs = argc > 3 ? atoi(argv[3]) : 512;
    Rec* rec; uint32_t* count; float* sink;
    CHECK(hipMalloc(&rec, sizeof(Rec) * max_rec)); CHECK(hipMalloc(&count, 4)); CHECK(hipMalloc(&sink, 4));
    hipStream_t s0, s1; CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    static Rec host[1 << 16];
    const char* forms[] = {"v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_fma_f32", "v_pk_mul_f32 + v_pk_add_f32", "scalar control"};
    const char* kinds[] = {"v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x2_f32", "VALU only", "nothing"};
    float* src; CHECK(hipMalloc(&src, sizeof(float) << 24)); CHECK(hipMemset(src, 0, sizeof(float) << 24));
    const char* kinds2[] = {"v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x2_f32", "VALU only", "v_pk_fma_f32 plain", "v_pk_mul_f32 + v_pk_add_f32", "LDS traffic", "global loads", "matrix + packed + LDS", "v_pk_fma_f32 crossed op_sel", "matrix f16 + scalar VALU", "matrix fp32 + packed + LDS", "matrix f16 16x16x32 + scalar", "nothing"};
    for (int crossed = 1; crossed >= 0; --crossed)
    for (int kind = 0; kind < 13; ++kind) {
        if (!crossed && kind != 7 && kind != 9) continue;
        CHECK(hipMemset(count, 0, 4));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int ni = n_iters / 4;
        if (kind == 0) hipLaunchKernelGGL(neighbour_kernel<0>, dim3(wgs), dim3(256), 0, s1, sink, n_iters);
        if (kind == 1) hipLaunchKernelGGL(neighbour_kernel<1>, dim3(wgs), dim3(256), 0, s1, sink, n_iters / 2);
        if (kind == 2) hipLaunchKernelGGL(neighbour_kernel<2>, dim3(wgs), dim3(256), 0, s1, sink, n_iters / 2);
        if (kind == 3) hipLaunchKernelGGL(neighbour2_kernel<3>, dim3(wgs), dim3(256), 0, s1, sink, src, ni * 4);
        if (kind == 4) hipLaunchKernelGGL(neighbour2_kernel<4>, dim3(wgs), dim3(256), 0, s1, sink, src, ni * 4);
        if (kind == 5) hipLaunchKernelGGL(neighbour2_kernel<5>, dim3(wgs), dim3(256), 0, s1, sink, src, ni * 4);
        if (kind == 6) hipLaunchKernelGGL(neighbour2_kernel<6>, dim3(wgs), dim3(256), 0, s1, sink, src, ni);
        if (kind == 7) hipLaunchKernelGGL(neighbour2_kernel<7>, dim3(wgs), dim3(256), 0, s1, sink, src, ni * 2);
        if (kind == 8) hipLaunchKernelGGL(neighbour2_kernel<8>, dim3(wgs), dim3(256), 0, s1, sink, src, ni * 4);
        if (kind == 9) hipLaunchKernelGGL(neighbour2_kernel<9>, dim3(wgs), dim3(256), 0, s1, sink, src, ni * 2);
        if (kind == 10) hipLaunchKernelGGL(neighbour2_kernel<10>, dim3(wgs), dim3(256), 0, s1, sink, src, ni * 2);
        if (kind == 11) hipLaunchKernelGGL(neighbour2_kernel<11>, dim3(wgs), dim3(256), 0, s1, sink, src, ni * 2);
        hipEventRecord(e0, s0);
        if (crossed) hipLaunchKernelGGL(victim_dense_kernel<true>, dim3(wgs), dim3(256), 0, s0, rec, count, v_iters / 8, max_rec);
        else hipLaunchKernelGGL(victim_dense_kernel<false>, dim3(wgs), dim3(256), 0, s0, rec, count, v_iters / 8, max_rec);
        hipEventRecord(e1, s0);
        CHECK(hipDeviceSynchronize());
        float tv = 0; hipEventElapsedTime(&tv, e0, e1);
        uint32_t n = 0; CHECK(hipMemcpy(&n, count, 4, hipMemcpyDeviceToHost));
        printf("neighbour %-30s victim dense v_pk_fma_f32, %-32s : %8u mismatches   (victim %.1f ms)\n", kinds2[kind], crossed ? "crossed operand selects" : "plain (weights duplicated)", n, tv);
        if (n) {
            const uint32_t m = n < (uint32_t)max_rec ? n : (uint32_t)max_rec;
            CHECK(hipMemcpy(host, rec, sizeof(Rec) * m, hipMemcpyDeviceToHost));
            unsigned q[4] = {0, 0, 0, 0}, h[4] = {0, 0, 0, 0}, par[2] = {0, 0};
            for (uint32_t i = 0; i < m; ++i) { q[(host[i].lane & 63) >> 4]++; h[host[i].half & 3]++; par[host[i].pad & 1]++; }
            printf("    lanes 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u    low only / high only / both: %u %u %u    accumulator even / odd: %u %u\n",
                   q[0], q[1], q[2], q[3], h[1], h[2], h[3], par[0], par[1]);
        }
    }
    for (int kind = 0; kind < 3; ++kind) {          // v_fma_mix_f32 with a high-half select beside the neighbours that break the packed-fp32 forms
        CHECK(hipMemset(count, 0, 4));
        if (kind == 0) hipLaunchKernelGGL(neighbour2_kernel<9>, dim3(wgs), dim3(256), 0, s1, sink, src, n_iters / 2);
        if (kind == 1) hipLaunchKernelGGL(neighbour2_kernel<7>, dim3(wgs), dim3(256), 0, s1, sink, src, n_iters / 2);
        if (kind == 2) hipLaunchKernelGGL(neighbour2_kernel<11>, dim3(wgs), dim3(256), 0, s1, sink, src, n_iters / 2);
        hipLaunchKernelGGL(victim_mix_kernel, dim3(wgs), dim3(256), 0, s0, rec, count, v_iters / 4, max_rec);
        CHECK(hipDeviceSynchronize());
        uint32_t n = 0; CHECK(hipMemcpy(&n, count, 4, hipMemcpyDeviceToHost));
        printf("neighbour %-30s victim v_fma_mix_f32 with low- and high-half selects        : %8u mismatches\n", kind == 0 ? "matrix f16 + scalar VALU" : kind == 1 ? "matrix + packed + LDS" : "matrix f16 16x16x32 + scalar", n);
    }
    for (int kind = 0; kind < 3; ++kind) {          // the split residuals' v_fma_mixlo_f16 / v_fma_mixhi_f16 pair
        CHECK(hipMemset(count, 0, 4));
        if (kind == 0) hipLaunchKernelGGL(neighbour2_kernel<9>, dim3(wgs), dim3(256), 0, s1, sink, src, n_iters / 2);
        if (kind == 1) hipLaunchKernelGGL(neighbour2_kernel<7>, dim3(wgs), dim3(256), 0, s1, sink, src, n_iters / 2);
        if (kind == 2) hipLaunchKernelGGL(neighbour2_kernel<11>, dim3(wgs), dim3(256), 0, s1, sink, src, n_iters / 2);
        hipLaunchKernelGGL(victim_mixhalf_kernel, dim3(wgs), dim3(256), 0, s0, rec, count, v_iters / 4, max_rec);
        CHECK(hipDeviceSynchronize());
        uint32_t n = 0; CHECK(hipMemcpy(&n, count, 4, hipMemcpyDeviceToHost));
        printf("neighbour %-30s victim v_fma_mixlo_f16 + v_fma_mixhi_f16 (high-half select)         : %8u mismatches\n", kind == 0 ? "matrix f16 + scalar VALU" : kind == 1 ? "matrix + packed + LDS" : "matrix f16 16x16x32 + scalar", n);
    }
    {   // what does a wrong result hold?  Checked after every row (one product per accumulator), beside the f16 matrix + scalar neighbour
        uint32_t* alt; CHECK(hipMalloc(&alt, 4)); CHECK(hipMemset(alt, 0, 4)); CHECK(hipMemset(count, 0, 4));
        hipLaunchKernelGGL(neighbour2_kernel<9>, dim3(wgs), dim3(256), 0, s1, sink, src, n_iters / 2);
        hipLaunchKernelGGL(victim_dense_kernel<true>, dim3(wgs), dim3(256), 0, s0, rec, count, v_iters / 16, max_rec, 0, alt);
        CHECK(hipDeviceSynchronize());
        uint32_t n = 0, na = 0; CHECK(hipMemcpy(&n, count, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&na, alt, 4, hipMemcpyDeviceToHost));
        printf("checked after every row: %u wrong results, %u of them are exactly the product the DEFAULT operand selects would give (the rest: neither word of the pair -- the operand of the last pass is not the victim's)\n", n, na);
    }
    for (int kind = 0; kind < (argc > 4 ? 4 : 0); ++kind)
        for (int form = 0; form < 4; ++form) {
            CHECK(hipMemset(count, 0, 4));
            hipEvent_t e0, e1, f0, f1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f0); hipEventCreate(&f1);
            hipEventRecord(f0, s1);
            if (kind == 0) hipLaunchKernelGGL(neighbour_kernel<0>, dim3(wgs), dim3(256), 0, s1, sink, n_iters);
            if (kind == 1) hipLaunchKernelGGL(neighbour_kernel<1>, dim3(wgs), dim3(256), 0, s1, sink, n_iters / 2);
            if (kind == 2) hipLaunchKernelGGL(neighbour_kernel<2>, dim3(wgs), dim3(256), 0, s1, sink, n_iters / 2);
            hipEventRecord(f1, s1);
            hipEventRecord(e0, s0);
            if (form == 0) launch_victim<0>(rec, count, wgs, v_iters, max_rec, s0);
            if (form == 1) launch_victim<1>(rec, count, wgs, v_iters, max_rec, s0);
            if (form == 2) launch_victim<2>(rec, count, wgs, v_iters, max_rec, s0);
            if (form == 3) launch_victim<3>(rec, count, wgs, v_iters, max_rec, s0);
            hipEventRecord(e1, s0);
            CHECK(hipDeviceSynchronize());
            float tv = 0, tn = 0; hipEventElapsedTime(&tv, e0, e1); hipEventElapsedTime(&tn, f0, f1);
            uint32_t n = 0; CHECK(hipMemcpy(&n, count, 4, hipMemcpyDeviceToHost));
            printf("neighbour %-24s victim %-32s : %8u mismatches   (victim %.1f ms, neighbour %.1f ms)\n", kinds[kind], forms[form], n, tv, tn);
            if (n) {
                const uint32_t m = n < (uint32_t)max_rec ? n : (uint32_t)max_rec;
                CHECK(hipMemcpy(host, rec, sizeof(Rec) * m, hipMemcpyDeviceToHost));
                unsigned q[4] = {0, 0, 0, 0}, h[4] = {0, 0, 0, 0}, wave[4] = {0, 0, 0, 0};
                for (uint32_t i = 0; i < m; ++i) { q[(host[i].lane & 63) >> 4]++; h[host[i].half & 3]++; wave[host[i].lane >> 6]++; }
                printf("    lanes 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u    low only / high only / both: %u %u %u    by wave of the workgroup: %u %u %u %u\n",
                       q[0], q[1], q[2], q[3], h[1], h[2], h[3], wave[0], wave[1], wave[2], wave[3]);
                for (uint32_t i = 0; i < (m < 6 ? m : 6); ++i)
                    printf("    wg %u lane %u it %u half %u got %08x (%.9g) want %08x (%.9g)\n", host[i].wg, host[i].lane, host[i].it, host[i].half,
                           host[i].got, *(float*)&host[i].got, host[i].want, *(float*)&host[i].want);
            }
        }
    return 0;
}
