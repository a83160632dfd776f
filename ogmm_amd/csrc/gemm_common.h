// Shared pieces of the GEMM engines (exact-fp32 and fp16x3-split): activation, tile->workgroup map, fused epilogue.
#pragma once
#include <type_traits>
#include "ogmm_common.h"

namespace ogmm_gemm_detail {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

// fp32 x4 -> binary16 hi x4 + lo x4 with hi = rn16(x), lo = rn16(x - hi), in 8 VALU instructions:
//   2 x v_cvt_pk_f16_f32 (hi pairs), 4 x v_fma_mix{lo,hi}_f16 (lo = rn16(fma(hi, -1, x)): the f16 source is read in place
//   and the fp32 result x - hi, which is exact, is rounded straight into the packed destination), 2 x v_max3_f32 |x| for
//   the overflow flag (|x| > 65504 makes hi infinite and the product garbage: the caller raises the device flag).
// hipcc's own lowering of the same arithmetic is 13 instructions + 12 for clamp/overflow tests; the staging of A is the
// largest non-MFMA cost of the loop (measured: 22 %), almost all of it VALU issue.
__device__ __forceinline__ void split4_f16(const f32x4 v, f16x4& hi, f16x4& lo, float& amax) {
    f16x2 h01, h23, l01, l23;
    asm volatile(
        "v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_cvt_pk_f16_f32 %1, %6, %7\n\t"
        "s_nop 0\n\t"
        "v_fma_mixlo_f16 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %1, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
        "s_nop 0\n\t"
        "v_fma_mixhi_f16 %2, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %1, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        : "=&v"(h01), "=&v"(h23), "=&v"(l01), "=&v"(l23)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    hi = f16x4{h01[0], h01[1], h23[0], h23[1]};
    lo = f16x4{l01[0], l01[1], l23[0], l23[1]};
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}

// The same split for fragments that are already in registers (LDS-DMA engine): the asm has no side effects, so it is not `volatile` and
// the scheduler may move it into the shadow of the matrix instructions.  The overflow test is one v_dot2_f32_f16 per PAIR of values:
// ovf += hi . hi turns inf / nan iff some |x| > 65504 made hi infinite (56 -> 16 VALU per K step and wave against the max-|x| form).
__device__ __forceinline__ void split4_f16_pure(const f32x4 v, f16x4& hi, f16x4& lo, float& ovf) {
    f16x2 h01, h23, l01, l23;
    asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_cvt_pk_f16_f32 %1, %6, %7\n\t"
        "s_nop 0\n\t"
        "v_fma_mixlo_f16 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %1, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
        "s_nop 0\n\t"
        "v_fma_mixhi_f16 %2, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %1, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        : "=&v"(h01), "=&v"(h23), "=&v"(l01), "=&v"(l23)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    hi = f16x4{h01[0], h01[1], h23[0], h23[1]};
    lo = f16x4{l01[0], l01[1], l23[0], l23[1]};
    ovf = __builtin_amdgcn_fdot2(h01, h01, ovf, false);
    ovf = __builtin_amdgcn_fdot2(h23, h23, ovf, false);
}

// One LDS-DMA instruction: 64 lanes x 16 bytes from `sbase + voff` (a wave-uniform 64-bit base in SGPRs plus a per-lane 32-bit byte offset:
// no address arithmetic on the vector ALU) to LDS bytes [lds_addr + 16 * lane, + 16).  It is inline assembly on purpose: hipcc models the
// builtin (__builtin_amdgcn_global_load_lds) as an LGKM event that may complete out of order, so with one in flight EVERY wait for a
// ds_read becomes `s_waitcnt lgkmcnt(0)` -- also for reads issued just before the wait -- and the fragment reads of the next MFMA group can no
// longer stay in flight behind the current group (measured: +25 % cycles in the K loop).  The compiler does not see this load: the caller
// counts vmcnt itself (asm `s_waitcnt vmcnt(N)`), owns M0 (no other M0 user in the kernel) and leaves nothing in flight at kernel end.
__device__ __forceinline__ void lds_dma16(unsigned voff, const void* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case OGMM_ACT_RELU: return fmaxf(v, 0.0f);
        case OGMM_ACT_LEAKY02: return v > 0.0f ? v : 0.2f * v;
        case OGMM_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

// Fused epilogue for a workgroup tile whose waves hold MT x NT 32x32 accumulators in the MFMA C/D layout
// (col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)):
//   v = act(acc * alpha * scale + shift) (+ Res); optional store; optional max over groups of pool_k rows
// (EdgeConv): per-lane run-combine -> LDS atomicMax on the int pattern (v >= 0 after ReLU) -> one coalesced store.
// `smem` must hold groups * BN ints and be free for reuse (the caller's K loop is finished).
template <int MT, int NT, int WM, int WN, bool POOL>
__device__ __forceinline__ void gemm_epilogue(const ogmm_gemm& g, f32x16 (&acc)[MT][NT], float* smem, int m0, int n0, int m_end,
                                              int zo, int zi, float alpha) {
    constexpr int BN = NT * 32 * WN, T = WM * WN * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    float* __restrict__ Cm = g.C ? g.C + zo * g.sC_o + zi * g.sC_i : nullptr;
    const float* __restrict__ Rm = g.Res ? g.Res + zo * g.sR_o + zi * g.sR_i : nullptr;
    const bool store_c = Cm != nullptr && (!POOL || g.store_c);
    int* pool_s = reinterpret_cast<int*>(smem);
    const int groups = POOL ? (m_end - m0) / g.pool_k : 0;
    if (POOL) {
        __syncthreads();
        for (int i = tid; i < groups * BN; i += T) pool_s[i] = 0;
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int cl = (wn * NT + j) * 32 + lr;      // column inside the tile
        const int col = n0 + cl;
        const bool col_ok = col < g.N;
        float cs = 1.0f, ct = 0.0f;
        if (!g.row_affine && col_ok) {
            if (g.scale) cs = g.scale[col];
            if (g.shift) ct = g.shift[col];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int cur_group = -1;
            float cur_max = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;   // row inside the tile
                const int row = m0 + rl;
                if (row < m_end && col_ok) {
                    float s = cs, sh = ct;
                    if (g.row_affine) {
                        s = g.scale ? g.scale[row] : 1.0f;
                        sh = g.shift ? g.shift[row] : 0.0f;
                    }
                    float v = apply_act(fmaf(acc[i][j][r] * alpha, s, sh), g.act);
                    if (Rm) v += Rm[(int64_t)row * g.ldr + col];
                    if (store_c) Cm[(int64_t)row * g.ldc + col] = v;
                    if (POOL) {
                        const int grp = rl / g.pool_k;
                        if (grp != cur_group) {
                            if (cur_group >= 0) atomicMax(&pool_s[cur_group * BN + cl], __float_as_int(cur_max));
                            cur_group = grp;
                            cur_max = v;
                        } else {
                            cur_max = fmaxf(cur_max, v);
                        }
                    }
                }
            }
            if (POOL && cur_group >= 0) atomicMax(&pool_s[cur_group * BN + cl], __float_as_int(cur_max));
        }
    }
    if (POOL) {
        __syncthreads();
        float* __restrict__ Pm = g.pool_out + zo * 0;   // pooled output is not batched
        const int64_t p0 = (int64_t)(m0 / g.pool_k);
        for (int i = tid; i < groups * BN; i += T) {
            const int p = i / BN, c = i % BN;
            if (n0 + c < g.N) Pm[(p0 + p) * g.ldp + n0 + c] = __int_as_float(pool_s[i]);
        }
    }
}

// Wide epilogue (no pooling): each wave transposes its 32 x (NT*32) accumulator slab through a private LDS patch so
// that a lane owns 4 CONSECUTIVE columns of one row, then applies alpha*scale+shift / activation / residual on float4s
// and stores with global_store_dwordx4 (16 lanes = one 256-byte row segment).  4x fewer store instructions than the
// one-dword-per-lane MFMA layout: the output phase of these GEMMs is store-issue bound (measured: 36 % of a
// 131072x512x512 launch), see HISTORY.md.  Needs N % 4 == 0, ldc % 4 == 0 (ldr % 4 == 0), 16-byte aligned C / Res.
// `smem` must provide waves * 32 * (NT*32 + 4) floats and be free (the K loop has passed its last barrier).
template <int MT, int NT, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue_wide(const ogmm_gemm& g, f32x16 (&acc)[MT][NT], float* smem, int m0, int n0, int m_end,
                                                   float alpha, bool direct_stores = false) {
    constexpr int LDC = NT * 32 + 4;
    constexpr int F4_PER_ROW = NT * 8;                      // float4 per patch row
    constexpr int ITER = 32 * F4_PER_ROW / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    float* patch = smem + wave * (32 * LDC);
    float* __restrict__ Cm = g.C;
    const float* __restrict__ Rm = g.Res;
    const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, zero4 = {0.f, 0.f, 0.f, 0.f};
    // column statistics of this wave's MT*32 rows (InstanceNorm / BatchNorm fusion), accumulated in fp64 from the first value on (round 6): the consumer forms
    // var = E[y^2] - mean^2, and with fp32 partial sums of y^2 that difference loses |mean|^2 / var units of 2^-24 -- a channel whose mean is 30 ... 100 standard
    // deviations (the sharp weight family has them: mean^2 / var up to 9500) got its normalised values 1e-5 off, which the parity tail's ill-conditioned pairs
    // amplify (tools/tail_bisect.py; DESIGN.md section 2 "Round 6").  The squares of fp32 values are exact in fp64.
    double tot1[4] = {0.0, 0.0, 0.0, 0.0}, tot2[4] = {0.0, 0.0, 0.0, 0.0};
    // Fast path: the workgroup's whole tile is inside the matrix and scale / shift are per column.  A lane's four columns are the same
    // in every pass (idx % F4_PER_ROW == lane % F4_PER_ROW), so scale / shift are fetched once, and every load and store below is
    // unconditional.  That matters beyond the instruction count: with loads or stores under per-lane conditions the compiler cannot
    // count the operations in flight, so each pass's wait for its (optional) loads becomes s_waitcnt vmcnt(0) -- which also waits for
    // the PREVIOUS pass's store to be acknowledged.  The general form below therefore issues one store per HBM round trip per wave
    // (the "store burst" of HISTORY.md: 14 % of a 131072x1024x1024 launch with the matrix cores idle).
    const bool tile_inside = m0 + WM * MT * 32 <= m_end && n0 + WN * NT * 32 <= g.N;
    if (tile_inside && !g.row_affine && 64 % F4_PER_ROW == 0 && g.act != OGMM_ACT_SIGMOID) {
        const int c4 = (lane % F4_PER_ROW) * 4, rl0 = lane / F4_PER_ROW;
        constexpr int RSTEP = 64 / F4_PER_ROW;             // patch rows covered per pass
        const int col = n0 + wn * NT * 32 + c4;
        const f32x4 sc = g.scale ? *reinterpret_cast<const f32x4*>(g.scale + col) : one4;
        const f32x4 sh = g.shift ? *reinterpret_cast<const f32x4*>(g.shift + col) : zero4;
        const bool stats = g.col_stats != nullptr;
        // The activation is resolved by a uniform branch around the whole tile (KIND 0 none, 1 ReLU, 2 LeakyReLU(0.2), 3 generic: max(v, lo)
        // then v > 0 ? v : slope * v), and a power-of-two alpha (the split engines' inverse weight scale) is folded into the column scale
        // (exactly: fma(v * 2^k, s, t) == fma(v, s * 2^k, t)): 1-2 VALU operations per element instead of 6 on the 128 accumulator
        // registers of a lane -- the epilogue's arithmetic, not only its stores, runs with the matrix cores idle.
        const float act_lo = g.act == OGMM_ACT_RELU ? 0.0f : -__builtin_inff(), act_slope = g.act == OGMM_ACT_LEAKY02 ? 0.2f : 1.0f;
        const bool alpha_pow2 = (__float_as_uint(alpha) & 0x7FFFFFu) == 0u && alpha > 0.0f;
        f32x4 scf = sc;
        if (alpha_pow2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) scf[e] = sc[e] * alpha;
        }
        auto block = [&](auto has_res, auto kind, int i) {
            constexpr int KIND = decltype(kind)::value;
            const int row0 = m0 + (wm * MT + i) * 32 + rl0;
            f32x4 rr[ITER];
            if constexpr (decltype(has_res)::value) {
#pragma unroll
                for (int q = 0; q < ITER; ++q) rr[q] = *reinterpret_cast<const f32x4*>(Rm + (int64_t)(row0 + q * RSTEP) * g.ldr + col);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + j * 32 + lr] = acc[i][j][r];
#pragma unroll
            for (int q = 0; q < ITER; ++q) {
                f32x4 v = *reinterpret_cast<const f32x4*>(&patch[(rl0 + q * RSTEP) * LDC + c4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (KIND == 0) v[e] = fmaf(v[e], scf[e], sh[e]);
                    else if constexpr (KIND == 1) v[e] = fmaxf(fmaf(v[e], scf[e], sh[e]), 0.0f);
                    else if constexpr (KIND == 2) { const float y = fmaf(v[e], scf[e], sh[e]); v[e] = y > 0.0f ? y : 0.2f * y; }
                    else {
                        const float y = fmaxf(fmaf(v[e] * alpha, sc[e], sh[e]), act_lo);
                        v[e] = y > 0.0f ? y : act_slope * y;
                    }
                }
                if constexpr (decltype(has_res)::value) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rr[q][e];
                }
                *reinterpret_cast<f32x4*>(Cm + (int64_t)(row0 + q * RSTEP) * g.ldc + col) = v;
                if (stats) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const double vd = (double)v[e]; tot1[e] += vd; tot2[e] = fma(vd, vd, tot2[e]); }
                }
            }
        };
        auto all_blocks = [&](auto has_res, auto kind) {
#pragma unroll
            for (int i = 0; i < MT; ++i) block(has_res, kind, i);
        };
        const int kind = !alpha_pow2 ? 3 : (g.act == OGMM_ACT_RELU ? 1 : (g.act == OGMM_ACT_LEAKY02 ? 2 : 0));
        if (direct_stores && !Rm && !stats && kind != 3) {
            // No residual, no statistics: store straight from the accumulator layout (lane = column, a register = a row: 32 lanes write one
            // 128-byte row segment) -- 128 dword stores per wave and no LDS transposition (16 ds_write + 4 ds_read + 4 wide stores per block).
            auto direct = [&](auto kind_c) {
                constexpr int KIND = decltype(kind_c)::value;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int cj = n0 + (wn * NT + j) * 32 + lr;
                    const float s1 = (g.scale ? g.scale[cj] : 1.0f) * alpha, t1 = g.shift ? g.shift[cj] : 0.0f;
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        float* __restrict__ cp = Cm + (int64_t)(m0 + (wm * MT + i) * 32 + 4 * lh) * g.ldc + cj;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float y = fmaf(acc[i][j][r], s1, t1);
                            if constexpr (KIND == 1) y = fmaxf(y, 0.0f);
                            else if constexpr (KIND == 2) y = y > 0.0f ? y : 0.2f * y;
                            cp[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldc] = y;
                        }
                    }
                }
            };
            if (kind == 0) direct(std::integral_constant<int, 0>{});
            else if (kind == 1) direct(std::integral_constant<int, 1>{});
            else direct(std::integral_constant<int, 2>{});
        } else
        if (Rm) {
            if (kind == 0) all_blocks(std::true_type{}, std::integral_constant<int, 0>{});
            else if (kind == 1) all_blocks(std::true_type{}, std::integral_constant<int, 1>{});
            else if (kind == 2) all_blocks(std::true_type{}, std::integral_constant<int, 2>{});
            else all_blocks(std::true_type{}, std::integral_constant<int, 3>{});
        } else {
            if (kind == 0) all_blocks(std::false_type{}, std::integral_constant<int, 0>{});
            else if (kind == 1) all_blocks(std::false_type{}, std::integral_constant<int, 1>{});
            else if (kind == 2) all_blocks(std::false_type{}, std::integral_constant<int, 2>{});
            else all_blocks(std::false_type{}, std::integral_constant<int, 3>{});
        }
    } else {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + j * 32 + lr] = acc[i][j][r];
        // (same wave: LDS operations complete in order, no barrier needed)
#pragma unroll
        for (int q = 0; q < ITER; ++q) {
            const int idx = q * 64 + lane;
            const int rl = idx / F4_PER_ROW, c4 = (idx % F4_PER_ROW) * 4;
            const int row = m0 + (wm * MT + i) * 32 + rl;
            const int col = n0 + wn * NT * 32 + c4;
            f32x4 v = *reinterpret_cast<const f32x4*>(&patch[rl * LDC + c4]);
            if (row < m_end && col < g.N) {
                f32x4 sc = one4, sh = zero4;
                if (g.row_affine) {
                    const float s1 = g.scale ? g.scale[row] : 1.0f, t1 = g.shift ? g.shift[row] : 0.0f;
                    sc = f32x4{s1, s1, s1, s1};
                    sh = f32x4{t1, t1, t1, t1};
                } else {
                    if (g.scale) sc = *reinterpret_cast<const f32x4*>(g.scale + col);
                    if (g.shift) sh = *reinterpret_cast<const f32x4*>(g.shift + col);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = apply_act(fmaf(v[e] * alpha, sc[e], sh[e]), g.act);
                if (Rm) {
                    const f32x4 rr = *reinterpret_cast<const f32x4*>(Rm + (int64_t)row * g.ldr + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rr[e];
                }
                *reinterpret_cast<f32x4*>(Cm + (int64_t)row * g.ldc + col) = v;
                if (g.col_stats) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const double vd = (double)v[e]; tot1[e] += vd; tot2[e] = fma(vd, vd, tot2[e]); }
                }
            }
        }
    }
    }
    if (g.col_stats) {
        // lanes that share (lane % F4_PER_ROW) hold the same 4 columns: fold them, then one fp64 atomic per column and statistic
#pragma unroll
        for (int o = F4_PER_ROW; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) { tot1[e] += __shfl_xor(tot1[e], o, 64); tot2[e] += __shfl_xor(tot2[e], o, 64); }
        const int col = n0 + wn * NT * 32 + (lane % F4_PER_ROW) * 4;
        const int first_row = m0 + wm * MT * 32;
        if (lane < F4_PER_ROW && col < g.N && first_row < m_end) {
            double* st = g.col_stats + (int64_t)((first_row >> 8) & g.col_stats_slot_mask) * g.col_stats_slot_stride + ((int64_t)(first_row / g.group_rows) * g.N + col) * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                atomicAdd(st + 2 * e, tot1[e]);
                atomicAdd(st + 2 * e + 1, tot2[e]);
            }
        }
    }
}

// Epilogue for a wave that owns ONE 32-row block against NB consecutive 32-column blocks (the 8 x 1 wave layout of gemm_f16x3_v8.hip), straight
// from the accumulator layout (lane = column, register = row: the 32 lanes of a half wave write one 128-byte row segment -- a full line --
// per store), no LDS for the values.  Requires the slab to be inside the matrix, per-column scale / shift, no pooling.  Column statistics
// (InstanceNorm fusion): a lane sums its 16 rows, the two half waves are folded with one cross-lane move, and the wave's NB * 32 partial sums go
// to `stat_lds` ([row block = stat_slot, by default the wave][NB * 32][2] DOUBLES -- 32 KiB for eight row blocks of 256 columns; floats in the NBS form --, the caller's K loop is over): the caller adds the eight waves up and issues ONE fp64 atomic per
// column and statistic per tile (per wave it would be 4096 atomics per tile: measured +10 % on the 1024-wide layers that feed a normalisation).
// RES_AHEAD (a caller with ~128 registers to spare: the 512-register engine): all NB * 16 residual values of the slab are requested before the first
// one is used, so that their latency is exposed once per slab instead of once per column block.
// NBS (its own instantiation of the calling kernel: an option inside the shared body cost every launch 10 % in round 3): the normalisation-backward fusion of
// struct ogmm_gemm.nb_* -- the value becomes dz = y * act'(x * nb_scale + nb_shift) with x read like a residual (Res / ldr), dz is stored, and the column
// statistics are {sum dz, sum dz * xhat} instead of {sum y, sum y^2}.
template <int NB, bool RES_AHEAD = false, bool NBS = false>
__device__ __forceinline__ void gemm_epilogue_rowblock(const ogmm_gemm& g, f32x16 (&acc)[NB], int row0, int col0, float alpha, float* stat_lds, int stat_slot = -1) {
    const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5, wave = stat_slot >= 0 ? stat_slot : (int)(threadIdx.x >> 6);
    float* __restrict__ Cm = g.C;
    const float* __restrict__ Rm = g.Res;
    const bool stats = g.col_stats != nullptr;
    if constexpr (NBS) {
        const int64_t gb = (int64_t)(row0 / g.group_rows) * g.N;          // the 32 rows of a block lie in one group (group_rows % 256 == 0)
        float xa[NB][16];
#pragma unroll
        for (int j = 0; j < NB; ++j) {          // every x value of the slab requested before the first one is used
            const float* __restrict__ rp = Rm + (int64_t)(row0 + 4 * lh) * g.ldr + col0 + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) xa[j][r] = rp[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldr];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int col = col0 + j * 32 + lr;
            const float s1 = (g.scale ? g.scale[col] : 1.0f) * alpha, t1 = g.shift ? g.shift[col] : 0.0f;
            const float nsc = g.nb_scale[gb + col], nsh = g.nb_shift[gb + col], nm = g.nb_mean[gb + col], nr = g.nb_rstd[gb + col];
            float* __restrict__ cp = Cm + (int64_t)(row0 + 4 * lh) * g.ldc + col;
            float sum1 = 0.0f, sum2 = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float y = fmaf(acc[j][r], s1, t1);
                const float x = xa[j][r];
                const float z = fmaf(x, nsc, nsh);
                const float dz = g.nb_act == OGMM_ACT_LEAKY02 ? (z > 0.0f ? y : 0.2f * y) : (z > 0.0f ? y : 0.0f);
                cp[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldc] = dz;
                sum1 += dz;
                sum2 = fmaf(dz, (x - nm) * nr, sum2);
            }
            sum1 += __shfl_xor(sum1, 32, 64);
            sum2 += __shfl_xor(sum2, 32, 64);
            if (lh == 0) {
                stat_lds[(wave * NB * 32 + j * 32 + lr) * 2] = sum1;
                stat_lds[(wave * NB * 32 + j * 32 + lr) * 2 + 1] = sum2;
            }
        }
        return;
    }
    auto run = [&](auto kind_c) {
        constexpr int KIND = decltype(kind_c)::value;
        float rra[RES_AHEAD ? NB : 1][16];
        if (RES_AHEAD && Rm) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float* __restrict__ rp = Rm + (int64_t)(row0 + 4 * lh) * g.ldr + col0 + j * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) rra[j][r] = rp[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldr];
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int col = col0 + j * 32 + lr;
            const float s1 = (g.scale ? g.scale[col] : 1.0f) * alpha, t1 = g.shift ? g.shift[col] : 0.0f;
            float* __restrict__ cp = Cm + (int64_t)(row0 + 4 * lh) * g.ldc + col;
            float rr[16];
            if (Rm) {
                if (RES_AHEAD) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) rr[r] = rra[j][r];
                } else {
                    const float* __restrict__ rp = Rm + (int64_t)(row0 + 4 * lh) * g.ldr + col;
#pragma unroll
                    for (int r = 0; r < 16; ++r) rr[r] = rp[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldr];
                }
            }
            double sum1 = 0.0, sum2 = 0.0;          // fp64 from the first value on: see gemm_epilogue_wide
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float y = fmaf(acc[j][r], s1, t1);
                if (KIND == OGMM_ACT_RELU) y = fmaxf(y, 0.0f);
                else if (KIND == OGMM_ACT_LEAKY02) y = y > 0.0f ? y : 0.2f * y;
                else if (KIND == OGMM_ACT_SIGMOID) y = 1.0f / (1.0f + expf(-y));
                if (Rm) y += rr[r];
                cp[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldc] = y;
                if (stats) { const double yd = (double)y; sum1 += yd; sum2 = fma(yd, yd, sum2); }
            }
            if (stats) {
                sum1 += __shfl_xor(sum1, 32, 64);
                sum2 += __shfl_xor(sum2, 32, 64);
                if (lh == 0) {
                    double* sd = reinterpret_cast<double*>(stat_lds);
                    sd[(wave * NB * 32 + j * 32 + lr) * 2] = sum1;
                    sd[(wave * NB * 32 + j * 32 + lr) * 2 + 1] = sum2;
                }
            }
        }
    };
    if (g.act == OGMM_ACT_RELU) run(std::integral_constant<int, OGMM_ACT_RELU>{});
    else if (g.act == OGMM_ACT_LEAKY02) run(std::integral_constant<int, OGMM_ACT_LEAKY02>{});
    else if (g.act == OGMM_ACT_SIGMOID) run(std::integral_constant<int, OGMM_ACT_SIGMOID>{});
    else run(std::integral_constant<int, OGMM_ACT_NONE>{});
}

// sum over the 32 lanes of a half wave (lanes that hold the same accumulator rows), result in every lane: quad swaps and row mirrors on the vector
// ALU's DPP path, one cross-row exchange
__device__ __forceinline__ float half_wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));           // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));           // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));          // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));          // row_mirror
    v += __shfl_xor(v, 16, 64);
    return v;
}

// The row-block epilogue with a Cout = 1 convolution behind it (ogmm_gemm.rd_*): the slab holds whole rows (NB * 32 = N columns), so the head's
// dot product is a sum over the lane's NB values per row plus a half-wave reduction; the map itself is stored only if C is given.
template <int NB>
__device__ __forceinline__ void gemm_epilogue_rowblock_rowdot(const ogmm_gemm& g, f32x16 (&acc)[NB], int row0, float alpha) {
    const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    float* __restrict__ Cm = g.C;
    const float* __restrict__ Rm = g.Res;
    // the per-column constants of all NB blocks first: behind the (optional) stores of a block the compiler could not move the next block's loads up
    float s1[NB], t1[NB], w[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int col = j * 32 + lr;
        s1[j] = (g.scale ? g.scale[col] : 1.0f) * alpha;
        t1[j] = g.shift ? g.shift[col] : 0.0f;
        w[j] = g.rd_w[col];
    }
    const float b = g.rd_b ? g.rd_b[0] : 0.0f;
    float rs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rs[r] = 0.0f;
    auto run = [&](auto kind_c) {
        constexpr int KIND = decltype(kind_c)::value;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int col = j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float y = fmaf(acc[j][r], s1[j], t1[j]);
                if (KIND == OGMM_ACT_RELU) y = fmaxf(y, 0.0f);
                else if (KIND == OGMM_ACT_LEAKY02) y = y > 0.0f ? y : 0.2f * y;
                else if (KIND == OGMM_ACT_SIGMOID) y = 1.0f / (1.0f + expf(-y));
                if (Rm || Cm) {
                    const int64_t row = row0 + 4 * lh + (r & 3) + 8 * (r >> 2);
                    if (Rm) y += Rm[row * g.ldr + col];
                    if (Cm) Cm[row * g.ldc + col] = y;
                }
                rs[r] = fmaf(y, w[j], rs[r]);
            }
        }
    };
    if (g.act == OGMM_ACT_RELU) run(std::integral_constant<int, OGMM_ACT_RELU>{});
    else if (g.act == OGMM_ACT_LEAKY02) run(std::integral_constant<int, OGMM_ACT_LEAKY02>{});
    else if (g.act == OGMM_ACT_SIGMOID) run(std::integral_constant<int, OGMM_ACT_SIGMOID>{});
    else run(std::integral_constant<int, OGMM_ACT_NONE>{});
#pragma unroll
    for (int r = 0; r < 16; ++r) rs[r] = half_wave_sum(rs[r]);
    if (lr == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) g.rd_out[(int64_t)(row0 + 4 * lh + (r & 3) + 8 * (r >> 2)) * g.rd_ld] = apply_act(rs[r] + b, g.rd_act);
    }
}

__device__ __forceinline__ bool wide_epilogue_ok(const ogmm_gemm& g) {
    return g.pool_k == 0 && g.C != nullptr && (g.N & 3) == 0 && (g.ldc & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0) &&
           (g.Res == nullptr || ((g.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(g.Res) & 15) == 0)) &&
           (g.row_affine || ((g.scale == nullptr || (reinterpret_cast<uintptr_t>(g.scale) & 15) == 0) &&
                             (g.shift == nullptr || (reinterpret_cast<uintptr_t>(g.shift) & 15) == 0)));
}

}  // namespace ogmm_gemm_detail
