// HBM-bound row/column kernels around the GEMM engine:
//   softmax over attention keys (models/attn.py:80), InstanceNorm1d+ReLU (models/attn.py:24-25),
//   channel L2-normalisation (models/gmmreg.py:74), Cout=1 convolutions (models/dgcnn.py:27,34),
//   and the row/column softmax-weighted sums of the overlap block (models/gmmreg.py:79-80).
// One wavefront per row with 16-byte lane loads where rows are contiguous; column reductions put 64
// consecutive channels on 64 consecutive lanes so every global access is a full 256-byte line.
#include "ogmm_common.h"
#include <algorithm>

namespace {

using namespace ogmm;

// ---------------------------------------------------------------- softmax over the last axis, in place
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, int64_t rows, int cols, int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* __restrict__ p = x + row * ld;
    float v[16];
    float m = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + i * 64;
        v[i] = c < cols ? p[c] : -__builtin_inff();
        m = fmaxf(m, v[i]);
    }
    m = wave_max(m);
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + i * 64;
        v[i] = c < cols ? expf(v[i] - m) : 0.0f;
        s += v[i];
    }
    s = wave_sum(s);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + i * 64;
        if (c < cols) p[c] = v[i] / s;
    }
}

// ---------------------------------------------------------------- InstanceNorm1d(affine=False) + ReLU, in place
// block = (cloud, 64-channel slab): 64 channels x 4 row lanes; statistics accumulate in fp64 like
// PyTorch's CPU batch-norm statistics (acc_type<float> = double); biased variance; the slab
// (N x 256 B) stays L2-resident for the second and third sweep.
__global__ __launch_bounds__(256) void instnorm_relu_kernel(float* __restrict__ x, int64_t ld, int N, int D, float eps) {
    __shared__ double red[4][64];
    __shared__ float s_alpha[64], s_beta[64];
    const int c = blockIdx.y, ch = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + ch;
    const bool ok = col < D;
    float* __restrict__ base = x + (int64_t)c * N * ld + col;
    double acc = 0.0;
    if (ok) for (int n = rl; n < N; n += 4) acc += base[(int64_t)n * ld];
    red[rl][ch] = acc;
    __syncthreads();
    const double mean = (red[0][ch] + red[1][ch] + red[2][ch] + red[3][ch]) / N;
    __syncthreads();
    acc = 0.0;
    if (ok) for (int n = rl; n < N; n += 4) { const double d = base[(int64_t)n * ld] - mean; acc += d * d; }
    red[rl][ch] = acc;
    __syncthreads();
    if (rl == 0) {
        const double var = (red[0][ch] + red[1][ch] + red[2][ch] + red[3][ch]) / N;
        const double inv = 1.0 / sqrt(var + (double)eps);
        s_alpha[ch] = (float)inv;
        s_beta[ch] = (float)(-mean * inv);
    }
    __syncthreads();
    const float a = s_alpha[ch], b = s_beta[ch];
    if (ok) for (int n = rl; n < N; n += 4) {
        float* q = base + (int64_t)n * ld;
        *q = fmaxf(fmaf(*q, a, b), 0.0f);
    }
}

// ---------------------------------------------------------------- fused-InstanceNorm statistics -> affine (see ogmm_gemm.col_stats)
// clear: the entry is zeroed behind the read -- every entry has exactly one reader, so the accumulating GEMM of the NEXT forward finds the buffer as it
// needs it and the caller's per-forward fill of it (a launch, and for the whole-forward buffer a 6 MB pass) goes away (round 5).
__global__ __launch_bounds__(256) void instnorm_finalize_kernel(double* __restrict__ st, int64_t n, int rows, float eps,
                                                                float* __restrict__ scale, float* __restrict__ shift, int clear) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double2 s12 = *reinterpret_cast<const double2*>(st + 2 * i);
    const double mean = s12.x / rows;
    const double var = fmax(s12.y / rows - mean * mean, 0.0);
    const double inv = 1.0 / sqrt(var + (double)eps);
    scale[i] = (float)inv;
    shift[i] = (float)(-mean * inv);
    if (clear) *reinterpret_cast<double2*>(st + 2 * i) = double2{0.0, 0.0};
}

// ---------------------------------------------------------------- activation rows -> split fragment-major B image
// x [rows][ld] fp32 -> hi/lo images [rows/32][K/16][64 lanes][8 halfs] (OGMM_PREC_F16X3_FRAG operand order), so that an
// ACTIVATION can be the B operand of the fp16x3 engine (the N x N similarity of models/gmmreg.py:75).  rows % 32 == 0 is
// not required: missing rows are zero.  K % 16 == 0.
using f16x8p = __attribute__((ext_vector_type(8))) _Float16;
__global__ __launch_bounds__(256) void pack_frag_kernel(const float* __restrict__ x, int64_t ld, int64_t rows, int K, int64_t rows_pad,
                                                        f16x8p* __restrict__ hi, f16x8p* __restrict__ lo) {
    const int64_t total = rows_pad / 32 * (K / 16) * 64;
    for (int64_t gI = (int64_t)blockIdx.x * 256 + threadIdx.x; gI < total; gI += (int64_t)gridDim.x * 256) {
        const int lane = (int)(gI & 63);
        const int64_t blk = gI >> 6;
        const int kb = (int)(blk % (K / 16));
        const int64_t nb = blk / (K / 16);
        const int64_t row = nb * 32 + (lane & 31);
        const int k0 = kb * 16 + (lane >> 5) * 8;
        f16x8p h = {0, 0, 0, 0, 0, 0, 0, 0}, l = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < rows) {
            const float4 a = *reinterpret_cast<const float4*>(x + row * ld + k0);
            const float4 b = *reinterpret_cast<const float4*>(x + row * ld + k0 + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xx = __builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
                const _Float16 hh = (_Float16)xx;
                h[e] = hh;
                l[e] = (_Float16)(xx - (float)hh);
            }
        }
        hi[gI] = h;
        lo[gI] = l;
    }
}

// ---------------------------------------------------------------- F.normalize(dim=channels) + split fragment-major B image in one kernel
// The tgt half of the normalised map is only ever the B operand of the similarity GEMM: a workgroup takes 32 rows, computes their norms
// (one wave per row, as l2norm_rows_kernel: same sums, same division) and then writes the rows' slice of the hi/lo images directly
// (pack_frag_kernel's layout; the second read of the rows comes from L2).  Saves writing and re-reading the fp32 map (2 x 134 MB at B = 64).
// Second job of the same launch (round 5; x_rn != nullptr): workgroups beyond the image's take 32 rows each of ANOTHER map and write 1 / max(|row|, 1e-12)
// (row_rnorm_kernel's sums and division) -- the src half's row scale of the similarity GEMM, which used to be a launch of its own behind this one.
__global__ __launch_bounds__(256) void l2norm_pack_frag_kernel(const float* __restrict__ x, int64_t ld, int64_t rows, int K, f16x8p* __restrict__ hi,
                                                               f16x8p* __restrict__ lo, int64_t pack_blocks, const float* __restrict__ x_rn, int64_t ld_rn,
                                                               int64_t rows_rn, float* __restrict__ rnorm_out) {
    __shared__ float nrm_s[32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int64_t)blockIdx.x >= pack_blocks) {
        const int64_t r0 = ((int64_t)blockIdx.x - pack_blocks) * 32;
        for (int r = wave; r < 32; r += 4) {
            const int64_t row = r0 + r;
            if (row >= rows_rn) break;
            const float* __restrict__ p = x_rn + row * ld_rn;
            float ss = 0.0f;
            for (int d = lane * 4; d < K; d += 256) {
                const float4 v = *reinterpret_cast<const float4*>(p + d);
                ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
            const float nrm = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
            if (lane == 0) rnorm_out[row] = 1.0f / nrm;
        }
        return;
    }
    const int64_t nb = blockIdx.x, row0 = nb * 32;
    for (int r = wave; r < 32; r += 4) {
        const int64_t row = row0 + r;
        float ss = 0.0f;
        if (row < rows) {
            const float* __restrict__ p = x + row * ld;
            for (int d = lane * 4; d < K; d += 256) {
                const float4 v = *reinterpret_cast<const float4*>(p + d);
                ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
        }
        const float nrm = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
        if (lane == 0) nrm_s[r] = nrm;
    }
    __syncthreads();
    const int r = lane & 31;
    const int64_t row = row0 + r;
    const float nrm = nrm_s[r];
    const int KB = K / 16;
    for (int kb = wave; kb < KB; kb += 4) {
        const int k0 = kb * 16 + (lane >> 5) * 8;
        f16x8p h = {0, 0, 0, 0, 0, 0, 0, 0}, l = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < rows) {
            const float4 a = *reinterpret_cast<const float4*>(x + row * ld + k0);
            const float4 b = *reinterpret_cast<const float4*>(x + row * ld + k0 + 4);
            const float v[8] = {a.x / nrm, a.y / nrm, a.z / nrm, a.w / nrm, b.x / nrm, b.y / nrm, b.z / nrm, b.w / nrm};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xx = __builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
                const _Float16 hh = (_Float16)xx;
                h[e] = hh;
                l[e] = (_Float16)(xx - (float)hh);
            }
        }
        const int64_t gI = (nb * KB + kb) * 64 + lane;
        hi[gI] = h;
        lo[gI] = l;
    }
}

// ---------------------------------------------------------------- F.normalize(dim=channels), one wave per row
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int D,
                                                          float* __restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* __restrict__ p = x + row * ldx;
    float ss = 0.0f;
    for (int d = lane * 4; d < D; d += 256) {
        const float4 v = *reinterpret_cast<const float4*>(p + d);
        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    const float nrm = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
    float* __restrict__ q = out + row * ldo;
    for (int d = lane * 4; d < D; d += 256) {
        float4 v = *reinterpret_cast<const float4*>(p + d);
        v.x /= nrm; v.y /= nrm; v.z /= nrm; v.w /= nrm;
        *reinterpret_cast<float4*>(q + d) = v;
    }
}

// ---------------------------------------------------------------- 1 / max(|row|, eps): F.normalize's divisor as a row scale (same sums as above), one wave per row
__global__ __launch_bounds__(256) void row_rnorm_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int D, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* __restrict__ p = x + row * ldx;
    float ss = 0.0f;
    for (int d = lane * 4; d < D; d += 256) {
        const float4 v = *reinterpret_cast<const float4*>(p + d);
        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    const float nrm = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
    if (lane == 0) out[row] = 1.0f / nrm;
}

// ---------------------------------------------------------------- Cout = 1 convolution, one wave per row
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int D,
                                                     const float* __restrict__ w, const float* __restrict__ b, int act,
                                                     float* __restrict__ y, int64_t ldy) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* __restrict__ p = x + row * ldx;
    float acc = 0.0f;
    for (int d = lane * 4; d < D; d += 256) {
        const float4 v = *reinterpret_cast<const float4*>(p + d);
        const float4 ww = *reinterpret_cast<const float4*>(w + d);
        acc = fmaf(v.w, ww.w, fmaf(v.z, ww.z, fmaf(v.y, ww.y, fmaf(v.x, ww.x, acc))));
    }
    acc = wave_sum(acc) + (b ? b[0] : 0.0f);
    if (act == OGMM_ACT_SIGMOID) acc = 1.0f / (1.0f + expf(-acc));
    else if (act == OGMM_ACT_RELU) acc = fmaxf(acc, 0.0f);
    else if (act == OGMM_ACT_LEAKY02) acc = acc > 0.0f ? acc : 0.2f * acc;
    if (lane == 0) y[row * ldy] = acc;
}

// ---------------------------------------------------------------- overlap block, row direction (one wave per row m)
//   wo_src[b][m] = sum_n softmax_n(S[b][m][:])[n] * o_src[b][n]
__global__ __launch_bounds__(256) void overlap_rows_kernel(const float* __restrict__ S, int N, const float* __restrict__ o_src,
                                                           int64_t ldo_in, float* __restrict__ wo_src, int64_t ldo,
                                                           float* __restrict__ stats /* [B][4][N] or NULL: rows 0,1 = row max, row sum */) {
    const int lane = threadIdx.x & 63, b = blockIdx.y;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= N) return;
    const float* __restrict__ row = S + ((int64_t)b * N + m) * N;
    const float* __restrict__ o = o_src + (int64_t)b * N * ldo_in;
    // the row is read once: up to 64 * OVR values per lane stay in registers between the max and the exp-sum pass
    constexpr int OVR = 16;
    float v[OVR];
    float mx = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < OVR; ++i) {
        const int n = lane + 64 * i;
        v[i] = n < N ? row[n] : -__builtin_inff();
        mx = fmaxf(mx, v[i]);
    }
    for (int n = lane + 64 * OVR; n < N; n += 64) mx = fmaxf(mx, row[n]);
    mx = wave_max(mx);
    float se = 0.0f, so = 0.0f;
#pragma unroll
    for (int i = 0; i < OVR; ++i) {
        const int n = lane + 64 * i;
        if (n < N) {
            const float e = expf(v[i] - mx);
            se += e;
            so = fmaf(e, o[(int64_t)n * ldo_in], so);
        }
    }
    for (int n = lane + 64 * OVR; n < N; n += 64) {
        const float e = expf(row[n] - mx);
        se += e;
        so = fmaf(e, o[(int64_t)n * ldo_in], so);
    }
    se = wave_sum(se);
    so = wave_sum(so);
    if (lane == 0) {
        wo_src[((int64_t)b * N + m) * ldo] = so / se;
        if (stats) { stats[((int64_t)b * 4 + 0) * N + m] = mx; stats[((int64_t)b * 4 + 1) * N + m] = se; }
    }
}

// ---------------------------------------------------------------- overlap block, column direction
//   wo_tgt[b][n] = sum_m softmax_m(S[b][:][n])[m] * o_tgt[b][m]; block = 64 columns x 4 row lanes, online softmax
__global__ __launch_bounds__(256) void overlap_cols_kernel(const float* __restrict__ S, int N, const float* __restrict__ o_tgt,
                                                           int64_t ldo_in, float* __restrict__ wo_tgt, int64_t ldo,
                                                           float* __restrict__ stats /* rows 2,3 = column max, column sum */) {
    __shared__ float sm[4][64], ss[4][64], st[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + cl;
    const float* __restrict__ Sb = S + (int64_t)b * N * N;
    const float* __restrict__ o = o_tgt + (int64_t)b * N * ldo_in;
    float mx = -__builtin_inff(), se = 0.0f, so = 0.0f;
    if (n < N) {
        // rows in chunks of 8 per row lane: eight independent loads in flight, one rescale per chunk instead of a data-dependent
        // branch per element
        constexpr int CH = 8;
        for (int m0 = rl * CH; m0 < N; m0 += 4 * CH) {
            float v[CH], om[CH];
            float cm = -__builtin_inff();
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int m = m0 + i;
                v[i] = m < N ? Sb[(int64_t)m * N + n] : -__builtin_inff();
                om[i] = m < N ? o[(int64_t)m * ldo_in] : 0.0f;
                cm = fmaxf(cm, v[i]);
            }
            const float nm = fmaxf(mx, cm);
            const float r = expf(mx - nm);            // exp(-inf) = 0 for the first chunk
            se *= r;
            so *= r;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const float e = expf(v[i] - nm);      // exp(-inf) = 0 for rows beyond N
                se += e;
                so = fmaf(e, om[i], so);
            }
            mx = nm;
        }
    }
    sm[rl][cl] = mx; ss[rl][cl] = se; st[rl][cl] = so;
    __syncthreads();
    if (rl == 0 && n < N) {
        float M = fmaxf(fmaxf(sm[0][cl], sm[1][cl]), fmaxf(sm[2][cl], sm[3][cl]));
        float E = 0.0f, T = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float r = expf(sm[i][cl] - M);
            E = fmaf(ss[i][cl], r, E);
            T = fmaf(st[i][cl], r, T);
        }
        wo_tgt[((int64_t)b * N + n) * ldo] = T / E;
        if (stats) { stats[((int64_t)b * 4 + 2) * N + n] = M; stats[((int64_t)b * 4 + 3) * N + n] = E; }
    }
}

// ---------------------------------------------------------------- overlap block, backward (training)
// With P1 = softmax_n(S[m][:]) and P2 = softmax_m(S[:][n]) rebuilt from the saved row / column max and sum:
//   dS[m][n]   = P1 a[m] (o_src[n] - wo_src[m]) + P2 b[n] (o_tgt[m] - wo_tgt[n]),   a = dL/dwo_src, b = dL/dwo_tgt
//   g_osrc[n] += sum_m a[m] P1[m][n]      (o_src is indexed along the tgt axis, models/gmmreg.py:79)
//   g_otgt[m]  = sum_n b[n] P2[m][n]
// One pass over S; a block owns ROWS_B rows (one wave per row at a time), column sums go through LDS and one global atomic
// per column and block.
constexpr int OVB_ROWS = 32;
__global__ __launch_bounds__(256) void overlap_bwd_kernel(const float* __restrict__ S, int N, const float* __restrict__ o_src,
                                                          const float* __restrict__ o_tgt, int64_t ldo_in, const float* __restrict__ wo_src,
                                                          const float* __restrict__ wo_tgt, int64_t ldo, const float* __restrict__ stats,
                                                          const float* __restrict__ g_wo_src, const float* __restrict__ g_wo_tgt, int64_t ldg,
                                                          float* __restrict__ dS, float* __restrict__ g_osrc, float* __restrict__ g_otgt, int64_t ldgo) {
    extern __shared__ float colacc[];                     // [N]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    const float* __restrict__ st = stats + (int64_t)b * 4 * N;
    for (int n = threadIdx.x; n < N; n += 256) colacc[n] = 0.0f;
    __syncthreads();
    const int m_hi = min(N, (int)(blockIdx.x + 1) * OVB_ROWS);
    for (int m = blockIdx.x * OVB_ROWS + wave; m < m_hi; m += 4) {
        const int64_t gm = (int64_t)b * N + m;
        const float a = g_wo_src[gm * ldg], wos = wo_src[gm * ldo], rmax = st[m], rinv = 1.0f / st[N + m];
        const float otm = o_tgt[gm * ldo_in];
        const float* __restrict__ row = S + gm * N;
        float* __restrict__ drow = dS + gm * N;
        float dot = 0.0f;
        for (int n = lane; n < N; n += 64) {
            const int64_t gn = (int64_t)b * N + n;
            const float s = row[n];
            const float p1 = expf(s - rmax) * rinv;
            const float p2 = expf(s - st[2 * N + n]) / st[3 * N + n];
            const float bn = g_wo_tgt[gn * ldg];
            drow[n] = p1 * a * (o_src[gn * ldo_in] - wos) + p2 * bn * (otm - wo_tgt[gn * ldo]);
            dot = fmaf(bn, p2, dot);
            atomicAdd(&colacc[n], a * p1);
        }
        dot = wave_sum(dot);
        if (lane == 0) g_otgt[gm * ldgo] = dot;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256) atomicAdd(&g_osrc[((int64_t)b * N + n) * ldgo], colacc[n]);
}

}  // namespace

extern "C" int ogmm_softmax_rows(float* x, int64_t rows, int cols, int64_t ld, void* stream) {
    OGMM_REQUIRE(x && rows > 0 && cols > 0 && cols <= 1024 && ld >= cols, "ogmm_softmax_rows: need 0 < cols <= 1024 <= ld (cols=%d)", cols);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ogmm::as_stream(stream), x, rows, cols, ld);
    return ogmm::check_launch("ogmm_softmax_rows");
}

extern "C" int ogmm_instnorm_relu(float* x, int64_t ld, int C, int N, int D, float eps, void* stream) {
    OGMM_REQUIRE(x && C > 0 && N > 0 && D > 0 && ld >= D, "ogmm_instnorm_relu: bad sizes C=%d N=%d D=%d", C, N, D);
    hipLaunchKernelGGL(instnorm_relu_kernel, dim3((D + 63) / 64, C), dim3(256), 0, ogmm::as_stream(stream), x, ld, N, D, eps);
    return ogmm::check_launch("ogmm_instnorm_relu");
}

extern "C" int ogmm_instnorm_finalize(double* col_stats, int64_t n_entries, int rows, float eps, float* scale, float* shift, int clear, void* stream) {
    OGMM_REQUIRE(col_stats && scale && shift && n_entries > 0 && rows > 0 && ogmm::aligned16(col_stats), "ogmm_instnorm_finalize: null / unaligned pointer or empty input");
    hipLaunchKernelGGL(instnorm_finalize_kernel, dim3((unsigned)((n_entries + 255) / 256)), dim3(256), 0, ogmm::as_stream(stream), col_stats,
                       n_entries, rows, eps, scale, shift, clear);
    return ogmm::check_launch("ogmm_instnorm_finalize");
}

extern "C" int ogmm_pack_frag(const float* x, int64_t ld, int64_t rows, int K, void* hi, void* lo, void* stream) {
    OGMM_REQUIRE(x && hi && lo && rows > 0 && K > 0 && K % 16 == 0 && ld % 4 == 0 && ogmm::aligned16(x) && ogmm::aligned16(hi) && ogmm::aligned16(lo),
                 "ogmm_pack_frag: K %% 16 == 0, ld %% 4 == 0 and 16-byte aligned pointers required");
    const int64_t rows_pad = (rows + 31) / 32 * 32;
    const int64_t total = rows_pad / 32 * (K / 16) * 64;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65535);
    hipLaunchKernelGGL(pack_frag_kernel, dim3(blocks), dim3(256), 0, ogmm::as_stream(stream), x, ld, rows, K, rows_pad,
                       reinterpret_cast<f16x8p*>(hi), reinterpret_cast<f16x8p*>(lo));
    return ogmm::check_launch("ogmm_pack_frag");
}

extern "C" int ogmm_l2norm_pack_frag(const float* x, int64_t ld, int64_t rows, int K, void* hi, void* lo, void* stream) {
    OGMM_REQUIRE(x && hi && lo && rows > 0 && K > 0 && K % 16 == 0 && ld % 4 == 0 && ogmm::aligned16(x) && ogmm::aligned16(hi) && ogmm::aligned16(lo),
                 "ogmm_l2norm_pack_frag: K %% 16 == 0, ld %% 4 == 0 and 16-byte aligned pointers required");
    const int64_t blocks = (rows + 31) / 32;
    OGMM_REQUIRE(blocks <= 2147483647LL, "ogmm_l2norm_pack_frag: too many rows");
    hipLaunchKernelGGL(l2norm_pack_frag_kernel, dim3((unsigned)blocks), dim3(256), 0, ogmm::as_stream(stream), x, ld, rows, K,
                       reinterpret_cast<f16x8p*>(hi), reinterpret_cast<f16x8p*>(lo), blocks, (const float*)nullptr, (int64_t)0, (int64_t)0, (float*)nullptr);
    return ogmm::check_launch("ogmm_l2norm_pack_frag");
}

extern "C" int ogmm_l2norm_pack_frag_rnorm(const float* x, int64_t ld, int64_t rows, int K, void* hi, void* lo, const float* x_rn, int64_t ld_rn, int64_t rows_rn,
                                           float* rnorm_out, void* stream) {
    OGMM_REQUIRE(x && hi && lo && x_rn && rnorm_out && rows > 0 && rows_rn > 0 && K > 0 && K % 16 == 0 && ld % 4 == 0 && ld_rn % 4 == 0 && ogmm::aligned16(x) &&
                 ogmm::aligned16(x_rn) && ogmm::aligned16(hi) && ogmm::aligned16(lo),
                 "ogmm_l2norm_pack_frag_rnorm: K %% 16 == 0, ld %% 4 == 0 and 16-byte aligned pointers required");
    const int64_t blocks = (rows + 31) / 32, blocks_rn = (rows_rn + 31) / 32;
    OGMM_REQUIRE(blocks + blocks_rn <= 2147483647LL, "ogmm_l2norm_pack_frag_rnorm: too many rows");
    hipLaunchKernelGGL(l2norm_pack_frag_kernel, dim3((unsigned)(blocks + blocks_rn)), dim3(256), 0, ogmm::as_stream(stream), x, ld, rows, K,
                       reinterpret_cast<f16x8p*>(hi), reinterpret_cast<f16x8p*>(lo), blocks, x_rn, ld_rn, rows_rn, rnorm_out);
    return ogmm::check_launch("ogmm_l2norm_pack_frag_rnorm");
}

extern "C" int ogmm_l2norm_rows(const float* x, int64_t ldx, int64_t rows, int D, float* out, int64_t ldo, void* stream) {
    OGMM_REQUIRE(x && out && rows > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ogmm::aligned16(x) && ogmm::aligned16(out),
                 "ogmm_l2norm_rows: D, ldx, ldo must be multiples of 4 and pointers 16-byte aligned");
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ogmm::as_stream(stream), x, ldx, rows, D, out, ldo);
    return ogmm::check_launch("ogmm_l2norm_rows");
}

extern "C" int ogmm_row_rnorm(const float* x, int64_t ldx, int64_t rows, int D, float* out, void* stream) {
    OGMM_REQUIRE(x && out && rows > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ogmm::aligned16(x), "ogmm_row_rnorm: D, ldx must be multiples of 4 and x 16-byte aligned");
    hipLaunchKernelGGL(row_rnorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ogmm::as_stream(stream), x, ldx, rows, D, out);
    return ogmm::check_launch("ogmm_row_rnorm");
}

extern "C" int ogmm_rowdot(const float* x, int64_t ldx, int64_t rows, int D, const float* w, const float* b, int act, float* y,
                           int64_t ldy, void* stream) {
    OGMM_REQUIRE(x && w && y && rows > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ogmm::aligned16(x) && ogmm::aligned16(w),
                 "ogmm_rowdot: D, ldx must be multiples of 4 and pointers 16-byte aligned");
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ogmm::as_stream(stream), x, ldx, rows, D, w, b, act, y, ldy);
    return ogmm::check_launch("ogmm_rowdot");
}

namespace {
using namespace ogmm;
// ---------------------------------------------------------------- overlap block, both directions in ONE pass over S
// A workgroup owns a 64-row x 1024-column tile of one pair's similarity matrix; wave w takes rows 16w .. 16w+15, a lane 16 columns
// (lane + 64 i) of the panel.  Four rows at a time are loaded (64 independent coalesced loads per lane) and used twice: for the row
// softmax-dot (wave reductions) and for a running (max, exp-sum, weighted exp-sum) per column, rescaled once per 4-row chunk.  Row
// results of a panel and column results of a row block are partial softmaxes (max, sum, dot); ogmm_overlap_finalize merges them.
// The two-kernel form read S twice (once of it column-wise) for 165 us at B = 64; S is 268 MB, i.e. ~55 us at the HBM rate.
constexpr int OVT_ROWS = 64, OVT_COLS = 1024;

__global__ __launch_bounds__(256) void overlap_tile_kernel(const float* __restrict__ S, int N, const float* __restrict__ o_src,
                                                           const float* __restrict__ o_tgt, int64_t ldo_in, float* __restrict__ rowpart,
                                                           float* __restrict__ colpart) {
    __shared__ float cm_s[3][3][OVT_COLS];               // waves 1..3: (max, sum, dot) per column
    const int rb = blockIdx.x, p = blockIdx.y, b = blockIdx.z;
    const int n_rb = gridDim.x, n_p = gridDim.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* __restrict__ Sb = S + (int64_t)b * N * N;
    const float* __restrict__ os = o_src + (int64_t)b * N * ldo_in;       // weights of the row direction, indexed by column
    const float* __restrict__ ot = o_tgt + (int64_t)b * N * ldo_in;       // weights of the column direction, indexed by row
    // exp(x), x <= 0, on v_exp_f32: the terms that carry weight have x near 0, where 2^(x log2 e) is as accurate as the range-reduced expf
    auto ex = [](float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); };
    const int c0 = p * OVT_COLS + lane;
    float osv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) osv[i] = c0 + 64 * i < N ? os[(int64_t)(c0 + 64 * i) * ldo_in] : 0.0f;
    float cmx[16], cse[16], cso[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { cmx[i] = -__builtin_inff(); cse[i] = 0.0f; cso[i] = 0.0f; }
    const int m_base = rb * OVT_ROWS + wave * 16;
    for (int m0 = m_base; m0 < m_base + 16 && m0 < N; m0 += 4) {
        float v[4][16];
        float otv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + r;
            otv[r] = m < N ? ot[(int64_t)m * ldo_in] : 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) v[r][i] = (m < N && c0 + 64 * i < N) ? Sb[(int64_t)m * N + c0 + 64 * i] : -__builtin_inff();
        }
        // column direction: one rescale per chunk
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float cm = fmaxf(fmaxf(v[0][i], v[1][i]), fmaxf(v[2][i], v[3][i]));
            const float nm = fmaxf(cmx[i], cm);
            const float sc = ex(cmx[i] - nm);           // first chunk: exp(-inf) = 0
            float se = cse[i] * sc, so = cso[i] * sc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = ex(v[r][i] - nm);       // rows past N: exp(-inf) = 0
                se += e;
                so = fmaf(e, otv[r], so);
            }
            cmx[i] = nm; cse[i] = se; cso[i] = so;
        }
        // row direction
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + r;
            float mx = -__builtin_inff();
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, v[r][i]);
            mx = wave_max(mx);
            float se = 0.0f, so = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = ex(v[r][i] - mx);
                se += e;
                so = fmaf(e, osv[i], so);
            }
            se = wave_sum(se);
            so = wave_sum(so);
            if (lane == 0 && m < N) {
                float* rp = rowpart + (((int64_t)b * n_p + p) * N + m) * 3;
                rp[0] = mx; rp[1] = se; rp[2] = so;
            }
        }
    }
    // merge the four waves' column partials (waves whose rows are all past N hold (-inf, 0, 0))
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            cm_s[wave - 1][0][lane + 64 * i] = cmx[i];
            cm_s[wave - 1][1][lane + 64 * i] = cse[i];
            cm_s[wave - 1][2][lane + 64 * i] = cso[i];
        }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int cl = lane + 64 * i, col = p * OVT_COLS + cl;
            float M = cmx[i];
#pragma unroll
            for (int w = 0; w < 3; ++w) M = fmaxf(M, cm_s[w][0][cl]);
            float r0 = expf(cmx[i] - M);
            float E = cse[i] * r0, T = cso[i] * r0;
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const float r = expf(cm_s[w][0][cl] - M);
                E = fmaf(cm_s[w][1][cl], r, E);
                T = fmaf(cm_s[w][2][cl], r, T);
            }
            if (col < N) {
                float* cp = colpart + (((int64_t)b * n_rb + rb) * N + col) * 3;
                cp[0] = M; cp[1] = E; cp[2] = T;
            }
        }
    }
}

// merges the partial softmaxes: rows over the column panels, columns over the row blocks
__global__ __launch_bounds__(256) void overlap_finalize_kernel(const float* __restrict__ rowpart, const float* __restrict__ colpart, int N, int n_p,
                                                               int n_rb, float* __restrict__ wo_src, float* __restrict__ wo_tgt, int64_t ldo,
                                                               float* __restrict__ stats) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 2 * N) return;
    const bool cols = idx >= N;
    const int x = cols ? idx - N : idx;
    const float* part = cols ? colpart + (int64_t)b * n_rb * N * 3 : rowpart + (int64_t)b * n_p * N * 3;
    const int n = cols ? n_rb : n_p;
    float M = -__builtin_inff();
    for (int t = 0; t < n; ++t) M = fmaxf(M, part[((int64_t)t * N + x) * 3]);
    float E = 0.0f, T = 0.0f;
    for (int t = 0; t < n; ++t) {
        const float* q = part + ((int64_t)t * N + x) * 3;
        const float r = expf(q[0] - M);
        E = fmaf(q[1], r, E);
        T = fmaf(q[2], r, T);
    }
    (cols ? wo_tgt : wo_src)[((int64_t)b * N + x) * ldo] = T / E;
    if (stats) {
        stats[((int64_t)b * 4 + (cols ? 2 : 0)) * N + x] = M;
        stats[((int64_t)b * 4 + (cols ? 3 : 1)) * N + x] = E;
    }
}
}  // namespace

extern "C" int64_t ogmm_overlap_cross_workspace_bytes(int B, int N) {
    const int64_t n_p = (N + OVT_COLS - 1) / OVT_COLS, n_rb = (N + OVT_ROWS - 1) / OVT_ROWS;
    return (int64_t)B * (n_p + n_rb) * N * 3 * (int64_t)sizeof(float);
}

// Second half of the FUSED overlap block: the similarity GEMM's epilogue (ogmm_gemm.ovl_rowpart / ovl_colpart, 256 x 256 tiles) left the
// (1, sum, dot) triples; this merges the N / 256 partials per row and per column.
extern "C" int ogmm_overlap_finalize(const float* rowpart, const float* colpart, int B, int N, float* wo_src, float* wo_tgt, int64_t ldo, void* stream) {
    OGMM_REQUIRE(rowpart && colpart && wo_src && wo_tgt && B > 0 && N > 0 && N % 256 == 0 && ldo >= 1, "ogmm_overlap_finalize: null pointer, or N not a multiple of 256");
    hipLaunchKernelGGL(overlap_finalize_kernel, dim3((2 * N + 255) / 256, B), dim3(256), 0, ogmm::as_stream(stream), rowpart, colpart, N, N / 256, N / 256, wo_src,
                       wo_tgt, ldo, (float*)nullptr);
    return ogmm::check_launch("ogmm_overlap_finalize");
}

// One pass over S.  stats may be NULL (eval); workspace: ogmm_overlap_cross_workspace_bytes(B, N) bytes.
extern "C" int ogmm_overlap_cross_ws(const float* S, int B, int N, const float* o_src, const float* o_tgt, int64_t ldo_in, float* wo_src,
                                     float* wo_tgt, int64_t ldo, float* stats, void* workspace, void* stream) {
    OGMM_REQUIRE(S && o_src && o_tgt && wo_src && wo_tgt && workspace && B > 0 && N > 0 && ldo_in >= 1 && ldo >= 1,
                 "ogmm_overlap_cross_ws: null pointer or empty input");
    const int n_p = (N + OVT_COLS - 1) / OVT_COLS, n_rb = (N + OVT_ROWS - 1) / OVT_ROWS;
    float* rowpart = reinterpret_cast<float*>(workspace);
    float* colpart = rowpart + (int64_t)B * n_p * N * 3;
    hipStream_t s = ogmm::as_stream(stream);
    hipLaunchKernelGGL(overlap_tile_kernel, dim3(n_rb, n_p, B), dim3(256), 0, s, S, N, o_src, o_tgt, ldo_in, rowpart, colpart);
    hipLaunchKernelGGL(overlap_finalize_kernel, dim3((2 * N + 255) / 256, B), dim3(256), 0, s, rowpart, colpart, N, n_p, n_rb, wo_src, wo_tgt, ldo, stats);
    return ogmm::check_launch("ogmm_overlap_cross_ws");
}

extern "C" int ogmm_overlap_cross(const float* S, int B, int N, const float* o_src, const float* o_tgt, int64_t ldo_in, float* wo_src,
                                  float* wo_tgt, int64_t ldo, void* stream) {
    OGMM_REQUIRE(S && o_src && o_tgt && wo_src && wo_tgt && B > 0 && N > 0 && ldo_in >= 1 && ldo >= 1, "ogmm_overlap_cross: null pointer or empty input");
    hipStream_t s = ogmm::as_stream(stream);
    hipLaunchKernelGGL(overlap_rows_kernel, dim3((N + 3) / 4, B), dim3(256), 0, s, S, N, o_src, ldo_in, wo_src, ldo, (float*)nullptr);
    hipLaunchKernelGGL(overlap_cols_kernel, dim3((N + 63) / 64, B), dim3(256), 0, s, S, N, o_tgt, ldo_in, wo_tgt, ldo, (float*)nullptr);
    return ogmm::check_launch("ogmm_overlap_cross");
}

extern "C" int ogmm_overlap_cross_train(const float* S, int B, int N, const float* o_src, const float* o_tgt, int64_t ldo_in, float* wo_src,
                                        float* wo_tgt, int64_t ldo, float* stats, void* stream) {
    OGMM_REQUIRE(S && o_src && o_tgt && wo_src && wo_tgt && stats && B > 0 && N > 0 && ldo_in >= 1 && ldo >= 1, "ogmm_overlap_cross_train: null pointer or empty input");
    hipStream_t s = ogmm::as_stream(stream);
    hipLaunchKernelGGL(overlap_rows_kernel, dim3((N + 3) / 4, B), dim3(256), 0, s, S, N, o_src, ldo_in, wo_src, ldo, stats);
    hipLaunchKernelGGL(overlap_cols_kernel, dim3((N + 63) / 64, B), dim3(256), 0, s, S, N, o_tgt, ldo_in, wo_tgt, ldo, stats);
    return ogmm::check_launch("ogmm_overlap_cross_train");
}

extern "C" int ogmm_overlap_cross_bwd(const float* S, int B, int N, const float* o_src, const float* o_tgt, int64_t ldo_in, const float* wo_src,
                                      const float* wo_tgt, int64_t ldo, const float* stats, const float* g_wo_src, const float* g_wo_tgt, int64_t ldg,
                                      float* dS, float* g_o_src, float* g_o_tgt, int64_t ldgo, void* stream) {
    OGMM_REQUIRE(S && o_src && o_tgt && wo_src && wo_tgt && stats && g_wo_src && g_wo_tgt && dS && g_o_src && g_o_tgt && B > 0 && N > 0,
                 "ogmm_overlap_cross_bwd: null pointer or empty input");
    OGMM_REQUIRE(ldgo == 1, "ogmm_overlap_cross_bwd: g_o_src / g_o_tgt must be dense [B][N] (zeroed here)");
    OGMM_REQUIRE(N <= 16384, "ogmm_overlap_cross_bwd: N <= 16384 (column accumulators live in LDS)");
    hipStream_t s = ogmm::as_stream(stream);
    (void)hipMemsetAsync(g_o_src, 0, sizeof(float) * (size_t)B * N, s);
    hipLaunchKernelGGL(overlap_bwd_kernel, dim3((N + OVB_ROWS - 1) / OVB_ROWS, B), dim3(256), (size_t)N * sizeof(float), s, S, N, o_src, o_tgt, ldo_in,
                       wo_src, wo_tgt, ldo, stats, g_wo_src, g_wo_tgt, ldg, dS, g_o_src, g_o_tgt, ldgo);
    return ogmm::check_launch("ogmm_overlap_cross_bwd");
}
