/*
 * ogmm_hip.h -- C ABI of libogmm_hip.so: the MI355X (gfx950) kernels of the overlap-guided GMM
 * registration hot path of gfmei/ogmm, GMMReg.forward(src, tgt, is_test=False).
 *
 * The reference has no FFI of its own: its only boundary is the Python nn.Module surface
 * (models/gmmreg.py:33,50).  The entry points below are what a binding for that path would call --
 * one per tensor-op cluster of the reference -- and each cites the reference code it replaces
 * (paths relative to the reference root).  The Python host in ogmm_amd/ binds them with ctypes
 * (INTEGRATION.md shows the stub); nothing here depends on PyTorch.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated; float = IEEE binary32; indices int32
 *   - point-major layouts: a cloud is xyz[N][3]; a feature map is feats[rows][ld] with the
 *     channel index contiguous (the reference is channel-major [B,C,N]; the host transposes the
 *     3-channel inputs once, every later tensor is produced point-major)
 *   - "C" clouds = 2B for a batch of B pairs (src clouds first, then tgt clouds)
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued, nothing synchronises
 *   - return value 0 = enqueued; non-zero = rejected (bad argument / launch failure), message in
 *     ogmm_last_error().  No entry point retains a pointer after it returns.
 */
#ifndef OGMM_HIP_H
#define OGMM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OGMM_ABI_VERSION 28

int ogmm_abi_version(void);
/* thread-local, valid until the next failing call on this thread */
const char* ogmm_last_error(void);

/* ---- K1: kNN graph.  lib/utils.py:12-44 (square_distance + knn); callers models/dgcnn.py:135,
 * lib/utils.py:52 <- models/attn.py:69.
 * dist(i,j) = max(1e-12, (-2 * fma(z_i,z_j, fma(y_i,y_j, x_i*x_j)) + |p_i|^2) + |p_j|^2)  -- the exact
 * fp32 rounding sequence of the reference on CPU, so distance rows reproduce bit-for-bit.  The kept SET is
 * exactly the one torch.topk keeps, including when rank k is an exact tie (its CPU kernel's libstdc++
 * heap-select / introselect is re-stated for such rows); the set is written in (distance, index) order
 * (torch's order among equal distances is unspecified and no caller depends on it).  1 <= k <= 32, k <= N. */
int ogmm_knn(const float* xyz /*[C][N][3]*/, int C, int N, int k, int32_t* idx /*[C][N][k]*/, void* stream);
/* (ABI 25) The forward's input stage, models/gmmreg.py:50: src, tgt [B][3][N] -> xyz [2B][N][3] (src clouds, then tgt clouds; torch.cat + transpose + contiguous). */
int ogmm_pack_clouds(const float* src, const float* tgt, int B, int N, float* xyz, void* stream);
/* (ABI 25) The head of the forward in one launch: the k-NN graph of the EdgeConv layers (lib/utils.py:37-44, as ogmm_knn: identical neighbour sets, rank-k
 * ties resolved with torch.topk's selection) and -- when idx5 != NULL -- the 5-NN graph of the positional encoding (lib/utils.py:52 <- models/attn.py:69: its
 * own topk call, own rank-5 ties) + the encoding's hidden maps (models/attn.py:65-73, as ogmm_pos_hidden: bit-identical), all from the one copy of the cloud the
 * workgroup holds in LDS.  workspace: ogmm_knn_pos_head_workspace_bytes(C, N) bytes of scratch (scan A's mark words: scan B visits only marked candidates).
 * ogmm_knn_pos_head_supported(N, k): 1 for 9 <= k <= 32 while cloud + candidate lists + the tie scratch fit 64 KiB of LDS (N <= 2816 at k = 20). */
int ogmm_knn_pos_head_supported(int N, int k);
int64_t ogmm_knn_pos_head_workspace_bytes(int C, int N);
int ogmm_knn_pos_head(const float* xyz, int C, int N, int k, int32_t* idx /*[C][N][k]*/, int32_t* idx5 /*[C][N][5] or NULL*/, const float* w_dis, const float* s_dis,
                      const float* t_dis, const float* w_ang, const float* s_ang, const float* t_ang, float* hid_dis /*[C*N][64]*/, float* hid_ang, void* workspace,
                      void* stream);
/* torch.topk(v, k, dim = -1, largest)[1] of a row-major [rows][n] map (row stride ldv) with the reference CPU kernel's choice among TIED values (the same
 * re-statement of ATen's selection as ogmm_knn's tie resolution): the Welsch term of the training loss (lib/loss.py:92, :95) takes the top_k points of the 0 / 1
 * ground-truth overlap labels, so with more than top_k ones the kept set -- and the loss -- depends on it.  idx [rows][k] in (value, index) order. */
int ogmm_topk_rows(const float* v, int64_t ldv, int rows, int n, int k, int largest, int32_t* idx, void* stream);

/* ---- K5: farthest point sampling.  lib/utils.py:170-198 (farthest_point_sample).
 * start != NULL: is_center=False with the torch.randint draw (:190) as an explicit input [C].
 * start == NULL: is_center=True (:183-188): running-min seeded with distances to the centroid.
 * `n_sets` independent samplings of the same clouds run in one launch (start is [n_sets][C],
 * ids is [n_sets][C][npoint]). */
int ogmm_fps(const float* xyz /*[C][N][3]*/, int C, int N, int npoint, int n_sets,
             const int32_t* start, int32_t* ids, void* stream);

/* ---- K6: row gather.  lib/utils.py:111-127 (index_points) as used by get_anchor_corrs :260-261.
 * out[c][s][:] = feats[(cloud_map ? cloud_map[c] : c) * N + ids[c'][s]][:], c' the same mapped cloud.
 * cloud_map lets the cross-attention read the OTHER cloud's anchors (models/gmmreg.py:71-72). */
int ogmm_gather_rows(const float* feats, int64_t ld, int C, int N, int D, const int32_t* ids /*[C][S]*/, int S,
                     const int32_t* cloud_map, float* out /*[C][S][D]*/, void* stream);

/* ---- the 1x1-convolution engine (K3 layers 2-4, K4, K8, K10, K11, K12, K13 similarity, K9 products):
 * every nn.Conv1d/Conv2d(kernel_size=1) of models/dgcnn.py:19-35,121-125, models/attn.py:21,34-57,91-92
 * and the einsum/matmul contractions of models/attn.py:79,82 and models/gmmreg.py:75.
 *   C[z][m][n] = act( alpha * sum_k A[z][m][k] * B[z][n][k] * scale[.] + shift[.] ) + Res[z][m][n]
 * A may be given as two column pieces [A | A2] (torch.cat on channels: models/attn.py:111,
 * models/gmmreg.py:82-83) against one B of row length K1+K2.  scale/shift (NULL = 1/0) index the
 * column n (folded eval-mode BatchNorm + conv bias) or the row m when row_affine != 0.
 * pool_k > 0 (EdgeConv, models/dgcnn.py:139-148): rows are edges in groups of pool_k per point;
 * besides C (stored only if store_c) pool_out[m / pool_k][n] = max over the group (inputs >= 0
 * after ReLU required: act must be OGMM_ACT_RELU).
 * K1, K2, lda, lda2, ldb multiples of 4; all base pointers 16-byte aligned.
 *
 * precision = OGMM_PREC_F32: exact fp32, v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain; 157 TFLOP/s peak).
 * precision = OGMM_PREC_F16X3: every fp32 operand x is split into two binary16 terms x = hi + lo
 *   (hi = rn16(x), lo = rn16(x - hi): 22 significand bits) and a*b is evaluated as hi*hi + hi*lo + lo*hi on
 *   v_mfma_f32_32x32x16_f16 with fp32 accumulation (3 of 2.5 PFLOP/s MFMAs instead of 16 fp32-rate ones).  The
 *   dropped lo*lo term is 2^-22 relative; measured end to end it is indistinguishable from the exact-fp32 path
 *   (DESIGN.md section 2).  A is split on the fly from fp32; B must be given pre-split as B_hi/B_lo
 *   (binary16 [N][ldb_h], ldb_h a multiple of 8, columns beyond K zero) possibly pre-scaled by a power of two
 *   that the caller folds into alpha.  |A| must stay below 65504: larger values are clamped and *overflow
 *   (device int, optional) is set non-zero. */
/* Engines behind OGMM_PREC_F16X3_FRAG (chosen by shape, same arithmetic and bit-identical results): >= 256 tiles of 256 x 256 with K1, K2
 * multiples of 32 run on the LDS-DMA engine (gemm_f16x3_v8.hip: both operands by global_load_lds, activations split in registers; the
 * InstanceNorm forms included, K1 + K2 <= 4096 with a_scale); other large shapes on the register-staged engine (gemm_f16x3_v4.hip); small
 * ones on 128 x 128 / 256 x 256 tiles (gemm_f16x3_v2.hip). */
enum { OGMM_ACT_NONE = 0, OGMM_ACT_RELU = 1, OGMM_ACT_LEAKY02 = 2, OGMM_ACT_SIGMOID = 3 };
/* bits of the device status word that kernels with on-chip / cross-workgroup protocols raise on a timed-out wait (checked by the host like the fp16 range flag) */
enum { OGMM_STATUS_EDGECONV_PROTOCOL = 2, OGMM_STATUS_EM_EXIT_PROTOCOL = 4 };
enum { OGMM_PREC_F32 = 0, OGMM_PREC_F16X3 = 1, OGMM_PREC_F16X3_FRAG = 2, OGMM_PREC_F16_FRAG = 3 };
/* OGMM_PREC_F16_FRAG (reduced precision, for BASELINE configs[2] which is quoted in bf16): the operands of OGMM_PREC_F16X3_FRAG, but
 * only the leading binary16 term of A and B is multiplied (11-bit mantissa, fp32 accumulate; 1/3 of the matrix instructions); shapes
 * outside the large-shape engine run as OGMM_PREC_F16X3_FRAG.
 * OGMM_PREC_F16X3_FRAG: same arithmetic as F16X3, but B_hi/B_lo are given as the fragment-major image
 *   image[n/32][k/16][lane 0..63][8 halfs], lane = ((k % 16) / 8) * 32 + n % 32, element = k % 8,
 * with n padded to a multiple of 256 and k to a multiple of 64 by zeros (two A pieces: the second piece starts at the
 * k-block K1/16, K1 % 64 == 0); ldb_h = padded K.  Each wave then reads its
 * v_mfma_f32_32x32x16_f16 B operands as coalesced 1 KiB loads without touching LDS (no batching in this mode). */

typedef struct ogmm_gemm {
    const float* A;  int64_t lda;  int32_t K1;
    const float* A2; int64_t lda2; int32_t K2;
    const float* B;  int64_t ldb;
    float* C;        int64_t ldc;
    const float* Res; int64_t ldr;
    int32_t M, N;
    int32_t batch_outer, batch_inner;            /* z = zo * batch_inner + zi; both >= 1 */
    int64_t sA_o, sA_i, sA2_o, sA2_i, sB_o, sB_i, sC_o, sC_i, sR_o, sR_i;   /* element strides */
    const float* scale; const float* shift; int32_t row_affine;
    float alpha;
    int32_t act;
    int32_t pool_k; float* pool_out; int64_t ldp; int32_t store_c;
    int32_t precision; const void* B_hi; const void* B_lo; int64_t ldb_h; int32_t* overflow;
    /* InstanceNorm fusion (models/attn.py:24-25), OGMM_PREC_F16X3_FRAG only, rows grouped per cloud (group_rows = N):
     *   col_stats != NULL: the epilogue also accumulates, per (row group, column), sum and sum of squares of the stored
     *     values into col_stats[group][column][2] (double, caller-zeroed)            -- the producer of the normalised map
     *   a_scale/a_shift != NULL ([group][K1+K2] float): A is read as relu(a * a_scale + a_shift) (relu iff a_relu)
     *     -- the consumer; ogmm_instnorm_finalize turns the statistics into a_scale / a_shift.
     * group_rows must be a multiple of the row tile (256; 128 for small problems). */
    double* col_stats; const float* a_scale; const float* a_shift; int32_t a_relu; int32_t group_rows;
    /* Overlap-block fusion (models/gmmreg.py:75-80), OGMM_PREC_F16X3_FRAG, batch_outer = pairs, M = N = points (multiples of 256):
     *   ovl_rowpart != NULL: the product is the similarity S of L2-normalised rows and is NOT stored (C is ignored); per 256 x 256 tile the epilogue
     *   leaves the partial softmax-dots  rows: sum_j exp(S_ij - 1) {1, o_col[j]}   columns: sum_i exp(S_ij - 1) {1, o_row[i]}   as (1, sum, dot)
     *   triples in ovl_rowpart [batch][N / 256][M][3] and ovl_colpart [batch][M / 256][N][3]; ogmm_overlap_finalize merges them into
     *   softmax(S, dim = 2) @ o_col and softmax(S^T, dim = 2) @ o_row.  o_row / o_col are read at [(batch * M + i) * ovl_ld] / [(batch * N + j) * ovl_ld].
 *   (models/gmmreg.py:79-80 passes src_o as o_col and tgt_o as o_row: its src_wo weights the row softmax with src_o indexed by the COLUMN.)
     *   row_rscale ([batch][M], may be NULL): S_ij is multiplied by row_rscale[i] first -- the A rows may then be left un-normalised
     *   (row_rscale = 1 / max(|row|, eps), ogmm_row_rnorm). */
    const float* ovl_orow; const float* ovl_ocol; int64_t ovl_ld; float* ovl_rowpart; float* ovl_colpart; const float* row_rscale;
    /* A Cout = 1 convolution fused behind this layer (the heads proj.3 and overlap.6 of models/gmmreg.py:30-47), OGMM_PREC_F16X3_FRAG, N == 256
     * (one column tile holds whole rows), M a multiple of 256:  rd_out[row * rd_ld] = rd_act(sum_col y[row][col] * rd_w[col] + rd_b[0]) with y the
     * value this layer would store.  C may then be NULL: the 256-wide map is not written at all.  ogmm_gemm_rowdot_fusable tells. */
    const float* rd_w; const float* rd_b; int32_t rd_act; float* rd_out; int64_t rd_ld;
    /* A rows gathered on the fly (lib/utils.py:111-127 index_points in front of a convolution: the anchors of models/gmmreg.py:54, 67-68),
     * OGMM_PREC_F16X3_FRAG, one A piece: output row m = c * a_gather_S + s reads A row map(c) * a_gather_N + a_gather_ids[map(c)][s] with
     * map(c) = a_gather_map ? a_gather_map[c] : c; a_gather_rows = rows of A (for the 32-bit offset check).  ogmm_gemm_gather_fusable tells. */
    const int32_t* a_gather_ids; const int32_t* a_gather_map; int32_t a_gather_S; int32_t a_gather_N; int64_t a_gather_rows;
    /* col_stats spread over several copies: row tile t adds into copy (t & col_stats_slot_mask), col_stats_slot_stride doubles apart (>= groups * N * 2;
     * the caller zeroes and afterwards sums the mask + 1 copies).  One copy (mask 0) serialises the atomics of every row tile of a group on the same
     * addresses: fine for <= 512 tiles, 12x slower than a separate pass on the 5.2 M-row per-edge maps of the training step (10240 tiles per group). */
    int32_t col_stats_slot_mask; int64_t col_stats_slot_stride;
    /* Per-layer term budget (OGMM_PREC_F16X3_FRAG): how many binary16 matrix instructions a product block may be evaluated with.
     *   0 or 3: a_lo w_hi + a_hi w_lo + a_hi w_hi (fp32-class, the default);
     *   2:      (a_hi + a_lo) w_hi -- the WEIGHT operand B is rounded to binary16 (its lo plane is not read), the activation keeps both terms;
     *   1:      a_hi w_hi -- both operands rounded to binary16 (the 4-wave engine, N >= 512, only).
     * A permission, not an order: engines / shapes without the cheaper form run all three terms (results then differ from the two-term form
     * by the weight's rounding, 2^-12 relative per product).  Which layers of the path tolerate it -- R, t within 1e-5 of the reference over the
     * parity distribution -- is measured, not assumed: tools/term_budget.py (CPU oracle with the same rounding) and HISTORY.md section 4. */
    int32_t terms;
    /* Normalisation-backward fusion (training; round 4): the GEMM computes dh = dY W (the gradient w.r.t. the ACTIVATION a = act(x * nb_scale + nb_shift) of a
     * normalised map x) and its epilogue turns it into dz = dh * act'(x * nb_scale + nb_shift), stores dz and accumulates the normalisation backward's two
     * column sums per group -- col_stats[g][c] = {sum dz, sum dz * xhat}, xhat = (x - nb_mean) * nb_rstd -- so that the separate reduction pass over (x, dh)
     * disappears; ogmm_norm_bwd_apply(x, dy = dz, act = NONE, sums = col_stats) finishes.  x = Res / ldr (same shape as C; no residual is added in this mode),
     * nb_* are [groups][N] float, group = row / group_rows (group_rows %% 256 == 0), nb_act in {RELU, LEAKY02}.  LDS-DMA engine (N >= 512, whole tiles) only. */
    const float* nb_mean; const float* nb_rstd; const float* nb_scale; const float* nb_shift; int32_t nb_act;
    /* A read TRANSPOSED (training; round 4, ABI 24): the logical operand is A[m][k] = A_mem[k * lda + m] -- A_mem is a row-major [K][M] map, e.g. the
     * upstream gradient dY [rows][Cout] of a weight gradient dW = dY^T X (M = Cout, K = rows, B = the fragment image of X^T, batch = row chunks with
     * sA_o = chunk_rows * lda).  The LDS-DMA engine fetches 4 k-rows x 64 m per DMA instruction and reads its operand fragments with transposing
     * ds_read2st64_b32; same products in the same order as the plain form on an explicitly transposed copy: bit-identical results, without the copy.
     * One A piece (K2 = 0), M %% 4 == 0, lda %% 4 == 0, K1 %% 32 == 0, no a_scale / gather / overlap / nb_* forms.  ogmm_gemm_atrans_supported tells.
     * a_colsum (optional, with a_trans): double [M], ADDED to: a_colsum[m] += sum over k (and over the batch) of A[m][k] -- the bias gradient dy.sum(0), gathered
     * from the operand fragments the engine reads anyway (fp32 tree over 8 values, fp64 from there on); the caller zeroes it. */
    int32_t a_trans; double* a_colsum;
} ogmm_gemm;

int ogmm_gemm_nt(const ogmm_gemm* desc /*HOST pointer*/, void* stream);
/* 1 if ogmm_gemm_nt takes the fused overlap block (ovl_rowpart) for B pairs of N points with D channels, else 0 (the caller then runs the
 * similarity GEMM into S and ogmm_overlap_cross_ws). */
int ogmm_gemm_overlap_fusable(int B, int N, int D);
/* 1 if ogmm_gemm_nt takes a fused Cout = 1 head (rd_out) behind an M x N layer with K1 + K2 input channels, else 0 */
int ogmm_gemm_rowdot_fusable(int M, int N, int K1, int K2);
/* 1 if ogmm_gemm_nt takes gathered A rows (a_gather_ids) for an M x N layer with K input channels over `rows` source rows, else 0 */
int ogmm_gemm_gather_fusable(int M, int N, int K, int64_t rows);
int ogmm_gemm_normbwd_fusable(int M, int N, int K, int group_rows);          /* the normalisation-backward fusion (ogmm_gemm.nb_*): 1 / 0 */
int ogmm_gemm_atrans_supported(int M, int N, int K, int64_t lda, int batch);  /* the transposed-A form (ogmm_gemm.a_trans) for `batch` products of M x N over K: 1 / 0 */
/* second half of the fused overlap block: merges the (1, sum, dot) triples the similarity GEMM left (models/gmmreg.py:79-80):
 * wo_src[(b N + i) ldo] = softmax(S_b, dim = 1)[i] . o_tgt, wo_tgt[(b N + j) ldo] = softmax(S_b^T, dim = 1)[j] . o_src */
int ogmm_overlap_finalize(const float* rowpart, const float* colpart, int B, int N, float* wo_src, float* wo_tgt, int64_t ldo, void* stream);
/* out[row] = 1 / max(|x[row, 0:D]|_2, 1e-12): the divisor of F.normalize as a row scale (ogmm_gemm.row_rscale) */
int ogmm_row_rnorm(const float* x, int64_t ldx, int64_t rows, int D, float* out, void* stream);

/* ---- K2+K3 layer 1: neighbour gather + cat(x_j - x_i, x_i) + conv 6->64 + BN + ReLU + max over k.
 * lib/utils.py:56-64, models/dgcnn.py:137-139.  h1 [C*N*k][64] (the un-pooled tensor feeds conv2),
 * pooled written to pool_out[point][0..63] with row stride ldp. */
int ogmm_edgeconv_first(const float* xyz, const int32_t* idx, int C, int N, int k,
                        const float* W /*[64][6]*/, const float* scale, const float* shift,
                        float* h1, float* pool_out, int64_t ldp, void* stream);

/* ---- K2+K3 fused: the whole EdgeConv chain (gather, conv1..conv4 + BN + ReLU, max over k after each layer) for
 * 7 <= k <= 32; xcat[point][0:64 | 64:128 | 128:256 | 256:512] = x1 | x2 | x3 | x4 (models/dgcnn.py:135-150).  Layers 2-4 take
 * their weights as OGMM_PREC_F16X3_FRAG images (h*, l*) with folded BN scale/shift (s*, t*) and the image's inverse
 * power-of-two scale (inv*).  No per-edge tensor is written to memory. */
int ogmm_edgeconv_fused(const float* xyz, const int32_t* idx, int C, int N, int k, const float* W1, const float* s1, const float* t1,
                        const void* h2, const void* l2, const float* s2, const float* t2, float inv2, const void* h3, const void* l3,
                        const float* s3, const float* t3, float inv3, const void* h4, const void* l4, const float* s4, const float* t4,
                        float inv4, float* xcat, int64_t ldx, void* stream);
/* The same chain as a producer / consumer pipeline over 32-row blocks (k = 20 only; edgeconv_pc.hip): four waves compute layers 1-3 of a block
 * each, four waves multiply every block with layer 4, so that every SIMD holds one vector-ALU-heavy and one matrix-heavy wave; no workgroup
 * barrier in the loop.  Same arguments, bit-identical xcat.  `status` (device int32[1] or NULL): the waves of a workgroup hand blocks over through LDS
 * sequence counters with bounded waits; a wait that runs into its limit (a protocol error: never observed, but a hung GPU is not an acceptable failure
 * mode) ORs OGMM_STATUS_EDGECONV_PROTOCOL into *status -- the host checks that word behind the forward, as it does the fp16 range flag -- and the
 * output is NaN-poisoned as well. */
int ogmm_edgeconv_pc(const float* xyz, const int32_t* idx, int C, int N, int k, const float* W1, const float* s1, const float* t1,
                        const void* h2, const void* l2, const float* s2, const float* t2, float inv2, const void* h3, const void* l3,
                        const float* s3, const float* t3, float inv3, const void* h4, const void* l4, const float* s4, const float* t4,
                        float inv4, float* xcat, int64_t ldx, int32_t* status, void* stream);
/* ---- K7 front half: PositionEncoding up to its two 64-channel hidden maps.  models/attn.py:65-73:
 * centroid, g=|p-c|^2 -> conv_dis.0 (1->64)+BN+LeakyReLU -> hid_dis; 5-NN offsets, cosine with the
 * global offset -> conv_ang1 (1->64)+BN+LeakyReLU -> max over k_pos -> hid_ang.  The two 64->D/2
 * convolutions that follow go through ogmm_gemm_nt. idx rows have stride idx_ld (first k_pos used). */
int ogmm_pos_hidden(const float* xyz, const int32_t* idx, int idx_ld, int k_pos, int C, int N,
                    const float* w_dis /*[64]*/, const float* s_dis, const float* t_dis,
                    const float* w_ang /*[64]*/, const float* s_ang, const float* t_ang,
                    float* hid_dis /*[C*N][64]*/, float* hid_ang /*[C*N][64]*/, void* stream);

/* ---- K9 fused: softmax(Q K^T * scale) V per (cloud, head) without materialising the scores.  models/attn.py:78-82.
 * q [C*N][ldq], k and v [C*M][ldk|ldv], out [C*N][ldo]; all head-major: head h owns columns [h*dh, (h+1)*dh)
 * (the host packs the projection weights that way, models/attn.py:96 interleaves d*H + h).  dh = 128, M in {32,64,128}.
 * fp16x3 split arithmetic on the matrix cores, fp32 softmax (see ogmm_gemm_nt).
 * workspace (device, ogmm_attention_workspace_bytes(C, M, H, dh) bytes, 16-byte aligned): non-NULL selects the transposed kernel (one
 * workgroup per (cloud, head) walks the query tiles and splits K / V into fragment order while staging them in LDS; the buffer itself is
 * only written by the OGMM_ATTN_PACKED / OGMM_ATTN_FRAG variants, which pack the K / V images there first).  With workspace == NULL the
 * first structure runs: every 128-query workgroup stages K / V itself (slower). */
int64_t ogmm_attention_workspace_bytes(int C, int M, int H, int dh);
int ogmm_attention(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, int C, int N, int M,
                   int H, int dh, float scale, float* out, int64_t ldo, void* workspace, void* stream);
/* The same with a term budget for the score product q k^T (HISTORY.md section 4): qk_terms 0 / 3 = three binary16 terms (fp32-class), 1 = both operands
 * rounded to binary16 -- one matrix instruction per block instead of three, no lo part of Q (a permission: kernels without that form run three). */
int ogmm_attention_terms(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, int C, int N, int M,
                         int H, int dh, float scale, float* out, int64_t ldo, int qk_terms, void* workspace, void* stream);

/* ---- K9 middle (unfused fallback): in-place softmax over the last axis (keys).  models/attn.py:80. cols <= 1024. */
int ogmm_softmax_rows(float* x, int64_t rows, int cols, int64_t ld, void* stream);

/* ---- K10 middle: InstanceNorm1d(affine=False, eps) + ReLU in place over the N points of each
 * (cloud, channel).  models/attn.py:24-25.  x [C][N][ld], D channels. */
int ogmm_instnorm_relu(float* x, int64_t ld, int C, int N, int D, float eps, void* stream);

/* statistics -> affine: mean = s1/rows, var = s2/rows - mean^2 (biased); scale = 1/sqrt(var+eps), shift = -mean*scale (models/attn.py:24).
 * clear != 0 (ABI 25): every entry is zeroed behind its read, so the buffer is ready for the next accumulation without a fill by the caller. */
int ogmm_instnorm_finalize(double* col_stats, int64_t n_entries, int rows, float eps, float* scale, float* shift, int clear, void* stream);

/* activation rows x [rows][ld] (fp32) -> split fragment-major image (OGMM_PREC_F16X3_FRAG B operand: [ceil(rows/32)][K/16][64][8]
 * binary16, hi and lo), so that an activation can be the B side of ogmm_gemm_nt (the similarity of models/gmmreg.py:75).
 * Rows beyond `rows` are zero.  K % 16 == 0.  For the 256-column tiles of the large engine pad rows to a multiple of 256 by
 * zero-filling the image (the caller owns the buffer: ceil(rows/32)*32 * K halfs per plane are written). */
int ogmm_pack_frag(const float* x, int64_t ld, int64_t rows, int K, void* hi, void* lo, void* stream);

/* ---- K13 pieces.  models/gmmreg.py:74: F.normalize over channels (eps 1e-12), rows of length D. */
int ogmm_l2norm_rows(const float* x, int64_t ldx, int64_t rows, int D, float* out, int64_t ldo, void* stream);
/* F.normalize(dim = channels) of x [rows][ld] written straight as the split fragment-major B image of ogmm_pack_frag (rows padded to a
 * multiple of 32 with zeros): the tgt side of the N x N similarity (models/gmmreg.py:74-75) never exists as an fp32 map. */
int ogmm_l2norm_pack_frag(const float* x, int64_t ld, int64_t rows, int K, void* hi, void* lo, void* stream);
/* (ABI 25) the same launch with a second job: rnorm_out[row] = 1 / max(|x_rn[row, 0:K]|_2, 1e-12) for the rows of ANOTHER map (ogmm_row_rnorm's result) --
 * both operands of the similarity GEMM of models/gmmreg.py:74-75 (tgt half as image, src half's row scale) are prepared by one kernel. */
int ogmm_l2norm_pack_frag_rnorm(const float* x, int64_t ld, int64_t rows, int K, void* hi, void* lo, const float* x_rn, int64_t ld_rn, int64_t rows_rn,
                                float* rnorm_out, void* stream);
/* Cout = 1 convolutions (proj.net.3, overlap.net.6): y[m] = act(dot(x[m][:], w) + b). */
int ogmm_rowdot(const float* x, int64_t ldx, int64_t rows, int D, const float* w, const float* b /*device [1]*/,
                int act, float* y, int64_t ldy, void* stream);
/* models/gmmreg.py:79-80 given S = src_fn^T tgt_fn [B][N][N]:
 *   wo_src[b][m] = sum_n softmax_n(S[b][m][:])[n] * o_src[b][n]     (sic: reference indexes o_src by n)
 *   wo_tgt[b][n] = sum_m softmax_m(S[b][:][n])[m] * o_tgt[b][m]
 * outputs are written with element stride `ldo` (they are channels 512/513 of the conv2 input). */
int ogmm_overlap_cross(const float* S, int B, int N, const float* o_src, const float* o_tgt, int64_t ldo_in,
                       float* wo_src, float* wo_tgt, int64_t ldo, void* stream);
/* The same in ONE pass over S (64-row x 1024-column tiles emit partial softmaxes for their rows and columns, a second small kernel
 * merges them).  stats: NULL, or [B][4][N] = row max, row exp-sum, column max, column exp-sum as ogmm_overlap_cross_train saves them.
 * workspace: ogmm_overlap_cross_workspace_bytes(B, N) bytes of device memory. */
int64_t ogmm_overlap_cross_workspace_bytes(int B, int N);
int ogmm_overlap_cross_ws(const float* S, int B, int N, const float* o_src, const float* o_tgt, int64_t ldo_in,
                          float* wo_src, float* wo_tgt, int64_t ldo, float* stats, void* workspace, void* stream);

/* ---- K15: overlap-weighted Sinkhorn k-means, the whole E/M loop on chip.  lib/utils.py:269-288
 * (wkeans_plus) with :69-108 (sinkhorn, log domain), :130-140 (gmm_params).  Centres start at
 * xyz[ids0]; p = o / max(sum o, 1e-4); per outer iteration: cost = cdist(xyz, mu)/tau, up to `sk_iters`
 * Sinkhorn sweeps, gamma = exp(K), nan->0, rows / max(rowsum, 1e-3), pi = mean, mu = gamma^T xyz /
 * (N pi + 1e-5).
 * Early exit of the sweeps (lib/utils.py:99-102): `thresh` > 0 ends an E-step's sweeps after the first sweep
 * whose residual sum|u - u0| + sum|v - v0|, averaged over the clouds of one reference call, is below it.  The
 * clouds [g * group_size, (g + 1) * group_size) form call group g (the reference calls wkeans_plus once for the
 * B src clouds and once for the B tgt clouds, models/gmmreg.py:100-101: group_size = B; 0 = all C clouds are one
 * call).  thresh <= 0: every sweep runs.  The exit couples the clouds of a group and nothing else: sharding a
 * batch over ranks changes the groups exactly as the reference's nn.DataParallel scatter does (train.py:190-191).
 *   resid  [C][iters][sk_iters] or NULL: every sweep's residual per cloud (NaN: the sweep did not run / was discarded)
 *   sweeps [C / group_size][iters] int32 or NULL: sweeps every E-step ran per call group (the reference's iteration count)
 *   exit_ws: ogmm_gmm_em_exit_workspace_bytes(...) bytes, 256-byte aligned; may be NULL when thresh <= 0 and resid == NULL.
 *            Its int32 word 1 (byte offset 4) is the call's PROTOCOL-ERROR word: the clouds of a group wait for each other's residuals with bounded
 *            polls, and a poll that runs into its limit (a lost workgroup) sets it to 1 and NaN-poisons pi / mu.  The host reads it behind the call
 *            (OGMM_STATUS_EM_EXIT_PROTOCOL in the model's status word), as it reads the fp16 range flag.
 * ogmm_gmm_em_chip_max_group(N, J): the largest group_size the on-chip kernels accept with thresh > 0 (the clouds of a
 * group wait for each other's residuals, so one resident round of workgroups must hold a whole group); 0 when the
 * shape does not run on chip at all.  ogmm_gmm_em_chip_cached(N, J): 1 when the N x J cost matrix stays in LDS
 * (the fast on-chip form); callers send everything else to ogmm_gmm_em_multi. */
int64_t ogmm_gmm_em_exit_workspace_bytes(int C, int N, int iters, int sk_iters, int group_size);
int ogmm_gmm_em_chip_max_group(int N, int J);
int ogmm_gmm_em_chip_cached(int N, int J);
int ogmm_gmm_em(const float* xyz, const float* o /*[C][N]*/, const int32_t* ids0 /*[C][J]*/, int C, int N, int J,
                int iters, int sk_iters, float epsilon, float tau, double thresh, int group_size,
                float* gamma /*[C][N][J]*/, float* pi /*[C][J]*/, float* mu /*[C][J][3]*/, float* resid, int32_t* sweeps,
                void* exit_ws, void* stream);

/* K15 for shapes whose N x J cost matrix exceeds one CU's LDS: the same loop as a fixed sequence of grid-wide kernels over a cost
 * matrix kept in `workspace` (ogmm_gmm_em_workspace_bytes, 256-byte aligned) -- or, for grids that fit one resident round, one launch whose
 * workgroups meet at per-cloud barriers -- all enqueued by this one call without host synchronisation.  Same arguments and results as
 * ogmm_gmm_em, any group size. */
int64_t ogmm_gmm_em_workspace_bytes(int C, int N, int J);
int ogmm_gmm_em_multi(const float* xyz, const float* o, const int32_t* ids0, int C, int N, int J, int iters, int sk_iters,
                      float epsilon, float tau, double thresh, int group_size, float* gamma, float* pi, float* mu, float* resid,
                      int32_t* sweeps, void* exit_ws, void* workspace, void* stream);

/* ---- K16: mu_feat = gamma^T feats / (N pi + 1e-5).  lib/utils.py:289 / :130-140. */
int ogmm_gmm_feat_mean(const float* gamma, const float* pi, const float* feats, int64_t ld, int C, int N, int J, int D,
                       float* mu_feat /*[C][J][D]*/, void* stream);

/* ---- K17+K18: cluster matching and the weighted rigid solve.  models/dgcnn.py:96-115 (GMMSVD, is_sk=False),
 * lib/utils.py:222-226, lib/se3.py:256-289.  One wavefront per pair; 3x3 SVD by Jacobi in registers (fp64). */
int ogmm_match_kabsch(const float* mu_s /*[B][J][3]*/, const float* mu_t, const float* f_s /*[B][J][D]*/, const float* f_t,
                      int B, int J, int D, float temperature, float* R /*[B][3][3]*/, float* t /*[B][3]*/,
                      float* scores /*[B][J][J] or NULL*/, void* stream);
/* lib/se3.py:256-289 alone: src, corr [B][3][J], w [B][J]. */
int ogmm_kabsch(const float* src, const float* corr, const float* w, int B, int J, float* R, float* t, void* stream);

/* ---- K19: CluLoss.  lib/loss.py:109-118 + :16-57 + lib/utils.py:244-254: anchors = feature of the
 * point nearest to each cluster centre; positives = mu_feat; InfoNCE with 2J-1 logits, label 0.
 * row_loss_sum[c] = sum over the cloud's 2J rows of the cross-entropy (the host divides by C*2J / applies
 * the 0.5 of models/gmmreg.py:110); near[c][j] = chosen point index. */
int ogmm_clu_infonce(const float* xyz, const float* mu, const float* feats, int64_t ld, const float* mu_feat,
                     int C, int N, int J, int D, float tau, float* row_loss_sum /*[C]*/, int32_t* near /*[C][J]*/, void* stream);

/* ---- K20: point-to-point ICP refinement of `forward(is_test=True)`.  lib/o3dutils.py:172-214 (reg_solver ->
 * refine_registration -> open3d registration_icp with TransformationEstimationPointToPoint), called at models/gmmreg.py:115-117
 * with max_corr_dist = 2 * overlap_radius and the network's (R, t) as the initial motion.  open3d's published algorithm,
 * default convergence criteria (pass max_iter 30, rel_fitness = rel_rmse = 1e-6), fp64, the whole loop for one pair on
 * one workgroup.  src [B][N][3], tgt [B][Nt][3] (Nt <= 8192), R0 [B][3][3], t0 [B][3] (NULL = identity) -> R, t;
 * fitness / rmse / iters [B] are optional. */
int ogmm_icp_point_to_point(const float* src, const float* tgt, int B, int N, int Nt, const float* R0, const float* t0,
                            float max_corr_dist, int max_iter, double rel_fitness, double rel_rmse,
                            float* R, float* t, float* fitness, float* rmse, int* iters, void* stream);

/* ---- K21: out[b][i] = min_j |a[b][i] - b[b][j]|^2 (direct-difference form), the reduction the evaluation metrics
 * take over lib/metric.py:193-194's [B][Na][Nb] matrix (chamfer / clipped chamfer / source error, lib/metric.py:221-236). */
int ogmm_min_sqdist(const float* a /*[B][Na][3]*/, const float* b /*[B][Nb][3]*/, int B, int Na, int Nb, float* out /*[B][Na]*/, void* stream);

/* ---- K22: R = V diag(1, 1, det(V U^T)) U^T for M = U S V^T, M [B][3][3] (NaN -> 0): the SVD step of the DeepGMR
 * baseline's `gmm_register` (baseline/deepgmr.py:28-34), fp64 Jacobi in registers like K18. */
int ogmm_rotation_from_cov(const float* M, int B, float* R, void* stream);

/* The same refinement for small batches (one workgroup per pair leaves most of the chip idle below ~256 pairs): every iteration is
 * two grid-wide launches (several workgroups per pair reduce the correspondence sums, one lane per pair updates the motion), all
 * enqueued by this call, finished pairs skipped through a flag in `workspace` (ogmm_icp_workspace_bytes, 16-byte aligned). */
int64_t ogmm_icp_workspace_bytes(int B, int N);
int ogmm_icp_point_to_point_ws(const float* src, const float* tgt, int B, int N, int Nt, const float* R0, const float* t0,
                               float max_corr_dist, int max_iter, double rel_fitness, double rel_rmse,
                               float* R, float* t, float* fitness, float* rmse, int* iters, void* workspace, void* stream);

/* =====================================================================================================
 * Training mode (`model.train()`): forward kernels that differ from eval, and the backward kernels.
 * The reference has no hand-written backward: autograd differentiates the model files; each entry cites the
 * forward expression whose derivative it computes.
 * ===================================================================================================== */

/* ---- T1: train-mode BatchNorm (models/dgcnn.py:126-130, :21-27; models/attn.py:34-57) and InstanceNorm1d
 * (models/attn.py:24) over row groups of a point-major map x [rows][cols] (row stride ldx): group g = rows
 * [g*group_rows, (g+1)*group_rows).  A BatchNorm group is one call of the shared layer (the src or the tgt half of the
 * stacked batch); an InstanceNorm group is one cloud.
 *   ogmm_colstats:        stats[g][c] = {sum x, sum x^2} in fp64 (zeroed by the call)
 *   ogmm_affine_act:      y = act(x * scale[g][c] + shift[g][c]),  act in {NONE, RELU, LEAKY02}
 *   ogmm_norm_bwd_reduce: sums[g][c] = {sum dz, sum dz * xhat}, dz = dy * act'(x*scale + shift), xhat = (x - mean) * rstd  (fp64, zeroed by the call)
 *   ogmm_norm_bwd_apply:  dx = scale * (dz - sums0/n - xhat * sums1/n),  n = group_rows
 * (the activation derivative is taken from the recomputed pre-activation, so the stored output is not re-read)
 *   ogmm_norm_finalize:   stats -> mean = s0/n, var = max(s1/n - mean^2, 0), rstd = 1/sqrt(var + eps), scale = gamma*rstd, shift = beta - mean*scale
 *                         (fp64, the float outputs rounded once; gamma / beta [cols] may be NULL; every output is [G][cols]) in one launch
 * The host turns sums into dgamma, dbeta. */
int ogmm_norm_finalize(const double* stats /*[G][cols][2]*/, int64_t groups, int cols, int64_t group_rows, double eps, const float* gamma, const float* beta,
                       float* scale, float* shift, float* mean, float* rstd, double* mean64, double* var64, void* stream);
/* dgamma[c] = sum over groups of sums[g][c][1], dbeta[c] = ... [0]: the affine parameters' gradients from ogmm_norm_bwd_reduce's (or the GEMM epilogue's) sums, one launch */
int ogmm_norm_param_grads(const double* sums /*[G][cols][2]*/, int groups, int cols, float* dgamma, float* dbeta, void* stream);
/* BatchNorm running statistics after G sequential train-mode calls of the shared layer (torch.nn.BatchNorm1d: running = (1 - momentum) running + momentum batch,
 * variance unbiased by n / (n - 1), n = group_rows; num_batches += G, may be NULL): mean64 / var64 [G][cols] as ogmm_norm_finalize leaves them */
int ogmm_bn_update_running(const double* mean64, const double* var64, int groups, int cols, int64_t group_rows, float momentum, float* running_mean,
                           float* running_var, int64_t* num_batches, void* stream);
int ogmm_colstats(const float* x, int64_t ldx, int64_t rows, int cols, int64_t group_rows, double* stats /*[G][cols][2]*/, void* stream);
int ogmm_affine_act(const float* x, int64_t ldx, int64_t rows, int cols, int64_t group_rows, const float* scale /*[G][cols]*/,
                    const float* shift, int act, float* y, int64_t ldy, void* stream);
/* Upstream gradient of the two backward kernels: dy (dense, may be NULL) plus, for a map that was max-pooled over the k rows of a
 * point, the gradient of the pooled map dpool [rows/k][cols] routed to the winning row arg [rows/k][cols] (NULL = none).
 * ogmm_affine_act_pool is the matching forward: y = act(x*scale + shift) (y may be NULL when only the pooled map is used) and
 * pooled[p][c] = max_j y[p*k + j][c] with the first maximising j in arg; group_points = group_rows / k. */
int ogmm_norm_bwd_reduce(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* dpool, int64_t ldp, const uint8_t* arg, int k,
                         int64_t rows, int cols, int64_t group_rows,
                         const float* scale /*[G][cols]*/, const float* shift, const float* mean, const float* rstd, int act,
                         double* sums /*[G][cols][2]*/, void* stream);
int ogmm_norm_bwd_apply(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* dpool, int64_t ldp, const uint8_t* arg, int k,
                        int64_t rows, int cols, int64_t group_rows,
                        const float* scale, const float* shift, const float* mean, const float* rstd, int act, const double* sums,
                        float* dx, int64_t lddx, void* stream);
int ogmm_affine_act_pool(const float* x, int64_t ldx, int64_t points, int k, int cols, int64_t group_points, const float* scale, const float* shift,
                         int act, float* y, int64_t ldy, float* pooled, int64_t ldp, uint8_t* arg, void* stream);

/* ---- T2: max over the k edges of a point on an un-fused per-edge map (models/dgcnn.py:139,142,145,148; models/attn.py:72):
 * out[p][c] = max_j h[p*k + j][c], arg[p][c] = first maximising j; backward routes dout to that edge and writes zeros elsewhere. */
int ogmm_maxpool_k(const float* h, int64_t ldh, int64_t points, int k, int cols, float* out, int64_t ldo, uint8_t* arg /*[points][cols]*/, void* stream);
int ogmm_maxpool_k_bwd(const float* dout, int64_t ldo, const uint8_t* arg, int64_t points, int k, int cols, float* dh, int64_t ldh, void* stream);

/* ---- T3: operands of the weight-gradient GEMM dW[n][k] = sum_r dY[r][n] X[r][k] (derivative of y = x W^T w.r.t. W) for
 * ogmm_gemm_nt, which contracts over the last axis of both operands.  The contraction is cut into S chunks of `chunk` rows
 * (chunk %% 64 == 0, r zero-padded to S*chunk) that run as the batch dimension of ogmm_gemm_nt (split-K); every chunk's
 * operand is stored compactly with row pitch `pitch` = chunk + 64 (a non power of two: HBM channel spread):
 *   ogmm_transpose_pad: out[s][c][rr] = x[s*chunk + rr][c]                       (A = dY^T, fp32; sA_o = cols*pitch floats, lda = pitch);
 *                       optionally the column sums of x on the way (the bias gradient of the same layer)
 *   ogmm_pack_frag_t:   per chunk the OGMM_PREC_F16X3_FRAG image of X^T, [n_pad/32][pitch/16][64 lanes][8 halfs], hi and lo
 *                       planes (ldb_h = pitch, sB_o = n_pad*pitch halfs); *overflow |= 1 if a finite |x| > 65504 was clamped.
 *                       a_scale != NULL ([rows/group_rows][cols] float, with a_shift): X is taken as relu(x * a_scale + a_shift)
 *                       (relu iff a_relu) -- the same read as struct ogmm_gemm.a_scale, for layers whose normalised input was
 *                       consumed by the forward GEMM that way and never written.
 * The S partial products (sC_o = n*k) are summed afterwards. */
int ogmm_transpose_pad(const float* x, int64_t ldx, int64_t rows, int cols, int64_t chunk, int64_t pitch, int S, float* out,
                       double* colsum /*NULL, or [colsum_slots][cols]: column sums of x (zeroed by the call; the caller adds the copies up)*/,
                       int colsum_slots /*power of two*/, void* stream);
int ogmm_pack_frag_t(const float* x, int64_t ldx, int64_t rows, int cols, int64_t chunk, int64_t pitch, int S, int n_pad, void* hi, void* lo,
                     int* overflow, const float* a_scale, const float* a_shift, int a_relu, int64_t group_rows, void* stream);

/* ---- T12: out = s_0 + ... + s_{n-1} for n <= 8 row-major [rows][cols] maps with their own row pitches (srcs, lds: HOST arrays of n entries): the
 * gradient of a feature map with several consumers in one pass instead of autograd's n - 1 pairwise adds. */
int ogmm_add_n(int n, const float* const* srcs, const int64_t* lds, int64_t rows, int cols, float* out, int64_t ldo, void* stream);

/* ---- T11: backward of the anchor attention (models/attn.py:78-82 under autograd): given dout = dL/dO of
 * O = softmax(Q K^T * scale) V it writes dq [C*N][lddq], dk and dv [C*M][lddk|lddv] (head-major columns like ogmm_attention; every
 * element of the three outputs is written).  The scores are re-formed per 32-query tile inside the kernel and never reach HBM;
 * exact fp32 on v_mfma_f32_32x32x2_f32.  Built for M = 128 anchors and dh = 128 (ogmm_attention_bwd_supported); rows of q / dout
 * 16-byte aligned, rows of k / v 8-byte aligned. */
int ogmm_attention_bwd_supported(int M, int dh);
int ogmm_attention_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* dout,
                       int64_t lddo, int C, int N, int M, int H, int dh, float scale, float* dq, int64_t lddq, float* dk, int64_t lddk,
                       float* dv, int64_t lddv, void* stream);
/* the same backward on the engines' fp16x3 arithmetic (round 5; the fp16x3 training step), every operand split into two binary16 in the kernel.
 * all_products = 0: only the two products that contract over the head dimension (S = Q K^T, dP = dO V^T); the three with P / dS as operands stay
 * exact fp32.  all_products = 1 (csrc/train_attn_bwd16.hip): all five -- P as P * 2^10, dS with a per-tile power of two taken from the tile's largest
 * |dS| (dQ un-scaled per tile, the running dK re-scaled exactly when the exponent changes).  `overflow` (device int32 or NULL) gets bit 0 when
 * q, k, v or dout exceed binary16's range. */
int ogmm_attention_bwd_f16x3(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* dout,
                             int64_t lddo, int C, int N, int M, int H, int dh, float scale, float* dq, int64_t lddq, float* dk, int64_t lddk,
                             float* dv, int64_t lddv, int all_products, int* overflow, void* stream);

/* ---- T9: weight gradient of the thin layers (per-edge EdgeConv maps, the 6 -> 64 edge layer, the 1 -> 64 positional layers):
 * part[s][n][k] = sum over the rows of stream s of dY[r][n] X[r][k], exact fp32 on v_mfma_f32_32x32x2_f32 (HBM-bound);
 * the caller sums the ogmm_weight_grad_thin_streams(n, k) partials (-1: shape not supported, n and k too wide).
 * Rows of dy / x must be aligned to the kernel's vector loads (4 floats for n > 64, 2 for n > 32; 2 floats for k > 32). */
int64_t ogmm_weight_grad_thin_streams(int n, int k);
int ogmm_weight_grad_thin(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t R, int n, int k, float* part, void* stream);
/* the same reduction on the engines' fp16x3 arithmetic (round 5: the fp32 form is matrix-bound on the 256 x 128 per-edge layer): every value is split
 * into two binary16 in registers, three v_mfma_f32_32x32x16_f16 per product block, fp32 accumulation; `overflow` (device int32 or NULL) gets bit 0 when
 * an operand exceeds binary16's range.  Same partial layout, same alignment rules. */
int ogmm_weight_grad_thin_f16x3(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t R, int n, int k, float* part, int* overflow,
                                void* stream);

/* ---- T10 (ABI 28): the small batched products of the training step that have no matrix-core shape (train_small.hip) -- until round 6 these were torch.matmul /
 * torch.bmm calls, i.e. hipBLASLt kernels; exact fp32 fmaf chains in ascending contraction order.
 * ogmm_small_bmm_nn: out[b][i][d] = sum_{j<m} S[b][i][j] X[b][j][d] (+ bias[d]); S and X by element strides (a transposed view costs nothing), out rows contiguous
 * (row pitch ldO); contraction m <= 1024, any rows / D.  Replaces: the forward y = x W^T of the thin layers emd.conv1 (6 -> 64, models/dgcnn.py:121,138),
 * pos.conv_dis.0 and pos.conv_ang1.0 (1 -> 64, models/attn.py:37-47) and their dx = dy W; the feature mean's backward df = gamma (dmu / (pi N + 1e-5))
 * (lib/utils.py:138-140); corr = softmax(...) mu_t (models/dgcnn.py:109); and the S^T dOut / dG X halves of the backward of both forms.
 * ogmm_small_bmm_nt: out[b][i][j] = alpha * sum_d A[b][i][d] B[b][j][d], one workgroup per 64 x 64 output tile, any D.  Replaces: the cluster-feature similarity of the matching
 * (lib/utils.py:222-226 <- models/dgcnn.py:107), the Gram matrix of the clustering loss (lib/loss.py:40-47) and d(scores) = dcorr mu_t^T. */
int ogmm_small_bmm_nn(const float* S, int64_t sS_b, int64_t sS_i, int64_t sS_j, const float* X, int64_t sX_b, int64_t sX_j, int64_t sX_d, const float* bias,
                      int batch, int64_t rows, int m, int D, float* out, int64_t sO_b, int64_t ldO, void* stream);
int ogmm_small_bmm_nt(const float* A, int64_t sA_b, int64_t ldA, const float* Bm, int64_t sB_b, int64_t ldB, int batch, int n, int m, int D, float alpha,
                      float* out, int64_t sO_b, int64_t ldO, void* stream);

/* out[rows[i]][0..D) += g[i][0..D) for i < n (fp32 atomics): the backward of index_points (lib/utils.py:111-127) -- the gradient rows of the anchors /
 * nearest points added into their feature map's gradient (torch's index_add_ until round 6).  rows: int64 row numbers < out_rows (not checked on the device). */
int ogmm_scatter_add_rows(float* out, int64_t ldo, int64_t out_rows, const int64_t* rows, const float* g, int64_t ldg, int64_t n, int D, void* stream);

/* ---- T8: the overlap block (models/gmmreg.py:75-80) in training: ogmm_overlap_cross that also saves the row / column softmax
 * statistics (stats [B][4][N] = row max, row sum, column max, column sum), and its backward: given a = dL/dwo_src, b = dL/dwo_tgt
 * (element stride ldg) it writes dS [B][N][N], g_o_src [B][N] (gradient of the logits the row pass reads, accumulated per column)
 * and g_o_tgt [B][N] in one pass over S.  dL/dfn then follows from dS by two GEMMs. */
int ogmm_overlap_cross_train(const float* S, int B, int N, const float* o_src, const float* o_tgt, int64_t ldo_in, float* wo_src, float* wo_tgt,
                             int64_t ldo, float* stats, void* stream);
int ogmm_overlap_cross_bwd(const float* S, int B, int N, const float* o_src, const float* o_tgt, int64_t ldo_in, const float* wo_src,
                           const float* wo_tgt, int64_t ldo, const float* stats, const float* g_wo_src, const float* g_wo_tgt, int64_t ldg,
                           float* dS, float* g_o_src, float* g_o_tgt, int64_t ldgo, void* stream);

/* ---- T4: backward of K18 (lib/se3.py:256-289): gradients w.r.t. src, corr [B][3][J] and w [B][J] given dL/dR [B][3][3] and
 * dL/dt [B][3] (either may be NULL = zero).  One lane per pair in fp64; the derivative of V D U^T through the 3x3 SVD in
 * closed form (no 1/(s_i^2 - s_j^2) blow-up for equal singular values).  Any of g_src / g_corr / g_w may be NULL. */
int ogmm_kabsch_bwd(const float* src, const float* corr, const float* w, int B, int J, const float* gR, const float* gt,
                    float* g_src, float* g_corr, float* g_w, void* stream);

/* ---- T5: index of the point nearest to each centre (lib/utils.py:244-254, torch.cdist + top-1): near [C][J]. */
int ogmm_nearest_point(const float* xyz, const float* mu /*[C][J][3]*/, int C, int N, int J, int32_t* near, void* stream);

/* Power-of-two scale of a weight for the binary16 split, on the device: scale_out [4 floats, ZERO on entry: [2], [3] are the kernel's scratch]; scale_out[0] = 2^e with max|W| 2^e in [2^top, 2^(top+1)) (e clamped to +-24),
 * inv_out[0 .. inv_len) = 2^-e (handed to ogmm_gemm_nt as its per-column `scale`).  The training step splits its weights every step
 * (train_ops._Linear): no host synchronisation, no cached exponent that could go stale. */
int ogmm_pow2_scale(const float* W, int64_t count, int top, float* scale_out, float* inv_out, int inv_len, void* stream);
/* The same scale (top = 10) AND the OGMM_PREC_F16X3_FRAG image of 2^e W in two launches (round 4: the step re-splits ~120 weights, each was 4-14 small launches):
 * W [rows][cols] contiguous fp32 (ld == cols); transpose = 0: B = W (N = rows; K = cols as the two pieces k1 | cols - k1 of struct ogmm_gemm, each padded with zeros
 * to a multiple of 64; k1 <= 0: one piece), transpose = 1: B = W^T (N = cols, K = rows: the operand of dX = dY W, without a transposed copy).  The image has n_pad
 * (%% 32 == 0, >= N; rows beyond N are zero) rows and ldb_h = padded K columns.  scratch4: four floats, ZERO on entry; on exit [0] = 2^e and [2], [3] are zero again,
 * so a pool of slots can be reused call after call.  inv_out[0 .. inv_len) = 2^-e. */
int ogmm_split_weight(const float* W, int64_t ld, int rows, int cols, int transpose, int k1, float* scratch4, float* inv_out, int inv_len, void* hi, void* lo,
                      int64_t ldb_h, int n_pad, void* stream);

/* ---- T6: constants of the input that feed trainable thin layers, un-fused for training:
 *   ogmm_edge_features: out[(c*N+i)*k + j][0..5] = (x_j - x_i, x_i), j over idx[c][i][:]          (lib/utils.py:47-66)
 *   ogmm_pos_features:  d2[c*N+i] = |p_i - centroid_c|^2;  alpha[(c*N+i)*k + j] = cos angle between the unit offset to the
 *                       j-th neighbour and the unit centroid offset (eps 1e-12)                     (models/attn.py:60-70) */
int ogmm_edge_features(const float* xyz, const int32_t* idx, int C, int N, int k, float* out /*[C*N*k][6]*/, void* stream);
int ogmm_pos_features(const float* xyz, const int32_t* idx, int C, int N, int k, const float* centroid /*[C][3]*/, float* d2, float* alpha, void* stream);

/* ---- T7: backward of ogmm_l2norm_rows (models/gmmreg.py:74): dx = g/n - x (x.g)/n^3, n = max(|x|, 1e-12). */
int ogmm_l2norm_rows_bwd(const float* x, int64_t ldx, const float* g, int64_t ldg, int64_t rows, int D, float* dx, int64_t lddx, void* stream);

/* ---- diagnostics (tools/edgeconv_time.py; not part of the hot path): OGMM_EDGECONV_PROBE=1 makes the fused EdgeConv kernels add the shader cycles of
 * their phases to a device counter, read and cleared here.  (The GEMM engines' ablation / clock-probe builds -- `precision` codes 12..40, 60..89,
 * 100..121 -- their ogmm_debug_v6/v8/v10_probe readers, the retired first LDS-DMA engine and the row-major-planes engine OGMM_PREC_F16X3 are a second
 * build of the engine sources in the tools-only libogmm_probe.so (entry ogmm_probe_gemm_nt, same descriptor); none of it is in this library or its ABI.) */
int ogmm_debug_edgeconv_probe(unsigned long long* host8);
int ogmm_debug_edgeconv_pc_probe(unsigned long long* host8);          /* OGMM_EDGECONV_PROBE=1: shader cycles per phase of the fused EdgeConv kernel */

#ifdef __cplusplus
}
#endif
#endif /* OGMM_HIP_H */
