/*
 * chirrup_amd.h -- C ABI of libchirrup_amd.so, the MI355X (gfx950) RWKV-7 decode kernels.
 *
 * This is the drop-in boundary B2 of SURVEY.md section 8(b): the entry points are what the
 * reference's torch binding for this path calls (file:line given per function, relative to
 * the leonsama/chirrup tree), with three additions every function shares:
 *   - an explicit `void *stream` (a hipStream_t; NULL = the null stream),
 *   - an `int` return: 0 on success, CHIRRUP_E_* (< 0) for arguments the kernel would
 *     mis-handle (the reference only has `assert(H*_N_==C)`), or a positive hipError_t,
 *   - where it shards a state pool: an optional `slot_idx` indirection.
 * Plain pointers and sizes only; no torch types.  All pointers are DEVICE pointers.
 * All kernels are stateless and re-entrant; they may be called concurrently from several host
 * threads / processes, each on its own device and stream, and may be captured in a hipGraph.
 */
#ifndef CHIRRUP_AMD_H
#define CHIRRUP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHIRRUP_OK 0
#define CHIRRUP_E_SHAPE (-1)     /* C != H*64, non-positive sizes, size limits */
#define CHIRRUP_E_NULL (-2)      /* a required pointer is NULL */
#define CHIRRUP_E_ALIGN (-3)     /* pointer / stride not aligned as documented */
#define CHIRRUP_E_UNSUPPORTED (-4)

/* Library / ABI version and the gfx target the kernels were built for ("gfx950").
 * CHIRRUP_ABI_VERSION changes whenever an exported prototype changes incompatibly; a caller compares it with
 * chirrup_abi_version() of the library it loaded before making any other call (chirrup_amd/lib.py does).
 *   1  round 1
 *   2  round 2: skinny_gemm_f16 / _f16_group / _f16_partial gained row_halves (and tile_counters) ahead of `stream`,
 *      skinny_gemm_f16_grouped gained w_tiled, skinny_gemm_group_workspace_bytes gained K, skinny_gemm_select removed
 *   3  round 3: chirrup_mm8_fuse gained out_planes (the struct grew); new entry points only otherwise (rwkv7_tmix_gemms,
 *      rwkv7_tmix_gemms_mm8, rwkv7_tmix_wkv7_fused_mm8, skinny_untile_weight, the clock probes)
 *   4  round 4: rwkv7_tmix_gemms / _mm8 gained `status` (a sticky status word of the caller's) ahead of spin_limit and the status
 *      word moved to the END of the sync words; rwkv7_commit_sampled gained status_src / status_dst ahead of `stream`;
 *      new: mm8t_seq_exact, mm8t_exact_workspace_bytes, mm8_dequant_f16, chirrup_device_cu_count, rwkv7_penalize_argmax_listed,
 *      rwkv7_commit_sampled_listed */
#define CHIRRUP_ABI_VERSION 4
int chirrup_abi_version(void);
const char *chirrup_target_arch(void);

/*
 * WKV7 recurrent state update over T timesteps (decode: T = 1).
 * Replaces: cuda_forward_seq  Albatross/cuda/rwkv7_state_fwd_fp16.cpp:6 (launcher
 *           Albatross/cuda/rwkv7_state_fwd_fp16.cu:313-316, kernel :26-97), reached from
 *           torch.ops.rwkv7_state_fwd_fp16.forward_seq (Albatross/rwkv7.py:151).
 * state     binary16 [n_slots][H][64][64], updated in place. Batch row b uses slot
 *           slot_idx[b] when slot_idx != NULL, else slot b. Slots are slot_stride ELEMENTS
 *           apart (0 = dense, H*64*64). 16-byte aligned; slot_stride % 8 == 0.
 *           Two batch rows must not name the same slot.
 * r,w,k,v,a,b  binary16 [B][T][C] contiguous.   y  binary16 [B][T][C] out.
 * elapsed_t int32 [B] (tokens already consumed by each row; feeds the decay dither).
 * Arithmetic: exactly spec A1 of SURVEY.md section 8 -- binary16 storage AND binary16
 * accumulate in the reference's two-lane order, one rounding per operation, no FMA.
 */
int wkv7_fwd_seq(int B, int T, int C, int H, void *state, const void *r, const void *w,
                 const void *k, const void *v, const void *a, const void *b, void *y,
                 const int32_t *elapsed_t, const int32_t *slot_idx, int64_t slot_stride,
                 void *stream);

/* Single-timestep form.  Replaces: cuda_forward_one  Albatross/cuda/rwkv7_state_fwd_fp16.cpp:7
 * (launcher .cu:318-322, kernel .cu:99-167; torch.ops.rwkv7_state_fwd_fp16.forward_one,
 * Albatross/rwkv7.py:96,132).  r..b, y are [B][C]. Same arithmetic as wkv7_fwd_seq with T=1. */
int wkv7_fwd_one(int B, int C, int H, void *state, const void *r, const void *w, const void *k,
                 const void *v, const void *a, const void *b, void *y, const int32_t *elapsed_t,
                 const int32_t *slot_idx, int64_t slot_stride, void *stream);
/* The state-independent part of the update taken out of the sequential loop, for chunks of many tokens: wkv7_decay computes
 * w~ = exp(-e^-0.5 * sigmoid(w)) - 1 + dither(elapsed_t[b] + t) for every row at once (cu:23, :59; w, w_out binary16
 * [B][T][C]); wkv7_fwd_seq_decayed is wkv7_fwd_seq given w~ instead of w.  Together bit-identical to wkv7_fwd_seq. */
int wkv7_decay(int B, int T, int C, const void *w, const int32_t *elapsed_t, void *w_out, void *stream);
int wkv7_fwd_seq_decayed(int B, int T, int C, int H, void *state, const void *r, const void *w_decayed, const void *k,
                         const void *v, const void *a, const void *b, void *y, const int32_t *elapsed_t,
                         const int32_t *slot_idx, int64_t slot_stride, void *stream);

/*
 * Sparse binary16 vector x matrix: out[c] += sum_{d : vec[d] != 0} vec[d] * mat[d][c].
 * Replaces: cuda_spmv_forward  Albatross/cuda/rwkv7_state_fwd_fp16.cpp:8 (launcher .cu:324-330,
 *           kernel .cu:222-310) and the ROCm path rwkv_mm_sparsity,
 *           Albatross/rwkv_mm_op_triton.py:6-61 (bsz = 1 channel-mix, Albatross/rwkv7.py:659).
 * vec [D], mat [D][C] row-major, out [C]; binary16. As in the reference `out` is ACCUMULATED
 * into, so the caller zeroes it (Albatross/rwkv7.py:65). Requires C % 8 == 0.
 * Accumulation is binary32 in row order (the Triton path's arithmetic) and deterministic.
 * workspace: device scratch of spmv_fp16_workspace_bytes(D, C) bytes (may be NULL if 0).
 */
int64_t spmv_fp16_workspace_bytes(int D, int C);
int spmv_fp16(int D, int C, const void *vec, const void *mat, void *out, void *workspace,
              void *stream);

/*
 * mm8 (w8a16): y[i][k] = sum_j x[i][j] * ((w[j][k] + 0.5) * rx[k] * ry[j] + mx[k] + my[j]).
 * Replaces: rwkv_pip::mm8_seq and rwkv_pip::mm8_seq_opt  scripts/test_mm8/rwkv_pip_wrapper.cpp:51-84, :206-211
 *           (kernels scripts/test_mm8/rwkv_pip_operators.cu:59-97 and the _opt / wmma forms :205-558).
 * x [B][N] binary16 (row stride x_stride elements), w [N][M] uint8 (row stride w_stride),
 * mx,rx [M], my,ry [N] binary16, y [B][M] binary16 (row stride y_stride).
 * Runs on the matrix cores: w is re-laid into K-contiguous tile images in the workspace (mm8_pack) and multiplied by
 * mm8t_seq, i.e. binary32 accumulate in the algebraically split form of scripts/test_mm8/benchmark.py:167-179
 * (xs = fp16(x*ry) through MFMA, rank-1 corrections after) -- the reference's own tolerance between its split and
 * direct forms is rtol 1e-3 (benchmark_pure_pytorch.py:92).  Needs N % 64 == 0, M % 128 == 0, 16-byte aligned x / w and
 * w_stride % 16 == 0; any other shape runs mm8_seq_direct.  Static weights: pack once with mm8_pack and call
 * mm8t_seq(w_tiled = 1) directly (what chirrup_amd.ops.mm8_seq does behind a per-tensor cache).
 * workspace: mm8_seq_workspace_bytes(B, N, M) bytes of device scratch (0 for the direct fallback).
 */
int64_t mm8_seq_workspace_bytes(int B, int N, int M);
int mm8_seq(int B, int N, int M, const void *x, int x_stride, const void *w, int w_stride,
            const void *mx, const void *rx, const void *my, const void *ry, void *y,
            int y_stride, void *workspace, void *stream);
int mm8_seq_opt(int B, int N, int M, const void *x, int x_stride, const void *w, int w_stride,
                const void *mx, const void *rx, const void *my, const void *ry, void *y,
                int y_stride, void *workspace, void *stream);
/* The as-coded kernel (rwkv_pip_operators.cu:59-83): binary32 accumulate over j in order, one rounding per operation;
 * bit-identical to oracle_mm8_seq.  Any shape; FP32 VALU, not MFMA -- parity and ragged shapes only. */
int mm8_seq_direct(int B, int N, int M, const void *x, int x_stride, const void *w, int w_stride,
                   const void *mx, const void *rx, const void *my, const void *ry, void *y,
                   int y_stride, void *stream);
/* w [N][M] uint8 (row stride w_stride, % 16 == 0; N % 64 == 0, M % 128 == 0) -> mm8_packed_bytes(N, M) = N*M bytes in the
 * layout mm8t_seq(w_tiled = 1) reads: the transposed matrix wT [M][N] cut into 8-KiB (128 x 64) tile images. */
int64_t mm8_packed_bytes(int N, int M);
int mm8_pack(int N, int M, const void *w, int w_stride, void *packed, void *stream);

/* GEMV form, binary32 output that the caller zeroes (the reference accumulates with
 * atomicAdd).  Replaces: rwkv_pip::mm8_one  scripts/test_mm8/rwkv_pip_wrapper.cpp:86-119
 * (kernel scripts/test_mm8/rwkv_pip_operators.cu:150-189). x [N], y [M] float. */
int mm8_one(int N, int M, const void *x, const void *w, int w_stride, const void *mx,
            const void *rx, const void *my, const void *ry, float *y, void *stream);

/*
 * Fused element-wise chains between the GEMMs of one layer.  They replace runs of separate torch
 * ops in Albatross/rwkv7.py (lines given); every op of a chain still rounds to binary16 once, as
 * in the reference's eager execution, reductions accumulate in binary32.  All tensors binary16,
 * rows of C channels, C % 64 == 0, 16-byte aligned.
 *
 * rwkv7_add_ln_mix: x_new = x (+ delta, may be NULL) -> x_out (may be NULL; may alias x when T == 1 -- with T > 1, a
 *   delta and n_mix > 0 it must be a different buffer: row t re-reads x[t-1] + delta[t-1] for the token shift);
 *   cur = LayerNorm(x_new; ln_w, ln_b, eps); with n_mix in {1, 6}: token shift against the previous
 *   row (t > 0) or prev_in[b] (t == 0), out[m] = cur + (prev - cur) * mix[m], prev_out[b] = cur of
 *   the last row; with n_mix == 0: out = cur.   x,delta,x_out,out[m]: [B][T][C]; prev_*: [B][C];
 *   mix: [n_mix][C]; out planes are out_stride elements apart.  T > 1 needs prev_out != prev_in.
 *   slot_idx (may be NULL): batch row b carries its token-shift state in row slot_idx[b] of prev_*.
 *   delta_partials (may be NULL; then delta must be NULL): float [delta_splits][B][T][C] split-K partial sums of
 *   the GEMM that produced delta (skinny_gemm_f16_partial); delta = binary16(sum over splits) is formed in this
 *   kernel's prologue, saving the GEMM's separate reduce launch.
 *   Replaces rwkv7.py:523 + :621-623 (n_mix 6), :531-533 + :675-677 (n_mix 1), :548-550 (n_mix 0).
 */
int rwkv7_add_ln_mix(int B, int T, int C, int n_mix, const void *x, const void *delta, void *x_out,
                     const void *ln_w, const void *ln_b, float eps, const void *prev_in, void *prev_out,
                     const void *mix, void *out, int64_t out_stride, const int32_t *slot_idx,
                     const float *delta_partials, int delta_splits, void *stream);
/* The same kernel next to mm8 (w8a16) GEMMs, whose split form (scripts/test_mm8/benchmark.py:167-179) has an activation
 * prologue (xs = fp16(x*ry), S = {sum xs, sum x*my, sum x} per row) and rank-1 corrections after the matrix product:
 *   in_*  : delta_partials hold the CORE sums of an mm8 product (mm8t_gemm_partial) with output scales in_rx, in_mx [C] and
 *           the row sums in_S [B*T][3] of ITS prologue: delta = fp16(rx*(sum - 1023.5*S0) + S1 + mx*S2) is formed here (the
 *           u8 kernels multiply by 1024 + q, so the core sums carry 1024*S0, removed with the +0.5*S0 term) -- the
 *           product's reduce launch folded into this kernel (as delta_partials does for the binary16 GEMMs);
 *   out_* : (n_mix == 1 only) besides out, write the prologue of the mm8 product that consumes out: out_xs [B][T][C] and
 *           out_S [B*T][3] for input scales out_ry, out_my [C] -- that product's prologue launch folded in.
 * Either half may be all-NULL; fuse == NULL is rwkv7_add_ln_mix. */
typedef struct {
    const void *in_rx, *in_mx;   /* binary16 [C] */
    const float *in_S;           /* [B*T][in_S_parts][3]: the row sums as in_S_parts partial sums each (added up here) */
    const void *out_ry, *out_my; /* binary16 [C] */
    void *out_xs;                /* binary16 [B][T][C] */
    float *out_S;                /* [B*T][3] */
    int in_S_parts;              /* 0 or 1: one sum per row; mm8_reduce_rows writes mm8_row_parts(C) of them, mm8t_gemm_fused mm8_tile_parts(C) */
    int out_planes;              /* 0 or 1: the prologue of mix plane 0 (n_mix = 1, or 6); 2..3 (n_mix = 6): of planes 0..out_planes-1
                                    (r, k, v) -- out_ry / out_my then [out_planes][C], out_xs [out_planes][B*T][C] (plane stride =
                                    out_stride), out_S [out_planes][B*T][3] */
} chirrup_mm8_fuse;
int rwkv7_add_ln_mix_mm8(int B, int T, int C, int n_mix, const void *x, const void *delta, void *x_out,
                         const void *ln_w, const void *ln_b, float eps, const void *prev_in, void *prev_out,
                         const void *mix, void *out, int64_t out_stride, const int32_t *slot_idx,
                         const float *delta_partials, int delta_splits, const chirrup_mm8_fuse *fuse, void *stream);

/* rwkv7.py:629-637: a = sigmoid(a_pre); kk = normalize(k*k_k) per 64-channel head;
 * k <- k*(1+(a-1)*k_a) in place; neg_kk = -kk; kka = kk*a; and, when v_first != NULL (layer > 0),
 * v <- v + (v_first - v)*sigmoid(vg_pre) in place.  rows = B*T. */
int rwkv7_tmix_mid(int64_t rows, int C, void *k, void *v, const void *a_pre, const void *vg_pre,
                   const void *v_first, const void *k_k, const void *k_a, void *neg_kk, void *kka,
                   void *stream);

/* rwkv7.py:647-649 up to the output projection: out = (group_norm_H(y; lnx_w, lnx_b, eps) +
 * (sum_head r*k*r_k) * v) * g. */
int rwkv7_tmix_post(int64_t rows, int C, const void *y, const void *r, const void *k, const void *v,
                    const void *g, const void *r_k, const void *lnx_w, const void *lnx_b, float eps,
                    void *out, void *stream);

/* LoRA hidden activations in place on `nplanes` consecutive planes of plane_elems binary16 values,
 * plane ids first_plane.. in the order [v, w, a, g]: w -> tanh (rwkv7.py:626), g -> sigmoid (:630),
 * a and v unchanged (:629, :637). */
int rwkv7_lora_act(int nplanes, int first_plane, int64_t plane_elems, void *hbuf, void *stream);

/*
 * Fused time-mix core (one launch per layer): the gating chain of rwkv7.py:629-637 (as rwkv7_tmix_mid),
 * the WKV7 update of wkv7_fwd_seq with a = -kk, b = kk*a, and the output chain of :647-649 (as
 * rwkv7_tmix_post), all inside the WKV7 kernel -- lane j of a head's wavefront holds channel j, the
 * head reductions (L2 norm, group-norm moments, bonus sum) are wavefront shuffles, and k', v', -kk,
 * kk*a, y never exist in HBM.  r, w, k, v are the RAW projections ([B][T][C]); out [B][T][C].
 * vg_pre / v_first both NULL for layer 0.  Same arithmetic (one rounding per torch op) as the three
 * separate kernels; only the order of the binary32 head reductions differs.
 */
int rwkv7_tmix_wkv7_fused(int B, int T, int C, int H, void *state, const void *r, const void *w, const void *k,
                          const void *v, const void *a_pre, const void *vg_pre, const void *v_first, const void *g,
                          const void *k_k, const void *k_a, const void *r_k, const void *lnx_w, const void *lnx_b,
                          float eps, void *out, const int32_t *elapsed_t, const int32_t *slot_idx,
                          int64_t slot_stride, void *stream);
/* ... with the mm8 (w8a16) activation prologue of the GEMM that consumes `out` (att.output as uint8 weights, scripts/test_mm8/
 * benchmark.py:167-173, :447-452): out receives xs = binary16(o * ry) instead of o, and S [B*T][H][3] (float) each head's share of
 * the row sums {sum xs, sum o*my, sum o} (the consumer adds the H parts in head order: rwkv7_add_ln_mix_mm8 with in_S_parts = H).
 * ry = NULL: identical to rwkv7_tmix_wkv7_fused. */
int rwkv7_tmix_wkv7_fused_mm8(int B, int T, int C, int H, void *state, const void *r, const void *w, const void *k, const void *v,
                              const void *a_pre, const void *vg_pre, const void *v_first, const void *g, const void *k_k,
                              const void *k_a, const void *r_k, const void *lnx_w, const void *lnx_b, float eps, void *out,
                              const int32_t *elapsed_t, const int32_t *slot_idx, int64_t slot_stride, const void *ry,
                              const void *my, float *S, void *stream);

/* rwkv7.py:678: x <- relu(x)**2 in place over n elements (n % 8 == 0). */
int rwkv7_relu_sq(int64_t n, void *x, void *stream);

/*
 * Skinny-M MFMA GEMM for the decode step: Y[M][N] = act(X[M][K] . W[N][K]^T + bias), M <= 256,
 * binary16 in / binary32 accumulate / binary16 out.  Plays the role of the reference's
 * F.linear / torch.matmul calls on the projection weights (Albatross/rwkv7.py:625-630, :649, :678-679,
 * :551) for batch sizes where the library GEMM is operand-ingest bound.
 * W row-major [N][K] (the layout the reference keeps its `*.weight` tensors in), row stride ldw
 * elements; K % 64 == 0, N % 4 == 0, ldx % 8 == 0, ldw % 8 == 0.  act: 0 none, 1 relu(.)^2 (rwkv7.py:678).
 * splits: K-split factor (0 = the library's choice: enough workgroups for the 256 CUs at the kernel's tile width);
 * needs skinny_gemm_workspace_bytes(M,N,K,splits) bytes of device scratch when that is > 0.
 * skinny_gemm_splits: the factor a call with these sizes uses (Z = problems of a batched launch, 1 otherwise).
 */
int64_t skinny_gemm_workspace_bytes(int M, int N, int K, int splits);
int skinny_gemm_splits(int N, int K, int Z, int splits);
/* As skinny_gemm_f16 without bias/activation, but leaves the `splits` binary32 partial results
 * [splits][M][N] in `partials` for the consumer to sum (see rwkv7_add_ln_mix). Returns the split count used
 * (> 0) or a negative CHIRRUP_E_* / positive hipError_t is NOT distinguishable here, so errors are < 0 only. */
int skinny_gemm_f16_partial(int M, int N, int K, const void *X, int ldx, const void *W, int64_t ldw, int w_tiled, int splits,
                            int row_halves, float *partials, void *stream);
/* Z independent skinny GEMMs in ONE launch (RWKV-7's receptance/key/value projections, Albatross/rwkv7.py:603-605,
 * and its four LoRA pairs, :626-637): Y[z] = act(X[z] . W[z]^T + bias[z]).  Problem z's operands start z * (their
 * batch stride, in elements) after problem 0's; bias may be NULL.  act: 0 none, 1 relu(.)^2, 4 + p: LoRA hidden
 * planes (v, w, a, g) with problem 0 = plane p: tanh on w, sigmoid on g.  K must be a multiple of 64, M <= 256.
 * Needs skinny_gemm_batched_workspace_bytes(...) bytes of scratch when splits != 1 or act != 0. */
int64_t skinny_gemm_batched_workspace_bytes(int Z, int M, int N, int K, int splits);
int skinny_gemm_f16_batched(int Z, int M, int N, int K, const void *X, int ldx, int64_t x_bs, const void *W, int64_t ldw,
                            int64_t w_bs, const void *bias, int64_t bias_bs, void *Y, int ldy, int64_t y_bs, int act,
                            int splits, void *workspace, void *stream);
/* The same with a reduction length per problem: problem z uses only the first k_of[z] columns of X[z] and W[z]
 * (k_of[z] <= K, multiple of 64; NULL = K for all).  Operands zero-padded to a common K (the LoRA ranks 96/128/128/480
 * packed as 512) are then not streamed beyond their real rank.  Z <= 8; splits must be 1 when k_of is given.
 * w_tiled: every W[z] is the tile image (skinny_tile_weight) of its [N][K] matrix, N % 128 == 0.
 * row_halves: as below, honoured for splits == 1 and act == 0 (the LoRA up-projections: 4 x 32 tiles of 2..8 K-blocks get
 * twice the workgroups and, with half the x image per stage, a deeper operand ring). */
int skinny_gemm_f16_grouped(int Z, int M, int N, int K, const int *k_of, const void *X, int ldx, int64_t x_bs, const void *W,
                            int64_t ldw, int64_t w_bs, int w_tiled, const void *bias, int64_t bias_bs, void *Y, int ldy,
                            int64_t y_bs, int act, int splits, int row_halves, void *workspace, void *stream);
/* Up to 8 GEMMs that share M, K, the row strides of x and W (ldx, ldw) and the split count, in ONE launch plus one
 * reduce launch: y_i = act_i(x_i . w_i^T + bias_i), w_i binary16 [n_i][K].  This is one RWKV-7 layer's receptance /
 * key / value projections together with its four LoRA down-projections and their activations (Albatross/rwkv7.py:
 * 625-637): seven independent GEMMs over the same token rows that the reference issues one by one.
 * act: 0 none, 1 relu(.)^2, 2 tanh, 3 sigmoid (applied to the binary16-rounded sum, like a separate torch op).
 * splits = 0: the library's choice.  Unsplit launches apply bias / activation in the GEMM epilogue; split ones go through
 * binary32 partials and one reduce launch: workspace = skinny_gemm_group_workspace_bytes(...) bytes, 256-byte aligned.
 *
 * row_halves = 1 (here, in skinny_gemm_f16, skinny_gemm_f16_partial and skinny_gemm_f16_grouped; honoured for M > 32 with
 * the 128-column kernel): every tile and K-slice is worked on by TWO workgroups, one per half of the rows -- twice the
 * workgroups without more partial planes (an unsplit launch then fills the chip with no partials and no reduce launch);
 * both stream the W tile, side by side on one XCD so that HBM sees it once.  splits = 0 then chooses for twice the tiles;
 * skinny_gemm_splits / *_workspace_bytes stay upper bounds.  Results are bit-identical to row_halves = 0 at the same
 * split count (a row's sums never depend on other rows). */
typedef struct {
    const void *x;     /* [M][ldx] binary16 */
    const void *w;     /* [n][ldw] binary16 */
    void *y;           /* [M][ldy] binary16 */
    const void *bias;  /* [n] binary16 or NULL */
    int n, ldy, act;
    int w_tiled;       /* w is in the tile-image layout (skinny_tile_weight); needs n % 128 == 0 */
} chirrup_gemm_problem;
int64_t skinny_gemm_group_workspace_bytes(int count, const chirrup_gemm_problem *problems, int M, int K, int splits);
/* tile_counters (skinny_gemm_f16_group, skinny_gemm_f16; may be NULL): skinny_gemm_pair_counters() ints, ZERO before their first
 * use and used by no other launch that may run concurrently (one set per stream).  With them a launch of at most 32 rows whose
 * split count is 2..4 (at most 48 KB of partials per tile) needs no reduce launch: every K-slice of a
 * tile writes its binary32 partial, the last to finish adds the partials in slice order (its own from on-chip sums) and applies
 * bias / activation itself; it leaves the counter at zero.  Bit-identical to the reduce-launch path.
 * The hand-off between the workgroups uses write-through stores, one agent-scope atomic per workgroup and L1-bypassing loads, no
 * fences -- a form measured on gfx950 for ONE workgroup per CU and memory from hipMalloc (MI355X_MICROARCH.md, inter-workgroup
 * visibility): `workspace` and `tile_counters` must be hipMalloc'ed device memory (not host-mapped, not managed), and the library
 * only takes this path for launches whose workgroups need more than half of a CU's LDS (so that two can never share a CU).
 * A launch that does not complete (device fault, process killed) may leave counters non-zero: zero them again before reuse --
 * chirrup_amd does so at the head of every captured decode graph and after any failed call. */
int skinny_gemm_pair_counters(void);
/* Diagnostic for bench.py / tools (process-wide, not thread-safe): while buf != NULL every 128-column GEMM launch of at most
 * `pairs` workgroups writes, per workgroup, the duration of its main loop as {shader-clock ticks (s_memtime), ticks of the
 * constant 100-MHz counter (s_memrealtime)} into buf (uint64 [2 * pairs]); ticks / (100-MHz ticks) * 100 = the clock in MHz the
 * CU ran the loop at.  When pairs >= 3 x the launch's workgroups, 4 absolute 100-MHz stamps per workgroup follow the pairs
 * (buf[2 * workgroups + 4 * wg + {0..3}]: kernel entry, main loop start, main loop end, epilogue stores done): the launch's
 * timeline (tools/gemm_timeline.py).  buf = NULL switches it off again. */
int skinny_gemm_clock_probe(void *buf, int pairs);
/* Experiment (tools/exp_warm.py): while stages > 0 every 128-column launch is preceded by a launch of the
 * same grid that touches the first `stages % 100` K-blocks of each workgroup's W tile (stages / 100: 1 = the next tile's bytes,
 * 3 = a tile half the matrix away, 2 = no loads but `stages % 100` x ~4 us of idling); sink: 4 writable bytes.  Process-wide. */
int skinny_gemm_warm_probe(int stages, void *sink);
/* The chip's clock without load: one wavefront spins `iters` dependent VALU operations; out (uint64 [2]) receives {shader-clock
 * ticks, 100-MHz ticks}. */
int chirrup_clock_probe(int iters, void *out, void *stream);
/* Diagnostic (tools/ln_timeline.py): while buf is set every rwkv7_add_ln_mix* launch writes 8 100-MHz stamps per workgroup
 * (entry, row data arrived, mean, variance, normalised, stores issued, stores acknowledged, unused); NULL switches it off. */
int rwkv7_ln_probe(void *buf);
/* Launch-cost probe (tools/launch_cost.py): `grid` workgroups of `block` lanes that hold lds_bytes of LDS, idle for sleep x ~4 us
 * and do nothing else. */
int chirrup_noop_launch(int grid, int block, int lds_bytes, int sleep, void *sink, void *stream);
int skinny_gemm_f16_group(int count, const chirrup_gemm_problem *problems, int M, int K, int ldx, int64_t ldw, int splits,
                          int row_halves, void *workspace, void *tile_counters, void *stream);
int skinny_gemm_f16(int M, int N, int K, const void *X, int ldx, const void *W, int64_t ldw, int w_tiled, const void *bias,
                    void *Y, int ldy, int act, int splits, int row_halves, void *workspace, void *tile_counters, void *stream);
/*
 * R/K/V and the WHOLE LoRA chain of one RWKV-7 layer (Albatross/rwkv7.py:625-630, :637: four down-projections, tanh / sigmoid,
 * four up-projections + bias) in ONE launch: the chain runs on the CUs the R/K/V tiles leave idle, beside them, instead of as
 * a second launch behind them (round 2: 12 us per layer at 7.2B / bsz 200).
 *   main[i]  (n_main <= 4):  y = x . w^T                                        (as chirrup_gemm_problem; act as there)
 *   lora[p]  (n_lora <= 4):  hid = act(x . w^T)  [M][n of ld_hid], then  y = hid[:, :k_up] . w_up^T + bias  [M][up_n of up_ldy]
 * 1 <= M <= 256 rows (row_halves = 1 and M > 32: two workgroups per tile over the two halves of the rows, as elsewhere; whole-row
 * tiles otherwise, M <= 128 -- at <= 32 rows the main tiles are then split 2..4 ways over K and reduced inside the launch like
 * skinny_gemm_f16's), shared K (% 64), ldx; w of lora[p] row-major [n][K] (row stride ldw, n % 64 == 0); w_up a tile image
 * (skinny_tile_weight) of [up_n][up_kimg] whose first k_up columns are multiplied (k_up <= n, % 64); act: 0 none, 2 tanh,
 * 3 sigmoid.  workspace: rwkv7_tmix_gemms_workspace_bytes bytes; sync: rwkv7_tmix_sync_words() ints -- both hipMalloc'ed, sync
 * ZERO before the first launch (every completed launch leaves it zero; the workgroups hand tiles to each other through it with
 * write-through stores, one agent-scope atomic per workgroup and ONE agent-scope acquire per consumer: cdna_hip_programming.md
 * Guideline 16).  Every wait is bounded (spin_limit polls of ~0.25 us, 0 = default ~0.1 s): if one gives up -- another tenant
 * held most of the chip that long -- the STATUS word is OR-ed non-zero and this launch's LoRA outputs are undefined.  status: an
 * int32 of the caller's (one per device is enough: launches only ever OR into it, nobody but its owner clears it), or NULL =
 * sync[rwkv7_tmix_status_word()], the LAST sync word; words [0, rwkv7_tmix_status_word()) may be zeroed between launches at
 * any time (a captured decode graph zeroes them at the head of every replay) -- the status word must not be, or a step that
 * gave up is erased by the next one (ABI 4; rwkv7_commit_sampled hands the word to the host behind every step's sampled ids).
 * One launch at a time per (workspace, sync).  All chain workgroups must be resident at once (they wait for each other):
 * CHIRRUP_E_UNSUPPORTED on a device with fewer compute units than the chain has workgroups (chirrup_device_cu_count()).
 */
typedef struct {
    const void *x;      /* [M][ldx] binary16 */
    const void *w;      /* down-projection [n][K] binary16, row stride ldw */
    void *hid;          /* [M][ld_hid] binary16: columns [0, n) are written */
    const void *w_up;   /* tile image of the up-projection [up_n][up_kimg] */
    const void *bias;   /* [up_n] binary16 or NULL */
    void *y;            /* [M][up_ldy] binary16 */
    int n, k_up, act;
} chirrup_lora_problem;
/* A uint8 (mm8, w8a16) main problem of rwkv7_tmix_gemms_mm8: y = mm8(x, w) in the split form of scripts/test_mm8/benchmark.py:
 * 167-179 -- xs [M][ldx] = binary16(x * ry) and S [M][3] = {sum xs, sum x*my, sum x} are the product's activation prologue
 * (rwkv7_add_ln_mix_mm8 writes them for r, k, v with out_planes = 3), w the K-contiguous uint8 weights [n][K] (tile images of
 * skinny_tile_weight_u8 when w_tiled), rx / mx [n] the column scales.  n % 8 == 0, ldy % 8 == 0, 16-byte aligned y / rx / mx. */
typedef struct {
    const void *xs;
    const void *w;
    void *y;
    const void *rx, *mx;
    const float *S;
    int n, ldy, w_tiled;
} chirrup_mm8_problem;
int rwkv7_tmix_sync_words(void);
int rwkv7_tmix_status_word(void);
int64_t rwkv7_tmix_gemms_workspace_bytes(int M, int K, int n_main, const chirrup_gemm_problem *main_problems, int n_lora,
                                         const chirrup_lora_problem *lora, int row_halves);
int rwkv7_tmix_gemms(int M, int K, int ldx, int64_t ldw, int n_main, const chirrup_gemm_problem *main_problems, int n_lora,
                     const chirrup_lora_problem *lora, int ld_hid, int up_n, int up_kimg, int up_ldy, int row_halves, void *workspace,
                     void *sync, void *status, int spin_limit, void *stream);
/* Compute units of the current device (what the time-mix launch sizes its chain by). */
int chirrup_device_cu_count(void);
/* ... with uint8 main problems (R/K/V as mm8 weights; the rank-1 corrections run in each tile's epilogue); the LoRA problems,
 * workspace (size it with the binary16 call's function, passing n and w_tiled of the uint8 problems) and sync as above. */
int rwkv7_tmix_gemms_mm8(int M, int K, int ldx, int64_t ldw, int n_main, const chirrup_mm8_problem *main_problems, int n_lora,
                         const chirrup_lora_problem *lora, int ld_hid, int up_n, int up_kimg, int up_ldy, int row_halves,
                         void *workspace, void *sync, void *status, int spin_limit, void *stream);
/* Weights in the ring kernel's tile-image layout (w_tiled = 1 above and in chirrup_gemm_problem): W [N][K] binary16,
 * N % 128 == 0, K % 64 == 0, re-laid so that each (128 rows x 64 k) tile is 16 KiB of consecutive bytes in the order the
 * kernel keeps it in LDS.  A 1-KiB LDS-DMA wave-instruction then reads 1 KiB of consecutive memory instead of eight
 * 128-B pieces of eight rows K*2 bytes apart.  Wt: N*K elements, must not overlap W.  The row-major matrix is still
 * what every other consumer (library GEMMs of the prefill path) needs. */
int skinny_tile_weight(int N, int K, const void *W, int64_t ldw, void *Wt, void *stream);
/* ... and back: the row-major matrix (row stride ldw) of a tile image, for callers that keep only the images and need a
 * row-major operand now and then (library GEMMs of prefill chunks above 256 rows). */
int skinny_untile_weight(int N, int K, const void *Wt, void *W, int64_t ldw, void *stream);
/* The same for mm8t_seq's uint8 weights wT [M_out][N_in] (N = M_out, K = N_in): 8-KiB tile images, w_tiled = 1 there. */
int skinny_tile_weight_u8(int N, int K, const void *W, int64_t ldw, void *Wt, void *stream);

/*
 * mm8 (w8a16) on the matrix cores: same quantisation and result as mm8_seq, but the uint8 weights
 * are given K-CONTIGUOUS, wT[M_out][N_in] (the transpose of the reference's w[N_in][M_out]; weights
 * are static, the transpose is done once at load).  Evaluated in the reference's algebraically split
 * form (scripts/test_mm8/benchmark.py:167-179): xs = fp16(x*ry) through MFMA against the exact
 * uint8->binary16 weights, then y = rx*(acc + 0.5*sum xs) + sum x*my + mx*sum x.  act as above.
 * Replaces the tiled / WMMA variants scripts/test_mm8/rwkv_pip_operators.cu:205-558.  Any B: more than 256
 * rows (chunked prefill) are processed 256 at a time, each block re-streaming the weights.
 */
int64_t mm8t_workspace_bytes(int B, int N_in, int M_out, int splits);
int mm8t_seq(int B, int N_in, int M_out, const void *x, int x_stride, const void *wT, int64_t w_stride, int w_tiled,
             const void *mx, const void *rx, const void *my, const void *ry, void *y, int y_stride, int act,
             int splits, void *workspace, void *stream);
/* mm8t_seq with the arithmetic of the reference's mm8_seq -- kernel_mm_seq_fp16i8, scripts/test_mm8/rwkv_pip_operators.cu:59-83:
 * every product and every sum in binary32 -- at matrix-core speed: x*ry (two binary16 numbers: <= 22 significant bits, exact in
 * binary32) is split EXACTLY into hi + lo binary16 operands, multiplied in two passes with binary32 accumulation and reduced
 * with the rank-1 corrections.  Differs from the as-coded kernel only by the order of the binary32 sums (~1e-5 of the row
 * scale; mm8t_seq, which rounds xs to binary16 like the reference's mm8_seq_opt, rwkv_pip_wrapper.cpp:148-191: 2e-3).  Same
 * arguments as mm8t_seq; workspace: mm8t_exact_workspace_bytes. */
/* The dequantised matrix of mm8t_seq's weights as binary16 [M_out][N_in] row-major: out[m][k] = fp16(((q + 0.5) * rx[m]) * ry[k] +
 * mx[m] + my[k]) (the as-coded dequantisation, rwkv_pip_operators.cu:76-79, rounded once).  For products of more than 256 rows
 * (chunked prefill), which are MFMA-bound: one pass into a reused scratch, then a binary16 GEMM (the reference's own mm8_seq_opt
 * casts the uint8 matrix to binary16 in front of cuBLAS the same way, rwkv_pip_wrapper.cpp:163-176).  wT as in mm8t_seq. */
int mm8_dequant_f16(int M_out, int N_in, const void *wT, int64_t w_stride, int w_tiled, const void *rx, const void *mx, const void *ry,
                    const void *my, void *out, void *stream);
int64_t mm8t_exact_workspace_bytes(int B, int N_in, int M_out, int splits);
int mm8t_seq_exact(int B, int N_in, int M_out, const void *x, int x_stride, const void *wT, int64_t w_stride, int w_tiled,
                   const void *mx, const void *rx, const void *my, const void *ry, void *y, int y_stride, int act,
                   int splits, void *workspace, void *stream);

/*
 * Penalties + greedy token selection in one pass (rows decoded with temperature 0).
 * Replaces chirrup/worker.py:724-740 for those rows: occurrence *= decay; logits -= alpha_presence +
 * occurrence * frequency_penalty (binary32, rounded back into the binary16 logits in place);
 * ids[row] = argmax (chirrup/utils/samplers.py:195-221 with temperature 0; ties -> lowest id).
 * logits [B][V] binary16; occurrence, alpha_presence float [n_slots][V]; penalty_decay,
 * frequency_penalty binary16 [n_slots]; row b uses slot slot_idx[b] (or b).  occurrence == NULL:
 * arg-max only.  V % 8 == 0.
 */
int rwkv7_penalize_argmax(int B, int V, void *logits, float *occurrence, const float *alpha_presence,
                          const void *penalty_decay, const void *frequency_penalty, const int32_t *slot_idx,
                          int32_t *ids, void *stream);
/* The device-side consequences of sampling ids[row] for slot slot_idx[row] (row when slot_idx is NULL), row < n
 * (reference: chirrup/worker.py:527-535): last_ids[slot] = id; occurrence[slot][id] += penalty_weight[id];
 * alpha_presence[slot][id] = presence[slot * presence_stride].  occurrence / alpha_presence fp32 [n_slots][V],
 * penalty_weight fp32 [V], last_ids int32 [n_slots].  An id outside [0, V) only updates last_ids.
 * status_dst (may be NULL): *status_dst = status_src ? *status_src : 0 -- the sticky launch-status word of the time-mix launches
 * (rwkv7_tmix_gemms) placed behind the ids (status_dst = ids + n of a buffer of n + 1), so that the ONE device-to-host copy of a
 * step's ids also tells the host whether that step's launches gave up a bounded wait. */
int rwkv7_commit_sampled(int n, int V, const int32_t *ids, const int32_t *slot_idx, int32_t *last_ids, float *occurrence,
                         const float *penalty_weight, float *alpha_presence, const float *presence, int64_t presence_stride,
                         const int32_t *status_src, int32_t *status_dst, void *stream);
/* The penalty step without the dense pass over the tables (round 4).  A slot's occurrence / alpha_presence rows are zero except at
 * the ids it has sampled, so beside the (unchanged, dense, exact) tables every slot keeps the list of those ids: pen_list int32
 * [n_slots][cap], pen_count int32 [n_slots] (entries used; -1 = more than cap distinct ids: the slot takes the dense pass until it
 * is reset), pen_bits uint32 [n_slots][V / 32] (an id is listed once).  rwkv7_commit_sampled_listed = rwkv7_commit_sampled + the
 * list update; rwkv7_penalize_argmax_listed = rwkv7_penalize_argmax touching only the listed entries -- per element the same
 * arithmetic, which is the identity on entries that are zero: logits, occurrence and ids are bit-identical to the dense entry
 * points'.  Resetting a slot: zero its two table rows (or their listed entries), its pen_bits row and pen_count[slot].  The rows
 * of ONE commit launch must address distinct slots.  V % 32 == 0. */
int rwkv7_penalize_argmax_listed(int B, int V, void *logits, float *occurrence, const float *alpha_presence, const void *penalty_decay,
                                 const void *frequency_penalty, const int32_t *slot_idx, int32_t *ids, const int32_t *pen_list,
                                 const int32_t *pen_count, int cap, void *stream);
int rwkv7_commit_sampled_listed(int n, int V, const int32_t *ids, const int32_t *slot_idx, int32_t *last_ids, float *occurrence,
                                const float *penalty_weight, float *alpha_presence, const float *presence, int64_t presence_stride,
                                const int32_t *status_src, int32_t *status_dst, int32_t *pen_list, int32_t *pen_count,
                                uint32_t *pen_bits, int cap, void *stream);

/* The two ends of a decode step around the layers (Albatross/rwkv7.py:503-517 `_pre`: the embedding gather; :561-563 `_post`:
 * state[2] += T), each one launch.  rwkv7_embed_rows: x[b][t] = emb[token], token = tokens[b*T + t], or -- when that is negative and
 * `feedback` is given -- feedback[slot_idx[b]] (feedback[b] without slot_idx): the id the sampler left for that slot one step ago
 * (chirrup/worker.py:527: the reference round-trips it through the host).  emb binary16 [V][C], C % 8 == 0, x binary16 [B*T][C]; a
 * token outside [0, V) gives a zero row.  zero_words / n_zero: int32 words zeroed by the same launch (the caller's launch-sync words:
 * tile counters and time-mix hand-off words of the stream), or NULL / 0.  elapsed_rows (or NULL): elapsed_rows[b] = elapsed_pool[slot
 * of row b] -- the slot table's step counters in batch-row order, as the WKV7 entries take them.
 * rwkv7_advance_elapsed: elapsed[slot_idx[b]] += T for b < B (elapsed[b] without slot_idx; slots distinct). */
int rwkv7_embed_rows(int B, int T, int C, int V, const void *emb, const int64_t *tokens, const int32_t *slot_idx,
                     const int32_t *feedback, void *x, int32_t *zero_words, int n_zero, const int32_t *elapsed_pool,
                     int32_t *elapsed_rows, void *stream);
int rwkv7_advance_elapsed(int B, int T, const int32_t *slot_idx, int32_t *elapsed, void *stream);
/* dst[slot_idx[b]][0..C) = src[slot_idx[b]][0..C) for b < B: binary16 tables [n_slots][C] (C % 8 == 0, slots distinct) -- the commit of
 * a chunk's token-shift carry into the slot table (the T > 1 form of rwkv7_add_ln_mix writes the carry to a side table). */
int rwkv7_copy_slot_rows(int B, int C, const int32_t *slot_idx, const void *src, void *dst, void *stream);

/*
 * Sort-free top-p / top-k / temperature sampling of n_rows rows of `logits` (binary16 [B][V], V <= 65536,
 * V % 8 == 0), row list `rows` (indices into logits / ids).  Semantics of
 * sample_logits_rwkv_pip_compatible (chirrup/utils/samplers.py:171-255): softmax, drop everything below
 * the probability at which the descending cumulative sum reaches top_p, optional top-k, probs**(1/T), one
 * draw.  temperature / top_p binary16 and top_k int32 are per-slot tables addressed through slot_idx[row]
 * (or row); uniform[i] in [0,1) is the random number for rows[i]; ids[rows[i]] receives the token.
 * Ties at the top-k boundary are all kept; the draw is an inverse-CDF walk in token order.
 */
int rwkv7_sample_topp(int n_rows, int V, const void *logits, const int32_t *rows, const void *temperature,
                      const void *top_p, const int32_t *top_k, const int32_t *slot_idx, const float *uniform,
                      int32_t *ids, void *stream);

/*
 * The pieces of mm8t_seq for a decode step that folds the mm8 prologue / reduce launches into the neighbouring kernels
 * (chirrup_amd/rwkv7.py, ffn_dtype = int8): LN kernel (prologue of ffn.key, rwkv7_add_ln_mix_mm8) -> mm8t_gemm_partial ->
 * mm8_reduce_rows (corrections + relu^2 of ffn.key AND the prologue of ffn.value) -> mm8t_gemm_partial -> next LN kernel
 * (corrections of ffn.value).  Same arithmetic and rounding points as mm8t_seq.
 * mm8t_gemm_partial: core sums of xs [B][N_in] (B <= 256) against wT into partials [splits][B][M_out]; returns the split
 *   count used (> 0) or a negative error.
 * mm8_reduce_rows: y = act(rx*(sum partials - 1023.5*S[.][0]) + S[.][1] + mx*S[.][2]) (y may be NULL), and when xs2 != NULL the
 *   prologue of the next product: xs2 = fp16(y*ry2), S2 = {sum xs2, sum y*my2, sum y} per row.  Row sums travel as partial
 *   sums: S is [B][S_parts][3], S2 is written as [B][mm8_row_parts(M_out)][3] (one part per 1024 columns), to be added in
 *   part order by the consumer (rwkv7_add_ln_mix_mm8: in_S_parts; this function: S_parts).
 */
int mm8_row_parts(int M_out);
int mm8t_gemm_partial(int B, int N_in, int M_out, const void *xs, int xs_stride, const void *wT, int64_t w_stride,
                      int w_tiled, int splits, int row_halves, float *partials, void *stream);
int mm8_reduce_rows(int B, int M_out, int splits, const float *partials, const void *rx, const void *mx, const float *S,
                    int S_parts, int act, void *y, int y_stride, const void *ry2, const void *my2, void *xs2, float *S2,
                    void *stream);
/* mm8t_gemm_partial + mm8_reduce_rows in ONE launch (unsplit; for products with enough 128-column tiles to fill the chip,
 * e.g. ffn.key with row_halves): corrections, relu^2 (act = 1), y (may be NULL) and the next product's prologue (xs2 and S2
 * [B][mm8_tile_parts(M_out)][3]; may be NULL) come out of the GEMM's own epilogue.  S: [B][S_parts][3].  B <= 256,
 * M_out < 32768, M_out % 8 == 0, 16-byte aligned vectors.  Same element arithmetic as mm8_reduce_rows; the row sums S2 are split per 128-column tile. */
int mm8_tile_parts(int M_out);
int mm8t_gemm_fused(int B, int N_in, int M_out, const void *xs, int xs_stride, const void *wT, int64_t w_stride, int w_tiled,
                    const void *rx, const void *mx, const float *S, int S_parts, int act, void *y, int y_stride,
                    const void *ry2, const void *my2, void *xs2, float *S2, int row_halves, void *stream);
/* mm8t_gemm_fused for few rows: the product is split `splits` ways over K (2..4; 0 = the library's choice) and the last of a tile's
 * workgroups to finish adds the other slices' sums to its own, in slice order, and runs the same epilogue -- no partials for a
 * consumer and no mm8_reduce_rows launch.  partials: room for splits x B x M_out binary32 values (hand-off slabs); tile_counters: as
 * skinny_gemm_f16.  B <= 64 and (splits - 1) * B <= 96, else CHIRRUP_E_UNSUPPORTED (use mm8t_gemm_partial + mm8_reduce_rows).
 * Same sums as mm8_reduce_rows over mm8t_gemm_partial at the same split count, bit for bit; S2 has mm8_tile_parts(M_out) parts per row
 * (as mm8t_gemm_fused), not mm8_row_parts. */
int mm8t_gemm_fused_split(int B, int N_in, int M_out, const void *xs, int xs_stride, const void *wT, int64_t w_stride, int w_tiled,
                          const void *rx, const void *mx, const float *S, int S_parts, int act, void *y, int y_stride,
                          const void *ry2, const void *my2, void *xs2, float *S2, int splits, float *partials,
                          void *tile_counters, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CHIRRUP_AMD_H */
