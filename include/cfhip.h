/*
 * cfhip.h — C-ABI of libcfhip.so: the MI355X (gfx950 / CDNA4) kernels behind carefree-learn's
 * data-parallel training hot path (ViT building blocks).
 *
 * The reference (carefree-learn v0.5.0, /root/reference) is 100 % Python and has no FFI of its
 * own: all arithmetic on this path is delegated to PyTorch ATen ops.  The "reference interface"
 * each entry point replaces is therefore the ATen call site inside the reference module, cited
 * per function below as file:line relative to /root/reference.
 *
 * Conventions
 *   - plain C types only; every tensor is a raw device pointer + explicit sizes / leading dims
 *     (in ELEMENTS of the tensor's dtype); row-major, last dimension contiguous.
 *   - bf16 = 16-bit brain float stored as uint16_t; "f32" = IEEE float.
 *   - every call takes the HIP stream (hipStream_t as void*); nothing synchronises the device,
 *     nothing allocates or frees device memory, no pointer is retained after return: the caller
 *     (PyTorch caching allocator) owns every buffer, scratch is passed in as `workspace`.
 *   - return 0 on success, negative on error; cfhip_last_error() gives the thread-local message.
 *   - re-entrant: callable from the Python main thread and the autograd worker thread.
 */
#ifndef CFHIP_H
#define CFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFHIP_VERSION 100 /* 0.1.0 */

/* error codes */
#define CFHIP_OK 0
#define CFHIP_ERR_INVALID (-1)   /* bad argument / unsupported shape */
#define CFHIP_ERR_LAUNCH (-2)    /* HIP launch failure */
#define CFHIP_ERR_WORKSPACE (-3) /* workspace too small */

int cfhip_version(void);
const char* cfhip_last_error(void);
/* tuning knobs (process-wide, used by the benchmarks' A/B runs; defaults are the shipped behaviour):
 *   "gemm_config"     -1 (shape heuristic, default) or 0..16 to force one tile configuration
 *                     (0: 128x128x64, 1: 128x128x32, 3: 128x64x64, 7: 256x256x32 two-group kernel,
 *                      8: 256x128x32 two-group kernel, 13: 256x256x64, 14 / 15: 192x128x64 on eight / four waves, 16: 128x256x64;
 *                      see csrc/gemm.hip)
 *   "gemm_heuristic"  1..8, which shape -> configuration table pick_config() uses (default 8: every M >= 1024 forward and
 *                     dX GEMM on 192x128x64 — four waves for outputs up to 1 024 columns wide, eight otherwise —, single dW GEMMs
 *                     on 128x128x32; 7: eight waves everywhere; 6: the round-2 table — forward on the BK = 64 configurations, dX on
 *                     the 256x128x32 two-group kernel)
 *   "gemm_group_n"    tile walk order of a GEMM launch: n > 0 (default 8): outputs wider than n tile columns are walked in
 *                     groups of n columns, all rows of a group first (an XCD's resident workgroups then share n B panels
 *                     that stay in its L2); n < 0: row groups of -n panels, columns outer; 0: rows outer, every column inner
 *   "ln_bwd_fused"    1 (default): cfhip_layernorm_bwd asked for dx AND dgamma / dbeta runs the one-launch kernel
 *                     (D a multiple of 8 up to 1280; round 5: not only multiples of 256); 2: only for D a multiple of 256
 *                     (rounds 3-4); 0: the round-1 one-wave-per-row kernel
 *   "gemm_cfg_nt_wide" / "gemm_cfg_nt" / "gemm_cfg_nn" / "gemm_cfg_tn"   tile configuration of one class of M >= 1024 GEMMs
 *                     (forward with N >= 2560, other forward, dX, dW); -1 (default): the heuristic table
 *   "attn_persistent" bits: 1 dK / dV pass, 2 dQ pass, 4 forward of the head_dim-64 kernels run as persistent 16-wave
 *                     workgroups (one per CU walking (batch, head) pairs, the next head's operands streaming into a second
 *                     LDS buffer) when no mask / not causal and 128 < T <= 256 — the ViT shape; default 7; 0: one workgroup
 *                     per head.  Results are bit-identical either way.
 *   "attn_pers_ctas"  workgroups of a persistent attention launch (default 256 = one per CU; measured in the ViT step:
 *                     128 / 192 / 224 / 256 / 512 -> 17.80 / 17.71 / 17.59 / 17.60 / 17.78 ms)
 *   "grouped_variant" ring of cfhip_gemm_bf16_grouped_tn: 0 (default) 5 slots, DMA 3 K-steps ahead; 1: 4 slots, 2 ahead;
 *                     2: 5 slots, 2 ahead; 3 / 4: DMA placement variants of 0.  +16: bias gradients reduced by the first tile
 *                     column alone instead of shared by the tile row; +32: row-major tile order (both: A/B runs);
 *   "attn_two_tiles"  bits (default 511) selecting the long-sequence attention forms (csrc/attn.hip): 1 forward, 2 dQ pass (4: also head_dim
 *                     > 48), 8 dK / dV pass of head_dim <= 64 on the two-tiles-per-wave kernels; 16: forward row sums through a ones column
 *                     when head_dim = 8 mod 16; 32: dK / dV pass of head_dim 72 .. 96 on the two-tile kernel; 64 (round 6): the plain
 *                     head_dim <= 48 dQ pass with S / dP on 32x32x16 MFMAs; 128 (round 6): head_dim 40, plain: dP - delta out of the matrix
 *                     pipe (two spare reduction columns) in the dQ and dK / dV passes; 256 (round 6): head_dim 72 .. 96 without a mask: forward and dQ
 *                     pass on the two-tile kernels.  "attn_one_pass" (default 2; 0 = two passes, 1 = the 16-wave
 *                     kernel at every length, 2 = 128 < T <= 224 on 8 waves with two key tiles each, bit-identical): dQ, dK, dV of a short
 *                     self-attention from one evaluation of S and dP.  "attn_short_max": longest head_dim-64 sequence on the LDS-resident kernels
 *   "conv_form"       tile form of cfhip_conv3x3_nhwc_bf16: -1 (default) = by shape (128x160x64 on four waves where 160-column tiles cover
 *                     Cout with at most 1/8 of waste — the UNet's 320 / 640 / 960 / 1280 / 1920 / 2560 — and Cin % 64 == 0; otherwise
 *                     the round-3 table); -2: the round-3 table everywhere; 0: 256x128x32 two-group kernel, 1: 128x128x32, 2: 128x160x32, 3: 128x160x64
 *   "conv_split"      -1 (default): the K split cfhip_conv3x3_workspace sizes for; n >= 1 forces n (A/B runs; size the workspace
 *                     with the option already set)
 * Unknown names are an error.  (The phase-timing ablation masks "gemm_ablate" / "attn_ablate" of round 1 are
 * not part of this library any more: they exist only in the -DCFHIP_ABLATE build that tools/build_variant.sh
 * writes to tools/libcfhip_ablate.so, selected by the tools through CFHIP_LIB.) */
int cfhip_set_option(const char* name, int value);

/* ------------------------------------------------------------------------------------------
 * K1/K2  GEMM  (replaces F.linear at modules/core/customs.py:89, attentions.py:214,
 *              nn.Linear.forward via hijacks.py:42-57, and its autograd backward)
 *
 *   C[m][n] = epilogue( sum_k A(m,k) * B(n,k) )          bf16 operands, fp32 accumulate (MFMA)
 *
 *   a_trans = 0: A(m,k) = A[m*lda + k]      a_trans = 1: A(m,k) = A[k*lda + m]
 *   b_trans = 0: B(n,k) = B[n*ldb + k]      b_trans = 1: B(n,k) = B[k*ldb + n]
 *   (forward  y = x W^T      : a_trans 0, b_trans 0, A = x [M,K],  B = W  [N,K])
 *   (backward dx = dy W      : a_trans 0, b_trans 1, A = dy [M,Kd], B = W  [Kd,N])
 *   (backward dW = dy^T x    : a_trans 1, b_trans 1, A = dy [Kd,M], B = x  [Kd,N])
 *
 *   epilogue (bias is fp32 [N] or NULL in every mode):
 *     CFHIP_EPI_NONE        C = acc + bias
 *     CFHIP_EPI_GELU        aux_out (bf16 [M,ldc], may be NULL) = acc + bias ; C = gelu_erf(acc + bias)
 *     CFHIP_EPI_RESIDUAL    C = acc + bias + aux_in (bf16 [M,ldc]; f32 when out_dtype == 1)
 *     CFHIP_EPI_DGELU       C = acc * gelu_erf'(aux_in)   (aux_in = saved pre-activation, bf16)
 *     CFHIP_EPI_QGELU / CFHIP_EPI_DQGELU   the same two with quick GELU x * sigmoid(1.702 x)
 *                           (activations.py "quick_gelu": the CLIP towers, multimodal/clip.py)
 *   out_dtype: 0 = bf16, 1 = f32.   accumulate != 0 (f32 output only): C += result.
 *   split_k > 1: the K range is cut in `split_k` slices, partial tiles go to `workspace`
 *     (needs split_k*M*N*4 bytes) and a second kernel reduces them (epilogue NONE only).
 *   bias_grad (layout (1,1) only, else NULL): f32 [M], (+)= sum_k A(m,k) — for dW = dY^T X this is
 *     the bias gradient colsum(dY), produced by the same kernel with a ones-operand MFMA
 *     (workspace then needs split_k*M*4 more bytes).
 * ------------------------------------------------------------------------------------------ */
#define CFHIP_EPI_NONE 0
#define CFHIP_EPI_GELU 1
#define CFHIP_EPI_RESIDUAL 2
#define CFHIP_EPI_DGELU 3
#define CFHIP_EPI_QGELU 4
#define CFHIP_EPI_DQGELU 5

int cfhip_gemm_bf16(const void* A, const void* B, void* C, const float* bias, const void* aux_in,
                    void* aux_out, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                    int a_trans, int b_trans, int epilogue, int out_dtype, int accumulate,
                    int split_k, void* workspace, size_t workspace_bytes, float* bias_grad,
                    int bias_grad_accumulate, void* stream);

/* The kernel instantiation (as a profiler prints it, e.g. "gemm_bf16_kernel<false, true, 0, Cfg<192, 128, 2, 2, 2, 64>>") that
 * cfhip_gemm_bf16 launches for an aligned problem of this shape under the current options; `out` holds at least 96 bytes.
 * For evidence lines (bench.py groups its in-step timings by it); not needed to run anything. */
int cfhip_gemm_kernel_name(int M, int N, int K, int a_trans, int b_trans, int epilogue, char* out, size_t out_bytes);

/* Grouped weight gradients: dW_i[M_i][N_i] (+)= dY_i^T X_i and db_i[M_i] (+)= colsum(dY_i) for `count` Linear layers in ONE
 * launch (the parameter half of F.linear's autograd backward, customs.py:89 / attentions.py:214: grad_weight =
 * grad_output^T @ input, grad_bias = grad_output.sum(0), for every Linear of one or more blocks).  Layout (1,1) of
 * cfhip_gemm_bf16: A = dY [K][M] (lda), B = X [K][N] (ldb), both bf16 token-major; C f32 [M][N] (ldc).  256 x 256 output
 * tiles of all problems share the chip, every tile runs its whole reduction: no split-K, no workspace, no second pass,
 * deterministic.  M, N, lda, ldb multiples of 8, ldc of 4, 16-byte aligned bases, K*ld*2 < 2 GiB.  `problems` is a HOST
 * array (copied into the kernel arguments, 24 per launch).  bias_grad may be NULL per problem.  With bias gradients the
 * workgroups of a tile row share the column-sum reduction and meet in a small workspace the LIBRARY owns, one per stream
 * (<= 16 streams; allocated on the first such launch of a stream — before any hipGraph capture); partial sums are added in a
 * fixed order, results do not depend on arrival order. */
typedef struct cfhip_gemm_problem {
  const void* A;
  const void* B;
  void* C;
  float* bias_grad;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int accumulate;            /* C += */
  int bias_grad_accumulate;  /* bias_grad += */
} cfhip_gemm_problem;
int cfhip_gemm_bf16_grouped_tn(const cfhip_gemm_problem* problems, int count, void* stream);

/* column sums of a bf16 matrix: out[n] (f32) (+)= sum_m X[m*ldx + n]   (bias gradients)
 * workspace: >= cfhip_colsum_workspace(M, N) bytes. */
size_t cfhip_colsum_workspace(int M, int N);
int cfhip_colsum_bf16(const void* X, float* out, int M, int N, int64_t ldx, int accumulate,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K5  LayerNorm over the last dim (replaces nn.LayerNorm built by NormFactory("layer"),
 *     modules/core/norms.py:88-89,118-119; call sites mixed_stacks/api.py:141,155,397-402)
 *   biased variance, eps inside the sqrt, fp32 statistics; gamma / beta fp32; dtype_flags bit 0: x is f32 (else bf16) —
 *   the residual stream is kept in f32, as it is in the reference's autocast run (head_token / pos_encoding are f32
 *   parameters, so `x + f(LN(x))` type-promotes to f32); bit 1: y is f32 (else bf16; f32 rows only) — a LayerNorm whose
 *   output IS the residual stream (the `embedding_norm` in front of CLIP's vision blocks, cv/encoder/transformer.py:60-64:
 *   nn.LayerNorm on an f32 input returns f32 under autocast; round 6, found by the full-size CLIP parity test).
 *   D % 4 == 0 and D <= 2048 (a row lives in the registers of one wave).
 *   x_row_stride / y_row_stride in elements (lets the head LN read token 0 of every sample).
 *   fwd saves mean / rstd (f32 [M]) for bwd.
 *   bwd: dx (bf16) = LN'(dy) [+ dx_add (bf16) if not NULL]; dgamma / dbeta (f32 [D]) (+)= ...
 *        workspace >= cfhip_layernorm_bwd_workspace(M, D) bytes.  dx == NULL: parameter gradients
 *        only; dgamma == dbeta == NULL: input gradient only (two launches that can run on two
 *        streams: dx is the critical path of backward, the parameter gradients are not).
 * ------------------------------------------------------------------------------------------ */
int cfhip_layernorm_fwd(const void* x, int dtype_flags, const float* gamma, const float* beta, void* y,
                        float* mean, float* rstd, int M, int D, int64_t x_row_stride,
                        int64_t y_row_stride, float eps, void* stream);
size_t cfhip_layernorm_bwd_workspace(int M, int D);
int cfhip_layernorm_bwd(const void* dy, const void* x, int x_is_f32, const float* gamma, const float* mean,
                        const float* rstd, const void* dx_add, void* dx, float* dgamma, float* dbeta,
                        int M, int D, int64_t dy_row_stride, int64_t x_row_stride,
                        int64_t dx_row_stride, int accumulate_param_grads, void* workspace,
                        size_t workspace_bytes, void* stream);
/* cfhip_layernorm_bwd in two calls: `_partials` = the row kernel (dx (+ dx_add), per-workgroup partial sums of dgamma | dbeta into
 * `workspace`, *rows_out = number of partial rows), `_reduce` = their column sums (+)= into dgamma / dbeta on any stream the caller
 * orders behind it.  The input gradient is on the backward's critical chain, the parameter gradients are not. */
int cfhip_layernorm_bwd_partials(const void* dy, const void* x, int x_is_f32, const float* gamma, const float* mean,
                                 const float* rstd, const void* dx_add, void* dx, int M, int D, int64_t dy_row_stride,
                                 int64_t x_row_stride, int64_t dx_row_stride, void* workspace, size_t workspace_bytes,
                                 int* rows_out, void* stream);
int cfhip_layernorm_bwd_reduce(void* workspace, int rows, int D, float* dgamma, float* dbeta, int accumulate, void* stream);
/* cfhip_layernorm_bwd / _partials with the residual-gradient stream carried in TWO bf16 words per element (round 6): hi = bf16(g),
 * lo = bf16(g - hi), i.e. 16 mantissa bits.  The reference's residual stream is f32 under autocast (x + f(LN(x)) with f32 x), so the
 * gradient that flows along it is f32 and only the matrix products see it rounded to bf16; with one bf16 word the running sum is
 * rounded twice per block, and parameters whose per-sample gradients cancel (CLIP's head token / positional encoding under the
 * contrastive loss) came out 1.5 x further from fp32 than the reference's own bf16 run.  dx_add_lo / dx_lo: the second words of
 * dx_add / dx (same strides; either may be NULL); the GEMMs of the block keep reading the first word = bf16(g), as the reference's
 * do.  rows_out != NULL selects the `_partials` form (row kernel only; reduce with cfhip_layernorm_bwd_reduce). */
int cfhip_layernorm_bwd2(const void* dy, const void* x, int x_is_f32, const float* gamma, const float* mean,
                         const float* rstd, const void* dx_add, const void* dx_add_lo, void* dx, void* dx_lo,
                         float* dgamma, float* dbeta, int M, int D, int64_t dy_row_stride, int64_t x_row_stride,
                         int64_t dx_row_stride, int accumulate_param_grads, void* workspace, size_t workspace_bytes,
                         int* rows_out, void* stream);
/* The reference's 4-D `LN` (modules/core/norms.py:30-46: NormFactory("layer_norm") on [B, C, H, W]): y = (x - mean_b) / (std_b + eps) *
 * weight[c] + bias[c], ONE mean and one UNBIASED standard deviation per sample over all C*H*W elements, eps added to the standard
 * deviation.  x / y / dy / dx bf16 NCHW (contiguous); weight / bias f32 [C] or both NULL (elementwise_affine = False); mean / std f32 [B]
 * written by the forward, read by the backward.  dx == NULL: parameter gradients only; dweight == dbias == NULL: dx only.
 * workspace >= cfhip_layernorm4d_workspace(B, C, HW) bytes.  Fixed summation order: results are reproducible bit for bit. */
size_t cfhip_layernorm4d_workspace(int B, int C, int HW);
int cfhip_layernorm4d_fwd(const void* x, const float* weight, const float* bias, void* y, float* mean, float* std, int B, int C,
                          int HW, float eps, void* workspace, size_t workspace_bytes, void* stream);
int cfhip_layernorm4d_bwd(const void* dy, const void* x, const float* weight, const float* mean, const float* std, void* dx,
                          float* dweight, float* dbias, int accumulate_param_grads, int B, int C, int HW, float eps, void* workspace,
                          size_t workspace_bytes, void* stream);
/* the two ends of that stream: f32 -> (hi, lo) and (hi, lo) -> f32 = hi + lo (exact in f32) */
int cfhip_split_f32_bf16x2(const float* src, void* hi, void* lo, int64_t n, void* stream);
int cfhip_join_bf16x2_f32(const void* hi, const void* lo, float* dst, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * K3/K4  fused scaled-dot-product attention (replaces F.scaled_dot_product_attention reached
 *     through sdp_attn, toolkit.py:953-963, from Attention.forward, attentions.py:254, and the
 *     head split / merge copies at attentions.py:180-185,270-275)
 *
 *   o[b,t,h,:] = softmax_j( q[b,t,h,:] . k[b,j,h,:] * scale  [+ mask] ) v[b,j,h,:]
 *
 *   q, k, v, o, dq, dk, dv, do: bf16, addressed as  ptr[b*stride_b + t*stride_t + h*DH + d]
 *   (so the packed [B,T,3,H,DH] projection output is consumed in place and o is written
 *   directly as [B,T,H*DH]).  head_dim DH must be 64.  Tq, Tk <= CFHIP_ATTN_MAX_T: the whole K/V of
 *   one head is LDS-resident (one workgroup per (b, h)); longer sequences: the same kernels with an
 *   outer loop over 256-row chunks (online softmax across chunks in the forward, one workgroup per
 *   128 query / key rows).  lse: f32 [B,H,Tq] (natural-log sum-exp of the scaled scores).
 *   mask: optional uint8 "keep" mask (1 = attend), addressed mask[b*ms_b + h*ms_h + i*ms_q + j]
 *   (strides may be 0 for broadcasting), or NULL.  causal != 0 additionally masks j > i.
 *   bwd needs delta: f32 [B,H,Tq] scratch (rowsum(do * o), computed inside).
 * ------------------------------------------------------------------------------------------ */
#define CFHIP_ATTN_MAX_T 256
#define CFHIP_ATTN_HEAD_DIM 64

int cfhip_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                   const uint8_t* mask, int B, int H, int Tq, int Tk, int64_t q_stride_b,
                   int64_t q_stride_t, int64_t kv_stride_b, int64_t kv_stride_t, int64_t o_stride_b,
                   int64_t o_stride_t, int64_t ms_b, int64_t ms_h, int64_t ms_q, float scale,
                   int causal, void* stream);

int cfhip_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                   const float* lse, float* delta, const uint8_t* mask, void* dq, void* dk, void* dv,
                   int B, int H, int Tq, int Tk, int64_t q_stride_b, int64_t q_stride_t,
                   int64_t kv_stride_b, int64_t kv_stride_t, int64_t o_stride_b, int64_t o_stride_t,
                   int64_t ms_b, int64_t ms_h, int64_t ms_q, float scale, int causal, int parts,
                   void* stream);
/* Attention with RETURNED weights — the reference's slow path `Attention.forward(require_weights=True)` / `customize_sdp`
 * (attentions.py:256-268: raw = q k^T / scaling, masked_fill(-inf), softmax, weights @ v).  The output comes from
 * cfhip_attn_fwd_dh with scale = 1 / scaling; cfhip_attn_probs materialises what the fused kernels never store:
 *     probs[b][h][i][j] (f32) = keep(b,h,i,j) ? exp(scale * q_i . k_j - lse[b][h][i]) : 0
 * and cfhip_attn_probs_bwd turns a gradient on those weights into its dq / dk contribution through the softmax
 *     t_i = sum_j dP_ij P_ij;  dS_ij = scale P_ij (dP_ij - t_i);  dq_i = sum_j dS_ij k_j;  dk_j = sum_i dS_ij q_i
 * (written as bf16 [B][T][H*head_dim] contiguous; the caller adds them to cfhip_attn_bwd_dh's; dv gets nothing).
 * ds_workspace: f32 [B][H][Tq][Tk].  fp32 VALU arithmetic, no atomics (deterministic); an inspection / auxiliary-loss path,
 * not a throughput path.  Addressing, mask and causal conventions as cfhip_attn_fwd_dh; head_dim a multiple of 8 up to 192. */
int cfhip_attn_probs(const void* q, const void* k, const float* lse, const uint8_t* mask, float* probs, int B, int H,
                     int Tq, int Tk, int head_dim, int64_t q_stride_b, int64_t q_stride_t, int64_t k_stride_b,
                     int64_t k_stride_t, int64_t ms_b, int64_t ms_h, int64_t ms_q, float scale, int causal, void* stream);
int cfhip_attn_probs_bwd(const void* q, const void* k, const float* lse, const uint8_t* mask, const float* d_probs,
                         float* ds_workspace, void* dq, void* dk, int B, int H, int Tq, int Tk, int head_dim,
                         int64_t q_stride_b, int64_t q_stride_t, int64_t k_stride_b, int64_t k_stride_t, int64_t ms_b,
                         int64_t ms_h, int64_t ms_q, float scale, int causal, void* stream);

/* The same with an explicit head_dim (any multiple of 8 up to 192, e.g. the 40 / 80 / 160-channel heads of the UNet's
 * SpatialTransformer, mixed_stacks/api.py:766-893; attentions.py:498-569): always the chunked general kernels unless
 * head_dim == 64 and both lengths fit the resident form.  Addressing ptr[b*stride_b + t*stride_t + h*head_dim + d]. */
int cfhip_attn_fwd_dh(const void* q, const void* k, const void* v, void* o, float* lse,
                      const uint8_t* mask, int B, int H, int Tq, int Tk, int head_dim, int64_t q_stride_b,
                      int64_t q_stride_t, int64_t kv_stride_b, int64_t kv_stride_t, int64_t o_stride_b,
                      int64_t o_stride_t, int64_t ms_b, int64_t ms_h, int64_t ms_q, float scale, int causal,
                      void* stream);
int cfhip_attn_bwd_dh(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                      const float* lse, float* delta, const uint8_t* mask, void* dq, void* dk, void* dv,
                      int B, int H, int Tq, int Tk, int head_dim, int64_t q_stride_b, int64_t q_stride_t,
                      int64_t kv_stride_b, int64_t kv_stride_t, int64_t o_stride_b, int64_t o_stride_t,
                      int64_t ms_b, int64_t ms_h, int64_t ms_q, float scale, int causal, int parts,
                      void* stream);
/* The same with DROPOUT ON THE ATTENTION PROBABILITIES (reference attentions.py:254 -> sdp_attn(..., dropout) ->
 * F.scaled_dot_product_attention(dropout_p); attentions.py:264-265 in the weights path): O = dropout(softmax(S)) V.
 * The keep mask is a pure function of (seed, offset, b, h, i, j) — one Philox4x32-10 call per 4 x 4 block of the score
 * matrix, counter = offset + ((b H + h) ceil(Tq / 4) + i / 4) ceil(Tk / 4) + j / 4, byte (j % 4) of word (i % 4) >=
 * round(256 p) keeps — so the two backward passes regenerate it from the pair the forward used (advance `offset` by
 * B H ceil(Tq / 4) ceil(Tk / 4) per call) and no mask is ever stored.  The probability is quantised to 1/256 (kept
 * values scale by 256 / (256 - round(256 p))).  Any head_dim / length: these run the chunked kernels.
 * cfhip_attn_dropout_mask writes that mask (uint8 [B][H][Tq][Tk], 1 = keep) for tests and debugging. */
int cfhip_attn_fwd_dropout(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* mask,
                           int B, int H, int Tq, int Tk, int head_dim, int64_t q_stride_b, int64_t q_stride_t,
                           int64_t kv_stride_b, int64_t kv_stride_t, int64_t o_stride_b, int64_t o_stride_t,
                           int64_t ms_b, int64_t ms_h, int64_t ms_q, float scale, int causal, float dropout_p,
                           uint64_t seed, uint64_t offset, void* stream);
int cfhip_attn_bwd_dropout(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                           const float* lse, float* delta, const uint8_t* mask, void* dq, void* dk, void* dv, int B,
                           int H, int Tq, int Tk, int head_dim, int64_t q_stride_b, int64_t q_stride_t,
                           int64_t kv_stride_b, int64_t kv_stride_t, int64_t o_stride_b, int64_t o_stride_t,
                           int64_t ms_b, int64_t ms_h, int64_t ms_q, float scale, int causal, int parts,
                           float dropout_p, uint64_t seed, uint64_t offset, void* stream);
int cfhip_attn_dropout_mask(void* mask_out, int B, int H, int Tq, int Tk, float dropout_p, uint64_t seed,
                            uint64_t offset, void* stream);
/* parts: 1 = dQ pass only, 2 = dK/dV pass only, 3 = both.  The two passes are independent kernels
 * (each recomputes S, dP and delta) and may be launched on two streams. */

/* ------------------------------------------------------------------------------------------
 * K8 (ViT patch embedding) + K7 glue
 *   im2row: img f32/bf16 [B,C,Hh,Ww] -> rows bf16 [B*gh*gw, C*P*P] in (c, ph, pw) order, so the
 *     stride==kernel Conv2d of VanillaPatchEmbed (high_level.py:172-188) becomes one K1 GEMM with
 *     the conv weight viewed as [out, C*P*P].
 *   assemble_tokens: x0[b,0,:] = head_token + pos[0]; x0[b,1+p,:] = patches[b,p,:] + pos[1+p]
 *     (mixed_stacks/api.py:419-438,209-228).  bwd: dpatches (bf16), dhead_token, dpos (f32, (+)=).
 * ------------------------------------------------------------------------------------------ */
int cfhip_im2row(const void* img, int img_is_bf16, void* rows, int B, int C, int Hh, int Ww, int P,
                 void* stream);
int cfhip_assemble_tokens_fwd(const void* patches, const float* head_token, const float* pos,
                              void* x0, int x0_is_f32, int B, int Np, int D, void* stream);
int cfhip_assemble_tokens_bwd(const void* dx0, void* dpatches, float* dhead_token, float* dpos,
                              int B, int Np, int D, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * element-wise helpers (K6 stand-alone forms, casts)
 * ------------------------------------------------------------------------------------------ */
int cfhip_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int cfhip_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);
int cfhip_gelu_fwd(const void* x, void* y, int64_t n, void* stream);              /* bf16 */
int cfhip_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream);
int cfhip_quick_gelu_fwd(const void* x, void* y, int64_t n, void* stream);        /* x * sigmoid(1.702 x) */
int cfhip_quick_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream);
/* DDPM forward process and objective (multimodal/diffusion/samplers/schema.py:90-112; models/cv/diffusion.py:44-94):
 *   q_sample: out[b][:] = sqrt_ac[t_b] * x[b][:] + sqrt_1mac[t_b] * noise[b][:]  (x, noise f32; t int64 [B]; the
 *     tables f32 [T]; f32 output bit-exact with the reference expression, or bf16)
 *   mse_loss: loss_sum (f32[1], caller zeroes) += sum_b mean_inner (pred - target)^2 ; dpred (bf16, may be NULL) =
 *     grad_scale * 2 (pred - target) / inner   (pred bf16 = the UNet output, target f32 = the noise) */
int cfhip_q_sample(const float* x, const float* noise, const int64_t* t, const float* sqrt_ac,
                   const float* sqrt_1mac, void* out, int out_is_f32, int64_t B, int64_t inner, void* stream);
int cfhip_mse_loss(const void* pred, const float* target, float* loss_sum, void* dpred, int64_t B, int64_t inner,
                   float grad_scale, void* stream);
/* DDPMStep.loss_fn (models/cv/diffusion.py:44-94) for every objective it has: per_sample[b] (f32, must be zero on entry)
 * += mean_inner f(pred - target), f = square (loss_type 0, "l2") or abs (1, "l1"); dpred (bf16, may be NULL) =
 * weight[b] * f'(pred - target) / inner.  weight f32 [B] carries everything the reference multiplies a sample's loss by
 * (l_simple_weight / exp(log_var[t]) + original_elbo_weight * lvlb_weights[t], over the batch size). */
int cfhip_diffusion_loss(const void* pred, const float* target, const float* weight, float* per_sample, void* dpred,
                         int64_t B, int64_t inner, int loss_type, void* stream);
/* dst[b * dst_bs + i] = src[b * src_bs + i] (bf16, i < n, everything a multiple of 4 elements): channel concat /
 * split of NCHW tensors — torch.cat(dim=1) of the UNet skip connections (multimodal/diffusion/unet.py:311-316) */
int cfhip_copy_strided_bf16(const void* src, void* dst, int64_t batch, int64_t n, int64_t src_batch_stride,
                            int64_t dst_batch_stride, void* stream);
/* two of them (same batch count) in one launch: both halves of torch.cat([a, b], dim=1) or of its backward split (unet.py:311-316) */
int cfhip_copy_strided2_bf16(const void* src_a, void* dst_a, int64_t n_a, int64_t src_a_batch_stride, int64_t dst_a_batch_stride,
                             const void* src_b, void* dst_b, int64_t n_b, int64_t src_b_batch_stride, int64_t dst_b_batch_stride,
                             int64_t batch, void* stream);
/* GEGLU (activations.py:150-158): out[m][c] = vg[m][c] * gelu_erf(vg[m][L + c]) for vg bf16 [M][2L]; bwd writes dvg */
int cfhip_geglu_fwd(const void* vg, void* out, int64_t M, int L, void* stream);
int cfhip_geglu_bwd(const void* dy, const void* vg, void* dvg, int64_t M, int L, void* stream);
int cfhip_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);
/* bf16 [R,C] -> bf16 [C,R] */
int cfhip_transpose_bf16(const void* src, void* dst, int R, int C, int64_t ld_src, int64_t ld_dst,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * K14  fused Adam / AdamW over a flat fp32 parameter arena (next-row §8f; replaces
 *     torch.optim.Adam(W).step() selected by optimizers.py:29-33)
 *   p, g, m, v: f32 [n].  p_bf16 (may be NULL): refreshed bf16 copy of the parameters for the
 *   next forward.  grad_scale multiplies g first (1/world_size for DDP-mean, clip coefficient).
 *   step >= 1 (bias correction).  decoupled != 0 -> AdamW.
 * ------------------------------------------------------------------------------------------ */
int cfhip_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int decoupled, int step,
                    float grad_scale, void* stream);
/* same update, step-dependent scalars read from device memory so the launch can be replayed from a
 * captured hipGraph: hyper[0..7] = lr, beta1, beta2, eps, weight_decay, 1-beta1^t, 1/sqrt(1-beta2^t),
 * grad_scale (f32, device). */
int cfhip_adam_step_dev(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n,
                        const float* hyper, int decoupled, void* stream);
/* EMA of the parameters (modules/common.py:126-137): ema = (1 - decay) * p + decay * ema over a flat f32 buffer
 * (16-byte aligned); two rounded products + one rounded sum like the reference's expression: bit-exact.
 * the reference clones every parameter every step. */
int cfhip_ema_update(float* ema, const float* p, int64_t n, float one_minus_decay, float decay, void* stream);
/* Implicit-GEMM 3x3 / stride 1 / padding 1 convolution on NHWC bf16 activations (F.conv2d reached from Conv2d.forward,
 * reference convs/basic.py:160-177, as used 61 times by the DDPM UNet, unet.py:76-322): Y[p][co] = bias[co] +
 * sum_{ky,kx,c} X[p + (ky-1)*W + (kx-1)][c] * Wk[co][(ky*3+kx)*Cin + c] over the pixels p of B images of H x W, taps
 * outside the image contribute zero.  X [B*H*W][Cin], Wk [Cout][9*Cin] (tap-major, channel-minor), Y [B*H*W][Cout],
 * all bf16; bias f32 [Cout] or NULL.  Cin % 32 == 0, Cout % 8 == 0.  No im2row matrix is materialised: each K-step of
 * the MFMA GEMM gathers 32 channels of one tap straight from X.  The input gradient is the same call with dY in place
 * of X and the filters rotated by 180 degrees and transposed: Wk'[c][(ky*3+kx)*Cout + co] = W[co][c][2-ky][2-kx].
 * Shapes with few output tiles split the reduction (deterministic ordered reduce of fp32 partials): pass
 * cfhip_conv3x3_workspace() bytes of scratch (0 when the shape does not split). */
size_t cfhip_conv3x3_workspace(int B, int H, int W, int Cin, int Cout);
int cfhip_conv3x3_nhwc_bf16(const void* X, const void* Wk, const float* bias, void* Y, int B, int H, int W, int Cin,
                            int Cout, void* workspace, size_t workspace_bytes, void* stream);
/* Weight gradient of the same convolution without an im2row matrix: dW[co][c][ky][kx] (+)= sum_p dY[p][co] *
 * X[p + (ky-1)*W + (kx-1)][c] (taps outside the image contribute zero), f32 in the reference's [Cout][Cin][3][3] layout
 * (`accumulate` != 0: added to dW).  dY [B*H*W][Cout], X [B*H*W][Cin] bf16 NHWC; Cin % 8 == 0, Cout % 8 == 0, H, W >= 2,
 * B*H*W*max(H, W) < 2^32.  bias_grad (f32 [Cout] or NULL) (+)= colsum(dY).  The GEMM writes fp32 partial sums per K slice
 * ([split_k][Cout][9*Cin], tap-major) into the workspace (cfhip_conv3x3_wgrad_workspace() bytes, needed for every
 * split_k >= 1); a second pass adds them in a fixed order and writes the filter layout: deterministic. */
size_t cfhip_conv3x3_wgrad_workspace(int Cin, int Cout, int split_k);
int cfhip_conv3x3_wgrad_nhwc_bf16(const void* dY, const void* X, float* dW, int accumulate, float* bias_grad,
                                  int bias_grad_accumulate, int B, int H, int W, int Cin, int Cout, int split_k,
                                  void* workspace, size_t workspace_bytes, void* stream);
/* Filter repacking for the two calls above, w bf16 [Cout][Cin][3][3] (Cin % 8 == 0):
 *   rotate == 0: out[co][(ky*3+kx)*Cin + c]        = w[co][c][ky][kx]       ([Cout][9*Cin], the forward's Wk)
 *   rotate != 0: out[c][(ky*3+kx)*Cout + co]       = w[co][c][2-ky][2-kx]   ([Cin][9*Cout], the input gradient's Wk') */
int cfhip_conv3x3_pack_filters(const void* w, void* out, int Cout, int Cin, int rotate, void* stream);
/* The same for 1 .. 64 filter banks in one launch: `table` is a HOST array of 5 int64 per problem {w, out, Cout, Cin, rotate} (the pointers
 * travel in the kernel arguments).  What a model calls once at the top of its forward instead of two packs per convolution
 * (functional.prepack_convs; reference: every `conv_nd(2, ..., 3, padding=1)` of unet.py / residual.py reaches F.conv2d with its own weight). */
int cfhip_conv3x3_pack_filters_grouped(const int64_t* table, int count, void* stream);
/* One idle wavefront for `microseconds` (1..100000) on `stream`.  Host-side stream self-check only (two streams
 * that share a ROCclr hardware queue run it back to back; the side streams of the backward pass and the RCCL
 * stream must not share the compute stream's queue -- reference counterpart: none, torch DDP owns its streams). */
int cfhip_spin(int microseconds, void* stream);
/* sum of squares of g (f32 [n]) into out[0] (+= ; caller zeroes) — gradient-norm clipping */
int cfhip_sumsq_f32(const float* g, float* out, int64_t n, void* stream);

/* softmax cross-entropy on f32 logits [B,C] with int64 labels [B]: loss_sum (f32[1], caller
 * zeroes) += sum_b CE_b ; dlogits (f32 [B,C]) = (softmax - onehot) * grad_scale.
 * (losses/basic.py:126-141; integer label gather is exact.) */
int cfhip_softmax_xent(const float* logits, const int64_t* labels, float* loss_sum, float* dlogits,
                       int B, int C, float grad_scale, void* stream);
/* focal loss (losses/basic.py:170-206, the examples' default): p = softmax(z) + eps,
 * L_b = -log(p_y) (1 - p_y)^gamma; loss_sum += sum_b L_b; dlogits = grad_scale * dL/dz. */
int cfhip_softmax_focal(const float* logits, const int64_t* labels, float* loss_sum, float* dlogits,
                        int B, int C, float gamma, float eps, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * K8 general form: Conv2d (groups 1) = im2row + K1 GEMM (convs/basic.py:104-177 -> F.conv2d)
 *   conv_im2row: x f32/bf16 [B,C,H,W] -> rows bf16 [B*Ho*Wo][Kp], k = (c, ky, kx), zero padding,
 *     columns K..Kp-1 zero (Kp = K rounded up to a multiple of 8: the GEMM operand rule), so
 *     y_rows = rows x W[Cout][Kp]^T (+bias) is one cfhip_gemm_bf16 call and dW the (tn) GEMM of
 *     the same rows;  Ho = (H + 2 pad - dil (kh - 1) - 1) / stride + 1.
 *   conv_row2im: dx bf16 [B,C,H,W] = gather-sum of drows bf16 [B*Ho*Wo][Kp] (= dY_rows x W): every
 *     input pixel sums its <= kh*kw contributions — no atomics, deterministic.
 *   transpose_batched: src f32/bf16 [batch][R][C] -> dst bf16 [batch][C][R]: NCHW <-> token-major.
 * K9/K10: BatchNorm over (B, inner) per channel of x [B][C][inner] (norms.py:20-27,90-93 ->
 *   nn.BatchNorm1d/2d): training = batch statistics (biased variance), running statistics updated
 *   with `momentum` and the UNBIASED variance; eval = running statistics.  y bf16, mean/rstd f32 [C]
 *   saved for backward; bwd: dgamma/dbeta f32 [C] ((+)= with accumulate), dx bf16 (NULL to skip).
 * LeakyReLU (slope 0 = ReLU; activations.py "leaky_relu_0.2", "ReLU"), AdaptiveAvgPool2d((1,1))
 *   (cv/encoder/vanilla.py:144): x [BC][inner] -> y [BC].
 * ------------------------------------------------------------------------------------------ */
int cfhip_conv_im2row(const void* x, int x_is_f32, void* rows, int B, int C, int H, int W, int kh,
                      int kw, int stride, int pad, int dil, int Kp, void* stream);
int cfhip_conv_row2im(const void* drows, void* dx, int B, int C, int H, int W, int kh, int kw,
                      int stride, int pad, int dil, int Kp, void* stream);
int cfhip_transpose_batched(const void* src, int src_is_f32, void* dst, int batch, int R, int C,
                            void* stream);
int cfhip_batchnorm_fwd(const void* x, int x_is_f32, const float* gamma, const float* beta, void* y,
                        float* mean, float* rstd, float* running_mean, float* running_var, int B, int C,
                        int inner, float eps, float momentum, int training, void* stream);
int cfhip_batchnorm_bwd(const void* dy, const void* x, int x_is_f32, const float* gamma,
                        const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                        int B, int C, int inner, int accumulate, int training, void* stream);
int cfhip_leaky_relu_fwd(const void* x, void* y, int64_t n, float slope, void* stream);
int cfhip_leaky_relu_bwd(const void* dy, const void* x, void* dx, int64_t n, float slope, void* stream);
int cfhip_avgpool_fwd(const void* x, void* y, int64_t BC, int inner, void* stream);
int cfhip_avgpool_bwd(const void* dy, void* dx, int64_t BC, int inner, void* stream);

/* ------------------------------------------------------------------------------------------
 * UNet residual-block pieces (convs/residual.py:86-253, multimodal/diffusion/unet.py:52-74)
 *   groupnorm: y = [SiLU](GroupNorm_G(x + add[b][c])) over x f32/bf16 [B][C][inner], statistics per
 *     (b, group) in fp32 (mean / rstd f32 [B*G] saved); `add` (f32 [B][C], may be NULL) is the
 *     time-embedding term `Linear(SiLU(t))[:, :, None, None]` the reference adds in front of norm2.
 *     bwd: dx bf16, per-batch partial dgamma / dbeta f32 [B][C] (sum them over B with
 *     cfhip_colsum / a column reduce), dadd f32 [B][C] = sum_inner dx (NULL when there was no add).
 *   silu_f32: the activation on the [B, 1280] time embedding;  upsample2 / avgpool2: nearest x2
 *     (F.interpolate(scale_factor=2, "nearest")) and 2x2 average pooling of bf16 [BC][H][W]
 *     (H, W = the SMALL size in both);  timestep_embedding: [cos(t f_i) | sin(t f_i)], f32.
 * ------------------------------------------------------------------------------------------ */
int cfhip_groupnorm_fwd(const void* x, int x_is_f32, const float* add, const float* gamma, const float* beta,
                        void* y, float* mean, float* rstd, int B, int C, int G, int inner, float eps,
                        int silu, void* stream);
int cfhip_groupnorm_bwd(const void* dy, const void* x, int x_is_f32, const float* add, const float* gamma,
                        const float* beta, const float* mean, const float* rstd, void* dx,
                        float* dgamma_part, float* dbeta_part, float* dadd, int B, int C, int G, int inner,
                        int silu, void* stream);
/* the same two kernels with ONE AFFINE PER SAMPLE: gamma / beta are f32 [B][C] when affine_batch_stride == C (0: the
 * plain [C] form above).  `norm2(net) * (1 + scale) + shift` of the scale-shift residual block (convs/residual.py:
 * 236-239) is GroupNorm with gamma_eff[b][c] = gamma[c] (1 + scale[b][c]), beta_eff[b][c] = beta[c] (1 + scale[b][c]) +
 * shift[b][c]; bwd then leaves d gamma_eff / d beta_eff in dgamma_part / dbeta_part [B][C] (no reduction over B). */
int cfhip_groupnorm_affine_fwd(const void* x, int x_is_f32, const float* add, const float* gamma, const float* beta,
                               void* y, float* mean, float* rstd, int B, int C, int G, int inner, float eps,
                               int silu, int affine_batch_stride, void* stream);
int cfhip_groupnorm_affine_bwd(const void* dy, const void* x, int x_is_f32, const float* add, const float* gamma,
                               const float* beta, const float* mean, const float* rstd, void* dx,
                               float* dgamma_part, float* dbeta_part, float* dadd, int B, int C, int G, int inner,
                               int silu, int affine_batch_stride, void* stream);
/* the same arithmetic with every (sample, group) cut into `splits` (1 .. 64, <= inner / 8) slices along `inner`, one
 * workgroup each — for few samples (B * G workgroups do not fill 256 CUs: the batch-1 256^2 UNet step).  inner % 8 == 0,
 * 16-byte aligned tensors, <= 128 channels per group.  Statistics: per-slice (mean, M2) merged in slice order (Chan's
 * update), so results do not depend on scheduling.  workspace: f32, B * G * splits * 2 (fwd) / B * G * splits * 3 * (C / G)
 * (bwd) elements, owned by the caller. */
int cfhip_groupnorm_split_fwd(const void* x, int x_is_f32, const float* add, const float* gamma, const float* beta,
                              void* y, float* mean, float* rstd, int B, int C, int G, int inner, float eps, int silu,
                              int affine_batch_stride, int splits, float* workspace, void* stream);
int cfhip_groupnorm_split_bwd(const void* dy, const void* x, int x_is_f32, const float* add, const float* gamma,
                              const float* beta, const float* mean, const float* rstd, void* dx, float* dgamma_part,
                              float* dbeta_part, float* dadd, int B, int C, int G, int inner, int silu,
                              int affine_batch_stride, int splits, float* workspace, void* stream);
int cfhip_silu_f32_fwd(const float* x, float* y, int64_t n, void* stream);
int cfhip_silu_f32_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream);
/* The per-block time-embedding projections of the UNet's residual blocks, every block in ONE launch (replaces `self.time_embedding`
 * = Sequential(SiLU, Linear) of each ResidualBlockWithTimeEmbedding: reference convs/residual.py:185-191,226-239, unet.py:154-190).
 *   out_i[b, :] = Linear_i(SiLU(emb[b, :])), i < count <= 32, emb f32 [B, K] (K % 32 == 0, K <= 2048) shared by all problems; bf16 SiLU(emb)
 *   and bf16 weights, f32 accumulation and output (what the per-block GEMMs computed).  `table`: a HOST array of 4 int64 per problem —
 *   forward {weight bf16 [N, K] (16-byte aligned), bias f32 [N] or 0, out f32 [B, N], N}; backward {weight, dY f32 [B, N] or 0 (no gradient
 *   for that output), bf16 copy of dY [B, N] to write or 0, N}.  t_bf16: SiLU(emb) as bf16 [B, K] (out; the weight-gradient GEMMs' operand).
 *   backward: d_emb[b, k] = SiLU'(emb) * sum_i sum_n bf16(dY_i[b, n]) W_i[n, k]: per-64-column partial sums in `partial` (f32,
 *   (sum_i ceil(N_i / 64)) * ceil8(B) * K elements, caller-owned), summed in block order: deterministic, no atomics. */
int cfhip_time_proj_fwd(const float* emb, int B, int K, const int64_t* table, int count, void* t_bf16, void* stream);
int cfhip_time_proj_bwd(const float* emb, int B, int K, const int64_t* table, int count, float* partial, float* d_emb, void* stream);
int cfhip_upsample2_fwd(const void* x, void* y, int64_t BC, int H, int W, void* stream);
int cfhip_upsample2_bwd(const void* dy, void* dx, int64_t BC, int H, int W, void* stream);
int cfhip_avgpool2_fwd(const void* x, void* y, int64_t BC, int Ho, int Wo, void* stream);
int cfhip_avgpool2_bwd(const void* dy, void* dx, int64_t BC, int Ho, int Wo, void* stream);
/* NHWC forms (round 5): the UNet keeps its activations as the rows the implicit-GEMM convolutions read and write — bf16 [B][H * W][C],
 * C % 8 == 0 — instead of hopping NCHW <-> NHWC around every convolution.
 *   groupnorm_nhwc: nn.GroupNorm (+ the per-(b, c) additive term in front, + SiLU behind) of convs/residual.py:194,226-247 on such rows;
 *     `splits` slices of rows per sample (one workgroup each), workspace = cfhip_groupnorm_nhwc_workspace(...) bytes;
 *     mean / rstd f32 [B * G]; bwd: dgamma_part / dbeta_part f32 [B][C] (per-sample sums: the caller reduces over B), dadd f32 [B][C]
 *     (with `add`); gamma / beta [C], or [B][C] with affine_batch_stride == C (the scale-shift norm); deterministic, no atomics.
 *   upsample2_nhwc: F.interpolate(scale_factor=2, mode="nearest") (residual.py:147) and its gradient; H, W = the small size. */
size_t cfhip_groupnorm_nhwc_workspace(int B, int C, int G, int splits, int backward, int with_add);
int cfhip_groupnorm_nhwc_fwd(const void* x, const float* add, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                             int B, int C, int G, int inner, float eps, int silu, int affine_batch_stride, int splits, float* workspace,
                             void* stream);
int cfhip_groupnorm_nhwc_bwd(const void* dy, const void* x, const float* add, const float* gamma, const float* beta, const float* mean,
                             const float* rstd, void* dx, float* dgamma_part, float* dbeta_part, float* dadd, int B, int C, int G, int inner,
                             int silu, int affine_batch_stride, int splits, float* workspace, void* stream);
int cfhip_upsample2_nhwc_fwd(const void* x, void* y, int64_t B, int H, int W, int C, void* stream);
int cfhip_upsample2_nhwc_bwd(const void* dy, void* dx, int64_t B, int H, int W, int C, void* stream);
/* nn.ReflectionPad2d in front of F.conv2d (reference convs/basic.py:61-75,114-115: Conv2d(padding="reflection[N]")):
 *   fwd: x [BC][H][W] (bf16, or f32 when x_is_f32) -> y bf16 [BC][H + pt + pb][W + pl + pr], mirrored without repeating the border;
 *   bwd: dx bf16 [BC][H][W] = gather of the <= 9 positions of dy that mirror onto each input pixel (deterministic).
 * Every pad is >= 0 and smaller than the extent it mirrors (torch's rule). */
int cfhip_reflect_pad2d_fwd(const void* x, int x_is_f32, void* y, int64_t BC, int H, int W, int pl, int pr, int pt, int pb,
                            void* stream);
int cfhip_reflect_pad2d_bwd(const void* dy, void* dx, int64_t BC, int H, int W, int pl, int pr, int pt, int pb, void* stream);
/* out[d] (+)= sum_r x[r][d] for a dense f32 [R][D] matrix (per-batch partials -> parameter gradient) */
int cfhip_colreduce_f32(const float* x, float* out, int R, int D, int accumulate, void* stream);
/* two [R, D] matrices into two outputs in one launch (GroupNorm's dgamma / dbeta partial sums: norms.py nn.GroupNorm's weight / bias gradients) */
int cfhip_colreduce2_f32(const float* x_a, float* out_a, const float* x_b, float* out_b, int R, int D, int accumulate, void* stream);
int cfhip_timestep_embedding(const int64_t* t, float* out, int B, int dim, float max_period, void* stream);


/* ------------------------------------------------------------------------------------------
 * CLIP text tower index work (multimodal/clip.py:209-256) — gathers are bit-exact
 *   embedding_fwd: out[n][:] = table[indices[n]][:] (+ pos[n % T][:]) — nn.Embedding lookup fused with
 *     the learned positional add (mixed_stacks/api.py:209-228); also the EOT pooling
 *     `net[arange(B), indices.argmax(-1)]` with table = the [B*T, D] stream and indices = b*T + argmax.
 *     table / pos f32, indices int64, out f32 or bf16, D % 4 == 0.
 *   embedding_bwd: dtable[indices[n]][:] += dy[n][:] (f32 atomics; caller zeroes / owns dtable;
 *     rows equal to padding_idx are skipped like nn.Embedding(padding_idx=...)).
 *   l2norm: y = x / ||x||_2 per f32 row (cftool.array.l2_normalize, no epsilon); inv_norm saved.
 * ------------------------------------------------------------------------------------------ */
int cfhip_embedding_fwd(const float* table, const int64_t* indices, const float* pos, void* out,
                        int out_is_f32, int64_t N, int D, int T, int64_t V, void* stream);
int cfhip_embedding_bwd(const void* dy, int dy_is_f32, const int64_t* indices, float* dtable, int64_t N,
                        int D, int64_t V, int64_t padding_idx, void* stream);
int cfhip_l2norm_fwd(const float* x, float* y, float* inv_norm, int64_t N, int D, void* stream);
int cfhip_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int64_t N, int D,
                     void* stream);
/* Similarity logits of the CLIP towers in fp32 (IPerceptor.forward, multimodal/schema.py:25-30:
 * `logit_scale.exp() * image_features @ text_features.t()`) and the products of its backward:
 * C[m][n] = alpha * alpha_dev[0] * sum_k A(m,k) * B(n,k), A(m,k) = a_trans ? A[k*lda + m] : A[m*lda + k] (B likewise),
 * C dense [M][N]; alpha_dev (device scalar, e.g. exp(logit_scale)) may be NULL.  fp32 FMA in k order: the operands are
 * not rounded to bf16.  dot: out[0] += sum a[i]*b[i] (d logit_scale = sum(dlogits * logits); caller zeroes).
 * The contrastive loss built on them (contrastive.py) has no counterpart in the reference: parity unpinned. */
int cfhip_sgemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb,
                    int a_trans, int b_trans, const float* alpha_dev, float alpha, void* stream);
int cfhip_dot_f32(const float* a, const float* b, float* out, int64_t n, void* stream);


/* ------------------------------------------------------------------------------------------
 * A12  dropout / stochastic depth (nn.Dropout at channel_mixers.py:30-41, mixed_stacks/api.py:130-158,
 *      ml_encoder.py:60-70, mappings.py:60-80; DropPath at modules/core/customs.py:429-446)
 *   dropout: y[i] = keep_i ? x[i] * s : 0 with s = 1 / (1 - p) in x's dtype (bf16 tensors: s rounded to bf16 first,
 *     as torch forms its noise `mask / (1 - p)` in the input's dtype) — bit-equal to torch given the mask.  keep_i = mask_in[i] != 0 when mask_in is given (one byte per element: parity tests inject the
 *     reference's mask), else u_i >= p with u from Philox4x32-10 keyed by `seed`, counter = offset + i / 4, lane i % 4
 *     — a pure function of (seed, offset, i), so the BACKWARD is the same call on dy with the same (seed, offset):
 *     no mask tensor is stored.  A call consumes ceil(n / 4) counters.  mask_out (optional) receives the mask.
 *     x / y bf16 or f32, 16-byte aligned; 0 <= p < 1.
 *   drop_path_mask: mask[b] = floor(keep_prob + u_b), 0 or 1 (customs.py:439-441), u_b from the same generator
 *     (counter = offset + b / 4).  drop_path: y[b][:] = (x[b][:] / keep_prob) * mask[b] — the reference's two
 *     roundings (customs.py:442) with a true IEEE division, `inner` elements per sample (inner % 4 == 0), bf16 or
 *     f32; bit-equal to the reference given the mask.  Its backward is the same call on dy.
 * ------------------------------------------------------------------------------------------ */
int cfhip_dropout(const void* x, void* y, int is_f32, int64_t n, float p, uint64_t seed, uint64_t offset,
                  const uint8_t* mask_in, uint8_t* mask_out, void* stream);
int cfhip_drop_path_mask(float* mask, int64_t B, float keep_prob, uint64_t seed, uint64_t offset, void* stream);
int cfhip_drop_path(const void* x, void* y, int is_f32, const float* mask, float keep_prob, int64_t B, int64_t inner,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * A16  tabular encoder (ml_encoder.Encoder.forward, modules/core/ml_encoder.py:171-209, + CommonMLModel.encode,
 *      models/ml/common.py:67-87) as one gather: out[b] = [numerical columns | one-hot blocks | embedding rows].
 *   plan: Fo entries of 6 int32 {src column, kind (0 copy / 1 one-hot / 2 embedding), payload (class id / table
 *     column), table id, dim (number of categories), table row pitch}, built by the host once per encoder.
 *   Categorical value v -> index: v >= dim -> 0 (the reference's out-of-bound imputation), else (int64)v
 *     (truncation, `.to(torch.long)`).  Integer index work: one-hot block and gathered rows are bit-exact.
 *   tables / dtables: device arrays of f32 base pointers, one per embedding column (`embeddings.<col>.weights`,
 *     [dim][pitch]); a NULL dtables entry skips that table's gradient.
 *   indices: int64 [B][K] of the K categorical columns (EncodingResult.indices).
 *   bwd: numerical outputs -> dx[b][src] (dx may be NULL), embedding outputs -> f32 atomic scatter-add into dtables
 *     (caller zeroes / owns them); one-hot outputs carry no gradient.
 * ------------------------------------------------------------------------------------------ */
int cfhip_ml_encode_fwd(const float* x, int64_t B, int F, int64_t x_row_stride, const int32_t* plan, int Fo,
                        const void* const* tables, float* out, void* stream);
int cfhip_ml_encode_indices(const float* x, int64_t B, int64_t x_row_stride, const int32_t* cols, const int32_t* dims,
                            int K, int64_t* indices, void* stream);
int cfhip_ml_encode_bwd(const float* dout, const float* x, int64_t B, int F, int64_t x_row_stride, const int32_t* plan,
                        int Fo, void* const* dtables, float* dx, void* stream);


/* ------------------------------------------------------------------------------------------
 * C1 / C2 / C3  collectives of the data-parallel exchange over RCCL / xGMI.  Replace what
 *   `accelerator.prepare` -> torch DDP would have done behind trainer.py:268-272 (gradient all-reduce, C1; the DDP
 *   constructor's parameter broadcast, C3) and add the embedding all-gather / reduce-scatter of a contrastive step (C2).
 *   One communicator per process (one process per GPU).  RCCL is reached with dlopen (the copy the process already
 *   holds — torch's — is preferred), so libcfhip.so has no link-time RCCL dependency.
 *   unique_id: rank 0 fills 128 bytes (ncclGetUniqueId); the caller ships them to the other ranks (any host channel:
 *     the launcher's TCP store, a file).  init: collective over all ranks; the CURRENT HIP device is the rank's GPU.
 *   Every collective is in-stream on `stream` (caller's comm stream), never synchronises the host.
 *   dtype: 0 = f32, 1 = bf16.  allreduce / broadcast are in place; allgather / reduce_scatter take `count_per_rank`
 *     elements per rank (recv of allgather / send of reduce_scatter hold world * count_per_rank).  Reductions are sums:
 *     the 1 / W of gradient averaging is the optimizer's grad_scale.
 * ------------------------------------------------------------------------------------------ */
int cfhip_comm_unique_id(void* out128);
int cfhip_comm_init(int rank, int world, const void* uid128, void** comm);
int cfhip_comm_destroy(void* comm);
/* ncclCommCount / ncclCommUserRank of the communicator: what RCCL reports, for the benchmark's evidence line (`rank` may be NULL) */
int cfhip_comm_count(void* comm, int* world, int* rank);
int cfhip_comm_allreduce(void* comm, void* buf, size_t count, int dtype, void* stream);
int cfhip_comm_allgather(void* comm, const void* send, void* recv, size_t count_per_rank, int dtype, void* stream);
int cfhip_comm_reduce_scatter(void* comm, const void* send, void* recv, size_t count_per_rank, int dtype, void* stream);
int cfhip_comm_broadcast(void* comm, void* buf, size_t count, int dtype, int root, void* stream);

/* ------------------------------------------------------------------------------------------
 * Grouped convolution (groups > 1, depthwise included): F.conv2d(net, w, bias, stride, padding, dilation, groups) reached
 * from Conv2d.forward (modules/core/convs/basic.py:160-177) and its two backward halves.  An option outside the named
 * benchmark configurations: direct kernels (one thread per output / input element, one workgroup per filter plane for the
 * weight gradient), fp32 accumulation, no atomics.  x / y / dy / dx bf16 NCHW; w bf16 [Cout][Cin / groups][kh][kw];
 * bias, dw, bias_grad f32; kh * kw <= 49 for the weight gradient.
 * ------------------------------------------------------------------------------------------ */
int cfhip_conv2d_grouped_fwd(const void* x, const void* w, const float* bias, void* y, int B, int Cin, int H, int W, int Cout,
                             int kh, int kw, int stride, int pad, int dil, int groups, void* stream);
int cfhip_conv2d_grouped_bwd_input(const void* dy, const void* w, void* dx, int B, int Cin, int H, int W, int Cout, int kh,
                                   int kw, int stride, int pad, int dil, int groups, void* stream);
int cfhip_conv2d_grouped_bwd_weight(const void* dy, const void* x, float* dw, int accumulate, float* bias_grad,
                                    int bias_grad_accumulate, int B, int Cin, int H, int W, int Cout, int kh, int kw,
                                    int stride, int pad, int dil, int groups, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CFHIP_H */
