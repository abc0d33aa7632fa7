/* maest_hip.h -- C ABI of libmaest_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * MAEST mel-spectrogram -> patchout-ViT hot path.
 *
 * The reference (palonso/MAEST) is pure Python with NO plugin / operator / FFI interface: its
 * boundary for this path is the Python API get_maest() / MAEST.forward() / predict_labels() /
 * Module.training_step() (models/maest.py:1467-1569, 831-939; models/module.py:73-102), and every
 * tensor op underneath is a stock torch / torchaudio call.  This header therefore declares the
 * entry points that the drop-in Python package (maest_amd/, same API as the reference) binds with
 * ctypes; each one cites the reference call site whose library op it replaces.
 *
 * Conventions
 *  - Every function returns int status: 0 = MAEST_OK; non-zero -> maest_last_error() has the text.
 *  - All tensor arguments are CALLER-OWNED DEVICE pointers (e.g. torch.Tensor.data_ptr()), row-major,
 *    with explicit sizes / leading dimensions in ELEMENTS.  The library never allocates, frees or
 *    retains memory and never synchronises: work is enqueued on `stream` (a hipStream_t passed as
 *    void*; NULL = the null stream), so every entry point is hipGraph-capturable.
 *  - dtype codes: MAEST_F32 (fp32, "parity mode": exact-fp32 MFMA) and MAEST_BF16 (bf16 operands,
 *    fp32 accumulate, "perf mode").  The residual stream, LayerNorm statistics, softmax statistics,
 *    logits and all parameter gradients are fp32 in both modes.
 *  - No torch types, no C++ types: plain pointers and integers only.
 */
#ifndef MAEST_HIP_H
#define MAEST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAEST_ABI_VERSION 8

#define MAEST_OK 0
#define MAEST_ERR_INVALID 1 /* bad argument (shape / alignment / dtype) */
#define MAEST_ERR_LAUNCH 2  /* HIP launch failure */

#define MAEST_F32 0
#define MAEST_BF16 1
#define MAEST_F32X3 2 /* fp32 tensors, matrix products in split-bf16 ("bf16 x 3") precision: accepted as the INPUT dtype
                         of maest_gemm_nt / maest_gemm_tn and as the dtype of maest_attn_fwd / maest_attn_bwd only (the
                         other entry points take MAEST_F32 for the same tensors); ~2^-16 per-product error on the
                         full-rate bf16 matrix pipe */

#define MAEST_BF16_QS 4 /* bf16 qkv tensor whose q columns hold q' = scale * log2(e) * q (the factor folded into the q rows of the qkv
                           projection's operand copy by maest_cast_weights_multi: q' is rounded once, forward and backward read the
                           same operand): accepted as the dtype of maest_attn_fwd(_rows) / maest_attn_bwd(_rows) only.  `scale` stays
                           the TRUE softmax scale; dQ comes out as the gradient with respect to the true q (so the projection's
                           dgrad / wgrad run on the unscaled weights), lse and the output are those of MAEST_BF16 up to rounding. */

/* The split-bf16 ("bf16 x 3") product as ONE bf16 GEMM of three times the depth (round 5): an fp32 value x is hi + lo with hi = bf16(x),
 * lo = bf16(x - hi); a row of K values is stored as 3 K bf16 values -- activations [ hi | hi | lo ] (MAEST_SPLIT3_A), weights
 * [ hi | lo | hi ] (MAEST_SPLIT3_B) -- so that the plain bf16 GEMM over K' = 3 K accumulates hi*hi + hi*lo + lo*hi, the three terms of
 * MAEST_F32X3, on the fast bf16 kernels (gemm_nt_ow.hip).  MAEST_SPLIT3_A: accepted as y_dtype of maest_layernorm_fwd /
 * maest_add_layernorm_fwd (y: bf16 [rows, 3 * 768], ldy = 2304); MAEST_SPLIT3_B as the dtype of maest_cast_weights_multi (dst: bf16
 * [rows, 3 * cols]; dst_t must be NULL); MAEST_F32X3_A3 as the dtype of maest_attn_fwd(_rows): MAEST_F32X3 with `out` written as
 * MAEST_SPLIT3_A rows (bf16 [B * N, 3 * 768]). */
#define MAEST_SPLIT3_A 5
#define MAEST_SPLIT3_B 6
#define MAEST_F32X3_A3 7

/* GEMM epilogues */
#define MAEST_F16 3 /* IEEE half: accepted as the INPUT dtype of maest_patch_im2col only (the loader's float16 mel batches,
                       discogs/dataset.py:58-67) */

#define MAEST_EPI_NONE 0     /* C = acc + bias                                           */
#define MAEST_EPI_GELU 1     /* C = gelu_erf(acc+bias) ; aux_out (optional) = gelu_erf'(acc+bias), saved for backward */
#define MAEST_EPI_RESIDUAL 2 /* C(fp32) = acc + bias + aux_in(fp32)                      */
#define MAEST_EPI_MUL 3      /* C = acc * aux_in   (aux_in in out_dtype; dgrad through GELU)  */
#define MAEST_EPI_ATOMIC 4   /* C(fp32) += acc   (split-K accumulate, C pre-zeroed)      */
#define MAEST_EPI_ROWDOT 5   /* internal to maest_gemm_nt_rowdot (not accepted by maest_gemm_nt) */

int maest_version(void);
const char* maest_last_error(void);

/* Which of the hand-scheduled kernels that own fixed registers are in this build.  maest_amd/build.py audits their code objects
 * (maest_amd/pw_audit.py); a kernel whose audit fails under some other hipcc is left out and the dispatch keeps the kernel it
 * replaced (same results: the forms are bit-equal for the GEMMs, equal to rounding for the attention forward).  *mask = OR of: */
#define MAEST_FORM_GEMM_NT_OW 1  /* gemm_nt_ow.hip: bf16 NT GEMM, one wave per SIMD (else the eight-wave kernel)            */
#define MAEST_FORM_GEMM_TN_OW 2  /* gemm_tn_ow.hip: bf16 wgrad GEMM, one wave per SIMD (else the eight-wave kernel)         */
#define MAEST_FORM_ATTN_FWD_PW 4 /* attn_fwd_pw.hip: persistent bf16 attention forward for N > 320 (else four-wave LDS-DMA) */
int maest_kernel_forms(int* mask);

/* ---- process-wide tuning / test switches.  Thread-safe (atomics); the defaults are taken from the environment
 * variables named below, which are read ONCE, at the first use of any switch.  restore_default != 0 ignores `value`.
 *   MAEST_OPT_GEMM_MIN_M    (env MAEST_GEMM_MIN_M,    default 8192): smallest M routed to the 256-row-tile GEMMs
 *   MAEST_OPT_GEMM_VARIANT  (env MAEST_GEMM_VARIANT,  default 0):    0 = full-line 256x256 kernels (bf16 operands, every epilogue
 *                            form incl. row-dot: four waves, one per SIMD, 128 x 128 outputs each -- gemm_nt_ow.hip / gemm_tn_ow.hip;
 *                            fp32 / split-bf16 operands and the 128-row tail tiles: eight waves -- gemm256.hip),
 *                            3 = as 0 with the eight-wave kernels for bf16 too (A/B, tests), 4 = 256-tile TN kernel at any qualifying
 *                            shape (1 / 2 selected two earlier 256-row kernels, removed in round 6)
 *   MAEST_OPT_GEMM_EPILOGUE (env MAEST_GEMM_EPILOGUE, default -1):   -1 = per-epilogue choice, 0/1/2 force a C-tile
 *                            epilogue form of the eight-wave full-line kernel (ignored by the one-wave-per-SIMD kernel, i.e. for
 *                            bf16 operands under MAEST_OPT_GEMM_VARIANT = 0) */
#define MAEST_OPT_GEMM_MIN_M 0
#define MAEST_OPT_GEMM_VARIANT 1
#define MAEST_OPT_GEMM_EPILOGUE 2
#define MAEST_OPT_ATTN_BWD 3 /* env MAEST_ATTN_BWD, default 0: fused one-pass attention backward where it applies
                                (bf16, N <= 320), query tiles fed by LDS-DMA; 1 = always the two-kernel dK/dV + dQ
                                form; (2 selected the register-fed fused form, removed in round 6: now as 0;)
                                3 = as 0 without the persistent form (one workgroup per (batch, head) at every shape).
                                4 = as 1 with the register-staged padded tiles (the bf16 two-kernel form otherwise streams its
                                tiles by LDS-DMA, bit-equal; A/B and tests).
                                Under 0, complete backward passes with 257 <= N <= 320 run the persistent form (one
                                workgroup per CU walks its (batch, head) items, the next item's K / V / query tiles
                                arriving while the current one computes) */
#define MAEST_OPT_GEMM_TAIL 5 /* env MAEST_GEMM_TAIL, default 1: the last partial round of a large NT GEMM runs in 128-row
                                 tiles (a second launch) when at most half a round of 256-row tiles is left over; 0: off;
                                 2: every tile a 128-row tile (tests) */
#define MAEST_OPT_ATTN_FWD 6 /* env MAEST_ATTN_FWD, default 0: bf16 attention forward by shape -- N > 320 (complete passes): the
                                persistent one-wave-per-SIMD kernel (attn_fwd_pw.hip: 96 query rows per wave, K / V tiles streamed
                                by LDS-DMA across work items); otherwise four-wave workgroups with K / V tiles fed by LDS-DMA into
                                unpadded bank-swizzled tiles; 1: the register-staged, padded-pitch form every other dtype uses;
                                2: the four-wave LDS-DMA form at every N; 3: the persistent form at every N (tests) */
#define MAEST_OPT_ATTN_FWD_WAVES 7 /* env MAEST_ATTN_FWD_WAVES, default 0: waves (32-query blocks) per workgroup of the DMA-fed bf16
                                      attention forward chosen by shape; 4 / 5 / 6 / 8 force one (tests, A/B) */
#define MAEST_OPT_TN_REDUCE 8 /* env MAEST_TN_REDUCE, default 0: split-K partials of the wgrad GEMM are combined with fp32 atomics; 1:
                                 maest_gemm_tn_ws uses the workspace it is given (partial tiles stored plainly, summed in split
                                 order by a second kernel: bit-reproducible dW; no measurable cost on the training step) */
#define MAEST_OPT_GEMM_WGS 9 /* env MAEST_GEMM_WGS, default 0: the bf16 NT GEMM (gemm_nt_ow.hip) launches one workgroup per tile; n > 0:
                               at most n workgroups, workgroup b walking tiles b, b + n, ... with the next tile's first operand units
                               requested from inside the epilogue (256 = one per CU; small values make a workgroup walk several tiles
                               at test shapes) */
#define MAEST_OPT_GEMM_PANEL 10 /* env MAEST_GEMM_PANEL, default 0: the bf16 NT GEMM (gemm_nt_ow.hip) walks its tiles row-major
                                  inside an XCD's range; -1: in column panels where a traffic estimate says so (B larger than ~3 MB: N = 3072
                                  at K = 768 in panels of 6, N = 2304 in 5 + 4); n > 0: panels of n tiles.  Results do not depend on it.
                                  Measured (profiles/r06_gemm_panels.txt): fabric reads of fc1 5.8 -> 4.0 x the operand bytes, time equal
                                  (they are Infinity-Cache hits), inference step +0.4 % -- hence off by default. */
#define MAEST_OPT_LN_BWD_BLOCKS 4 /* env MAEST_LN_BWD_BLOCKS, default 1024: workgroup cap of the LayerNorm backward grid */
int maest_set_option(int opt, int value, int restore_default);
int maest_get_option(int opt, int* value);
/* ABI 7: an override of one switch for the CALLING THREAD only (clear != 0 removes it; `value` is then ignored).  Launches made by this
 * thread see the override, every other thread the process-wide value; maest_get_option on this thread returns the override.  The
 * engine sets the launch form of a forward / backward pass this way (maest_amd/maest.py: _gemm_form), so that two models driven
 * from two threads of one process cannot change each other's launches. */
int maest_set_option_thread(int opt, int value, int clear);

/* ---- K8, K10-K12, K13 head, K4 (im2col form) and their dgrad / wgrad ---------------------------
 * nn.Linear: models/maest.py:353,355,361,376 ; :197-199,203-206 ; :572,579 ; nn.Conv2d :238-240.
 *   C[M,N] = epilogue( sum_k A[m,k] * B[n,k] )      A:[M,K] lda, B:[N,K] ldb (both k-contiguous)
 * in_dtype = dtype of A and B; out_dtype = dtype of C / aux_out (and aux_in for MUL).
 * GELU uses libm erf in fp32 mode and a 4-term erf of the Abramowitz-Stegun 7.1.26 form (|err| <= 1.7e-6) in bf16 mode.
 * K must be a multiple of 64 (bf16) / 32 (fp32): callers zero-pad.  bias: fp32 [N] or NULL. */
int maest_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, int in_dtype,
                  void* C, int64_t ldc, int out_dtype, int M, int N, int K,
                  const float* bias, int epi, const void* aux_in, void* aux_out, int64_t ld_aux,
                  int split_k, void* stream);

/* ---- the same GEMM (no epilogue other than the bias) plus, out of the same C-tile pass, the dot products of every row of
 * the STORED C (rounded to out_dtype) with the same row of `other` (out_dtype, [M, N], ld_other), per group of 64
 * columns:
 *   rowdot[((m / rows_per_item) * (N / 64) + g) * rows_per_item + m % rows_per_item] = sum_{c < 64} C[m, 64 g + c] * other[m, 64 g + c]
 * (fp32, [M / rows_per_item, N / 64, rows_per_item]).  With C = dO (the gradient of the attention output, produced by the
 * dgrad GEMM of the output projection, models/maest.py:376), other = O and rows_per_item = tokens per clip this is the
 * `delta = rowsum(dO * O)` per (clip, head, query) the attention backward needs: pass it to maest_attn_bwd_rows with
 * out = NULL and the separate pass over dO and O disappears.  N % 64 == 0, M % rows_per_item == 0.  Shapes the 256-row
 * tile kernels do not take run the plain GEMM followed by a small reduction kernel (same result). */
int maest_gemm_nt_rowdot(const void* A, int64_t lda, const void* B, int64_t ldb, int in_dtype, void* C, int64_t ldc,
                         int out_dtype, int M, int N, int K, const float* bias, const void* other, int64_t ld_other,
                         float* rowdot, int rows_per_item, void* stream);

/* ---- wgrad + bias grad, no transposed copies ("TN": both operands token-major as they sit in HBM) ----
 *   C[M,N] (fp32, ACCUMULATED: zero it first) += sum_k A[k,m] * B[k,n]      A:[K,M] lda, B:[K,N] ldb
 *   colsum[m] (fp32, ACCUMULATED, may be NULL) += sum_k A[k,m]              (= the bias gradient)
 * dW = dY^T X of nn.Linear backward with A = dY [tokens, out], B = X [tokens, in].  Any K (the token
 * tail is zero-filled in LDS); rows of A / B must be 16-byte multiples (lda / ldb) and may be wider
 * than M / N.  split_k partials are combined with fp32 atomics; split_k = 0 picks it automatically. */
int maest_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int dtype, float* C,
                  int64_t ldc, int M, int N, int K, float* colsum, int split_k, void* stream);
/* The same with a caller-owned scratch buffer (ABI 5): the DETERMINISTIC form of the split-K combine, selected by
 * MAEST_OPT_TN_REDUCE = 1.  Given `workspace_bytes` >= the figure maest_gemm_tn_workspace_bytes reports, of 16-byte aligned device
 * memory (and C 16-byte aligned, ldc % 4 == 0), the K-split partial tiles are written to it with plain 16-byte stores and a second
 * kernel adds them to C in split order: no atomics on C, so dW is bit-reproducible from run to run (colsum still uses atomics).
 * Measured against the atomics (profiles/r03_ab_tn_workspace_combine.txt): the GEMM + reduce pair timed alone is 21 % faster at the
 * proj shape (28 splits), 6 % at qkv, equal at fc1 / fc2; the training step is equal within the pairs' spread -- opt-in because
 * there is no gain to claim, not because it costs.  workspace = NULL, a smaller buffer, the option at 0, or a shape
 * for which maest_gemm_tn_workspace_bytes reports 0 = exactly maest_gemm_tn.  The workspace is dead when the call's work on
 * `stream` has completed. */
int maest_gemm_tn_workspace_bytes(int dtype, int M, int N, int K, int split_k, int64_t* bytes);
int maest_gemm_tn_ws(const void* A, int64_t lda, const void* B, int64_t ldb, int dtype, float* C, int64_t ldc, int M, int N,
                     int K, float* colsum, int split_k, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- 2-D transpose with zero padding: dst[c, r] = src[r, c], dst rows padded to ld_dst ----------
 * (operand preparation for the wgrad GEMMs: dW = dY^T X needs both operands token-contiguous). */
int maest_transpose(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int cols,
                    int dtype, void* stream);

/* ---- fp32 -> bf16 cast of parameters, optionally also emitting the transpose ---------------------
 * dst[r,c] = bf16(src[r,c]) ; dst_t[c,r] = bf16(src[r,c]) (dst / dst_t may be NULL).
 * With dtype == MAEST_F32 it is a plain copy / transpose (parity mode). */
int maest_cast_weights(const float* src, void* dst, void* dst_t, int rows, int cols, int dtype,
                       void* stream);
/* The same for n parameters in one launch (HOST arrays of n device pointers / shapes; dst[i] or dst_t[i] may
 * be NULL): the operand-copy refresh after an optimizer step (autocast's per-call weight casts in the
 * reference, ex_maest.py:51 precision="16-mixed").  scaled_rows (HOST array of n, or NULL): rows [0, scaled_rows[i]) of dst[i] --
 * not of dst_t[i] -- are multiplied by row_scale before the rounding: the q rows of a qkv projection's forward operand copy carry
 * scale * log2(e) for MAEST_BF16_QS attention (the transposed copy, which serves the dgrad, stays unscaled). */
int maest_cast_weights_multi(int n, const float* const* src, void* const* dst, void* const* dst_t,
                             const int* rows, const int* cols, const int* scaled_rows, float row_scale, int dtype,
                             void* stream);

/* ---- K7 LayerNorm over the last dim (nn.LayerNorm: models/maest.py:395,405,499,553,571) ---------
 * x: fp32 [rows, cols] (ldx); y: y_dtype [rows, cols] (ldy); mean/rstd: fp32 [rows] or NULL.
 * cols must be 768. */
int maest_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y,
                        int64_t ldy, int y_dtype, float* mean, float* rstd, int rows, int cols,
                        float eps, void* stream);
/* The residual add fused into the LayerNorm that follows it (models/maest.py:418-419 then :395/:405 of the next
 * LayerNorm): x_out[r,:] = x[r,:] + delta[r,:] (fp32, contiguous [rows, 768]; may alias x) and y = LN(x_out).
 * delta: the proj / fc2 Linear output including its bias, in delta_dtype. */
int maest_add_layernorm_fwd(const float* x, const void* delta, int delta_dtype, float* x_out, const float* gamma,
                            const float* beta, void* y, int y_dtype, float* mean, float* rstd, int rows, int cols,
                            float eps, void* stream);
/* dx_out[r,:] = dres[r,:] (may be NULL) + LN'(dy)[r,:]   (fp32), plus an optional copy of dx_out in
 * `dx_lp_dtype` (operand of the next dgrad GEMM).  dgamma/dbeta (fp32 [cols]) are ACCUMULATED
 * (atomics) -- zero them first. */
int maest_layernorm_bwd(const void* dy, int64_t lddy, int dy_dtype, const float* x, int64_t ldx,
                        const float* gamma, const float* mean, const float* rstd,
                        const float* dres, float* dx_out, void* dx_lp, int dx_lp_dtype,
                        float* dgamma, float* dbeta, int rows, int cols, void* stream);

/* The same with a COMPACT residual gradient: dres is [rows / n_tok][n_head][768] -- the first n_head tokens of every
 * clip of n_tok tokens; the other tokens' residual gradient is zero.  (Last block of the network: only the tokens the
 * head reads, cls and dist, carry a gradient above it -- models/maest.py:819-826.)  n_head = 0: dense dres / NULL. */
int maest_layernorm_bwd_headres(const void* dy, int64_t lddy, int dy_dtype, const float* x, int64_t ldx,
                                const float* gamma, const float* mean, const float* rstd,
                                const float* dres, float* dx_out, void* dx_lp, int dx_lp_dtype,
                                float* dgamma, float* dbeta, int rows, int cols, int n_tok, int n_head, void* stream);

/* ---- K9 fused softmax attention (Attention.forward: models/maest.py:362-375) ---------------------
 * qkv: [B*N, 2304] with column = s*768 + h*64 + d  (s = 0,1,2 for q,k,v), exactly the layout the
 * reference's qkv Linear produces before its reshape/permute (:362-363).
 * out: [B*N, 768] (column = h*64 + d), i.e. the reference's (attn @ v).transpose(1,2).reshape(B,N,C).
 * lse: fp32 [B, 12, N] log-sum-exp of the scaled scores (saved for backward) or NULL. */
int maest_attn_fwd(const void* qkv, void* out, float* lse, int B, int N, int dtype, float scale,
                   void* stream);
/* delta: fp32 [B,12,N] workspace (rowsum(dO*O)); dqkv: [B*N, 2304] same layout as qkv.
 * out == NULL: `delta` already holds rowsum(dO*O) (maest_gemm_nt_rowdot) and is only read. */
int maest_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                   float* delta, void* dqkv, int B, int N, int dtype, float scale, void* stream);
/* Attention of the LAST block, where only the first q_rows tokens of a clip (cls, dist) are read by what follows
 * (final norm + head, models/maest.py:819-826): the same functions restricted to those queries; keys / values stay
 * complete.  Forward: rows / lse of queries >= roundup(q_rows, 32) are NOT written.  Backward: dout rows
 * [q_rows, roundup(q_rows, 32)) must be zero, later rows are not read (nor are out / lse there); dK, dV complete, dQ = 0
 * for every query >= q_rows.  q_rows = N is maest_attn_fwd / maest_attn_bwd.  Backward with q_rows < N is served by
 * the fused bf16 kernel only (N <= 320): MAEST_ERR_INVALID otherwise. */
int maest_attn_fwd_rows(const void* qkv, void* out, float* lse, int B, int N, int dtype, float scale, int q_rows,
                        void* stream);
int maest_attn_bwd_rows(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                        int B, int N, int dtype, float scale, int q_rows, void* stream);
/* Rows of the first n_head tokens of every clip, [clips][n_tok][768] -> compact [clips][n_head][768] (dtype fp32 / bf16),
 * and back: dst rows [0, n_pad) of every clip = the compact rows followed by zeros, rows >= n_pad untouched. */
int maest_gather_head_rows(const void* src, int clips, int n_tok, int n_head, int dtype, void* dst, void* stream);
int maest_scatter_head_rows(const void* src, int clips, int n_tok, int n_head, int n_pad, int dtype, void* dst,
                            void* stream);

/* ---- K16 + K4 operand: mixup + im2col of the 16x16 / stride-10 patches ---------------------------
 * PatchEmbed.forward (models/maest.py:243-256) fused with Module.training_step's mixup
 * (models/module.py:77-83) and with EVERY patchout variant of models/maest.py:678-780 (structured time /
 * frequency, fixed index lists, interleaved, unstructured): the host resolves them into one list of kept
 * patch tokens and dropped patches are never computed -- mathematically identical to compute-then-drop.
 * x: [B, F, T] in x_dtype = MAEST_F32 or MAEST_F16 (the loader hands out float16 mel batches, discogs/dataset.py:58-67;
 * a half sample is widened exactly as x.float() would, in the load); perm: int32 [B] or NULL; lam: fp32 [B] or NULL (x' = lam*x + (1-lam)*x[perm]).
 * tok_ft: int32 [P, 2] = (frequency patch index f, time patch index t) of each kept token, in sequence
 * order.  out: dtype [B*P, 256], row = b*P + j, col = ky*16 + kx, patch origin (10 f, 10 t).
 * Optional K17 SpecMasking fused into the same load (helpers/spec_masking.py:27-33; the loader masks each clip
 * before the batch is mixed up, discogs/datamodule.py:140-152): t_stripes int32 [B, n_t, 2] / f_stripes int32
 * [B, n_f, 2] = (start, width) per clip, as in maest_spec_mask; a masked sample reads as 0.0 (for the clip and,
 * with its own stripes, for its mixup partner).  n_t = n_f = 0 (pointers may be NULL) disables it. */
int maest_patch_im2col(const void* x, int x_dtype, int B, int F, int T, const int32_t* perm, const float* lam,
                       const int32_t* tok_ft, int P, const int32_t* t_stripes, int n_t,
                       const int32_t* f_stripes, int n_f, void* out, int dtype, void* stream);
/* The same with the convolution's stride as an argument (ABI 8): patch origin (stride_f * f, stride_t * t).  The reference takes the
 * strides as constructor arguments (get_maest(stride_f=, stride_t=): models/maest.py:1505-1507, 1537; PatchEmbed: models/maest.py:214-241) and
 * only warns that the checkpoints were trained with (10, 10); maest_patch_im2col is this entry with (10, 10). */
int maest_patch_im2col_strided(const void* x, int x_dtype, int B, int F, int T, int stride_f, int stride_t, const int32_t* perm,
                               const float* lam, const int32_t* tok_ft, int P, const int32_t* t_stripes, int n_t,
                               const int32_t* f_stripes, int n_f, void* out, int dtype, void* stream);

/* ---- K5 + K6: positional add + token assembly (models/maest.py:645-675, 769, 785-796) ------------
 * patches: fp32 [B*P, 768] (conv output incl. bias); x0: fp32 [B, 2 + P, 768]
 *   x0[b,0] = cls + new_pos[0]; x0[b,1] = dist + new_pos[1];
 *   x0[b, 2 + j] = (patches[b*P + j] + time_pos[:, toffset + t_j]) + freq_pos[:, f_j]
 * time_pos: fp32 [768, Tt]; freq_pos: fp32 [768, Fg]. */
int maest_token_assemble(const float* patches, const float* cls_token, const float* dist_token,
                         const float* new_pos, const float* freq_pos, const float* time_pos, int Fg,
                         int Tt, int toffset, const int32_t* tok_ft, int B, int P, float* x0,
                         void* stream);
/* Backward of the above: dx0 fp32 [B, 2+P, 768] -> dpatches (dtype, [B*P, 768]) and ACCUMULATED
 * fp32 grads d_cls[768], d_dist[768], d_new_pos[2,768], d_freq_pos[768,Fg], d_time_pos[768,Tt]. */
int maest_token_assemble_bwd(const float* dx0, int B, int P, int Fg, int Tt, int toffset,
                             const int32_t* tok_ft, void* dpatches, int dtype, float* d_cls,
                             float* d_dist, float* d_new_pos, float* d_freq_pos, float* d_time_pos,
                             void* stream);

/* ---- K13 / K14: final LayerNorm on the cls/dist tokens + feature pooling -------------------------
 * models/maest.py:806-810, 905-906: xn = LN_eps(x)[:, 0:2]; cls = xn[:,0]; dist = xn[:,1];
 * feat = (cls + dist)/2.   x: fp32 [B, N, 768].  Outputs fp32 [B,768] each; mean/rstd fp32 [B,2]. */
int maest_head_pool_fwd(const float* x, int B, int N, const float* gamma, const float* beta, float eps,
                        float* cls, float* dist, float* feat, float* mean, float* rstd, void* stream);
/* d_cls_total = d_cls + d_feat/2 (either may be NULL), same for dist; writes dx fp32 [B,N,768]
 * rows 0,1 (all other rows are ZEROED), accumulates dgamma/dbeta. */
int maest_head_pool_bwd(const float* d_cls, const float* d_dist, const float* d_feat, const float* x,
                        int B, int N, const float* gamma, const float* mean, const float* rstd,
                        float* dx, float* dgamma, float* dbeta, void* stream);
/* early-exit embedding (models/maest.py:825-829): emb[b] = cat(x[b,0], x[b,1], mean(x[b,2:], 0)) */
int maest_embed_pool(const float* x, int B, int N, float* emb, void* stream);

/* ---- K18: BCE-with-logits, mean reduction (models/module.py:90, 299-301) ------------------------
 * loss (fp32 scalar, ACCUMULATED: zero it first) += weight * mean(max(z,0) - z*y + log1p(exp(-|z|)))
 * dlogits = weight * (sigmoid(z) - y) / (rows*cols)  (or NULL).
 * Optional fused label mixup: y' = lam[b]*y[b] + (1-lam[b])*y[perm[b]]. */
int maest_bce_logits(const float* z, const float* y, const int32_t* perm, const float* lam, int rows,
                     int cols, float weight, float* loss, float* dlogits, void* stream);

/* ---- K15: predict_labels tail (models/maest.py:936-938): act[c] = mean_b sigmoid(z[b,c]) */
int maest_sigmoid_mean(const float* z, int rows, int cols, float* act, void* stream);

/* ---- column sums (bias gradients): out[c] += sum_r src[r,c]  (fp32 out, ACCUMULATED) */
int maest_colsum(const void* src, int64_t ld, int rows, int cols, int dtype, float* out, void* stream);

/* ---- K17 SpecMasking on explicit stripes (helpers/spec_masking.py:27-33): zero x[b,:,s:s+w] for the
 * time stripes and x[b,s:s+w,:] for the frequency stripes.  stripes: int32 [B, n, 2] (start,width). */
int maest_spec_mask(float* x, int B, int F, int T, const int32_t* t_stripes, int n_t,
                    const int32_t* f_stripes, int n_f, void* stream);

/* ---- K1-K3 fused log-mel front end (models/helpers/melspectrogram.py:47-60) -----------------------
 * wave: fp32 [B, S]; out: fp32 [B, 96, T] with T = 1 + S/256.  window: fp32 [512] periodic Hann;
 * fb_start/fb_len: int32 [96] first FFT bin and number of non-zero taps of each mel band;
 * fb_w: fp32 [96, fb_stride] tap weights; twiddle: fp32 [512, 2] = (re, im) of exp(-2*pi*i*k/512).
 * out = (log10(1 + log_scale * mel) - norm_mean) / norm_2std.  Requires S > 256 (reflect padding). */
int maest_logmel(const float* wave, int B, int S, const float* window, const float* twiddle,
                 const int32_t* fb_start, const int32_t* fb_len, const float* fb_w, int fb_stride,
                 float log_scale, float norm_mean, float norm_2std, float* out, void* stream);

/* ---- second mel parameterisation: AugmentMelSTFT (models/preprocess.py:17-128; north_star names the file, the
 * reference's MAEST path never calls it).  wave: fp32 [B, S] at 32 kHz; out: fp32 [B, n_mels, T], T = 1 + (S-1)/320.
 * y = pre0*x[n] + pre1*x[n+1] (pre-emphasis, reference [-0.97, 1]); STFT n_fft 1024 / hop 320 / center / reflect with
 * `window` = the 800-point non-periodic Hann zero-padded to 1024; power; band-sparse filterbank over 513 bins
 * (fb_start / fb_len / fb_w as in maest_logmel); out = (log(mel + log_eps) + norm_add) / norm_div.
 * twiddle: fp32 [1024, 2] = exp(-2*pi*i*k/1024).  n_mels <= 128. */
int maest_augment_mel(const float* wave, int B, int S, const float* window, const float* twiddle,
                      const int32_t* fb_start, const int32_t* fb_len, const float* fb_w, int fb_stride,
                      int n_mels, float pre0, float pre1, float log_eps, float norm_add, float norm_div,
                      float* out, void* stream);

/* ---- optimizer-side helper: scale a flat fp32 gradient bucket (after the RCCL all-reduce) */
int maest_scale_f32(float* x, int64_t n, float alpha, void* stream);
/* x *= *alpha_dev with the factor read from DEVICE memory: the upstream gradient of the scalar loss applied to the
 * logit gradients (autograd of F.binary_cross_entropy_with_logits, models/module.py:90) without a host round trip */
int maest_scale_dev_f32(float* x, int64_t n, const float* alpha_dev, void* stream);
/* dst[r, c] = cast(src[r, c]) for c < cols and 0 for cols <= c < ld_dst: fp32 rows (the logit gradients [B, C]) ->
 * zero-padded operand rows of the head's dgrad / wgrad GEMMs (K must be a multiple of 64 there) */
int maest_cast_rows(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int cols, int dtype,
                    void* stream);
/* stochastic weight averaging of n parameters in one launch (HOST arrays of device pointers / sizes):
 * avg[i] += (cur[i] - avg[i]) * inv_count   (Lightning StochasticWeightAveraging.avg_fn, helpers/swa_callback.py) */
int maest_swa_update_multi(int n, float* const* avg, const float* const* cur, const int64_t* numel,
                           float inv_count, void* stream);
/* x = (x + add) / div in place (AugmentMelSTFT's "fast normalization" after the training-time masks,
 * models/preprocess.py:129) */
int maest_affine_f32(float* x, int64_t n, float add, float div, void* stream);

/* ---- on-disk mel chunks -> network input (the step before the hot path) ------------
 * Replaces DiscogsDataset.load_melspectrogram (discogs/dataset.py:69-140) + norm_func
 * (discogs/datamodule.py:126-136) for a whole batch.
 * frames: raw float16 rows [sum_b frames_read[b], n_bands] as stored on disk (helpers/
 * melspectrogram_extractor.py:44-48), clip b starting at row row_start[b] (device arrays).
 * out: fp32 [B, n_bands, T]; clip b = its frames_read[b] <= T frames, zero padded to T, the padding
 * centred by a roll of (T - frames_read) / 2, transposed; with normalize != 0 then (x - mean) / div with both
 * constants and both results rounded to float16 exactly as numpy evaluates the reference expression. */
int maest_melfile_assemble(const uint16_t* frames, const int64_t* row_start, const int32_t* frames_read,
                           int B, int n_bands, int T, int normalize, float norm_mean, float norm_div,
                           float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAEST_HIP_H */
