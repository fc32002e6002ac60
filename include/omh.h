/*
 * omh.h — C ABI of libomh.so, the MI355X (gfx950) kernel library under the
 * Wan2.1 DiT / 3D-causal-VAE path of johndpope/OmniHuman-1-hack.
 *
 * The reference has no FFI for this path: its boundary is the Python
 * nn.Module surface (WanModel.forward, WanVAE.encode/decode — SURVEY.md §8b)
 * and every arithmetic op is delegated to PyTorch aten / flash-attn CUDA
 * kernels.  Each entry point below replaces one of those delegated ops and
 * cites the reference line(s) whose arithmetic it takes over; the host-side
 * Python in omnihuman-1-hack_amd/wan/ keeps the reference's call surface and
 * binds these symbols through ctypes.
 *
 * Conventions
 *  - plain C types only: device pointers (void* / float*), sizes, strides in
 *    ELEMENTS, a hipStream_t passed as void*.  No torch types.
 *  - every call is asynchronous on the given stream and allocates nothing;
 *    the caller owns all buffers.  The only process state is the option
 *    table below (omh_set_option; omh_set_deterministic is one of its
 *    entries): it is seeded from the environment ONCE, at the first call
 *    into the library, and the launch path never reads the environment.
 *  - return 0 on success, a negative OMH_E_* code on a rejected argument,
 *    or the positive hipError_t of a failed launch.
 *  - bf16 = 16-bit brain float (upper half of an IEEE fp32), fp32 accumulate
 *    everywhere.
 */
#ifndef OMH_H
#define OMH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMH_ABI_VERSION 12

#define OMH_E_BADARG   (-1)   /* null pointer / non-positive size             */
#define OMH_E_ALIGN    (-2)   /* pointer or leading dimension not aligned     */
#define OMH_E_SHAPE    (-3)   /* shape not supported by this kernel           */

typedef void* omh_stream_t;   /* hipStream_t */

int omh_abi_version(void);
/* gfx arch string the library was built for ("gfx950"). */
const char* omh_build_arch(void);
/* Deterministic mode (ABI v6; debugging aid for the training step).  The backward's few
 * remaining fp32 atomics — column sums of the bias gradients, the gate gradient of omh_gated_residual_bwd, the
 * split-K of omh_gemm_bf16_tn, the input gradient of omh_dense_f32_bwd — make two runs of one step differ in the last
 * bits.  With the mode on, each of those launches gives every output element to ONE workgroup that adds in a fixed
 * order (no split K, one row block): the whole step then repeats bit for bit, at a few percent of its speed.
 * on < 0 only queries.  The initial value is the environment's OMH_DETERMINISTIC (0 / 1).  Returns the mode in force. */
int omh_set_deterministic(int on);

/* Process options (ABI v10).  Every dispatch switch of the library — kernel-family overrides used by the parity tests
 * (one product on every kernel that can take it) and by A/B timing — is an entry of one table:
 *   ATTN_KERNEL ("w64" / "base")   ATTN_SPLIT ("0" / "tail")   W64_SPLIT ("0")   W64_VARIANT (ablation builds only)
 *   GEMM_KERNEL ("w64" / "8w")   GEMM_TILE ("big" / "small" / "tiny")   GEMM_RULE   GEMM_GROUP_M   GEMM_SPLITK ("0")
 *   GEMM_QKV ("0" / "1")   GEMM_W64_R192 / N192 / BF16M / GBWD / GAUX ("0" / "1")   GEMM_W64_P256 (experiment builds only)
 *   RMS_PAIR_ROW ("0" / "1": omh_rmsnorm_rope_bf16_pair's two-workgroup / one-wave-per-row form)
 *   GEMM_TN_W64 ("0" / "1")   GEMM_TN_TILE   GEMM_TN_GROUP_TILE ("big" / "small")   GEMM_TN_SPLIT (count)
 *   CONV_TILE ("w64" / "wide" / "small")   CONV_W64   CONV_WIDE_MIN   CONV_FUSE_NORM   CONV_KW3   CONV_PERSIST
 *   CONV_W64_UP2   LN_RPW ("1" / "2" / "4")   DETERMINISTIC ("0" / "1")
 * An unset option means "the library decides" (the shipped dispatch).  The table is seeded from the environment
 * (variable OMH_<NAME>) exactly once, at the first call into the library; afterwards only omh_set_option changes it —
 * no entry point calls getenv on the launch path.  Setting an option is not synchronised against concurrent launches.
 *   omh_set_option(key, value): key with or without the "OMH_" prefix; value NULL or "" unsets; at most 47 characters.
 *                               (NULL, NULL) restores every option to its start-up (environment) value.
 *                               Returns 0, OMH_E_BADARG for an unknown key, OMH_E_SHAPE for a value that is too long.
 *   omh_get_option(key): the value in force, NULL when unset or unknown (the pointer is valid until the next set).
 *   omh_option_count / omh_option_name(i): enumerate the table (DETERMINISTIC is not enumerated). */
int omh_set_option(const char* key, const char* value);
const char* omh_get_option(const char* key);
int omh_option_count(void);
const char* omh_option_name(int index);

/* ------------------------------------------------------------------------
 * GEMM  C[m][n] = epilogue( sum_k A[m][k] * B[n][k] )          (bf16 MFMA)
 * Replaces aten addmm under every nn.Linear of the DiT
 * (model.py:125-128,144-146,160,176-178,185,272-274,344,465-467) and the
 * patch-embedding Conv3d-as-GEMM (model.py:463,515).
 * A: [M,K] bf16 row-major (lda), B: [N,K] bf16 row-major (ldb) — i.e. the
 * nn.Linear weight layout [out,in].  K must be a multiple of 8; lda/ldb
 * multiples of 8; A/B 16-byte aligned.
 * ---------------------------------------------------------------------- */
enum {
    OMH_EPI_BF16      = 0,  /* C bf16  = acc + bias                               */
    OMH_EPI_F32       = 1,  /* C fp32  = acc + bias                               */
    OMH_EPI_GELU_BF16 = 2,  /* C bf16  = gelu_tanh(acc + bias)   (model.py:273)   */
    OMH_EPI_RESID     = 3,  /* C fp32 += (acc + bias) * gate     (model.py:296,313,328) */
    OMH_EPI_F32_ACCUM = 4,  /* C fp32 += acc + bias               (used by backward) */
    OMH_EPI_GELU_ERF_BF16 = 5, /* C bf16 = gelu_erf(acc + bias)   (i2v MLPProj, model.py:369) */
    OMH_EPI_GELU_BWD_BF16 = 6, /* C bf16 = acc * gelu_tanh'(aux)   (backward of model.py:273 fused into the dgrad GEMM) */
    /* ABI v10 — the fused q | k | v projection of the self-attention (model.py:144-146,152-153 on concatenated
       weights [3 dim, dim]): columns n < n_split go to C bf16 [M, ldc] = acc + bias[n] (q | k, what the norm kernel
       reads); columns n >= n_split go TRANSPOSED to aux: aux[(n - n_split) * ldaux + m] bf16 = acc + bias[n] — V^T
       [dim, ldaux >= M], the layout the attention kernel reads (omh_attn_args.vt).  One launch reads the activations
       once.  batch == 1, bias [N] or none, M and n_split multiples of 8, aux 16-byte aligned, ldaux a multiple of 8;
       columns m >= M of aux are not written.  Same bits as the two separate products
       (BF16 on [M, n_split], and BF16 + OMH_BIAS_M with the operands swapped). */
    OMH_EPI_BF16_SPLIT_T = 7
};
enum { OMH_BIAS_NONE = 0, OMH_BIAS_N = 1, OMH_BIAS_M = 2 };

typedef struct omh_gemm_args {
    const void* A; const void* B; void* C;
    int32_t M, N, K;
    int32_t lda, ldb, ldc;
    int32_t batch;                /* >=1; grid.z                                  */
    int64_t strideA, strideB, strideC;   /* per-batch element strides            */
    int32_t epilogue;             /* OMH_EPI_*                                    */
    int32_t bias_mode;            /* OMH_BIAS_*                                   */
    const float* bias;            /* [N] or [M] fp32, may be NULL                 */
    /* OMH_EPI_RESID gate = gate_const + gate0[n] + gate1[(m / gate_rows) * gate1_stride + n];
       either pointer may be NULL (treated as 0).                                 */
    const float* gate0; const float* gate1;
    int64_t gate1_stride; int32_t gate_rows; float gate_const;
    /* != 0: B is K-MAJOR, [K,N] bf16 row-major (ldb >= N, N a multiple of 8):  C = epi(sum_k A[m][k] B[k][n]).
       The input gradient of an nn.Linear, dx = dy W, on the weight as stored [out,in] (autograd of the Linears
       above under distilled_trainer.py:289-301) — no transposed weight copy.  Epilogues BF16 / F32 / F32_ACCUM. */
    int32_t b_kmajor;
    /* ABI v5 — fused epilogues of the training step (autograd of model.py:296,313,328,273 under
       distilled_trainer.py:289-301), batch == 1, the gemm_bf16.hip kernels:
       c_in : OMH_EPI_RESID reads the old residual from c_in (layout of C) instead of C: the out-of-place
              x_out = x_in + (acc + bias) * gate that keeps x_in for the backward.  NULL: in place.
       aux  : bf16 [M, >= N], row pitch ldaux (multiple of 8), 16-byte aligned.
              OMH_EPI_RESID          OUT  aux = bf16(acc + bias): the branch output the gate gradient needs
              OMH_EPI_GELU_BF16      OUT  aux = bf16(acc + bias): the pre-activation the GELU backward needs
              OMH_EPI_GELU_BWD_BF16  IN   the forward's pre-activation                                        */
    const float* c_in; void* aux; int32_t ldaux;
    /* ABI v9 — split K for few-row, long-contraction products (the FFN-down projection and the FFN-up input gradient of
       model.py:272-274,328 at one or two [16,1,60,104] clips: 56 / 104 tiles of 256 x 192 on 256 CUs).  With
       workspace_bytes >= omh_gemm_workspace_bytes(args) > 0 the contraction is cut into S = 2..4 equal slices, slice s
       of every tile is one workgroup of the 256 x 192 stream writing its fp32 partial to workspace[s], and a second
       launch adds the slices in the order s = 0..S-1 (no atomics: bit-repeatable), the bias, and applies the epilogue
       (OMH_EPI_RESID with c_in / aux / gates, OMH_EPI_F32).  S depends on (M, N, K) only.  NULL / too small: the
       unsplit kernels, as before.  16-byte aligned. */
    void* workspace; int64_t workspace_bytes;
    /* ABI v10 — OMH_EPI_BF16_SPLIT_T: first column that goes (transposed) to aux. */
    int32_t n_split;
} omh_gemm_args;

int omh_gemm_bf16(const omh_gemm_args* args, omh_stream_t stream);
/* Bytes of `workspace` with which omh_gemm_bf16 would split the contraction of this product (0: it would not). */
int64_t omh_gemm_workspace_bytes(const omh_gemm_args* args);

/* bf16 GEMM with both operands K-MAJOR (row-major [K, *]):  C[m][n] (+)= sum_k A[k][m] * B[k][n],  fp32 C.
 * The weight gradient of every nn.Linear in the training step — dW[out][in] = sum_r dy[r][out] x[r][in], what
 * autograd computes for model.py's Linears under distilled_trainer.py:289-301 — on dy and x as the backward
 * produces them (no transposed copies).  M, N, lda, ldb multiples of 8; K arbitrary; accumulate != 0 adds to C. */
typedef struct omh_gemm_tn_args {
    const void* A; const void* B; float* C;
    int32_t M, N, K;
    int32_t lda, ldb, ldc;
    int32_t accumulate;
} omh_gemm_tn_args;

int omh_gemm_bf16_tn(const omh_gemm_tn_args* args, omh_stream_t stream);

/* Up to OMH_TN_GROUP_MAX such products in ONE launch: all the weight gradients of one block's backward (q|k|v, o,
 * cross q, cross k|v, cross o, FFN 1, FFN 2).  Each alone is 144 ... 840 tiles of 128 x 128 over a contraction of 6 240
 * rows and leaves most of the chip idle or needs a split K with fp32 atomics; together they fill it, every tile runs
 * its whole K range, and the result is bit-repeatable.  first_tile / total_tiles are scratch filled by the library.
 * Products with M % 256 == 0 (every weight of the DiT) run on the 256 x 384 k-major stream kernel (gemm_tn_w64.hip:
 * persistent workgroups over the tiles of the whole group, 192 tiles for a block's attention weights = one round of
 * the chip) when the group is worth a launch; same MFMA, same order over k, same bits as the 128 x 128 tiles.
 * OMH_GEMM_TN_W64 = 0 / 1 forces the choice (also for omh_gemm_bf16_tn). */
#define OMH_TN_GROUP_MAX 12
typedef struct omh_gemm_tn_group {
    int32_t n;
    omh_gemm_tn_args problem[OMH_TN_GROUP_MAX];
    int32_t first_tile[OMH_TN_GROUP_MAX]; int32_t total_tiles;
} omh_gemm_tn_group;
int omh_gemm_bf16_tn_grouped(const omh_gemm_tn_group* group, omh_stream_t stream);

/* ------------------------------------------------------------------------
 * Flash attention forward, head_dim 128, bf16, non-causal, key-length mask.
 * Replaces flash_attn.flash_attn_varlen_func as called from
 * attention.py:96-127 for self-attention (model.py:151-156) and
 * cross-attention (model.py:181,221-223): out[b,i,h,:] =
 * softmax_j<k_lens[b]( q[b,i,h,:].k[b,j,h,:] * scale ) v[b,j,h,:].
 *   q  : [B, Lq, H, 128] bf16, element strides (q_bs, q_rs), head h at +h*128
 *   k  : [B, Lk, H, 128] bf16, strides (k_bs, k_rs)
 *   vt : V transposed, [B, H*128, ldv] bf16 (row = h*128+d, column = key);
 *        ldv >= roundup(Lk,64), columns >= Lk must hold finite values
 *   o  : [B, Lq, H, 128] bf16, strides (o_bs, o_rs)
 *   k_lens: int32 [B] device pointer or NULL (= Lk); rows with k_lens 0 give 0.
 * ---------------------------------------------------------------------- */
typedef struct omh_attn_args {
    const void* q; const void* k; const void* vt; void* o;
    const int32_t* k_lens;
    int32_t B, H, Lq, Lk;
    int64_t q_bs, q_rs, k_bs, k_rs, vt_bs, o_bs, o_rs;
    int32_t ldv;
    float scale;                  /* softmax scale, 1/sqrt(128) in the reference  */
    float* lse;                   /* optional [B,H,Lq] fp32 log-sum-exp (natural log) for backward; may be NULL */
    /* != 0: q already carries the factor scale*log2(e) (folded into the producing norm kernel, see
       omh_rmsnorm_rope_bf16 out_scale: one rounding to bf16 instead of two); `scale` is then only documentation.
       0: the long-sequence kernel multiplies q by scale*log2(e) itself and re-rounds it to bf16 (a 2^-9 relative
       perturbation of q); the short-sequence kernels apply the factor to the fp32 scores.                        */
    int32_t q_prescaled;
    /* Optional scratch for the long-sequence kernel's split-KV tail (omh_flash_attn_workspace_bytes() bytes, 16-byte
       aligned, contents irrelevant): when the number of 256-query tiles is a multiple of the 256 CUs plus a small
       remainder, the remainder is split over the keys instead of costing a whole extra round.  NULL: no split. */
    void* workspace; int64_t workspace_bytes;
    /* ABI v5, optional: the output once more in fp32, [B, Lq, H, 128] with o's strides (o_bs, o_rs, in elements),
       16-byte aligned — the training step keeps it for the backward's delta_i = sum_d dO_id O_id (a bf16 O there
       leaves a 2^-9 residue in every row sum of dS).  Served by the short-sequence kernel only: a call that sets it
       does not take the long-sequence kernels. */
    float* o32;
    /* ABI v8.  OMH_ATTN_SHORT_KERNEL: run the short-sequence kernel whatever the shape (the training forward and its
       re-run under use_checkpoint must take the same kernel whether or not they ask for lse / o32).
       OMH_ATTN_ALLOW_SPLIT: the short-sequence kernel may split the LAST, partly filled round of its workgroups over the
       keys (workers write fp32 partial results into `workspace`, a small kernel combines them — the flash-decoding
       reduction): 624 workgroups on 512 slots (4 clips x 12 heads x 13 query tiles, the training step) then take 1.25
       rounds instead of 2.  The split plan depends on B, so a sample's last bits depend on its batch: only callers that do
       not need batch invariance set it (the training step; the inference path does not). */
    int32_t flags;
    /* ABI v10, optional: int32 [B] device pointer or NULL (= Lq).  The reference's flash_attention(q_lens=...)
       (attention.py:24-60,79-80: queries past q_lens[b] are cut out of the packed varlen batch): output rows
       i >= q_lens[b] are written as ZERO (o, o32; lse = -inf).  Served by the short-sequence kernel, unsplit. */
    const int32_t* q_lens;
    /* ABI v12: the band of flash_attention(causal=, window_size=(left, right)) (attention.py:24-60,96-127, flash-attn's
       bottom-right aligned local attention): query i of a sample with qlen = q_lens[b] (or Lq) queries and
       klen = k_lens[b] (or Lk) keys attends key j iff  i + klen - qlen - window_left <= j <= i + klen - qlen + window_right.
       A side < 0 is unbounded: (-1, -1) = full attention (the only setting the reference's own callers use; the one
       the long-sequence stream and the backward serve), causal = (left, 0).  Rows whose band holds no key are written as
       zero (lse = -inf).  A bounded side selects the short-sequence kernel, unsplit; forward only. */
    int32_t window_left, window_right;
} omh_attn_args;
#define OMH_ATTN_SHORT_KERNEL 1
#define OMH_ATTN_ALLOW_SPLIT  2

int omh_flash_attn_fwd_d128(const omh_attn_args* args, omh_stream_t stream);
/* Scratch size omh_flash_attn_fwd_d128 can use for these shapes (0: none needed). */
int64_t omh_flash_attn_workspace_bytes(const omh_attn_args* args);

/* ------------------------------------------------------------------------
 * Flash attention backward, head_dim 128 (training step: the autograd of
 * flash_attn_varlen_func under the per-block checkpoint, model.py:544-548,
 * distilled_trainer.py:289-301).  With P = exp(scale q k^T - lse) over keys
 * < k_lens[b]:  dP = dO V^T,  dV = P^T dO,  dS = P (dP - rowsum(P dP)),
 * dQ = scale dS K,  dK = scale dS^T Q.   No atomics: repeatable bit for bit.
 *   q, dout    : [B, Lq, H, 128] bf16 (strides q_bs,q_rs / o_bs,o_rs); o: unused (may be NULL)
 *   k, v       : [B, Lk, H, 128] bf16 (k_bs, k_rs)
 *   qt, dot    : Q^T and dO^T, [B, H*128, ldq] bf16 (batch stride qt_bs), kt: K^T [B, H*128, ldk] (kt_bs);
 *                ldq >= roundup(Lq,32), ldk >= roundup(Lk,32), pad columns ZERO (omh_transpose_bf16 into a
 *                zero-filled buffer)
 *   lse        : [B, H, Lq] fp32 from omh_flash_attn_fwd_d128;  delta: [B, H, Lq] fp32 workspace (written)
 *   dq         : [B, Lq, H, 128] fp32 (dq_bs, dq_rs);  dk, dv: [B, Lk, H, 128] fp32 (dk_bs, dk_rs); fully written
 * ---------------------------------------------------------------------- */
typedef struct omh_attn_bwd_args {
    const void* q; const void* k; const void* v; const void* o; const void* dout;
    const void* qt; const void* dot; const void* kt;
    const float* lse; float* delta;
    void* dq; void* dk; void* dv;          /* fp32, or bf16 with out_bf16 */
    const int32_t* k_lens;
    int32_t B, H, Lq, Lk;
    int64_t q_bs, q_rs, k_bs, k_rs, o_bs, o_rs, dq_bs, dq_rs, dk_bs, dk_rs, qt_bs, kt_bs;
    int32_t ldq, ldk;
    float scale;
    /* ABI v5.  q_prescaled != 0: q (and qt) carry the factor scale*log2(e) as in the forward call
       (omh_attn_args.q_prescaled): P is recomputed from the very operands the forward used; dq is still the gradient
       with respect to the UNSCALED normalised q (what omh_rmsnorm_rope_bwd expects), dk = dS^T q' / log2(e).
       out_bf16 != 0: dq / dk / dv are bf16 (strides in bf16 elements, multiples of 4) — the operand type of the
       weight-gradient GEMMs that follow, no cast pass. */
    int32_t q_prescaled, out_bf16;
    /* o32 != NULL: the forward's fp32 output (omh_attn_args.o32, strides o_bs / o_rs).  Selects the round-3 kernels
       (csrc/attention_bwd2.hip): delta = rowsum(dO * o32) in the dQ kernel's prologue instead of a first pass over the
       keys, 64-position tiles staged by LDS-DMA into a double buffer, the transposed operands read from the
       row-major tiles with ds_read_b64_tr_b16 — qt / dot / kt are not read (may be NULL). */
    const float* o32;
    /* ABI v7 (round-3 kernels only, o32 != NULL).  phase 0: everything (the dQ kernel computes delta, then dK/dV).
       1: delta only (a small kernel, the same arithmetic as the dQ kernel's prologue); 2: dQ only, reading the delta a
       phase-1 call left in `delta`; 3: dK/dV only, likewise.  Phases 2 and 3 are independent of each other: a caller
       may issue them on two streams (the dQ kernel's 624 workgroups on 512 slots and the dK/dV kernel's 624 on 256
       each leave most of their last round idle at 4 clips x 1560 positions). */
    int32_t phase;
    /* ABI v8 (round-3 kernels only).  Optional scratch, omh_flash_attn_bwd_workspace_bytes() bytes, 16-byte aligned:
       with it the last, partly filled round of workgroups of the dQ kernel is split over the keys and that of the
       dK/dV kernel over the queries; the workers' fp32 partial sums are added in a fixed order by a small kernel (no
       atomics: still repeatable bit for bit for a given shape).  NULL: no split. */
    void* workspace; int64_t workspace_bytes;
} omh_attn_bwd_args;

int omh_flash_attn_bwd_d128(const omh_attn_bwd_args* args, omh_stream_t stream);
/* Scratch size the call can use (0: none needed / no split for these shapes); depends on B, H, Lq, Lk, phase. */
int64_t omh_flash_attn_bwd_workspace_bytes(const omh_attn_bwd_args* args);

/* ------------------------------------------------------------------------
 * LayerNorm (no affine) fused with adaLN modulation, fp32 in -> bf16 out.
 * Replaces WanLayerNorm + "x*(1+scale)+shift" (model.py:91-104,292-293,
 * 314-315,358) and the affine norm3 (model.py:263-265,313).
 *   y[r][c] = xhat[r][c] * (mul_const + mul0[c] + mul1[b*mul1_stride + c])
 *                        + (add0[c] + add1[b*add1_stride + c]),  b = r / rows_per_batch
 * any of mul0/mul1/add0/add1 may be NULL.  dim % 4 == 0, dim <= 8192.
 * ---------------------------------------------------------------------- */
int omh_layernorm_modulate(const float* x, void* y_bf16, int64_t rows, int32_t dim, float eps,
                           float mul_const, const float* mul0, const float* mul1, int64_t mul1_stride,
                           const float* add0, const float* add1, int64_t add1_stride,
                           int64_t rows_per_batch, omh_stream_t stream);

/* ------------------------------------------------------------------------
 * WanRMSNorm over the full model dim (+ optional 3-axis RoPE), fp32 -> bf16.
 * Replaces WanRMSNorm (model.py:72-88) and rope_apply (model.py:42-69) on
 * the q/k projections (model.py:144-145,152-153,176-177,216-219).
 *   x: [rows, ldx] fp32 (first `dim` columns used); y: [rows, dim] bf16
 *   weight: [dim] fp32 or NULL (no gain, used when qk_norm is off -> plain cast)
 *   do_norm: 0 = skip normalisation (Identity), 1 = RMS-normalise
 *   rope: cos/sin tables [rope_len, head_dim/2] fp32 or NULL for no rotation;
 *   grid: int32 [B,3] (f,h,w) on the device; rows = B*seq_len, token s of
 *   sample b sits at row b*seq_len+s; tokens >= f*h*w are not rotated.
 *   The first cf = c-2*(c/3) pair-columns use f, next c/3 use h, last c/3 use w.
 * ---------------------------------------------------------------------- */
int omh_rmsnorm_rope(const float* x, int64_t ldx, void* y_bf16, int64_t rows, int32_t dim,
                     const float* weight, float eps, int32_t do_norm,
                     const float* rope_cos, const float* rope_sin, int32_t rope_len,
                     int32_t head_dim, const int32_t* grid, int32_t seq_len, omh_stream_t stream);

/* Same with a bf16 input [rows, ldx] (the inference path keeps the q|k projection in bf16: half the
 * HBM traffic of this HBM-bound step; the training recompute uses the fp32 form above).  out_scale multiplies the
 * result in fp32 before the rounding to bf16: the self-attention q is produced with out_scale =
 * head_dim^-0.5 * log2(e) (attention.py:96-127 softmax_scale), so that the attention kernel's exponentials need no
 * per-score multiply (omh_attn_args.q_prescaled). */
int omh_rmsnorm_rope_bf16(const void* x_bf16, int64_t ldx, void* y_bf16, int64_t rows, int32_t dim,
                          const float* weight, float eps, int32_t do_norm,
                          const float* rope_cos, const float* rope_sin, int32_t rope_len,
                          int32_t head_dim, const int32_t* grid, int32_t seq_len, float out_scale,
                          omh_stream_t stream);
/* ABI v9: omh_rmsnorm_rope_bf16 on TWO column segments of the same rows in one launch — q and k of the self-attention out
 * of the fused q|k projection (model.py:144-145: norm_q / norm_k + rope_apply): segment 1 reads x + seg_x elements, with
 * its own gain, output and output scale.  The same row arithmetic as two calls (same bits). */
int omh_rmsnorm_rope_bf16_pair(const void* x_bf16, int64_t ldx, int64_t seg_x, void* y0_bf16, void* y1_bf16, int64_t rows,
                               int32_t dim, const float* weight0, const float* weight1, float eps, int32_t do_norm,
                               const float* rope_cos, const float* rope_sin, int32_t rope_len, int32_t head_dim,
                               const int32_t* grid, int32_t seq_len, float out_scale0, float out_scale1, omh_stream_t stream);

/* fp32 -> bf16 cast (round to nearest even) of a contiguous buffer. */
int omh_cast_f32_bf16(const float* x, void* y_bf16, int64_t n, omh_stream_t stream);

/* ------------------------------------------------------------------------
 * Patchify: latent [C,F,H,W] fp32 -> token matrix [f*h*w, Kp] bf16 with
 * column (c*pt + a)*ph*pw + i*pw + j  = x[c, f*pt+a, h*ph+i, w*pw+j]; columns
 * >= C*pt*ph*pw (up to Kp) are zero.  With the Conv3d weight flattened to
 * [dim, C*pt*ph*pw] this turns patch_embedding (model.py:463,515-518) into a GEMM.
 * ---------------------------------------------------------------------- */
int omh_patchify(const float* x, void* tokens_bf16, int32_t C, int32_t F, int32_t H, int32_t W,
                 int32_t pt, int32_t ph, int32_t pw, int32_t Kp, omh_stream_t stream);

/* Unpatchify (model.py:565-588): head output [f*h*w, pt*ph*pw*Cout] fp32 ->
 * [Cout, f*pt, h*ph, w*pw] fp32, einsum 'fhwpqrc->cfphqwr'. */
int omh_unpatchify(const float* tokens, float* out, int32_t Cout, int32_t f, int32_t h, int32_t w,
                   int32_t pt, int32_t ph, int32_t pw, omh_stream_t stream);

/* ------------------------------------------------------------------------
 * Small fp32 dense layer for the time embedding (model.py:469-471,526-528):
 *   y[b][n] = act_out( sum_k act_in(x[b][k]) * W[n][k] + bias[n] )
 * act: 0 none, 1 SiLU.  One wave per output element; B is tiny (batch).
 * ---------------------------------------------------------------------- */
int omh_dense_f32(const float* x, const float* W, const float* bias, float* y,
                  int32_t B, int32_t N, int32_t K, int32_t act_in, int32_t act_out,
                  omh_stream_t stream);

/* sinusoidal_embedding_1d (model.py:17-27), computed in fp64, stored fp32:
 * out[b] = [cos(t_b * 10000^(-i/half)) | sin(...)], i < half = dim/2. */
int omh_sinusoidal_embedding(const float* t, float* out, int32_t B, int32_t dim, omh_stream_t stream);

/* ------------------------------------------------------------------------
 * Classifier-free guidance + one flow-matching UniPC (bh2, predict-x0) step
 * on the latent, fused into one pass (text2video.py:243-252;
 * fm_solvers_unipc.py:279-331 convert_model_output, :350-484 predictor,
 * :486-626 corrector, :655-739 step).  Scalar coefficients come from the
 * sigma schedule on the host.  All tensors fp32, n elements.
 *   v      = uncond + guide * (cond - uncond)
 *   m_t    = x - sigma * v                                   -> mt_out
 *   x_c    = use_corr ? ca_last*last + ca_m1*m1 + ca_m2*m2 + ca_mt*m_t : x   -> xc_out
 *   x_next = pb_x*x_c + pb_mt*m_t + pb_m1*m1                  -> x_next
 * m1 / m2 (the previous two x0 predictions) and `last` may be NULL when their
 * coefficient is 0.  xc_out may alias `last`; x_next may alias x.
 * ---------------------------------------------------------------------- */
int omh_cfg_unipc_step(const float* cond, const float* uncond, const float* x, const float* last,
                       const float* m1, const float* m2, float* mt_out, float* xc_out, float* x_next,
                       int64_t n, float guide, float sigma, int32_t use_corr,
                       float ca_last, float ca_m1, float ca_m2, float ca_mt,
                       float pb_x, float pb_mt, float pb_m1, omh_stream_t stream);

/* ----------------------------------------------------------------------
 * ABI v11.  Measurement, not product: TFLOP/s this GPU sustains on back-to-back
 * v_mfma_f32_32x32x16_bf16 with one wave per SIMD and no memory traffic
 * (one workgroup of 4 waves per CU, iters x 64 MFMAs each), with constant
 * operands (random_operands = 0) or with operands that toggle like data (1):
 * the clock follows the power the multipliers draw, so the second figure is
 * the matrix-pipe ceiling a kernel on real data can be measured against.
 * bench.py reports both beside the data-sheet peak.  scratch: >= 256 floats
 * per CU of device memory (overwritten).  Synchronises `stream`.
 * ---------------------------------------------------------------------- */
int omh_probe_mfma_tflops(int32_t random_operands, int32_t iters, float* scratch, int64_t scratch_floats,
                          float* tflops_out, omh_stream_t stream);

/* ABI v12.  A HIP stream confined to a subset of the CUs (hipExtStreamCreateWithCUMask), for the training step's
 * second stream (the weight-gradient GEMMs that run beside the backward of distilled_trainer.py:289-301): cus_per_32 of
 * every 32 consecutive CU-mask bits are enabled, from the low end (high = 0) or the high end (high != 0), so that a
 * "low k" and a "high 32 - k" stream partition the chip and every XCD contributes equally.  The caller owns the stream
 * (omh_stream_destroy).  Measurement / scheduling aid: no kernel of the library depends on it. */
int omh_stream_create_cu_mask(int32_t cus_per_32, int32_t high, omh_stream_t* stream_out);
int omh_stream_destroy(omh_stream_t stream);

/* ========================================================================
 * 3D causal VAE (seaweed_apt/wan/modules/vae.py).  Activations are
 * channels-last bf16 [T, H, W, C]; a causal conv's temporal history (the
 * reference's feat_cache slots, vae.py:205-217) is the leading frames of its
 * input buffer.
 * ====================================================================== */

/* Implicit-GEMM convolution (CausalConv3d 3x3x3 / 3x1x1 / 1x1x1, Conv2d 3x3
 * stride 1 or 2, optional folded nearest-2x upsample; vae.py:17-36,76-96).
 *   y[to][yo][xo][co] = bias[co] + resid[...] + sum w[co][((kt*KH+dy)*KW+dx)*Cin+ci] *
 *                       x[to*stride_t + kt][S(yo*stride_hw + dy - pad_h)][S(xo*stride_hw + dx - pad_w)][ci]
 *   S(i) = i (or i>>1 with up2), zero outside [0, Hin) x [0, Win) (x2 with up2).
 *   split_n > 0: the Cout channels are Cout/split_n consecutive output FRAMES
 *   of split_n channels each (temporal upsample interleave, vae.py:134-137).
 * Cin % 8 == 0; (Tout-1)*stride_t + KT <= Tin. */
typedef struct omh_conv_args {
    const void* x; const void* w; const float* bias; const void* resid; void* y;
    int32_t Tin, Hin, Win, Cin;
    int32_t Tout, Hout, Wout, Cout;
    int32_t KT, KH, KW;
    int32_t stride_t, stride_hw, pad_h, pad_w;
    int32_t up2, out_f32, split_n;
    int32_t resid_f32;            /* != 0: resid is fp32 (the fp32 residual trunk of the VAE executor, ABI v4) */
    /* ABI v7: the NEXT layer's RMS_norm + SiLU on this convolution's output (vae.py:39-54 under :195-197, :203-205),
     * fused into the producing kernel's epilogue where a workgroup holds all Cout channels of a voxel (the stream
     * kernel at Cout = 96) and run as omh_rms_silu_cl(_f32in) on y behind the convolution everywhere else — same
     * values either way.  norm_gamma: fp32 [Cout] or NULL (no norm); norm_out: bf16 [Tout,Hout,Wout,Cout];
     * norm_only != 0: the caller does not need y itself (still a valid buffer: the un-fused route goes through it). */
    const float* norm_gamma; void* norm_out; int32_t norm_only;
    /* ABI v10 — the fp32-faithful VAE mode's convolution (the reference's VAE computes in fp32, vae.py:619-624,649-663)
     * on split-bf16 PAIRS: != 0 says that x and w carry, per 16 real channels c0..c0+15, the 32 values
     * [hi(c0..c0+15) | lo(c0..c0+15)] with hi = bf16(v), lo = bf16(v - hi) (omh_split3_f32 pattern 2), i.e. Cin = 2 x the
     * real channel count, and that the product wanted is x_hi w_hi + x_lo w_hi + x_hi w_lo per channel (= x w - x_lo w_lo,
     * an fp32-class product with fp32 accumulation).  Served by the stream kernel only (3x3(x3) "same" convolutions of
     * the residual blocks and the folded-upsample convolutions, fp32 output, Cin % 32 == 0, Cout = 96 or a multiple of
     * 192): three MFMA products per channel block that share their LDS fragments — a third less staging, LDS reads and
     * barriers per MFMA than the same product as a 3 C-channel convolution over [hi | lo | hi] x [hi | hi | lo]
     * (pattern 0 / 1, which every other layer keeps).  omh_conv_pair_supported() says whether a call would be taken
     * (by the layer's geometry only, never by the number of frames); omh_conv_cl_bf16 returns OMH_E_SHAPE otherwise. */
    int32_t pair;
} omh_conv_args;

int omh_conv_cl_bf16(const omh_conv_args* args, omh_stream_t stream);
/* 1 if omh_conv_cl_bf16 would take `args` with args->pair != 0 (the pointers are not dereferenced), else 0. */
int omh_conv_pair_supported(const omh_conv_args* args);

/* RMS_norm over channels (+ optional SiLU) per voxel (vae.py:39-54,195-197):
 *   y[p][c] = act( x[p][c] / max(||x[p]||_2, 1e-12) * sqrt(C) * gamma[c] ),  x,y bf16 [P, C]. */
int omh_rms_silu_cl(const void* x_bf16, const float* gamma, void* y_bf16, int64_t P, int32_t C,
                    int32_t do_silu, omh_stream_t stream);
/* The same on an fp32 input (the residual trunk, which the reference keeps in fp32: WanVAE(dtype=torch.float),
 * vae.py:619-624): statistics and normalisation in fp32, one rounding to bf16 at the convolution's input. */
int omh_rms_silu_cl_f32in(const float* x_f32, const float* gamma, void* y_bf16, int64_t P, int32_t C,
                          int32_t do_silu, omh_stream_t stream);

/* In-place ReLU on a bf16 buffer (n % 8 == 0): the nn.ReLU between the Conv3d layers of the OmniHuman pose
 * guider (Omnihuman/omnihuman_wan_t2v.py:37-45,149-157), whose convolutions run on omh_conv_cl_bf16.
 * A negative NaN becomes +0 as well (torch.relu would keep it): inputs here are finite conv outputs. */
int omh_relu_bf16(void* x_bf16, int64_t n, omh_stream_t stream);
/* Its backward for the training step of the pose guider (omnihuman_wan_t2v.py:453-488 through :149-157):
 * g = y > 0 ? dy : 0 with y the ReLU's output; bf16, n % 8 == 0. */
int omh_relu_bwd_bf16(const void* dy_bf16, const void* y_bf16, void* g_bf16, int64_t n, omh_stream_t stream);

/* Layout/precision converts at the VAE boundary (vae.py:547-553, 535-540, 661):
 *   nchw_to_cl : y[t][h][w][c] = bf16( x[c][t0+t][h][w] * mul[c] + add[c] ), c < C; channels C..Cp-1 zero
 *   cl_to_nchw : y[c][t0+t][h][w] = clamp( (x[t][h][w][c] + add[c]) * mul[c], lo, hi ), x fp32 with Cp channels
 * The NCTHW tensor has t_total frames; the channels-last chunk covers frames [t0, t0+T).
 * mul/add may be NULL (1 / 0). */
int omh_nchw_to_cl(const float* x, void* y_bf16, int32_t C, int32_t T, int32_t H, int32_t W, int32_t Cp,
                   const float* mul, const float* add, int32_t t_total, int32_t t0, omh_stream_t stream);
int omh_cl_to_nchw(const float* x, float* y, int32_t C, int32_t T, int32_t H, int32_t W, int32_t Cp,
                   const float* mul, const float* add, float lo, float hi, int32_t t_total, int32_t t0,
                   omh_stream_t stream);

/* ---- ABI v8: the fp32-faithful VAE mode (the reference computes its VAE in fp32: WanVAE(dtype=torch.float) ->
 * amp.autocast(dtype=self.dtype), vae.py:619-624,649-663).  Every convolution / GEMM operand is carried as a bf16
 * PAIR hi = bf16(x), lo = bf16(x - hi) (x = hi + lo to 2^-17), laid out as three channel blocks so that the UNCHANGED
 * bf16 MFMA kernels compute an fp32-class product as a contraction three times as long:
 *   activations  [hi | lo | hi]   (pattern 0)        weights  [hi | hi | lo]   (pattern 1)
 *   sum over the 3 Cp channels = x_hi w_hi + x_lo w_hi + x_hi w_lo = x w - x_lo w_lo      (fp32 accumulate)
 * omh_split3_f32: x fp32 [rows, C] (pitch ldx) -> y bf16 [rows, 3 Cp] (pitch ldy >= 3 Cp), channels C..Cp-1 zero;
 *   Cp % 4 == 0.  omh_rms_silu_cl_split3: omh_rms_silu_cl_f32in with a pattern-0 result [P, 3 C] (IEEE exp /
 *   division).  omh_nchw_to_cl_f32 / omh_softmax_rows_f32: the boundary convert and the row softmax with fp32 results
 *   (then omh_split3_f32). */
int omh_split3_f32(const float* x, int64_t ldx, void* y_bf16, int64_t ldy, int64_t rows, int32_t C, int32_t Cp,
                   int32_t pattern, omh_stream_t stream);
int omh_rms_silu_cl_split3(const float* x_f32, const float* gamma, void* y_bf16x3, int64_t P, int32_t C,
                           int32_t do_silu, omh_stream_t stream);
/* ABI v10 — the PAIR layout of omh_conv_args.pair: omh_split3_f32 with pattern 2 writes y bf16 [rows, 2 Cp] (pitch
 * ldy >= 2 Cp, Cp % 16 == 0): per 16 channels c0.. the 32 values [hi(c0..c0+15) | lo(c0..c0+15)] — activations and
 * weights alike; omh_rms_silu_cl_pair is omh_rms_silu_cl_split3 with that result layout [P, 2 C] (C % 16 == 0). */
int omh_rms_silu_cl_pair(const float* x_f32, const float* gamma, void* y_bf16x2, int64_t P, int32_t C,
                         int32_t do_silu, omh_stream_t stream);
int omh_nchw_to_cl_f32(const float* x, float* y, int32_t C, int32_t T, int32_t H, int32_t W, int32_t Cp,
                       const float* mul, const float* add, int32_t t_total, int32_t t0, omh_stream_t stream);
int omh_softmax_rows_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t R, int32_t L, float scale,
                         omh_stream_t stream);

/* Row softmax for the VAE's single-head mid-block attention (vae.py:252):
 *   y[r][j] = bf16( softmax_j( x[r][j] * scale ) ), x fp32 [R, L] (ldx), y bf16 (ldy). */
int omh_softmax_rows(const float* x, int64_t ldx, void* y_bf16, int64_t ldy, int64_t R, int32_t L, float scale,
                     omh_stream_t stream);

/* ========================================================================
 * Backward pass of the DiT (training step, seaweed_apt/distilled_trainer.py:
 * 241-316; the reference obtains it from autograd over aten).  Matrix products
 * (dgrad / wgrad / attention dQ,dK,dV) reuse omh_gemm_bf16 on transposed
 * operands; gradients of parameters and modulation vectors are ACCUMULATED
 * (fp32 atomics) into caller-zeroed buffers.
 * ====================================================================== */

/* out[b][c][r] = in[b][r][c], bf16; ld_out >= R (pad columns untouched). */
int omh_transpose_bf16(const void* in, void* out, int32_t R, int32_t C, int64_t ld_in, int64_t ld_out,
                       int32_t batch, int64_t bs_in, int64_t bs_out, omh_stream_t stream);
/* out[c] += sum_r x[r][c]  (x bf16 or fp32) — bias gradients. */
int omh_colsum_accum(const void* x, int32_t is_bf16, int64_t ld, float* out, int64_t R, int32_t C, omh_stream_t stream);
/* The same for up to OMH_COLSUM_MAX matrices in one launch (all bias gradients of one block backward).
 * blocks[] is scratch filled by the library. */
#define OMH_COLSUM_MAX 16
typedef struct omh_colsum_batch {
    int32_t n;
    const void* x[OMH_COLSUM_MAX]; float* out[OMH_COLSUM_MAX];
    int64_t ld[OMH_COLSUM_MAX]; int64_t R[OMH_COLSUM_MAX];
    int32_t C[OMH_COLSUM_MAX]; int32_t is_bf16[OMH_COLSUM_MAX]; int32_t blocks[OMH_COLSUM_MAX];
} omh_colsum_batch;
int omh_colsum_accum_multi(const omh_colsum_batch* batch, omh_stream_t stream);
/* GELU-tanh on bf16 (model.py:273) and its backward dx = dy * gelu'(x_pre). */
int omh_gelu_tanh_bf16(const void* x, void* y, int64_t n, omh_stream_t stream);
int omh_gelu_tanh_bwd_bf16(const void* dy, const void* x_pre, void* dx, int64_t n, omh_stream_t stream);

/* exact (erf) GELU, the activation of the i2v image-embedding MLP (model.py:366), forward and backward — the
 * training step of the i2v backbone: y = 0.5 x (1 + erf(x/sqrt 2)); dx = dy (Phi(x) + x phi(x)).  bf16, n elements. */
int omh_gelu_erf_bf16(const void* x_bf16, void* y_bf16, int64_t n, omh_stream_t stream);
int omh_gelu_erf_bwd_bf16(const void* dy_bf16, const void* x_pre_bf16, void* dx_bf16, int64_t n, omh_stream_t stream);
/* Gated residual (model.py:296,328): xo = xi + y*gate; gate = gate_const + gate0[c] + gate1[b*stride + c].
 * bwd: dy = bf16(dx*gate); dgate[b*dgate_stride + c] += sum_rows dx*y (skipped when y or dgate is NULL). */
int omh_gated_residual_fwd(const float* xi, const void* y_bf16, float* xo, int64_t rows, int32_t dim,
                           float gate_const, const float* gate0, const float* gate1, int64_t gate1_stride,
                           int64_t rows_per_batch, omh_stream_t stream);
int omh_gated_residual_bwd(const float* dx, const void* y_bf16, void* dy_bf16, float* dgate, int64_t dgate_stride,
                           int64_t rows, int32_t dim, float gate_const, const float* gate0, const float* gate1,
                           int64_t gate1_stride, int64_t rows_per_batch, omh_stream_t stream);
/* Backward of omh_layernorm_modulate: dx_accum += dL/dx; dmul[b*dstride+c] += dy*xhat; dadd[..] += dy. */
int omh_layernorm_modulate_bwd(const float* x, const float* dy, float* dx_accum, int64_t rows, int32_t dim,
                               float eps, float mul_const, const float* mul0, const float* mul1, int64_t mul1_stride,
                               float* dmul, float* dadd, int64_t dstride, int64_t rows_per_batch, omh_stream_t stream);
/* Backward of omh_rmsnorm_rope: dy fp32 [rows, dim] (grad of the rotated, normalised output) ->
 * dx bf16 [rows, lddx]; dweight[c] += ... (may be NULL). */
int omh_rmsnorm_rope_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, void* dx_bf16, int64_t lddx,
                         float* dweight, int64_t rows, int32_t dim, const float* weight, float eps, int32_t do_norm,
                         const float* rope_cos, const float* rope_sin, int32_t rope_len, int32_t head_dim,
                         const int32_t* grid, int32_t seq_len, omh_stream_t stream);
/* The same with x and / or dy in bf16 (x_bf16 / dy_bf16 != 0; strides in elements, multiples of 4): the backward of
 * omh_rmsnorm_rope_bf16 on the very bf16 projection the forward normalised (model.py:144-145 under
 * distilled_trainer.py:289-301), fed by omh_flash_attn_bwd_d128 with out_bf16.  dy is the gradient of the UNSCALED
 * output (out_scale not applied).  dx may alias dy (a row is read whole before it is written). */
int omh_rmsnorm_rope_bwd_t(const void* x, int32_t x_bf16, int64_t ldx, const void* dy, int32_t dy_bf16, int64_t lddy,
                           void* dx_bf16, int64_t lddx, float* dweight, int64_t rows, int32_t dim, const float* weight,
                           float eps, int32_t do_norm, const float* rope_cos, const float* rope_sin, int32_t rope_len,
                           int32_t head_dim, const int32_t* grid, int32_t seq_len, omh_stream_t stream);
/* Round-3 forms of the two norm backwards (csrc/dit_backward2.hip): a workgroup takes 8 rows of ONE batch element,
 * per-column parameter sums are combined in a fixed order (registers -> LDS -> one partial per workgroup in
 * `workspace` -> a second launch adds the partials of a column in workgroup order): no atomics, bit-repeatable.
 *
 * omh_layernorm_modulate_bwd2 = omh_layernorm_modulate_bwd (dy fp32 or bf16) and, optionally, the gated-residual
 * backward of the NEXT branch (omh_gated_residual_bwd) applied to each row of dx as soon as it is final:
 *   dy_next[r][c] = bf16(dx[r][c] * gate),  gate = gate_const + gate0[c] + gate1[b * gate1_stride + c]
 *   dgate[b * dgate_stride + c] += sum_rows dx * y_next            (only with y_next and dgate)
 * workspace: omh_layernorm_modulate_bwd2_workspace(rows, dim, rows_per_batch) floats, 16-byte aligned. */
/* ABI v9: the second launch of one of the two functions below, handed back to the caller instead of being issued
 * (`deferred` != NULL in the argument struct: the struct is filled in, the partials stay in `workspace`, which the caller
 * keeps alive): a block's backward collects the five it produces on its main stream and issues them as ONE launch with
 * omh_partial_colsum_multi — same sums in the same order, four launches fewer per block. */
typedef struct omh_partial_reduce {
    const float* part; int32_t nj, np, nb, dim, grid_y;
    float* out[3]; int64_t stride[3];
} omh_partial_reduce;
#define OMH_PARTIAL_REDUCE_MAX 8
typedef struct omh_partial_reduce_batch { int32_t n; omh_partial_reduce e[OMH_PARTIAL_REDUCE_MAX]; } omh_partial_reduce_batch;
int omh_partial_colsum_multi(const omh_partial_reduce_batch* batch, omh_stream_t stream);

typedef struct omh_ln_bwd_args {
    const float* x; const void* dy; int32_t dy_bf16; float* dx;
    int64_t rows; int32_t dim; float eps; float mul_const;
    const float* mul0; const float* mul1; int64_t mul1_stride;
    float* dmul; float* dadd; int64_t dstride; int64_t rows_per_batch;
    void* dy_next; const void* y_next; float gate_const; const float* gate0; const float* gate1; int64_t gate1_stride;
    float* dgate; int64_t dgate_stride;
    float* workspace; int64_t workspace_floats;
    omh_partial_reduce* deferred;          /* ABI v9; NULL: the partials are added here */
} omh_ln_bwd_args;
int64_t omh_layernorm_modulate_bwd2_workspace(int64_t rows, int32_t dim, int64_t rows_per_batch);
int omh_layernorm_modulate_bwd2(const omh_ln_bwd_args* args, omh_stream_t stream);

/* omh_rmsnorm_rope_bwd_t on n_seg = 1 or 2 column segments of the same rows in one launch (q and k of the
 * self-attention: segment s reads x + s*seg_x, dy + s*seg_dy, writes dx + s*seg_dx — element offsets — with its own
 * weight[s] / dweight[s]; with n_seg = 2 the dweight pointers are both set or both NULL).  dx may alias dy.
 * workspace: omh_rmsnorm_rope_bwd2_workspace(rows, dim, n_seg) floats (only read when a dweight is set). */
typedef struct omh_rms_bwd_args {
    const void* x; int32_t x_bf16; int64_t ldx; const void* dy; int32_t dy_bf16; int64_t lddy; void* dx; int64_t lddx;
    int32_t n_seg; int64_t seg_x, seg_dy, seg_dx;
    const float* weight[2]; float* dweight[2];
    int64_t rows; int32_t dim; float eps; int32_t do_norm;
    const float* rope_cos; const float* rope_sin; int32_t rope_len, head_dim; const int32_t* grid; int32_t seq_len;
    float* workspace; int64_t workspace_floats;
    omh_partial_reduce* deferred;          /* ABI v9; NULL: the partials are added here */
} omh_rms_bwd_args;
int64_t omh_rmsnorm_rope_bwd2_workspace(int64_t rows, int32_t dim, int32_t n_seg);
int omh_rmsnorm_rope_bwd2(const omh_rms_bwd_args* args, omh_stream_t stream);
/* dS = P * (dP - sum_j P*dP) * scale per row (softmax backward of the unfused attention backward). */
int omh_softmax_bwd_rows(const void* p_bf16, int64_t ldp, const float* dp, int64_t lddp, void* ds_bf16, int64_t ldds,
                         int64_t R, int32_t L, float scale, omh_stream_t stream);
/* Adjoint of omh_unpatchify: g fp32 [Cout, f*pt, h*ph, w*pw] -> dtok bf16 [f*h*w, pt*ph*pw*Cout]. */
int omh_unpatchify_bwd(const float* g, void* dtok_bf16, int32_t Cout, int32_t f, int32_t h, int32_t w,
                       int32_t pt, int32_t ph, int32_t pw, omh_stream_t stream);
/* Backward of omh_dense_f32 (act_out must have been 0): dW += dy^T act(x), db += sum_b dy,
 * dx (=|+=) (dy W) * act'(x).  dW_accum / dx may be NULL. */
int omh_dense_f32_bwd(const float* x, const float* W, const float* dy, float* dW_accum, float* db_accum,
                      float* dx, int32_t dx_accumulate, int32_t B, int32_t N, int32_t K, int32_t act_in,
                      omh_stream_t stream);

/* Fused AdamW (decoupled weight decay, bias-corrected; torch.optim.AdamW semantics,
 * distilled_trainer.py:69-75) on fp32 parameters; grads are divided by grad_scale first
 * (GradScaler-style loss scaling, distilled_trainer.py:95,301).  step >= 1. */
int omh_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int32_t step, float grad_scale, omh_stream_t stream);
/* The same update for n_tensors tensors in one launch: table is a DEVICE array of n_tensors x 5 int64
 * {param ptr, grad ptr, exp_avg ptr, exp_avg_sq ptr, numel}; all tensors share `step`. */
int omh_adamw_multi(const int64_t* table, int32_t n_tensors, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int32_t step, float grad_scale, omh_stream_t stream);
/* bf16 operand copies of the fp32 master weights, all in ONE launch (they go stale with every optimizer step,
 * distilled_trainer.py:376-380: the reference's autocast re-casts every weight on every use).  table: DEVICE array of
 * n_entries x 9 int64 {src fp32 [rows, cols] contiguous, dst, dstT, rows, cols, ld_dst, ld_dstT, first_tile, kind};
 * kind 0: dst bf16 [rows, ld_dst] = bf16(src) and dstT bf16 [cols, ld_dstT] = its transpose (either may be 0), one
 * workgroup per 64 x 64 tile; kind 1: dst fp32 = src, one workgroup per 4096 elements.  first_tile = prefix sum of the
 * entries' workgroup counts (ascending), total_tiles = their sum. */
int omh_pack_weights_multi(const int64_t* table, int32_t n_entries, int64_t total_tiles, omh_stream_t stream);
/* ABI v8: omh_adamw_multi and omh_pack_weights_multi in ONE pass — the optimizer step writes the bf16 operand copies
 * itself instead of a second launch re-reading every updated parameter.  table: DEVICE array of n_entries x 12 int64
 * {param, grad, exp_avg, exp_avg_sq (fp32), dst, dstT, rows, cols, ld_dst, ld_dstT, first_tile, kind}; kind 0: a
 * [rows, cols] weight with bf16 copies (dst / dstT, either may be 0), one workgroup per 64 x 64 tile; kind 1: fp32 copy
 * dst = updated param; kind 2: no copy; kinds 1, 2: one workgroup per 4096 elements.  Same arithmetic as the two
 * separate entry points; the copies are the bf16 images of the parameters just written. */
int omh_adamw_pack_multi(const int64_t* table, int32_t n_entries, int64_t total_tiles, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int32_t step, float grad_scale, omh_stream_t stream);
/* EMA of the weights, ema = decay*ema + (1-decay)*p (distilled_trainer.py:319-334). */
int omh_ema_update(float* ema, const float* p, int64_t n, float decay, omh_stream_t stream);
/* ABI v10: the same update for every parameter of a model in ONE launch (the loop of distilled_trainer.py:319-334 is
 * ~825 tensors for Wan2.1-1.3B).  table: DEVICE array of n_entries x 4 int64 {ema, p (fp32), numel, first_chunk}, one
 * workgroup per 4096-element chunk, first_chunk = running sum of ceil(numel / 4096); total_chunks = its final value.
 * Same arithmetic per element as omh_ema_update. */
int omh_ema_update_multi(const int64_t* table, int32_t n_entries, int64_t total_chunks, float decay, omh_stream_t stream);

/* ========================================================================
 * Prompt-side encoders (run once per prompt): the umT5 text encoder of seaweed_apt/wan/modules/t5.py:272-322 and
 * the CLIP vision tower of seaweed_apt/wan/modules/clip.py:209-301.  Their Linear layers run on omh_gemm_bf16;
 * attention (head_dim 64 / 80, <= 512 keys) is two batched omh_gemm_bf16 calls around omh_softmax_bias_rows.
 * ====================================================================== */

/* nn.Embedding lookup (t5.py:306): out[r][:] = table[ids[r]][:], fp32; ids int64 on the device, clamped to the table. */
int omh_gather_rows_f32(const float* table, const int64_t* ids, float* out, int64_t rows, int32_t dim, int64_t vocab,
                        omh_stream_t stream);
/* The same from a bf16 table (the umT5 checkpoint is bf16: the embedding is never widened to fp32 as a whole). */
int omh_gather_rows_bf16(const void* table_bf16, const int64_t* ids, float* out, int64_t rows, int32_t dim, int64_t vocab,
                         omh_stream_t stream);
/* T5LayerNorm (t5.py:55-69): y = weight * x * rsqrt(mean(x^2) + eps); either result pointer may be NULL. */
int omh_rmsnorm_f32(const float* x, const float* weight, float eps, float* y_f32, void* y_bf16, int64_t rows,
                    int32_t dim, omh_stream_t stream);
/* nn.LayerNorm with affine, fp32 in and out (clip.py:47-50; the pre_norm of the embedded tokens, clip.py:288-289). */
int omh_layernorm_f32(const float* x, const float* weight, const float* bias, float eps, float* y, int64_t rows,
                      int32_t dim, omh_stream_t stream);
/* Softmax rows with T5's additive attention terms (t5.py:101-113; also clip.py:82 with bucket = table = NULL):
 *   y[h*L + i][j] = bf16( softmax_j( x[h*L + i][j] * scale + table[bucket[i*L + j]][h] ) ),  j < klen;
 * keys j >= klen get weight 0 (the reference fills finfo.min), columns L..ldy-1 are zeroed.  x fp32 [H*L, ldx],
 * bucket int32 [L, L] (relative-position bucket of (i, j), t5.py:244-268, computed on the host), table fp32 [nb, H]. */
int omh_softmax_bias_rows(const float* x, int64_t ldx, void* y_bf16, int64_t ldy, int32_t H, int32_t L, float scale,
                          const int32_t* bucket, const float* table, int32_t klen, omh_stream_t stream);
/* out = a * b, bf16, n even: the gated product of T5FeedForward (t5.py:137). */
int omh_mul_bf16(const void* a_bf16, const void* b_bf16, void* out_bf16, int64_t n, omh_stream_t stream);
/* ViT token assembly (clip.py:280-287): out[b][0] = cls + pos[0], out[b][1+i] = tok[b][i] + pos[1+i]; fp32. */
int omh_vit_embed(const float* tok, const float* cls, const float* pos, float* out, int32_t B, int32_t n, int32_t dim,
                  omh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OMH_H */
