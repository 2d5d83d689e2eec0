/*
 * clipfs.h -- C ABI of libclipfs_hip.so: the MI355X (gfx950) engine behind the
 * reference's Python API for the CLIP few-shot hot path.
 *
 * The reference (Dokumushikun/jittor-clip-fewshot) has NO FFI / plugin layer:
 * every op below is executed today by Jittor-generated CUDA kernels reached from
 * Python.  Each entry point cites the reference Python site
 * whose arithmetic it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named h_*; tensors are fp32,
 *     row-major, densely packed unless a leading dimension is given;
 *   - activations are token-major: row m = b * L + l  (the reference is
 *     sequence-first [L, N, d]; only the memory order differs, not the values);
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream);
 *   - return value: 0 = ok, CLIPFS_EINVAL = bad argument (nothing launched),
 *     otherwise 1000 + hipError_t of the failed launch;
 *   - no entry point allocates or frees device memory or synchronises the stream: all buffers (workspaces
 *     included) are owned by the caller (PyTorch-ROCm tensors).  The only state kept is per host thread:
 *     the last error string and, while enabled, the GEMM timing events (clipfs_gemm_timing).
 */
#ifndef CLIPFS_H
#define CLIPFS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLIPFS_OK 0
#define CLIPFS_EINVAL 1
#define CLIPFS_HIPERR_BASE 1000

/* Version 2 (round 3): the descriptor structs (clipfs_gemm_args, clipfs_tower) start with `struct_size` = sizeof of the
 * struct the CALLER was compiled against; the library rejects (CLIPFS_EINVAL, nothing launched) a size it does not know
 * instead of reading fields at shifted offsets.  New fields are appended at the END of a struct only; a change that moves
 * an existing field or a parameter of an entry point bumps this number.  Version 1 (rounds 1-2) had no size member and
 * changed layout once without a bump: binaries built against it must be rebuilt. */
#define CLIPFS_ABI_VERSION 2
int clipfs_abi_version(void);
/* sha256 prefixes (16 hex digits) of the sources the library was built from: all of csrc/ + this header, and the
 * fp32 GEMM alone (gemm.hip, gemm_common.h, common.h).  "unstamped" when built without build.py. */
const char* clipfs_source_stamp(void);
const char* clipfs_gemm_source_stamp(void);
/* human readable description of the last CLIPFS_EINVAL on this thread */
const char* clipfs_last_error(void);

/* ------------------------------------------------------------------ GEMM --
 * C[M,N] = epilogue( alpha * A[M,K] * B[N,K]^T )      fp32 MFMA (v_mfma_f32_32x32x2_f32)
 * Replaces every jittor.nn.linear / matmul on the path: packed QKV (jclip/mha.py:140),
 * LinearLoRA (lora_train_vlp.py:286-306), out_proj (mha.py:461), MLP (jclip/model.py:34-39),
 * patch-embed conv (model.py:87-91,105), visual.proj / text_projection (model.py:124,213-214),
 * cosine logits (lora_train_vlp.py:995), and their input-gradients.
 * Epilogue, in this order (each part optional):
 *     v  = alpha * acc + bias[n]
 *     v += lora_scale * sum_j lora_t[m, seg*r + j] * lora_b[n, j]      seg = n / lora_seg_width
 *     if act == 1:  (aux_out[m,n] = v);  v = v * sigmoid(1.702 v)        QuickGELU, model.py:24-27
 *     if act == 2:  v = v * dQuickGELU(aux_in[m,n])                      backward of act 1
 *     v += residual[rm, n]                                               model.py:60-61
 *     if act == 3:  v = max(v, 0)                                        ReLU after the residual add (the ResNet-50
 *                                                                        bottleneck of the MoCo branch, slow_pace.py:1239)
 *     C[out_row(m), n] = v
 * a_mode 0: A is row-major [M,K] with leading dimension lda.
 * a_mode 1: A is an NCHW image batch [B,3,R,R]; row m = (b, py, px) patch, column
 *           k = (c, ky, kx)  (im2col on the fly, kernel = stride = patch).  Then
 *           out_row(m) = b*out_tokens + 1 + p and residual row rm = 1 + p  (p = m % P):
 *           class-token slot skipped, positional embedding added (model.py:105-114).
 */
typedef struct clipfs_gemm_args {
  size_t struct_size;      /* = sizeof(clipfs_gemm_args) of the caller's header (ABI check) */
  const float* A;
  const float* B;          /* [N,K] row-major, ldb */
  float* C;                /* [*,N] row-major, ldc */
  int M, N, K;
  int lda, ldb, ldc;
  float alpha;
  const float* bias;       /* [N] or NULL */
  const float* residual;   /* [*,N] (ld = ldres) or NULL */
  int ldres;
  int act;                 /* 0 none, 1 QuickGELU, 2 multiply by dQuickGELU(aux_in), 3 ReLU (after the residual; fp32 only) */
  float* aux_out;          /* act 1: pre-activation copy (ld = ldc) or NULL */
  const float* aux_in;     /* act 2: saved pre-activation (ld = ldc) */
  const float* lora_t;     /* [M, lora_nseg * lora_r] or NULL */
  const float* lora_b;     /* [N, lora_r] */
  int lora_r, lora_nseg, lora_seg_width;
  float lora_scale;
  int a_mode;              /* 0 dense, 1 patch im2col */
  int img_res, patch, out_tokens; /* a_mode 1 */
  const void* B_planes;    /* optional 16-bit copy of B (a_mode 0, K % 32 == 0, ldb == K); NULL = exact fp32 */
  int b_format;            /* 1: bf16 hi/lo planes (clipfs_split_bf16) -> split-bf16 x3 kernel;
                              2: one f16 plane (clipfs_convert_f16) -> f16 MFMA kernel (cfg-5's fp16 path) */
  float* workspace;        /* optional split-K / stream-K slab scratch, 16-byte aligned (NULL: never split) */
  size_t workspace_floats;
  const void* A_f16;       /* optional f16 copy of A [M,K] (ld = lda): with b_format 2 selects the f16 x f16 kernel
                              (operands stream HBM -> LDS -> MFMA untouched); A may then be NULL */
  void* C_f16;             /* f16 x f16 kernel only: also (or, with C == NULL, only) write the result as f16, ld = ldc */
  int aux_f16;             /* f16 x f16 kernel only: aux_out / aux_in hold f16 values (fp16 storage of the saved
                              pre-activation), ld = ldc */
  int* counters;           /* optional arrival counters (split-K, stream-K): >= clipfs_gemm_counter_ints(M,N,K) ints, ZERO on
                              entry (the kernel leaves them zero); with `workspace` enables those schedules */
  size_t counters_ints;
} clipfs_gemm_args;
/* Ordering: everything the call does is ordered on `stream` (work queued on it before the call happens-before, work
 * queued after it happens-after).  The f16 x f16 path may run part of the rows on an internal high-priority side
 * stream, forked from and joined back into `stream` with events inside the call (one side stream per caller stream
 * and host thread, created on first use). */
int clipfs_gemm_nt(const clipfs_gemm_args* args, void* stream);
/* Products with few output tiles (small per-rank batches, the one-row-per-sequence products of the last block) are
 * cut along K into `clipfs_gemm_splits` slices.  Every slice writes its raw partial tile to `workspace` and the slice
 * that arrives LAST at the tile's arrival counter (`counters`) adds the slices in slice order and applies the epilogue,
 * inside the same launch -- deterministic, no float atomics, no second kernel.  Enabled when workspace_floats >=
 * clipfs_gemm_workspace_floats(M,N,K) AND counters_ints >= clipfs_gemm_counter_ints(M,N,K) (both 0 for shapes that
 * are never split); `counters` must be zero on entry and is left zero. */
/* planes[0..n) = bf16(src), planes[n..2n) = bf16(src - hi): the frozen-weight half of the opt-in split-bf16
 * ("bf16 x 3") GEMM: a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate.
 * `planes` holds 2*n bf16 values (4*n bytes); n % 8 == 0. */
int clipfs_split_bf16(const float* src, void* planes, size_t n, void* stream);
/* dst[0..n) = f16(src): weights for the fp16 MFMA mode (v_mfma_f32_32x32x16_f16, fp32 accumulate; activations are
 * rounded to f16 in the staging path).  Tolerance is that of fp16 products (~5e-4 relative), stated in the tests. */
int clipfs_convert_f16(const float* src, void* dst, size_t n, void* stream);
int clipfs_gemm_splits(int M, int N, int K);
size_t clipfs_gemm_workspace_floats(int M, int N, int K);
/* Stream-K (opt-in: environment CLIPFS_GEMM_SK=1|2; dense fp32 products with K % 32 == 0, when `workspace` and `counters`
 * are both supplied; clipfs_gemm_counter_ints returns 0 while it is off): the tiles x K-steps
 * iteration space is cut into equal contiguous runs, one per resident workgroup, instead of one tile per workgroup
 * (whose last round leaves most CUs idle at the path's 2-9 tiles per slot).  A tile shared by several runs is summed by
 * the last run to arrive, in run order (bitwise reproducible, no float atomics, nobody waits).  `workspace` holds the
 * partial-tile slabs, `counters` one arrival counter per tile: zeroed once by the caller, left zero by every launch. */
size_t clipfs_gemm_counter_ints(int M, int N, int K);
/* Diagnostics for bench.py's roofline leg (never enabled inside a timed region): while enabled, every
 * GEMM launch of the calling thread is bracketed by HIP events on its launch stream;
 * clipfs_gemm_timing_collect synchronises on them and returns the summed kernel time, the summed
 * algorithmic FLOPs (2*M*N*K) and the number of launches since the last collect. */
int clipfs_gemm_timing(int enable);
int clipfs_gemm_timing_collect(double* total_ms, double* total_flops, int* launches);
/* algorithmic HBM bytes (every operand, result and epilogue tensor counted once) of the launches summed by the last
 * clipfs_gemm_timing_collect of the calling thread */
double clipfs_gemm_timing_last_bytes(void);

/* ------------------------------------------------------------- LayerNorm --
 * y = (x - mean) / sqrt(var + eps) * gamma + beta over the last dim (biased var).
 * Replaces jittor nn.LayerNorm, jclip/model.py:17-21,115,121,209.
 * `x` rows are read with stride ldx (lets ln_post read the class-token rows in place).
 * mean/rstd (each [rows]) may be NULL (inference). */
int clipfs_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* y,
                         float* mean, float* rstd, int rows, int width, float eps, void* stream);
/* dx = dres + LN'(dy)   (dres may be NULL); gamma frozen: no dgamma/dbeta. */
int clipfs_layernorm_bwd(const float* dy, const float* x, int ldx, const float* gamma, const float* mean,
                         const float* rstd, const float* dres, float* dx, int lddx, int rows, int width,
                         void* stream);
/* fp16 storage mode: the same kernels writing an f16 copy of the result [rows, width] for the GEMM that consumes it
 * (y16 / dx16 may be NULL; in the forward y may be NULL when only the f16 operand is needed). */
int clipfs_layernorm_fwd_f16(const float* x, int ldx, const float* gamma, const float* beta, float* y, void* y16,
                             float* mean, float* rstd, int rows, int width, float eps, void* stream);
int clipfs_layernorm_bwd_f16(const float* dy, const float* x, int ldx, const float* gamma, const float* mean,
                             const float* rstd, const float* dres, float* dx, void* dx16, int lddx, int rows, int width,
                             void* stream);
/* LayerNorm forward with the adapter's down-projection in the same pass: y = LN(x) (and / or its f16 copy y16) and
 * t[row, s*r + j] = sum_k dropout_s(y)[row, k] * A[s*r + j, k], exactly clipfs_layernorm_fwd(_f16) followed by
 * clipfs_lora_down on y (same Philox counters, so clipfs_lora_bwd regenerates the same masks; t agrees to fp32 summation
 * order).  Replaces ln_1 + lora_A(dropout(x)) of the adapted q/k/v projections, jclip/model.py:115 with
 * lora_train_vlp.py:296-306.  Covered: nseg == 3, r in {1, 2, 4} (clipfs_layernorm_fwd_lora_ok); other shapes take the
 * two separate calls.  keep_bits (may be NULL): as for clipfs_lora_down. */
int clipfs_layernorm_fwd_lora_ok(int width, int r, int nseg);
int clipfs_layernorm_fwd_lora(const float* x, int ldx, const float* gamma, const float* beta, float* y, void* y16,
                              float* mean, float* rstd, int rows, int width, float eps, const float* A, float* t, int r,
                              int nseg, unsigned seg_mask, float p, uint64_t seed, uint32_t stream_base, uint32_t drow0,
                              void* keep_bits, void* stream);

/* ------------------------------------------------------------- attention --
 * qkv [B*L, 3*d] (q | k | v, head h at columns h*64..), out [B*L, d] heads merged.
 * out = softmax(q k^T / sqrt(64) + causal mask) v per (b, head), head_dim = 64.
 * Replaces scaled_dot_product_attention + reshapes + permute(2,0,1,3):
 * jclip/mha.py:55-83,439-458 and lora_train_vlp.py:339-367,492-501.
 * Scores/probabilities never leave the CU (reference: [N*H,L,L] round trip in HBM). */
int clipfs_attention_fwd(const float* qkv, float* out, float* lse, int batch, int seq, int heads, int causal,
                         void* stream);
/* dqkv from dout, recomputing the probabilities from qkv.
 * Default path, seq <= 288: exact-fp32 MFMA kernels (v_mfma_f32_32x32x2_f32, scores kept transposed so that the
 * softmax statistics are per-lane scalars).  The forward writes lse [clipfs_attention_lse_floats] (log-sum-exp of
 * the scaled scores per (batch, head, query)) when given the buffer; the backward takes the forward's `out`, that
 * `lse` and a `work` buffer of the same size.
 * Without lse: seq <= 96 falls back to register/LDS-resident VALU kernels that recompute the softmax (out, lse and
 * work may be NULL); seq > 96 requires lse; seq > 288 runs streaming kernels with an online softmax. */
int clipfs_attention_bwd(const float* qkv, const float* dout, const float* out, const float* lse, float* dqkv,
                         float* work, int batch, int seq, int heads, int causal, void* stream);
size_t clipfs_attention_lse_floats(int batch, int seq, int heads);
/* fp16 storage mode (cfg-5), seq <= 288: the same function with both contractions on
 * v_mfma_f32_32x32x16_f16 (operands rounded to f16 in the staging path; softmax statistics, accumulators and
 * outputs fp32).  lse as above (may be NULL when no backward follows). */
int clipfs_attention_f16_fwd(const void* qkv, int qkv_f16, float* out, void* out16, float* lse, int batch, int seq,
                             int heads, int causal, void* stream);
/* dqkv from (qkv, dout, out, lse) of clipfs_attention_f16_fwd; work: batch*heads*seq floats (D_i = dO_i . O_i).
 * out16 / dqkv16 (may be NULL): f16 copies of out / dqkv, the A operands of the GEMMs that follow; with dqkv16 given,
 * dqkv itself may be NULL (fp16 storage mode: every consumer reads the f16 image, see clipfs_lora_bwd_f16dy).
 * qkv_f16 != 0: qkv is an f16 tensor [B*L, 3*d] (the QKV GEMM's f16 output: fp16 storage), else fp32.
 * dout_f16 != 0: dout is an f16 tensor [B*L, d] (the output-projection dgrad GEMM's f16-only result), else fp32. */
int clipfs_attention_f16_bwd(const void* qkv, int qkv_f16, const void* dout, int dout_f16, const float* out,
                             const float* lse, float* dqkv, void* dqkv16, float* work, int batch, int seq, int heads,
                             int causal, void* stream);

/* ------------------------------------------------------------------ LoRA --
 * t[m, s*r + j] = sum_k drop_s(x)[m,k] * A[s*r + j, k]       (the "down" half of
 * lora_train_vlp.py:302: x @ (B@A)^T == (x @ A^T) @ B^T; the reference materialises B@A).
 * drop_s = Philox dropout of nn.Dropout(p) (:298-299), one independent stream per
 * segment s: stream id = stream_base + s, element (drow0 + m, k) as documented in DESIGN.md
 * (drow0 = index of row 0 in the GLOBAL batch: a data-parallel shard draws the masks of the
 * one-process run); p = 0 or seed == 0 disables dropout.  seg_mask bit s = 0 leaves t[:, s*r..] = 0. */
int clipfs_lora_down(const float* x, const float* A, float* t, int rows, int width, int r, int nseg,
                     unsigned seg_mask, float p, uint64_t seed, uint32_t stream_base, uint32_t drow0,
                     void* keep_bits, void* stream);
/* keep_bits (may be NULL): uint16 [rows, width/4]; with dropout active the forward records its masks there -- bit
 * 4*s + e of entry (m, c) set <=> element (m, 4*c + e) was kept for segment s -- and clipfs_lora_bwd / _f16dy given the
 * same buffer read them instead of evaluating Philox again (identical masks by construction; the adapter's dA / dx
 * products were VALU-bound on the generator).  Matrix-core kernels only: clipfs_lora_keep_bits_ok(...) != 0. */
int clipfs_lora_keep_bits_ok(int width, int segw, int r, int nseg);
/* Backward of the adapter pair for one linear with nseg stacked segments:
 *   dt[m, s*r+j]  = scale * sum_n dy[m, s*segw + n] * B[s*segw + n, j]
 *   dB[s*segw+n,j] += scale * sum_m dy[m, s*segw+n] * t[m, s*r+j]
 *   dA[s*r+j, k]  += sum_m dt[m, s*r+j] * drop_s(x)[m,k]
 *   dx[m,k]       += sum_{s,j} dt[m, s*r+j] * A[s*r+j,k] * dropscale_s(m,k)   (if dx != NULL)
 * work: caller scratch, >= clipfs_lora_bwd_work_floats(...) floats. */
size_t clipfs_lora_bwd_work_floats(int rows, int width, int r, int nseg);
int clipfs_lora_bwd(const float* dy, const float* x, const float* t, const float* A, const float* B,
                    float* dt, float* dA, float* dB, float* dx, int rows, int width, int segw, int r,
                    int nseg, unsigned seg_mask, float scale, float p, uint64_t seed, uint32_t stream_base,
                    uint32_t drow0, const void* keep_bits, float* work, void* stream);
/* The same with dy given as its f16 image [rows, nseg*segw] (fp16 storage mode: the tensor the dgrad GEMM consumes), so
 * that the two passes over dy move half the bytes and the fp32 dy need not exist.  Matrix-core kernels only:
 * clipfs_lora_bwd_f16dy_ok(width, segw, r, nseg) != 0 says a shape is covered (r <= 16, width % 128 == 0, ...). */
int clipfs_lora_bwd_f16dy_ok(int width, int segw, int r, int nseg);
int clipfs_lora_bwd_f16dy(const void* dy16, const float* x, const float* t, const float* A, const float* B,
                          float* dt, float* dA, float* dB, float* dx, int rows, int width, int segw, int r,
                          int nseg, unsigned seg_mask, float scale, float p, uint64_t seed, uint32_t stream_base,
                          uint32_t drow0, const void* keep_bits, float* work, void* stream);

/* --------------------------------------------------------- token assembly --
 * vit: x[b,0,:] = class_embedding + pos[0]; x[b, 1+P+i, :] = vpt[i]  (jclip/model.py:109-114,
 * jclip/model1.py:192-194).  Patch rows are written by the patch GEMM epilogue. */
int clipfs_vit_fill_special(float* x, const float* class_emb, const float* pos, const float* vpt, int batch,
                            int tokens, int n_patch, int n_vpt, int width, void* stream);
/* text: x[c,l,:] = (ctx row if 1 <= l <= n_ctx and ctx != NULL else table[ids[c,l]]) + pos[l]
 * (jclip/model.py:203-205; slow_pace.py:185-199,838). */
int clipfs_text_embed(const int64_t* ids, const float* table, const float* pos, const float* ctx, int n_ctx,
                      float* x, int n, int seq, int width, void* stream);
/* dtok[i,:] += sum_c dx[c, first+i, :]: gradient of tokens shared by every sequence -- the text prompt
 * ctx (first = 1, slow_pace.py:198-199) or the VPT tokens (first = 1 + patches, model1.py:192-194). */
int clipfs_token_rows_grad(const float* dx, float* dtok, int n, int seq, int width, int n_tok, int first,
                           void* stream);
/* C[m,n] = alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]: strided VALU product for the two tiny
 * logits-backward products whose reduction runs over the classes (403: not 16-byte friendly). */
int clipfs_matmul_small(const float* A, const float* B, float* C, int M, int N, int K, long sam, long sak,
                        long sbk, long sbn, float alpha, void* stream);
/* out[c,:] = x[c*seq + argmax_l ids[c,l], :]   (EOT row, jclip/model.py:213-214) ; idx_out optional */
int clipfs_gather_eot(const float* x, const int64_t* ids, float* out, int32_t* idx_out, int n, int seq,
                      int width, void* stream);
/* scatter-add of the above: dx[c*seq + idx[c], :] = dy[c,:], all other rows zero */
int clipfs_scatter_rows(const float* dy, const int32_t* idx, float* dx, int n, int seq, int width, void* stream);
/* one row per sequence, the building blocks of the sparse last-block backward (clipfs_tower_bwd_sparse):
 *   gather:  out[c, :] = src[(c*seq + idx[c]) * ld + 0..width)          add:  dx[c*seq + idx[c], :] += src[c, :] */
int clipfs_gather_seq_rows(const float* src, size_t ld, const int32_t* idx, float* out, int n, int seq, int width,
                           void* stream);
int clipfs_add_seq_rows(const float* src, const int32_t* idx, float* dx, int n, int seq, int width, void* stream);
/*   put:     dst[(c*seq + idx[c]) * ld + 0..width) = src[c, :]  (other rows untouched; clipfs_tower_fwd_rows)
 *   eot_index: idx[c] = argmax_l ids[c, l], first maximum (the EOT position, jclip/model.py:213-214) */
int clipfs_put_seq_rows(const float* src, const int32_t* idx, float* dst, size_t ld, int n, int seq, int width,
                        void* stream);
int clipfs_eot_index(const int64_t* ids, int32_t* idx, int n, int seq, void* stream);
/* gather / put for a tensor kept as f16 (the saved pre-GELU activation of the fp16 storage mode); ld in halves, the
 * compact side [n, width] is fp32 */
int clipfs_gather_seq_rows_f16(const void* src_f16, size_t ld, const int32_t* idx, float* out, int n, int seq, int width,
                               void* stream);
int clipfs_put_seq_rows_f16(const float* src, const int32_t* idx, void* dst_f16, size_t ld, int n, int seq, int width,
                            void* stream);

/* --------------------------------------------------------- BPE tokenizer --
 * Native merge loop of the CLIP byte-pair encoder (jclip/simple_tokenizer.py:88-129; host code, no GPU work).
 * create: `merges` = the merges text after its header line, one "left right" per line; returns an opaque handle or
 * NULL.  encode: word w = UTF-8 bytes [offsets[w], offsets[w+1]) of `words` (already cleaned, lower-cased and split by
 * the caller); writes each word's vocabulary ids consecutively into ids (capacity cap) and their number into
 * counts[w]; returns the total or -1. */
void* clipfs_bpe_create(const char* merges, size_t n_bytes, int n_merges);
void clipfs_bpe_destroy(void* handle);
long clipfs_bpe_encode(const void* handle, const uint8_t* words, const int32_t* offsets, int n_words, int32_t* ids,
                       int32_t* counts, long cap);

/* ------------------------------------------------- MoCo ResNet-50 branch --
 * Data movement of the frozen ResNet-50 feature extractor (slow_pace.py:1237-1271,1677-1680; forward only, NHWC
 * activations; convolutions run as clipfs_gemm_nt with BatchNorm folded into weights / bias and ReLU = act 3):
 *   nchw_to_nhwc : y[n,h,w,c] = x[n,c,h,w]
 *   im2col_nhwc  : col[(n,ho,wo), (ky,kx,c)] = x[n, ho*stride-pad+ky, wo*stride-pad+kx, c] (0 outside), row length Kp
 *                  (>= kh*kw*C, multiple of 4, tail zero) -- the A operand of the convolution's GEMM
 *   maxpool3x3s2 : torchvision's MaxPool2d(3, 2, 1) on NHWC;   global_avgpool : mean over HW -> [N, C] */
int clipfs_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, void* stream);
int clipfs_im2col_nhwc(const float* x, float* col, int N, int H, int W, int C, int kh, int kw, int stride, int pad,
                       int Kp, void* stream);
int clipfs_maxpool3x3s2_nhwc(const float* x, float* y, int N, int H, int W, int C, void* stream);
int clipfs_global_avgpool_nhwc(const float* x, float* y, int N, int HW, int C, void* stream);

/* ---------------------------------------------------------- head / loss --
 * y = x / ||x||_2 per row; inv_norm[row] saved (may be NULL).  jclip/model.py:222-224. */
int clipfs_l2norm_fwd(const float* x, float* y, float* inv_norm, int rows, int width, void* stream);
/* dx = inv_norm * (dy - y * <dy,y>) */
int clipfs_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int rows, int width,
                      void* stream);
/* per class: normalise each template embedding, mean, normalise (lora_train_vlp.py:978-990 /
 * clip_classifier :647-666).  emb [C*t, d] (class major), out [C, d]. */
int clipfs_class_mean_fwd(const float* emb, float* out, int classes, int templates, int width, void* stream);
int clipfs_class_mean_bwd(const float* emb, const float* dout, float* demb, int classes, int templates, int width,
                          void* stream);
/* mean softmax cross entropy + dlogits (= (softmax - onehot) / rows * grad_scale);
 * loss_rows [2*rows] scratch: per-sample losses, then per-row hit flags; loss_sum [1] = sum of the
 * per-sample losses (fixed summation order).  lora_train_vlp.py:997.
 * correct [1] (may be NULL) counts argmax == target (cls_acc :638-644). */
int clipfs_cross_entropy(const float* logits, const int64_t* target, float* dlogits, float* loss_rows,
                         float* loss_sum, int32_t* correct, int rows, int classes, float grad_scale,
                         void* stream);
/* top-k labels per row, ties broken towards the smaller class index (cls_acc :639, test.py:1738) */
int clipfs_topk(const float* logits, int32_t* labels, int rows, int classes, int k, void* stream);
/* Channel_LP (slow_pace.py:1195-1206): y = (scale1*x + bias1) W^T + b via clipfs_gemm_nt after this
 * affine; logit_normalize (:1276-1280): (z - rowmean)/std_all, std unbiased clamped at 1e-6.
 * work: >= 2 floats. */
int clipfs_channel_affine(const float* x, const float* scale1, const float* bias1, float* y, int rows,
                          int width, void* stream);
int clipfs_logit_normalize(const float* z, float* out, float* work, int rows, int classes, void* stream);
/* backward of logit_normalize (head training, slow_pace.py:1671-1675): dz from dzn, statistics recomputed from z */
int clipfs_logit_normalize_bwd(const float* z, const float* dzn, float* dz, int rows, int classes, void* stream);
/* out[c] = sum_r x[r,c] * (y ? y[r,c] : 1): bias / per-channel scale gradients of the head (fixed row order) */
int clipfs_colsum(const float* x, const float* y, float* out, int rows, int cols, void* stream);
/* Stage-2 self-consistency losses (slow_pace.py:1653-1658).
 * l1_loss: loss[0] = mean |a - b| (jittor nn.l1_loss); da (may be NULL) = sign(a - b) * grad_scale / n.
 * kl_logits: kl_div(log_softmax(logits), log_softmax(target_logits)) of slow_pace.py:1170-1177 per row:
 *   loss_rows[r] = sum_j q_j (log q_j - log p_j); dlogits (may be NULL) = (p - q) * grad_scale; the caller sums the rows
 *   and divides by numel (:1658). */
int clipfs_l1_loss(const float* a, const float* b, size_t n, float* loss, float* da, float grad_scale, void* stream);
int clipfs_kl_logits(const float* logits, const float* target_logits, float* loss_rows, float* dlogits, int rows,
                     int classes, float grad_scale, void* stream);

/* ------------------------------------------------------------- optimiser --
 * jittor.optim.AdamW.step (lora_train_vlp.py:946,1002): p *= 1 - lr*wd; m,v update;
 * p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).  One launch over the flat LoRA+prompt buffer. */
int clipfs_adamw(float* p, const float* g, float* m, float* v, size_t n, int step, float lr, float beta1,
                 float beta2, float eps, float weight_decay, float grad_scale, void* stream);

/* -------------------------------------------------------------------- MTA --
 * solve_mta (lora_train_vlp.py:742-811; slow_pace.py:1363-1433), one workgroup per image.
 * feats [n_img, V, d] unit rows (row 0 = centre view), text [C, d] (unit rows; the reference
 * passes its transpose [d, C]).  mode_out [n_img, d], logits_out [n_img, C] = 100 * mode . text
 * (either may be NULL).  work: >= clipfs_mta_work_floats floats.  views >= 5 (k = int(0.3 (V-1)) >= 1).
 * The diagonal of cdist is clamped at 0 before the sqrt (deviation from :740). */
size_t clipfs_mta_work_floats(int n_img, int views, int width, int classes);
int clipfs_mta(const float* feats, const float* text, float* mode_out, float* logits_out, float* work,
               int n_img, int views, int width, int classes, void* stream);

/* ----------------------------------------------------------- TTA views --
 * One kernel writes every normalised fp32 view [n_views, 3, S, S] of one uint8 HWC source image resident in HBM:
 * crop -> PIL-exact 8-bit resize (bilinear or bicubic, Pillow Resample.c restated) -> S x S window -> optional
 * horizontal flip -> (u8 - 255 mean) / (255 std).  Replaces the CPU/PIL view generation of ood.py:946-958,1084-1089
 * and jclip/clip.py:130-144 (Resize 256 bicubic + CenterCrop; RandomResizedCrop bilinear + RandomHorizontalFlip).
 * recs: int32 [n_views, 10] = {top, left, h, w, flip, out_w, out_h, win_x, win_y, filter(0 bilinear, 1 bicubic)};
 * boxes are sampled on the host (clipfs/views.py).  Needs max(h/out_h, w/out_w) * support * 2 + 1 <= 24 taps. */
int clipfs_tta_views(const uint8_t* image, int height, int width, const int32_t* recs, int n_views, int out_size,
                     const float* mean, const float* stdv, float* out, void* stream);

/* --------------------------------------------------------- tower drivers --
 * C++ sequencing of the kernels above for one transformer tower, so that one call
 * from Python enqueues a whole forward or backward (no per-kernel interpreter cost).
 * Replaces Transformer / ResidualAttentionBlock / VisionTransformer.execute and
 * CLIP.encode_text (jclip/model.py:42-126,202-215), PlainMultiheadAttentionLoRA
 * (lora_train_vlp.py:431-506) and their Jittor autograd. */
typedef struct clipfs_block {
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  const float *w_qkv, *b_qkv, *w_qkv_t; /* [3d,d], [3d], transposed copy [d,3d] for dgrad */
  const float *w_o, *b_o, *w_o_t;       /* [d,d] */
  const float *w_fc, *b_fc, *w_fc_t;    /* [4d,d], [4d], [d,4d] */
  const float *w_pr, *b_pr, *w_pr_t;    /* [d,4d], [d], [4d,d] */
  /* LoRA on q,k,v (stacked: A [3r,d], B [3d,r]) and on the out projection (A [r,d], B [d,r]) */
  const float *lora_a_qkv, *lora_b_qkv, *lora_a_o, *lora_b_o;
  float *g_lora_a_qkv, *g_lora_b_qkv, *g_lora_a_o, *g_lora_b_o; /* gradient slots (accumulated into) */
  unsigned lora_mask;                   /* bit0 q, bit1 k, bit2 v, bit3 o */
  /* optional 16-bit copies of the four weights and of their transposed copies (clipfs_split_bf16 /
   * clipfs_convert_f16, format in clipfs_tower.weight_format); when present the tower's GEMMs use the
   * bf16 x 3 or f16 MFMA kernel, otherwise the exact fp32 MFMA kernel */
  const void *w_qkv_p, *w_o_p, *w_fc_p, *w_pr_p, *w_qkv_t_p, *w_o_t_p, *w_fc_t_p, *w_pr_t_p;
} clipfs_block;

typedef struct clipfs_tower {
  size_t struct_size;       /* = sizeof(clipfs_tower) of the caller's header (ABI check) */
  size_t block_size;        /* = sizeof(clipfs_block): the stride of `blocks` */
  int width, heads, layers, seq, causal;
  int lora_r;
  float lora_scale, lora_dropout;
  uint64_t dropout_seed;    /* 0 = no dropout (eval) */
  uint32_t dropout_stream0; /* stream id of layer 0 segment 0; layer l uses stream0 + 4*l + s */
  uint32_t dropout_row0;    /* global index of this call's first token row (data-parallel shards); 0 otherwise */
  const clipfs_block* blocks; /* HOST array [layers] of device pointers */
  int weight_format;        /* format of the blocks' *_p copies: 0 none (exact fp32), 1 bf16 hi/lo, 2 f16 */
  int* gemm_counters;       /* optional: >= clipfs_tower_counter_ints(t, batch) ints, zeroed ONCE by the caller (every
                               GEMM leaves them zero) -- enables the stream-K GEMM schedule; one buffer per stream */
  size_t gemm_counters_ints;
} clipfs_tower;

/* floats needed per tower call for saved activations / scratch */
size_t clipfs_tower_saved_floats(const clipfs_tower* t, int batch);
size_t clipfs_tower_scratch_floats(const clipfs_tower* t, int batch);
size_t clipfs_tower_counter_ints(const clipfs_tower* t, int batch);
/* x [batch*seq, width] in/out (residual stream, updated in place).  saved == NULL: inference
 * (nothing kept); else activations for clipfs_tower_bwd are written to `saved`. */
int clipfs_tower_fwd(const clipfs_tower* t, float* x, int batch, float* saved, float* scratch, void* stream);
/* The same forward when the caller reads ONE row per sequence of the result (rows[c] = its token index: the class token,
 * jclip/model.py:121-124, or the EOT token, :213-214).  In the LAST block everything after the attention is row-wise,
 * so the output projection, LayerNorm 2 and the MLP run on `batch` rows instead of batch*seq: 9 d^2 MACs per skipped
 * token, 5.9 % of the image tower's forward and 6.0 % of the text tower's at cfg-2.  On return x holds the block output
 * at rows c*seq + rows[c] ONLY (the other rows keep the last block's input); `saved` is complete for
 * clipfs_tower_bwd_sparse with the same rows and NOT for clipfs_tower_bwd.  Falls back to clipfs_tower_fwd in the
 * cases clipfs_tower_bwd_sparse falls back (clipfs_tower_rows_mode() == 0: an o-projection adapter in the last block,
 * seq < 8, CLIPFS_DENSE_BWD=1), so the two always agree.  In the fp16 storage mode the `batch`-row products use the
 * fp32 master weights. */
int clipfs_tower_fwd_rows(const clipfs_tower* t, float* x, const int32_t* rows, int batch, float* saved, float* scratch,
                          void* stream);
int clipfs_tower_rows_mode(const clipfs_tower* t);
/* dx [batch*seq, width] in/out: gradient wrt the tower output on entry, wrt its input on exit.
 * stop_at_input != 0: the gradient wrt the tower input is not needed (image tower without VPT:
 * block 0's LN1 backward and q/k/v dgrad are skipped). */
int clipfs_tower_bwd(const clipfs_tower* t, float* dx, int batch, const float* saved, float* scratch,
                     int stop_at_input, void* stream);
/* The same backward when the gradient wrt the tower output is non-zero in ONE row per sequence only -- which is what
 * both towers receive: the image head reads the class token (jclip/model.py:121-124), the text head the EOT token
 * (:213-214).  dxs [batch, width] holds those rows, rows[b] their token index; dx [batch*seq, width] is written (no
 * zero-filled input needed) with the gradient wrt the tower input.  In the LAST block the MLP and output-projection
 * input-gradients are row-wise, so they run on `batch` rows instead of batch*seq (exact: the skipped rows are exact
 * zeros): 9 d^2 MACs per skipped token, 1.4 % (image) + 1.6 % (text) of the cfg-2 step.  Falls back to the dense path
 * (scatter + clipfs_tower_bwd) for o-projection adapters in the last block and seq < 8 (clipfs_tower_rows_mode). */
int clipfs_tower_bwd_sparse(const clipfs_tower* t, const float* dxs, const int32_t* rows, float* dx, int batch,
                            const float* saved, float* scratch, int stop_at_input, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLIPFS_H */
