// Host-side sequencing of one transformer tower (forward and backward): the C++ runtime that replaces
// Jittor's graph executor for Transformer / ResidualAttentionBlock (jclip/model.py:42-77),
// PlainMultiheadAttentionLoRA (lora_train_vlp.py:431-506) and their autograd.  One call enqueues
// every kernel of the pass on the caller's stream; no allocation, no synchronisation.
//
// Residual stream without copies: block l reads x_in[l], the out-projection epilogue writes
// x_mid[l] = x_in[l] + attn, the c_proj epilogue writes x_in[l+1] = x_mid[l] + mlp, so the tensors the
// backward needs are exactly the ones the forward had to produce anyway.
#include "common.h"

#include <stdlib.h>

namespace clipfs {

static inline size_t al4(size_t n) { return (n + 3) & ~(size_t)3; }

// fp16 storage mode runs attention on the f16 MFMA kernels (attention_f16.hip; sequences up to 288 tokens)
static inline bool f16_attention(const clipfs_tower* t) { return t->weight_format == 2 && t->seq <= 288; }
// ... and then qkv itself is stored as f16 (written by the QKV GEMM, read by the attention kernels): needs the
// f16 x f16 GEMM for the LoRA'd projection, i.e. a segment width that is a multiple of its 128-column tiles
static inline bool qkv_f16(const clipfs_tower* t) {
  return f16_attention(t) && (t->width % 128) == 0 && t->lora_r <= 16;
}

struct SavedLayout {
  size_t x_in, stat1, h1, t_qkv, qkv, att, lse, t_o, x_mid, stat2, u, keep, total;
};

// the q/k/v adapters' dropout masks travel from the forward to the backward as keep bits (2 bytes per float4 of the
// LayerNorm output, clipfs_lora_keep_bits_ok) instead of being regenerated from Philox in the dA and dx products
static inline bool keep_bits_slot(const clipfs_tower* t) {  // the slot exists (layout: independent of the step's seed)
  return t->lora_r > 0 && t->lora_dropout > 0.f && clipfs_lora_keep_bits_ok(t->width, t->width, t->lora_r, 3);
}
static inline bool keep_bits_saved(const clipfs_tower* t) { return keep_bits_slot(t) && t->dropout_seed != 0; }

static SavedLayout saved_layout(const clipfs_tower* t, size_t M) {
  const size_t d = t->width, r = t->lora_r > 0 ? t->lora_r : 0;
  SavedLayout L;
  size_t o = 0;
  L.x_in = o;  o += al4(M * d);
  L.stat1 = o; o += al4(2 * M);
  L.h1 = o;    o += al4(M * d);
  L.t_qkv = o; o += al4(M * 3 * r);
  L.qkv = o;   o += al4(qkv_f16(t) ? (M * 3 * d + 1) / 2 : M * 3 * d);  // fp16 mode: q | k | v saved as f16
  L.att = o;   o += al4(M * d);
  // log-sum-exp rows: the long-sequence fp32 kernels (0 floats for seq <= 96) and every f16 MFMA attention need them
  L.lse = o;   o += al4(t->weight_format == 2 ? (M / t->seq) * t->seq * (size_t)t->heads
                                              : clipfs_attention_lse_floats((int)(M / t->seq), t->seq, t->heads));
  L.t_o = o;   o += al4(M * r);
  L.x_mid = o; o += al4(M * d);
  L.stat2 = o; o += al4(2 * M);
  L.u = o;     o += al4(t->weight_format == 2 ? M * 2 * d : M * 4 * d);  // fp16 mode: pre-activation saved as f16
  L.keep = o;  o += keep_bits_slot(t) ? al4((M * (d / 4) + 1) / 2) : 0;    // uint16 per float4 of h1
  L.total = o;
  return L;
}

struct ScratchLayout {
  size_t h, big, b3, b1, dt, work, gemm_ws, gemm_ws_floats, a16, c16, total;
  size_t counter_ints;  // stream-K arrival counters the tower's largest GEMM needs (a separate, caller-zeroed buffer)
};

static ScratchLayout scratch_layout(const clipfs_tower* t, size_t M) {
  const size_t d = t->width, r = t->lora_r > 0 ? t->lora_r : 0;
  ScratchLayout S;
  size_t o = 0;
  S.h = o;    o += al4(M * d);
  S.big = o;  o += al4(M * 4 * d);
  S.b3 = o;   o += al4(M * 3 * d);
  S.b1 = o;   o += al4(M * d);
  S.dt = o;   o += al4(M * 4 * r);
  S.work = o; o += r ? al4(clipfs_lora_bwd_work_floats((int)M, (int)d, (int)r, 3)) : 0;
  // split-K scratch for the largest of the tower's GEMM shapes (0 unless the row count is small)
  size_t ws = 0, cnt = 0;
  const int shapes[7][2] = {{3 * (int)d, (int)d}, {(int)d, (int)d}, {4 * (int)d, (int)d}, {(int)d, 4 * (int)d},
                            {(int)d, 3 * (int)d}, {4 * (int)d, (int)d}, {(int)d, (int)d}};
  // ... at the tower's row count and at one row per sequence (the compact last block: clipfs_tower_fwd_rows /
  // clipfs_tower_bwd_sparse run their products on `batch` rows)
  const size_t row_counts[2] = {M, M / (size_t)t->seq};
  for (int k = 0; k < 2; ++k)
    for (int i = 0; i < 7; ++i) {
      if (!row_counts[k]) continue;
      const size_t w = clipfs_gemm_workspace_floats((int)row_counts[k], shapes[i][0], shapes[i][1]);
      const size_t c = clipfs_gemm_counter_ints((int)row_counts[k], shapes[i][0], shapes[i][1]);
      ws = w > ws ? w : ws;
      cnt = c > cnt ? c : cnt;
    }
  S.counter_ints = cnt;
  S.gemm_ws = o; S.gemm_ws_floats = ws; o += al4(ws);
  // fp16 storage mode (weight_format 2): f16 images of the GEMM operands, M x 4d halves each
  S.a16 = o; o += t->weight_format == 2 ? al4(M * 2 * d) : 0;
  S.c16 = o; o += t->weight_format == 2 ? al4(M * 2 * d) : 0;
  S.total = o;
  return S;
}

static int check_tower(const clipfs_tower* t, int batch) {
  CLIPFS_REQUIRE(t && t->blocks, "tower: null descriptor");
  CLIPFS_REQUIRE(t->struct_size == sizeof(clipfs_tower) && t->block_size == sizeof(clipfs_block),
                 "tower: descriptor built against another clipfs.h (struct_size %zu / block_size %zu, library has %zu / %zu)",
                 t->struct_size, t->block_size, sizeof(clipfs_tower), sizeof(clipfs_block));
  CLIPFS_REQUIRE(batch > 0 && t->layers > 0 && t->seq > 0 && t->heads > 0 && t->width == t->heads * 64,
                 "tower: width %d must be heads %d * 64", t->width, t->heads);
  CLIPFS_REQUIRE(t->lora_r >= 0 && t->lora_r <= 16, "tower: lora rank %d unsupported", t->lora_r);
  if (t->weight_format == 2)  // fp16 storage mode chains f16 results between GEMMs: every block needs all its f16 weights
    for (int l = 0; l < t->layers; ++l) {
      const clipfs_block& b = t->blocks[l];
      CLIPFS_REQUIRE(b.w_qkv_p && b.w_o_p && b.w_fc_p && b.w_pr_p, "tower: block %d lacks f16 weight copies", l);
    }
  return CLIPFS_OK;
}

// Per-call state of a tower pass (no hidden / thread-local state: everything a GEMM of the pass needs travels here)
struct TowerCtx {
  float* ws = nullptr;          // split-K / stream-K scratch (inside the call's scratch buffer)
  size_t ws_floats = 0;
  int* counters = nullptr;      // stream-K arrival counters (caller-zeroed; every launch leaves them zero)
  size_t counters_ints = 0;
  int b_format = 0;             // format of the blocks' 16-bit weight copies
  void* a16 = nullptr;          // fp16 mode: f16 image of the A operand (converted per GEMM)
  void* c16 = nullptr;          // fp16 mode: f16 output handed from one GEMM to the next
};

static TowerCtx make_ctx(const clipfs_tower* t, float* scratch, const ScratchLayout& SC) {
  TowerCtx c;
  c.ws = scratch + SC.gemm_ws;
  c.ws_floats = SC.gemm_ws_floats;
  c.counters = t->gemm_counters;
  c.counters_ints = t->gemm_counters_ints;
  c.b_format = t->weight_format;
  c.a16 = t->weight_format == 2 ? scratch + SC.a16 : nullptr;
  c.c16 = t->weight_format == 2 ? scratch + SC.c16 : nullptr;
  return c;
}

enum GemmChain { CHAIN_NONE = 0, CHAIN_OUT16 = 1, CHAIN_IN16 = 2 };

// chain: CHAIN_OUT16 = the result is only the next GEMM's A operand: in fp16 mode write it as f16 alone;
//        CHAIN_IN16  = A is the previous GEMM's CHAIN_OUT16 result.
// a16_ready: f16 image of A written by the producing kernel (LayerNorm / attention); NULL = convert here.
static int gemm(const TowerCtx& cx, const float* A, const float* B, const void* Bp, float* C, int M, int N, int K,
                const float* bias, const float* res, int act, float* aux_out, const float* aux_in, const float* lt,
                const float* lb, int r, int nseg, int segw, float lscale, hipStream_t st, int chain = CHAIN_NONE,
                const void* a16_ready = nullptr, void* c16_only = nullptr) {
  clipfs_gemm_args a = {};
  a.struct_size = sizeof(a);
  a.B_planes = Bp;
  a.b_format = cx.b_format;
  a.workspace = cx.ws;
  a.workspace_floats = cx.ws_floats;
  a.counters = cx.counters;
  a.counters_ints = cx.counters_ints;
  a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K;
  a.lda = K; a.ldb = K; a.ldc = N; a.alpha = 1.f;
  a.bias = bias; a.residual = res; a.ldres = N;
  a.act = act; a.aux_out = aux_out; a.aux_in = aux_in;
  a.lora_t = lt; a.lora_b = lb; a.lora_r = r; a.lora_nseg = nseg; a.lora_seg_width = segw; a.lora_scale = lscale;
  if (cx.b_format == 2 && Bp && cx.a16 && (K % 32) == 0 && (!lt || (segw % 128 == 0 && r <= 16))) {
    if (chain & CHAIN_IN16) {
      a.A_f16 = cx.c16;
    } else if (a16_ready) {
      a.A_f16 = a16_ready;  // the producing kernel already wrote the f16 image
    } else {
      CLIPFS_CHECK(clipfs_convert_f16(A, cx.a16, (size_t)M * K, st));
      a.A_f16 = cx.a16;
    }
    if (chain & CHAIN_OUT16) {
      a.C_f16 = cx.c16;
      a.C = nullptr;
    }
    a.aux_f16 = 1;  // fp16 storage of the saved pre-activation (only the f16 x f16 GEMMs read or write it)
    if (c16_only) {  // the result is kept as f16 alone (qkv in fp16 storage)
      a.C_f16 = c16_only;
      a.C = nullptr;
    }
  } else {
    CLIPFS_REQUIRE(!c16_only, "tower: the f16 x f16 GEMM is required for an f16-only result");
  }
  return clipfs_gemm_nt(&a, st);
}

}  // namespace clipfs

using namespace clipfs;

extern "C" size_t clipfs_tower_saved_floats(const clipfs_tower* t, int batch) {
  if (!t || batch <= 0 || t->struct_size != sizeof(clipfs_tower)) return 0;
  return saved_layout(t, (size_t)batch * t->seq).total * (size_t)t->layers;
}

extern "C" size_t clipfs_tower_scratch_floats(const clipfs_tower* t, int batch) {
  if (!t || batch <= 0 || t->struct_size != sizeof(clipfs_tower)) return 0;
  return scratch_layout(t, (size_t)batch * t->seq).total;
}

extern "C" size_t clipfs_tower_counter_ints(const clipfs_tower* t, int batch) {
  if (!t || batch <= 0 || t->struct_size != sizeof(clipfs_tower)) return 0;
  return scratch_layout(t, (size_t)batch * t->seq).counter_ints;
}

// Last block "one row per sequence" mode (clipfs_tower_fwd_rows / clipfs_tower_bwd_sparse): both directions must agree,
// the compact forward leaves the skipped rows of x_mid / u / stat2 unwritten.
static bool last_block_rows_ok(const clipfs_tower* t) {
  static const bool force_dense = getenv("CLIPFS_DENSE_BWD") && atoi(getenv("CLIPFS_DENSE_BWD")) != 0;  // A/B aid
  const clipfs_block& b = t->blocks[t->layers - 1];
  const bool lora_o = b.lora_a_o && (b.lora_mask & 8u);
  return !(force_dense || lora_o || t->seq < 8);
}

// rows == NULL: every row of every block.  rows != NULL (and last_block_rows_ok): the LAST block's output projection,
// LayerNorm 2 and MLP run on the `batch` rows c * seq + rows[c] only.
static int tower_fwd_impl(const clipfs_tower* t, float* x, const int32_t* rows, int batch, float* saved, float* scratch,
                          void* stream) {
  CLIPFS_CHECK(check_tower(t, batch));
  CLIPFS_REQUIRE(x && scratch, "tower_fwd: null buffer");
  hipStream_t st = (hipStream_t)stream;
  const int M = batch * t->seq, d = t->width, r = t->lora_r;
  const SavedLayout SL = saved_layout(t, (size_t)M);
  const ScratchLayout SC = scratch_layout(t, (size_t)M);
  const bool train = saved != nullptr;
  const TowerCtx cx = make_ctx(t, scratch, SC);
  CLIPFS_REQUIRE(!t->gemm_counters || t->gemm_counters_ints >= SC.counter_ints,
                 "tower: gemm_counters holds %zu ints, %zu needed", t->gemm_counters_ints, SC.counter_ints);
  // dropout follows the caller's train MODE (is_training(), lora_train_vlp.py:297-298), carried by a non-zero seed;
  // `saved` only decides whether activations are kept (a no-grad forward in train mode still drops)
  const uint64_t seed = t->dropout_seed;
  if (train) {
    hipError_t e = hipMemcpyAsync(saved + SL.x_in, x, (size_t)M * d * sizeof(float), hipMemcpyDeviceToDevice, st);
    CLIPFS_REQUIRE(e == hipSuccess, "tower_fwd: memcpy failed: %s", hipGetErrorString(e));
  }
  for (int l = 0; l < t->layers; ++l) {
    const clipfs_block& b = t->blocks[l];
    float* sv = train ? saved + (size_t)l * SL.total : nullptr;
    const float* x_in = train ? sv + SL.x_in : x;
    float* h1 = train ? sv + SL.h1 : scratch + SC.h;
    float* qkv = train ? sv + SL.qkv : scratch + SC.b3;
    float* att = train ? sv + SL.att : scratch + SC.b1;
    float* x_mid = train ? sv + SL.x_mid : x;
    float* t_qkv = train ? sv + SL.t_qkv : scratch + SC.dt;
    float* t_o = train ? sv + SL.t_o : scratch + SC.dt + al4((size_t)M * 3 * r);
    float* x_next = train ? (l + 1 < t->layers ? saved + (size_t)(l + 1) * SL.total + SL.x_in : x) : x;
    const unsigned qkv_mask = b.lora_a_qkv ? (b.lora_mask & 7u) : 0u;
    const bool lora_o = b.lora_a_o && (b.lora_mask & 8u);
    const uint32_t ds = t->dropout_stream0 + 4u * (uint32_t)l;

    // fp16 storage mode: producers write the f16 image of every GEMM operand next to (or instead of) the fp32 tensor
    void* h16 = cx.a16;                                                        // [M, d] halves: ln1 / attention / ln2 / dx
    void* dqkv16 = cx.a16 ? (void*)((char*)cx.a16 + (size_t)M * d * 2) : nullptr;  // [M, 3d] halves
    (void)dqkv16;
    void* keep = (train && qkv_mask && keep_bits_saved(t)) ? (void*)(sv + SL.keep) : nullptr;
    if (qkv_mask && clipfs_layernorm_fwd_lora_ok(d, r, 3)) {
      // small ranks: the adapter's down-projection rides on the LayerNorm pass (the row is in registers there)
      CLIPFS_CHECK(clipfs_layernorm_fwd_lora(x_in, d, b.ln1_g, b.ln1_b, h1, h16, train ? sv + SL.stat1 : nullptr,
                                             train ? sv + SL.stat1 + M : nullptr, M, d, 1e-5f, b.lora_a_qkv, t_qkv, r, 3, qkv_mask,
                                             t->lora_dropout, seed, ds, t->dropout_row0, keep, st));
    } else {
      if (h16)
        CLIPFS_CHECK(clipfs_layernorm_fwd_f16(x_in, d, b.ln1_g, b.ln1_b, h1, h16, train ? sv + SL.stat1 : nullptr,
                                              train ? sv + SL.stat1 + M : nullptr, M, d, 1e-5f, st));
      else
        CLIPFS_CHECK(clipfs_layernorm_fwd(x_in, d, b.ln1_g, b.ln1_b, h1, train ? sv + SL.stat1 : nullptr,
                                          train ? sv + SL.stat1 + M : nullptr, M, d, 1e-5f, st));
      if (qkv_mask)
        CLIPFS_CHECK(clipfs_lora_down(h1, b.lora_a_qkv, t_qkv, M, d, r, 3, qkv_mask, t->lora_dropout, seed, ds, t->dropout_row0,
                                      keep, st));
    }
    const bool q16 = qkv_f16(t);
    CLIPFS_CHECK(gemm(cx, h1, b.w_qkv, b.w_qkv_p, qkv, M, 3 * d, d, b.b_qkv, nullptr, 0, nullptr, nullptr, qkv_mask ? t_qkv : nullptr,
                      b.lora_b_qkv, r, 3, d, t->lora_scale, st, CHAIN_NONE, h16, q16 ? (void*)qkv : nullptr));
    const void* att16 = nullptr;
    if (f16_attention(t)) {
      CLIPFS_CHECK(clipfs_attention_f16_fwd(qkv, q16, att, h16, train ? sv + SL.lse : nullptr, batch, t->seq, t->heads,
                                            t->causal, st));
      att16 = h16;
    } else
      CLIPFS_CHECK(clipfs_attention_fwd(qkv, att, train ? sv + SL.lse : nullptr, batch, t->seq, t->heads, t->causal, st));
    if (rows && l == t->layers - 1) {
      // ---- the rest of the LAST block on one row per sequence: the head reads nothing else (jclip/model.py:121-124,
      // :213-214) and every remaining operation is row-wise.  Compact buffers live in the MLP scratch (Ms (13 d + 2)
      // floats <= M 4 d for seq >= 4); what the sparse backward gathers (x_mid, u, LayerNorm-2 statistics) is put back
      // at those rows of the saved tensors, the block output at those rows of x.
      const int seq = t->seq, Ms = batch;
      float* att_s = scratch + SC.big;
      float* xin_s = att_s + (size_t)Ms * d;
      float* xmid_s = xin_s + (size_t)Ms * d;
      float* h2_s = xmid_s + (size_t)Ms * d;
      float* xout_s = h2_s + (size_t)Ms * d;
      float* g_s = xout_s + (size_t)Ms * d;
      float* u_s = g_s + (size_t)Ms * 4 * d;
      float* mean_s = u_s + (size_t)Ms * 4 * d;
      float* rstd_s = mean_s + al4((size_t)Ms);
      CLIPFS_CHECK(clipfs_gather_seq_rows(att, (size_t)d, rows, att_s, Ms, seq, d, st));
      CLIPFS_CHECK(clipfs_gather_seq_rows(x_in, (size_t)d, rows, xin_s, Ms, seq, d, st));
      // fp16 storage mode: these `batch`-row products use the fp32 master weights (plane argument NULL) -- the f16 kernels
      // are built for tens of thousands of rows -- and the pre-GELU rows go back into the f16 tensor the dense path keeps
      const bool f16m = t->weight_format == 2;
      CLIPFS_CHECK(gemm(cx, att_s, b.w_o, f16m ? nullptr : b.w_o_p, xmid_s, Ms, d, d, b.b_o, xin_s, 0, nullptr, nullptr, nullptr, nullptr,
                        0, 0, 0, 0.f, st));
      CLIPFS_CHECK(clipfs_layernorm_fwd(xmid_s, d, b.ln2_g, b.ln2_b, h2_s, train ? mean_s : nullptr, train ? rstd_s : nullptr, Ms,
                                        d, 1e-5f, st));
      CLIPFS_CHECK(gemm(cx, h2_s, b.w_fc, f16m ? nullptr : b.w_fc_p, g_s, Ms, 4 * d, d, b.b_fc, nullptr, 1, train ? u_s : nullptr,
                        nullptr, nullptr, nullptr, 0, 0, 0, 0.f, st));
      CLIPFS_CHECK(gemm(cx, g_s, b.w_pr, f16m ? nullptr : b.w_pr_p, xout_s, Ms, d, 4 * d, b.b_pr, xmid_s, 0, nullptr, nullptr, nullptr,
                        nullptr, 0, 0, 0, 0.f, st));
      CLIPFS_CHECK(clipfs_put_seq_rows(xout_s, rows, x, (size_t)d, Ms, seq, d, st));
      if (train) {
        CLIPFS_CHECK(clipfs_put_seq_rows(xmid_s, rows, sv + SL.x_mid, (size_t)d, Ms, seq, d, st));
        if (f16m)
          CLIPFS_CHECK(clipfs_put_seq_rows_f16(u_s, rows, sv + SL.u, (size_t)4 * d, Ms, seq, 4 * d, st));
        else
          CLIPFS_CHECK(clipfs_put_seq_rows(u_s, rows, sv + SL.u, (size_t)4 * d, Ms, seq, 4 * d, st));
        CLIPFS_CHECK(clipfs_put_seq_rows(mean_s, rows, sv + SL.stat2, 1, Ms, seq, 1, st));
        CLIPFS_CHECK(clipfs_put_seq_rows(rstd_s, rows, sv + SL.stat2 + M, 1, Ms, seq, 1, st));
      }
      break;
    }
    if (lora_o)
      CLIPFS_CHECK(clipfs_lora_down(att, b.lora_a_o, t_o, M, d, r, 1, 1u, t->lora_dropout, seed, ds + 3, t->dropout_row0, nullptr, st));
    CLIPFS_CHECK(gemm(cx, att, b.w_o, b.w_o_p, x_mid, M, d, d, b.b_o, x_in, 0, nullptr, nullptr, lora_o ? t_o : nullptr, b.lora_b_o, r,
                      1, d, t->lora_scale, st, CHAIN_NONE, att16));
    float* h2 = scratch + SC.h;
    if (h16)  // only the f16 image is consumed (by the c_fc GEMM)
      CLIPFS_CHECK(clipfs_layernorm_fwd_f16(x_mid, d, b.ln2_g, b.ln2_b, nullptr, h16, train ? sv + SL.stat2 : nullptr,
                                            train ? sv + SL.stat2 + M : nullptr, M, d, 1e-5f, st));
    else
      CLIPFS_CHECK(clipfs_layernorm_fwd(x_mid, d, b.ln2_g, b.ln2_b, h2, train ? sv + SL.stat2 : nullptr,
                                        train ? sv + SL.stat2 + M : nullptr, M, d, 1e-5f, st));
    float* gbuf = scratch + SC.big;
    CLIPFS_CHECK(gemm(cx, h2, b.w_fc, b.w_fc_p, gbuf, M, 4 * d, d, b.b_fc, nullptr, 1, train ? sv + SL.u : nullptr, nullptr, nullptr,
                      nullptr, 0, 0, 0, 0.f, st, CHAIN_OUT16, h16));
    CLIPFS_CHECK(gemm(cx, gbuf, b.w_pr, b.w_pr_p, x_next, M, d, 4 * d, b.b_pr, x_mid, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0,
                      0.f, st, CHAIN_IN16));
  }
  return CLIPFS_OK;
}

extern "C" int clipfs_tower_fwd(const clipfs_tower* t, float* x, int batch, float* saved, float* scratch, void* stream) {
  return tower_fwd_impl(t, x, nullptr, batch, saved, scratch, stream);
}

extern "C" int clipfs_tower_fwd_rows(const clipfs_tower* t, float* x, const int32_t* rows, int batch, float* saved,
                                     float* scratch, void* stream) {
  CLIPFS_REQUIRE(t && rows, "tower_fwd_rows: null argument");
  CLIPFS_CHECK(check_tower(t, batch));
  return tower_fwd_impl(t, x, last_block_rows_ok(t) ? rows : nullptr, batch, saved, scratch, stream);
}

extern "C" int clipfs_tower_rows_mode(const clipfs_tower* t) {
  return (t && t->blocks && t->layers > 0 && last_block_rows_ok(t)) ? 1 : 0;
}

// blocks l_hi ... 0 of the backward; dx [batch*seq, width] in/out
static int tower_bwd_range(const clipfs_tower* t, float* dx, int batch, const float* saved, float* scratch,
                           int stop_at_input, hipStream_t st, int l_hi) {
  const int M = batch * t->seq, d = t->width, r = t->lora_r;
  const SavedLayout SL = saved_layout(t, (size_t)M);
  const ScratchLayout SC = scratch_layout(t, (size_t)M);
  const TowerCtx cx = make_ctx(t, scratch, SC);
  const uint64_t seed = t->dropout_seed;
  void* h16 = cx.a16;                                                           // f16 image of dx (then of d ln-out ...)
  void* dqkv16 = cx.a16 ? (void*)((char*)cx.a16 + (size_t)M * d * 2) : nullptr;  // [M, 3d] halves
  if (h16) CLIPFS_CHECK(clipfs_convert_f16(dx, h16, (size_t)M * d, st));       // later images come from LayerNorm backward
  for (int l = l_hi; l >= 0; --l) {
    const clipfs_block& b = t->blocks[l];
    const float* sv = saved + (size_t)l * SL.total;
    CLIPFS_REQUIRE(b.w_pr_t && b.w_fc_t && b.w_o_t && b.w_qkv_t, "tower_bwd: block %d lacks transposed weights", l);
    CLIPFS_REQUIRE(t->weight_format != 2 || (b.w_pr_t_p && b.w_fc_t_p && b.w_o_t_p && b.w_qkv_t_p),
                   "tower_bwd: block %d lacks f16 copies of the transposed weights", l);
    const unsigned qkv_mask = b.lora_a_qkv ? (b.lora_mask & 7u) : 0u;
    const bool lora_o = b.lora_a_o && (b.lora_mask & 8u);
    const uint32_t ds = t->dropout_stream0 + 4u * (uint32_t)l;
    float* du = scratch + SC.big;
    float* dh = scratch + SC.h;
    float* datt = scratch + SC.b1;
    float* dqkv = scratch + SC.b3;
    float* dt = scratch + SC.dt;
    float* work = scratch + SC.work;
    // MLP: du = (dx Wpr) * gelu'(u) ; dh2 = du Wfc ; dx += LN2'(dh2)
    CLIPFS_CHECK(gemm(cx, dx, b.w_pr_t, b.w_pr_t_p, du, M, 4 * d, d, nullptr, nullptr, 2, nullptr, sv + SL.u, nullptr, nullptr, 0, 0, 0,
                      0.f, st, CHAIN_OUT16, h16));
    CLIPFS_CHECK(gemm(cx, du, b.w_fc_t, b.w_fc_t_p, dh, M, d, 4 * d, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0.f,
                      st, CHAIN_IN16));
    if (h16)
      CLIPFS_CHECK(clipfs_layernorm_bwd_f16(dh, sv + SL.x_mid, d, b.ln2_g, sv + SL.stat2, sv + SL.stat2 + M, dx, dx, h16, d,
                                            M, d, st));
    else
      CLIPFS_CHECK(clipfs_layernorm_bwd(dh, sv + SL.x_mid, d, b.ln2_g, sv + SL.stat2, sv + SL.stat2 + M, dx, dx, d, M, d,
                                        st));
    // attention output projection.  fp16 storage mode without an o-projection adapter: its only consumer is the f16
    // attention backward, which rounds dO to f16 for its MFMA operands anyway -- the GEMM writes the f16 image alone (into
    // the same scratch slot): a quarter of the epilogue bytes of an fp32 result, half the bytes the attention kernels stage
    const bool datt16 = f16_attention(t) && !lora_o && h16 != nullptr;
    CLIPFS_CHECK(gemm(cx, dx, b.w_o_t, b.w_o_t_p, datt, M, d, d, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0.f,
                      st, CHAIN_NONE, h16, datt16 ? (void*)datt : nullptr));
    if (lora_o) {
      CLIPFS_REQUIRE(b.g_lora_a_o && b.g_lora_b_o, "tower_bwd: block %d o-LoRA gradient slots missing", l);
      CLIPFS_CHECK(clipfs_lora_bwd(dx, sv + SL.att, sv + SL.t_o, b.lora_a_o, b.lora_b_o, dt, b.g_lora_a_o, b.g_lora_b_o,
                                   datt, M, d, d, r, 1, 1u, t->lora_scale, t->lora_dropout, seed, ds + 3, t->dropout_row0, nullptr, work, st));
    }
    // (the D_i work vector of the long-sequence kernels lives in the dt scratch slot's neighbour: reuse `dh`, dead here)
    const void* dqkv16_ready = nullptr;
    // fp16 storage mode: the dgrad GEMM and (matrix-core shapes) the adapter backward read the f16 image of dqkv, so the
    // attention backward does not write the fp32 tensor at all (404 MB per ViT-L/14 block at 128 images)
    const bool dy16 = f16_attention(t) && dqkv16 && (!qkv_mask || clipfs_lora_bwd_f16dy_ok(d, d, r, 3));
    if (f16_attention(t)) {
      CLIPFS_CHECK(clipfs_attention_f16_bwd(sv + SL.qkv, qkv_f16(t), datt, datt16 ? 1 : 0, sv + SL.att, sv + SL.lse,
                                            dy16 ? nullptr : dqkv, dqkv16, dh, batch, t->seq, t->heads, t->causal, st));
      dqkv16_ready = dqkv16;
    } else
      CLIPFS_CHECK(clipfs_attention_bwd(sv + SL.qkv, datt, sv + SL.att, sv + SL.lse, dqkv, dh, batch, t->seq, t->heads,
                                        t->causal, st));
    const bool need_dx = !(l == 0 && stop_at_input);
    if (need_dx)
      CLIPFS_CHECK(gemm(cx, dqkv, b.w_qkv_t, b.w_qkv_t_p, dh, M, d, 3 * d, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0,
                        0.f, st, CHAIN_NONE, dqkv16_ready));
    if (qkv_mask) {
      CLIPFS_REQUIRE(b.g_lora_a_qkv && b.g_lora_b_qkv, "tower_bwd: block %d LoRA gradient slots missing", l);
      if (dy16)
        CLIPFS_CHECK(clipfs_lora_bwd_f16dy(dqkv16, sv + SL.h1, sv + SL.t_qkv, b.lora_a_qkv, b.lora_b_qkv, dt, b.g_lora_a_qkv,
                                           b.g_lora_b_qkv, need_dx ? dh : nullptr, M, d, d, r, 3, qkv_mask, t->lora_scale,
                                           t->lora_dropout, seed, ds, t->dropout_row0, keep_bits_saved(t) ? (const void*)(sv + SL.keep) : nullptr, work, st));
      else
        CLIPFS_CHECK(clipfs_lora_bwd(dqkv, sv + SL.h1, sv + SL.t_qkv, b.lora_a_qkv, b.lora_b_qkv, dt, b.g_lora_a_qkv,
                                     b.g_lora_b_qkv, need_dx ? dh : nullptr, M, d, d, r, 3, qkv_mask, t->lora_scale,
                                     t->lora_dropout, seed, ds, t->dropout_row0, keep_bits_saved(t) ? (const void*)(sv + SL.keep) : nullptr, work, st));
    }
    if (need_dx) {
      if (h16)
        CLIPFS_CHECK(clipfs_layernorm_bwd_f16(dh, sv + SL.x_in, d, b.ln1_g, sv + SL.stat1, sv + SL.stat1 + M, dx, dx, h16, d,
                                              M, d, st));
      else
        CLIPFS_CHECK(clipfs_layernorm_bwd(dh, sv + SL.x_in, d, b.ln1_g, sv + SL.stat1, sv + SL.stat1 + M, dx, dx, d, M, d,
                                          st));
    }
  }
  return CLIPFS_OK;
}


extern "C" int clipfs_tower_bwd(const clipfs_tower* t, float* dx, int batch, const float* saved, float* scratch,
                                int stop_at_input, void* stream) {
  CLIPFS_CHECK(check_tower(t, batch));
  CLIPFS_REQUIRE(dx && saved && scratch, "tower_bwd: null buffer");
  const ScratchLayout SC = scratch_layout(t, (size_t)batch * t->seq);
  CLIPFS_REQUIRE(!t->gemm_counters || t->gemm_counters_ints >= SC.counter_ints,
                 "tower: gemm_counters holds %zu ints, %zu needed", t->gemm_counters_ints, SC.counter_ints);
  return tower_bwd_range(t, dx, batch, saved, scratch, stop_at_input, (hipStream_t)stream, t->layers - 1);
}

extern "C" int clipfs_tower_bwd_sparse(const clipfs_tower* t, const float* dxs, const int32_t* rows, float* dx, int batch,
                                       const float* saved, float* scratch, int stop_at_input, void* stream) {
  CLIPFS_CHECK(check_tower(t, batch));
  CLIPFS_REQUIRE(dxs && rows && dx && saved && scratch, "tower_bwd_sparse: null buffer");
  hipStream_t st = (hipStream_t)stream;
  const int seq = t->seq, M = batch * seq, d = t->width, r = t->lora_r, Ms = batch;
  const SavedLayout SL = saved_layout(t, (size_t)M);
  const ScratchLayout SC = scratch_layout(t, (size_t)M);
  CLIPFS_REQUIRE(!t->gemm_counters || t->gemm_counters_ints >= SC.counter_ints,
                 "tower: gemm_counters holds %zu ints, %zu needed", t->gemm_counters_ints, SC.counter_ints);
  const int l = t->layers - 1;
  const clipfs_block& b = t->blocks[l];
  if (!last_block_rows_ok(t)) {  // dense fall-back: the row gradients scattered into zeros
    CLIPFS_CHECK(clipfs_scatter_rows(dxs, rows, dx, batch, seq, d, st));
    return tower_bwd_range(t, dx, batch, saved, scratch, stop_at_input, st, l);
  }
  CLIPFS_REQUIRE(b.w_pr_t && b.w_fc_t && b.w_o_t && b.w_qkv_t, "tower_bwd: block %d lacks transposed weights", l);
  const TowerCtx cx = make_ctx(t, scratch, SC);
  const float* sv = saved + (size_t)l * SL.total;
  // compact (one row per sequence) buffers live in the MLP scratch, which this block does not otherwise use:
  // Ms (8 d + 4 d + 2) floats <= M 4 d for seq >= 3
  float* u_s = scratch + SC.big;
  float* du_s = u_s + (size_t)Ms * 4 * d;
  float* xmid_s = du_s + (size_t)Ms * 4 * d;
  float* dh_s = xmid_s + (size_t)Ms * d;
  float* dxm_s = dh_s + (size_t)Ms * d;
  float* datt_s = dxm_s + (size_t)Ms * d;
  float* mean_s = datt_s + (size_t)Ms * d;
  float* rstd_s = mean_s + al4((size_t)Ms);
  // ---- MLP and output projection on the `batch` rows that carry gradient ----
  // fp16 storage mode: the `batch`-row products use the fp32 master weights (plane argument NULL); the saved pre-GELU
  // activation is an f16 tensor there
  const bool f16m = t->weight_format == 2;
  if (f16m)
    CLIPFS_CHECK(clipfs_gather_seq_rows_f16(sv + SL.u, (size_t)4 * d, rows, u_s, Ms, seq, 4 * d, st));
  else
    CLIPFS_CHECK(clipfs_gather_seq_rows(sv + SL.u, (size_t)4 * d, rows, u_s, Ms, seq, 4 * d, st));
  CLIPFS_CHECK(clipfs_gather_seq_rows(sv + SL.x_mid, (size_t)d, rows, xmid_s, Ms, seq, d, st));
  CLIPFS_CHECK(clipfs_gather_seq_rows(sv + SL.stat2, 1, rows, mean_s, Ms, seq, 1, st));
  CLIPFS_CHECK(clipfs_gather_seq_rows(sv + SL.stat2 + M, 1, rows, rstd_s, Ms, seq, 1, st));
  CLIPFS_CHECK(gemm(cx, dxs, b.w_pr_t, f16m ? nullptr : b.w_pr_t_p, du_s, Ms, 4 * d, d, nullptr, nullptr, 2, nullptr, u_s, nullptr, nullptr,
                    0, 0, 0, 0.f, st));
  CLIPFS_CHECK(gemm(cx, du_s, b.w_fc_t, f16m ? nullptr : b.w_fc_t_p, dh_s, Ms, d, 4 * d, nullptr, nullptr, 0, nullptr, nullptr, nullptr,
                    nullptr, 0, 0, 0, 0.f, st));
  CLIPFS_CHECK(clipfs_layernorm_bwd(dh_s, xmid_s, d, b.ln2_g, mean_s, rstd_s, dxs, dxm_s, d, Ms, d, st));
  CLIPFS_CHECK(gemm(cx, dxm_s, b.w_o_t, f16m ? nullptr : b.w_o_t_p, datt_s, Ms, d, d, nullptr, nullptr, 0, nullptr, nullptr, nullptr,
                    nullptr, 0, 0, 0, 0.f, st));
  // ---- attention and the QKV projection see every row again ----
  float* dh = scratch + SC.h;
  float* datt = scratch + SC.b1;
  float* dqkv = scratch + SC.b3;
  float* dt = scratch + SC.dt;
  float* work = scratch + SC.work;
  CLIPFS_CHECK(clipfs_scatter_rows(datt_s, rows, datt, batch, seq, d, st));
  const unsigned qkv_mask = b.lora_a_qkv ? (b.lora_mask & 7u) : 0u;
  const void* dqkv16_ready = nullptr;
  void* dqkv16 = (f16_attention(t) && cx.a16) ? (void*)((char*)cx.a16 + (size_t)M * d * 2) : nullptr;  // [M, 3d] halves
  const bool dy16 = dqkv16 && (!qkv_mask || clipfs_lora_bwd_f16dy_ok(d, d, r, 3));  // as in tower_bwd_range
  if (f16_attention(t)) {
    CLIPFS_CHECK(clipfs_attention_f16_bwd(sv + SL.qkv, qkv_f16(t), datt, 0, sv + SL.att, sv + SL.lse, dy16 ? nullptr : dqkv, dqkv16,
                                          dh, batch, seq, t->heads, t->causal, st));
    dqkv16_ready = dqkv16;
  } else
    CLIPFS_CHECK(clipfs_attention_bwd(sv + SL.qkv, datt, sv + SL.att, sv + SL.lse, dqkv, dh, batch, seq, t->heads, t->causal, st));
  const uint32_t ds = t->dropout_stream0 + 4u * (uint32_t)l;
  const bool need_dx = !(l == 0 && stop_at_input);
  if (need_dx)
    CLIPFS_CHECK(gemm(cx, dqkv, b.w_qkv_t, b.w_qkv_t_p, dh, M, d, 3 * d, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0,
                      0, 0.f, st, CHAIN_NONE, dqkv16_ready));
  if (qkv_mask) {
    CLIPFS_REQUIRE(b.g_lora_a_qkv && b.g_lora_b_qkv, "tower_bwd: block %d LoRA gradient slots missing", l);
    if (dy16)
      CLIPFS_CHECK(clipfs_lora_bwd_f16dy(dqkv16, sv + SL.h1, sv + SL.t_qkv, b.lora_a_qkv, b.lora_b_qkv, dt, b.g_lora_a_qkv,
                                         b.g_lora_b_qkv, need_dx ? dh : nullptr, M, d, d, r, 3, qkv_mask, t->lora_scale,
                                         t->lora_dropout, t->dropout_seed, ds, t->dropout_row0, keep_bits_saved(t) ? (const void*)(sv + SL.keep) : nullptr,
                                         work, st));
    else
      CLIPFS_CHECK(clipfs_lora_bwd(dqkv, sv + SL.h1, sv + SL.t_qkv, b.lora_a_qkv, b.lora_b_qkv, dt, b.g_lora_a_qkv, b.g_lora_b_qkv,
                                   need_dx ? dh : nullptr, M, d, d, r, 3, qkv_mask, t->lora_scale, t->lora_dropout,
                                   t->dropout_seed, ds, t->dropout_row0, keep_bits_saved(t) ? (const void*)(sv + SL.keep) : nullptr, work, st));
  }
  if (!need_dx) return CLIPFS_OK;
  CLIPFS_CHECK(clipfs_layernorm_bwd(dh, sv + SL.x_in, d, b.ln1_g, sv + SL.stat1, sv + SL.stat1 + M, nullptr, dx, d, M, d, st));
  CLIPFS_CHECK(clipfs_add_seq_rows(dxm_s, rows, dx, batch, seq, d, st));  // the residual branch around the attention
  return tower_bwd_range(t, dx, batch, saved, scratch, stop_at_input, st, l - 1);
}
