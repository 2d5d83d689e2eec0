// Exact-fp32 attention for SHORT sequences (L <= 96: ViT-B/32's 50 / 54 tokens, the 77-token text tower) on
// v_mfma_f32_16x16x4_f32: forward, dQ pass, dK/dV pass (jclip/mha.py:55-83,439-458).  Same mathematics and the same
// transposed-score scheme as attention_mfma.hip (32x32x2 tiles, kept for 96 < L <= 288); what changes and why:
//
//   * 16-token tiles.  77 tokens are 5 tiles (80) instead of 3 x 32 (96), and the causal mask skips whole 16 x 16
//     tiles: 15 of 25 tile pairs carry work for the text tower where the 32-token tiling computed 6 of 9 (3840 vs 6144
//     score elements for 3003 useful ones).  One wave per own tile: 4 waves at L = 50, 5 at L = 77.
//   * EVERY MFMA operand comes from LDS or registers.  Each tensor of the other side has ONE image in LDS, transposed
//     ([64 features][TP tokens], TP = 16 NT + 8): products that contract over FEATURES (scores, dP) read it with
//     ds_read_b32 (lane = token, one feature row per MFMA step), products that contract over TOKENS (PV, dQ, dK, dV) with
//     ds_read_b128 (lane = feature row, 4 consecutive tokens = 4 MFMA steps).  TP = 8 (mod 16) makes the b128 reads
//     conflict free and the b32 reads 2-way.  The 32x32 kernels fetched the row-major tiles from global memory / L2
//     inside the tile loop and waited ~1-2 us per tile for them (26-32 % MFMA-busy, profiles/r01).
//   * no online softmax: a query's scores over all <= 96 keys stay in registers (4 per key tile), so the row maximum
//     and sum are computed once -- no per-tile rescale of the output accumulators, one lane exchange per query;
//   * independent accumulator chains are interleaved (the 16x16x4 MFMA issues every 32 cycles but a dependent one waits
//     40): the score phase runs the chains of all key tiles step by step, the token-axis products the four feature tiles.
//
// Register map of v_mfma_f32_16x16x4_f32 (D = A B + C, 16 x 16 x 4, wave64): lane l supplies A[i = l & 15][k = l >> 4],
// B[k = l >> 4][j = l & 15] and holds D[i = 4 (l >> 4) + r][j = l & 15] in register r = 0..3.  As everywhere in this
// library the assignment of a sum's terms to K slots is free as long as A and B agree: in a feature contraction lane
// group g = l >> 4 takes features 16 g .. 16 g + 15 (step s -> feature 16 g + s), in a token contraction group g takes
// tokens 4 g .. 4 g + 3 of the tile (step r -> token 4 g + r), which is exactly the D layout of the score tile.
#include "common.h"

#include <stdlib.h>

namespace clipfs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int A16_HD = 64;
constexpr float A16_LOG2E = 1.4426950408889634f;
constexpr int A16_MAX_TILES = 6;  // L <= 96

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// transposed fp32 image of a head's [L][64] slice: tr[f * TP + tok]; tokens [L, 16 NT) zero-filled (a masked
// probability is exactly 0, so the padding must not hold NaN / Inf bit patterns).  Staging is split in two so that a
// workgroup can keep the NEXT head's rows travelling (global -> registers) while it computes the current one: these
// kernels move ~40 bytes per MFMA and run at the speed the memory system delivers, i.e. at the bytes kept in flight.
template <int NT>
__device__ __forceinline__ void a16_fetch(const float* __restrict__ src, size_t ld, int L, f32x4 (&v)[4]) {
  constexpr int Lp = 16 * NT;
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // Lp * 16 chunks of 4 features over 64 NT threads = 4 each
    const int idx = (int)threadIdx.x + k * 64 * NT;
    const int c = idx / Lp, tok = idx - c * Lp;  // token fastest: conflict-free ds_write_b32 in a16_put
    // unconditional load of a clamped row (padding rows are zeroed in a16_put): straight-line code, so the compiler
    // can COUNT these loads in its s_waitcnt vmcnt(N) and leave them in flight behind older loads it has to wait for
    v[k] = *reinterpret_cast<const f32x4*>(src + (size_t)min(tok, L - 1) * ld + 4 * c);
  }
}
template <int NT>
__device__ __forceinline__ void a16_put(float* tr, const f32x4 (&v)[4], int L) {
  constexpr int Lp = 16 * NT, TP = Lp + 8;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = (int)threadIdx.x + k * 64 * NT;
    const int c = idx / Lp, tok = idx - c * Lp;
#pragma unroll
    for (int j = 0; j < 4; ++j) tr[(4 * c + j) * TP + tok] = tok < L ? v[k][j] : 0.f;
  }
}

// the own tile's rows as B operands of a feature contraction: w[s] = row (t0 + lane & 15), feature 16 (lane >> 4) + s
__device__ __forceinline__ void a16_load_own(const float* __restrict__ src, size_t ld, int tok, int lane, float (&w)[16]) {
  const float* p = src + (size_t)tok * ld + 16 * (lane >> 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e) w[4 * j + e] = v[e];
  }
}

// NA feature contractions at once, their chains interleaved step by step:
//   acc[t][r] = sum_f other[16 t + 4 g + r][f] * own[lane & 15][f]        (other rows from the transposed image)
template <int NA, int TP>
__device__ __forceinline__ void a16_scores(const float* tr, int lane, const float (&own)[16], f32x4 (&acc)[A16_MAX_TILES]) {
  const float* p = tr + (16 * (lane >> 4)) * TP + (lane & 15);
#pragma unroll
  for (int t = 0; t < NA; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 16; ++s)
#pragma unroll
    for (int t = 0; t < NA; ++t) acc[t] = mfma16(p[s * TP + 16 * t], own[s], acc[t]);
}

// token contraction over tile t of the other side: o[ft] += sum_r tr[16 ft + (lane & 15)][16 t + 4 g + r] * w[r]
template <int TP>
__device__ __forceinline__ void a16_accum(const float* tr, int lane, int t, const f32x4& w, f32x4 (&o)[4]) {
  const float* p = tr + (lane & 15) * TP + 16 * t + 4 * (lane >> 4);
  f32x4 a[4];
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) a[ft] = *reinterpret_cast<const f32x4*>(p + 16 * ft * TP);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) o[ft] = mfma16(a[ft][r], w[r], o[ft]);
}

// o[ft][r] = feature 16 ft + 4 (lane >> 4) + r of the lane's own row: four 16-byte stores
__device__ __forceinline__ void a16_store(float* __restrict__ dst, const f32x4 (&o)[4], float scale, int lane) {
  float* p = dst + 4 * (lane >> 4);
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) *reinterpret_cast<f32x4*>(p + 16 * ft) = o[ft] * scale;
}

__device__ __forceinline__ float a16_allreduce_max(float v) {  // over the 4 lane groups holding one own row
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float a16_allreduce_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ---------------------------------------------------------------------------------------------------------
// One workgroup per (batch, head).  Tried and measured slower (MI355X, cfg-2 shapes): persistent workgroups that walk
// several heads with the next head's rows prefetched into registers during the current head's MFMAs (the extra 32-48
// registers cost a wave per SIMD and the per-head barriers serialise the causal tower's uneven waves: text forward
// 64 -> 98 us), and reading the score operands one four-step block ahead (no change: the other waves of the SIMD
// already cover the LDS latency).
#define A16_SWITCH_SCORES(na, IMG, OWN, ACC)                                   \
  switch (na) {                                                                \
    case 1: a16_scores<1, TP>(IMG, lane, OWN, ACC); break;               \
    case 2: a16_scores<(NT >= 2 ? 2 : 1), TP>(IMG, lane, OWN, ACC); break; \
    case 3: a16_scores<(NT >= 3 ? 3 : 1), TP>(IMG, lane, OWN, ACC); break; \
    case 4: a16_scores<(NT >= 4 ? 4 : 1), TP>(IMG, lane, OWN, ACC); break; \
    case 5: a16_scores<(NT >= 5 ? 5 : 1), TP>(IMG, lane, OWN, ACC); break; \
    default: a16_scores<(NT >= 6 ? 6 : 1), TP>(IMG, lane, OWN, ACC); break; \
  }

// forward: own = 16 queries per wave.  S^T = K Q^T (lane = query, registers = keys), exact softmax over the
// registers, O^T = V^T P^T.
template <int NT>
__global__ __launch_bounds__(64 * NT) void attention16_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                  float* __restrict__ lse, int L, int H, int causal) {
  extern __shared__ __attribute__((aligned(16))) float a16_smem[];
  constexpr int TP = 16 * NT + 8;
  float* sKt = a16_smem;
  float* sVt = sKt + 64 * TP;
  const int lane = threadIdx.x & 63;
  const int qt = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int d = H * A16_HD;
  const size_t ld = (size_t)3 * d;
  const int g = lane >> 4;
  const int q_tok = 16 * qt + (lane & 15);
  const float c = 0.125f * A16_LOG2E;
  const int klim = causal ? q_tok : L - 1;
  const int na = causal ? qt + 1 : NT;  // key tiles with work (wave-uniform)
  auto head = [&](int item) { return qkv + (size_t)(item / H) * L * ld + (size_t)(item % H) * A16_HD; };
  f32x4 kst[4], vst[4];
  const int item = blockIdx.x;  // = b * H + h
  a16_fetch<NT>(head(item) + d, ld, L, kst);
  a16_fetch<NT>(head(item) + 2 * d, ld, L, vst);
  const int b = item / H, h = item % H;
  float qf[16];
  a16_load_own(head(item), ld, min(q_tok, L - 1), lane, qf);
  a16_put<NT>(sKt, kst, L);
  a16_put<NT>(sVt, vst, L);
  __syncthreads();
  f32x4 s[A16_MAX_TILES];
  A16_SWITCH_SCORES(na, sKt, qf, s)
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t < na) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * t + 4 * g + r;
        s[t][r] = (key < L && key <= klim) ? s[t][r] * c : -INFINITY;
        m = fmaxf(m, s[t][r]);
      }
    }
  m = a16_allreduce_max(m);  // key 0 is never masked: m is finite
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t < na) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[t][r] = __builtin_amdgcn_exp2f(s[t][r] - m);
        l += s[t][r];
      }
    }
  l = a16_allreduce_sum(l);
  f32x4 o[4];
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) o[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t < na) a16_accum<TP>(sVt, lane, t, s[t], o);
  if (q_tok < L) {
    a16_store(out + ((size_t)b * L + q_tok) * d + h * A16_HD, o, 1.f / l, lane);
    if (lse && g == 0) lse[((size_t)b * H + h) * L + q_tok] = (m + log2f(l)) * (1.f / A16_LOG2E);
  }
}

// dQ pass (own = 16 queries).  S^T = K Q^T ; P^T = exp2(S^T c - lse) ; dP^T = V dO^T ; dS^T = P^T (dP^T - D) / 8 ;
// dQ^T = K^T dS^T.  D_i = dO_i . O_i from the own rows in registers.
template <int NT>
__device__ __forceinline__ void attention16_bwd_q_body(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                       const float* __restrict__ out, const float* __restrict__ lse,
                                                       float* __restrict__ dqkv, int L, int H, int causal, int item,
                                                       float* a16_smem) {
  constexpr int TP = 16 * NT + 8;
  float* sKt = a16_smem;
  float* sVt = sKt + 64 * TP;
  const int lane = threadIdx.x & 63;
  const int qt = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int d = H * A16_HD;
  const size_t ld = (size_t)3 * d;
  const int g = lane >> 4;
  const int q_tok = 16 * qt + (lane & 15), q_cl = min(q_tok, L - 1);
  const float c = 0.125f * A16_LOG2E;
  const int klim = causal ? q_tok : L - 1;
  const int na = causal ? qt + 1 : NT;
  auto head = [&](int item) { return qkv + (size_t)(item / H) * L * ld + (size_t)(item % H) * A16_HD; };
  f32x4 kst[4], vst[4];
  a16_fetch<NT>(head(item) + d, ld, L, kst);
  a16_fetch<NT>(head(item) + 2 * d, ld, L, vst);
  const int b = item / H, h = item % H;
  float qf[16], gf[16];
  a16_load_own(head(item), ld, q_cl, lane, qf);
  a16_load_own(dout + (size_t)b * L * d + (size_t)h * A16_HD, (size_t)d, q_cl, lane, gf);
  float Di = 0.f;
  {
    const float* op = out + ((size_t)b * L + q_cl) * d + h * A16_HD + 16 * g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 ov = *reinterpret_cast<const f32x4*>(op + 4 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) Di = fmaf(gf[4 * j + e], ov[e], Di);
    }
  }
  const float lse2 = lse[((size_t)b * H + h) * L + q_cl] * A16_LOG2E;
  a16_put<NT>(sKt, kst, L);
  a16_put<NT>(sVt, vst, L);
  Di = a16_allreduce_sum(Di);
  __syncthreads();
  f32x4 s[A16_MAX_TILES], dp[A16_MAX_TILES];
  A16_SWITCH_SCORES(na, sKt, qf, s)
  A16_SWITCH_SCORES(na, sVt, gf, dp)
  f32x4 acc[4];
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) acc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t < na) {
      f32x4 ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * t + 4 * g + r;
        const float p = (key < L && key <= klim) ? __builtin_amdgcn_exp2f(s[t][r] * c - lse2) : 0.f;
        ds[r] = p * (dp[t][r] - Di) * 0.125f;
      }
      a16_accum<TP>(sKt, lane, t, ds, acc);
    }
  if (q_tok < L) a16_store(dqkv + ((size_t)b * L + q_tok) * ld + h * A16_HD, acc, 1.f, lane);
}

// dK/dV pass (own = 16 keys).  S = Q K^T ; P = exp2(S c - lse) ; dP = dO V^T ; dS = P (dP - D) / 8 ;
// dV^T = dO^T P ; dK^T = Q^T dS.  lse and D vary with the register index (rows = queries): 4-float groups from LDS;
// D is recomputed here from the staged dO rows and the matching O rows.
template <int NT>
__device__ __forceinline__ void attention16_bwd_kv_body(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                        const float* __restrict__ out, const float* __restrict__ lse,
                                                        float* __restrict__ dqkv, int L, int H, int causal, int item,
                                                        float* a16_smem) {
  constexpr int Lp = 16 * NT, TP = Lp + 8;
  float* sQt = a16_smem;
  float* sGt = sQt + 64 * TP;
  float* sLse = sGt + 64 * TP;  // [Lp], log2 units
  float* sD = sLse + Lp;
  float* sPart = sD + Lp;       // [16 feature chunks][Lp]: partial dO . O sums of the staging threads
  const int lane = threadIdx.x & 63;
  const int kt = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int d = H * A16_HD;
  const size_t ld = (size_t)3 * d;
  const int g = lane >> 4;
  const int k_tok = 16 * kt + (lane & 15), k_cl = min(k_tok, L - 1);
  const float c = 0.125f * A16_LOG2E;
  // query tiles with work: t >= kt under the causal mask (wave-uniform); they run as tiles 0 .. na-1 of a shifted image
  const int t_first = causal ? kt : 0;
  const int na = NT - t_first;
  const float* sQ1 = sQt + 16 * t_first;
  const float* sG1 = sGt + 16 * t_first;
  auto qhead = [&](int item) { return qkv + (size_t)(item / H) * L * ld + (size_t)(item % H) * A16_HD; };
  auto ghead = [&](int item) { return dout + (size_t)(item / H) * L * d + (size_t)(item % H) * A16_HD; };
  const int vtok = min((int)threadIdx.x, L - 1);  // thread i < 16 NT carries element i of the head's lse / D vectors
  f32x4 qst[4], gst[4], ost[4];
  a16_fetch<NT>(qhead(item), ld, L, qst);
  a16_fetch<NT>(ghead(item), (size_t)d, L, gst);
  a16_fetch<NT>(out + (size_t)(item / H) * L * d + (size_t)(item % H) * A16_HD, (size_t)d, L, ost);
  const int b = item / H, h = item % H;
  float kf[16], vf[16];
  a16_load_own(qhead(item) + d, ld, k_cl, lane, kf);
  a16_load_own(qhead(item) + 2 * d, ld, k_cl, lane, vf);
  const float lse_i = lse[(size_t)item * L + vtok];  // item = b * H + h
  a16_put<NT>(sQt, qst, L);
  a16_put<NT>(sGt, gst, L);
  // D_i = dO_i . O_i from the rows this pass stages anyway (the dQ pass computes its own copy in registers, so the two
  // passes do not depend on each other and run as ONE launch): each staging thread holds 4 features of 4 (chunk, token)
  // pairs; the 16 chunk sums of a token are added in chunk order by the thread that owns the token => reproducible
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = (int)threadIdx.x + k * 64 * NT;
    const int c = idx / Lp, tok = idx - c * Lp;
    sPart[c * Lp + tok] = fmaf(gst[k][3], ost[k][3], fmaf(gst[k][2], ost[k][2], fmaf(gst[k][1], ost[k][1], gst[k][0] * ost[k][0])));
  }
  __syncthreads();
  if ((int)threadIdx.x < Lp) {
    float d_i = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) d_i += sPart[c * Lp + threadIdx.x];
    sLse[threadIdx.x] = (int)threadIdx.x < L ? lse_i * A16_LOG2E : 0.f;
    sD[threadIdx.x] = (int)threadIdx.x < L ? d_i : 0.f;
  }
  __syncthreads();
  f32x4 s[A16_MAX_TILES], dp[A16_MAX_TILES];
  A16_SWITCH_SCORES(na, sQ1, kf, s)
  A16_SWITCH_SCORES(na, sG1, vf, dp)
  f32x4 av[4], ak[4];
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) {
    av[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    ak[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int u = 0; u < NT; ++u)
    if (u < na) {
      const int t = t_first + u;
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + 16 * t + 4 * g);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + 16 * t + 4 * g);
      f32x4 p, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = 16 * t + 4 * g + r;
        p[r] = (qi < L && (!causal || qi >= k_tok)) ? __builtin_amdgcn_exp2f(s[u][r] * c - l4[r]) : 0.f;
        ds[r] = p[r] * (dp[u][r] - d4[r]) * 0.125f;
      }
      a16_accum<TP>(sGt, lane, t, p, av);
      a16_accum<TP>(sQt, lane, t, ds, ak);
    }
  if (k_tok < L) {
    float* kp = dqkv + ((size_t)b * L + k_tok) * ld + d + h * A16_HD;
    a16_store(kp, ak, 1.f, lane);
    a16_store(kp + d, av, 1.f, lane);
  }
}

// The backward as ONE launch: workgroup 2 i takes the dQ pass of head i, workgroup 2 i + 1 its dK/dV pass (neighbours in
// dispatch order, so the head's q, k, v, dO rows are fetched from HBM once and found in L2 by the other).  Round 2 ran
// them as two launches with D handed through memory; at the per-rank sizes of the 8-GPU step each was ~15 us of work
// behind its own launch.
template <int NT>
__global__ __launch_bounds__(64 * NT) void attention16_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                  const float* __restrict__ out, const float* __restrict__ lse,
                                                                  float* __restrict__ dqkv, int L, int H, int causal) {
  extern __shared__ __attribute__((aligned(16))) float a16_smem[];
  const int item = blockIdx.x >> 1;  // = b * H + h
  if (blockIdx.x & 1)
    attention16_bwd_kv_body<NT>(qkv, dout, out, lse, dqkv, L, H, causal, item, a16_smem);
  else
    attention16_bwd_q_body<NT>(qkv, dout, out, lse, dqkv, L, H, causal, item, a16_smem);
}

// ---- host side (called from attention_mfma.hip) -------------------------------------------------------------

bool attention16_enabled(int seq) {
  static const int cfg = getenv("CLIPFS_ATTN16") ? atoi(getenv("CLIPFS_ATTN16")) : 1;  // 0: the 32x32 kernels (A/B aid)
  return cfg != 0 && seq <= 16 * A16_MAX_TILES;
}

static size_t a16_lds(int nt, bool vectors) {  // vectors: lse, D and the 16 partial-D rows of the dK/dV pass
  return ((size_t)2 * 64 * (16 * nt + 8) + (vectors ? 18 * (size_t)16 * nt : 0)) * sizeof(float);
}

template <int NT>
static int a16_fwd(const float* qkv, float* out, float* lse, int batch, int seq, int heads, int causal, hipStream_t st) {
  static bool attr = false;
  if (!attr && a16_lds(NT, false) > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention16_fwd_kernel<NT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)a16_lds(NT, false));
    attr = true;
  }
  hipLaunchKernelGGL(attention16_fwd_kernel<NT>, dim3(batch * heads), dim3(64 * NT), a16_lds(NT, false), st, qkv, out, lse, seq,
                     heads, causal);
  return launch_status();
}

template <int NT>
static int a16_bwd(const float* qkv, const float* dout, const float* out, const float* lse, float* dqkv, float* work,
                   int batch, int seq, int heads, int causal, hipStream_t st) {
  (void)work;  // D_i no longer travels through memory
  static bool attr = false;
  if (!attr && a16_lds(NT, true) > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention16_bwd_kernel<NT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)a16_lds(NT, true));
    attr = true;
  }
  hipLaunchKernelGGL(attention16_bwd_kernel<NT>, dim3(2 * batch * heads), dim3(64 * NT), a16_lds(NT, true), st, qkv, dout, out,
                     lse, dqkv, seq, heads, causal);
  return launch_status();
}

int attention16_fwd(const float* qkv, float* out, float* lse, int batch, int seq, int heads, int causal, hipStream_t st) {
  switch ((seq + 15) / 16) {
    case 1: return a16_fwd<1>(qkv, out, lse, batch, seq, heads, causal, st);
    case 2: return a16_fwd<2>(qkv, out, lse, batch, seq, heads, causal, st);
    case 3: return a16_fwd<3>(qkv, out, lse, batch, seq, heads, causal, st);
    case 4: return a16_fwd<4>(qkv, out, lse, batch, seq, heads, causal, st);
    case 5: return a16_fwd<5>(qkv, out, lse, batch, seq, heads, causal, st);
    default: return a16_fwd<6>(qkv, out, lse, batch, seq, heads, causal, st);
  }
}

int attention16_bwd(const float* qkv, const float* dout, const float* out, const float* lse, float* dqkv, float* work,
                    int batch, int seq, int heads, int causal, hipStream_t st) {
  switch ((seq + 15) / 16) {
    case 1: return a16_bwd<1>(qkv, dout, out, lse, dqkv, work, batch, seq, heads, causal, st);
    case 2: return a16_bwd<2>(qkv, dout, out, lse, dqkv, work, batch, seq, heads, causal, st);
    case 3: return a16_bwd<3>(qkv, dout, out, lse, dqkv, work, batch, seq, heads, causal, st);
    case 4: return a16_bwd<4>(qkv, dout, out, lse, dqkv, work, batch, seq, heads, causal, st);
    case 5: return a16_bwd<5>(qkv, dout, out, lse, dqkv, work, batch, seq, heads, causal, st);
    default: return a16_bwd<6>(qkv, dout, out, lse, dqkv, work, batch, seq, heads, causal, st);
  }
}

}  // namespace clipfs
