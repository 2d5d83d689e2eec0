// Exact-fp32 attention on the matrix cores (v_mfma_f32_32x32x2_f32, the instruction of the dense GEMMs): the
// function of attention.hip (jclip/mha.py:55-83,439-458) for sequences up to 288 tokens, forward, dQ and dK/dV.
// Same structure as attention_f16.hip, with fp32 operands end to end (products exact in fp32, fp32 accumulate; only
// the summation order differs from the VALU kernels):
//
//   * one workgroup per (batch, head), one wave per 32-token tile of the "own" side (queries in the forward and the
//     dQ pass, keys in the dK/dV pass); the own rows sit in registers as MFMA B operands (lane = row lane & 31,
//     features 32 (lane >> 5) .. + 31: the K-slot <-> feature assignment of an MFMA sum is free as long as A and B
//     agree, so each half-wave simply takes one contiguous half of the 64 features);
//   * scores are produced TRANSPOSED (S^T[other][own] = Other Own^T), so a lane holds 16 other-side tokens of ONE
//     own token: softmax statistics are per-lane scalars (+ one exchange with lane ^ 32) and P^T / dS^T are already
//     the B operand of the token-axis product; its A operand is read from the TRANSPOSED image of the other side
//     kept in LDS ([64 features][tokens], 16-byte reads of 4 consecutive tokens);
//   * the row-major other-side tiles (A operand of the score product) are read straight from global memory /
//     L2 (each lane 128 contiguous bytes), so LDS holds only the transposed images: 26 KB per head at L = 77.
#include "common.h"

#include <stdlib.h>

namespace clipfs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int AM_HD = 64;
constexpr float AM_LOG2E = 1.4426950408889634f;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float am_xor32(float v) { return __shfl_xor(v, 32, 64); }

// transposed fp32 image of a head's [L][64] slice: tr[f * TP + tok], TP = Lp + 4 (conflict-free ds_read_b128 for
// lanes = consecutive features); tokens [L, Lp) zero-filled.
__device__ __forceinline__ void am_stage_T(const float* __restrict__ src, size_t ld, int L, int Lp, float* tr) {
  const int TP = Lp + 4;
  for (int idx = threadIdx.x; idx < Lp * 16; idx += (int)blockDim.x) {
    const int c = idx / Lp, tok = idx - c * Lp;  // token fastest: conflict-free ds_write_b32
    f32x4 v;
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (tok < L) v = *reinterpret_cast<const f32x4*>(src + (size_t)tok * ld + 4 * c);
#pragma unroll
    for (int j = 0; j < 4; ++j) tr[(4 * c + j) * TP + tok] = v[j];
  }
}

// own rows as B operands: own[i] = row (t0 + lane & 31), feature 32 (lane >> 5) + i
__device__ __forceinline__ void am_load_own(const float* __restrict__ src, size_t ld, int t0, int L, int lane, float (&own)[32]) {
  const float* p = src + (size_t)min(t0 + (lane & 31), L - 1) * ld + 32 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e) own[4 * j + e] = v[e];
  }
}

// acc[other token t0 + reg-row][own token = lane] = sum_f other[tok][f] own[lane][f]; other rows from global memory
__device__ __forceinline__ f32x16 am_scores_T(const float* __restrict__ other, size_t ld, int t0, int L, int lane,
                                              const float (&own)[32]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* p = other + (size_t)min(t0 + (lane & 31), L - 1) * ld + 32 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = mfma32(v[e], own[4 * j + e], acc);
  }
  return acc;
}

// acc[feature dt*32 + reg-row][own = lane] += sum over the tile's 32 tokens of tr[feature][t0 + tok] * w[tok][own],
// w = the lane's 16 registers (tokens (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
__device__ __forceinline__ f32x16 am_accum_T(const float* tr, int TP, int dt, int t0, int lane, const f32x16& w, f32x16 acc) {
  const float* p = tr + (dt * 32 + (lane & 31)) * TP + t0 + 4 * (lane >> 5);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + 8 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = mfma32(v[e], w[4 * g + e], acc);
  }
  return acc;
}

__device__ __forceinline__ void am_store_T(float* __restrict__ dst, const f32x16 (&o)[2], float scale, int lane) {
  // lane holds features dt*32 + 8 g + 4 (lane >> 5) + {0..3} of its own row: 16-byte stores
  float* p = dst + 4 * (lane >> 5);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = o[t][4 * g + j] * scale;
      *reinterpret_cast<f32x4*>(p + 32 * t + 8 * g) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attention_mfma_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                 float* __restrict__ lse, int L, int H, int causal) {
  extern __shared__ __attribute__((aligned(16))) float am_smem[];
  const int Lp = (L + 31) & ~31, TP = Lp + 4;
  float* sVt = am_smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = (int)blockDim.x >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * AM_HD;
  const size_t ld = (size_t)3 * d;
  const float* q0 = qkv + (size_t)b * L * ld + (size_t)h * AM_HD;
  am_stage_T(q0 + 2 * d, ld, L, Lp, sVt);
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  const float c = 0.125f * AM_LOG2E;
  for (int qt = wave; qt * 32 < L; qt += nw) {
    const int q_tok = qt * 32 + fr;
    float qf[32];
    am_load_own(q0, ld, qt * 32, L, lane, qf);
    f32x16 o[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int kend = causal ? min(L, qt * 32 + 32) : L;
    const int klim = causal ? q_tok : L - 1;
    for (int k0 = 0; k0 < kend; k0 += 32) {
      f32x16 s = am_scores_T(q0 + d, ld, k0, L, lane, qf);
      float mt = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        s[r] = (key < L && key <= klim) ? s[r] * c : -INFINITY;
        mt = fmaxf(mt, s[r]);
      }
      mt = fmaxf(mt, am_xor32(mt));
      const float mn = fmaxf(m, mt);
      const float f = __builtin_amdgcn_exp2f(m - mn);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - mn);
        ps += s[r];
      }
      l = l * f + ps;
      m = mn;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= f;
        o[t] = am_accum_T(sVt, TP, t, k0, lane, s, o[t]);
      }
    }
    l += am_xor32(l);
    if (q_tok < L) {
      am_store_T(out + ((size_t)b * L + q_tok) * d + h * AM_HD, o, 1.f / l, lane);
      if (lse && fh == 0) lse[((size_t)b * H + h) * L + q_tok] = (m + log2f(l)) * (1.f / AM_LOG2E);
    }
  }
}

// dQ pass (own = queries).  S^T = K Q^T ; P^T = exp2(S^T c - lse) ; dP^T = V dO^T ; dS^T = P^T (dP^T - D) / 8 ;
// dQ^T += K^T dS^T.  Also writes D_i = dO_i . O_i for the second pass.
__global__ __launch_bounds__(256) void attention_mfma_bwd_q_kernel(const float* __restrict__ qkv,
                                                                   const float* __restrict__ dout,
                                                                   const float* __restrict__ out,
                                                                   const float* __restrict__ lse, float* __restrict__ dqkv,
                                                                   float* __restrict__ Dbuf, int L, int H, int causal) {
  extern __shared__ __attribute__((aligned(16))) float am_smem[];
  const int Lp = (L + 31) & ~31, TP = Lp + 4;
  float* sKt = am_smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = (int)blockDim.x >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * AM_HD;
  const size_t ld = (size_t)3 * d;
  const float* q0 = qkv + (size_t)b * L * ld + (size_t)h * AM_HD;
  am_stage_T(q0 + d, ld, L, Lp, sKt);
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  const float c = 0.125f * AM_LOG2E;
  for (int qt = wave; qt * 32 < L; qt += nw) {
    const int q_tok = qt * 32 + fr, q_cl = min(q_tok, L - 1);
    float qf[32], gf[32];
    am_load_own(q0, ld, qt * 32, L, lane, qf);
    am_load_own(dout + (size_t)b * L * d + (size_t)h * AM_HD, (size_t)d, qt * 32, L, lane, gf);
    float Di = 0.f;
    {
      const float* op = out + ((size_t)b * L + q_cl) * d + h * AM_HD + 32 * fh;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 ov = *reinterpret_cast<const f32x4*>(op + 4 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e) Di = fmaf(gf[4 * j + e], ov[e], Di);
      }
      Di += am_xor32(Di);
    }
    const float lse2 = lse[((size_t)b * H + h) * L + q_cl] * AM_LOG2E;
    if (q_tok < L && fh == 0) Dbuf[((size_t)b * H + h) * L + q_tok] = Di;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int kend = causal ? min(L, qt * 32 + 32) : L;
    const int klim = causal ? q_tok : L - 1;
    for (int k0 = 0; k0 < kend; k0 += 32) {
      const f32x16 s = am_scores_T(q0 + d, ld, k0, L, lane, qf);
      const f32x16 dp = am_scores_T(q0 + 2 * d, ld, k0, L, lane, gf);
      f32x16 ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        const float p = (key < L && key <= klim) ? __builtin_amdgcn_exp2f(s[r] * c - lse2) : 0.f;
        ds[r] = p * (dp[r] - Di) * 0.125f;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t] = am_accum_T(sKt, TP, t, k0, lane, ds, acc[t]);
    }
    if (q_tok < L) am_store_T(dqkv + ((size_t)b * L + q_tok) * ld + h * AM_HD, acc, 1.f, lane);
  }
}

// dK/dV pass (own = keys).  S = Q K^T ; P = exp2(S c - lse) ; dP = dO V^T ; dS = P (dP - D) / 8 ;
// dV^T += dO^T P ; dK^T += Q^T dS.  lse and D vary with the register index (rows = queries): 4-float groups from LDS.
__global__ __launch_bounds__(256) void attention_mfma_bwd_kv_kernel(const float* __restrict__ qkv,
                                                                    const float* __restrict__ dout,
                                                                    const float* __restrict__ lse,
                                                                    const float* __restrict__ Dbuf,
                                                                    float* __restrict__ dqkv, int L, int H, int causal) {
  extern __shared__ __attribute__((aligned(16))) float am_smem[];
  const int Lp = (L + 31) & ~31, TP = Lp + 4;
  float* sQt = am_smem;
  float* sGt = sQt + 64 * TP;
  float* sLse = sGt + 64 * TP;  // [Lp], log2 units
  float* sD = sLse + Lp;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = (int)blockDim.x >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * AM_HD;
  const size_t ld = (size_t)3 * d;
  const float* q0 = qkv + (size_t)b * L * ld + (size_t)h * AM_HD;
  const float* g0 = dout + (size_t)b * L * d + (size_t)h * AM_HD;
  am_stage_T(q0, ld, L, Lp, sQt);
  am_stage_T(g0, (size_t)d, L, Lp, sGt);
  for (int i = threadIdx.x; i < Lp; i += (int)blockDim.x) {
    sLse[i] = i < L ? lse[((size_t)b * H + h) * L + i] * AM_LOG2E : 0.f;
    sD[i] = i < L ? Dbuf[((size_t)b * H + h) * L + i] : 0.f;
  }
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  const float c = 0.125f * AM_LOG2E;
  for (int kt = wave; kt * 32 < L; kt += nw) {
    const int k_tok = kt * 32 + fr;
    float kf[32], vf[32];
    am_load_own(q0 + d, ld, kt * 32, L, lane, kf);
    am_load_own(q0 + 2 * d, ld, kt * 32, L, lane, vf);
    f32x16 av[2], ak[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        av[t][r] = 0.f;
        ak[t][r] = 0.f;
      }
    for (int i0 = causal ? kt * 32 : 0; i0 < L; i0 += 32) {
      const f32x16 s = am_scores_T(q0, ld, i0, L, lane, kf);
      const f32x16 dp = am_scores_T(g0, (size_t)d, i0, L, lane, vf);
      f32x16 p, ds;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + i0 + 8 * g + 4 * fh);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + i0 + 8 * g + 4 * fh);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          const int qi = i0 + 8 * g + 4 * fh + j;
          p[r] = (qi < L && (!causal || qi >= k_tok)) ? __builtin_amdgcn_exp2f(s[r] * c - l4[j]) : 0.f;
          ds[r] = p[r] * (dp[r] - d4[j]) * 0.125f;
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        av[t] = am_accum_T(sGt, TP, t, i0, lane, p, av[t]);
        ak[t] = am_accum_T(sQt, TP, t, i0, lane, ds, ak[t]);
      }
    }
    if (k_tok < L) {
      float* kp = dqkv + ((size_t)b * L + k_tok) * ld + d + h * AM_HD;
      am_store_T(kp, ak, 1.f, lane);
      am_store_T(kp + d, av, 1.f, lane);
    }
  }
}

// ---- host side (called from attention.hip) ----------------------------------------------------------------

// attention_mfma16.hip: 16-token tiles on v_mfma_f32_16x16x4_f32, every operand in LDS (sequences up to 96 tokens)
bool attention16_enabled(int seq);
int attention16_fwd(const float* qkv, float* out, float* lse, int batch, int seq, int heads, int causal, hipStream_t st);
int attention16_bwd(const float* qkv, const float* dout, const float* out, const float* lse, float* dqkv, float* work,
                    int batch, int seq, int heads, int causal, hipStream_t st);

bool attention_mfma_enabled() {
  static const int cfg = getenv("CLIPFS_ATTN_MFMA") ? atoi(getenv("CLIPFS_ATTN_MFMA")) : 1;  // 0: VALU kernels (A/B aid)
  return cfg != 0;
}

static int am_threads(int seq) {
  const int tiles = (seq + 31) / 32;
  return 64 * (tiles < 4 ? tiles : 4);
}

static size_t am_lds(int seq, int images, bool vectors) {
  const int Lp = (seq + 31) & ~31;
  return ((size_t)images * 64 * (Lp + 4) + (vectors ? 2 * (size_t)Lp : 0)) * sizeof(float);
}

int attention_mfma_fwd(const float* qkv, float* out, float* lse, int batch, int seq, int heads, int causal, hipStream_t st) {
  if (attention16_enabled(seq)) return attention16_fwd(qkv, out, lse, batch, seq, heads, causal, st);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_mfma_fwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(attention_mfma_fwd_kernel, dim3(batch * heads), dim3(am_threads(seq)), am_lds(seq, 1, false), st, qkv,
                     out, lse, seq, heads, causal);
  return launch_status();
}

int attention_mfma_bwd(const float* qkv, const float* dout, const float* out, const float* lse, float* dqkv, float* work,
                       int batch, int seq, int heads, int causal, hipStream_t st) {
  if (attention16_enabled(seq)) return attention16_bwd(qkv, dout, out, lse, dqkv, work, batch, seq, heads, causal, st);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_mfma_bwd_q_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_mfma_bwd_kv_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(attention_mfma_bwd_q_kernel, dim3(batch * heads), dim3(am_threads(seq)), am_lds(seq, 1, false), st, qkv,
                     dout, out, lse, dqkv, work, seq, heads, causal);
  CLIPFS_CHECK(launch_status());
  hipLaunchKernelGGL(attention_mfma_bwd_kv_kernel, dim3(batch * heads), dim3(am_threads(seq)), am_lds(seq, 2, true), st, qkv,
                     dout, lse, work, dqkv, seq, heads, causal);
  return launch_status();
}

}  // namespace clipfs
