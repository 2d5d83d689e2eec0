// MFMA attention for the fp16 storage mode (cfg-5: ViT-L/14 image tower L = 257, text tower L = 77 causal; head_dim 64): the same function
// as attention.hip (jclip/mha.py:55-83,439-458) with the two contractions on v_mfma_f32_32x32x16_f16, softmax
// statistics and every accumulator in fp32.
//
// One workgroup per (batch, head).  The head's K, V (forward / dQ pass) or Q, dO (dK/dV pass) are converted to f16
// once and kept in LDS in two images: row-major [token][64] for the contraction over the feature axis (lane =
// token row, ds_read_b128) and transposed [64][token] for the contraction over the token axis (ds_read_b64).
// A wave owns a tile of 32 "own" tokens whose operands stay in registers and walks over the 32-token tiles of the
// other side.
//
// The layout trick that keeps softmax in registers: scores are computed TRANSPOSED, S^T[key][query] = K Q^T, so the
// MFMA result layout (column = lane & 31, 16 rows per lane) gives every lane 16 keys of ONE query.  Row max / row
// sum are then in-lane reductions plus one exchange with lane ^ 32, the running (m, l) and the rescale factor are
// per-lane scalars, and P^T is -- without any data movement -- exactly the B operand of the next product
// O^T[d][query] += V^T[d][key] P^T[key][query]  (the MFMA sums over k in any order as long as A and B agree, so the
// A operand reads V^T at the keys the lane's registers hold: 4 consecutive keys at +0 and 4 at +8).
#include "common.h"

namespace clipfs {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int AF_HD = 64;
constexpr int AF_MAXL = 288;        // 9 tiles of 32 tokens
constexpr int AF_ROW = 72;          // row-major image: halves per token row (64 + 8 pad: conflict-free ds_read_b128)
// transposed image: halves per feature row = Lp + 4 (= 4 * odd for Lp % 32 == 0: conflict-free ds_read_b64)
constexpr float AF_LOG2E = 1.4426950408889634f;

__device__ __forceinline__ f16x8 cvt8(const f32x4& a, const f32x4& b) {
  f16x8 h;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = (_Float16)a[j];
    h[4 + j] = (_Float16)b[j];
  }
  return h;
}

// eight consecutive features as f16: converted from fp32, or taken as they are from an f16 tensor (fp16 storage of qkv)
__device__ __forceinline__ f16x8 load8(const float* p) {
  return cvt8(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4));
}
__device__ __forceinline__ f16x8 load8(const _Float16* p) { return *reinterpret_cast<const f16x8*>(p); }

// Stage `L` token rows (fp32 or f16, 64 features at `src`, row stride ld) as f16 into a row-major image and/or a transposed
// image; tokens [L, Lp) are zero-filled so masked lanes multiply finite values.
// One work item = 4 consecutive tokens x 8 features, loaded ONCE (4 x 16 / 32 bytes) for both images: the row-major
// image takes 4 ds_write_b128, the transposed one 8 ds_write_b64 (feature row f: the 4 tokens are consecutive halves).
// Round 2 wrote the transposed image with 32 ds_write_b16 per item and loaded the source once per image; with one
// workgroup per CU (the images fill the LDS) that staging ran un-overlapped in front of every head's MFMA phase.
template <typename T>
__device__ __forceinline__ void stage_head(const T* __restrict__ src, size_t ld, int L, int Lp, _Float16* rowm,
                                           _Float16* tr) {
  const int TP = Lp + 4;
  const int groups = Lp >> 2;  // Lp % 32 == 0
  for (int idx = threadIdx.x; idx < groups * 8; idx += (int)blockDim.x) {
    const int c = idx / groups, tok0 = (idx - c * groups) * 4;  // token group fastest: consecutive lanes write consecutive
    f16x8 h[4];                                                  // 8-byte pieces of one feature row of the transposed image
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int j = 0; j < 8; ++j) h[q][j] = (_Float16)0.f;
      if (tok0 + q < L) h[q] = load8(src + (size_t)(tok0 + q) * ld + 8 * c);
    }
    if (rowm) {
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f16x8*>(rowm + (tok0 + q) * AF_ROW + 8 * c) = h[q];
    }
    if (tr) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        *reinterpret_cast<f16x4*>(tr + (8 * c + j) * TP + tok0) = f16x4{h[0][j], h[1][j], h[2][j], h[3][j]};
    }
  }
}

// The wave's own 32 rows as MFMA operands: frag[s] = row (t0 + lane & 31), features 16 s + 8 (lane >> 5) .. + 7.
template <typename T>
__device__ __forceinline__ void load_own(const T* __restrict__ src, size_t ld, int t0, int L, int lane, f16x8 (&frag)[4]) {
  const T* p = src + (size_t)min(t0 + (lane & 31), L - 1) * ld + 8 * (lane >> 5);
#pragma unroll
  for (int s = 0; s < 4; ++s) frag[s] = load8(p + 16 * s);
}

// acc[tile rows = other-side tokens t0.. (register index)][cols = own tokens (lane)] = other_rowmajor . own^T
__device__ __forceinline__ f32x16 scores_T(const _Float16* rowm, int t0, int lane, const f16x8 (&own)[4]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const _Float16* p = rowm + (t0 + (lane & 31)) * AF_ROW + 8 * (lane >> 5);
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(p + 16 * s), own[s], acc, 0, 0, 0);
  return acc;
}

// A operand of the token-axis contraction for key slice u (16 tokens) of the tile at t0: feature row (dt*32 + lane&31),
// tokens t0 + 16 u + 4 fh + {0..3} and + 8 + {0..3}  (the tokens registers 8u .. 8u+7 of the lane hold).
__device__ __forceinline__ f16x8 load_T(const _Float16* tr, int TP, int dt, int t0, int u, int lane) {
  const _Float16* p = tr + (dt * 32 + (lane & 31)) * TP + t0 + 16 * u + 4 * (lane >> 5);
  const f16x4 lo = *reinterpret_cast<const f16x4*>(p);
  const f16x4 hi = *reinterpret_cast<const f16x4*>(p + 8);
  f16x8 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = lo[j];
    v[4 + j] = hi[j];
  }
  return v;
}

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
template <typename TQ>
__global__ __launch_bounds__(320) void attention_f16_fwd_kernel(const TQ* __restrict__ qkv, float* __restrict__ out,
                                                                float* __restrict__ lse, int L, int H, int causal,
                                                                _Float16* __restrict__ out16) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int Lp = (L + 31) & ~31, TP = Lp + 4;
  _Float16* sK = reinterpret_cast<_Float16*>(smem_raw);  // [Lp][AF_ROW]
  _Float16* sVt = sK + Lp * AF_ROW;                       // [64][TP]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = (int)blockDim.x >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * AF_HD;
  const size_t ld = (size_t)3 * d;
  const TQ* q0 = qkv + (size_t)b * L * ld + (size_t)h * AF_HD;
  stage_head(q0 + d, ld, L, Lp, sK, nullptr);
  stage_head(q0 + 2 * d, ld, L, Lp, nullptr, sVt);
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  const float c = 0.125f * AF_LOG2E;  // scores in log2 units
  for (int qt = wave; qt * 32 < L; qt += nw) {
    const int q_tok = qt * 32 + fr;
    f16x8 qf[4];
    load_own(q0, ld, qt * 32, L, lane, qf);
    f32x16 o[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int kend = causal ? min(L, qt * 32 + 32) : L;  // causal: key tiles up to the diagonal tile
    const int klim = causal ? q_tok : L - 1;             // last visible key of this lane's query
    for (int k0 = 0; k0 < kend; k0 += 32) {
      f32x16 s = scores_T(sK, k0, lane, qf);
      float mt = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        s[r] = (key < L && key <= klim) ? s[r] * c : -INFINITY;
        mt = fmaxf(mt, s[r]);
      }
      mt = fmaxf(mt, xor32(mt));
      const float mn = fmaxf(m, mt);
      const float f = __builtin_amdgcn_exp2f(m - mn);
      float ps = 0.f;
      f16x8 pf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[r] - mn);
        ps += p;
        pf[r >> 3][r & 7] = (_Float16)p;
      }
      l = l * f + ps;
      m = mn;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(load_T(sVt, TP, t, k0, u, lane), pf[u], o[t], 0, 0, 0);
      }
    }
    l += xor32(l);
    const float inv = 1.f / l;
    if (q_tok < L) {
      float* op = out + ((size_t)b * L + q_tok) * d + h * AF_HD + 4 * fh;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = o[t][4 * g4 + j] * inv;
          *reinterpret_cast<f32x4*>(op + 32 * t + 8 * g4) = v;
          if (out16)
            *reinterpret_cast<f16x4*>(out16 + ((size_t)b * L + q_tok) * d + h * AF_HD + 4 * fh + 32 * t + 8 * g4) =
                f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        }
      if (lse && fh == 0) lse[((size_t)b * H + h) * L + q_tok] = (m + log2f(l)) * (1.f / AF_LOG2E);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// backward, pass 1: dQ (and D_i = dO_i . O_i for pass 2).  Own side = queries; K (both images) and V in LDS.
//   S^T = K Q^T ; P^T = exp(S^T c - lse) ; dP^T = V dO^T ; dS^T = P^T (dP^T - D) / 8 ; dQ^T += K^T dS^T
// ---------------------------------------------------------------------------------------------------------
template <typename TQ, typename TG>
__global__ __launch_bounds__(576) void attention_f16_bwd_q_kernel(const TQ* __restrict__ qkv,
                                                                  const TG* __restrict__ dout,
                                                                  const float* __restrict__ out,
                                                                  const float* __restrict__ lse, float* __restrict__ dqkv,
                                                                  float* __restrict__ Dbuf, int L, int H, int causal,
                                                                  _Float16* __restrict__ dqkv16) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int Lp = (L + 31) & ~31, TP = Lp + 4;
  _Float16* sK = reinterpret_cast<_Float16*>(smem_raw);
  _Float16* sV = sK + Lp * AF_ROW;
  _Float16* sKt = sV + Lp * AF_ROW;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = (int)blockDim.x >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * AF_HD;
  const size_t ld = (size_t)3 * d;
  const TQ* q0 = qkv + (size_t)b * L * ld + (size_t)h * AF_HD;
  stage_head(q0 + d, ld, L, Lp, sK, sKt);
  stage_head(q0 + 2 * d, ld, L, Lp, sV, nullptr);
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  const float c = 0.125f * AF_LOG2E;
  for (int qt = wave; qt * 32 < L; qt += nw) {
    const int q_tok = qt * 32 + fr, q_cl = min(q_tok, L - 1);
    f16x8 qf[4], gf[4];
    load_own(q0, ld, qt * 32, L, lane, qf);
    float Di = 0.f;
    {
      const TG* gp = dout + ((size_t)b * L + q_cl) * d + h * AF_HD + 8 * fh;
      const float* op = out + ((size_t)b * L + q_cl) * d + h * AF_HD + 8 * fh;
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        f32x4 g0, g1;
        if constexpr (sizeof(TG) == 2) {  // dO given as its f16 image (written by the output-projection dgrad GEMM)
          gf[sidx] = *reinterpret_cast<const f16x8*>(gp + 16 * sidx);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            g0[j] = (float)gf[sidx][j];
            g1[j] = (float)gf[sidx][4 + j];
          }
        } else {
          g0 = *reinterpret_cast<const f32x4*>(gp + 16 * sidx);
          g1 = *reinterpret_cast<const f32x4*>(gp + 16 * sidx + 4);
          gf[sidx] = cvt8(g0, g1);
        }
        const f32x4 o0 = *reinterpret_cast<const f32x4*>(op + 16 * sidx), o1 = *reinterpret_cast<const f32x4*>(op + 16 * sidx + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) Di += g0[j] * o0[j] + g1[j] * o1[j];
      }
      Di += xor32(Di);
    }
    const float lse2 = lse[((size_t)b * H + h) * L + q_cl] * AF_LOG2E;
    if (q_tok < L && fh == 0) Dbuf[((size_t)b * H + h) * L + q_tok] = Di;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int kend = causal ? min(L, qt * 32 + 32) : L;
    const int klim = causal ? q_tok : L - 1;
    for (int k0 = 0; k0 < kend; k0 += 32) {
      const f32x16 s = scores_T(sK, k0, lane, qf);
      const f32x16 dp = scores_T(sV, k0, lane, gf);
      f16x8 dsf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        const float p = (key < L && key <= klim) ? __builtin_amdgcn_exp2f(s[r] * c - lse2) : 0.f;
        dsf[r >> 3][r & 7] = (_Float16)(p * (dp[r] - Di) * 0.125f);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(load_T(sKt, TP, t, k0, u, lane), dsf[u], acc[t], 0, 0, 0);
    }
    if (q_tok < L) {
      float* op = dqkv + ((size_t)b * L + q_tok) * ld + h * AF_HD + 4 * fh;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[t][4 * g4 + j];
          if (dqkv) *reinterpret_cast<f32x4*>(op + 32 * t + 8 * g4) = v;
          if (dqkv16)
            *reinterpret_cast<f16x4*>(dqkv16 + ((size_t)b * L + q_tok) * ld + h * AF_HD + 4 * fh + 32 * t + 8 * g4) =
                f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// backward, pass 2: dK, dV.  Own side = keys; Q and dO (both images each) plus lse and D in LDS.
//   S = Q K^T ; P = exp(S c - lse) ; dP = dO V^T ; dS = P (dP - D) / 8 ; dV^T += dO^T P ; dK^T += Q^T dS
// (rows of the MFMA result = queries, so lse and D vary with the register index: read as 4-float groups from LDS).
// ---------------------------------------------------------------------------------------------------------
template <typename TQ, typename TG>
__global__ __launch_bounds__(576) void attention_f16_bwd_kv_kernel(const TQ* __restrict__ qkv,
                                                                   const TG* __restrict__ dout,
                                                                   const float* __restrict__ lse,
                                                                   const float* __restrict__ Dbuf,
                                                                   float* __restrict__ dqkv, int L, int H, int causal,
                                                                   _Float16* __restrict__ dqkv16) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int Lp = (L + 31) & ~31, TP = Lp + 4;
  _Float16* sQ = reinterpret_cast<_Float16*>(smem_raw);
  _Float16* sG = sQ + Lp * AF_ROW;
  _Float16* sQt = sG + Lp * AF_ROW;
  _Float16* sGt = sQt + 64 * TP;
  float* sLse = reinterpret_cast<float*>(sGt + 64 * TP);  // [Lp] in log2 units
  float* sD = sLse + Lp;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = (int)blockDim.x >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * AF_HD;
  const size_t ld = (size_t)3 * d;
  const TQ* q0 = qkv + (size_t)b * L * ld + (size_t)h * AF_HD;
  stage_head(q0, ld, L, Lp, sQ, sQt);
  stage_head(dout + (size_t)b * L * d + (size_t)h * AF_HD, (size_t)d, L, Lp, sG, sGt);
  for (int i = threadIdx.x; i < Lp; i += (int)blockDim.x) {
    sLse[i] = i < L ? lse[((size_t)b * H + h) * L + i] * AF_LOG2E : 0.f;
    sD[i] = i < L ? Dbuf[((size_t)b * H + h) * L + i] : 0.f;
  }
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  const float c = 0.125f * AF_LOG2E;
  for (int kt = wave; kt * 32 < L; kt += nw) {
    const int k_tok = kt * 32 + fr;
    f16x8 kf[4], vf[4];
    load_own(q0 + d, ld, kt * 32, L, lane, kf);
    load_own(q0 + 2 * d, ld, kt * 32, L, lane, vf);
    f32x16 av[2], ak[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        av[t][r] = 0.f;
        ak[t][r] = 0.f;
      }
    for (int i0 = causal ? kt * 32 : 0; i0 < L; i0 += 32) {  // causal: query tiles from the diagonal tile on
      const f32x16 s = scores_T(sQ, i0, lane, kf);
      const f32x16 dp = scores_T(sG, i0, lane, vf);
      f16x8 pf[2], dsf[2];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + i0 + 8 * g4 + 4 * fh);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + i0 + 8 * g4 + 4 * fh);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g4 + j;
          const int qi = i0 + 8 * g4 + 4 * fh + j;
          const float p = (qi < L && (!causal || qi >= k_tok)) ? __builtin_amdgcn_exp2f(s[r] * c - l4[j]) : 0.f;
          pf[r >> 3][r & 7] = (_Float16)p;
          dsf[r >> 3][r & 7] = (_Float16)(p * (dp[r] - d4[j]) * 0.125f);
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          av[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(load_T(sGt, TP, t, i0, u, lane), pf[u], av[t], 0, 0, 0);
          ak[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(load_T(sQt, TP, t, i0, u, lane), dsf[u], ak[t], 0, 0, 0);
        }
    }
    if (k_tok < L) {
      float* kp = dqkv + ((size_t)b * L + k_tok) * ld + d + h * AF_HD + 4 * fh;
      float* vp = kp + d;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f32x4 k4, v4;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            k4[j] = ak[t][4 * g4 + j];
            v4[j] = av[t][4 * g4 + j];
          }
          if (dqkv) {
            *reinterpret_cast<f32x4*>(kp + 32 * t + 8 * g4) = k4;
            *reinterpret_cast<f32x4*>(vp + 32 * t + 8 * g4) = v4;
          }
          if (dqkv16) {
            _Float16* k16 = dqkv16 + ((size_t)b * L + k_tok) * ld + d + h * AF_HD + 4 * fh + 32 * t + 8 * g4;
            *reinterpret_cast<f16x4*>(k16) = f16x4{(_Float16)k4[0], (_Float16)k4[1], (_Float16)k4[2], (_Float16)k4[3]};
            *reinterpret_cast<f16x4*>(k16 + d) = f16x4{(_Float16)v4[0], (_Float16)v4[1], (_Float16)v4[2], (_Float16)v4[3]};
          }
        }
    }
  }
}

static size_t af_lds_bytes(int L, int images_rowmajor, int images_transposed) {
  const int Lp = (L + 31) & ~31;
  return ((size_t)images_rowmajor * Lp * AF_ROW + (size_t)images_transposed * 64 * (Lp + 4)) * sizeof(_Float16);
}

}  // namespace clipfs

using namespace clipfs;

static int check_af(const void* a, const void* b, int batch, int seq, int heads) {
  CLIPFS_REQUIRE(a && b, "attention_f16: null pointer");
  CLIPFS_REQUIRE(batch > 0 && heads > 0 && seq > 0 && seq <= AF_MAXL, "attention_f16: seq %d outside 1..%d", seq, AF_MAXL);
  return CLIPFS_OK;
}

static int af_threads(int seq, int max_waves) {
  const int tiles = (seq + 31) / 32;
  return 64 * (tiles < max_waves ? tiles : max_waves);
}

template <typename TQ>
static void af_set_attrs() {
  static bool done = false;
  if (done) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f16_fwd_kernel<TQ>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done = true;
}

template <typename TQ>
static int af_fwd(const void* qkv, float* out, void* out16, float* lse, int batch, int seq, int heads, int causal,
                  hipStream_t st) {
  af_set_attrs<TQ>();
  hipLaunchKernelGGL(attention_f16_fwd_kernel<TQ>, dim3(batch * heads), dim3(af_threads(seq, 5)), af_lds_bytes(seq, 1, 1), st,
                     reinterpret_cast<const TQ*>(qkv), out, lse, seq, heads, causal, reinterpret_cast<_Float16*>(out16));
  return launch_status();
}

template <typename TQ, typename TG>
static int af_bwd(const void* qkv, const void* dout, const float* out, const float* lse, float* dqkv, void* dqkv16,
                  float* work, int batch, int seq, int heads, int causal, hipStream_t st) {
  const int threads = af_threads(seq, 9);
  const size_t lds_q = af_lds_bytes(seq, 2, 1);
  const size_t lds_kv = af_lds_bytes(seq, 2, 2) + 2 * (size_t)((seq + 31) & ~31) * sizeof(float);
  static bool attr = false;  // per instantiation
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f16_bwd_q_kernel<TQ, TG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f16_bwd_kv_kernel<TQ, TG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((attention_f16_bwd_q_kernel<TQ, TG>), dim3(batch * heads), dim3(threads), lds_q, st,
                     reinterpret_cast<const TQ*>(qkv), reinterpret_cast<const TG*>(dout), out, lse, dqkv, work, seq, heads, causal,
                     reinterpret_cast<_Float16*>(dqkv16));
  CLIPFS_CHECK(launch_status());
  hipLaunchKernelGGL((attention_f16_bwd_kv_kernel<TQ, TG>), dim3(batch * heads), dim3(threads), lds_kv, st,
                     reinterpret_cast<const TQ*>(qkv), reinterpret_cast<const TG*>(dout), lse, work, dqkv, seq, heads, causal,
                     reinterpret_cast<_Float16*>(dqkv16));
  return launch_status();
}

extern "C" int clipfs_attention_f16_fwd(const void* qkv, int qkv_f16, float* out, void* out16, float* lse, int batch, int seq,
                                        int heads, int causal, void* stream) {
  CLIPFS_CHECK(check_af(qkv, out, batch, seq, heads));
  CLIPFS_REQUIRE(aligned16(qkv) && aligned16(out), "attention_f16_fwd: misaligned pointer");
  return qkv_f16 ? af_fwd<_Float16>(qkv, out, out16, lse, batch, seq, heads, causal, (hipStream_t)stream)
                 : af_fwd<float>(qkv, out, out16, lse, batch, seq, heads, causal, (hipStream_t)stream);
}

extern "C" int clipfs_attention_f16_bwd(const void* qkv, int qkv_f16, const void* dout, int dout_f16, const float* out,
                                        const float* lse, float* dqkv, void* dqkv16, float* work, int batch, int seq, int heads,
                                        int causal, void* stream) {
  CLIPFS_CHECK(check_af(qkv, dqkv ? (const void*)dqkv : dqkv16, batch, seq, heads));  // the fp32 result is optional beside the f16 one
  CLIPFS_REQUIRE(dout && out && lse && work, "attention_f16_bwd: null pointer");
  CLIPFS_REQUIRE(aligned16(qkv) && aligned16(dout) && aligned16(out) && aligned16(dqkv) && aligned16(dqkv16),
                 "attention_f16_bwd: misaligned pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dout_f16)
    return qkv_f16 ? af_bwd<_Float16, _Float16>(qkv, dout, out, lse, dqkv, dqkv16, work, batch, seq, heads, causal, st)
                   : af_bwd<float, _Float16>(qkv, dout, out, lse, dqkv, dqkv16, work, batch, seq, heads, causal, st);
  return qkv_f16 ? af_bwd<_Float16, float>(qkv, dout, out, lse, dqkv, dqkv16, work, batch, seq, heads, causal, st)
                 : af_bwd<float, float>(qkv, dout, out, lse, dqkv, dqkv16, work, batch, seq, heads, causal, st);
}
