// Self-attention forward / backward for head_dim 64 and short sequences (CLIP ViT-B/32: L = 50, 54 with
// VPT; text: L = 77), one workgroup (4 waves) per (batch, head).
//
// ~1 % of the layer's FLOPs, and the fp32 matrix rate equals the fp32 vector rate on gfx950, so this runs on
// the VALU -- what matters is operand delivery.  v1 read one LDS word per FMA (LDS-bound, 20 % VALU
// utilisation); here every inner loop is  FMA(register operand, broadcast operand):
//   * "row" operands (K rows, V rows: lane = key j) and "column" operands (V / K / Q / dO columns:
//     lane = feature d) live in VGPRs, loaded once per wave;
//   * the broadcast operand is either a wave-uniform global row (q_i, dO_i: scalar/broadcast loads) or a
//     probability / score-gradient vector that the wave parked in LDS and reads back 4 at a time with one
//     same-address ds_read_b128 (1 LDS instruction per 4 FMAs);
//   * softmax max / sum are DPP wave reductions (common.h).
// The [L, L] matrices of the reference (jclip/mha.py:79-83: a 30-77 MB HBM round trip per layer) stay on
// chip: registers in the forward, LDS (P^T, dS^T, dS) in the backward.  Heads are merged by the store
// address (no permute pass, mha.py:458).
#include "common.h"

#include <stdlib.h>

namespace clipfs {

constexpr int HD = 64;
constexpr int KSTRIDE = 68;  // staged K/V row stride (floats): 16-byte aligned, conflict-free ds_read_b128 by row

typedef float f32x4 __attribute__((ext_vector_type(4)));

// cooperative copy of L rows of 64 floats (global row stride ld) into LDS rows of KSTRIDE floats
__device__ __forceinline__ void stage_rows(const float* __restrict__ base, size_t ld, float* __restrict__ dst, int L,
                                           int tid) {
  for (int idx = tid; idx < L * (HD / 4); idx += (int)blockDim.x) {
    const int r = idx >> 4, c = idx & 15;
    *reinterpret_cast<f32x4*>(dst + r * KSTRIDE + 4 * c) = *reinterpret_cast<const f32x4*>(base + (size_t)r * ld + 4 * c);
  }
}

// lane j takes row j (+64 per kk) of a staged [L][KSTRIDE] image into registers; rows >= L are zero
template <int KPL>
__device__ __forceinline__ void rows_to_regs(const float* __restrict__ s, int lane, int L, float (&reg)[KPL][HD]) {
#pragma unroll
  for (int kk = 0; kk < KPL; ++kk) {
    const int j = lane + 64 * kk;
    const bool ok = j < L;
    const float* row = s + (ok ? j : 0) * KSTRIDE;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * c);
      reg[kk][4 * c + 0] = ok ? v[0] : 0.f;
      reg[kk][4 * c + 1] = ok ? v[1] : 0.f;
      reg[kk][4 * c + 2] = ok ? v[2] : 0.f;
      reg[kk][4 * c + 3] = ok ? v[3] : 0.f;
    }
  }
}

// lane d takes column d of L global rows (coalesced 256-byte row reads); entries >= L are zero
template <int LMAX>
__device__ __forceinline__ void col_to_regs(const float* __restrict__ base, size_t ld, int lane, int L,
                                            float (&reg)[LMAX]) {
  // unconditional (row-clamped) loads so all of them are in flight together; zero-select afterwards
#pragma unroll
  for (int j = 0; j < LMAX; ++j) reg[j] = base[(size_t)min(j, L - 1) * ld + lane];
#pragma unroll
  for (int j = 0; j < LMAX; ++j) reg[j] = j < L ? reg[j] : 0.f;
}

// dot of the wave-uniform global row `u` (64 floats) with each of this lane's register rows
// (`nkeys` = number of keys that are not masked for this row: key blocks of 64 beyond it are skipped)
template <int KPL>
__device__ __forceinline__ void dot_rows(const float* __restrict__ u, const float (&reg)[KPL][HD], float (&out)[KPL],
                                         int nkeys = 64 * KPL) {
  f32x4 q[HD / 4];
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) q[c] = *reinterpret_cast<const f32x4*>(u + 4 * c);  // same address in every lane
#pragma unroll
  for (int kk = 0; kk < KPL; ++kk) {
    float a0 = 0.f, a1 = 0.f;
    if (64 * kk < nkeys) {  // wave-uniform
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) {
        a0 = fmaf(q[c][0], reg[kk][4 * c + 0], a0);
        a1 = fmaf(q[c][1], reg[kk][4 * c + 1], a1);
        a0 = fmaf(q[c][2], reg[kk][4 * c + 2], a0);
        a1 = fmaf(q[c][3], reg[kk][4 * c + 3], a1);
      }
    }
    out[kk] = a0 + a1;
  }
}

// scores -> probabilities for query row i (lane = key): scale after the dot (mha.py:79), mask, softmax
template <int KPL>
__device__ __forceinline__ void softmax_row(float (&s)[KPL], int i, int lane, int L, int causal) {
  float m = -INFINITY;
#pragma unroll
  for (int kk = 0; kk < KPL; ++kk) {
    const int j = lane + 64 * kk;
    s[kk] *= 0.125f;
    if (j >= L || (causal && j > i)) s[kk] = -INFINITY;
    m = fmaxf(m, s[kk]);
  }
  m = wave_max(m);
  float sum = 0.f;
#pragma unroll
  for (int kk = 0; kk < KPL; ++kk) {
    s[kk] = __expf(s[kk] - m);  // exp(-inf) = 0 for masked keys
    sum += s[kk];
  }
  sum = wave_sum(sum);
#pragma unroll
  for (int kk = 0; kk < KPL; ++kk) s[kk] = s[kk] / sum;
}

// sum_j vec[j] * col[j] for j < n: vec = 16-byte aligned LDS vector read as same-address b128 broadcasts.
// Guarded in groups of 16 (4 reads in flight per group); vec must be finite up to the group boundary and
// col[j >= L] == 0, so the overhang contributes exact zeros.
template <int LMAX>
__device__ __forceinline__ float bcast_dot(const float* __restrict__ vec, const float (&col)[LMAX], int n, int lo = 0) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
  for (int g = 0; g < LMAX / 16; ++g) {
    if (16 * g < n && 16 * g + 16 > lo) {  // entries below lo are exact zeros (causal mask) in vec
      f32x4 p[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) p[t] = *reinterpret_cast<const f32x4*>(vec + 16 * g + 4 * t);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a0 = fmaf(p[t][0], col[16 * g + 4 * t + 0], a0);
        a1 = fmaf(p[t][1], col[16 * g + 4 * t + 1], a1);
        a2 = fmaf(p[t][2], col[16 * g + 4 * t + 2], a2);
        a3 = fmaf(p[t][3], col[16 * g + 4 * t + 3], a3);
      }
    }
  }
  return (a0 + a1) + (a2 + a3);
}

// two query rows (uniform global rows u0, u1) against this lane's K rows read from the staged LDS image:
// each ds_read_b128 of K feeds 8 FMAs (used when the K rows do not fit in VGPRs next to the V column)
template <int KPL>
__device__ __forceinline__ void dot_rows_lds2(const float* __restrict__ u0, const float* __restrict__ u1,
                                              const float* __restrict__ sK, int lane, int L, float (&o0)[KPL],
                                              float (&o1)[KPL]) {
  const float* row[KPL];
#pragma unroll
  for (int kk = 0; kk < KPL; ++kk) {
    const int j = lane + 64 * kk;
    row[kk] = sK + (j < L ? j : 0) * KSTRIDE;  // rows >= L are masked to -inf afterwards
    o0[kk] = o1[kk] = 0.f;
  }
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    const f32x4 qa = *reinterpret_cast<const f32x4*>(u0 + 4 * c);
    const f32x4 qb = *reinterpret_cast<const f32x4*>(u1 + 4 * c);
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
      const f32x4 k = *reinterpret_cast<const f32x4*>(row[kk] + 4 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o0[kk] = fmaf(qa[e], k[e], o0[kk]);
        o1[kk] = fmaf(qb[e], k[e], o1[kk]);
      }
    }
  }
}

template <int LMAX>
__global__ __launch_bounds__(512) void attention_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            int L, int H, int causal) {
  constexpr int KPL = (LMAX + 63) / 64;
  constexpr int PROW = 64 * KPL;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sK = smem;                 // [L][KSTRIDE] staged K rows
  float* sP = smem + L * KSTRIDE;   // [16][PROW] probability rows of the waves' current queries
  const int NW = (int)blockDim.x >> 6;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * HD;
  const size_t ld = (size_t)3 * d;
  const float* q0 = qkv + (size_t)b * L * ld + (size_t)h * HD;
  float* o0 = out + (size_t)b * L * d + (size_t)h * HD;
  stage_rows(q0 + d, ld, sK, L, tid);
  float vcol[LMAX];
  col_to_regs<LMAX>(q0 + 2 * d, ld, lane, L, vcol);  // V[:, lane]
  __syncthreads();
  if constexpr (KPL == 1) {
    float krow[KPL][HD];
    rows_to_regs<KPL>(sK, lane, L, krow);
    float* myP = sP + wave * PROW;
    for (int i = wave; i < L; i += NW) {
      float s[KPL];
      dot_rows<KPL>(q0 + (size_t)i * ld, krow, s, causal ? i + 1 : L);
      softmax_row<KPL>(s, i, lane, L, causal);
      myP[lane] = s[0];
      __builtin_amdgcn_wave_barrier();  // same wave writes then reads: LDS is in order per wave
      const float o = bcast_dot<LMAX>(myP, vcol, causal ? i + 1 : L);
      __builtin_amdgcn_wave_barrier();
      o0[(size_t)i * d + lane] = o;
    }
  } else {
    float* pa = sP + (2 * wave) * PROW;
    float* pb = pa + PROW;
    for (int i = wave; i < L; i += 2 * NW) {
      const int i2 = i + NW;
      const bool two = i2 < L;
      float sa[KPL], sb[KPL];
      dot_rows_lds2<KPL>(q0 + (size_t)i * ld, q0 + (size_t)(two ? i2 : i) * ld, sK, lane, L, sa, sb);
      softmax_row<KPL>(sa, i, lane, L, causal);
      softmax_row<KPL>(sb, two ? i2 : i, lane, L, causal);
#pragma unroll
      for (int kk = 0; kk < KPL; ++kk) {
        pa[lane + 64 * kk] = sa[kk];
        pb[lane + 64 * kk] = sb[kk];
      }
      __builtin_amdgcn_wave_barrier();
      const float oa = bcast_dot<LMAX>(pa, vcol, causal ? i + 1 : L);
      const float ob = bcast_dot<LMAX>(pb, vcol, causal ? (two ? i2 : i) + 1 : L);
      __builtin_amdgcn_wave_barrier();
      o0[(size_t)i * d + lane] = oa;
      if (two) o0[(size_t)i2 * d + lane] = ob;
    }
  }
}

// Backward: dQ = scale dS K, dK = scale dS^T Q, dV = P^T dO, dS = P * (dP - rowsum(P * dP)), dP = dO V^T.
// P is recomputed (phase A1), never stored in HBM.  Phases (register sets are reused between phases):
//   A1  lane = key, K rows in VGPRs : P_i  -> LDS P^T[j][i]
//   A2  lane = key, V rows in VGPRs : dP_i, dS_i -> LDS dS^T[j][i] and dS[i][j]   (scale folded into dS)
//   C1  lane = d, dO and Q columns in VGPRs, per key j : dV[j], dK[j]  (P^T[j][:], dS^T[j][:] broadcast)
//   C2  lane = d, K column in VGPRs, per query i      : dQ[i]         (dS[i][:] broadcast)
template <int LMAX>
__global__ __launch_bounds__(512) void attention_bwd_kernel(const float* __restrict__ qkv,
                                                            const float* __restrict__ dout,
                                                            float* __restrict__ dqkv, int L, int H, int causal, int LP) {
  constexpr int KPL = (LMAX + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sPT = smem;              // [L][LP]  P^T
  float* sdST = sPT + L * LP;     // [L][LP]  dS^T
  float* sdS = sdST + L * LP;     // [L][LP]  dS   (aliases the staging buffer below: LP >= KSTRIDE is not needed,
  float* sKV = sdS;               //  [L][KSTRIDE] staged K rows, then V rows -- dead before dS is written)
  const int NW = (int)blockDim.x >> 6;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * HD;
  const size_t ld = (size_t)3 * d;
  const float* q0 = qkv + (size_t)b * L * ld + (size_t)h * HD;
  const float* do0 = dout + (size_t)b * L * d + (size_t)h * HD;
  float* dq0 = dqkv + (size_t)b * L * ld + (size_t)h * HD;
  for (int idx = tid; idx < 2 * L * LP; idx += (int)blockDim.x) smem[idx] = 0.f;  // P^T, dS^T: finite zero padding
  stage_rows(q0 + d, ld, sKV, L, tid);
  __syncthreads();
  {
    float rows[KPL][HD];
    rows_to_regs<KPL>(sKV, lane, L, rows);  // K rows
    __syncthreads();
    stage_rows(q0 + 2 * d, ld, sKV, L, tid);  // V rows replace K rows in the staging buffer
    // A1: probabilities
    for (int i = wave; i < L; i += NW) {
      float s[KPL];
      dot_rows<KPL>(q0 + (size_t)i * ld, rows, s, causal ? i + 1 : L);
      softmax_row<KPL>(s, i, lane, L, causal);
#pragma unroll
      for (int kk = 0; kk < KPL; ++kk) {
        const int j = lane + 64 * kk;
        if (j < L) sPT[j * LP + i] = s[kk];
      }
    }
    __syncthreads();
    rows_to_regs<KPL>(sKV, lane, L, rows);  // V rows
    __syncthreads();                         // staging buffer is dead from here: it becomes dS
    for (int idx = tid; idx < L * LP; idx += (int)blockDim.x) sdS[idx] = 0.f;
    __syncthreads();
    // A2: dP, dS
    for (int i = wave; i < L; i += NW) {
      float dp[KPL], pv[KPL];
      dot_rows<KPL>(do0 + (size_t)i * d, rows, dp, causal ? i + 1 : L);
      float rs = 0.f;
#pragma unroll
      for (int kk = 0; kk < KPL; ++kk) {
        const int j = lane + 64 * kk;
        pv[kk] = j < L ? sPT[j * LP + i] : 0.f;
        rs = fmaf(pv[kk], dp[kk], rs);
      }
      rs = wave_sum(rs);
#pragma unroll
      for (int kk = 0; kk < KPL; ++kk) {
        const int j = lane + 64 * kk;
        const float ds = pv[kk] * (dp[kk] - rs) * 0.125f;
        if (j < L) {
          sdST[j * LP + i] = ds;
          sdS[i * LP + j] = ds;
        }
      }
    }
  }
  __syncthreads();
  {
    // C1: dV[j] = sum_i P_ij dO_i , dK[j] = sum_i dS_ij Q_i
    float doc[LMAX], qc[LMAX];
    col_to_regs<LMAX>(do0, (size_t)d, lane, L, doc);
    col_to_regs<LMAX>(q0, ld, lane, L, qc);
    for (int j = wave; j < L; j += NW) {
      const int lo = causal ? j : 0;  // P_ij = dS_ij = 0 for i < j
      const float av = bcast_dot<LMAX>(sPT + j * LP, doc, L, lo);
      const float ak = bcast_dot<LMAX>(sdST + j * LP, qc, L, lo);
      dq0[(size_t)j * ld + 2 * d + lane] = av;
      dq0[(size_t)j * ld + d + lane] = ak;
    }
  }
  {
    // C2: dQ[i] = sum_j dS_ij K_j
    float kc[LMAX];
    col_to_regs<LMAX>(q0 + d, ld, lane, L, kc);
    for (int i = wave; i < L; i += NW) dq0[(size_t)i * ld + lane] = bcast_dot<LMAX>(sdS + i * LP, kc, causal ? i + 1 : L);
  }
}


// ---------------------------------------------------------------------------------------------------------
// Long sequences (ViT-L/14: L = 257): K/V no longer fit next to a wave's registers, so these kernels stream key
// (or query) blocks of 64 from L2 with an online softmax -- one wave per query row (forward, dQ) or per key row
// (dK, dV).  Correct for any L; not tuned (the B/32 path never takes them).  lse[(b*H + h)*L + i] =
// max_i + log(sum_i) of the scaled scores is written by the forward and reused by the backward.
// ---------------------------------------------------------------------------------------------------------

__device__ __forceinline__ float dot64_row(const f32x4 (&u)[HD / 4], const float* __restrict__ row) {
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    const f32x4 k = *reinterpret_cast<const f32x4*>(row + 4 * c);
    a0 = fmaf(u[c][0], k[0], a0);
    a1 = fmaf(u[c][1], k[1], a1);
    a0 = fmaf(u[c][2], k[2], a0);
    a1 = fmaf(u[c][3], k[3], a1);
  }
  return a0 + a1;
}

// acc[lane] += sum_{t < n} vec[t] * base[(j0 + t) * ld + lane]   (vec: the wave's LDS row of 64 broadcast values)
__device__ __forceinline__ float axpy_block(const float* __restrict__ vec, const float* __restrict__ base, size_t ld,
                                            int lane, int n, float acc) {
  for (int t = 0; t < n; t += 4) {
    const f32x4 p = *reinterpret_cast<const f32x4*>(vec + t);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (t + e < n) acc = fmaf(p[e], base[(size_t)(t + e) * ld + lane], acc);
  }
  return acc;
}

__global__ __launch_bounds__(256) void attention_generic_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                    float* __restrict__ lse, int L, int H, int causal) {
  __shared__ __attribute__((aligned(16))) float sP[4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int i = blockIdx.y * 4 + wave;
  if (i >= L) return;
  const int d = H * HD;
  const size_t ld = (size_t)3 * d;
  const float* q0 = qkv + (size_t)b * L * ld + (size_t)h * HD;
  f32x4 q[HD / 4];
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) q[c] = *reinterpret_cast<const f32x4*>(q0 + (size_t)i * ld + 4 * c);
  float m = -INFINITY, l = 0.f, o = 0.f;
  const int jend = causal ? i + 1 : L;
  for (int j0 = 0; j0 < jend; j0 += 64) {
    const int j = j0 + lane;
    float s = -INFINITY;
    if (j < jend) s = dot64_row(q, q0 + d + (size_t)j * ld) * 0.125f;
    const float mn = fmaxf(m, wave_max(s));
    const float pj = __expf(s - mn);  // 0 for masked lanes
    const float f = __expf(m - mn);   // 0 on the first block (m = -inf)
    l = l * f + wave_sum(pj);
    sP[wave][lane] = pj;
    __builtin_amdgcn_wave_barrier();
    o = axpy_block(sP[wave], q0 + 2 * d + (size_t)j0 * ld, ld, lane, min(64, jend - j0), o * f);
    __builtin_amdgcn_wave_barrier();
    m = mn;
  }
  out[((size_t)b * L + i) * d + h * HD + lane] = o / l;
  if (lse && lane == 0) lse[((size_t)b * H + h) * L + i] = m + __logf(l);
}

// dQ and D_i = dO_i . O_i   (one wave per query row)
__global__ __launch_bounds__(256) void attention_generic_bwd_q_kernel(const float* __restrict__ qkv,
                                                                      const float* __restrict__ dout,
                                                                      const float* __restrict__ out,
                                                                      const float* __restrict__ lse,
                                                                      float* __restrict__ dqkv, float* __restrict__ Dbuf,
                                                                      int L, int H, int causal) {
  __shared__ __attribute__((aligned(16))) float sP[4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int i = blockIdx.y * 4 + wave;
  if (i >= L) return;
  const int d = H * HD;
  const size_t ld = (size_t)3 * d;
  const float* q0 = qkv + (size_t)b * L * ld + (size_t)h * HD;
  const float* do_i = dout + ((size_t)b * L + i) * d + h * HD;
  f32x4 q[HD / 4], g[HD / 4];
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    q[c] = *reinterpret_cast<const f32x4*>(q0 + (size_t)i * ld + 4 * c);
    g[c] = *reinterpret_cast<const f32x4*>(do_i + 4 * c);
  }
  const float Di = wave_sum(do_i[lane] * out[((size_t)b * L + i) * d + h * HD + lane]);
  const float li = lse[((size_t)b * H + h) * L + i];
  if (lane == 0) Dbuf[((size_t)b * H + h) * L + i] = Di;
  float acc = 0.f;
  const int jend = causal ? i + 1 : L;
  for (int j0 = 0; j0 < jend; j0 += 64) {
    const int j = j0 + lane;
    float ds = 0.f;
    if (j < jend) {
      const float pj = __expf(dot64_row(q, q0 + d + (size_t)j * ld) * 0.125f - li);
      const float dp = dot64_row(g, q0 + 2 * d + (size_t)j * ld);
      ds = pj * (dp - Di) * 0.125f;
    }
    sP[wave][lane] = ds;
    __builtin_amdgcn_wave_barrier();
    acc = axpy_block(sP[wave], q0 + d + (size_t)j0 * ld, ld, lane, min(64, jend - j0), acc);
    __builtin_amdgcn_wave_barrier();
  }
  dqkv[((size_t)b * L + i) * ld + h * HD + lane] = acc;
}

// dK and dV   (one wave per key row; queries streamed in blocks of 64, lane = query)
__global__ __launch_bounds__(256) void attention_generic_bwd_kv_kernel(const float* __restrict__ qkv,
                                                                       const float* __restrict__ dout,
                                                                       const float* __restrict__ lse,
                                                                       const float* __restrict__ Dbuf,
                                                                       float* __restrict__ dqkv, int L, int H, int causal) {
  __shared__ __attribute__((aligned(16))) float sP[4][64];
  __shared__ __attribute__((aligned(16))) float sS[4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int j = blockIdx.y * 4 + wave;
  if (j >= L) return;
  const int d = H * HD;
  const size_t ld = (size_t)3 * d;
  const float* q0 = qkv + (size_t)b * L * ld + (size_t)h * HD;
  const float* do0 = dout + (size_t)b * L * d + (size_t)h * HD;
  f32x4 k[HD / 4], v[HD / 4];
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    k[c] = *reinterpret_cast<const f32x4*>(q0 + d + (size_t)j * ld + 4 * c);
    v[c] = *reinterpret_cast<const f32x4*>(q0 + 2 * d + (size_t)j * ld + 4 * c);
  }
  float av = 0.f, ak = 0.f;
  const int ibeg = causal ? j : 0;  // P_ij = 0 for i < j
  for (int i0 = ibeg & ~63; i0 < L; i0 += 64) {
    const int i = i0 + lane;
    float pj = 0.f, ds = 0.f;
    if (i < L && i >= ibeg) {
      const size_t st = ((size_t)b * H + h) * L + i;
      pj = __expf(dot64_row(k, q0 + (size_t)i * ld) * 0.125f - lse[st]);
      const float dp = dot64_row(v, do0 + (size_t)i * d);
      ds = pj * (dp - Dbuf[st]) * 0.125f;
    }
    sP[wave][lane] = pj;
    sS[wave][lane] = ds;
    __builtin_amdgcn_wave_barrier();
    const int n = min(64, L - i0);
    av = axpy_block(sP[wave], do0 + (size_t)i0 * d, (size_t)d, lane, n, av);
    ak = axpy_block(sS[wave], q0 + (size_t)i0 * ld, ld, lane, n, ak);
    __builtin_amdgcn_wave_barrier();
  }
  dqkv[((size_t)b * L + j) * ld + 2 * d + h * HD + lane] = av;
  dqkv[((size_t)b * L + j) * ld + d + h * HD + lane] = ak;
}


// ---------------------------------------------------------------------------------------------------------
// Medium sequences (96 < L <= 300; ViT-L/14: 257): one workgroup of 8 waves per (batch, head) with the head's K and V
// (or Q and dO) resident in LDS -- 2 x L x 68 floats = 140 KB at L = 257 -- instead of being re-streamed from L2 per
// query row.  A wave owns one query (key) row at a time: its row is wave-uniform (scalar loads), the LDS rows are
// read by the lane that owns that key (16 conflict-free ds_read_b128 at row stride 68) and column-wise
// (ds_read_b32, lane = feature) for the accumulations; online softmax over key blocks of 64.
// ---------------------------------------------------------------------------------------------------------

// acc[lane] += sum_{t < n} vec[t] * rows[(j0 + t) * KSTRIDE + lane]     (rows: LDS image, vec: the wave's LDS slot)
__device__ __forceinline__ float axpy_block_lds(const float* __restrict__ vec, const float* __restrict__ rows, int lane,
                                                int n, float acc) {
  float a0 = acc, a1 = 0.f;
  int t = 0;
  for (; t + 4 <= n; t += 4) {
    const f32x4 p = *reinterpret_cast<const f32x4*>(vec + t);
    a0 = fmaf(p[0], rows[(t + 0) * KSTRIDE + lane], a0);
    a1 = fmaf(p[1], rows[(t + 1) * KSTRIDE + lane], a1);
    a0 = fmaf(p[2], rows[(t + 2) * KSTRIDE + lane], a0);
    a1 = fmaf(p[3], rows[(t + 3) * KSTRIDE + lane], a1);
  }
  for (; t < n; ++t) a0 = fmaf(vec[t], rows[t * KSTRIDE + lane], a0);
  return a0 + a1;
}

__device__ __forceinline__ void stage_rows_n(const float* __restrict__ base, size_t ld, float* __restrict__ dst, int L) {
  for (int idx = threadIdx.x; idx < L * (HD / 4); idx += (int)blockDim.x) {
    const int r = idx >> 4, c = idx & 15;
    *reinterpret_cast<f32x4*>(dst + r * KSTRIDE + 4 * c) = *reinterpret_cast<const f32x4*>(base + (size_t)r * ld + 4 * c);
  }
}

__global__ __launch_bounds__(512) void attention_lds_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                float* __restrict__ lse, int L, int H, int causal) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sK = smem;
  float* sV = sK + L * KSTRIDE;
  float* sP = sV + L * KSTRIDE;  // [8][64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * HD;
  const size_t ld = (size_t)3 * d;
  const float* q0 = qkv + (size_t)b * L * ld + (size_t)h * HD;
  stage_rows_n(q0 + d, ld, sK, L);
  stage_rows_n(q0 + 2 * d, ld, sV, L);
  __syncthreads();
  float* myP = sP + wave * 64;
  for (int i = wave; i < L; i += 8) {
    f32x4 q[HD / 4];
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) q[c] = *reinterpret_cast<const f32x4*>(q0 + (size_t)i * ld + 4 * c);  // uniform
    float m = -INFINITY, l = 0.f, o = 0.f;
    const int jend = causal ? i + 1 : L;
    for (int j0 = 0; j0 < jend; j0 += 64) {
      const int j = j0 + lane;
      float sc = -INFINITY;
      if (j < jend) sc = dot64_row(q, sK + j * KSTRIDE) * 0.125f;
      const float mn = fmaxf(m, wave_max(sc));
      const float pj = __expf(sc - mn);
      const float f = __expf(m - mn);
      l = l * f + wave_sum(pj);
      myP[lane] = pj;
      __builtin_amdgcn_wave_barrier();
      o = axpy_block_lds(myP, sV + j0 * KSTRIDE, lane, min(64, jend - j0), o * f);
      __builtin_amdgcn_wave_barrier();
      m = mn;
    }
    out[((size_t)b * L + i) * d + h * HD + lane] = o / l;
    if (lse && lane == 0) lse[((size_t)b * H + h) * L + i] = m + __logf(l);
  }
}

__global__ __launch_bounds__(512) void attention_lds_bwd_q_kernel(const float* __restrict__ qkv,
                                                                  const float* __restrict__ dout,
                                                                  const float* __restrict__ out,
                                                                  const float* __restrict__ lse, float* __restrict__ dqkv,
                                                                  float* __restrict__ Dbuf, int L, int H, int causal) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sK = smem;
  float* sV = sK + L * KSTRIDE;
  float* sP = sV + L * KSTRIDE;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * HD;
  const size_t ld = (size_t)3 * d;
  const float* q0 = qkv + (size_t)b * L * ld + (size_t)h * HD;
  stage_rows_n(q0 + d, ld, sK, L);
  stage_rows_n(q0 + 2 * d, ld, sV, L);
  __syncthreads();
  float* myP = sP + wave * 64;
  for (int i = wave; i < L; i += 8) {
    const float* do_i = dout + ((size_t)b * L + i) * d + h * HD;
    f32x4 q[HD / 4], g[HD / 4];
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      q[c] = *reinterpret_cast<const f32x4*>(q0 + (size_t)i * ld + 4 * c);
      g[c] = *reinterpret_cast<const f32x4*>(do_i + 4 * c);
    }
    const float Di = wave_sum(do_i[lane] * out[((size_t)b * L + i) * d + h * HD + lane]);
    const float li = lse[((size_t)b * H + h) * L + i];
    if (lane == 0) Dbuf[((size_t)b * H + h) * L + i] = Di;
    float acc = 0.f;
    const int jend = causal ? i + 1 : L;
    for (int j0 = 0; j0 < jend; j0 += 64) {
      const int j = j0 + lane;
      float ds = 0.f;
      if (j < jend) {
        const float pj = __expf(dot64_row(q, sK + j * KSTRIDE) * 0.125f - li);
        ds = pj * (dot64_row(g, sV + j * KSTRIDE) - Di) * 0.125f;
      }
      myP[lane] = ds;
      __builtin_amdgcn_wave_barrier();
      acc = axpy_block_lds(myP, sK + j0 * KSTRIDE, lane, min(64, jend - j0), acc);
      __builtin_amdgcn_wave_barrier();
    }
    dqkv[((size_t)b * L + i) * ld + h * HD + lane] = acc;
  }
}

__global__ __launch_bounds__(512) void attention_lds_bwd_kv_kernel(const float* __restrict__ qkv,
                                                                   const float* __restrict__ dout,
                                                                   const float* __restrict__ lse,
                                                                   const float* __restrict__ Dbuf,
                                                                   float* __restrict__ dqkv, int L, int H, int causal) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sQ = smem;
  float* sG = sQ + L * KSTRIDE;  // dO rows
  float* sP = sG + L * KSTRIDE;  // [8][64] p, then [8][64] dS
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * HD;
  const size_t ld = (size_t)3 * d;
  const float* q0 = qkv + (size_t)b * L * ld + (size_t)h * HD;
  stage_rows_n(q0, ld, sQ, L);
  stage_rows_n(dout + (size_t)b * L * d + (size_t)h * HD, (size_t)d, sG, L);
  __syncthreads();
  float* myP = sP + wave * 64;
  float* myS = sP + 8 * 64 + wave * 64;
  const float* lse0 = lse + ((size_t)b * H + h) * L;
  const float* D0 = Dbuf + ((size_t)b * H + h) * L;
  for (int j = wave; j < L; j += 8) {
    f32x4 k[HD / 4], v[HD / 4];
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      k[c] = *reinterpret_cast<const f32x4*>(q0 + d + (size_t)j * ld + 4 * c);
      v[c] = *reinterpret_cast<const f32x4*>(q0 + 2 * d + (size_t)j * ld + 4 * c);
    }
    float av = 0.f, ak = 0.f;
    const int ibeg = causal ? j : 0;
    for (int i0 = ibeg & ~63; i0 < L; i0 += 64) {
      const int i = i0 + lane;
      float pj = 0.f, ds = 0.f;
      if (i < L && i >= ibeg) {
        pj = __expf(dot64_row(k, sQ + i * KSTRIDE) * 0.125f - lse0[i]);
        ds = pj * (dot64_row(v, sG + i * KSTRIDE) - D0[i]) * 0.125f;
      }
      myP[lane] = pj;
      myS[lane] = ds;
      __builtin_amdgcn_wave_barrier();
      const int n = min(64, L - i0);
      av = axpy_block_lds(myP, sG + i0 * KSTRIDE, lane, n, av);
      ak = axpy_block_lds(myS, sQ + i0 * KSTRIDE, lane, n, ak);
      __builtin_amdgcn_wave_barrier();
    }
    dqkv[((size_t)b * L + j) * ld + 2 * d + h * HD + lane] = av;
    dqkv[((size_t)b * L + j) * ld + d + h * HD + lane] = ak;
  }
}

// attention_mfma.hip: exact-fp32 MFMA kernels for seq <= 288 (the default path; the VALU kernels below remain for
// longer sequences, for backward calls without the forward's lse, and as the CLIPFS_ATTN_MFMA=0 comparison)
bool attention_mfma_enabled();
int attention_mfma_fwd(const float* qkv, float* out, float* lse, int batch, int seq, int heads, int causal, hipStream_t st);
int attention_mfma_bwd(const float* qkv, const float* dout, const float* out, const float* lse, float* dqkv, float* work,
                       int batch, int seq, int heads, int causal, hipStream_t st);
constexpr int ATTN_MFMA_MAX = 288;

static int check_attn(const char* what, int batch, int seq, int heads, int max_seq) {
  CLIPFS_REQUIRE(batch > 0 && heads > 0 && seq > 0 && seq <= max_seq, "%s: batch %d seq %d heads %d unsupported (seq <= %d)",
                 what, batch, seq, heads, max_seq);
  return CLIPFS_OK;
}

// padded row length of the LDS [L][LP] matrices: >= L rounded up to 4 (b128 broadcast reads), LP / 4 odd so the
// transposed (stride-LP) writes spread over 8 distinct bank groups
static int padded_lp(int seq) {
  int lp = (seq + 3) & ~3;
  if (((lp >> 2) & 1) == 0) lp += 4;
  return lp;
}

}  // namespace clipfs

using namespace clipfs;

constexpr int ATTN_FAST_MAX = 96;   // register/LDS-resident kernels up to here (forward-only inference: 128)
constexpr int ATTN_MAX_SEQ = 4096;
constexpr int ATTN_LDS_MAX = 288;    // (2 * L * 68 + 16 * 64) floats <= 160 KiB: K and V (or Q and dO) of a head stay in LDS

extern "C" int clipfs_attention_fwd(const float* qkv, float* out, float* lse, int batch, int seq, int heads, int causal,
                                    void* stream) {
  CLIPFS_CHECK(check_attn("attention_fwd", batch, seq, heads, ATTN_MAX_SEQ));
  CLIPFS_REQUIRE(qkv && out && aligned16(qkv), "attention_fwd: null or misaligned pointer");
  hipStream_t st = (hipStream_t)stream;
  if (attention_mfma_enabled() && seq <= ATTN_MFMA_MAX && aligned16(out))
    return attention_mfma_fwd(qkv, out, lse, batch, seq, heads, causal, st);
  if ((seq > 128 || (seq > ATTN_FAST_MAX && lse)) && seq <= ATTN_LDS_MAX) {
    const size_t lds = ((size_t)2 * seq * KSTRIDE + 8 * 64) * sizeof(float);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_lds_fwd_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
    hipLaunchKernelGGL(attention_lds_fwd_kernel, dim3(batch * heads), dim3(512), lds, st, qkv, out, lse, seq, heads, causal);
    return launch_status();
  }
  if (seq > 128 || (seq > ATTN_FAST_MAX && lse)) {
    hipLaunchKernelGGL(attention_generic_fwd_kernel, dim3(batch * heads, (seq + 3) / 4), dim3(256), 0, st, qkv, out, lse,
                       seq, heads, causal);
    return launch_status();
  }
  static const int thr_cfg = getenv("CLIPFS_ATTN_FWD_THREADS") ? atoi(getenv("CLIPFS_ATTN_FWD_THREADS")) : 0;
  const dim3 grid(batch * heads), block(thr_cfg ? thr_cfg : 256);
  const size_t lds1 = ((size_t)seq * KSTRIDE + 16 * 64) * sizeof(float);
  const size_t lds2 = ((size_t)seq * KSTRIDE + 16 * 128) * sizeof(float);
  if (seq <= 64)
    hipLaunchKernelGGL((attention_fwd_kernel<64>), grid, block, lds1, st, qkv, out, seq, heads, causal);
  else if (seq <= 80)
    hipLaunchKernelGGL((attention_fwd_kernel<80>), grid, block, lds2, st, qkv, out, seq, heads, causal);
  else if (seq <= 96)
    hipLaunchKernelGGL((attention_fwd_kernel<96>), grid, block, lds2, st, qkv, out, seq, heads, causal);
  else
    hipLaunchKernelGGL((attention_fwd_kernel<128>), grid, block, lds2, st, qkv, out, seq, heads, causal);
  return launch_status();
}

extern "C" int clipfs_attention_bwd(const float* qkv, const float* dout, const float* out, const float* lse, float* dqkv,
                                    float* work, int batch, int seq, int heads, int causal, void* stream) {
  CLIPFS_CHECK(check_attn("attention_bwd", batch, seq, heads, ATTN_MAX_SEQ));
  CLIPFS_REQUIRE(qkv && dout && dqkv && aligned16(qkv) && aligned16(dout), "attention_bwd: null or misaligned pointer");
  hipStream_t st = (hipStream_t)stream;
  if (attention_mfma_enabled() && seq <= ATTN_MFMA_MAX && out && lse && work && aligned16(out) && aligned16(dqkv))
    return attention_mfma_bwd(qkv, dout, out, lse, dqkv, work, batch, seq, heads, causal, st);
  if (seq > ATTN_FAST_MAX) {
    CLIPFS_REQUIRE(out && lse && work, "attention_bwd: seq %d > %d needs the forward's out and lse and a work buffer", seq,
                   ATTN_FAST_MAX);
    if (seq <= ATTN_LDS_MAX) {
      const size_t lds = ((size_t)2 * seq * KSTRIDE + 16 * 64) * sizeof(float);
      static bool attr = false;
      if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_lds_bwd_q_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_lds_bwd_kv_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
      }
      hipLaunchKernelGGL(attention_lds_bwd_q_kernel, dim3(batch * heads), dim3(512), lds, st, qkv, dout, out, lse, dqkv,
                         work, seq, heads, causal);
      CLIPFS_CHECK(launch_status());
      hipLaunchKernelGGL(attention_lds_bwd_kv_kernel, dim3(batch * heads), dim3(512), lds, st, qkv, dout, lse, work, dqkv,
                         seq, heads, causal);
      return launch_status();
    }
    const dim3 grid(batch * heads, (seq + 3) / 4);
    hipLaunchKernelGGL(attention_generic_bwd_q_kernel, grid, dim3(256), 0, st, qkv, dout, out, lse, dqkv, work, seq, heads,
                       causal);
    CLIPFS_CHECK(launch_status());
    hipLaunchKernelGGL(attention_generic_bwd_kv_kernel, grid, dim3(256), 0, st, qkv, dout, lse, work, dqkv, seq, heads,
                       causal);
    return launch_status();
  }
  const int lp = padded_lp(seq);
  // P^T, dS^T, and dS (whose storage first serves as the [seq][KSTRIDE] K/V staging buffer) + 16 floats overhang
  const size_t third = (size_t)seq * (lp > KSTRIDE ? lp : KSTRIDE);
  const size_t lds = ((size_t)2 * seq * lp + third + 16) * sizeof(float);
  CLIPFS_REQUIRE(lds <= 160 * 1024, "attention_bwd: seq %d needs %zu bytes of LDS (> 160 KiB)", seq, lds);
  static const int thr_cfg = getenv("CLIPFS_ATTN_BWD_THREADS") ? atoi(getenv("CLIPFS_ATTN_BWD_THREADS")) : 0;
  const dim3 grid(batch * heads), block(thr_cfg ? thr_cfg : (seq > 64 ? 512 : 256));
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_bwd_kernel<64>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_bwd_kernel<80>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_bwd_kernel<96>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (seq <= 64)
    hipLaunchKernelGGL((attention_bwd_kernel<64>), grid, block, lds, st, qkv, dout, dqkv, seq, heads, causal, lp);
  else if (seq <= 80)
    hipLaunchKernelGGL((attention_bwd_kernel<80>), grid, block, lds, st, qkv, dout, dqkv, seq, heads, causal, lp);
  else
    hipLaunchKernelGGL((attention_bwd_kernel<96>), grid, block, lds, st, qkv, dout, dqkv, seq, heads, causal, lp);
  return launch_status();
}

extern "C" size_t clipfs_attention_lse_floats(int batch, int seq, int heads) {
  if (attention_mfma_enabled() && seq <= ATTN_MFMA_MAX) return (size_t)batch * heads * seq;
  return seq > ATTN_FAST_MAX ? (size_t)batch * heads * seq : 0;
}
