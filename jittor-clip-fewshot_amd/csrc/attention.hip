// Self-attention forward / backward for head_dim 64 and short sequences (L <= 128: CLIP
// ViT-B/32 L = 50 or 54 with VPT, text L = 77), one workgroup (4 waves) per (batch, head).
//
// This is ~1 % of the layer's FLOPs (2*2*L^2*64 per head vs 24*L*d^2 per image for the GEMMs) and
// the fp32 matrix rate equals the fp32 vector rate on gfx950, so it runs on the VALU:
//   * scores: key-per-lane.  Lane j keeps K[j][0:64] in 64 VGPRs (second set for L > 64); the
//     query row is read from LDS as a broadcast; s_ij never leaves registers;
//   * softmax: wave64 max / sum by cross-lane shuffles;
//   * PV: lane = output column d; p_ij is broadcast with v_readlane, V[j][:] is an LDS row
//     (conflict free).  Heads are merged by the store address (no permute pass).
// The [L, L] score / probability matrix of the reference (jclip/mha.py:79-83, a round trip through
// HBM of 30-77 MB per layer) stays in registers (forward) or LDS (backward).
#include "common.h"

namespace clipfs {

constexpr int HD = 64;

__device__ __forceinline__ float bcast_lane(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// scores of query row `i` against this lane's keys (raw dot * scale, masked to -inf)
template <int KPL>
__device__ __forceinline__ void row_scores(const float* __restrict__ sQrow, const float (&kreg)[KPL][HD], int i,
                                           int lane, int L, int causal, float (&s)[KPL]) {
#pragma unroll
  for (int kk = 0; kk < KPL; ++kk) s[kk] = 0.f;
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    const float4 q = *reinterpret_cast<const float4*>(sQrow + 4 * c);
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
      s[kk] = fmaf(q.x, kreg[kk][4 * c + 0], s[kk]);
      s[kk] = fmaf(q.y, kreg[kk][4 * c + 1], s[kk]);
      s[kk] = fmaf(q.z, kreg[kk][4 * c + 2], s[kk]);
      s[kk] = fmaf(q.w, kreg[kk][4 * c + 3], s[kk]);
    }
  }
#pragma unroll
  for (int kk = 0; kk < KPL; ++kk) {
    const int j = lane + 64 * kk;
    s[kk] *= 0.125f;  // 1/sqrt(64), applied after the dot product (mha.py:79)
    if (j >= L || (causal && j > i)) s[kk] = -INFINITY;
  }
}

template <int KPL>
__device__ __forceinline__ void row_softmax(float (&s)[KPL]) {
  float m = s[0];
#pragma unroll
  for (int kk = 1; kk < KPL; ++kk) m = fmaxf(m, s[kk]);
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

template <int KPL>
__device__ __forceinline__ void load_rows_to_regs(const float* __restrict__ base, size_t row_stride, int lane, int L,
                                                  float (&reg)[KPL][HD]) {
#pragma unroll
  for (int kk = 0; kk < KPL; ++kk) {
    const int j = lane + 64 * kk;
    if (j < L) {
      const float4* src = reinterpret_cast<const float4*>(base + (size_t)j * row_stride);
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) {
        const float4 v = src[c];
        reg[kk][4 * c + 0] = v.x;
        reg[kk][4 * c + 1] = v.y;
        reg[kk][4 * c + 2] = v.z;
        reg[kk][4 * c + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int c = 0; c < HD; ++c) reg[kk][c] = 0.f;
    }
  }
}

__device__ __forceinline__ void stage_rows(const float* __restrict__ base, size_t row_stride, float* __restrict__ dst,
                                           int L, int tid) {
  for (int idx = tid; idx < L * (HD / 4); idx += 256) {
    const int r = idx >> 4, c = idx & 15;
    *reinterpret_cast<float4*>(dst + r * HD + 4 * c) =
        *reinterpret_cast<const float4*>(base + (size_t)r * row_stride + 4 * c);
  }
}

template <int KPL>
__global__ __launch_bounds__(256) void attention_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            int L, int H, int causal) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sQ = smem;           // [L][64]
  float* sV = smem + L * HD;  // [L][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * HD;
  const size_t ld = (size_t)3 * d;
  const float* q0 = qkv + (size_t)b * L * ld + (size_t)h * HD;
  stage_rows(q0, ld, sQ, L, tid);
  stage_rows(q0 + 2 * d, ld, sV, L, tid);
  float kreg[KPL][HD];
  load_rows_to_regs<KPL>(q0 + d, ld, lane, L, kreg);
  __syncthreads();
  for (int i = wave; i < L; i += 4) {
    float s[KPL];
    row_scores<KPL>(sQ + i * HD, kreg, i, lane, L, causal, s);
    row_softmax<KPL>(s);
    const int jmax = causal ? i + 1 : L;
    float o = 0.f;
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
      const int jend = min(jmax - 64 * kk, 64);
      for (int j = 0; j < jend; ++j) o = fmaf(bcast_lane(s[kk], j), sV[(j + 64 * kk) * HD + lane], o);
    }
    out[((size_t)b * L + i) * d + h * HD + lane] = o;
  }
}

// Backward: dQ = scale * dS K, dK = scale * dS^T Q, dV = P^T dO with dS = P * (dP - rowsum(P*dP)),
// dP = dO V^T.  P is recomputed (pass 1), never stored in HBM.
template <int KPL>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const float* __restrict__ qkv,
                                                            const float* __restrict__ dout,
                                                            float* __restrict__ dqkv, int L, int H, int causal) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LP = L + 1;  // padded row of the [L][L] matrices: column reads are conflict free
  float* sQ = smem;
  float* sK = sQ + L * HD;
  float* sdO = sK + L * HD;
  float* sP = sdO + L * HD;  // [L][LP]
  float* sdS = sP + L * LP;  // [L][LP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * HD;
  const size_t ld = (size_t)3 * d;
  const float* q0 = qkv + (size_t)b * L * ld + (size_t)h * HD;
  const float* do0 = dout + (size_t)b * L * d + (size_t)h * HD;
  float* dq0 = dqkv + (size_t)b * L * ld + (size_t)h * HD;
  stage_rows(q0, ld, sQ, L, tid);
  stage_rows(q0 + d, ld, sK, L, tid);
  stage_rows(do0, (size_t)d, sdO, L, tid);
  float reg[KPL][HD];
  load_rows_to_regs<KPL>(q0 + d, ld, lane, L, reg);  // K rows
  __syncthreads();
  // pass 1: probabilities
  for (int i = wave; i < L; i += 4) {
    float s[KPL];
    row_scores<KPL>(sQ + i * HD, reg, i, lane, L, causal, s);
    row_softmax<KPL>(s);
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
      const int j = lane + 64 * kk;
      if (j < L) sP[i * LP + j] = s[kk];
    }
  }
  load_rows_to_regs<KPL>(q0 + 2 * d, ld, lane, L, reg);  // V rows replace K rows
  __syncthreads();
  // pass 2: dP, dS (kept in LDS for dK) and dQ rows
  for (int i = wave; i < L; i += 4) {
    float dp[KPL], pv[KPL];
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) dp[kk] = 0.f;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      const float4 g = *reinterpret_cast<const float4*>(sdO + i * HD + 4 * c);
#pragma unroll
      for (int kk = 0; kk < KPL; ++kk) {
        dp[kk] = fmaf(g.x, reg[kk][4 * c + 0], dp[kk]);
        dp[kk] = fmaf(g.y, reg[kk][4 * c + 1], dp[kk]);
        dp[kk] = fmaf(g.z, reg[kk][4 * c + 2], dp[kk]);
        dp[kk] = fmaf(g.w, reg[kk][4 * c + 3], dp[kk]);
      }
    }
    float rs = 0.f;
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
      const int j = lane + 64 * kk;
      pv[kk] = j < L ? sP[i * LP + j] : 0.f;
      rs = fmaf(pv[kk], dp[kk], rs);
    }
    rs = wave_sum(rs);
    float ds[KPL];
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
      const int j = lane + 64 * kk;
      ds[kk] = pv[kk] * (dp[kk] - rs) * 0.125f;
      if (j < L) sdS[i * LP + j] = ds[kk];
    }
    const int jmax = causal ? i + 1 : L;
    float acc = 0.f;
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
      const int jend = min(jmax - 64 * kk, 64);
      for (int j = 0; j < jend; ++j) acc = fmaf(bcast_lane(ds[kk], j), sK[(j + 64 * kk) * HD + lane], acc);
    }
    dq0[(size_t)i * ld + lane] = acc;
  }
  __syncthreads();
  // pass 3: per key j, dV[j] = sum_i P_ij dO_i and dK[j] = sum_i dS_ij Q_i  (lane = column)
  for (int j = wave; j < L; j += 4) {
    const int i0 = causal ? j : 0;  // P_ij = 0 for i < j under the causal mask
    float av = 0.f, ak = 0.f;
    for (int ib = i0 & ~63; ib < L; ib += 64) {
      const int ii = ib + lane;
      const float pcol = ii < L ? sP[ii * LP + j] : 0.f;   // lane ii holds P[ii][j]
      const float dcol = ii < L ? sdS[ii * LP + j] : 0.f;  // and dS[ii][j]
      const int lo = max(i0 - ib, 0), hi = min(L - ib, 64);
      for (int t = lo; t < hi; ++t) {
        av = fmaf(bcast_lane(pcol, t), sdO[(ib + t) * HD + lane], av);
        ak = fmaf(bcast_lane(dcol, t), sQ[(ib + t) * HD + lane], ak);
      }
    }
    dq0[(size_t)j * ld + d + lane] = ak;
    dq0[(size_t)j * ld + 2 * d + lane] = av;
  }
}

static int check_attn(const char* what, int batch, int seq, int heads, int max_seq) {
  CLIPFS_REQUIRE(batch > 0 && heads > 0 && seq > 0 && seq <= max_seq, "%s: batch %d seq %d heads %d unsupported (seq <= %d)",
                 what, batch, seq, heads, max_seq);
  return CLIPFS_OK;
}

}  // namespace clipfs

using namespace clipfs;

extern "C" int clipfs_attention_fwd(const float* qkv, float* out, int batch, int seq, int heads, int causal,
                                    void* stream) {
  CLIPFS_CHECK(check_attn("attention_fwd", batch, seq, heads, 128));
  CLIPFS_REQUIRE(qkv && out && aligned16(qkv), "attention_fwd: null or misaligned pointer");
  const size_t lds = (size_t)2 * seq * HD * sizeof(float);
  const dim3 grid(batch * heads), block(256);
  if (seq <= 64)
    hipLaunchKernelGGL((attention_fwd_kernel<1>), grid, block, lds, (hipStream_t)stream, qkv, out, seq, heads, causal);
  else {
    static bool attr = false;
    if (!attr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_fwd_kernel<2>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * HD * 4);
      attr = true;
    }
    hipLaunchKernelGGL((attention_fwd_kernel<2>), grid, block, lds, (hipStream_t)stream, qkv, out, seq, heads, causal);
  }
  return launch_status();
}

extern "C" int clipfs_attention_bwd(const float* qkv, const float* dout, float* dqkv, int batch, int seq, int heads,
                                    int causal, void* stream) {
  CLIPFS_CHECK(check_attn("attention_bwd", batch, seq, heads, 96));  // LDS: 3*L*64 + 2*L*(L+1) floats <= 160 KiB
  CLIPFS_REQUIRE(qkv && dout && dqkv && aligned16(qkv) && aligned16(dout), "attention_bwd: null or misaligned pointer");
  const size_t lds = ((size_t)3 * seq * HD + (size_t)2 * seq * (seq + 1)) * sizeof(float);
  const dim3 grid(batch * heads), block(256);
  static bool attr = false;
  if (!attr) {
    const int maxlds = (3 * 96 * HD + 2 * 96 * 97) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_bwd_kernel<1>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_bwd_kernel<2>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
    attr = true;
  }
  if (seq <= 64)
    hipLaunchKernelGGL((attention_bwd_kernel<1>), grid, block, lds, (hipStream_t)stream, qkv, dout, dqkv, seq, heads,
                       causal);
  else
    hipLaunchKernelGGL((attention_bwd_kernel<2>), grid, block, lds, (hipStream_t)stream, qkv, dout, dqkv, seq, heads,
                       causal);
  return launch_status();
}
