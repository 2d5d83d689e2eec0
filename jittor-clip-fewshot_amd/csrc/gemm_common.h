// Pieces shared by the fp32-MFMA GEMM (gemm.hip) and the split-bf16 GEMM (gemm_bf16.hip).
#pragma once
#include "common.h"

namespace clipfs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmParams {
  clipfs_gemm_args a;
  int n_blocks_n;  // number of BN-wide column blocks
  int patches;     // a_mode 1: patches per image (G*G)
  int grid_g;      // a_mode 1: patches per side
  int splits;      // split-K factor (1 = none): unit u = tile * splits + split, split s covers K-steps [s*nk/S, (s+1)*nk/S)
  float* part;     // splits > 1: raw partial sums, slab s at part + s * M * N (row-major, ld = N)
  int gm;          // super-tile height in m-blocks (tile order: gm m-blocks x all n-blocks, m fastest)
  int ablate;      // tuning aid (CLIPFS_GEMM_ABLATE): 1 no global prefetch, 2 no LDS store, 4 no barrier -- WRONG RESULTS
  // stream-K launch (gemm_sk_kernel): the tiles x K-steps iteration space is cut into gridDim.x equal contiguous runs
  int sk_nk;        // K-steps per tile
  int sk_tile0;     // first tile of the stream-K part (= number of tiles dealt whole)
  int sk_dp_rounds; // whole tiles per run
  int sk_q, sk_r;   // stream-K piece of run w: K-steps [w q + min(w, r), ...) of length q + (w < r)
  float* sk_part;  // partial-tile slabs: slot (2 w + kind) of BM*BN floats, w = run index, kind 0 = the run's first
                   // (head) tile, 1 = its last (tail) tile
  int* sk_cnt;     // arrival counter per tile (zero on entry, zero again on exit)
};

constexpr int BK = 32;

// The fused epilogue for one output element (order documented in include/clipfs.h); used by the GEMM
// kernel and by the split-K combine kernel.
// lora_done: the rank-r term is already inside accv (added on the matrix cores after the K loop; alpha == 1 there)
__device__ __forceinline__ void epilogue_store(const clipfs_gemm_args& g, int patches, int m, int n, float accv,
                                               bool lora_done = false) {
  float v = g.alpha * accv + (g.bias ? g.bias[n] : 0.f);
  if (g.lora_t && !lora_done) {
    const int lseg = n / g.lora_seg_width;
    const float* lb = g.lora_b + (size_t)n * g.lora_r;
    const float* t = g.lora_t + (size_t)m * (g.lora_nseg * g.lora_r) + lseg * g.lora_r;
    float d = 0.f;
    for (int jj = 0; jj < g.lora_r; ++jj) d = fmaf(t[jj], lb[jj], d);
    v = fmaf(g.lora_scale, d, v);
  }
  size_t orow = (size_t)m, rrow = (size_t)m;
  if (g.a_mode == 1) {
    const int b = m / patches, pp = m - b * patches;
    orow = (size_t)b * g.out_tokens + 1 + pp;
    rrow = (size_t)(1 + pp);
  }
  if (g.act == 1) {
    if (g.aux_out) g.aux_out[orow * g.ldc + n] = v;
    v = quick_gelu(v);
  } else if (g.act == 2) {
    v *= quick_gelu_grad(g.aux_in[orow * g.ldc + n]);
  }
  if (g.residual) v += g.residual[rrow * g.ldres + n];
  if (g.act == 3) v = fmaxf(v, 0.f);  // ReLU AFTER the residual add (ResNet bottleneck: relu(conv + identity))
  g.C[orow * g.ldc + n] = v;
}


// What every 32x32-MFMA GEMM kernel does after its K loop, for the TM x TN accumulator tiles of one wave whose
// first row / column are mw / nw (C/D map of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) +
// 4 (lane >> 5)):
//  1. rank-r LoRA up-projection on the matrix cores: ceil(r / 2) more K-steps of v_mfma_f32_32x32x2_f32 with
//     A = lora_scale * t[m, seg r + k], B = lora_b[n, k] (a 32-column tile lies inside one segment: the host checks
//     lora_seg_width % 32 == 0) -- the per-element epilogue would cost 2 x 16-byte loads + r FMAs per output;
//  2. the epilogue.  Dense rows and the whole block tile inside the problem (full_tile): every option is a
//     wave-uniform branch around a 16-element pass with all of its loads issued together; otherwise the generic
//     per-element path (ragged edges, patch row remap).
template <int TM, int TN>
__device__ __forceinline__ void finish_tiles(const clipfs_gemm_args& g, int patches, f32x16 (&acc)[TM][TN], int mw, int nw,
                                             bool full_tile, int lane) {
  const int fr = lane & 31, fh = lane >> 5;
  const int M = g.M, N = g.N;
  const bool lora_mfma = g.lora_t && g.alpha == 1.f;
  if (lora_mfma) {
    const int r = g.lora_r, tw = g.lora_nseg * r;
    const int rsteps = (r + 1) >> 1;
    for (int st = 0; st < rsteps; ++st) {
      const int k = 2 * st + fh;
      const float kmask = k < r ? 1.f : 0.f;
      const int kc2 = min(k, r - 1);
      float bv[TN];
      int segs[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nb = nw + j * 32;
        segs[j] = min(nb, N - 1) / g.lora_seg_width;
        bv[j] = g.lora_b[(size_t)min(nb + fr, N - 1) * r + kc2] * kmask;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = min(mw + i * 32 + fr, M - 1);
        const float* tp = g.lora_t + (size_t)m * tw + kc2;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float av = g.lora_scale * tp[segs[j] * r];
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  }
  if (g.a_mode == 0 && full_tile && (lora_mfma || !g.lora_t)) {
    const int ldc = g.ldc;
    // Every per-element LOAD of the wave's tiles (residual rows, else the saved pre-activation) is issued before the
    // first STORE: vmcnt retires in issue order, so a load queued behind a store also waits for that store's
    // acknowledgement (microseconds under load) -- round 2 loaded tile (i, j)'s residual after tile (i, j-1)'s stores.
    const bool pre_res = g.residual != nullptr, pre_aux = !pre_res && g.act == 2;
    float pre[TM][TN][16];
    if (pre_res || pre_aux) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int mb = mw + i * 32 + 4 * fh, n = nw + j * 32 + fr;
          const float* q = pre_res ? g.residual + (size_t)mb * g.ldres + n : g.aux_in + (size_t)mb * ldc + n;
          const int ldq = pre_res ? g.ldres : ldc;
#pragma unroll
          for (int r = 0; r < 16; ++r) pre[i][j][r] = q[((r & 3) + 8 * (r >> 2)) * ldq];
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = nw + j * 32 + fr;
      const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mb = mw + i * 32 + 4 * fh;
        const size_t base = (size_t)mb * ldc + n;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = g.alpha * acc[i][j][r] + bias;
        if (g.act == 1) {
          if (g.aux_out) {
            float* q = g.aux_out + base;
#pragma unroll
            for (int r = 0; r < 16; ++r) q[((r & 3) + 8 * (r >> 2)) * ldc] = v[r];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = quick_gelu(v[r]);
        } else if (g.act == 2) {
          if (pre_aux) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] *= quick_gelu_grad(pre[i][j][r]);
          } else {  // act 2 together with a residual (not on the path): the pre-activation is loaded here
            const float* q = g.aux_in + base;
            float u[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = q[((r & 3) + 8 * (r >> 2)) * ldc];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] *= quick_gelu_grad(u[r]);
          }
        }
        if (pre_res) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] += pre[i][j][r];
        }
        if (g.act == 3) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        float* q = g.C + base;
#pragma unroll
        for (int r = 0; r < 16; ++r) q[((r & 3) + 8 * (r >> 2)) * ldc] = v[r];
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = nw + j * 32 + fr;
    if (n >= N) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (m < M) epilogue_store(g, patches, m, n, acc[i][j][r], lora_mfma);
      }
    }
  }
}


// Tile order shared by both kernels: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the
// XCD, speed only); every XCD gets a CONTIGUOUS run of the unit list.
__device__ __forceinline__ int xcd_contiguous_unit() {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, li = bid >> 3, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + li;
}

}  // namespace clipfs
