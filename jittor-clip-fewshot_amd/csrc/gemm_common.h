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
  int ablate;      // tuning aid (CLIPFS_GEMM_ABLATE): 1 no global prefetch, 2 no LDS store, 4 no barrier -- WRONG RESULTS
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
  g.C[orow * g.ldc + n] = v;
}


// Tile order shared by both kernels: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the
// XCD, speed only); every XCD gets a CONTIGUOUS run of the unit list.
__device__ __forceinline__ int xcd_contiguous_unit() {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, li = bid >> 3, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + li;
}

}  // namespace clipfs
