// Split-bf16 ("bf16 x 3") NT GEMM:  C = epi( A * B^T ) with fp32 operands represented as hi + lo bf16 pairs and
// three bf16 MFMA products per pair, accumulated in fp32:
//        a * b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (a_hi = bf16(a), a_lo = bf16(a - a_hi))
// SURVEY.md section 7 names this as the alternative to the exact-fp32 MFMA: v_mfma_f32_32x32x16_bf16 runs at 16x
// the FLOP rate of v_mfma_f32_32x32x2_f32, so three products are still 5.3x faster; per product the dropped
// terms are ~2^-17 relative.  Measured on the full ViT-B/32 + text towers (oracle emulation) the 100 x cosine
// logits move by 2.2e-4 (exact fp32: 1e-5; tolerance 1e-3).  It is an opt-in precision mode of the engine.
//
//   * B (frozen weights) is split ONCE into two bf16 planes [N, K] (clipfs_split_bf16) and streamed straight into
//     LDS with global_load_lds_dwordx4 (bytes per element: 2 + 2, the same as fp32);
//   * A (fp32 activations) is split on the fly in the staging path: global fp32 -> VGPR -> hi/lo -> two LDS planes,
//     so no other kernel changes and HBM traffic stays that of the fp32 GEMM;
//   * LDS image per plane: [rows][32 k] bf16 = 4 chunks of 16 B per row, chunk XOR (row >> 2) & 3
//     (conflict-free ds_read_b128 of the 8-wide bf16 fragments: lane l holds k = 8 (l >> 5) .. +7 of row l & 31);
//   * block tile BM x BN x 32, 4 waves (2 x 2), double-buffered, one barrier per K-step; epilogue and tile order are
//     shared with the fp32 kernel (gemm_common.h).
#include "gemm_common.h"

#include <stdlib.h>

namespace clipfs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16b(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi[j] = (__bf16)x0[j];
    hi[4 + j] = (__bf16)x1[j];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    lo[j] = (__bf16)(x0[j] - (float)hi[j]);
    lo[4 + j] = (__bf16)(x1[j] - (float)hi[4 + j]);
  }
}

__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ hi,
                                                         __bf16* __restrict__ lo, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(src + 8 * i);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(src + 8 * i + 4);
    bf16x8 h, l;
    split8(x0, x1, h, l);
    *reinterpret_cast<bf16x8*>(hi + 8 * i) = h;
    *reinterpret_cast<bf16x8*>(lo + 8 * i) = l;
  }
}

struct Bf16Params {
  GemmParams g;
  const void* b_hi;  // [N, K] 16-bit plane (bf16 hi, or the single f16 plane)
  const void* b_lo;  // bf16 lo plane (unused for f16)
};

__device__ __forceinline__ f16x8 to_f16x8(const f32x4& x0, const f32x4& x1) {
  f16x8 h;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = (_Float16)x0[j];
    h[4 + j] = (_Float16)x1[j];
  }
  return h;
}

__global__ __launch_bounds__(256) void convert_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst,
                                                          size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256)
    *reinterpret_cast<f16x8*>(dst + 8 * i) = to_f16x8(*reinterpret_cast<const f32x4*>(src + 8 * i),
                                                      *reinterpret_cast<const f32x4*>(src + 8 * i + 4));
}

// NPL = number of 16-bit planes per operand: 2 = bf16 hi/lo (3 MFMA products), 1 = f16 (1 product; cfg-5's fp16 path)
template <int BM, int BN, int NPL>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(const Bf16Params bp) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int TM = BM / 64, TN = BN / 64;          // 32x32 tiles per wave (2 x 2 waves)
  constexpr int A_UNITS = BM * 4 / 256;              // (row, 8-k chunk) units staged per thread
  constexpr int B_INSTR = NPL * BN / 16 / 4;         // global_load_lds instructions per wave per K-step
  constexpr int PLANE_A = BM * 64, PLANE_B = BN * 64;  // bytes
  constexpr int STAGE = NPL * PLANE_A + NPL * PLANE_B;

  const GemmParams& p = bp.g;
  const clipfs_gemm_args& g = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = g.M, N = g.N, K = g.K;

  int tile = xcd_contiguous_unit();
  constexpr int GM = 8;
  const int nbn = p.n_blocks_n;
  const int grp = tile / (GM * nbn);
  const int rem = tile - grp * (GM * nbn);
  const int mb_total = (M + BM - 1) / BM;
  const int gmm = min(GM, mb_total - grp * GM);
  const int m0 = (grp * GM + rem % gmm) * BM;
  const int n0 = (rem / gmm) * BN;

  // ---- A staging (fp32 -> registers -> hi/lo planes) ----------------------------------------------------
  const float* a_src[A_UNITS];
  int a_off[A_UNITS];
#pragma unroll
  for (int i = 0; i < A_UNITS; ++i) {
    const int u = tid + 256 * i, row = u >> 2, c = u & 3;
    a_src[i] = g.A + (size_t)min(m0 + row, M - 1) * g.lda + 8 * c;
    a_off[i] = row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
  }
  // ---- B staging (pre-split planes -> LDS directly): instruction q covers 16 rows x 64 B of one plane ------------
  const char* b_src[B_INSTR];
  int b_off[B_INSTR];
  {
    const int uw = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
      const int q = uw + 4 * i, plane = q / (BN / 16), r0 = 16 * (q % (BN / 16));
      const int row = r0 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
      const char* base = reinterpret_cast<const char*>(plane ? bp.b_lo : bp.b_hi);
      b_src[i] = base + ((size_t)min(n0 + row, N - 1) * g.ldb + 8 * c) * 2;
      b_off[i] = NPL * PLANE_A + plane * PLANE_B + r0 * 64;
    }
  }
  // ---- fragment addresses ---------------------------------------------------------------------------------
  const int fr = lane & 31, fh = lane >> 5;
  int a_frag[TM], b_frag[TN], a_swz[TM], b_swz[TN];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int row = wm * (BM / 2) + t * 32 + fr;
    a_frag[t] = row * 64;
    a_swz[t] = (row >> 2) & 3;
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int row = wn * (BN / 2) + t * 32 + fr;
    b_frag[t] = NPL * PLANE_A + row * 64;
    b_swz[t] = (row >> 2) & 3;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 areg[A_UNITS][2];
  auto load_a = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < A_UNITS; ++i) {
      areg[i][0] = *reinterpret_cast<const f32x4*>(a_src[i] + kt * BK);
      areg[i][1] = *reinterpret_cast<const f32x4*>(a_src[i] + kt * BK + 4);
    }
  };
  auto glds_b = [&](int kt, int stage) __attribute__((always_inline)) {
    char* s = smem_raw + stage * STAGE;
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) glds16b(b_src[i] + (size_t)kt * BK * 2, s + b_off[i]);
  };
  auto store_a = [&](int stage) __attribute__((always_inline)) {
    char* s = smem_raw + stage * STAGE;
#pragma unroll
    for (int i = 0; i < A_UNITS; ++i) {
      if constexpr (NPL == 2) {
        bf16x8 hi, lo;
        split8(areg[i][0], areg[i][1], hi, lo);
        *reinterpret_cast<bf16x8*>(s + a_off[i]) = hi;
        *reinterpret_cast<bf16x8*>(s + PLANE_A + a_off[i]) = lo;
      } else {
        *reinterpret_cast<f16x8*>(s + a_off[i]) = to_f16x8(areg[i][0], areg[i][1]);
      }
    }
  };
  auto compute = [&](const char* s) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if constexpr (NPL == 2) {
        bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const int o = a_frag[t] + (((2 * kb + fh) ^ a_swz[t]) << 4);
          ah[t] = *reinterpret_cast<const bf16x8*>(s + o);
          al[t] = *reinterpret_cast<const bf16x8*>(s + PLANE_A + o);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          const int o = b_frag[t] + (((2 * kb + fh) ^ b_swz[t]) << 4);
          bh[t] = *reinterpret_cast<const bf16x8*>(s + o);
          bl[t] = *reinterpret_cast<const bf16x8*>(s + PLANE_B + o);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      } else {
        f16x8 av[TM], bv[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t)
          av[t] = *reinterpret_cast<const f16x8*>(s + a_frag[t] + (((2 * kb + fh) ^ a_swz[t]) << 4));
#pragma unroll
        for (int t = 0; t < TN; ++t)
          bv[t] = *reinterpret_cast<const f16x8*>(s + b_frag[t] + (((2 * kb + fh) ^ b_swz[t]) << 4));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  const int nk = K / BK;
  load_a(0);
  glds_b(0, 0);
  store_a(0);
  __syncthreads();
  for (int kt = 0; kt + 1 < nk; ++kt) {
    load_a(kt + 1);
    glds_b(kt + 1, (kt + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(smem_raw + (kt & 1) * STAGE);
    store_a((kt + 1) & 1);
    __syncthreads();
  }
  compute(smem_raw + ((nk - 1) & 1) * STAGE);

  finish_tiles<TM, TN>(g, p.patches, acc, m0 + wm * (BM / 2), n0 + wn * (BN / 2), m0 + BM <= M && n0 + BN <= N, lane);
}

template <int BM, int BN, int NPL>
static int launch_bf16(Bf16Params& bp, hipStream_t stream) {
  const clipfs_gemm_args& a = bp.g.a;
  bp.g.n_blocks_n = (a.N + BN - 1) / BN;
  const int mb = (a.M + BM - 1) / BM;
  const size_t lds = 2 * (size_t)(NPL * BM * 64 + NPL * BN * 64);
  static bool attr = false;
  if (!attr && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3_kernel<BM, BN, NPL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, NPL>), dim3(mb * bp.g.n_blocks_n), dim3(256), lds, stream, bp);
  return launch_status();
}

// called from clipfs_gemm_nt when args->B_planes is set and the shape qualifies
int gemm_bf16x3_dispatch(const GemmParams& base, hipStream_t stream) {
  Bf16Params bp;
  bp.g = base;
  bp.g.splits = 1;
  const clipfs_gemm_args& a = base.a;
  bp.b_hi = a.B_planes;
  bp.b_lo = reinterpret_cast<const char*>(a.B_planes) + (size_t)a.N * a.ldb * 2;
  static const int tile_cfg = getenv("CLIPFS_BF16_TILE") ? atoi(getenv("CLIPFS_BF16_TILE")) : 0;
  const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
  const bool big = tile_cfg == 1 || (tile_cfg == 0 && t128 >= 1024);
  if (a.b_format == 2) return big ? launch_bf16<128, 128, 1>(bp, stream) : launch_bf16<64, 128, 1>(bp, stream);
  return big ? launch_bf16<128, 128, 2>(bp, stream) : launch_bf16<64, 128, 2>(bp, stream);
}

}  // namespace clipfs

using namespace clipfs;

extern "C" int clipfs_convert_f16(const float* src, void* dst, size_t n, void* stream) {
  CLIPFS_REQUIRE(src && dst && n > 0 && (n & 7) == 0 && aligned16(src) && aligned16(dst),
                 "convert_f16: n must be a multiple of 8, pointers 16-byte aligned");
  const size_t n8 = n / 8;
  const unsigned blocks = (unsigned)((n8 + 255) / 256 > 8192 ? 8192 : (n8 + 255) / 256);
  hipLaunchKernelGGL(convert_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src,
                     reinterpret_cast<_Float16*>(dst), n8);
  return launch_status();
}

extern "C" int clipfs_split_bf16(const float* src, void* planes, size_t n, void* stream) {
  CLIPFS_REQUIRE(src && planes && n > 0 && (n & 7) == 0 && aligned16(src) && aligned16(planes),
                 "split_bf16: n must be a multiple of 8, pointers 16-byte aligned");
  __bf16* hi = reinterpret_cast<__bf16*>(planes);
  const size_t n8 = n / 8;
  const unsigned blocks = (unsigned)((n8 + 255) / 256 > 8192 ? 8192 : (n8 + 255) / 256);
  hipLaunchKernelGGL(split_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, hi, hi + n, n8);
  return launch_status();
}
