// f16 x f16 -> fp32-accumulate NT GEMM for the fp16 storage mode (cfg-5: ViT-L/14):  C = epi( A16 * B16^T ).
//
// Both operands already live in HBM as f16 (the producing kernels write an f16 copy of every tensor that feeds a
// GEMM; weights are converted once), so the whole K-loop is: global_load_lds_dwordx4 -> LDS -> ds_read_b128 ->
// v_mfma_f32_32x32x16_f16.  No register staging, no conversion.
//   * block tile BM x BN x 32, 4 waves as 2 x 2, wave tile (BM/2) x (BN/2) in 32x32 MFMA tiles; 256x128 is the
//     main shape (128 fp32 accumulators per lane, two workgroups per CU);
//   * LDS image per operand: [rows][32 k] f16 = 4 chunks of 16 B per row, chunk XOR (row >> 2) & 3 -- the same
//     conflict-free layout as gemm_bf16.hip; three stages filled two K-steps ahead, one barrier per K-step;
//   * the rank-r LoRA up-projection (r <= 16) is ONE extra MFMA K-step per tile: A-fragment = the rows of t,
//     B-fragment = lora_scale * lora_b (zero padded to k = 16), both converted in registers;
//   * epilogue as gemm_common.h documents it (bias, QuickGELU / x dQuickGELU, residual), writing fp32 C and/or an
//     f16 copy C16 for the next GEMM.
// Rows that do not fill a whole BM block are better served by a second launch with a smaller tile (the caller --
// gemm_f16_dispatch -- cuts M so that the big launch is an exact number of rounds over the CUs).
#include "gemm_common.h"

#include <stdint.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>
#include <vector>

namespace clipfs {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// QuickGELU and its derivative with v_exp_f32 / v_rcp_f32 (1 ulp) instead of the IEEE division of common.h: at
// K = 1024 the 128 activations per lane of a 256x128 tile cost as many cycles as a third of the K loop.
__device__ __forceinline__ float fast_sigmoid_1702(float u) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * u));
}
__device__ __forceinline__ float quick_gelu_fast(float u) { return u * fast_sigmoid_1702(u); }
__device__ __forceinline__ float quick_gelu_grad_fast(float u) {
  const float sg = fast_sigmoid_1702(u);
  return sg * (1.f + 1.702f * u * (1.f - sg));
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct F16Params {
  clipfs_gemm_args a;
  const _Float16* A16;
  const _Float16* B16;
  _Float16* C16;
  int m_begin;      // first row of this launch (rows [m_begin, m_end) of the problem)
  int m_end;
  int n_blocks_n;
  int row_major_epilogue;  // 256 x 256 kernel: epilogue through LDS with 8 / 16-byte row accesses (needs 4-column alignment)
};

// What both f16 kernels do after their K loop for the TM x TN accumulator tiles of one wave whose first row / column
// are mw / nw: the rank-r LoRA up-projection as ONE more MFMA K-step, then the fused epilogue.
template <int TM, int TN>
__device__ __forceinline__ void f16_lora_step(const F16Params& p, f32x16 (&acc)[TM][TN], int mw, int nw, int n0, int lane) {
  const clipfs_gemm_args& g = p.a;
  const int fr = lane & 31, fh = lane >> 5;
  const int Mend = p.m_end, N = g.N;
  // ---- LoRA up-projection: one more K-step of 16 (rank zero-padded), operands converted in registers ----------
  if (g.lora_t) {
    const int r = g.lora_r;
    const int seg = n0 / g.lora_seg_width;  // the host guarantees a block tile lies inside one segment
    f16x8 av[TM], bv[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int m = min(mw + t * 32 + fr, Mend - 1);
      const float* tp = g.lora_t + (size_t)m * (g.lora_nseg * r) + seg * r;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * fh + j;
        av[t][j] = (_Float16)(tp[min(k, r - 1)] * (k < r ? 1.f : 0.f));
      }
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const int n = min(nw + t * 32 + fr, N - 1);
      const float* lb = g.lora_b + (size_t)n * r;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * fh + j;
        bv[t][j] = (_Float16)(lb[min(k, r - 1)] * (k < r ? g.lora_scale : 0.f));
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i], bv[j], acc[i][j], 0, 0, 0);
  }
}

// the same step for 8 x 4 accumulator tiles of v_mfma_f32_16x16x32_f16 (rank zero-padded to k = 32)
__device__ __forceinline__ void f16_lora_step16(const F16Params& p, f32x4 (&acc)[8][4], int mw, int nw, int n0, int lane) {
  const clipfs_gemm_args& g = p.a;
  if (!g.lora_t) return;
  const int li = lane & 15, lg = lane >> 4;
  const int Mend = p.m_end, N = g.N;
  const int r = g.lora_r;
  const int seg = n0 / g.lora_seg_width;
  f16x8 bv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = min(nw + t * 16 + li, N - 1);
    const float* lb = g.lora_b + (size_t)n * r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * lg + j;
      bv[t][j] = (_Float16)(lb[min(k, r - 1)] * (k < r ? g.lora_scale : 0.f));
    }
  }
  f16x8 av[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int m = min(mw + t * 16 + li, Mend - 1);
    const float* tp = g.lora_t + (size_t)m * (g.lora_nseg * r) + seg * r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * lg + j;
      av[t][j] = (_Float16)(tp[min(k, r - 1)] * (k < r ? 1.f : 0.f));
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[i], bv[j], acc[i][j], 0, 0, 0);
}

template <int TM, int TN>
__device__ __forceinline__ void f16_finish(const F16Params& p, f32x16 (&acc)[TM][TN], int mw, int nw, int n0, bool full_tile,
                                           int lane) {
  const clipfs_gemm_args& g = p.a;
  const int fr = lane & 31, fh = lane >> 5;
  const int Mend = p.m_end, N = g.N;
  f16_lora_step<TM, TN>(p, acc, mw, nw, n0, lane);

  // ---- epilogue: lane owns column n of each 32x32 tile and 16 of its rows (4 groups of 4 consecutive) ---------
  // Row offsets inside a 32x32 tile: (r & 3) + 8 (r >> 2) + 4 fh.  Every option is a wave-uniform branch around a
  // 16-element pass so the common cases stay straight-line.
  const int ldc = g.ldc;
  if (full_tile) {  // whole tile inside the problem: no predicates, loads of a pass issued together
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
            if (g.aux_f16) {
              _Float16* q = reinterpret_cast<_Float16*>(g.aux_out) + base;
#pragma unroll
              for (int r = 0; r < 16; ++r) q[((r & 3) + 8 * (r >> 2)) * ldc] = (_Float16)v[r];
            } else {
              float* q = g.aux_out + base;
#pragma unroll
              for (int r = 0; r < 16; ++r) q[((r & 3) + 8 * (r >> 2)) * ldc] = v[r];
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = quick_gelu_fast(v[r]);
        } else if (g.act == 2) {
          float u[16];
          if (g.aux_f16) {
            const _Float16* q = reinterpret_cast<const _Float16*>(g.aux_in) + base;
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = (float)q[((r & 3) + 8 * (r >> 2)) * ldc];
          } else {
            const float* q = g.aux_in + base;
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = q[((r & 3) + 8 * (r >> 2)) * ldc];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] *= quick_gelu_grad_fast(u[r]);
        }
        if (g.residual) {
          const float* q = g.residual + (size_t)mb * g.ldres + n;
          float u[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) u[r] = q[((r & 3) + 8 * (r >> 2)) * g.ldres];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] += u[r];
        }
        if (g.C) {
          float* q = g.C + base;
#pragma unroll
          for (int r = 0; r < 16; ++r) q[((r & 3) + 8 * (r >> 2)) * ldc] = v[r];
        }
        if (p.C16) {
          _Float16* q = p.C16 + base;
#pragma unroll
          for (int r = 0; r < 16; ++r) q[((r & 3) + 8 * (r >> 2)) * ldc] = (_Float16)v[r];
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = nw + j * 32 + fr;
    const bool n_ok = n < N;
    const float bias = (g.bias && n_ok) ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = mw + i * 32 + 4 * fh;
      const size_t base = (size_t)mb * ldc + n;
      int ok = 0;  // bit r set: element r is inside the problem
#pragma unroll
      for (int r = 0; r < 16; ++r) ok |= (n_ok && mb + (r & 3) + 8 * (r >> 2) < Mend) ? (1 << r) : 0;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = g.alpha * acc[i][j][r] + bias;
      if (g.act == 1) {
        if (g.aux_out) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (ok >> r & 1) {
              const size_t o = base + (size_t)((r & 3) + 8 * (r >> 2)) * ldc;
              if (g.aux_f16)
                reinterpret_cast<_Float16*>(g.aux_out)[o] = (_Float16)v[r];
              else
                g.aux_out[o] = v[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = quick_gelu_fast(v[r]);
      } else if (g.act == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (ok >> r & 1) {
            const size_t o = base + (size_t)((r & 3) + 8 * (r >> 2)) * ldc;
            v[r] *= quick_gelu_grad_fast(g.aux_f16 ? (float)reinterpret_cast<const _Float16*>(g.aux_in)[o] : g.aux_in[o]);
          }
      }
      if (g.residual) {
        const float* q = g.residual + (size_t)mb * g.ldres + n;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (ok >> r & 1) v[r] += q[((r & 3) + 8 * (r >> 2)) * g.ldres];
      }
      if (g.C) {
        float* q = g.C + base;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (ok >> r & 1) q[((r & 3) + 8 * (r >> 2)) * ldc] = v[r];
      }
      if (p.C16) {
        _Float16* q = p.C16 + base;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (ok >> r & 1) q[((r & 3) + 8 * (r >> 2)) * ldc] = (_Float16)v[r];
      }
    }
  }
}

template <int BM, int BN, int NST>
__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(const F16Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_INSTR = BM / 16 / 4, B_INSTR = BN / 16 / 4;  // global_load_lds per wave per K-step
  constexpr int PLANE_A = BM * 64, PLANE_B = BN * 64;          // bytes
  constexpr int STAGE = PLANE_A + PLANE_B;

  const clipfs_gemm_args& g = p.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int Mend = p.m_end, N = g.N, K = g.K;

  const int tile = xcd_contiguous_unit();
  constexpr int GM = 8;
  const int nbn = p.n_blocks_n;
  const int grp = tile / (GM * nbn);
  const int rem = tile - grp * (GM * nbn);
  const int mb_total = (Mend - p.m_begin + BM - 1) / BM;
  const int gmm = min(GM, mb_total - grp * GM);
  const int m0 = p.m_begin + (grp * GM + rem % gmm) * BM;
  const int n0 = (rem / gmm) * BN;

  // ---- staging: instruction q of an operand covers 16 rows x 64 B ---------------------------------------
  const char* a_src[A_INSTR];
  const char* b_src[B_INSTR];
  int a_off[A_INSTR], b_off[B_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int r0 = 16 * (wave + 4 * i), row = r0 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
    a_src[i] = reinterpret_cast<const char*>(p.A16 + (size_t)min(m0 + row, Mend - 1) * g.lda + 8 * c);
    a_off[i] = r0 * 64;
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    const int r0 = 16 * (wave + 4 * i), row = r0 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
    b_src[i] = reinterpret_cast<const char*>(p.B16 + (size_t)min(n0 + row, N - 1) * g.ldb + 8 * c);
    b_off[i] = PLANE_A + r0 * 64;
  }
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
    b_frag[t] = PLANE_A + row * 64;
    b_swz[t] = (row >> 2) & 3;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto stage_in = [&](int kt, int stage) __attribute__((always_inline)) {
    char* s = smem_raw + stage * STAGE;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) glds16(a_src[i] + (size_t)kt * (BK * 2), s + a_off[i]);
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) glds16(b_src[i] + (size_t)kt * (BK * 2), s + b_off[i]);
  };
  // All twelve fragment reads of a K-step are issued before its first MFMA (48 VGPRs): one exposed LDS latency per
  // K-step instead of one per group of four MFMAs; the second half's reads complete under the first half's MFMAs.
  auto compute = [&](const char* s) __attribute__((always_inline)) {
    f16x8 av[2][TM], bv[2][TN];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int t = 0; t < TM; ++t)
        av[kb][t] = *reinterpret_cast<const f16x8*>(s + a_frag[t] + (((2 * kb + fh) ^ a_swz[t]) << 4));
#pragma unroll
      for (int t = 0; t < TN; ++t)
        bv[kb][t] = *reinterpret_cast<const f16x8*>(s + b_frag[t] + (((2 * kb + fh) ^ b_swz[t]) << 4));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[kb][i], bv[kb][j], acc[i][j], 0, 0, 0);
  };

  // NST = 3 LDS stages, loads issued two K-steps ahead (a K-step is ~1k cycles of MFMA per SIMD, HBM/L2 latency is
  // of that order): at step kt wait until this wave's loads of stage kt have landed (the newer stage may still be
  // in flight), barrier (everyone's stage kt is in LDS and everyone is done reading stage kt-1), refill the stage
  // kt-1 occupied with step kt+2, compute.  NST = 2 (24 KiB at 64 x 128: fits beside a 128 KiB ping-pong workgroup,
  // see gemm_f16_dispatch): one K-step ahead, vmcnt(0) at the barrier.
  constexpr int PER_STAGE = A_INSTR + B_INSTR;  // vmcnt events per wave per stage
  const int nk = K / BK;
  stage_in(0, 0);
  if (NST >= 3 && nk > 1) stage_in(1, 1);
  int cur = 0, nxt = NST - 1;
  for (int kt = 0; kt < nk; ++kt) {
    if (NST >= 3 && kt + 1 < nk)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NST - 1 < nk) stage_in(kt + NST - 1, nxt);
    compute(smem_raw + cur * STAGE);
    cur = cur == NST - 1 ? 0 : cur + 1;
    nxt = nxt == NST - 1 ? 0 : nxt + 1;
  }

  f16_finish<TM, TN>(p, acc, m0 + wm * (BM / 2), n0 + wn * (BN / 2), n0, m0 + BM <= Mend && n0 + BN <= N, lane);
}

// Row-major epilogue of the 256 x 256 kernel.  The MFMA result layout gives a lane ONE column and 16 rows of each
// 32 x 32 tile, so a register-side epilogue moves 2 - 4 bytes per lane and instruction (128 values x up to four tensors
// per lane: at K = 1024 it took longer than the K loop).  Here the raw accumulators of 128 tile rows at a time go
// through the LDS ring (128 rows x 256 columns x 4 B = its 128 KiB exactly) and all 512 threads then walk the half tile
// row-major, 4 columns per thread: bias, saved pre-activation, residual, fp32 C and f16 C16 are all 8 / 16-byte accesses
// of whole rows (a wave covers one 1 KiB row segment per instruction).
// T16: the accumulators are 8 x 4 tiles of v_mfma_f32_16x16x32_f16 (lane: column l & 15, rows 4 (l >> 4) + r); a store
// instruction then covers four row groups 4 rows apart, which all fall on the same 16 banks of a 1 KiB-pitch image -- the
// 16-float column blocks are XORed with the row group ((row >> 2) & 3) on both sides.
#ifdef CLIPFS_STAMPS
__device__ unsigned long long clipfs_f16_stamps[8192 * 10];
#define F16_STAMP(i)                                                                                  \
  do {                                                                                                \
    if (threadIdx.x == 0 && blockIdx.x < 8192) {                                                      \
      clipfs_f16_stamps[blockIdx.x * 10 + (i)] = __builtin_amdgcn_s_memtime();                         \
      if ((i) == 1) clipfs_f16_stamps[blockIdx.x * 10 + 4] = __builtin_amdgcn_s_memrealtime();        \
      if ((i) == 2) clipfs_f16_stamps[blockIdx.x * 10 + 5] = __builtin_amdgcn_s_memrealtime();        \
    }                                                                                                 \
  } while (0)
}  // namespace clipfs
extern "C" int clipfs_debug_read_f16_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(clipfs::clipfs_f16_stamps), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
namespace clipfs {
#else
#define F16_STAMP(i) do {} while (0)
#endif

// Epilogue of the 256 x 256 kernels through LDS: the accumulators are transposed through the (now idle) ring in two
// passes of 128 tile rows, then every thread owns 4 consecutive columns of 16 rows of the pass and does row-major
// 16-byte accesses.
// Round 3: in-kernel stamps (scripts/gemm_f16_stamps.py) showed the round-2 form of this epilogue taking 37 % (f16-only
// result) to 66 % (fp32 result + residual) of a K = 1024 tile's lifetime.  Waits, not bandwidth:
//   * `__syncthreads()` is a workgroup-scope fence + barrier and the fence is `s_waitcnt vmcnt(0)`: the second pass's
//     staging waited until EVERY global store of the first pass had been acknowledged by the memory system;
//   * the per-row loads (residual, saved pre-activation) were issued one row at a time behind the previous row's
//     store, and vmcnt retires in issue order, so each load also waited for that store: 32 store round trips per tile.
// Now (a) the staging barriers are raw `s_barrier`s behind `s_waitcnt lgkmcnt(0)` (only LDS traffic has to be ordered);
// (b) a pass is the UPPER or LOWER 64 rows of every wave's 128-row tile (not one wave row), so all eight waves stage,
// and after pass 0 every wave has freed half of its accumulators: (c) that room holds ALL per-row loads of a pass at
// once -- pass 0's are issued before its staging, pass 1's right after it, i.e. before any store of the tile, so no
// wait for a load ever includes a store; (d) the bias (a thread's columns never change) is loaded once.  Same
// arithmetic, same order per element: bitwise the same results.  Measured (stamps, K = 1024 tiles): 27.5 -> 15.6 k cycles
// (f16-only result), 97 -> 44 k (fp32 result + residual: 512 KB per tile at 11.6 B/cycle/CU = this CU's share of HBM),
// 67 -> 31 k (f16 result x saved pre-activation); cfg-5 step 110.4 -> 102.8 ms.  Tried and dropped: starting the first
// round's workgroups 0-3 delay steps apart so that the CUs' epilogue bursts do not coincide (no change up to 8 k
// cycles per step: the epilogue is bound per CU, not by a chip-wide burst).
template <bool T16, bool WIDE, class AccT>
__device__ __forceinline__ void f16_epilogue_lds(const F16Params& p, AccT& acc, float* lds, int m0, int n0, int wm, int wn,
                                                 int tid) {
  const clipfs_gemm_args& g = p.a;
  const int lane = tid & 63, fr = lane & 31, fh = lane >> 5;
  const int Mend = p.m_end, N = g.N, ldc = g.ldc;
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  constexpr int RP = 16;                   // rows per thread and pass
  const int c4 = tid & 63, rr = tid >> 6;  // this thread: columns n .. n + 3 of staging rows rr + 8 k
  const int n = n0 + 4 * c4;
  const bool n_ok = n < N;                 // N % 4 == 0 (host): the 4 columns are inside together
  const int nc = n_ok ? n : 0;             // clamped column for the (unused) loads of an outside thread
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (g.bias) b4 = *reinterpret_cast<const f32x4*>(g.bias + nc);
  const bool has_res = g.residual != nullptr, act2 = g.act == 2;
  const _Float16* aux16 = reinterpret_cast<const _Float16*>(g.aux_in);
  // staging row lr (0 .. 127) of pass ps <-> tile row (lr >> 6) * 128 + ps * 64 + (lr & 63)
  auto tile_row = [&](int ps, int lr) { return (lr >> 6) * 128 + ps * 64 + (lr & 63); };

  // One load buffer per pass: the residual rows (fp32 x 4) OR the saved pre-activation (f16 x 4 in the low two dwords).
  // The host never sends both, nor an fp32 pre-activation, to this epilogue (launch_f16_pp: register-epilogue kernel).
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x4 ld0[RP], ld1[RP];
  auto issue_loads = [&](int ps, f32x4 (&r)[RP]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < RP; ++k) {
      const int m = min(m0 + tile_row(ps, rr + 8 * k), Mend - 1);  // clamped: rows past the end are never stored
      if (has_res) {
        r[k] = *reinterpret_cast<const f32x4*>(g.residual + (size_t)m * g.ldres + nc);
      } else {
        const f32x2 h = *reinterpret_cast<const f32x2*>(aux16 + (size_t)m * ldc + nc);
        r[k][0] = h[0];
        r[k][1] = h[1];
      }
    }
  };
  auto stage = [&](int ps) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS-only ordering: global stores are NOT drained
    __builtin_amdgcn_s_barrier();                        // the ring (or the previous pass's image) is no longer read
    asm volatile("" ::: "memory");
    if constexpr (T16) {
      const int li = lane & 15, lg = lane >> 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            lds[(wm * 64 + i * 16 + 4 * lg + r) * 256 + ((wn * 64 + j * 16 + li) ^ (16 * lg))] = acc[ps * 4 + i][j][r];
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            lds[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * 256 + wn * 64 + j * 32 + fr] = acc[ps * 2 + i][j][r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto finish = [&](int ps, const f32x4 (&r)[RP]) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < RP; kb += 4) {
      f32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = rr + 8 * (kb + j);
        v[j] = *reinterpret_cast<const f32x4*>(lds + row * 256 + ((4 * c4) ^ (T16 ? 16 * ((row >> 2) & 3) : 0)));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m0 + tile_row(ps, rr + 8 * (kb + j));
        const bool ok = n_ok && m < Mend;
        f32x4 x = g.bias ? g.alpha * v[j] + b4 : g.alpha * v[j];
        const size_t o = (size_t)m * ldc + n;
        if (g.act == 1) {
          if (g.aux_out && ok)
            *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(g.aux_out) + o) =
                f16x4{(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3]};
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = quick_gelu_fast(x[e]);
        } else if (act2) {
          const f16x4 u = __builtin_bit_cast(f16x4, f32x2{r[kb + j][0], r[kb + j][1]});
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] *= quick_gelu_grad_fast((float)u[e]);
        }
        if (has_res) x += r[kb + j];
        if (ok) {
          if (g.C) *reinterpret_cast<f32x4*>(g.C + o) = x;
          if (p.C16) *reinterpret_cast<f16x4*>(p.C16 + o) = f16x4{(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3]};
        }
      }
    }
  };

  // ---- f16-only result (the QKV, c_fc and d(c_proj) products: 69 % of the tower's GEMM FLOPs): 8 columns per thread ----
  // The row loop is bound by the NUMBER of vector-memory instructions (stamps: ~40 cycles per wave-store whatever its
  // width), so a thread takes 8 consecutive columns of 8 rows per pass: one 16-byte store per row instead of two 8-byte
  // ones (and one 16-byte load of the saved pre-activation).  Same arithmetic per element.
  typedef _Float16 f16x8e __attribute__((ext_vector_type(8)));
  // (WIDE is chosen on the host -- f16_wide_epilogue_ok -- and is a template parameter so that the two paths do not
  // share a register allocation: as one runtime branch the fp32-result path spilled and ran 3x slower)
  if constexpr (WIDE) {
    constexpr int RW = 8;                      // rows per thread and pass
    const int c8 = tid & 31, r16 = tid >> 5;   // columns n8 .. n8 + 7 of staging rows r16 + 16 k
    const int n8 = n0 + 8 * c8;
    const bool n8_ok = n8 < N;
    const int n8c = n8_ok ? n8 : 0;
    f32x4 ba = {0.f, 0.f, 0.f, 0.f}, bb = ba;
    if (g.bias) {
      ba = *reinterpret_cast<const f32x4*>(g.bias + n8c);
      bb = *reinterpret_cast<const f32x4*>(g.bias + n8c + 4);
    }
    f32x4 lw0[RW], lw1[RW];  // act 2: eight halves of the saved pre-activation per row
    auto issue8 = [&](int ps, f32x4 (&r)[RW]) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < RW; ++k) {
        const int m = min(m0 + tile_row(ps, r16 + 16 * k), Mend - 1);
        r[k] = *reinterpret_cast<const f32x4*>(aux16 + (size_t)m * ldc + n8c);
      }
    };
    auto finish8 = [&](int ps, const f32x4 (&r)[RW]) __attribute__((always_inline)) {
#pragma unroll
      for (int kb = 0; kb < RW; kb += 2) {
        f32x4 va[2], vb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = r16 + 16 * (kb + j);
          const int sw = T16 ? 16 * ((row >> 2) & 3) : 0;
          va[j] = *reinterpret_cast<const f32x4*>(lds + row * 256 + ((8 * c8) ^ sw));
          vb[j] = *reinterpret_cast<const f32x4*>(lds + row * 256 + ((8 * c8 + 4) ^ sw));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int m = m0 + tile_row(ps, r16 + 16 * (kb + j));
          const bool ok = n8_ok && m < Mend;
          float x[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x[e] = g.bias ? g.alpha * va[j][e] + ba[e] : g.alpha * va[j][e];
            x[4 + e] = g.bias ? g.alpha * vb[j][e] + bb[e] : g.alpha * vb[j][e];
          }
          const size_t o = (size_t)m * ldc + n8;
          if (g.act == 1) {
            if (g.aux_out && ok) {
              f16x8e h;
#pragma unroll
              for (int e = 0; e < 8; ++e) h[e] = (_Float16)x[e];
              *reinterpret_cast<f16x8e*>(reinterpret_cast<_Float16*>(g.aux_out) + o) = h;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = quick_gelu_fast(x[e]);
          } else if (act2) {
            const f16x8e u = __builtin_bit_cast(f16x8e, r[kb + j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] *= quick_gelu_grad_fast((float)u[e]);
          }
          if (ok) {
            f16x8e h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (_Float16)x[e];
            *reinterpret_cast<f16x8e*>(p.C16 + o) = h;
          }
        }
      }
    };
    if (act2) issue8(0, lw0);
    stage(0);
    F16_STAMP(6);
    __builtin_amdgcn_sched_barrier(0);
    if (act2) issue8(1, lw1);
    __builtin_amdgcn_sched_barrier(0);
    finish8(0, lw0);
    F16_STAMP(7);
    __builtin_amdgcn_sched_barrier(0);
    stage(1);
    F16_STAMP(8);
    finish8(1, lw1);
    return;
  } else {
  const bool any_loads = has_res || act2;
  if (any_loads) issue_loads(0, ld0);  // in flight across the staging of pass 0
  stage(0);
  F16_STAMP(6);
  __builtin_amdgcn_sched_barrier(0);
  if (any_loads) issue_loads(1, ld1);  // half of the accumulators are dead now; still ahead of every store
  __builtin_amdgcn_sched_barrier(0);
  finish(0, ld0);
  F16_STAMP(7);
  __builtin_amdgcn_sched_barrier(0);
  stage(1);
  F16_STAMP(8);
  finish(1, ld1);
  }
}

// ---- 256 x 256 "ping-pong" kernel ---------------------------------------------------------------------------------
// The kernel above tops out where every two-barrier-per-K-step structure does (cdna_hip_programming.md section 5, "the
// step-3 structure's ceiling"): the two workgroups of a CU drift into the same phase, each wave's MFMAs wait for its
// own LDS reads, and the LDS-DMA is drained to vmcnt(0)-ish depth at every barrier.  This one follows the guide's
// recipe for getting past it -- ONE workgroup per CU whose prefetch stays in flight across barriers, counted vmcnt, raw
// barriers, and the two waves of every SIMD deliberately out of phase:
//   * block tile 256 x 256 x 32, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 (the same 128 accumulators per lane);
//     a third less LDS-DMA per MFMA than 256 x 128;
//   * LDS ring of FOUR stages (4 x 32 KiB = 128 KiB), LDS-DMA issued THREE K-steps ahead (~3 x 1024 cycles: an HBM
//     miss lands in time), retired with `s_waitcnt vmcnt(8)` -- two younger stages stay in flight;
//   * per K-step every wave runs LOADS (issue the DMA of step p+3, read ALL fragments of step p, retire its DMA of
//     step p+1, lgkmcnt(0)) | barrier | 16 MFMAs at raised priority | barrier.  Wave group 1 (the second wave of each
//     SIMD) executes one extra barrier up front, so it is permanently one barrier behind: while group 0 issues MFMAs,
//     group 1 is in LOADS, and vice versa -- the matrix pipe always has a wave whose operands are already in
//     registers.  Group 0 executes the matching extra barrier at the end.
// Hazards (global barrier numbers; group 0: A_p = 2p+1, B_p = 2p+2; group 1: A_p = 2p+2, B_p = 2p+3):
//   RAW  stage p+1 is read in LOADS_{p+1}.  Every wave retires ITS share of stage p+1 (vmcnt) inside LOADS_p, i.e.
//        before its A_p <= #2p+2; group 0 reads after B_p = #2p+2, group 1 after #2p+3.
//   WAR  ring slot (p+3) % 4 = (p-1) % 4 is overwritten by DMA issued in LOADS_p, i.e. after B_{p-1} >= #2p; all reads
//        of stage p-1 were complete (lgkmcnt(0)) before the reader's A_{p-1} <= #2p.
template <bool RM>
__global__ __launch_bounds__(512, 2) void gemm_f16_pp_kernel(const F16Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int BM = 256, BN = 256, TM = 4, TN = 2;  // four LDS stages
  constexpr int PLANE = 256 * 64;     // bytes of one operand plane of a stage
  constexpr int STAGE = 2 * PLANE;    // 32 KiB
  constexpr int PER_STAGE = 4;        // LDS-DMA instructions per wave per stage (2 for A, 2 for B)

  const clipfs_gemm_args& g = p.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wm is also the ping-pong group: waves 0-3 / 4-7 = first / second wave of a SIMD
  const int Mend = p.m_end, N = g.N, K = g.K;

  const int tile = xcd_contiguous_unit();
  constexpr int GM = 4;
  const int nbn = p.n_blocks_n;
  const int grp = tile / (GM * nbn);
  const int rem = tile - grp * (GM * nbn);
  const int mb_total = (Mend - p.m_begin + BM - 1) / BM;
  const int gmm = min(GM, mb_total - grp * GM);
  const int m0 = p.m_begin + (grp * GM + rem % gmm) * BM;
  const int n0 = (rem / gmm) * BN;

  // staging: instruction i of an operand covers rows 16 (wave + 8 i) .. + 15, 64 B each
  const char* a_src[2];
  const char* b_src[2];
  int st_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r0 = 16 * (wave + 8 * i), row = r0 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
    a_src[i] = reinterpret_cast<const char*>(p.A16 + (size_t)min(m0 + row, Mend - 1) * g.lda + 8 * c);
    b_src[i] = reinterpret_cast<const char*>(p.B16 + (size_t)min(n0 + row, N - 1) * g.ldb + 8 * c);
    st_off[i] = r0 * 64;
  }
  const int fr = lane & 31, fh = lane >> 5;
  int a_frag[TM], b_frag[TN], a_swz[TM], b_swz[TN];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int row = wm * 128 + t * 32 + fr;
    a_frag[t] = row * 64;
    a_swz[t] = (row >> 2) & 3;
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int row = wn * 64 + t * 32 + fr;
    b_frag[t] = PLANE + row * 64;
    b_swz[t] = (row >> 2) & 3;
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto stage_in = [&](int kt, int slot) __attribute__((always_inline)) {
    char* s = smem_raw + slot * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(a_src[i] + (size_t)kt * (BK * 2), s + st_off[i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(b_src[i] + (size_t)kt * (BK * 2), s + PLANE + st_off[i]);
  };

  const int nk = K / BK;
  // prologue: three stages in flight, stage 0 landed and published
  stage_in(0, 0);
  if (nk > 1) stage_in(1, 1);
  if (nk > 2) stage_in(2, 2);
  if (nk > 2)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_STAGE) : "memory");
  else if (nk > 1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                 // #0: stage 0 readable by everyone
  asm volatile("" ::: "memory");
  if (wm == 1) __builtin_amdgcn_s_barrier();    // group 1 falls one barrier behind (pairs with group 0's A_0)

  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // ---- LOADS ----
    if (kt + 3 < nk) stage_in(kt + 3, (slot + 3) & 3);  // into the slot step kt-1 used: released at #2 kt
    const char* s = smem_raw + slot * STAGE;
    f16x8 av[2][TM], bv[2][TN];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int t = 0; t < TN; ++t) bv[kb][t] = *reinterpret_cast<const f16x8*>(s + b_frag[t] + (((2 * kb + fh) ^ b_swz[t]) << 4));
#pragma unroll
      for (int t = 0; t < TM; ++t) av[kb][t] = *reinterpret_cast<const f16x8*>(s + a_frag[t] + (((2 * kb + fh) ^ a_swz[t]) << 4));
    }
    // retire this wave's share of step kt+1; stages kt+2 and kt+3 (where issued) stay in flight
    const int younger = min(2, nk - 2 - kt);  // stages issued after kt+1
    if (younger >= 2)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PER_STAGE) : "memory");
    else if (younger == 1)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PER_STAGE) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // A_kt
    asm volatile("" ::: "memory");
    // ---- MFMA ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[kb][i], bv[kb][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();  // B_kt
    asm volatile("" ::: "memory");
    slot = (slot + 1) & 3;
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();  // group 0 catches the extra barrier of group 1
  if constexpr (RM) {
    f16_lora_step<TM, TN>(p, acc, m0 + wm * 128, n0 + wn * 64, n0, lane);
    f16_epilogue_lds<false, false>(p, acc, reinterpret_cast<float*>(smem_raw), m0, n0, wm, wn, tid);
  } else {
    f16_finish<TM, TN>(p, acc, m0 + wm * 128, n0 + wn * 64, n0, m0 + BM <= Mend && n0 + BN <= N, lane);
  }
}

// ---- 256 x 256 x 64, four phases per K-tile -------------------------------------------------------------------------
// The ping-pong kernel above is the guide's "minimum 2-phase" form (one LOADS | MFMA pair of 16 MFMAs per 32-wide
// K-step).  This one is the finer schedule the guide measures ~1.3x faster (cdna_hip_programming.md, "The 256^2 8-phase
// template"): the same tile, wave grid (2 x 4, wave tile 128 x 64), 128 KiB of LDS and barrier-staggered wave groups, but
//   * K-tiles of 64, each computed as FOUR phases of 16 MFMAs (16x16x32; T16 = false: 8 of 32x32x16, the A/B aid) = one
//     64 x 32 quadrant of the wave tile over K = 64: quadrants (A0,B0) (A0,B1) (A1,B1) (A1,B0), A0/A1 = the wave's rows 0-63 / 64-127, B0/B1 = its columns
//     0-31 / 32-63, so a phase reads 12 / 4 / 8 / 0 fragments (ds_read_b128) and the other wave of the SIMD, half a
//     phase behind, always has a quadrant whose operands are in registers;
//   * the LDS ring is 2 K-tiles x 4 HALF-tiles of 16 KiB, a half-tile = one quadrant operand of ALL waves (A0: rows
//     wr*128 + [0,64) of both wave rows, B0: columns wc*64 + [0,32) of the four wave columns, ...), stored as
//     [2 k-planes][128 rows][64 B] with the 16-byte-chunk XOR (row >> 2) & 3 of the other f16 kernels (conflict free);
//   * ONE half-tile (2 global_load_lds per wave) is staged per phase, in the order the quadrants are read
//     (A0, B0, B1, A1), five phases ahead of its first read; every phase retires exactly the half-tile the NEXT phase
//     reads with a counted `s_waitcnt vmcnt(8)`: four half-tiles stay in flight across the barriers, vmcnt never
//     drains inside the loop.
// Phase G = 4 t + p (p = 1..4) of K-tile t:  reads {A0,B0 | B1 | A1 | -}(t);  stages sequence element G + 5
// ({B1,A1}(t+1), {A0,B0}(t+2));  waits until element G + 1 has landed;  barrier;  lgkmcnt(0);  the MFMAs;  barrier.
// Hazards (global barrier numbers: group 0 (waves 0-3) passes #2G+1 and #2G+2 in phase G, group 1 one later each):
//   RAW  element G+1 is retired by every wave before its first barrier of phase G (<= #2G+2) and read in phase G+1
//        (group 0 after #2G+2, group 1 after #2G+3).  {A0,B0}(t+1) are retired in phases (t,3) and (t,4).
//   WAR  a slot is restaged >= 2 phases after its last read: A0 / B0 read in (t,1), restaged in (t,3) / (t,4); B1 read
//        in (t,2), restaged in (t+1,1); A1 read in (t,3), restaged in (t+1,2).  A read of phase G is complete
//        (lgkmcnt(0)) before the reader's second barrier of that phase (<= #2G+3); the restage of phase G+2 is issued
//        after #2G+4.

template <bool T16, bool WIDE>
__global__ __launch_bounds__(512, 2) void gemm_f16_ph_kernel(const F16Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  F16_STAMP(0);
  constexpr int BM = 256, BN = 256;
  constexpr int HPLANE = 128 * 64;   // bytes of one k-plane (32 k) of a half-tile
  constexpr int HT = 2 * HPLANE;     // 16 KiB
  constexpr int KT_BYTES = 128;      // bytes of a K-tile along a row

  const clipfs_gemm_args& g = p.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wm is also the stagger group
  const int Mend = p.m_end, N = g.N, K = g.K;

  const int tile = xcd_contiguous_unit();
  constexpr int GM = 4;
  const int nbn = p.n_blocks_n;
  const int grp = tile / (GM * nbn);
  const int rem = tile - grp * (GM * nbn);
  const int mb_total = (Mend - p.m_begin + BM - 1) / BM;
  const int gmm = min(GM, mb_total - grp * GM);
  const int m0 = p.m_begin + (grp * GM + rem % gmm) * BM;
  const int n0 = (rem / gmm) * BN;

  // staging: a wave's instruction covers rows 16 wave .. + 15 of a half-tile plane (lane: row l / 4, chunk l % 4); the
  // swizzle is applied to the SOURCE chunk.  Half-tile types j: 0 = A0, 1 = B0, 2 = B1, 3 = A1.
  const char *src_a0, *src_b0, *src_b1, *src_a1;
  {
    const int r = 16 * wave + (lane >> 2);  // row of the half-tile
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    const int ar = m0 + (r >> 6) * 128 + (r & 63);
    const int br = n0 + (r >> 5) * 64 + (r & 31);
    src_a0 = reinterpret_cast<const char*>(p.A16 + (size_t)min(ar, Mend - 1) * g.lda + 8 * c);
    src_a1 = reinterpret_cast<const char*>(p.A16 + (size_t)min(ar + 64, Mend - 1) * g.lda + 8 * c);
    src_b0 = reinterpret_cast<const char*>(p.B16 + (size_t)min(br, N - 1) * g.ldb + 8 * c);
    src_b1 = reinterpret_cast<const char*>(p.B16 + (size_t)min(br + 32, N - 1) * g.ldb + 8 * c);
  }
  const int st_off = 16 * wave * 64;
  // element (K-tile u, type j) -> slot 4 (u & 1) + j
  auto stage = [&](int u, int j, const char* sp) __attribute__((always_inline)) {
    char* s = smem_raw + (4 * (u & 1) + j) * HT + st_off;
    const char* q = sp + (size_t)u * KT_BYTES;
    glds16(q, s);
    glds16(q + 64, s + HPLANE);
  };
  // fragment addresses inside a half-tile: row R, plane pl (32 k each), 16-byte chunk c of the row's 64 bytes.
  //   32x32x16: lane = row l & 31, K-step s (two per plane) -> chunk 2 s + (l >> 5);   tiles: A 2 per quadrant, B 1
  //   16x16x32: lane = row l & 15, one K-step per plane   -> chunk l >> 4;            tiles: A 4 per quadrant, B 2
  constexpr int NA = T16 ? 4 : 2, NB = T16 ? 2 : 1;  // tiles of a quadrant operand
  constexpr int TR = T16 ? 16 : 32;                  // tile rows
  const int fr = lane & (TR - 1), fh = lane / TR;    // fh: 0..1 (32-row tiles) / 0..3 (16-row tiles)
  int a_row[NA], a_swz[NA], b_row[NB], b_swz[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int R = wm * 64 + i * TR + fr;
    a_row[i] = R * 64;
    a_swz[i] = (R >> 2) & 3;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int R = wn * 32 + i * TR + fr;
    b_row[i] = R * 64;
    b_swz[i] = (R >> 2) & 3;
  }
  // fragment c of a tile: plane c >> 1 (32x32x16: K-step c & 1 of it) / plane c (16x16x32)
  constexpr int NC = T16 ? 2 : 4;
  auto read_a = [&](f16x8 (&f)[NA][NC], int u, int j) __attribute__((always_inline)) {
    const char* s = smem_raw + (4 * (u & 1) + j) * HT;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int pl = T16 ? c : (c >> 1), ch = T16 ? fh : 2 * (c & 1) + fh;
        f[i][c] = *reinterpret_cast<const f16x8*>(s + pl * HPLANE + a_row[i] + ((ch ^ a_swz[i]) << 4));
      }
  };
  auto read_b = [&](f16x8 (&f)[NB][NC], int u, int j) __attribute__((always_inline)) {
    const char* s = smem_raw + (4 * (u & 1) + j) * HT;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int pl = T16 ? c : (c >> 1), ch = T16 ? fh : 2 * (c & 1) + fh;
        f[i][c] = *reinterpret_cast<const f16x8*>(s + pl * HPLANE + b_row[i] + ((ch ^ b_swz[i]) << 4));
      }
  };

  typedef typename std::conditional<T16, f32x4, f32x16>::type acc_t;
  acc_t acc[2 * NA][2 * NB];  // [row tile][column tile] of the 128 x 64 wave tile
#pragma unroll
  for (int i = 0; i < 2 * NA; ++i)
#pragma unroll
    for (int j = 0; j < 2 * NB; ++j)
#pragma unroll
      for (int r = 0; r < (T16 ? 4 : 16); ++r) acc[i][j][r] = 0.f;
  f16x8 fa0[NA][NC], fa1[NA][NC], fb0[NB][NC], fb1[NB][NC];

  // one quadrant: rows I0 .., columns J0 .. (in tiles); 8 (32x32x16) or 16 (16x16x32) MFMAs, K-step outermost so that
  // consecutive MFMAs never share an accumulator
#define PH_SYNC_MFMA(VM, FA, FB, I0, J0)                                                          \
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");                                       \
  __builtin_amdgcn_s_barrier();                                                                   \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                              \
  __builtin_amdgcn_sched_barrier(0);                                                              \
  __builtin_amdgcn_s_setprio(1);                                                                  \
  _Pragma("unroll") for (int c = 0; c < NC; ++c) _Pragma("unroll") for (int i = 0; i < NA; ++i)   \
      _Pragma("unroll") for (int j = 0; j < NB; ++j) {                                            \
    if constexpr (T16)                                                                            \
      acc[I0 * NA + i][J0 * NB + j] =                                                             \
          __builtin_amdgcn_mfma_f32_16x16x32_f16(FA[i][c], FB[j][c], acc[I0 * NA + i][J0 * NB + j], 0, 0, 0); \
    else                                                                                          \
      acc[I0 * NA + i][J0 * NB + j] =                                                             \
          __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[i][c], FB[j][c], acc[I0 * NA + i][J0 * NB + j], 0, 0, 0); \
  }                                                                                               \
  __builtin_amdgcn_s_setprio(0);                                                                  \
  __builtin_amdgcn_sched_barrier(0);                                                              \
  __builtin_amdgcn_s_barrier();                                                                   \
  asm volatile("" ::: "memory");

  const int nk = K / 64;  // >= 2 (host)
  // prologue: K-tile 0 and {A0,B0}(1) in flight; {A0,B0}(0) landed and published
  stage(0, 0, src_a0);
  stage(0, 1, src_b0);
  stage(0, 2, src_b1);
  stage(0, 3, src_a1);
  stage(1, 0, src_a0);
  stage(1, 1, src_b0);
  // 16x16x32: the LoRA term goes into the accumulators FIRST, while no fragment registers are live (its 48 operand
  // registers beside 128 accumulators and 64 fragments spilled around the last K-tile when it came last).  Its loads are
  // younger than the LDS-DMA above, so waiting for them retires the whole prologue: the counted wait below then falls through.
  if constexpr (T16) f16_lora_step16(p, acc, m0 + wm * 128, n0 + wn * 64, n0, lane);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // #0
  asm volatile("" ::: "memory");
  if (wm == 1) __builtin_amdgcn_s_barrier();  // group 1 falls one barrier behind
  F16_STAMP(1);

  int t = 0;
  for (; t + 2 < nk; ++t) {  // steady state: every stage exists, four half-tiles stay in flight
    read_a(fa0, t, 0);
    read_b(fb0, t, 1);
    stage(t + 1, 2, src_b1);
    PH_SYNC_MFMA(8, fa0, fb0, 0, 0)
    read_b(fb1, t, 2);
    stage(t + 1, 3, src_a1);
    PH_SYNC_MFMA(8, fa0, fb1, 0, 1)
    read_a(fa1, t, 3);
    stage(t + 2, 0, src_a0);
    PH_SYNC_MFMA(8, fa1, fb1, 1, 1)
    stage(t + 2, 1, src_b0);
    PH_SYNC_MFMA(8, fa1, fb0, 1, 0)
  }
  {  // K-tile nk-2: its phases 3 and 4 have nothing left to stage
    read_a(fa0, t, 0);
    read_b(fb0, t, 1);
    stage(t + 1, 2, src_b1);
    PH_SYNC_MFMA(8, fa0, fb0, 0, 0)
    read_b(fb1, t, 2);
    stage(t + 1, 3, src_a1);
    PH_SYNC_MFMA(8, fa0, fb1, 0, 1)
    read_a(fa1, t, 3);
    PH_SYNC_MFMA(6, fa1, fb1, 1, 1)
    PH_SYNC_MFMA(4, fa1, fb0, 1, 0)
    ++t;
  }
  {  // K-tile nk-1
    read_a(fa0, t, 0);
    read_b(fb0, t, 1);
    PH_SYNC_MFMA(2, fa0, fb0, 0, 0)
    read_b(fb1, t, 2);
    PH_SYNC_MFMA(0, fa0, fb1, 0, 1)
    read_a(fa1, t, 3);
    PH_SYNC_MFMA(0, fa1, fb1, 1, 1)
    PH_SYNC_MFMA(0, fa1, fb0, 1, 0)
  }
#undef PH_SYNC_MFMA
  __builtin_amdgcn_sched_barrier(0);  // nothing of the epilogue (LoRA operand loads) is hoisted into the last phases
  if (wm == 0) __builtin_amdgcn_s_barrier();  // group 0 catches the extra barrier of group 1
  F16_STAMP(2);
  if constexpr (!T16) f16_lora_step<4, 2>(p, acc, m0 + wm * 128, n0 + wn * 64, n0, lane);
  f16_epilogue_lds<T16, WIDE>(p, acc, reinterpret_cast<float*>(smem_raw), m0, n0, wm, wn, tid);
  F16_STAMP(3);
}

static int launch_f16_pp(F16Params& p, hipStream_t stream) {
  p.n_blocks_n = (p.a.N + 255) / 256;
  {
    const clipfs_gemm_args& a = p.a;
    static const int cfg = getenv("CLIPFS_F16_EPILOGUE") ? atoi(getenv("CLIPFS_F16_EPILOGUE")) : 1;  // 0: register epilogue (A/B aid)
    auto al = [](const void* q, size_t n) { return q == nullptr || ((uintptr_t)q & (n - 1)) == 0; };
    p.row_major_epilogue = cfg != 0 && (a.N & 3) == 0 && (a.ldc & 3) == 0 && (!a.residual || (a.ldres & 3) == 0) &&
                           al(a.C, 16) && al(p.C16, 8) && al(a.bias, 16) && al(a.residual, 16) &&
                           al(a.aux_out, 8) && al(a.aux_in, 8) &&
                           // the LDS epilogue reads / writes the pre-activation as f16 only and keeps ONE per-row load
                           // buffer (residual or pre-activation): everything else goes to the register-epilogue kernel
                           ((!a.aux_out && a.act != 2) || a.aux_f16) && !(a.residual && a.act == 2);
  }
  const int mb = (p.m_end - p.m_begin + 255) / 256;
  const size_t lds = 4 * (size_t)(2 * 256 * 64);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_pp_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_pp_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_ph_kernel<true, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_ph_kernel<true, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_ph_kernel<false, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  static const int ph_cfg = getenv("CLIPFS_F16_PHASED") ? atoi(getenv("CLIPFS_F16_PHASED")) : 1;  // 0: the 2-phase kernel, 2: phased on 32x32x16 MFMAs (A/B aids)
  // MFMA shape of the phased kernel.  The board sits at its power cap under f16 MFMA load and holds a higher clock on
  // 16x16x32 than on 32x32x16 (MI355X_MICROARCH.md, DVFS give-back item 7): 1.58 vs 1.76 us per K-tile and round at
  // M = 32768, N = 4096 (scripts/ksweep_f16.py: K = 128 ... 4096 150.6 ... 915.3 us vs 154.1 ... 1000.8), so 16x16x32
  // is the default; 32x32x16 stays as CLIPFS_F16_PHASED=2.
  const bool phased = p.row_major_epilogue && ph_cfg != 0 && (p.a.K % 64) == 0 && p.a.K >= 128;
  const bool shape16 = ph_cfg != 2;
  // f16-only result without a residual: the 8-columns-per-thread epilogue (16-byte stores / pre-activation loads)
  const clipfs_gemm_args& ga = p.a;
  const bool wide = !ga.C && p.C16 && !ga.residual && (ga.N & 7) == 0 && (ga.ldc & 7) == 0 &&
                    ((reinterpret_cast<uintptr_t>(p.C16) | reinterpret_cast<uintptr_t>(ga.aux_out) |
                      reinterpret_cast<uintptr_t>(ga.aux_in) | reinterpret_cast<uintptr_t>(ga.bias)) & 15) == 0;
  if (phased && shape16 && wide)
    hipLaunchKernelGGL((gemm_f16_ph_kernel<true, true>), dim3(mb * p.n_blocks_n), dim3(512), lds, stream, p);
  else if (phased && shape16)
    hipLaunchKernelGGL((gemm_f16_ph_kernel<true, false>), dim3(mb * p.n_blocks_n), dim3(512), lds, stream, p);
  else if (phased)
    hipLaunchKernelGGL((gemm_f16_ph_kernel<false, false>), dim3(mb * p.n_blocks_n), dim3(512), lds, stream, p);
  else if (p.row_major_epilogue)
    hipLaunchKernelGGL(gemm_f16_pp_kernel<true>, dim3(mb * p.n_blocks_n), dim3(512), lds, stream, p);
  else
    hipLaunchKernelGGL(gemm_f16_pp_kernel<false>, dim3(mb * p.n_blocks_n), dim3(512), lds, stream, p);
  return launch_status();
}

template <int BM, int BN, int NST = 3>
static int launch_f16(F16Params& p, hipStream_t stream) {
  p.n_blocks_n = (p.a.N + BN - 1) / BN;
  const int mb = (p.m_end - p.m_begin + BM - 1) / BM;
  const size_t lds = NST * (size_t)(BM * 64 + BN * 64);
  static bool attr = false;
  if (!attr && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_kernel<BM, BN, NST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, NST>), dim3(mb * p.n_blocks_n), dim3(256), lds, stream, p);
  return launch_status();
}

// ---- leftover rows beside the ping-pong kernel ----------------------------------------------------------------
// The rows the 256 x 256 launch does not take (M mod 256, or a thin last round) are a launch of a few dozen small
// workgroups whose duration is one tile's K loop (15 - 40 us at K = 1024 ... 4096): behind the big launch on the
// same stream that is pure latency, ~290 times per cfg-5 step.  A ping-pong workgroup leaves 32 KiB of LDS and 96
// registers per lane and SIMD free, which is exactly a two-stage 64 x 128 workgroup: the leftover launch goes to a
// side stream forked off the caller's stream BEFORE the big launch and joined after it, so both grids are resident
// together and the small one finishes in the shadow of the big one.  One side stream + two events per caller stream,
// created on first use, per host thread.
#define F16_HIP(call)                                                          \
  do {                                                                         \
    hipError_t _e = (call);                                                    \
    if (_e != hipSuccess) {                                                    \
      set_error("gemm_f16: %s failed: %s", #call, hipGetErrorString(_e));      \
      return CLIPFS_HIPERR_BASE + (int)_e;                                     \
    }                                                                          \
  } while (0)

struct SideStream {
  hipStream_t stream;
  hipEvent_t fork, join;
};

static SideStream* side_stream_for(hipStream_t main) {
  static thread_local std::vector<std::pair<hipStream_t, SideStream>> table;
  for (auto& e : table)
    if (e.first == main) return &e.second;
  SideStream s;
  // highest priority: its own hardware queue (never the one the caller's stream is mapped to), and the few small
  // workgroups are dispatched as soon as their dependencies are met
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  static const int prio_cfg = getenv("CLIPFS_F16_SIDE_PRIO") ? atoi(getenv("CLIPFS_F16_SIDE_PRIO")) : 0;  // 0 high, 1 low, 2 normal
  const int prio = prio_cfg == 1 ? least : prio_cfg == 2 ? 0 : greatest;
  if (hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, prio) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess)
    return nullptr;
  table.emplace_back(main, s);
  return &table.back().second;
}

// rows [p.m_begin, p.m_end) on the 4-wave kernels: 256 x 128 tiles on the rows that fill whole rounds of 512 workgroup
// slots (2 per CU), the leftover rows in 64-row tiles behind them; small problems on small tiles
static int dispatch_rows_4wave(F16Params& p, int tile_cfg, hipStream_t stream) {
  const clipfs_gemm_args& a = p.a;
  const int M = p.m_end - p.m_begin;
  const bool lora_ok_256 = !a.lora_t || a.lora_seg_width % 128 == 0;
  if (tile_cfg == 2) return launch_f16<64, 128>(p, stream);
  if (tile_cfg == 1) return launch_f16<128, 128>(p, stream);
  if (tile_cfg == 3 && lora_ok_256) return launch_f16<256, 128>(p, stream);
  const int nbn = (a.N + 127) / 128;
  if ((long)((M + 255) / 256) * nbn < 512) {
    return ((long)((M + 127) / 128) * nbn >= 512) ? launch_f16<128, 128>(p, stream) : launch_f16<64, 128>(p, stream);
  }
  const int mb = (M + 255) / 256;
  const long tiles = (long)mb * nbn;
  const long over = tiles % 512;  // tiles beyond the last full round of 512 slots
  int peel_blocks = 0;
  if (tiles > 512 && over > 0 && over * 8 <= 512) peel_blocks = (int)((over + nbn - 1) / nbn);
  const int begin = p.m_begin, end = p.m_end;
  const int split = begin + (mb - peel_blocks) * 256;
  if (peel_blocks == 0 || split >= end || split <= begin) return launch_f16<256, 128>(p, stream);
  p.m_end = split;
  CLIPFS_CHECK((launch_f16<256, 128>(p, stream)));
  p.m_begin = split;
  p.m_end = end;
  return launch_f16<64, 128>(p, stream);
}

// called from clipfs_gemm_nt when args->A_f16 is set
int gemm_f16_dispatch(const clipfs_gemm_args& a, hipStream_t stream) {
  F16Params p;
  p.a = a;
  p.A16 = reinterpret_cast<const _Float16*>(a.A_f16);
  p.B16 = reinterpret_cast<const _Float16*>(a.B_planes);
  p.C16 = reinterpret_cast<_Float16*>(a.C_f16);
  p.m_begin = 0;
  p.m_end = a.M;
  p.row_major_epilogue = 0;
  // tuning aid CLIPFS_F16_TILE: 1 / 2 / 3 force the 128x128 / 64x128 / 256x128 4-wave kernel, 4 forces the 256x256
  // ping-pong kernel on every row, 5 disables it
  static const int tile_cfg = getenv("CLIPFS_F16_TILE") ? atoi(getenv("CLIPFS_F16_TILE")) : 0;
  const bool lora_ok_pp = !a.lora_t || a.lora_seg_width % 256 == 0;
  if (tile_cfg == 4 && lora_ok_pp) return launch_f16_pp(p, stream);
  if (tile_cfg >= 1 && tile_cfg <= 3) return dispatch_rows_4wave(p, tile_cfg, stream);
  // 256 x 256 ping-pong tiles (one workgroup per CU) on the rows that fill whole rounds over the CUs -- a last round is
  // accepted when it is at least 80 % full -- and the 4-wave kernels on what is left
  if (tile_cfg != 5 && lora_ok_pp && a.K >= 4 * BK) {
    static int cus = 0;
    if (!cus) {
      int dev = 0, v = 0;
      cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
             v > 0) ? v : 256;
    }
    const int nbn = (a.N + 255) / 256;
    const int mb = a.M / 256;  // whole 256-row blocks
    long tiles = (long)mb * nbn;
    int use_mb = mb;
    static const int fill = getenv("CLIPFS_F16_PP_FILL") ? atoi(getenv("CLIPFS_F16_PP_FILL")) : 30;  // percent
    if (tiles >= cus) {
      const long rem = tiles % cus;
      if (rem != 0 && rem * 100 < (long)cus * fill) use_mb = (int)((tiles - rem) / nbn);
    } else if (tiles * 100 < (long)cus * fill) {
      use_mb = 0;
    }
    if (use_mb > 0 && (long)use_mb * 256 * 5 >= (long)a.M * 3) {
      const int split = use_mb * 256;
      static const int side_cfg = getenv("CLIPFS_F16_SIDE") ? atoi(getenv("CLIPFS_F16_SIDE")) : 1;  // 0: same stream (A/B aid)
      SideStream* side = (side_cfg != 0 && split < a.M) ? side_stream_for(stream) : nullptr;
      if (side) {
        // leftover rows first, on the side stream (ordered after everything already queued on `stream`)
        F16_HIP(hipEventRecord(side->fork, stream));
        F16_HIP(hipStreamWaitEvent(side->stream, side->fork, 0));
        F16Params q = p;
        q.m_begin = split;
        q.m_end = a.M;
        const long small_tiles = (long)((a.M - split + 63) / 64) * ((a.N + 127) / 128);
        if (small_tiles <= 1024)
          CLIPFS_CHECK((launch_f16<64, 128, 2>(q, side->stream)));
        else
          CLIPFS_CHECK(dispatch_rows_4wave(q, 0, side->stream));
        F16_HIP(hipEventRecord(side->join, side->stream));
      }
      p.m_end = split;
      CLIPFS_CHECK(launch_f16_pp(p, stream));
      if (side) {
        F16_HIP(hipStreamWaitEvent(stream, side->join, 0));
        return CLIPFS_OK;
      }
      if (p.m_end >= a.M) return CLIPFS_OK;
      p.m_begin = p.m_end;
      p.m_end = a.M;
    }
  }
  return dispatch_rows_4wave(p, 0, stream);
}

}  // namespace clipfs
