// fp32 "NT" GEMM on the CDNA4 matrix cores:  C[M,N] = epi( alpha * A[M,K] * B[N,K]^T ).
//
// v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fma chain) -- the only MFMA that meets the
// 1e-3-on-100x-cosine-logits budget of the north star (SURVEY.md section 7 "hard parts").
//
// Block tile BM x BN x 32, 256 threads = 4 waves (2 x 2), wave tile (BM/2) x (BN/2) made of
// 32x32 MFMA tiles.  Both operands are K-contiguous in memory, so a staging thread moves 16-byte
// K-chunks: global -> registers (issued one K-step ahead) -> LDS (double buffered, one barrier
// per K-step).  LDS image per operand: [rows][32 floats] with the 16-byte chunk index XOR-swizzled
// by (row >> 1) & 7, which makes both the ds_write_b128 of the staging pass and the ds_read_b128 of
// the fragment reads bank-conflict free (read groups are 16 lanes of distinct rows; MI355X_MICROARCH
// "LDS").  A lane's 4 consecutive k values feed 4 successive MFMAs: MFMA e of k-group q multiplies
// k = 8q+e (lanes 0-31) and k = 8q+4+e (lanes 32-63) -- any pairing is legal as long as A and B agree.
//
// Ragged M and N are handled by clamping the staged row index and predicating the stores; a K tail
// (K % 32 != 0) is zero filled.  Epilogue order is documented in include/clipfs.h.
#include "common.h"
#include "gemm_common.h"

#include <stdlib.h>

#include <vector>

// Timing-only ablations of the K loop (skip the prefetch / the LDS stores / the barrier / the fragment reads / the epilogue:
// WRONG results) exist only in a diagnostic build: `HIPCC_EXTRA=-DCLIPFS_ABLATION_BUILD python build.py --force`, then
// CLIPFS_GEMM_ABLATE=<bits> (scripts/ablate_gemm.py).  The shipped library compiles them out.
#ifdef CLIPFS_ABLATION_BUILD
#define CLIPFS_ABLATE(mask, bit) ((mask) & (bit))
#else
#define CLIPFS_ABLATE(mask, bit) 0
#endif

#ifdef CLIPFS_STAMPS
// Diagnostic build only (HIPCC_EXTRA=-DCLIPFS_STAMPS, build.py --tag stamps): s_memtime at four points of every
// workgroup of gemm_nt_kernel + the hardware ids of the CU it ran on, written to a buffer nothing else reads
// (scripts/gemm_stamps.py).
__device__ unsigned long long clipfs_gemm_stamps[16384 * 6];
#define CLIPFS_STAMP(i)                                                                                     \
  do {                                                                                                      \
    if (threadIdx.x == 0 && blockIdx.x < 16384) {                                                           \
      clipfs_gemm_stamps[blockIdx.x * 6 + (i)] = __builtin_amdgcn_s_memtime();                              \
      if ((i) == 1) clipfs_gemm_stamps[blockIdx.x * 6 + 4] = __builtin_amdgcn_s_memrealtime();              \
      if ((i) == 2) clipfs_gemm_stamps[blockIdx.x * 6 + 5] = __builtin_amdgcn_s_memrealtime();              \
      if (0) {                                                                                       \
        unsigned hw, xcc;                                                                                   \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                  \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                \
        clipfs_gemm_stamps[blockIdx.x * 6 + 4] = hw;                                                        \
        clipfs_gemm_stamps[blockIdx.x * 6 + 5] = xcc;                                                       \
      }                                                                                                     \
    }                                                                                                       \
  } while (0)
extern "C" int clipfs_debug_read_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(clipfs_gemm_stamps), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#else
#define CLIPFS_STAMP(i) do {} while (0)
#endif

namespace clipfs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // first-class vector: stays in VGPRs (HIP's float4 struct arrays went to scratch)
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

// 16-byte global -> LDS copy without a VGPR round trip (global_load_lds_dwordx4): the LDS destination is
// wave-uniform base + lane * 16, the global source is per lane -- so the XOR swizzle goes on the SOURCE.
__device__ __forceinline__ void glds16(const float* gsrc, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// AMODE 0: dense A, K % 32 == 0 (the hot configuration: the K loop is pointer bumps + 16-byte loads only)
// AMODE 1: patch im2col with patch == 32 (one K-step == one (channel, ky) image row segment)
// AMODE 2: generic (dense with a K tail, or any patch size): per-element address arithmetic
template <int BM, int BN, int AMODE>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WM = BM >= 64 ? 2 : 1;   // wave grid WM x WN (4 waves); BM = 32 puts the 4 waves side by side
  constexpr int WN = 4 / WM;
  constexpr int TM = BM / (32 * WM);     // 32x32 tiles per wave along M
  constexpr int TN = BN / (32 * WN);     // ... along N
  constexpr int A_CHUNKS = BM * 8 / 256; // 16-byte chunks staged per thread
  constexpr int B_CHUNKS = BN * 8 / 256;
  constexpr int STAGE_FLOATS = (BM + BN) * BK;

  const clipfs_gemm_args& g = p.a;
  CLIPFS_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // Tile order.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD, speed only),
  // each XCD has a private 4 MB L2: give every XCD a CONTIGUOUS run of the tile list, and order the list in
  // super-tiles of GM m-blocks x all n-blocks (m fastest) so the ~96 tiles an XCD runs at once share
  // 8 A panels and 12 B panels instead of ~40 + 18 (FETCH_SIZE: DESIGN.md section 8; GM = 8 fetches 10-25 % less than
  // 16 on four of the five path shapes, 4 and 32 more; the time does not move: the kernel is MFMA-bound).
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, li = bid >> 3, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + li;
  }
  const int split = tile % p.splits;  // consecutive units = the K slices of one tile (same XCD: shared panels)
  tile /= p.splits;
  const int GM = p.gm;
  const int nbn = p.n_blocks_n;
  const int grp = tile / (GM * nbn);
  const int rem = tile - grp * (GM * nbn);
  const int mb_total = (g.M + BM - 1) / BM;
  const int gm = min(GM, mb_total - grp * GM);  // the last group may be shorter
  const int m0 = (grp * GM + rem % gm) * BM;
  const int n0 = (rem / gm) * BN;
  const int M = g.M, N = g.N, K = g.K;

  // ---- staging addresses -------------------------------------------------------------------
  const float* a_src[A_CHUNKS];
  int a_lds[A_CHUNKS];
  const int kc = tid & 7;  // chunk (4 floats) inside the 32-wide K-step, same for every chunk id
#pragma unroll
  for (int i = 0; i < A_CHUNKS; ++i) {
    const int row = (tid >> 3) + 32 * i;
    const int m = min(m0 + row, M - 1);
    if (AMODE == 0 || AMODE == 3 || (AMODE == 2 && g.a_mode == 0)) {
      a_src[i] = g.A + (size_t)m * g.lda + (AMODE == 0 ? kc * 4 : 0);
    } else {
      const int b = m / p.patches, pp = m - b * p.patches;
      const int py = pp / p.grid_g, px = pp - py * p.grid_g;
      a_src[i] = g.A + ((size_t)b * 3 * g.img_res + (size_t)py * g.patch) * g.img_res + (size_t)px * g.patch +
                 (AMODE == 1 ? kc * 4 : 0);
    }
    a_lds[i] = row * BK + ((kc ^ ((row >> 1) & 7)) << 2);
  }
  const float* b_src[B_CHUNKS];
  int b_lds[B_CHUNKS];
#pragma unroll
  for (int i = 0; i < B_CHUNKS; ++i) {
    const int row = (tid >> 3) + 32 * i;
    const int n = min(n0 + row, N - 1);
    b_src[i] = g.B + (size_t)n * g.ldb + (AMODE != 2 ? kc * 4 : 0);
    b_lds[i] = BM * BK + row * BK + ((kc ^ ((row >> 1) & 7)) << 2);
  }

  f32x4 a_reg[A_CHUNKS], b_reg[B_CHUNKS];
  auto load_global = [&](int kt) __attribute__((always_inline)) {
    if (AMODE == 0 || AMODE == 3) {
      const int k0 = kt * BK;
#pragma unroll
      for (int i = 0; i < A_CHUNKS; ++i) a_reg[i] = *reinterpret_cast<const f32x4*>(a_src[i] + k0);
#pragma unroll
      for (int i = 0; i < B_CHUNKS; ++i) b_reg[i] = *reinterpret_cast<const f32x4*>(b_src[i] + k0);
    } else if (AMODE == 1) {
      // patch == 32 == BK: K-step kt is image row (c = kt / 32, ky = kt % 32) of every patch
      const int k0 = kt * BK;
      const size_t off = ((size_t)(kt >> 5) * g.img_res + (kt & 31)) * g.img_res;
#pragma unroll
      for (int i = 0; i < A_CHUNKS; ++i) a_reg[i] = *reinterpret_cast<const f32x4*>(a_src[i] + off);
#pragma unroll
      for (int i = 0; i < B_CHUNKS; ++i) b_reg[i] = *reinterpret_cast<const f32x4*>(b_src[i] + k0);
    } else {
      const int k = kt * BK + kc * 4;
      const bool k_ok = k < K;  // K % 4 == 0 is checked on the host
      if (g.a_mode == 0) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i)
          a_reg[i] = k_ok ? *reinterpret_cast<const f32x4*>(a_src[i] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        const int pp2 = g.patch * g.patch;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
          float t[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kk = k + e;
            const int c2 = kk / pp2, r2 = kk - c2 * pp2;
            const int ky2 = r2 / g.patch, kx2 = r2 - ky2 * g.patch;
            t[e] = kk < K ? a_src[i][((size_t)c2 * g.img_res + ky2) * g.img_res + kx2] : 0.f;
          }
          a_reg[i] = f32x4{t[0], t[1], t[2], t[3]};
        }
      }
#pragma unroll
      for (int i = 0; i < B_CHUNKS; ++i)
        b_reg[i] = k_ok ? *reinterpret_cast<const f32x4*>(b_src[i] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_lds = [&](int stage) __attribute__((always_inline)) {
    float* s = smem + stage * STAGE_FLOATS;
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) *reinterpret_cast<f32x4*>(s + a_lds[i]) = a_reg[i];
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) *reinterpret_cast<f32x4*>(s + b_lds[i]) = b_reg[i];
  };

  // ---- AMODE 3: dense operands copied global -> LDS directly (no VGPR staging, no ds_write) -------------
  // one global_load_lds_dwordx4 covers 8 tile rows x 128 B = 1 KiB of the LDS image (lane l: row l / 8,
  // slot l % 8); wave w owns row groups w, w + 4, ...; the swizzle is applied to the per-lane SOURCE chunk.
  const float* ga_src[A_CHUNKS];
  const float* gb_src[B_CHUNKS];
  int ga_lds[A_CHUNKS], gb_lds[B_CHUNKS];
  if (AMODE == 3) {
    const int uw = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
      const int r0 = 8 * (uw + 4 * i), r = r0 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      ga_src[i] = g.A + (size_t)min(m0 + r, M - 1) * g.lda + 4 * c;
      ga_lds[i] = r0 * BK;
    }
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) {
      const int r0 = 8 * (uw + 4 * i), r = r0 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      gb_src[i] = g.B + (size_t)min(n0 + r, N - 1) * g.ldb + 4 * c;
      gb_lds[i] = BM * BK + r0 * BK;
    }
  }
  auto glds_stage = [&](int kt, int stage) __attribute__((always_inline)) {
    float* s = smem + stage * STAGE_FLOATS;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) glds16(ga_src[i] + k0, s + ga_lds[i]);
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) glds16(gb_src[i] + k0, s + gb_lds[i]);
  };

  // ---- fragment read addresses -----------------------------------------------------------------
  const int fr = lane & 31, fh = lane >> 5;
  const int swz = (fr >> 1) & 7;  // (row >> 1) & 7 : tile bases are multiples of 32
  int a_frag[TM], b_frag[TN];
#pragma unroll
  for (int t = 0; t < TM; ++t) a_frag[t] = (wm * (BM / WM) + t * 32 + fr) * BK;
#pragma unroll
  for (int t = 0; t < TN; ++t) b_frag[t] = BM * BK + (wn * (BN / WN) + t * 32 + fr) * BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk_all = (K + BK - 1) / BK;
  const int kt0 = (int)((long)split * nk_all / p.splits);
  const int nk = (int)((long)(split + 1) * nk_all / p.splits) - kt0;  // K-steps of this unit (>= 1: host guarantees)
  auto compute = [&](const float* s) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = (((2 * q + fh) ^ swz) << 2);
      f32x4 av[TM], bv[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) av[t] = *reinterpret_cast<const f32x4*>(s + a_frag[t] + ch);
#pragma unroll
      for (int t = 0; t < TN; ++t) bv[t] = *reinterpret_cast<const f32x4*>(s + b_frag[t] + ch);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float ae = av[i][e];
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float be = bv[j][e];
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, be, acc[i][j], 0, 0, 0);
          }
        }
      }
    }
  };
  if (AMODE == 3) {
    glds_stage(kt0, 0);
    __syncthreads();  // emits vmcnt(0) for the LDS-DMA in flight, then the barrier
    CLIPFS_STAMP(1);
    for (int kt = 0; kt + 1 < nk; ++kt) {
      glds_stage(kt0 + kt + 1, (kt + 1) & 1);  // the other stage was last read before the previous barrier
      compute(smem + (kt & 1) * STAGE_FLOATS);
      __syncthreads();
    }
    compute(smem + ((nk - 1) & 1) * STAGE_FLOATS);
  } else {
  // prologue, steady state (prefetch unconditionally: the staged registers must stay in VGPRs), tail
  load_global(kt0);
  store_lds(0);
  __syncthreads();
  for (int kt = 0; kt + 1 < nk; ++kt) {
    if (!CLIPFS_ABLATE(p.ablate, 1)) load_global(kt0 + kt + 1);  // global -> registers, one K-step ahead
    __builtin_amdgcn_sched_barrier(0);         // keep the loads ABOVE the MFMAs (hipcc sinks them to the ds_write otherwise)
    compute(smem + (kt & 1) * STAGE_FLOATS);   // 32 MFMAs per wave hide the load latency
    if (!CLIPFS_ABLATE(p.ablate, 2)) store_lds((kt + 1) & 1);  // registers -> the other LDS stage
    if (!CLIPFS_ABLATE(p.ablate, 4)) __syncthreads();
  }
  compute(smem + ((nk - 1) & 1) * STAGE_FLOATS);
  }

  CLIPFS_STAMP(2);
  // ---- LoRA up-projection on the matrix cores + epilogue (gemm_common.h) ------------------------------------
  if (p.splits > 1) {
    // Split-K without a combine launch: every K slice publishes its raw accumulators as a slab (register order: one
    // 16-byte store per lane, fully coalesced) WRITE-THROUGH (sc1), takes a ticket on the tile's arrival counter, and
    // the slice that arrives LAST adds the slabs in slice order -- bitwise reproducible whatever the arrival order --
    // and runs the same LoRA + epilogue code as an unsplit tile.  Nobody waits, so no residency assumption; every slab
    // load is an sc1 load (bypasses this CU's L1, which no other CU's store refreshes): no release / acquire fence
    // (same hand-off as the stream-K kernel below; MI355X_MICROARCH.md "Valid forms").  The counter is left zero.
    constexpr int SLAB = BM * BN;
    const int S = p.splits;
    {
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(p.part + ((size_t)tile * S + split) * SLAB, 0, SLAB * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const f32x4 val = f32x4{acc[i][j][4 * v], acc[i][j][4 * v + 1], acc[i][j][4 * v + 2], acc[i][j][4 * v + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, val), rs,
                                                   (((i * TN + j) * 4 + v) * 256 + tid) * 16, 0, /*sc1*/ 16);
          }
    }
    int* flag = reinterpret_cast<int*>(smem + 2 * STAGE_FLOATS);  // one word behind the two stages (the host adds it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores ...
    __syncthreads();                                    // ... before ONE lane signals (also: everybody is done with the stages)
    if (tid == 0) {
      const int old = __hip_atomic_fetch_add(p.sk_cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = old == S - 1;
      if (last) __hip_atomic_store(p.sk_cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    if (S > 2) {  // (s0 + s1) + s2 + ...: every slab from memory, the own one included (it was stored write-through)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    for (int c = 0; c < S; ++c) {
      if (S == 2 && c == split) continue;  // two slices: a + b == b + a, the own slab stays in registers
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(p.part + ((size_t)tile * S + c) * SLAB, 0, SLAB * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x4 u[4];
#pragma unroll
          for (int v = 0; v < 4; ++v)
            u[v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                 rs, (((i * TN + j) * 4 + v) * 256 + tid) * 16, 0, /*sc1*/ 16));
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * v + e] += u[v][e];
        }
    }
  }
  finish_tiles<TM, TN>(g, p.patches, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), m0 + BM <= M && n0 + BN <= N, lane);
  CLIPFS_STAMP(3);
}

// ---- stream-K ------------------------------------------------------------------------------------------------
// The path's shapes put 2.3 - 9.4 tiles on each of the 512 resident workgroup slots, so a one-tile-per-workgroup grid
// ends in a mostly idle round (1200 tiles = 2.34 rounds run for 3: profiles/r01).  Here gridDim.x persistent
// workgroups ("runs", two per CU) share the work evenly:
//   * the LAST sk_tiles tiles (between one and two per run) form an iteration space tiles x K-steps that is cut into
//     gridDim.x equal contiguous pieces -- a piece covers the end of one tile and the start of the next;
//   * the tiles before them are dealt whole, round by round, in the same XCD-contiguous order as gemm_nt_kernel (the
//     64 runs of an XCD work on 64 neighbouring tiles: shared A / B panels in that XCD's L2).
// Whole tiles go through the fused epilogue directly.  A partial tile's raw accumulators go to a slab and the LAST
// contributor to arrive (agent-scope arrival counter; nobody ever waits, so no residency assumption) adds the slabs in
// run order -- bitwise reproducible whatever the arrival order -- and runs the same LoRA + epilogue code.  Slabs are
// stored write-through (sc1) and read with sc1 loads: no release / acquire fence (cdna_hip_programming.md section 5,
// "In-launch split-K reduction").  A run is ONE software pipeline: the first K-step of the next tile is fetched into
// the free LDS stage while the current tile runs its last K-step and its epilogue.
struct SkSeg {
  int tile, kb, ke;
};

#ifndef CLIPFS_SK_WAVES
#define CLIPFS_SK_WAVES 2  // waves per SIMD the stream-K kernel is compiled for (2: <= 256 registers)
#endif
template <int BM, int BN, int NSTAGE>
__global__ __launch_bounds__(256, CLIPFS_SK_WAVES) void gemm_sk_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WM = BM >= 64 ? 2 : 1;
  constexpr int WN = 4 / WM;
  constexpr int TM = BM / (32 * WM);
  constexpr int TN = BN / (32 * WN);
  constexpr int A_CHUNKS = BM * 8 / 256;
  constexpr int B_CHUNKS = BN * 8 / 256;
  constexpr int STAGE_FLOATS = (BM + BN) * BK;
  constexpr int SLAB_BYTES = BM * BN * 4;

  const clipfs_gemm_args& g = p.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wave / WN, wn = wave % WN;
  const int M = g.M, N = g.N;
  const int nk = p.sk_nk;
  const int runs = gridDim.x;
  const int w = xcd_contiguous_unit();  // runs that are neighbours in the tile list sit on one XCD (shared L2 panels)
  // NSTAGE LDS stages: the LDS-DMA runs NSTAGE - 1 K-steps ahead of the MFMAs (2 stages = one K-step = ~1.7 us of two
  // workgroups' MFMAs; a third stage measured no gain at 64 x 128 and would cost the 128 x 128 tile its second workgroup)
  int* flag = reinterpret_cast<int*>(smem + NSTAGE * STAGE_FLOATS);  // one word behind the stages

  // ---- this run's work list: its piece of the stream-K iteration space first, then its whole tiles ----
  const int q = p.sk_q, r = p.sk_r;                     // piece length q (+1 for the first r runs)
  const int sk_it0 = w * q + min(w, r);
  const int sk_it1 = sk_it0 + q + (w < r ? 1 : 0);
  // whole tiles: XCD x owns dp tiles [x * dpx, (x+1) * dpx), dpx = dp_rounds * (runs / 8) (host: runs % 8 == 0 when dp_rounds > 0)
  const int rpx = runs >> 3;
  const int xcd = rpx > 0 ? w / rpx : 0, li = rpx > 0 ? w - xcd * rpx : 0;
  struct Cursor {  // position in the work list; the loads run two K-steps ahead of the MFMAs, each side has its own
    int it, rd;
  };
  auto next_seg = [&](Cursor& c, SkSeg& sg) -> bool {
    if (c.it < sk_it1) {
      const int t = c.it / nk;
      sg.tile = p.sk_tile0 + t;
      sg.kb = c.it - t * nk;
      sg.ke = min(nk, sg.kb + (sk_it1 - c.it));
      c.it += sg.ke - sg.kb;
      return true;
    }
    if (c.rd < p.sk_dp_rounds) {
      sg.tile = (xcd * p.sk_dp_rounds + c.rd) * rpx + li;
      sg.kb = 0;
      sg.ke = nk;
      ++c.rd;
      return true;
    }
    return false;
  };
  // run index holding stream-K iteration x / first iteration of run c
  auto run_of = [&](int x) -> int { return x < r * (q + 1) ? x / (q + 1) : r + (x - r * (q + 1)) / q; };
  auto begin_of = [&](int c) -> int { return c * q + min(c, r); };

  const int fr = lane & 31, fh = lane >> 5;
  const int swz = (fr >> 1) & 7;
  int a_frag[TM], b_frag[TN];
#pragma unroll
  for (int t = 0; t < TM; ++t) a_frag[t] = (wm * (BM / WM) + t * 32 + fr) * BK;
#pragma unroll
  for (int t = 0; t < TN; ++t) b_frag[t] = BM * BK + (wn * (BN / WN) + t * 32 + fr) * BK;

  const int GM = p.gm, nbn = p.n_blocks_n;
  const int mb_total = (M + BM - 1) / BM;
  auto tile_origin = [&](int tile, int& m0, int& n0) {  // super-tiles of gm m-blocks x all n-blocks, m fastest
    const int grp = tile / (GM * nbn);
    const int rem = tile - grp * (GM * nbn);
    const int gmh = min(GM, mb_total - grp * GM);
    m0 = (grp * GM + rem % gmh) * BM;
    n0 = (rem / gmh) * BN;
  };
  const float* ga_src[A_CHUNKS];
  const float* gb_src[B_CHUNKS];
  auto set_sources = [&](int m0, int n0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
      const int rr = 8 * (uw + 4 * i) + (lane >> 3);
      const int c = (lane & 7) ^ ((rr >> 1) & 7);
      ga_src[i] = g.A + (size_t)min(m0 + rr, M - 1) * g.lda + 4 * c;
    }
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) {
      const int rr = 8 * (uw + 4 * i) + (lane >> 3);
      const int c = (lane & 7) ^ ((rr >> 1) & 7);
      gb_src[i] = g.B + (size_t)min(n0 + rr, N - 1) * g.ldb + 4 * c;
    }
  };
  auto glds_stage = [&](int kt, int stage) __attribute__((always_inline)) {
    float* s = smem + stage * STAGE_FLOATS;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) glds16(ga_src[i] + k0, s + 8 * (uw + 4 * i) * BK);
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) glds16(gb_src[i] + k0, s + BM * BK + 8 * (uw + 4 * i) * BK);
  };

  // ---- the K-step, software pipelined by hand --------------------------------------------------------------------
  // A K-step is four groups of 8 MFMAs (512 cycles of the SIMD's matrix pipe each); the fragments of group q+1 are read
  // from LDS right after the first MFMAs of group q were issued, so their latency runs in the shadow of that group
  // (hipcc left alone emits read -> wait -> 8 MFMAs, exposing ~100+ cycles per group whenever the SIMD's other wave is
  // in a prologue / epilogue / barrier).  The workgroup barrier sits INSIDE the last group: once a wave holds its last
  // fragments it has finished with the stage, waits for its own LDS-DMA of the next stage (issued a whole K-step ago),
  // passes the barrier and reads the NEXT K-step's first fragments while its remaining six MFMAs drain.
  struct Frag {
    f32x4 a[TM], b[TN];
  };
  auto read_frag = [&](Frag& f, const float* st, int qq) __attribute__((always_inline)) {
    const int ch = (((2 * qq + fh) ^ swz) << 2);
#pragma unroll
    for (int t = 0; t < TM; ++t) f.a[t] = *reinterpret_cast<const f32x4*>(st + a_frag[t] + ch);
#pragma unroll
    for (int t = 0; t < TN; ++t) f.b[t] = *reinterpret_cast<const f32x4*>(st + b_frag[t] + ch);
  };

  // ---- load side: walks the work list two K-steps ahead of the MFMA side ----
  Cursor lcur = {sk_it0, 0};
  SkSeg lseg;
  bool lhave = next_seg(lcur, lseg);
  if (!lhave) return;
  int lkt = lseg.kb, lstage = 0, issued = 0;
  {
    int lm0, ln0;
    tile_origin(lseg.tile, lm0, ln0);
    set_sources(lm0, ln0);
  }
  [[maybe_unused]] const int abl = p.ablate;
  auto issue_next = [&]() __attribute__((always_inline)) {  // LDS-DMA of the next K-step of the list (no-op at its end)
    if (!lhave) return;
    if (!CLIPFS_ABLATE(abl, 1) || issued < 2) glds_stage(lkt, lstage);
    lstage = lstage == NSTAGE - 1 ? 0 : lstage + 1;
    ++issued;
    if (++lkt == lseg.ke) {
      lhave = next_seg(lcur, lseg);
      if (lhave) {
        int lm0, ln0;
        tile_origin(lseg.tile, lm0, ln0);
        set_sources(lm0, ln0);
        lkt = lseg.kb;
      }
    }
  };

  // ---- MFMA side ----
  Cursor ccur = {sk_it0, 0};
  SkSeg cur;
  bool have = next_seg(ccur, cur);
  int cstage = 0, done = 0;  // stage holding the current K-step; K-steps finished
#pragma unroll
  for (int i = 0; i < NSTAGE - 1; ++i) issue_next();
  Frag f0, f1;
  if (NSTAGE >= 3 && issued > 1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * (A_CHUNKS + B_CHUNKS)) : "memory");  // the first K-step has landed (later ones may still travel)
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frag(f0, smem + cstage * STAGE_FLOATS, 0);
  while (have) {
    SkSeg nxt;
    const bool have_n = next_seg(ccur, nxt);
    int m0, n0;
    tile_origin(cur.tile, m0, n0);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) acc[i][j][rr] = 0.f;
    auto mfma_part = [&](const Frag& f, int e0, int e1) __attribute__((always_inline)) {
#pragma unroll
      for (int e = e0; e < e1; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][e], f.b[j][e], acc[i][j], 0, 0, 0);
    };
    const int n_steps = cur.ke - cur.kb;
    for (int kt = 0; kt < n_steps; ++kt) {
      // one K-step on stage `cstage`; on entry f0 = its group-0 fragments
      issue_next();  // K-step done + NSTAGE - 1 -> the stage released at the barrier inside the previous K-step
      const float* st = smem + cstage * STAGE_FLOATS;
      const int nstage = cstage == NSTAGE - 1 ? 0 : cstage + 1;
      __builtin_amdgcn_sched_barrier(0);
      mfma_part(f0, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (!CLIPFS_ABLATE(abl, 8)) read_frag(f1, st, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_part(f0, 1, 4);
      mfma_part(f1, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (!CLIPFS_ABLATE(abl, 8)) read_frag(f0, st, 2);
      __builtin_amdgcn_sched_barrier(0);
      mfma_part(f1, 1, 4);
      mfma_part(f0, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (!CLIPFS_ABLATE(abl, 8)) read_frag(f1, st, 3);
      __builtin_amdgcn_sched_barrier(0);
      mfma_part(f0, 1, 4);
      mfma_part(f1, 0, 1);  // needs the last fragments: this wave is done reading the stage
      __builtin_amdgcn_sched_barrier(0);
      ++done;
      if (CLIPFS_ABLATE(abl, 4)) {
        if (!CLIPFS_ABLATE(abl, 8)) read_frag(f0, smem + nstage * STAGE_FLOATS, 0);
      } else if (issued > done) {  // another K-step follows (this tile's or the next one's): barrier INSIDE the last group
        if (NSTAGE >= 3 && issued >= done + NSTAGE - 1)
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * (A_CHUNKS + B_CHUNKS)) : "memory");  // all but the youngest K-steps' LDS-DMA
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // everybody's share of the next stage has landed, everybody is done with this one
        asm volatile("" ::: "memory");
        read_frag(f0, smem + nstage * STAGE_FLOATS, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_part(f1, 1, 4);
      __builtin_amdgcn_sched_barrier(0);
      cstage = nstage;
    }
    const int mw = m0 + wm * (BM / WM), nw = n0 + wn * (BN / WN);
    const bool full_tile = m0 + BM <= M && n0 + BN <= N;

    if (CLIPFS_ABLATE(abl, 16)) {
      if (acc[0][0][0] == 123.456f) g.C[0] = 1.f;  // keep the accumulators alive
    } else if (cur.kb == 0 && cur.ke == nk) {
      finish_tiles<TM, TN>(g, p.patches, acc, mw, nw, full_tile, lane);
    } else {
      // ---- partial tile: publish the slab WRITE-THROUGH (sc1 stores: no release fence, which would write back the
      // whole XCD L2 -- full of everybody's freshly written C tiles -- once per slab), take a ticket ----
      const int t_lo_it = (cur.tile - p.sk_tile0) * nk, t_hi_it = t_lo_it + nk - 1;
      const int kind = sk_it0 > t_lo_it ? 0 : 1;  // the piece starts inside this tile (its head) / before it (its tail)
      {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(p.sk_part + ((size_t)(2 * w + kind)) * (BM * BN), 0, SLAB_BYTES,
                                                         0x00020000);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const f32x4 val = f32x4{acc[i][j][4 * v], acc[i][j][4 * v + 1], acc[i][j][4 * v + 2], acc[i][j][4 * v + 3]};
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, val), rs,
                                                     (((i * TN + j) * 4 + v) * 256 + tid) * 16, 0, /*sc1*/ 16);
            }
      }
      const int w_lo = run_of(t_lo_it), w_hi = run_of(t_hi_it);  // pieces contributing to this tile
      const int nc = w_hi - w_lo + 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores ...
      __syncthreads();                                    // ... before ONE lane signals
      if (tid == 0) {
        const int old = __hip_atomic_fetch_add(p.sk_cnt + cur.tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == nc - 1;
        if (last) __hip_atomic_store(p.sk_cnt + cur.tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero again for the next launch
        *flag = last;
      }
      __syncthreads();
      if (*flag) {
        // the reducer: slabs are added in run order, so the sum does not depend on who arrived last (two contributors,
        // the usual case: a + b == b + a, the own slab stays in registers).  EVERY slab load is an sc1 load (bypasses
        // this CU's L1, which no other CU's store ever refreshes), so no acquire fence is needed either.
        if (nc > 2) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int rr = 0; rr < 16; ++rr) acc[i][j][rr] = 0.f;
        }
        for (int c = w_lo; c <= w_hi; ++c) {
          if (nc == 2 && c == w) continue;
          const int ckind = begin_of(c) > t_lo_it ? 0 : 1;
          const auto rs = __builtin_amdgcn_make_buffer_rsrc(p.sk_part + ((size_t)(2 * c + ckind)) * (BM * BN), 0, SLAB_BYTES,
                                                           0x00020000);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              f32x4 u[4];  // one 32 x 32 accumulator tile at a time (register budget of the 128 x 128 kernel)
#pragma unroll
              for (int v = 0; v < 4; ++v)
                u[v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                     rs, (((i * TN + j) * 4 + v) * 256 + tid) * 16, 0, /*sc1*/ 16));
#pragma unroll
              for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * v + e] += u[v][e];
            }
        }
        finish_tiles<TM, TN>(g, p.patches, acc, mw, nw, full_tile, lane);
      }
    }
    cur = nxt;
    have = have_n;
  }
}

// ---- optional per-launch timing (bench.py roofline leg): HIP events on the launch stream around every
// GEMM while enabled.  Off by default; never on in the timed region.
struct TimedLaunch {
  hipEvent_t start, stop;
  double flops;
  double bytes;  // algorithmic HBM bytes: each operand, result and epilogue tensor touched once
};
static thread_local double g_last_bytes = 0.0;
static thread_local bool g_timing = false;
static thread_local std::vector<TimedLaunch>* g_timed = nullptr;

template <int BM, int BN, int AMODE>
static int launch(const GemmParams& p, hipStream_t stream) {
  const int mb = (p.a.M + BM - 1) / BM;
  // 8 KiB more than the two stages need: two workgroups per CU instead of three.  The K loop is as fast with two
  // (the MFMA pipe is the shared resource either way) and the third of the CU left free lets the other tower's
  // LayerNorm / attention / LoRA kernels run beside the GEMM: +1.3 % on the step, +3.4 % on the 8-GPU per-rank step.
  static const int lds_pad = getenv("CLIPFS_GEMM_LDS_PAD") ? atoi(getenv("CLIPFS_GEMM_LDS_PAD")) : 8;  // KiB
  const size_t lds = 2 * (size_t)(BM + BN) * BK * sizeof(float) + (size_t)lds_pad * 1024 + 64;  // + the split-K flag word
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<BM, BN, AMODE>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, AMODE>), dim3(mb * p.n_blocks_n * p.splits), dim3(256), lds, stream, p);
  return launch_status();
}

static int cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      n = v;
    else
      n = 256;  // MI355X
  }
  return n;
}

constexpr int SK_MIN_STEPS = 8;  // K-steps per run below which the slab hand-off would outweigh the balance it buys

// Stream-K configuration.  OPT-IN (CLIPFS_GEMM_SK=1 by shape, 2 always; default 0).  Measured on MI355X (64 x 128 tiles,
// 2 persistent workgroups per CU; scripts/ablate_gemm.py, same box): stand-alone, against one tile per workgroup, it gains
// 2-3 % where a tile has >= 24 K-steps and the grid ends in a partial round (K = 1536 ... 3072 shapes), 5-15 % on the
// per-rank shapes of the 8-GPU run (150 - 600 tiles), loses 1-9 % at K = 512 (the slab hand-off of a 13 us tile is not
// paid back) -- cfg-2 shape mix 76.2 % vs 75.4 % of the fp32 MFMA peak.  INSIDE the train step it loses 1.7 % (2831 vs
// 2882 img/s; per-rank batches: no difference): the two towers already run on two streams, so the idle CUs of one
// GEMM's last round are filled by the other tower's kernels, while a persistent grid that owns every CU keeps them out.
// 128 x 128 tiles (a third less LDS-DMA / LDS-read traffic per MFMA) and 3 workgroups per CU were both slower.
struct SkConfig {
  bool use;
  int per_cu;  // persistent workgroups per CU
};

static inline int sk_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

static inline SkConfig sk_config(int M, int N, int K) {
  static const int mode = sk_env("CLIPFS_GEMM_SK", 0);  // 0: never (one tile per workgroup), 1: by shape, 2: always
  static const int per_cu = sk_env("CLIPFS_GEMM_SK_PER_CU", 2);
  SkConfig c;
  c.per_cu = per_cu > 0 ? per_cu : 2;
  const long tiles = (long)((M + 63) / 64) * ((N + 127) / 128);
  const int nk = K / BK;
  const long slots = (long)cu_count() * c.per_cu;
  c.use = mode == 2 || (mode == 1 && (nk >= 24 || (tiles < 2 * slots && tiles * nk >= 2L * SK_MIN_STEPS * 64)));
  return c;
}

// Schedule of a stream-K launch over `tiles` tiles of `nk` K-steps: `runs` persistent workgroups; the last `sk_tiles`
// tiles (one to two per run) are cut evenly along tiles x K-steps, the others are dealt whole in `dp_rounds` rounds.
struct SkPlan {
  int runs, dp_rounds, tile0, sk_tiles, q, r;
};

static inline SkPlan sk_plan(long tiles, int nk, int per_cu) {
  SkPlan s;
  const long slots = (long)cu_count() * per_cu;
  long runs = tiles * nk / SK_MIN_STEPS;
  if (runs < tiles) runs = tiles;  // short K: never fewer runs than tiles (a run per tile is the plain schedule)
  if (runs > slots) runs = slots;
  if (runs > tiles * nk) runs = tiles * nk;
  s.runs = (int)(runs < 1 ? 1 : runs);
  s.dp_rounds = 0;
  if ((s.runs & 7) == 0 && tiles >= 2L * s.runs) s.dp_rounds = (int)(tiles / s.runs) - 1;
  s.tile0 = s.dp_rounds * s.runs;
  s.sk_tiles = (int)(tiles - s.tile0);
  const long iters = (long)s.sk_tiles * nk;
  s.q = (int)(iters / s.runs);
  s.r = (int)(iters % s.runs);
  return s;
}

template <int BM, int BN, int NSTAGE>
static int launch_sk(GemmParams& p, int per_cu, hipStream_t stream) {
  const int mb = (p.a.M + BM - 1) / BM;
  p.n_blocks_n = (p.a.N + BN - 1) / BN;
  p.sk_nk = p.a.K / BK;
  const SkPlan s = sk_plan((long)mb * p.n_blocks_n, p.sk_nk, per_cu);
  p.sk_tile0 = s.tile0;
  p.sk_dp_rounds = s.dp_rounds;
  p.sk_q = s.q;
  p.sk_r = s.r;
  p.sk_part = p.a.workspace;
  p.sk_cnt = p.a.counters;
  // the stages + the flag word, padded so that exactly `per_cu` workgroups fit a CU's 160 KiB (the rest of the CU stays
  // free for the other tower's LayerNorm / attention / LoRA kernels on the side stream)
  size_t lds = NSTAGE * (size_t)(BM + BN) * BK * sizeof(float) + 64;
  const size_t want = (size_t)(160 * 1024 / (per_cu + 1)) + 1024;
  if (lds < want && want * per_cu <= 160 * 1024) lds = want;
  static size_t attr_lds = 0;
  if (lds > attr_lds && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sk_kernel<BM, BN, NSTAGE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_lds = lds;
  }
  hipLaunchKernelGGL((gemm_sk_kernel<BM, BN, NSTAGE>), dim3(s.runs), dim3(256), lds, stream, p);
  return launch_status();
}

}  // namespace clipfs

using namespace clipfs;

// Tile height: 64 x 128, or 32 x 128 when that balances the 256 CUs better.  All tiles of a launch that fits the
// resident slots start together and share their CU's MFMA pipe, so a launch lasts ceil(tiles / 256) tile-times of
// the busiest CU: cost = ceil(tiles / 256) * rows-per-tile, with 8 % on top for the 32-row tile (it re-reads B
// twice as often).  Measured on the per-rank shapes of the strong-scaling run (scripts/bench_small_m.py): e.g. 600
// tiles of 64 rows (3 vs 2.34 per CU) lose 7-9 % against 1200 tiles of 32 rows, 450 tiles of 64 rows win by 9 %.
static inline int gemm_bm(int M, int N) {
  static const int force = getenv("CLIPFS_GEMM_BM") ? atoi(getenv("CLIPFS_GEMM_BM")) : 0;  // tuning aid: 32 / 64
  if (force == 32 || force == 64) return force;
  const long nbn = (N + 127) / 128;
  const long t64 = (long)((M + 63) / 64) * nbn, t32 = (long)((M + 31) / 32) * nbn;
  const double cost64 = (double)((t64 + 255) / 256) * 64.0, cost32 = (double)((t32 + 255) / 256) * 32.0 * 1.08;
  return cost32 < cost64 ? 32 : 64;
}

// Split-K factor for a [M,N,K] product: only when even the 32 x 128 tiling leaves the 512 resident workgroup slots
// short of work (strong-scaling per-rank batches, the one-row-per-sequence products of the last block) and K is long
// enough to cut.  The slices of a tile are combined by the last one to arrive inside the same launch (no second
// kernel), so the factor aims at ~one unit per slot, with at least 4 K-steps per unit.
extern "C" int clipfs_gemm_splits(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  static const int force = getenv("CLIPFS_GEMM_SPLITS") ? atoi(getenv("CLIPFS_GEMM_SPLITS")) : 0;  // tuning aid
  if (force > 0) return (K + BK - 1) / BK >= 2 * force ? force : 1;
  const int bm = gemm_bm(M, N);
  const long tiles = (long)((M + bm - 1) / bm) * ((N + 127) / 128);
  const int nk = (K + BK - 1) / BK;
  if (tiles >= 384 || nk < 8) return 1;
  long s = (448 + tiles - 1) / tiles;
  if (s > nk / 4) s = nk / 4;
  if (s > 8) s = 8;
  return s < 1 ? 1 : (int)s;
}

// tiles of the tiling gemm_nt_impl picks for a split product (0 when it does not split)
static inline long splitk_tiles(int M, int N, int K) {
  if (clipfs_gemm_splits(M, N, K) <= 1) return 0;
  const int bm = gemm_bm(M, N);
  return (long)((M + bm - 1) / bm) * ((N + 127) / 128);
}

static inline bool sk_enabled() {
  static const bool on = sk_env("CLIPFS_GEMM_SK", 0) != 0;
  return on;
}

// stream-K geometry of a dense [M,N,K] product: tiles (of the finer 64 x 128 tiling: bounds the counter array for either
// tile height), floats of slab space (a head and a tail slab per run)
static inline void sk_geometry(int M, int N, int K, long* tiles, size_t* slab_floats) {
  const SkConfig c = sk_config(M, N, K);
  *tiles = (long)((M + 63) / 64) * ((N + 127) / 128);
  *slab_floats = c.use ? (size_t)2 * sk_plan(*tiles, K / BK, c.per_cu).runs * 64 * 128 : 0;
}

extern "C" size_t clipfs_gemm_workspace_floats(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int s = clipfs_gemm_splits(M, N, K);
  // one slab of a whole tile (BM x 128 floats) per K slice and tile
  size_t need = s > 1 ? (size_t)s * (size_t)splitk_tiles(M, N, K) * gemm_bm(M, N) * 128 : 0;
  if (sk_enabled() && (K % BK) == 0) {  // stream-K: a head and a tail slab per run
    long tiles;
    size_t sk;
    sk_geometry(M, N, K, &tiles, &sk);
    need = sk > need ? sk : need;
  }
  return need;
}

extern "C" size_t clipfs_gemm_counter_ints(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  size_t need = (size_t)splitk_tiles(M, N, K);  // split-K: one arrival counter per tile
  if (sk_enabled() && (K % BK) == 0) {
    long tiles;
    size_t slab;
    sk_geometry(M, N, K, &tiles, &slab);
    if (slab && (size_t)tiles > need) need = (size_t)tiles;
  }
  return need;
}

extern "C" int clipfs_gemm_timing(int enable) {
  g_timing = enable != 0;
  return CLIPFS_OK;
}

extern "C" double clipfs_gemm_timing_last_bytes(void) { return g_last_bytes; }

extern "C" int clipfs_gemm_timing_collect(double* total_ms, double* total_flops, int* launches) {
  double ms = 0.0, fl = 0.0, by = 0.0;
  int n = 0;
  if (g_timed) {
    for (TimedLaunch& t : *g_timed) {
      float e = 0.f;
      if (hipEventSynchronize(t.stop) == hipSuccess && hipEventElapsedTime(&e, t.start, t.stop) == hipSuccess) {
        ms += e;
        fl += t.flops;
        by += t.bytes;
        ++n;
      }
      (void)hipEventDestroy(t.start);
      (void)hipEventDestroy(t.stop);
    }
    g_timed->clear();
  }
  g_last_bytes = by;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = n;
  return CLIPFS_OK;
}

namespace clipfs {
int gemm_bf16x3_dispatch(const GemmParams& base, hipStream_t stream);  // gemm_bf16.hip
int gemm_f16_dispatch(const clipfs_gemm_args& a, hipStream_t stream);  // gemm_f16.hip
}

static int gemm_nt_impl(const clipfs_gemm_args* args, void* stream);

extern "C" int clipfs_gemm_nt(const clipfs_gemm_args* args, void* stream) {
  if (!g_timing) return gemm_nt_impl(args, stream);
  TimedLaunch tl;
  (void)hipEventCreate(&tl.start);
  (void)hipEventCreate(&tl.stop);
  tl.flops = args ? 2.0 * args->M * (double)args->N * args->K : 0.0;
  tl.bytes = 0.0;
  if (args) {
    const double mk = (double)args->M * args->K, nk = (double)args->N * args->K, mn = (double)args->M * args->N;
    const double ab = args->A_f16 ? 2.0 : 4.0, bb = args->B_planes ? (args->b_format == 1 ? 4.0 : 2.0) : 4.0;
    tl.bytes = ab * mk + bb * nk + (args->C ? 4.0 : 0.0) * mn + (args->C_f16 ? 2.0 : 0.0) * mn +
               4.0 * mn * ((args->residual ? 1 : 0) + (args->aux_out && args->act == 1 ? 1 : 0) + (args->act == 2 ? 1 : 0));
  }
  (void)hipEventRecord(tl.start, (hipStream_t)stream);
  const int rc = gemm_nt_impl(args, stream);
  (void)hipEventRecord(tl.stop, (hipStream_t)stream);
  if (!g_timed) g_timed = new std::vector<TimedLaunch>();
  g_timed->push_back(tl);
  return rc;
}

static int gemm_nt_impl(const clipfs_gemm_args* args, void* stream) {
  CLIPFS_REQUIRE(args != nullptr, "gemm: null args");
  const clipfs_gemm_args& a = *args;
  CLIPFS_REQUIRE(a.struct_size == sizeof(clipfs_gemm_args),
                 "gemm: args built against another clipfs.h (struct_size %zu, library has %zu)", a.struct_size,
                 sizeof(clipfs_gemm_args));
  CLIPFS_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: bad dims %d %d %d", a.M, a.N, a.K);
  if (a.A_f16) {  // f16 x f16 kernel
    CLIPFS_REQUIRE(a.B_planes && a.b_format == 2, "gemm: A_f16 needs the f16 copy of B (b_format 2)");
    CLIPFS_REQUIRE(a.C || a.C_f16, "gemm: no output");
    CLIPFS_REQUIRE(a.a_mode == 0 && (a.K % BK) == 0 && (a.lda & 7) == 0 && (a.ldb & 7) == 0 && a.lda >= a.K && a.ldb >= a.K &&
                       a.ldc >= a.N && aligned16(a.A_f16) && aligned16(a.B_planes),
                   "gemm f16: K %% 32, lda/ldb %% 8, 16-byte aligned operands required");
    CLIPFS_REQUIRE(a.act >= 0 && a.act <= 2 && (a.act != 2 || a.aux_in) && (!a.residual || a.ldres >= a.N), "gemm f16: bad epilogue args");
    if (a.lora_t)
      CLIPFS_REQUIRE(a.lora_b && a.lora_r > 0 && a.lora_r <= 16 && a.lora_nseg > 0 && a.lora_seg_width % 128 == 0 &&
                         a.lora_seg_width * a.lora_nseg >= a.N, "gemm f16: lora rank <= 16 and segment width %% 128 required");
    return gemm_f16_dispatch(a, (hipStream_t)stream);
  }
  CLIPFS_REQUIRE(!a.C_f16 && !a.aux_f16, "gemm: C_f16 / aux_f16 belong to the f16 x f16 kernel only");
  CLIPFS_REQUIRE(a.A && a.B && a.C, "gemm: null operand");
  CLIPFS_REQUIRE((a.K & 3) == 0 && (a.ldb & 3) == 0 && aligned16(a.B), "gemm: K and ldb must be multiples of 4, B 16-byte aligned");
  CLIPFS_REQUIRE(a.ldb >= a.K && a.ldc >= a.N, "gemm: leading dimension too small");
  CLIPFS_REQUIRE(a.act >= 0 && a.act <= 3, "gemm: bad act %d", a.act);
  CLIPFS_REQUIRE(a.act != 3 || !a.B_planes, "gemm: act 3 (ReLU) belongs to the exact fp32 kernels");
  CLIPFS_REQUIRE(a.act != 2 || a.aux_in, "gemm: act 2 needs aux_in");
  CLIPFS_REQUIRE(!a.residual || a.ldres >= a.N, "gemm: ldres too small");
  GemmParams p;
  p.a = a;
  p.patches = 0;
#ifdef CLIPFS_ABLATION_BUILD
  static const int ablate_cfg = getenv("CLIPFS_GEMM_ABLATE") ? atoi(getenv("CLIPFS_GEMM_ABLATE")) : 0;
  p.ablate = ablate_cfg;
#else
  p.ablate = 0;
#endif
  p.grid_g = 0;
  static const int gm_cfg = getenv("CLIPFS_GEMM_GM") ? atoi(getenv("CLIPFS_GEMM_GM")) : 8;  // tuning aid
  p.gm = gm_cfg > 0 ? gm_cfg : 8;
  if (a.a_mode == 0) {
    CLIPFS_REQUIRE((a.lda & 3) == 0 && a.lda >= a.K && aligned16(a.A), "gemm: lda must be a multiple of 4 and >= K, A 16-byte aligned");
  } else {
    CLIPFS_REQUIRE(a.a_mode == 1, "gemm: bad a_mode %d", a.a_mode);
    CLIPFS_REQUIRE(a.patch > 0 && a.img_res % a.patch == 0, "gemm: image %d not divisible by patch %d", a.img_res, a.patch);
    p.grid_g = a.img_res / a.patch;
    p.patches = p.grid_g * p.grid_g;
    CLIPFS_REQUIRE(a.K == 3 * a.patch * a.patch, "gemm: patch mode needs K == 3*patch^2");
    CLIPFS_REQUIRE(a.M % p.patches == 0 && a.out_tokens >= p.patches + 1, "gemm: patch mode shape mismatch");
    CLIPFS_REQUIRE((a.patch & 3) != 0 || ((a.img_res & 3) == 0 && aligned16(a.A)), "gemm: image rows must be 16-byte aligned");
  }
  if (a.lora_t) {
    CLIPFS_REQUIRE(a.lora_b && a.lora_r > 0 && a.lora_nseg > 0 && a.lora_seg_width > 0, "gemm: bad lora args");
    CLIPFS_REQUIRE(a.lora_seg_width % 32 == 0 && a.lora_seg_width * a.lora_nseg >= a.N, "gemm: lora segment width must be a multiple of 32 and cover N");
  }
  hipStream_t s = (hipStream_t)stream;
  p.splits = 1;
  p.part = nullptr;
  p.n_blocks_n = 0;
  if (a.B_planes && (a.b_format == 1 || a.b_format == 2) && a.a_mode == 0 && (a.K % BK) == 0 && a.ldb == a.K)
    return gemm_bf16x3_dispatch(p, s);
  p.sk_cnt = nullptr;
  {
    // split-K needs the slab scratch AND the (caller-zeroed) arrival counters; dense fp32 operands only
    const int want = clipfs_gemm_splits(a.M, a.N, a.K);
    if (want > 1 && a.a_mode == 0 && (a.K % BK) == 0 && a.workspace && a.counters && aligned16(a.workspace) &&
        a.workspace_floats >= clipfs_gemm_workspace_floats(a.M, a.N, a.K) &&
        a.counters_ints >= (size_t)splitk_tiles(a.M, a.N, a.K)) {
      p.splits = want;
      p.part = a.workspace;
      p.sk_cnt = a.counters;
    }
  }
  // 64x128 tiles keep the tile count a large multiple of the CU count at the path's shapes
  // (M = 12800: 200 x N/128 tiles), 128x128 is used when there are plenty of tiles anyway.
  static const int tile_cfg = getenv("CLIPFS_GEMM_TILE") ? atoi(getenv("CLIPFS_GEMM_TILE")) : 0;  // tuning aid
  if (tile_cfg == 1 && a.a_mode == 0 && (a.K % BK) == 0) {
    p.splits = 1;  // the slab geometry is that of the default tiling
    p.n_blocks_n = (a.N + 127) / 128;
    return launch<128, 128, 3>(p, s);
  }
  if (tile_cfg == 2 && a.a_mode == 0 && (a.K % BK) == 0) {
    p.splits = 1;
    p.n_blocks_n = (a.N + 63) / 64;
    return launch<128, 64, 0>(p, s);
  }
  // dense, K % 32 == 0, scratch + counters supplied: stream-K (every run the same number of K-steps)
  if (sk_enabled() && tile_cfg == 0 && a.a_mode == 0 && (a.K % BK) == 0 && a.workspace && a.counters &&
      sk_config(a.M, a.N, a.K).use &&
      (long)((a.M + 63) / 64) * ((a.N + 127) / 128) * 2 * (a.K / BK) < (1L << 30) &&
      a.workspace_floats >= clipfs_gemm_workspace_floats(a.M, a.N, a.K) &&
      a.counters_ints >= clipfs_gemm_counter_ints(a.M, a.N, a.K)) {
    CLIPFS_REQUIRE(aligned16(a.workspace), "gemm: workspace must be 16-byte aligned");
    p.splits = 1;
    return launch_sk<64, 128, 2>(p, sk_config(a.M, a.N, a.K).per_cu, s);
  }
  p.n_blocks_n = (a.N + 127) / 128;
  if (gemm_bm(a.M, a.N) == 32 && a.a_mode == 0 && (a.K % BK) == 0) return launch<32, 128, 3>(p, s);
  static const int glds_cfg = getenv("CLIPFS_GEMM_GLDS") ? atoi(getenv("CLIPFS_GEMM_GLDS")) : 1;  // 0: register staging (A/B aid)
  if (a.a_mode == 0 && (a.K % BK) == 0) return glds_cfg ? launch<64, 128, 3>(p, s) : launch<64, 128, 0>(p, s);
  if (a.a_mode == 1 && a.patch == 32 && (a.img_res & 3) == 0) return launch<64, 128, 1>(p, s);
  return launch<64, 128, 2>(p, s);
}
