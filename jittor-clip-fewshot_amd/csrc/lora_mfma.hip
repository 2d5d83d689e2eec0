// Rank-r LoRA products on the matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32: the fp32 MFMA of the dense GEMMs
// in its 16x16 shape, whose N = 16 fits ranks up to 16).  Same five products, same Philox dropout stream and same
// results (up to summation order) as the one-wave-per-row kernels in lora.hip, which remain for shapes outside
// r <= 16 / width % 64 == 0.  At r = 16 (cfg-5) the scalar kernels spend ~50 wave reductions per row; here every
// product is a tall-skinny MFMA GEMM whose big operand is read once with 16-byte loads:
//
//   down  t[m, s r + j]   = sum_k drop_s(x)[m, k] A[s r + j, k]          A-operand = x rows, B-operand = A rows
//   dt    dt[m, s r + j]  = scale sum_n dy[m, s w + n] B[s w + n, j]
//   dB    dB[n, j]       += scale sum_m dy[m, n] t[m, seg(n) r + j]      reduction over rows: K index = row
//   dA    dA[s r + j, k] += sum_m dt[m, s r + j] drop_s(x)[m, k]
//   dx    dx[m, k]       += sum_s dropscale_s(m, k) sum_j dt[m, s r + j] A[s r + j, k]
//
// MFMA 16x16x4 layouts: A operand lane l = A[i = l & 15][k = l >> 4], B operand lane l = B[k = l >> 4][j = l & 15],
// result lane l = D[i = 4 (l >> 4) + v][j = l & 15], v = 0..3.  The k (and, for dB/dA/dx, the column) assignment is
// free as long as both operands agree, so a lane always loads FOUR CONSECUTIVE floats (one float4 = one Philox call
// = 4 dropout multipliers) and feeds them to four successive MFMAs.
#include "common.h"

namespace clipfs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 zero4() {
  f32x4 z;
  z[0] = z[1] = z[2] = z[3] = 0.f;
  return z;
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// fp16 storage mode: the incoming gradient is read from its f16 image (the dgrad GEMM's A operand): half the bytes
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld4(const _Float16* p) {
  const f16x4 h = *reinterpret_cast<const f16x4*>(p);
  f32x4 v;
  v[0] = (float)h[0];
  v[1] = (float)h[1];
  v[2] = (float)h[2];
  v[3] = (float)h[3];
  return v;
}

// Sum the four waves' partial accumulators (the waves of a block split the reduction axis) through LDS; wave 0 gets
// the total.  red: [4 waves][NACC][64 lanes] f32x4.
template <int NACC>
__device__ __forceinline__ void block_sum4(f32x4 (&acc)[NACC], f32x4* red, int wave, int lane) {
  if (wave != 0) {
#pragma unroll
    for (int s = 0; s < NACC; ++s) red[((wave - 1) * NACC + s) * 64 + lane] = acc[s];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int s = 0; s < NACC; ++s) acc[s] += red[(w * NACC + s) * 64 + lane];
  }
}

// t = drop(x) A^T.  One block per 16 rows; its 4 waves take a quarter of the columns each (rows / 16 waves alone
// would leave ~2 waves per SIMD).
template <int NSEG>
__global__ __launch_bounds__(256) void lora_down_mfma_kernel(const float* __restrict__ x, const float* __restrict__ A,
                                                             float* __restrict__ t, int rows, int width, int r,
                                                             unsigned seg_mask, float p, uint64_t seed,
                                                             uint32_t stream_base, uint32_t drow0,
                                                             uint16_t* __restrict__ keep_bits) {
  __shared__ f32x4 red[3 * NSEG * 64];
  const int lane = threadIdx.x & 63, li = lane & 15, kg = lane >> 4;
  const int wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * 16;
  const int m = min(row0 + li, rows - 1);
  const bool drop = p > 0.f && seed != 0;
  const uint32_t thr = dropout_threshold(p);
  const float inv_keep = 1.f / (1.f - p);
  const float* xr = x + (size_t)m * width + 4 * kg;
  const float* ar = A + (size_t)min(li, r - 1) * width + 4 * kg;
  f32x4 acc[NSEG];
#pragma unroll
  for (int s = 0; s < NSEG; ++s) acc[s] = zero4();
  const int cw = width >> 2;  // columns per wave (width % 64 == 0)
  const int cend = (wave + 1) * cw;
  for (int c0 = wave * cw; c0 < cend; c0 += 64) {  // 4 steps of 16 columns, loads first
    f32x4 xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int cu = c0 + 16 * u;
      xv[u] = ld4(xr + min(cu, cend - 16));
      if (cu >= cend) xv[u] = zero4();
    }
    uint32_t kb[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
      if (!((seg_mask >> s) & 1u)) continue;
      f32x4 wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) wv[u] = ld4(ar + (size_t)s * r * width + min(c0 + 16 * u, cend - 16));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x4 xs = xv[u];
        if (drop) {
          const float4 mk = dropout_scale4(seed, stream_base + s, drow0 + (uint32_t)m, (uint32_t)(((c0 + 16 * u) >> 2) + kg), thr, inv_keep);
          xs[0] *= mk.x;
          xs[1] *= mk.y;
          xs[2] *= mk.z;
          xs[3] *= mk.w;
          kb[u] |= keep_bits4(mk) << (4 * s);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[s] = mfma16(xs[e], wv[u][e], acc[s]);
      }
    }
    if (keep_bits && drop && row0 + li < rows) {  // the masks of this pass, for the backward (4 bits per segment and float4)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c0 + 16 * u < cend) keep_bits[(size_t)m * (width >> 2) + ((c0 + 16 * u) >> 2) + kg] = (uint16_t)kb[u];
    }
  }
  block_sum4<NSEG>(acc, red, wave, lane);
  if (wave == 0 && li < r) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int mo = row0 + 4 * kg + v;
      if (mo < rows) {
#pragma unroll
        for (int s = 0; s < NSEG; ++s) t[(size_t)mo * (NSEG * r) + s * r + li] = ((seg_mask >> s) & 1u) ? acc[s][v] : 0.f;
      }
    }
  }
}

// dt = scale * dy_seg B_seg.  One block (index bx) per 16 rows.
template <int NSEG, typename TY>
__device__ __forceinline__ void lora_dt_mfma_body(const TY* __restrict__ dy, const float* __restrict__ B,
                                                  float* __restrict__ dt, int rows, int segw, int r, unsigned seg_mask,
                                                  float scale, int bx, f32x4* red) {
  const int lane = threadIdx.x & 63, li = lane & 15, kg = lane >> 4;
  const int wave = threadIdx.x >> 6;
  const int row0 = bx * 16;
  const int m = min(row0 + li, rows - 1);
  const int cw = segw >> 2;
  const TY* dr = dy + (size_t)m * NSEG * segw + 4 * kg;
  const int jj = min(li, r - 1);
  f32x4 acc[NSEG];
#pragma unroll
  for (int s = 0; s < NSEG; ++s) acc[s] = zero4();
#pragma unroll
  for (int s = 0; s < NSEG; ++s) {
    if (!((seg_mask >> s) & 1u)) continue;
    const float* bs = B + ((size_t)s * segw + 4 * kg) * r + jj;
    const int cend = (wave + 1) * cw;
    for (int c0 = wave * cw; c0 < cend; c0 += 64) {
      f32x4 g[4];
      float bv[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cu = min(c0 + 16 * u, cend - 16);
        g[u] = ld4(dr + s * segw + cu);
        if (c0 + 16 * u >= cend) g[u] = zero4();
        const float* bp = bs + (size_t)cu * r;
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[u][e] = bp[e * r];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[s] = mfma16(g[u][e], bv[u][e], acc[s]);
    }
  }
  block_sum4<NSEG>(acc, red, wave, lane);
  if (wave == 0 && li < r) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int mo = row0 + 4 * kg + v;
      if (mo < rows) {
#pragma unroll
        for (int s = 0; s < NSEG; ++s) dt[(size_t)mo * (NSEG * r) + s * r + li] = scale * acc[s][v];
      }
    }
  }
}

// dB partials: part[slice][n][j] = sum_{m in slice} dy[m, n] t[m, seg(n) r + j].  One wave per (64 columns, slice);
// lane i owns columns n0 + 4 i + e of MFMA tile e.
template <typename TY>
__device__ __forceinline__ void lora_db_mfma_body(const TY* __restrict__ dy, const float* __restrict__ t,
                                                  float* __restrict__ part, int rows, int cols, int segw, int nseg, int r,
                                                  int rows_per_slice, int bx, int by) {
  const int lane = threadIdx.x & 63, li = lane & 15, kg = lane >> 4;
  const int n0 = (bx * 4 + (threadIdx.x >> 6)) * 64;
  if (n0 >= cols) return;
  const int slice = by;
  const int m0 = slice * rows_per_slice, m1 = min(rows, m0 + rows_per_slice);
  const int s = n0 / segw;
  const int tw = nseg * r;
  const TY* dp = dy + n0 + 4 * li;
  const float* tp = t + s * r + min(li, r - 1);
  f32x4 acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = zero4();

  for (int mb = m0; mb < m1; mb += 16) {  // 4 steps of 4 rows, loads first
    f32x4 g[4];
    float tv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = mb + 4 * u + kg;
      const bool ok = m < m1;
      const int mc = ok ? m : m1 - 1;
      g[u] = ld4(dp + (size_t)mc * cols);
      tv[u] = tp[(size_t)mc * tw];
      if (!ok) {
        g[u] = zero4();
        tv[u] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = mfma16(g[u][e], tv[u], acc[e]);
  }
  if (li < r) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int v = 0; v < 4; ++v) part[((size_t)slice * cols + n0 + 4 * (4 * kg + v) + e) * r + li] = acc[e][v];
  }
}

// dA partials: part[slice][s r + j][k] = sum_{m in slice} dt[m, s r + j] drop_s(x)[m, k].  One wave per (64 columns,
// slice); lane i owns columns k0 + 4 i + e of MFMA tile e (so one Philox call covers the lane's float4 of x).
template <int NSEG>
__device__ __forceinline__ void lora_da_mfma_body(const float* __restrict__ x, const float* __restrict__ dt,
                                                  float* __restrict__ part, int rows, int width, int r, unsigned seg_mask,
                                                  float p, uint64_t seed, uint32_t stream_base, uint32_t drow0,
                                                  int rows_per_slice, int bx, int by,
                                                  const uint16_t* __restrict__ keep_bits) {
  const int lane = threadIdx.x & 63, li = lane & 15, kg = lane >> 4;
  const int k0 = (bx * 4 + (threadIdx.x >> 6)) * 64;
  if (k0 >= width) return;
  const int slice = by;
  const int m0 = slice * rows_per_slice, m1 = min(rows, m0 + rows_per_slice);
  const bool drop = p > 0.f && seed != 0;
  const uint32_t thr = dropout_threshold(p);
  const float inv_keep = 1.f / (1.f - p);
  const int tw = NSEG * r;
  const float* xp = x + k0 + 4 * li;
  const float* dp = dt + min(li, r - 1);
  const uint32_t c4 = (uint32_t)((k0 >> 2) + li);
  f32x4 acc[NSEG][4];
#pragma unroll
  for (int s = 0; s < NSEG; ++s)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[s][e] = zero4();
  for (int mb = m0; mb < m1; mb += 8) {  // 2 steps of 4 rows, loads first
    f32x4 xv[2];
    float g[2][NSEG];
    int mcs[2];
    uint32_t kbu[2] = {0u, 0u};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = mb + 4 * u + kg;
      const bool ok = m < m1;
      mcs[u] = ok ? m : m1 - 1;
      xv[u] = ld4(xp + (size_t)mcs[u] * width);
      if (keep_bits) kbu[u] = keep_bits[(size_t)mcs[u] * (width >> 2) + c4];
      if (!ok) xv[u] = zero4();
#pragma unroll
      for (int s = 0; s < NSEG; ++s) g[u][s] = dp[(size_t)mcs[u] * tw + s * r];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < NSEG; ++s) {
        if (!((seg_mask >> s) & 1u)) continue;
        f32x4 xs = xv[u];
        if (drop) {
          const float4 mk = keep_bits ? keep_scale4(kbu[u] >> (4 * s), inv_keep)
                                      : dropout_scale4(seed, stream_base + s, drow0 + (uint32_t)mcs[u], c4, thr, inv_keep);
          xs[0] *= mk.x;
          xs[1] *= mk.y;
          xs[2] *= mk.z;
          xs[3] *= mk.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[s][e] = mfma16(g[u][s], xs[e], acc[s][e]);
      }
  }
  // result tile e: D[i = rank index 4 kg + v][j = li] <-> column k0 + 4 li + e: one float4 per (s, v)
#pragma unroll
  for (int s = 0; s < NSEG; ++s)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int j = 4 * kg + v;
      if (j < r) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[s][e][v];
        *reinterpret_cast<f32x4*>(part + ((size_t)slice * tw + s * r + j) * width + k0 + 4 * li) = o;
      }
    }
}

// dx[m, k] += sum_s dropscale_s(m, k) sum_j dt[m, s r + j] A[s r + j, k].  One block per 16 rows (its 4 waves take a
// quarter of the columns each); per 16-column tile
// the result D[i <-> column k0 + i][j <-> row] gives a lane 4 consecutive columns of one row: a float4 of dx.
template <int NSEG, int RQ>  // RQ = ceil(r / 4) MFMA K-steps
__device__ __forceinline__ void lora_dx_mfma_body(const float* __restrict__ dt, const float* __restrict__ A,
                                                  float* __restrict__ dx, int rows, int width, int r, unsigned seg_mask,
                                                  float p, uint64_t seed, uint32_t stream_base, uint32_t drow0, int bx,
                                                  const uint16_t* __restrict__ keep_bits) {
  const int lane = threadIdx.x & 63, li = lane & 15, kg = lane >> 4;
  const int wave = threadIdx.x >> 6;
  const int row0 = bx * 16;
  const int m = row0 + li;
  const int mc = min(m, rows - 1);
  const int cw = width >> 2;
  const bool drop = p > 0.f && seed != 0;
  const uint32_t thr = dropout_threshold(p);
  const float inv_keep = 1.f / (1.f - p);
  float dtv[NSEG][RQ];  // B operand: k = rank index 4 q + kg, j = row li
  const float* ap[NSEG][RQ];
  float amask[RQ];
#pragma unroll
  for (int q = 0; q < RQ; ++q) {
    const int j = 4 * q + kg;
    amask[q] = j < r ? 1.f : 0.f;
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
      dtv[s][q] = dt[(size_t)mc * (NSEG * r) + s * r + min(j, r - 1)] * amask[q];
      ap[s][q] = A + (size_t)(s * r + min(j, r - 1)) * width + li;
    }
  }
  float* xr = dx + (size_t)mc * width + 4 * kg;
  for (int k0 = wave * cw; k0 < (wave + 1) * cw; k0 += 32) {  // two 16-column tiles, loads first (cw % 32 == 0)
    f32x4 tot[2];
    float av[2][NSEG][RQ];
    uint32_t kbu[2] = {0u, 0u};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      tot[u] = ld4(xr + k0 + 16 * u);
      if (keep_bits) kbu[u] = keep_bits[(size_t)mc * (width >> 2) + ((k0 + 16 * u) >> 2) + kg];
#pragma unroll
      for (int s = 0; s < NSEG; ++s)
#pragma unroll
        for (int q = 0; q < RQ; ++q) av[u][s][q] = ap[s][q][k0 + 16 * u];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int s = 0; s < NSEG; ++s) {
        if (!((seg_mask >> s) & 1u)) continue;
        f32x4 acc = zero4();
#pragma unroll
        for (int q = 0; q < RQ; ++q) acc = mfma16(av[u][s][q], dtv[s][q], acc);
        if (drop) {
          const float4 mk = keep_bits ? keep_scale4(kbu[u] >> (4 * s), inv_keep)
                                      : dropout_scale4(seed, stream_base + s, drow0 + (uint32_t)mc, (uint32_t)(((k0 + 16 * u) >> 2) + kg), thr, inv_keep);
          acc[0] *= mk.x;
          acc[1] *= mk.y;
          acc[2] *= mk.z;
          acc[3] *= mk.w;
        }
        tot[u] += acc;
      }
      if (m < rows) *reinterpret_cast<f32x4*>(xr + k0 + 16 * u) = tot[u];
    }
  }
}

// ---- the backward as three launches ---------------------------------------------------------------------------
// dt and dB both read dy and do not depend on each other; dA and dx both need dt and touch different tensors.  Each
// pair is ONE launch whose leading blocks do the first product and whose trailing blocks do the second (the bodies
// above, unchanged: same arithmetic, same results): at the per-rank sizes of the 8-GPU step every one of these products
// is a few microseconds of work behind ~10 us of launch, and the two halves of a pair fill each other's tail.
template <int NSEG, typename TY>
__global__ __launch_bounds__(256) void lora_db_dt_mfma_kernel(const TY* __restrict__ dy, const float* __restrict__ B,
                                                              const float* __restrict__ t, float* __restrict__ dt,
                                                              float* __restrict__ part_b, int rows, int segw, int r,
                                                              unsigned seg_mask, float scale, int rows_per_slice,
                                                              int gx_b, int n_db) {
  __shared__ f32x4 red[3 * NSEG * 64];
  const int b = blockIdx.x;
  if (b < n_db)
    lora_db_mfma_body<TY>(dy, t, part_b, rows, NSEG * segw, segw, NSEG, r, rows_per_slice, b % gx_b, b / gx_b);
  else
    lora_dt_mfma_body<NSEG, TY>(dy, B, dt, rows, segw, r, seg_mask, scale, b - n_db, red);
}

template <int NSEG, int RQ>
__global__ __launch_bounds__(256) void lora_da_dx_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dt,
                                                              const float* __restrict__ A, float* __restrict__ part_a,
                                                              float* __restrict__ dx, int rows, int width, int r,
                                                              unsigned seg_mask, float p, uint64_t seed,
                                                              uint32_t stream_base, uint32_t drow0, int rows_per_slice,
                                                              int gx_a, int n_da, const uint16_t* __restrict__ keep_bits) {
  const int b = blockIdx.x;
  if (b < n_da)
    lora_da_mfma_body<NSEG>(x, dt, part_a, rows, width, r, seg_mask, p, seed, stream_base, drow0, rows_per_slice, b % gx_a,
                            b / gx_a, keep_bits);
  else
    lora_dx_mfma_body<NSEG, RQ>(dt, A, dx, rows, width, r, seg_mask, p, seed, stream_base, drow0, b - n_da, keep_bits);
}

// ---- host side (called from lora.hip) ---------------------------------------------------------------------

bool lora_mfma_ok(int width, int segw, int r, int nseg) {
  return r >= 1 && r <= 16 && (nseg == 1 || nseg == 3) && (width % 128) == 0 && (segw % 64) == 0;
}

// rows per reduction slice: enough slices to put ~6000 waves (column groups x slices) on the chip, few enough that
// the partial sums stay small; never below 64 rows (the work buffer is sized for 64-row slices)
static int lora_mfma_slice_rows(int rows, int col_groups) {
  int sr = 2048;
  while (sr > 64 && (long)col_groups * ((rows + sr - 1) / sr) < 6144) sr >>= 1;
  return sr;
}

int lora_down_mfma(const float* x, const float* A, float* t, int rows, int width, int r, int nseg, unsigned seg_mask,
                   float p, uint64_t seed, uint32_t stream_base, uint32_t drow0, uint16_t* keep_bits, hipStream_t st) {
  const dim3 grid((rows + 15) / 16);
  if (nseg == 1)
    hipLaunchKernelGGL((lora_down_mfma_kernel<1>), grid, dim3(256), 0, st, x, A, t, rows, width, r, seg_mask, p, seed, stream_base, drow0, keep_bits);
  else
    hipLaunchKernelGGL((lora_down_mfma_kernel<3>), grid, dim3(256), 0, st, x, A, t, rows, width, r, seg_mask, p, seed, stream_base, drow0, keep_bits);
  return launch_status();
}

template <int NSEG, int RQ>
static void launch_da_dx(const float* x, const float* dt, const float* A, float* part_a, float* dx, int rows, int width, int r,
                         unsigned seg_mask, float p, uint64_t seed, uint32_t stream_base, uint32_t drow0, int sr_a, int slices_a,
                         const uint16_t* keep_bits, hipStream_t st) {
  const int gx_a = (width + 255) / 256, n_da = gx_a * slices_a;
  const int n_dx = dx ? (rows + 15) / 16 : 0;
  hipLaunchKernelGGL((lora_da_dx_mfma_kernel<NSEG, RQ>), dim3(n_da + n_dx), dim3(256), 0, st, x, dt, A, part_a, dx, rows, width,
                     r, seg_mask, p, seed, stream_base, drow0, sr_a, gx_a, n_da, keep_bits);
}

// reduce2(part_b, dB, nb, slices_b, scale_b, part_a, dA, na, slices_a, scale_a): both slice sums in one launch
typedef void (*lora_reduce2_fn)(const float*, float*, size_t, int, float, const float*, float*, size_t, int, float, hipStream_t);

template <int NSEG, typename TY>
static int lora_bwd_mfma_n(const TY* dy, const float* x, const float* t, const float* A, const float* B, float* dt,
                           float* dA, float* dB, float* dx, int rows, int width, int segw, int r, unsigned seg_mask,
                           float scale, float p, uint64_t seed, uint32_t stream_base, uint32_t drow0, const uint16_t* keep_bits,
                           float* work, hipStream_t st, lora_reduce2_fn reduce2) {
  const int cols = NSEG * segw;
  const int sr_b = lora_mfma_slice_rows(rows, cols / 64), slices_b = (rows + sr_b - 1) / sr_b;
  const int sr_a = lora_mfma_slice_rows(rows, width / 64), slices_a = (rows + sr_a - 1) / sr_a;
  const int n_rows16 = (rows + 15) / 16;
  const size_t nb = (size_t)cols * r, na = (size_t)NSEG * r * width;
  float* part_b = work;
  float* part_a = work + (((size_t)slices_b * nb + 3) & ~(size_t)3);
  // 1: dB partials || dt
  const int gx_b = (cols + 255) / 256, n_db = gx_b * slices_b;
  hipLaunchKernelGGL((lora_db_dt_mfma_kernel<NSEG, TY>), dim3(n_db + n_rows16), dim3(256), 0, st, dy, B, t, dt, part_b, rows,
                     segw, r, seg_mask, scale, sr_b, gx_b, n_db);
  CLIPFS_CHECK(launch_status());
  // 2: dA partials || dx
  switch ((r + 3) / 4) {
    case 1:
      launch_da_dx<NSEG, 1>(x, dt, A, part_a, dx, rows, width, r, seg_mask, p, seed, stream_base, drow0, sr_a, slices_a, keep_bits, st);
      break;
    case 2:
      launch_da_dx<NSEG, 2>(x, dt, A, part_a, dx, rows, width, r, seg_mask, p, seed, stream_base, drow0, sr_a, slices_a, keep_bits, st);
      break;
    case 3:
      launch_da_dx<NSEG, 3>(x, dt, A, part_a, dx, rows, width, r, seg_mask, p, seed, stream_base, drow0, sr_a, slices_a, keep_bits, st);
      break;
    default:
      launch_da_dx<NSEG, 4>(x, dt, A, part_a, dx, rows, width, r, seg_mask, p, seed, stream_base, drow0, sr_a, slices_a, keep_bits, st);
      break;
  }
  CLIPFS_CHECK(launch_status());
  // 3: both slice sums
  reduce2(part_b, dB, nb, slices_b, scale, part_a, dA, na, slices_a, 1.0f, st);
  return launch_status();
}

int lora_bwd_mfma(const float* dy, const float* x, const float* t, const float* A, const float* B, float* dt, float* dA,
                  float* dB, float* dx, int rows, int width, int segw, int r, int nseg, unsigned seg_mask, float scale,
                  float p, uint64_t seed, uint32_t stream_base, uint32_t drow0, const uint16_t* keep_bits, float* work,
                  hipStream_t st, lora_reduce2_fn reduce) {
  if (nseg == 1)
    return lora_bwd_mfma_n<1, float>(dy, x, t, A, B, dt, dA, dB, dx, rows, width, segw, r, seg_mask, scale, p, seed, stream_base, drow0,
                                     keep_bits, work, st, reduce);
  return lora_bwd_mfma_n<3, float>(dy, x, t, A, B, dt, dA, dB, dx, rows, width, segw, r, seg_mask, scale, p, seed, stream_base, drow0,
                                   keep_bits, work, st, reduce);
}

// the same with dy given as its f16 image [rows, nseg * segw] (fp16 storage mode)
int lora_bwd_mfma_f16dy(const void* dy16, const float* x, const float* t, const float* A, const float* B, float* dt, float* dA,
                        float* dB, float* dx, int rows, int width, int segw, int r, int nseg, unsigned seg_mask, float scale,
                        float p, uint64_t seed, uint32_t stream_base, uint32_t drow0, const uint16_t* keep_bits,
                        float* work, hipStream_t st, lora_reduce2_fn reduce) {
  const _Float16* dy = reinterpret_cast<const _Float16*>(dy16);
  if (nseg == 1)
    return lora_bwd_mfma_n<1, _Float16>(dy, x, t, A, B, dt, dA, dB, dx, rows, width, segw, r, seg_mask, scale, p, seed, stream_base,
                                        drow0, keep_bits, work, st, reduce);
  return lora_bwd_mfma_n<3, _Float16>(dy, x, t, A, B, dt, dA, dB, dx, rows, width, segw, r, seg_mask, scale, p, seed, stream_base,
                                      drow0, keep_bits, work, st, reduce);
}

}  // namespace clipfs
