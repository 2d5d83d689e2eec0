// LoRA adapter kernels.  The reference materialises B@A ([d,d]) and runs a second full d x d GEMM
// per adapted projection (lora_train_vlp.py:218-221,302: +26 % FLOPs); here the rank-r form is kept:
//   down : t = drop(x) A^T           [rows, nseg*r]   (this file, HBM-bound: one read of x)
//   up   : y += scale * t B^T        (fused into the GEMM epilogue, gemm.hip)
// and the backward is three skinny products that each read their big operand exactly once.
#include "common.h"

#include <stdlib.h>

namespace clipfs {

constexpr int LORA_MAX_CHUNKS = 8;  // width <= 2048
constexpr int LORA_MAX_OUT = 64;    // nseg * r

// one wave per row
__global__ __launch_bounds__(256) void lora_down_kernel(const float* __restrict__ x, const float* __restrict__ A,
                                                        float* __restrict__ t, int rows, int width, int r, int nseg,
                                                        unsigned seg_mask, float p, uint64_t seed,
                                                        uint32_t stream_base, uint32_t drow0) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = width >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * width);
  float4 v[LORA_MAX_CHUNKS];
#pragma unroll
  for (int i = 0; i < LORA_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) v[i] = xr[c];
  }
  const bool drop = p > 0.f && seed != 0;
  const uint32_t thr = dropout_threshold(p);
  const float inv_keep = 1.f / (1.f - p);
  float* trow = t + (size_t)row * (nseg * r);
  for (int s = 0; s < nseg; ++s) {
    if (!((seg_mask >> s) & 1u)) {
      if (lane < r) trow[s * r + lane] = 0.f;
      continue;
    }
    float4 xs[LORA_MAX_CHUNKS];
#pragma unroll
    for (int i = 0; i < LORA_MAX_CHUNKS; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        xs[i] = v[i];
        if (drop) {
          const float4 m = dropout_scale4(seed, stream_base + s, drow0 + (uint32_t)row, (uint32_t)c, thr, inv_keep);
          xs[i].x *= m.x;
          xs[i].y *= m.y;
          xs[i].z *= m.z;
          xs[i].w *= m.w;
        }
      }
    }
    for (int j = 0; j < r; ++j) {
      const float4* ar = reinterpret_cast<const float4*>(A + (size_t)(s * r + j) * width);
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < LORA_MAX_CHUNKS; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
          const float4 a = ar[c];
          acc += (xs[i].x * a.x + xs[i].y * a.y) + (xs[i].z * a.z + xs[i].w * a.w);
        }
      }
      acc = wave_sum(acc);
      if (lane == 0) trow[s * r + j] = acc;
    }
  }
}

// dt[m, s*r+j] = scale * sum_n dy[m, s*segw+n] * B[s*segw+n, j]     one wave per row, 16-byte loads:
// a lane takes 4 consecutive n (one float4 of dy) and the 4 matching rows of B.
template <int R>
__global__ __launch_bounds__(256) void lora_dt_kernel(const float* __restrict__ dy, const float* __restrict__ B,
                                                      float* __restrict__ dt, int rows, int segw, int nseg,
                                                      unsigned seg_mask, float scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* dr = dy + (size_t)row * nseg * segw;
  float* out = dt + (size_t)row * nseg * R;
  for (int s = 0; s < nseg; ++s) {
    if (!((seg_mask >> s) & 1u)) {
      if (lane < R) out[s * R + lane] = 0.f;
      continue;
    }
    float acc[R];
#pragma unroll
    for (int j = 0; j < R; ++j) acc[j] = 0.f;
    for (int n4 = lane; n4 < (segw >> 2); n4 += 64) {
      const float4 g = *reinterpret_cast<const float4*>(dr + s * segw + 4 * n4);
      const float* b = B + ((size_t)s * segw + 4 * n4) * R;
      const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < R; ++j) acc[j] = fmaf(gv[e], b[e * R + j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const float v = wave_sum(acc[j]);
      if (lane == 0) out[s * R + j] = scale * v;
    }
  }
}

// Column-parallel tall reduction:  part[slice][c][j] = sum_{m in slice} X[m, c] * T[m, toff(c) + j]
// thread = one column c (coalesced across lanes), T row values are wave-uniform broadcasts.
// Used for dB (X = dy, T = t) -- deterministic two stage sum (no float atomics).
template <int R>
__global__ __launch_bounds__(256) void lora_db_partial_kernel(const float* __restrict__ dy, const float* __restrict__ t,
                                                              float* __restrict__ part, int rows, int cols, int segw,
                                                              int nseg, int rows_per_slice) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int slice = blockIdx.y;
  if (c >= cols) return;
  const int s = c / segw;
  const int m0 = slice * rows_per_slice, m1 = min(rows, m0 + rows_per_slice);
  float acc[R];
#pragma unroll
  for (int j = 0; j < R; ++j) acc[j] = 0.f;
  const int tw = nseg * R;
  for (int m = m0; m < m1; ++m) {
    const float g = dy[(size_t)m * cols + c];
    const float* tr = t + (size_t)m * tw + s * R;
#pragma unroll
    for (int j = 0; j < R; ++j) acc[j] = fmaf(g, tr[j], acc[j]);
  }
  float* o = part + ((size_t)slice * cols + c) * R;
#pragma unroll
  for (int j = 0; j < R; ++j) o[j] = acc[j];
}

// dA partials: thread = 4 consecutive columns k of x; acc[s][j] over the slice's rows, dropout
// multipliers regenerated from the Philox stream (never stored).  One wave per (256 columns, row slice).
template <int R, int NSEG>
__global__ __launch_bounds__(64) void lora_da_partial_kernel(const float* __restrict__ x, const float* __restrict__ dt,
                                                             float* __restrict__ part, int rows, int width,
                                                             unsigned seg_mask, float p, uint64_t seed,
                                                             uint32_t stream_base, uint32_t drow0, int rows_per_slice) {
  const int c4 = blockIdx.x * 64 + threadIdx.x;  // chunk of 4 columns
  const int slice = blockIdx.y;
  if (c4 * 4 >= width) return;
  const int m0 = slice * rows_per_slice, m1 = min(rows, m0 + rows_per_slice);
  const bool drop = p > 0.f && seed != 0;
  const uint32_t thr = dropout_threshold(p);
  const float inv_keep = 1.f / (1.f - p);
  float4 acc[NSEG][R];
#pragma unroll
  for (int s = 0; s < NSEG; ++s)
#pragma unroll
    for (int j = 0; j < R; ++j) acc[s][j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
  for (int m = m0; m < m1; ++m) {
    const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)m * width + 4 * c4);
    const float* dr = dt + (size_t)m * (NSEG * R);
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
      if (!((seg_mask >> s) & 1u)) continue;
      float4 xs = xv;
      if (drop) {
        const float4 mk = dropout_scale4(seed, stream_base + s, drow0 + (uint32_t)m, (uint32_t)c4, thr, inv_keep);
        xs.x *= mk.x;
        xs.y *= mk.y;
        xs.z *= mk.z;
        xs.w *= mk.w;
      }
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const float g = dr[s * R + j];
        acc[s][j].x = fmaf(g, xs.x, acc[s][j].x);
        acc[s][j].y = fmaf(g, xs.y, acc[s][j].y);
        acc[s][j].z = fmaf(g, xs.z, acc[s][j].z);
        acc[s][j].w = fmaf(g, xs.w, acc[s][j].w);
      }
    }
  }
  // part layout [slice][s*R + j][width]
#pragma unroll
  for (int s = 0; s < NSEG; ++s)
#pragma unroll
    for (int j = 0; j < R; ++j)
      *reinterpret_cast<float4*>(part + ((size_t)slice * (NSEG * R) + s * R + j) * width + 4 * c4) = acc[s][j];
}

// out[i] += scale * sum_slice part[slice][i].  64 outputs x 16 slice groups per block; group g sums slices
// g, g+16, ... and the 16 group sums are added in a fixed order => bitwise reproducible.
__device__ __forceinline__ void reduce_slices_body(const float* __restrict__ part, float* __restrict__ out, size_t n,
                                                   int slices, float scale, unsigned block, float (*red)[64]) {
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const size_t i = (size_t)block * 64 + lane;
  float acc = 0.f;
  if (i < n)
    for (int s = grp; s < slices; s += 16) acc += part[(size_t)s * n + i];
  red[grp][lane] = acc;
  __syncthreads();
  if (grp == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][lane];
    out[i] += scale * t;
  }
}

__global__ __launch_bounds__(1024) void reduce_slices_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                             size_t n, int slices, float scale) {
  __shared__ float red[16][64];
  reduce_slices_body(part, out, n, slices, scale, blockIdx.x, red);
}

__global__ __launch_bounds__(1024) void reduce_slices2_kernel(const float* __restrict__ part0, float* __restrict__ out0,
                                                              size_t n0, int slices0, float scale0,
                                                              const float* __restrict__ part1, float* __restrict__ out1,
                                                              size_t n1, int slices1, float scale1, unsigned nblk0) {
  __shared__ float red[16][64];
  if (blockIdx.x < nblk0)
    reduce_slices_body(part0, out0, n0, slices0, scale0, blockIdx.x, red);
  else
    reduce_slices_body(part1, out1, n1, slices1, scale1, blockIdx.x - nblk0, red);
}

// dx[m,k] += sum_{s,j} dt[m, s*r+j] * A[s*r+j, k] * dropscale_s(m,k)       one wave per row
__global__ __launch_bounds__(256) void lora_dx_kernel(const float* __restrict__ dt, const float* __restrict__ A,
                                                      float* __restrict__ dx, int rows, int width, int r, int nseg,
                                                      unsigned seg_mask, float p, uint64_t seed,
                                                      uint32_t stream_base, uint32_t drow0) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = width >> 2;
  const bool drop = p > 0.f && seed != 0;
  const uint32_t thr = dropout_threshold(p);
  const float inv_keep = 1.f / (1.f - p);
  const float* dr = dt + (size_t)row * nseg * r;
  float4* xr = reinterpret_cast<float4*>(dx + (size_t)row * width);
  for (int c = lane; c < nch; c += 64) {
    float4 tot = xr[c];
    for (int s = 0; s < nseg; ++s) {
      if (!((seg_mask >> s) & 1u)) continue;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = 0; j < r; ++j) {
        const float g = dr[s * r + j];
        const float4 av = *reinterpret_cast<const float4*>(A + (size_t)(s * r + j) * width + 4 * c);
        a.x = fmaf(g, av.x, a.x);
        a.y = fmaf(g, av.y, a.y);
        a.z = fmaf(g, av.z, a.z);
        a.w = fmaf(g, av.w, a.w);
      }
      if (drop) {
        const float4 mk = dropout_scale4(seed, stream_base + s, drow0 + (uint32_t)row, (uint32_t)c, thr, inv_keep);
        a.x *= mk.x;
        a.y *= mk.y;
        a.z *= mk.z;
        a.w *= mk.w;
      }
      tot.x += a.x;
      tot.y += a.y;
      tot.z += a.z;
      tot.w += a.w;
    }
    xr[c] = tot;
  }
}

// rows per reduction slice: 64 for large row counts, smaller when there are few rows so that the partial
// kernels still put >= ~1000 waves on the chip (per-rank batches of 32 images: M = 1600)
static inline int lora_slice_rows(int rows) {
  int r = 64;
  while (r > 8 && (rows + r - 1) / r < 256) r >>= 1;
  return r;
}

// lora_mfma.hip
bool lora_mfma_ok(int width, int segw, int r, int nseg);
int lora_down_mfma(const float* x, const float* A, float* t, int rows, int width, int r, int nseg, unsigned seg_mask,
                   float p, uint64_t seed, uint32_t stream_base, uint32_t drow0, uint16_t* keep_bits, hipStream_t st);
typedef void (*lora_reduce2_fn)(const float*, float*, size_t, int, float, const float*, float*, size_t, int, float, hipStream_t);
int lora_bwd_mfma_f16dy(const void* dy16, const float* x, const float* t, const float* A, const float* B, float* dt, float* dA,
                        float* dB, float* dx, int rows, int width, int segw, int r, int nseg, unsigned seg_mask, float scale,
                        float p, uint64_t seed, uint32_t stream_base, uint32_t drow0, const uint16_t* keep_bits,
                        float* work, hipStream_t st, lora_reduce2_fn reduce);
int lora_bwd_mfma(const float* dy, const float* x, const float* t, const float* A, const float* B, float* dt, float* dA,
                  float* dB, float* dx, int rows, int width, int segw, int r, int nseg, unsigned seg_mask, float scale,
                  float p, uint64_t seed, uint32_t stream_base, uint32_t drow0, const uint16_t* keep_bits, float* work,
                  hipStream_t st, lora_reduce2_fn reduce);

// the dB and dA slice sums of one adapter backward in one launch (blocks [0, nblk0) take the first)
static void launch_reduce_slices2(const float* part0, float* out0, size_t n0, int slices0, float scale0, const float* part1,
                                  float* out1, size_t n1, int slices1, float scale1, hipStream_t st) {
  const unsigned nblk0 = (unsigned)((n0 + 63) / 64), nblk1 = (unsigned)((n1 + 63) / 64);
  hipLaunchKernelGGL(reduce_slices2_kernel, dim3(nblk0 + nblk1), dim3(1024), 0, st, part0, out0, n0, slices0, scale0, part1,
                     out1, n1, slices1, scale1, nblk0);
}

static bool use_lora_mfma() {
  static const int cfg = getenv("CLIPFS_LORA_MFMA") ? atoi(getenv("CLIPFS_LORA_MFMA")) : 1;  // 0: scalar kernels (A/B aid)
  return cfg != 0;
}

}  // namespace clipfs

using namespace clipfs;

// whether the forward can record its dropout masks as keep bits for the backward (matrix-core kernels, <= 4 segments)
extern "C" int clipfs_lora_keep_bits_ok(int width, int segw, int r, int nseg) {
  static const int cfg = getenv("CLIPFS_LORA_KEEP_BITS") ? atoi(getenv("CLIPFS_LORA_KEEP_BITS")) : 1;  // 0: Philox again in the backward (A/B aid)
  return (cfg != 0 && use_lora_mfma() && lora_mfma_ok(width, segw, r, nseg) && segw == width && nseg <= 4) ? 1 : 0;
}

extern "C" int clipfs_lora_down(const float* x, const float* A, float* t, int rows, int width, int r, int nseg,
                                unsigned seg_mask, float p, uint64_t seed, uint32_t stream_base, uint32_t drow0, void* keep_bits,
                                void* stream) {
  CLIPFS_REQUIRE(x && A && t, "lora_down: null pointer");
  CLIPFS_REQUIRE(!keep_bits || clipfs_lora_keep_bits_ok(width, width, r, nseg),
                 "lora_down: keep bits are recorded by the matrix-core kernels only (width %d r %d nseg %d)", width, r, nseg);
  CLIPFS_REQUIRE(rows > 0 && width > 0 && (width & 3) == 0 && width <= 256 * LORA_MAX_CHUNKS, "lora_down: width %d unsupported", width);
  CLIPFS_REQUIRE(r > 0 && r <= 64 && nseg > 0 && nseg <= 4 && nseg * r <= LORA_MAX_OUT, "lora_down: r %d nseg %d unsupported", r, nseg);
  CLIPFS_REQUIRE(p >= 0.f && p < 1.f, "lora_down: dropout p %f out of range", (double)p);
  CLIPFS_REQUIRE(aligned16(x) && aligned16(A), "lora_down: misaligned pointer");
  if (use_lora_mfma() && lora_mfma_ok(width, width, r, nseg))
    return lora_down_mfma(x, A, t, rows, width, r, nseg, seg_mask, p, seed, stream_base, drow0,
                          reinterpret_cast<uint16_t*>(keep_bits), (hipStream_t)stream);
  hipLaunchKernelGGL(lora_down_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, A, t, rows, width, r,
                     nseg, seg_mask, p, seed, stream_base, drow0);
  return launch_status();
}

extern "C" size_t clipfs_lora_bwd_work_floats(int rows, int width, int r, int nseg) {
  const int sr = lora_slice_rows(rows);
  const size_t slices = (size_t)(rows + sr - 1) / sr;
  // dB partials: slices * (nseg*segw) * r with segw <= 4*width (MLP never adapted; q/k/v/o segw == width)
  // dA partials: slices * nseg * r * width
  return slices * (size_t)nseg * r * width * 2 + 64;
}

template <int R>
static int lora_bwd_r(const float* dy, const float* x, const float* t, const float* A, const float* B, float* dt,
                      float* dA, float* dB, float* dx, int rows, int width, int segw, int nseg, unsigned seg_mask,
                      float scale, float p, uint64_t seed, uint32_t stream_base, uint32_t drow0, float* work, hipStream_t st) {
  const int sr = lora_slice_rows(rows);
  const int slices = (rows + sr - 1) / sr;
  const int cols = nseg * segw;
  hipLaunchKernelGGL((lora_dt_kernel<R>), dim3((rows + 3) / 4), dim3(256), 0, st, dy, B, dt, rows, segw, nseg, seg_mask,
                     scale);
  CLIPFS_CHECK(launch_status());
  // dB
  float* part_b = work;
  hipLaunchKernelGGL((lora_db_partial_kernel<R>), dim3((cols + 255) / 256, slices), dim3(256), 0, st, dy, t, part_b,
                     rows, cols, segw, nseg, sr);
  CLIPFS_CHECK(launch_status());
  const size_t nb = (size_t)cols * R;
  hipLaunchKernelGGL(reduce_slices_kernel, dim3((unsigned)((nb + 63) / 64)), dim3(1024), 0, st, part_b, dB, nb, slices,
                     scale);
  CLIPFS_CHECK(launch_status());
  // dA
  float* part_a = work + (size_t)slices * nb;
  const dim3 ga((width / 4 + 63) / 64, slices);
  switch (nseg) {
    case 1:
      hipLaunchKernelGGL((lora_da_partial_kernel<R, 1>), ga, dim3(64), 0, st, x, dt, part_a, rows, width, seg_mask, p,
                         seed, stream_base, drow0, sr);
      break;
    case 3:
      hipLaunchKernelGGL((lora_da_partial_kernel<R, 3>), ga, dim3(64), 0, st, x, dt, part_a, rows, width, seg_mask, p,
                         seed, stream_base, drow0, sr);
      break;
    default:
      set_error("lora_bwd: nseg %d unsupported (1 or 3)", nseg);
      return CLIPFS_EINVAL;
  }
  CLIPFS_CHECK(launch_status());
  const size_t na = (size_t)nseg * R * width;
  hipLaunchKernelGGL(reduce_slices_kernel, dim3((unsigned)((na + 63) / 64)), dim3(1024), 0, st, part_a, dA, na, slices,
                     1.0f);
  CLIPFS_CHECK(launch_status());
  if (dx) {
    hipLaunchKernelGGL(lora_dx_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, dt, A, dx, rows, width, R, nseg, seg_mask,
                       p, seed, stream_base, drow0);
    CLIPFS_CHECK(launch_status());
  }
  return CLIPFS_OK;
}

extern "C" int clipfs_lora_bwd(const float* dy, const float* x, const float* t, const float* A, const float* B,
                               float* dt, float* dA, float* dB, float* dx, int rows, int width, int segw, int r,
                               int nseg, unsigned seg_mask, float scale, float p, uint64_t seed, uint32_t stream_base, uint32_t drow0,
                               const void* keep_bits, float* work, void* stream) {
  CLIPFS_REQUIRE(dy && x && t && A && B && dt && dA && dB && work, "lora_bwd: null pointer");
  CLIPFS_REQUIRE(!keep_bits || (clipfs_lora_keep_bits_ok(width, segw, r, nseg) && aligned16(dy) && aligned16(dx ? dx : x)),
                 "lora_bwd: keep bits are read by the matrix-core kernels only (width %d r %d nseg %d)", width, r, nseg);
  CLIPFS_REQUIRE(rows > 0 && width > 0 && (width & 3) == 0 && segw == width, "lora_bwd: width %d segw %d unsupported (segw must equal width)", width, segw);
  CLIPFS_REQUIRE(p >= 0.f && p < 1.f, "lora_bwd: dropout p out of range");
  CLIPFS_REQUIRE(aligned16(x) && aligned16(A) && aligned16(work) && (!dx || aligned16(dx)), "lora_bwd: misaligned pointer");
  hipStream_t st = (hipStream_t)stream;
  if (use_lora_mfma() && lora_mfma_ok(width, segw, r, nseg) && aligned16(dy) && aligned16(dx ? dx : x))
    return lora_bwd_mfma(dy, x, t, A, B, dt, dA, dB, dx, rows, width, segw, r, nseg, seg_mask, scale, p, seed, stream_base, drow0,
                         reinterpret_cast<const uint16_t*>(keep_bits), work, st, launch_reduce_slices2);
#define CLIPFS_LORA_CASE(RR)                                                                                       \
  case RR:                                                                                                         \
    return lora_bwd_r<RR>(dy, x, t, A, B, dt, dA, dB, dx, rows, width, segw, nseg, seg_mask, scale, p, seed,       \
                          stream_base, drow0, work, st)
  switch (r) {
    CLIPFS_LORA_CASE(1);
    CLIPFS_LORA_CASE(2);
    CLIPFS_LORA_CASE(4);
    CLIPFS_LORA_CASE(8);
    CLIPFS_LORA_CASE(16);
    default:
      set_error("lora_bwd: rank %d unsupported (1, 2, 4, 8, 16)", r);
      return CLIPFS_EINVAL;
  }
#undef CLIPFS_LORA_CASE
}

// fp16 storage mode: the incoming gradient dy is read from its f16 image (what the dgrad GEMM consumes anyway) -- half
// the bytes of the two passes over dy, and the producer (clipfs_attention_f16_bwd) no longer has to write the fp32
// tensor at all.  Matrix-core kernels only: clipfs_lora_bwd_f16dy_ok says whether a shape is covered.
extern "C" int clipfs_lora_bwd_f16dy_ok(int width, int segw, int r, int nseg) {
  return (use_lora_mfma() && lora_mfma_ok(width, segw, r, nseg) && segw == width) ? 1 : 0;
}

extern "C" int clipfs_lora_bwd_f16dy(const void* dy16, const float* x, const float* t, const float* A, const float* B,
                                     float* dt, float* dA, float* dB, float* dx, int rows, int width, int segw, int r,
                                     int nseg, unsigned seg_mask, float scale, float p, uint64_t seed, uint32_t stream_base,
                                     uint32_t drow0, const void* keep_bits, float* work, void* stream) {
  CLIPFS_REQUIRE(dy16 && x && t && A && B && dt && dA && dB && work, "lora_bwd_f16dy: null pointer");
  CLIPFS_REQUIRE(rows > 0 && clipfs_lora_bwd_f16dy_ok(width, segw, r, nseg),
                 "lora_bwd_f16dy: width %d segw %d r %d nseg %d is outside the matrix-core kernels", width, segw, r, nseg);
  CLIPFS_REQUIRE(p >= 0.f && p < 1.f, "lora_bwd_f16dy: dropout p out of range");
  CLIPFS_REQUIRE(aligned16(dy16) && aligned16(x) && aligned16(A) && aligned16(work) && (!dx || aligned16(dx)),
                 "lora_bwd_f16dy: misaligned pointer");
  CLIPFS_REQUIRE(!keep_bits || clipfs_lora_keep_bits_ok(width, segw, r, nseg), "lora_bwd_f16dy: keep bits not covered");
  return lora_bwd_mfma_f16dy(dy16, x, t, A, B, dt, dA, dB, dx, rows, width, segw, r, nseg, seg_mask, scale, p, seed, stream_base,
                             drow0, reinterpret_cast<const uint16_t*>(keep_bits), work, (hipStream_t)stream, launch_reduce_slices2);
}
