// LayerNorm forward / backward and row L2-normalisation: HBM-bound, one wave64 per row,
// 16-byte loads, statistics by cross-lane shuffles (no LDS).  Algorithmic bytes per row:
// fwd 2*width*4 (+8 for mean/rstd), bwd 3..4*width*4.
#include "common.h"

#include <stdlib.h>

namespace clipfs {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int LN_MAX_CHUNKS = 8;  // 8 * 64 lanes * 4 floats = width <= 2048

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* x, int ldx,  // x may alias y (in place)
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int rows, int width, float eps,
                                                            _Float16* __restrict__ y16 = nullptr) {  // optional f16 copy (or only output)
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = width >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * ldx);
  float4 v[LN_MAX_CHUNKS];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      v[i] = xr[c];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)width;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)width + eps);
  float4* yr = y ? reinterpret_cast<float4*>(y + (size_t)row * width) : nullptr;
  f16x4* yh = y16 ? reinterpret_cast<f16x4*>(y16 + (size_t)row * width) : nullptr;
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const float4 g = g4[c], b = b4[c];
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      if (yr) yr[c] = o;
      if (yh) yh[c] = f16x4{(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
    }
  }
  if (mean_out && lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

// LayerNorm forward with the adapter's down-projection in the same pass (ranks up to 4 on q, k, v: cfg-2 / cfg-3):
//   y = LN(x) ;  t[row, s r + j] = sum_k drop_s(y)[row, k] A[s r + j, k]      (lora_train_vlp.py:296-306: lora_A(dropout(x)))
// The normalised row is in the wave's registers when it is stored, so the separate clipfs_lora_down launch (a second
// read of y plus, at the per-rank sizes of the 8-GPU step, ~10 us of launch for ~2 us of work) is folded in: per float4
// of y one Philox call per adapted segment (the same (stream, row, column group) counter as the stand-alone kernel, so
// the backward's masks agree), NSEG * R dot products against A (12 rows of <= 8 KiB: cache resident), NSEG * R wave sums.
template <int NSEG, int R>
__global__ __launch_bounds__(256) void layernorm_fwd_lora_kernel(const float* x, int ldx, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* y,
                                                                 float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                 int rows, int width, float eps, _Float16* __restrict__ y16,
                                                                 const float* __restrict__ A, float* __restrict__ t,
                                                                 unsigned seg_mask, float p, uint64_t seed,
                                                                 uint32_t stream_base, uint32_t drow0,
                                                                 uint16_t* __restrict__ keep_bits) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = width >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * ldx);
  float4 v[LN_MAX_CHUNKS];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      v[i] = xr[c];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)width;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)width + eps);
  float4* yr = y ? reinterpret_cast<float4*>(y + (size_t)row * width) : nullptr;
  f16x4* yh = y16 ? reinterpret_cast<f16x4*>(y16 + (size_t)row * width) : nullptr;
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  const float4* A4 = reinterpret_cast<const float4*>(A);
  const bool drop = p > 0.f && seed != 0;
  const uint32_t thr = dropout_threshold(p);
  const float inv_keep = 1.f / (1.f - p);
  float acc[NSEG][R];
#pragma unroll
  for (int sg = 0; sg < NSEG; ++sg)
#pragma unroll
    for (int j = 0; j < R; ++j) acc[sg][j] = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const float4 g = g4[c], b = b4[c];
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      if (yr) yr[c] = o;
      if (yh) yh[c] = f16x4{(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
      uint32_t kb = 0u;
#pragma unroll
      for (int sg = 0; sg < NSEG; ++sg) {
        if (!((seg_mask >> sg) & 1u)) continue;
        float4 xs = o;
        if (drop) {
          const float4 mk = dropout_scale4(seed, stream_base + sg, drow0 + (uint32_t)row, (uint32_t)c, thr, inv_keep);
          xs.x *= mk.x;
          xs.y *= mk.y;
          xs.z *= mk.z;
          xs.w *= mk.w;
          kb |= keep_bits4(mk) << (4 * sg);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const float4 a = A4[(size_t)(sg * R + j) * nch + c];
          acc[sg][j] = fmaf(xs.w, a.w, fmaf(xs.z, a.z, fmaf(xs.y, a.y, fmaf(xs.x, a.x, acc[sg][j]))));
        }
      }
      if (keep_bits && drop) keep_bits[(size_t)row * nch + c] = (uint16_t)kb;  // the masks, for the backward
    }
  }
#pragma unroll
  for (int sg = 0; sg < NSEG; ++sg)
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const float tot = wave_sum(acc[sg][j]);
      if (lane == 0) t[(size_t)row * (NSEG * R) + sg * R + j] = tot;  // segments outside seg_mask: 0
    }
  if (mean_out && lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            int ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in,
                                                            const float* dres, float* dx,  // may alias (in-place residual add)
                                                            int lddx, int rows, int width,
                                                            _Float16* __restrict__ dx16 = nullptr) {  // optional f16 copy [rows, width]
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = width >> 2;
  const float mean = mean_in[row], rstd = rstd_in[row];
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * ldx);
  const float4* dyr = reinterpret_cast<const float4*>(dy + (size_t)row * width);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  float4 xh[LN_MAX_CHUNKS], gd[LN_MAX_CHUNKS];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const float4 xv = xr[c], d = dyr[c], g = g4[c];
      xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
      gd[i] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
      s1 += (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
      s2 += (gd[i].x * xh[i].x + gd[i].y * xh[i].y) + (gd[i].z * xh[i].z + gd[i].w * xh[i].w);
    }
  }
  const float c1 = wave_sum(s1) / (float)width;
  const float c2 = wave_sum(s2) / (float)width;
  float4* dxr = reinterpret_cast<float4*>(dx + (size_t)row * lddx);
  const float4* rr = dres ? reinterpret_cast<const float4*>(dres + (size_t)row * lddx) : nullptr;
  f16x4* dxh = dx16 ? reinterpret_cast<f16x4*>(dx16 + (size_t)row * width) : nullptr;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      float4 o;
      o.x = rstd * (gd[i].x - c1 - xh[i].x * c2);
      o.y = rstd * (gd[i].y - c1 - xh[i].y * c2);
      o.z = rstd * (gd[i].z - c1 - xh[i].z * c2);
      o.w = rstd * (gd[i].w - c1 - xh[i].w * c2);
      if (rr) {
        const float4 r = rr[c];
        o.x += r.x;
        o.y += r.y;
        o.z += r.z;
        o.w += r.w;
      }
      dxr[c] = o;
      if (dxh) dxh[c] = f16x4{(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
    }
  }
}

__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ inv_out, int rows, int width) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = width >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * width);
  float4 v[LN_MAX_CHUNKS];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      v[i] = xr[c];
      s += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
  }
  const float inv = 1.f / sqrtf(wave_sum(s));
  float4* yr = reinterpret_cast<float4*>(y + (size_t)row * width);
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) yr[c] = make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
  }
  if (inv_out && lane == 0) inv_out[row] = inv;
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ inv_in, float* __restrict__ dx,
                                                         int rows, int width) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = width >> 2;
  const float4* yr = reinterpret_cast<const float4*>(y + (size_t)row * width);
  const float4* dr = reinterpret_cast<const float4*>(dy + (size_t)row * width);
  float4 yv[LN_MAX_CHUNKS], dv[LN_MAX_CHUNKS];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      yv[i] = yr[c];
      dv[i] = dr[c];
      s += (yv[i].x * dv[i].x + yv[i].y * dv[i].y) + (yv[i].z * dv[i].z + yv[i].w * dv[i].w);
    }
  }
  const float dot = wave_sum(s);
  const float inv = inv_in[row];
  float4* xr = reinterpret_cast<float4*>(dx + (size_t)row * width);
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nch)
      xr[c] = make_float4(inv * (dv[i].x - yv[i].x * dot), inv * (dv[i].y - yv[i].y * dot),
                          inv * (dv[i].z - yv[i].z * dot), inv * (dv[i].w - yv[i].w * dot));
  }
}

static int check_rows(const char* what, int rows, int width) {
  CLIPFS_REQUIRE(rows > 0 && width > 0 && (width & 3) == 0 && width <= 256 * LN_MAX_CHUNKS,
                 "%s: rows %d width %d unsupported (width must be a multiple of 4, <= %d)", what, rows, width,
                 256 * LN_MAX_CHUNKS);
  return CLIPFS_OK;
}

}  // namespace clipfs

using namespace clipfs;

extern "C" int clipfs_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* y,
                                    float* mean, float* rstd, int rows, int width, float eps, void* stream) {
  CLIPFS_CHECK(check_rows("layernorm_fwd", rows, width));
  CLIPFS_REQUIRE(x && gamma && beta && y, "layernorm_fwd: null pointer");
  CLIPFS_REQUIRE((mean == nullptr) == (rstd == nullptr), "layernorm_fwd: mean and rstd must both be given or both NULL");
  CLIPFS_REQUIRE(ldx >= width && (ldx & 3) == 0 && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta),
                 "layernorm_fwd: alignment");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta,
                     y, mean, rstd, rows, width, eps);
  return launch_status();
}

extern "C" int clipfs_layernorm_bwd(const float* dy, const float* x, int ldx, const float* gamma, const float* mean,
                                    const float* rstd, const float* dres, float* dx, int lddx, int rows, int width,
                                    void* stream) {
  CLIPFS_CHECK(check_rows("layernorm_bwd", rows, width));
  CLIPFS_REQUIRE(dy && x && gamma && mean && rstd && dx, "layernorm_bwd: null pointer");
  CLIPFS_REQUIRE(ldx >= width && (ldx & 3) == 0 && lddx >= width && (lddx & 3) == 0 && aligned16(x) && aligned16(dy) &&
                     aligned16(dx) && aligned16(gamma) && (!dres || aligned16(dres)),
                 "layernorm_bwd: alignment");
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, x, ldx, gamma,
                     mean, rstd, dres, dx, lddx, rows, width);
  return launch_status();
}

extern "C" int clipfs_l2norm_fwd(const float* x, float* y, float* inv_norm, int rows, int width, void* stream) {
  CLIPFS_CHECK(check_rows("l2norm_fwd", rows, width));
  CLIPFS_REQUIRE(x && y && aligned16(x) && aligned16(y), "l2norm_fwd: null or misaligned pointer");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, inv_norm, rows,
                     width);
  return launch_status();
}

extern "C" int clipfs_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int rows, int width,
                                 void* stream) {
  CLIPFS_CHECK(check_rows("l2norm_bwd", rows, width));
  CLIPFS_REQUIRE(dy && y && inv_norm && dx && aligned16(dy) && aligned16(y) && aligned16(dx),
                 "l2norm_bwd: null or misaligned pointer");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, y, inv_norm, dx,
                     rows, width);
  return launch_status();
}

// fp16 storage mode: the same kernels with an f16 copy of the result for the GEMM that consumes it (y / the fp32
// result may be NULL in the forward when only the f16 operand is needed).
extern "C" int clipfs_layernorm_fwd_f16(const float* x, int ldx, const float* gamma, const float* beta, float* y, void* y16,
                                        float* mean, float* rstd, int rows, int width, float eps, void* stream) {
  CLIPFS_CHECK(check_rows("layernorm_fwd_f16", rows, width));
  CLIPFS_REQUIRE(x && gamma && beta && (y || y16), "layernorm_fwd_f16: null pointer");
  CLIPFS_REQUIRE((mean == nullptr) == (rstd == nullptr), "layernorm_fwd_f16: mean and rstd must both be given or both NULL");
  CLIPFS_REQUIRE(ldx >= width && (ldx & 3) == 0 && aligned16(x) && (!y || aligned16(y)) && (!y16 || aligned16(y16)) &&
                     aligned16(gamma) && aligned16(beta), "layernorm_fwd_f16: alignment");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta,
                     y, mean, rstd, rows, width, eps, reinterpret_cast<_Float16*>(y16));
  return launch_status();
}

extern "C" int clipfs_layernorm_bwd_f16(const float* dy, const float* x, int ldx, const float* gamma, const float* mean,
                                        const float* rstd, const float* dres, float* dx, void* dx16, int lddx, int rows,
                                        int width, void* stream) {
  CLIPFS_CHECK(check_rows("layernorm_bwd_f16", rows, width));
  CLIPFS_REQUIRE(dy && x && gamma && mean && rstd && dx, "layernorm_bwd_f16: null pointer");
  CLIPFS_REQUIRE(ldx >= width && (ldx & 3) == 0 && lddx >= width && (lddx & 3) == 0 && aligned16(x) && aligned16(dy) &&
                     aligned16(dx) && aligned16(gamma) && (!dres || aligned16(dres)) && (!dx16 || aligned16(dx16)),
                 "layernorm_bwd_f16: alignment");
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, x, ldx, gamma,
                     mean, rstd, dres, dx, lddx, rows, width, reinterpret_cast<_Float16*>(dx16));
  return launch_status();
}

// LayerNorm + adapter down-projection in one pass; covered shapes: 3 segments (q, k, v), rank 1, 2 or 4
extern "C" int clipfs_layernorm_fwd_lora_ok(int width, int r, int nseg) {
  static const int cfg = getenv("CLIPFS_LN_LORA") ? atoi(getenv("CLIPFS_LN_LORA")) : 1;  // 0: separate launches (A/B aid)
  return (cfg != 0 && nseg == 3 && (r == 1 || r == 2 || r == 4) && width > 0 && (width & 3) == 0 && width <= 256 * LN_MAX_CHUNKS) ? 1 : 0;
}

extern "C" int clipfs_layernorm_fwd_lora(const float* x, int ldx, const float* gamma, const float* beta, float* y, void* y16,
                                         float* mean, float* rstd, int rows, int width, float eps, const float* A, float* t,
                                         int r, int nseg, unsigned seg_mask, float p, uint64_t seed, uint32_t stream_base,
                                         uint32_t drow0, void* keep_bits, void* stream) {
  CLIPFS_CHECK(check_rows("layernorm_fwd_lora", rows, width));
  CLIPFS_REQUIRE(x && gamma && beta && (y || y16) && A && t, "layernorm_fwd_lora: null pointer");
  CLIPFS_REQUIRE((mean == nullptr) == (rstd == nullptr), "layernorm_fwd_lora: mean and rstd must both be given or both NULL");
  CLIPFS_REQUIRE(clipfs_layernorm_fwd_lora_ok(width, r, nseg), "layernorm_fwd_lora: width %d r %d nseg %d is not covered", width, r, nseg);
  CLIPFS_REQUIRE(p >= 0.f && p < 1.f, "layernorm_fwd_lora: dropout p %f out of range", (double)p);
  CLIPFS_REQUIRE(ldx >= width && (ldx & 3) == 0 && aligned16(x) && (!y || aligned16(y)) && (!y16 || aligned16(y16)) &&
                     aligned16(gamma) && aligned16(beta) && aligned16(A), "layernorm_fwd_lora: alignment");
  const dim3 grid((rows + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  _Float16* h = reinterpret_cast<_Float16*>(y16);
  uint16_t* kb = reinterpret_cast<uint16_t*>(keep_bits);
  switch (r) {
    case 1:
      hipLaunchKernelGGL((layernorm_fwd_lora_kernel<3, 1>), grid, dim3(256), 0, st, x, ldx, gamma, beta, y, mean, rstd, rows, width,
                         eps, h, A, t, seg_mask, p, seed, stream_base, drow0, kb);
      break;
    case 2:
      hipLaunchKernelGGL((layernorm_fwd_lora_kernel<3, 2>), grid, dim3(256), 0, st, x, ldx, gamma, beta, y, mean, rstd, rows, width,
                         eps, h, A, t, seg_mask, p, seed, stream_base, drow0, kb);
      break;
    default:
      hipLaunchKernelGGL((layernorm_fwd_lora_kernel<3, 4>), grid, dim3(256), 0, st, x, ldx, gamma, beta, y, mean, rstd, rows, width,
                         eps, h, A, t, seg_mask, p, seed, stream_base, drow0, kb);
      break;
  }
  return launch_status();
}
