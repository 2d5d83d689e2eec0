// Shared helpers for the clipfs HIP sources (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "clipfs.h"

namespace clipfs {

void set_error(const char* fmt, ...);

#define CLIPFS_REQUIRE(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      ::clipfs::set_error(__VA_ARGS__);      \
      return CLIPFS_EINVAL;                  \
    }                                        \
  } while (0)

inline int launch_status() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("kernel launch failed: %s", hipGetErrorString(e));
    return CLIPFS_HIPERR_BASE + (int)e;
  }
  return CLIPFS_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define CLIPFS_CHECK(expr)          \
  do {                              \
    int _rc = (expr);               \
    if (_rc != CLIPFS_OK) return _rc; \
  } while (0)

// ---- wave64 reductions: DPP butterfly inside each row of 16 lanes, row_bcast across rows, one
// v_readlane of lane 63 to broadcast (no LDS crossbar: ds_bpermute shuffles cost ~6 dependent LDS round
// trips per reduction, the DPP form ~6 VALU slots).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float identity, float v) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1, 0xf>(0.f, v);   // quad_perm [1,0,3,2]
  v += dpp_f<0x4E, 0xf>(0.f, v);   // quad_perm [2,3,0,1]
  v += dpp_f<0x141, 0xf>(0.f, v);  // row_half_mirror
  v += dpp_f<0x140, 0xf>(0.f, v);  // row_mirror        -> every lane: sum of its row of 16
  v += dpp_f<0x142, 0xa>(0.f, v);  // row_bcast:15 into rows 1,3
  v += dpp_f<0x143, 0xc>(0.f, v);  // row_bcast:31 into rows 2,3 -> lanes 48..63: total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  const float ninf = -INFINITY;
  v = fmaxf(v, dpp_f<0xB1, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<0x4E, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<0x141, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<0x140, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<0x142, 0xa>(ninf, v));
  v = fmaxf(v, dpp_f<0x143, 0xc>(ninf, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---- Philox4x32-10 dropout stream (bit-identical to oracle/clip_oracle.py:dropout_keep_mask) ----
// element (row, col): counter = (col / 4, row, stream, 0), key = (seed_lo, seed_hi), word col % 4.
struct u32x4 {
  uint32_t x, y, z, w;
};
__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: the
    // integer multiplies are quarter rate and were ~half of the LoRA kernels' instruction time
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0;
    c1 = lo1;
    c2 = n2;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return {c0, c1, c2, c3};
}
__device__ __forceinline__ uint32_t dropout_threshold(float p) {
  double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
}
// multipliers (0 or inv_keep) for the 4 consecutive columns starting at col4*4 of `row`
__device__ __forceinline__ float4 dropout_scale4(uint64_t seed, uint32_t stream, uint32_t row, uint32_t col4,
                                                 uint32_t thr, float inv_keep) {
  u32x4 r = philox4x32_10(col4, row, stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
  return make_float4(r.x >= thr ? inv_keep : 0.f, r.y >= thr ? inv_keep : 0.f, r.z >= thr ? inv_keep : 0.f,
                     r.w >= thr ? inv_keep : 0.f);
}

// The same multipliers from / to 4 stored keep bits (bit e = column 4*col4 + e kept): the forward pass of an adapted
// projection records its masks as 4 bits per segment and float4 so that the backward reads 2 bytes per float4 instead of
// evaluating Philox again (three 10-round evaluations per float4 made the adapter's dA / dx kernels VALU-bound).
__device__ __forceinline__ float4 keep_scale4(uint32_t bits, float inv_keep) {
  return make_float4((bits & 1u) ? inv_keep : 0.f, (bits & 2u) ? inv_keep : 0.f, (bits & 4u) ? inv_keep : 0.f,
                     (bits & 8u) ? inv_keep : 0.f);
}
__device__ __forceinline__ uint32_t keep_bits4(const float4& mk) {
  return (mk.x != 0.f ? 1u : 0u) | (mk.y != 0.f ? 2u : 0u) | (mk.z != 0.f ? 4u : 0u) | (mk.w != 0.f ? 8u : 0u);
}

__device__ __forceinline__ float quick_gelu(float u) { return u / (1.f + __expf(-1.702f * u)); }
__device__ __forceinline__ float quick_gelu_grad(float u) {
  const float s = 1.f / (1.f + __expf(-1.702f * u));
  return s * (1.f + 1.702f * u * (1.f - s));
}

}  // namespace clipfs
