// Data-movement kernels of the MoCo-v3 ResNet-50 auxiliary branch (slow_pace.py:1237-1271 load_moco,
// :1677-1680 loss_aux; the model itself is jittor.models.resnet.resnet50 = torchvision's ResNet-50 v1.5).
// The branch is a FROZEN feature extractor run forward only.  Every convolution is a GEMM on the fp32-MFMA kernel
// (gemm.hip): activations are NHWC, so a 1x1 / stride-1 convolution IS a GEMM on the activation tensor, and the
// others read an im2col matrix written by the kernel below; BatchNorm (inference statistics) is folded into the
// convolution's weights and bias on the host, ReLU (after the residual add) is the GEMM epilogue's act 3.
// What is left for this file: the NCHW -> NHWC input permute, im2col, the 3x3/2 max pool and the global average pool
// -- all HBM-bound gathers with 16-byte accesses along the channel axis.
#include "common.h"

namespace clipfs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// x [N, C, H, W] -> y [N, H, W, C]
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C,
                                                            int H, int W) {
  const size_t total = (size_t)N * H * W * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    size_t p = i / C;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const int n = (int)(p / H);
    y[i] = x[(((size_t)n * C + c) * H + h) * W + w];
  }
}

// x [N, H, W, C] -> col [N * Ho * Wo, Kp], column (ky, kx, c) (c fastest), columns [kh*kw*C, Kp) zero.
// CV = channels moved per thread (4 when C % 4 == 0: 16-byte loads / stores, else 1)
template <int CV>
__global__ __launch_bounds__(256) void im2col_nhwc_kernel(const float* __restrict__ x, float* __restrict__ col, int N, int H,
                                                           int W, int C, int kh, int kw, int stride, int pad, int Ho, int Wo,
                                                           int Kp) {
  const int kcols = Kp / CV;
  const size_t total = (size_t)N * Ho * Wo * kcols;
  const int K = kh * kw * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int kc = (int)(i % kcols) * CV;
    size_t m = i / kcols;
    const int wo = (int)(m % Wo);
    size_t q = m / Wo;
    const int ho = (int)(q % Ho);
    const int n = (int)(q / Ho);
    float v[CV];
#pragma unroll
    for (int e = 0; e < CV; ++e) v[e] = 0.f;
    if (kc < K) {
      const int c = kc % C, t = kc / C;
      const int kx = t % kw, ky = t / kw;
      const int hi = ho * stride - pad + ky, wi = wo * stride - pad + kx;
      if (hi >= 0 && hi < H && wi >= 0 && wi < W) {
        const float* src = x + (((size_t)n * H + hi) * W + wi) * C + c;
        if (CV == 4) {
          const f32x4 u = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
          for (int e = 0; e < CV; ++e) v[e] = u[e];
        } else {
          v[0] = src[0];
        }
      }
    }
    float* dst = col + m * Kp + kc;
    if (CV == 4)
      *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1 % CV], v[2 % CV], v[3 % CV]};
    else
      dst[0] = v[0];
  }
}

// 3x3 / stride 2 / pad 1 max pool on NHWC (C % 4 == 0); padding never wins (-inf)
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H,
                                                                 int W, int C, int Ho, int Wo) {
  const int c4n = C / 4;
  const size_t total = (size_t)N * Ho * Wo * c4n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4;
    size_t m = i / c4n;
    const int wo = (int)(m % Wo);
    size_t q = m / Wo;
    const int ho = (int)(q % Ho);
    const int n = (int)(q / Ho);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int hi = 2 * ho - 1 + ky, wi = 2 * wo - 1 + kx;
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) {
          const f32x4 u = *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + hi) * W + wi) * C + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) best[e] = fmaxf(best[e], u[e]);
        }
      }
    *reinterpret_cast<f32x4*>(y + m * C + c) = best;
  }
}

// y[n, c] = mean over the HW positions of x[n, :, c]   (one wave per (n, 64-channel group): lanes = channels)
__global__ __launch_bounds__(64) void global_avgpool_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int HW,
                                                                 int C) {
  const int groups = (C + 63) / 64;
  const int n = blockIdx.x / groups, c = (blockIdx.x % groups) * 64 + threadIdx.x;
  if (c >= C) return;
  const float* p = x + (size_t)n * HW * C + c;
  float s = 0.f;
  for (int i = 0; i < HW; ++i) s += p[(size_t)i * C];  // sequential over positions: the summation order of a plain loop
  y[(size_t)n * C + c] = s / (float)HW;
}

static inline unsigned rn_grid(size_t total) {
  const size_t b = (total + 255) / 256;
  return (unsigned)(b > 65535 ? 65535 : (b < 1 ? 1 : b));
}

}  // namespace clipfs

using namespace clipfs;

extern "C" int clipfs_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, void* stream) {
  CLIPFS_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad args");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(rn_grid((size_t)N * C * H * W)), dim3(256), 0, (hipStream_t)stream, x, y, N, C, H,
                     W);
  return launch_status();
}

extern "C" int clipfs_im2col_nhwc(const float* x, float* col, int N, int H, int W, int C, int kh, int kw, int stride, int pad,
                                  int Kp, void* stream) {
  CLIPFS_REQUIRE(x && col && N > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0,
                 "im2col_nhwc: bad args");
  CLIPFS_REQUIRE(Kp >= kh * kw * C && (Kp & 3) == 0, "im2col_nhwc: Kp %d must be >= kh*kw*C and a multiple of 4", Kp);
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  CLIPFS_REQUIRE(Ho > 0 && Wo > 0, "im2col_nhwc: empty output");
  hipStream_t st = (hipStream_t)stream;
  if ((C & 3) == 0 && aligned16(x) && aligned16(col)) {
    const size_t total = (size_t)N * Ho * Wo * (Kp / 4);
    hipLaunchKernelGGL((im2col_nhwc_kernel<4>), dim3(rn_grid(total)), dim3(256), 0, st, x, col, N, H, W, C, kh, kw, stride, pad,
                       Ho, Wo, Kp);
  } else {
    const size_t total = (size_t)N * Ho * Wo * Kp;
    hipLaunchKernelGGL((im2col_nhwc_kernel<1>), dim3(rn_grid(total)), dim3(256), 0, st, x, col, N, H, W, C, kh, kw, stride, pad,
                       Ho, Wo, Kp);
  }
  return launch_status();
}

extern "C" int clipfs_maxpool3x3s2_nhwc(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  CLIPFS_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && aligned16(x) && aligned16(y),
                 "maxpool3x3s2_nhwc: bad args (C %% 4 == 0, 16-byte aligned)");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3x3s2_nhwc_kernel, dim3(rn_grid((size_t)N * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     x, y, N, H, W, C, Ho, Wo);
  return launch_status();
}

extern "C" int clipfs_global_avgpool_nhwc(const float* x, float* y, int N, int HW, int C, void* stream) {
  CLIPFS_REQUIRE(x && y && N > 0 && HW > 0 && C > 0, "global_avgpool_nhwc: bad args");
  hipLaunchKernelGGL(global_avgpool_nhwc_kernel, dim3(N * ((C + 63) / 64)), dim3(64), 0, (hipStream_t)stream, x, y, HW, C);
  return launch_status();
}
