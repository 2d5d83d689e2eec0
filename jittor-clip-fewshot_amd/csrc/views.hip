// GPU-side view generation for MTA test-time augmentation.  In the reference the 1 + 512
// views per image are made by 8 CPU workers with PIL (ood.py:946-958,1084-1089; jclip/clip.py:130-144):
//     centre view : Resize(256, BICUBIC) -> CenterCrop(224) -> ImageNormalize -> ToTensor
//     crops       : RandomResizedCrop(224, scale, BILINEAR) -> RandomHorizontalFlip -> ImageNormalize -> ToTensor
// and that stage, not the ViT, dominates TTA wall-clock.  Here the source image stays in HBM as uint8 HWC and one
// kernel writes every normalised fp32 view [n, 3, S, S] directly.  The resampling is PIL's 8-bit path restated
// exactly (Pillow src/libImaging/Resample.c: separable filter with support scaled by max(in/out, 1), double
// coefficients normalised then quantised to 22 fractional bits, horizontal pass rounded to uint8 before the vertical
// pass) so the pixels are bit-identical to Image.crop(...).resize(...); tests/test_views_gpu.py checks that against
// PIL itself.  The crop boxes / flips are sampled on the host (clipfs/views.py) -- integer work, microseconds.
#include "common.h"

namespace clipfs {

constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int VIEW_KMAX = 24;  // taps per axis: ceil(support) * 2 + 1, support = filter support * max(scale, 1)

struct ViewRec {  // one row of the int32 [n, 10] descriptor
  int top, left, h, w;      // crop box in the source image
  int flip;                 // horizontal flip after the resize
  int out_w, out_h;         // size the crop is resized to
  int win_x, win_y;         // top-left of the S x S window taken from the resized image (CenterCrop)
  int filter;               // 0 bilinear, 1 bicubic (a = -0.5)
};

__device__ __forceinline__ double pil_filter(int kind, double x) {
  if (x < 0.0) x = -x;
  if (kind == 0) return x < 1.0 ? 1.0 - x : 0.0;
  const double a = -0.5;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc for ONE output coordinate `xx` of an axis (in_size -> out_size)
__device__ __forceinline__ int pil_coeffs(int kind, int in_size, int out_size, int xx, int* __restrict__ kq, int& xmin) {
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = (kind == 0 ? 1.0 : 2.0) * filterscale;
  const double center = 0.0 + (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (xmax > VIEW_KMAX) xmax = VIEW_KMAX;  // host rejects scales that would need more taps
  double k[VIEW_KMAX];
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    const double w = pil_filter(kind, (x + xmin - center + 0.5) * ss);
    k[x] = w;
    ww += w;
  }
  for (int x = 0; x < xmax; ++x) {
    double v = k[x];
    if (ww != 0.0) v /= ww;
    kq[x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
  }
  return xmax;
}

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// one thread = one output pixel (all 3 channels)
__global__ __launch_bounds__(256) void tta_views_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                        const ViewRec* __restrict__ recs, int S,
                                                        const float* __restrict__ mean, const float* __restrict__ stdv,
                                                        float* __restrict__ out) {
  const int v = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= S * S) return;
  const ViewRec r = recs[v];
  const int oy = pix / S, ox = pix % S;
  const int sx = r.flip ? S - 1 - ox : ox;  // flip acts on the final S x S view
  int kx[VIEW_KMAX], ky[VIEW_KMAX];
  int xmin, ymin;
  const int nx = pil_coeffs(r.filter, r.w, r.out_w, r.win_x + sx, kx, xmin);
  const int ny = pil_coeffs(r.filter, r.h, r.out_h, r.win_y + oy, ky, ymin);
  int acc[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
  for (int yy = 0; yy < ny; ++yy) {
    const uint8_t* row = img + ((size_t)(r.top + ymin + yy) * W + (r.left + xmin)) * 3;
    int h0 = 1 << (PRECISION_BITS - 1), h1 = h0, h2 = h0;
    for (int xx = 0; xx < nx; ++xx) {
      h0 += (int)row[3 * xx + 0] * kx[xx];
      h1 += (int)row[3 * xx + 1] * kx[xx];
      h2 += (int)row[3 * xx + 2] * kx[xx];
    }
    acc[0] += clip8(h0) * ky[yy];  // the horizontal pass is rounded to uint8 before the vertical pass (as in Pillow)
    acc[1] += clip8(h1) * ky[yy];
    acc[2] += clip8(h2) * ky[yy];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float px = (float)clip8(acc[c]);
    // jittor ImageNormalize on a PIL image: (u8 - mean * 255) * ((1 / 255) / std)
    out[(((size_t)v * 3 + c) * S + oy) * S + ox] = (px - mean[c] * 255.f) * ((1.f / 255.f) / stdv[c]);
  }
}

}  // namespace clipfs

using namespace clipfs;

extern "C" int clipfs_tta_views(const uint8_t* image, int height, int width, const int32_t* recs, int n_views, int out_size,
                                const float* mean, const float* stdv, float* out, void* stream) {
  CLIPFS_REQUIRE(image && recs && mean && stdv && out, "tta_views: null pointer");
  CLIPFS_REQUIRE(height > 0 && width > 0 && n_views > 0 && out_size > 0, "tta_views: bad dims");
  hipLaunchKernelGGL(tta_views_kernel, dim3((out_size * out_size + 255) / 256, n_views), dim3(256), 0, (hipStream_t)stream,
                     image, height, width, reinterpret_cast<const ViewRec*>(recs), out_size, mean, stdv, out);
  return launch_status();
}
