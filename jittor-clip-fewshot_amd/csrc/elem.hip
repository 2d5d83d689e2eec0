// Token assembly, classifier head, loss, top-k and optimiser kernels (all HBM/latency bound,
// tiny next to the GEMMs; each reads its operands once with 16-byte accesses where rows allow).
#include "common.h"

namespace clipfs {

// ---- ViT special tokens: class row and VPT rows (jclip/model.py:109-114, model1.py:192-194) ----
__global__ __launch_bounds__(256) void vit_fill_special_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                                               const float* __restrict__ pos,
                                                               const float* __restrict__ vpt, int batch, int tokens,
                                                               int n_patch, int n_vpt, int width) {
  const int per_img = (1 + n_vpt) * width;
  const size_t total = (size_t)batch * per_img;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int b = (int)(i / per_img), rem = (int)(i % per_img);
    const int r = rem / width, c = rem % width;
    if (r == 0)
      x[((size_t)b * tokens) * width + c] = cls[c] + pos[c];
    else
      x[((size_t)b * tokens + n_patch + r) * width + c] = vpt[(size_t)(r - 1) * width + c];
  }
}

// ---- text embedding (jclip/model.py:203-205; prompt tokens slow_pace.py:185-199) ----
__global__ __launch_bounds__(256) void text_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                         const float* __restrict__ pos, const float* __restrict__ ctx,
                                                         int n_ctx, float* __restrict__ x, int n, int seq, int width) {
  const int nch = width >> 2;
  const size_t total = (size_t)n * seq * nch;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % nch);
    const size_t tok = i / nch;
    const int l = (int)(tok % seq);
    const float4* src;
    if (ctx && l >= 1 && l <= n_ctx)
      src = reinterpret_cast<const float4*>(ctx + (size_t)(l - 1) * width);
    else
      src = reinterpret_cast<const float4*>(table + (size_t)ids[tok] * width);
    const float4 e = src[c];
    const float4 pe = reinterpret_cast<const float4*>(pos + (size_t)l * width)[c];
    reinterpret_cast<float4*>(x + tok * width)[c] = make_float4(e.x + pe.x, e.y + pe.y, e.z + pe.z, e.w + pe.w);
  }
}

// dctx[i, :] += sum_c dx[c, first + i, :]
__global__ __launch_bounds__(256) void text_ctx_grad_kernel(const float* __restrict__ dx, float* __restrict__ dctx, int n,
                                                            int seq, int width, int n_ctx, int first) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_ctx * width) return;
  const int r = i / width, c = i % width;
  float acc = 0.f;
  for (int k = 0; k < n; ++k) acc += dx[((size_t)k * seq + first + r) * width + c];
  dctx[i] += acc;
}

// EOT row = first position of the largest token id (jclip/model.py:213-214)
__global__ __launch_bounds__(64) void gather_eot_kernel(const float* __restrict__ x, const int64_t* __restrict__ ids,
                                                        float* __restrict__ out, int32_t* __restrict__ idx_out, int seq,
                                                        int width) {
  const int c = blockIdx.x, lane = threadIdx.x;
  long long best = -1;
  int besti = 0;
  for (int l = lane; l < seq; l += 64) {
    const long long v = ids[(size_t)c * seq + l];
    if (v > best) {
      best = v;
      besti = l;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const long long ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(besti, off, 64);
    if (ov > best || (ov == best && oi < besti)) {
      best = ov;
      besti = oi;
    }
  }
  if (idx_out && lane == 0) idx_out[c] = besti;
  const float* src = x + ((size_t)c * seq + besti) * width;
  for (int k = lane; k < width; k += 64) out[(size_t)c * width + k] = src[k];
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ dy, const int32_t* __restrict__ idx,
                                                           float* __restrict__ dx, int n, int seq, int width) {
  const size_t total = (size_t)n * seq * width;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % width);
    const size_t tok = i / width;
    const int c = (int)(tok / seq), l = (int)(tok % seq);
    dx[i] = (l == idx[c]) ? dy[(size_t)c * width + k] : 0.f;
  }
}

// out[c, :] = src[(c * seq + idx[c]) * ld + 0..width)  -- one row per sequence (the class token / the EOT token)
__global__ __launch_bounds__(256) void gather_seq_rows_kernel(const float* __restrict__ src, size_t ld,
                                                              const int32_t* __restrict__ idx, float* __restrict__ out,
                                                              int n, int seq, int width) {
  const size_t total = (size_t)n * width;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % width), c = (int)(i / width);
    out[i] = src[((size_t)c * seq + idx[c]) * ld + k];
  }
}

// dx[(c * seq + idx[c]) * width + k] += src[c, k]
__global__ __launch_bounds__(256) void add_seq_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                           float* __restrict__ dx, int n, int seq, int width) {
  const size_t total = (size_t)n * width;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % width), c = (int)(i / width);
    dx[((size_t)c * seq + idx[c]) * width + k] += src[i];
  }
}

// dst[(c * seq + idx[c]) * ld + k] = src[c, k]   (the other rows are left alone)
__global__ __launch_bounds__(256) void put_seq_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                           float* __restrict__ dst, size_t ld, int n, int seq, int width) {
  const size_t total = (size_t)n * width;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % width), c = (int)(i / width);
    dst[((size_t)c * seq + idx[c]) * ld + k] = src[i];
  }
}

// the same two for a tensor stored as f16 (fp16 storage mode: the saved pre-GELU activation); the compact side stays fp32
__global__ __launch_bounds__(256) void gather_seq_rows_f16_kernel(const _Float16* __restrict__ src, size_t ld,
                                                                  const int32_t* __restrict__ idx, float* __restrict__ out,
                                                                  int n, int seq, int width) {
  const size_t total = (size_t)n * width;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % width), c = (int)(i / width);
    out[i] = (float)src[((size_t)c * seq + idx[c]) * ld + k];
  }
}
__global__ __launch_bounds__(256) void put_seq_rows_f16_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                               _Float16* __restrict__ dst, size_t ld, int n, int seq, int width) {
  const size_t total = (size_t)n * width;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % width), c = (int)(i / width);
    dst[((size_t)c * seq + idx[c]) * ld + k] = (_Float16)src[i];
  }
}

// idx[c] = argmax_l ids[c, l] (first maximum: the EOT token has the largest id, jclip/model.py:213-214)
__global__ __launch_bounds__(64) void eot_index_kernel(const int64_t* __restrict__ ids, int32_t* __restrict__ idx, int n, int seq) {
  const int c = blockIdx.x, lane = threadIdx.x;
  long best = -1;
  int at = 0;
  for (int l = lane; l < seq; l += 64) {
    const long v = ids[(size_t)c * seq + l];
    if (v > best) {
      best = v;
      at = l;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const long ob = __shfl_xor(best, off, 64);
    const int oa = __shfl_xor(at, off, 64);
    if (ob > best || (ob == best && oa < at)) {
      best = ob;
      at = oa;
    }
  }
  if (lane == 0) idx[c] = at;
}

// ---- per class: normalise templates, mean, normalise (lora_train_vlp.py:978-990) ----  one wave per class
__global__ __launch_bounds__(64) void class_mean_fwd_kernel(const float* __restrict__ emb, float* __restrict__ out,
                                                            int templates, int width) {
  const int c = blockIdx.x, lane = threadIdx.x;
  constexpr int MAXW = 32;  // width <= 2048
  float acc[MAXW];
#pragma unroll
  for (int i = 0; i < MAXW; ++i) acc[i] = 0.f;
  for (int t = 0; t < templates; ++t) {
    const float* e = emb + ((size_t)c * templates + t) * width;
    float v[MAXW];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
      const int k = lane + 64 * i;
      v[i] = k < width ? e[k] : 0.f;
      s = fmaf(v[i], v[i], s);
    }
    const float nrm = sqrtf(wave_sum(s));
#pragma unroll
    for (int i = 0; i < MAXW; ++i) acc[i] += v[i] / nrm;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    acc[i] = acc[i] / (float)templates;
    s = fmaf(acc[i], acc[i], s);
  }
  const float nrm = sqrtf(wave_sum(s));
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int k = lane + 64 * i;
    if (k < width) out[(size_t)c * width + k] = acc[i] / nrm;
  }
}

// backward of class_mean_fwd: demb from dout, recomputing the normalised templates.  one wave per class
__global__ __launch_bounds__(64) void class_mean_bwd_kernel(const float* __restrict__ emb, const float* __restrict__ dout,
                                                            float* __restrict__ demb, int templates, int width) {
  const int c = blockIdx.x, lane = threadIdx.x;
  constexpr int MAXW = 32;
  float acc[MAXW];
#pragma unroll
  for (int i = 0; i < MAXW; ++i) acc[i] = 0.f;
  for (int t = 0; t < templates; ++t) {
    const float* e = emb + ((size_t)c * templates + t) * width;
    float v[MAXW];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
      const int k = lane + 64 * i;
      v[i] = k < width ? e[k] : 0.f;
      s = fmaf(v[i], v[i], s);
    }
    const float nrm = sqrtf(wave_sum(s));
#pragma unroll
    for (int i = 0; i < MAXW; ++i) acc[i] += v[i] / nrm;
  }
  // m = acc / T ; out = m / |m| ; dm = (dout - out <dout, out>) / |m| ; each template gets dm / T
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    acc[i] = acc[i] / (float)templates;
    s = fmaf(acc[i], acc[i], s);
  }
  const float mn = sqrtf(wave_sum(s));
  float g[MAXW];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int k = lane + 64 * i;
    g[i] = k < width ? dout[(size_t)c * width + k] : 0.f;
    dot = fmaf(g[i], acc[i] / mn, dot);
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < MAXW; ++i) g[i] = (g[i] - (acc[i] / mn) * dot) / (mn * (float)templates);
  for (int t = 0; t < templates; ++t) {
    const float* e = emb + ((size_t)c * templates + t) * width;
    float v[MAXW];
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
      const int k = lane + 64 * i;
      v[i] = k < width ? e[k] : 0.f;
      s2 = fmaf(v[i], v[i], s2);
    }
    const float nrm = sqrtf(wave_sum(s2));
    float d2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXW; ++i) d2 = fmaf(g[i], v[i] / nrm, d2);
    d2 = wave_sum(d2);
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
      const int k = lane + 64 * i;
      if (k < width) demb[((size_t)c * templates + t) * width + k] = (g[i] - (v[i] / nrm) * d2) / nrm;
    }
  }
}

// ---- softmax cross entropy, one wave per row (lora_train_vlp.py:997) ----
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits,
                                                            const int64_t* __restrict__ target,
                                                            float* __restrict__ dlogits, float* __restrict__ loss_rows,
                                                            int32_t* __restrict__ correct_rows, int rows, int classes,
                                                            float grad_scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* z = logits + (size_t)row * classes;
  float m = -INFINITY;
  int am = 0;
  for (int k = lane; k < classes; k += 64) {
    const float v = z[k];
    if (v > m) {
      m = v;
      am = k;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(m, off, 64);
    const int oi = __shfl_xor(am, off, 64);
    if (ov > m || (ov == m && oi < am)) {
      m = ov;
      am = oi;
    }
  }
  float s = 0.f;
  for (int k = lane; k < classes; k += 64) s += __expf(z[k] - m);
  s = wave_sum(s);
  const int tgt = (int)target[row];
  const float lse = __logf(s);
  if (lane == 0) {
    loss_rows[row] = lse - (z[tgt] - m);
    if (correct_rows) correct_rows[row] = (am == tgt) ? 1 : 0;
  }
  if (dlogits) {
    const float inv = grad_scale / ((float)rows * s);
    for (int k = lane; k < classes; k += 64) {
      float gk = __expf(z[k] - m) * inv;
      if (k == tgt) gk -= grad_scale / (float)rows;
      dlogits[(size_t)row * classes + k] = gk;
    }
  }
}

// deterministic final reduction of the per-row losses / hit flags (single wave, fixed order)
__global__ __launch_bounds__(64) void ce_finish_kernel(const float* __restrict__ loss_rows,
                                                       const int32_t* __restrict__ correct_rows,
                                                       float* __restrict__ loss_sum, int32_t* __restrict__ correct,
                                                       int rows) {
  const int lane = threadIdx.x;
  float s = 0.f;
  int c = 0;
  for (int r = lane; r < rows; r += 64) {
    s += loss_rows[r];
    if (correct_rows) c += correct_rows[r];
  }
  s = wave_sum(s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if (lane == 0) {
    loss_sum[0] = s;
    if (correct) correct[0] = c;
  }
}

// ---- top-k (k <= 8): repeated arg-max with ties to the smaller index, one wave per row ----
__global__ __launch_bounds__(64) void topk_kernel(const float* __restrict__ logits, int32_t* __restrict__ labels,
                                                  int classes, int k) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const float* z = logits + (size_t)row * classes;
  float prev_v = INFINITY;
  int prev_i = -1;
  for (int t = 0; t < k; ++t) {
    float m = -INFINITY;
    int am = 0x7fffffff;
    for (int c = lane; c < classes; c += 64) {
      const float v = z[c];
      // candidates strictly after the previous pick in (value desc, index asc) order
      const bool after = (v < prev_v) || (v == prev_v && c > prev_i);
      if (after && (v > m || (v == m && c < am))) {
        m = v;
        am = c;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(m, off, 64);
      const int oi = __shfl_xor(am, off, 64);
      if (ov > m || (ov == m && oi < am)) {
        m = ov;
        am = oi;
      }
    }
    if (lane == 0) labels[(size_t)row * k + t] = am;
    prev_v = m;
    prev_i = am;
  }
}

// ---- Channel_LP affine (slow_pace.py:1195-1206) and logit_normalize (:1276-1280) ----
__global__ __launch_bounds__(256) void channel_affine_kernel(const float* __restrict__ x, const float* __restrict__ s1,
                                                             const float* __restrict__ b1, float* __restrict__ y,
                                                             size_t total, int width) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % width);
    y[i] = s1[c] * x[i] + b1[c];
  }
}

// single workgroup: global mean / unbiased std (clamped) over all elements, then per-row centring
__global__ __launch_bounds__(1024) void logit_normalize_kernel(const float* __restrict__ z, float* __restrict__ out,
                                                               float* __restrict__ work, int rows, int classes) {
  __shared__ float red[16];
  __shared__ float stat[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t n = (size_t)rows * classes;
  float s = 0.f;
  for (size_t i = tid; i < n; i += 1024) s += z[i];
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    stat[0] = t / (float)n;
  }
  __syncthreads();
  const float mean = stat[0];
  float q = 0.f;
  for (size_t i = tid; i < n; i += 1024) {
    const float d = z[i] - mean;
    q = fmaf(d, d, q);
  }
  q = wave_sum(q);
  __syncthreads();
  if (lane == 0) red[wave] = q;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    stat[1] = sqrtf(fmaxf(t / (float)(n - 1), 1e-6f));
    if (work) {
      work[0] = mean;
      work[1] = stat[1];
    }
  }
  __syncthreads();
  const float sd = stat[1];
  for (int r = wave; r < rows; r += 16) {
    float rs = 0.f;
    for (int c = lane; c < classes; c += 64) rs += z[(size_t)r * classes + c];
    const float rm = wave_sum(rs) / (float)classes;
    for (int c = lane; c < classes; c += 64) out[(size_t)r * classes + c] = (z[(size_t)r * classes + c] - rm) / sd;
  }
}

// backward of logit_normalize: zn = (z - rowmean(z)) / sigma, sigma = sqrt(max(sum((z - mean_all)^2)/(n-1), 1e-6))
//   dz = (dzn - rowmean(dzn)) / sigma  -  (sum(dzn * zn) / sigma) * (z - mean_all) / ((n - 1) * sigma)   [2nd term 0 if clamped]
__global__ __launch_bounds__(1024) void logit_normalize_bwd_kernel(const float* __restrict__ z,
                                                                   const float* __restrict__ dzn, float* __restrict__ dz,
                                                                   int rows, int classes) {
  __shared__ float red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t n = (size_t)rows * classes;
  auto block_total = [&](float v) -> float {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    return t;
  };
  float s = 0.f;
  for (size_t i = tid; i < n; i += 1024) s += z[i];
  const float mean = block_total(s) / (float)n;
  float q = 0.f;
  for (size_t i = tid; i < n; i += 1024) {
    const float d = z[i] - mean;
    q = fmaf(d, d, q);
  }
  const float var = block_total(q) / (float)(n - 1);
  const bool clamped = var < 1e-6f;
  const float sd = sqrtf(fmaxf(var, 1e-6f));
  // g = sum_ij dzn_ij * (z_ij - rowmean_i) ; needs the row means first: each wave owns rows r = wave, wave+16, ...
  float g = 0.f;
  for (int r = wave; r < rows; r += 16) {
    float rs = 0.f;
    for (int c = lane; c < classes; c += 64) rs += z[(size_t)r * classes + c];
    const float rm = wave_sum(rs) / (float)classes;
    for (int c = lane; c < classes; c += 64) g = fmaf(dzn[(size_t)r * classes + c], z[(size_t)r * classes + c] - rm, g);
  }
  g = block_total(g);  // = sigma * sum(dzn * zn)
  const float coef = clamped ? 0.f : g / (sd * sd * sd * (float)(n - 1));
  for (int r = wave; r < rows; r += 16) {
    float ds = 0.f;
    for (int c = lane; c < classes; c += 64) ds += dzn[(size_t)r * classes + c];
    const float dm = wave_sum(ds) / (float)classes;
    for (int c = lane; c < classes; c += 64) {
      const size_t i = (size_t)r * classes + c;
      dz[i] = (dzn[i] - dm) / sd - coef * (z[i] - mean);
    }
  }
}

// out[c] (+)= sum_r x[r,c] * (y ? y[r,c] : 1)     one thread per column, fixed row order
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                     float* __restrict__ out, int rows, int cols) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float acc = 0.f;
  for (int r = 0; r < rows; ++r) {
    const float v = x[(size_t)r * cols + c];
    acc = y ? fmaf(v, y[(size_t)r * cols + c], acc) : acc + v;
  }
  out[c] = acc;
}

// ---- AdamW over the flat trainable buffer (lora_train_vlp.py:946,1002) ----
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                                    float b1, float b2, float eps, float wd, float inv_sqrt_bc2,
                                                    float step_size, float grad_scale) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * grad_scale;
  float pi = p[i] * (1.f - lr * wd);
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
  pi -= step_size * mi / denom;
  p[i] = pi;
  m[i] = mi;
  v[i] = vi;
}

// C[m,n] = alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]   (tiny products whose reduction dim is
// not 16-byte friendly, e.g. over the 403 classes in the logits backward)
__global__ __launch_bounds__(256) void matmul_small_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                           float* __restrict__ C, int M, int N, int K, long sam, long sak,
                                                           long sbk, long sbn, float alpha) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(A[m * sam + k * sak], B[k * sbk + n * sbn], acc);
  C[i] = alpha * acc;
}

static inline unsigned grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}


// ---- stage-2 self-consistency losses (slow_pace.py:1653-1658, kl_div :1170-1177) ----------------------------------
// L1: loss = mean |a - b| ; da = sign(a - b) * grad_scale / n.  One 1024-thread block: fixed thread -> element
// assignment and a fixed LDS tree, so the sum is bitwise reproducible (n is a few 1e5 on the path).
__global__ __launch_bounds__(1024) void l1_loss_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                       float* __restrict__ loss, float* __restrict__ da, float gscale) {
  __shared__ float red[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float g = gscale / (float)n;
  float acc = 0.f;
  for (size_t i = threadIdx.x; i < n; i += 1024) {
    const float d = a[i] - b[i];
    acc += fabsf(d);
    if (da) da[i] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    loss[0] = t / (float)n;
  }
}

// KL(softmax(t) || softmax(x)) per row:  loss_rows[r] = sum_j q_j (log q_j - log p_j),  dx = (p - q) * grad_scale.
// One wave per row (403 classes); both log-softmaxes with max subtraction.
__global__ __launch_bounds__(256) void kl_logits_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                        float* __restrict__ loss_rows, float* __restrict__ dx, int rows,
                                                        int cols, float gscale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * cols;
  const float* tr = t + (size_t)row * cols;
  float mx = -INFINITY, mt = -INFINITY;
  for (int j = lane; j < cols; j += 64) {
    mx = fmaxf(mx, xr[j]);
    mt = fmaxf(mt, tr[j]);
  }
  mx = wave_max(mx);
  mt = wave_max(mt);
  float sx = 0.f, st = 0.f;
  for (int j = lane; j < cols; j += 64) {
    sx += __expf(xr[j] - mx);
    st += __expf(tr[j] - mt);
  }
  sx = wave_sum(sx);
  st = wave_sum(st);
  const float lx = mx + __logf(sx), lt = mt + __logf(st);
  float acc = 0.f;
  for (int j = lane; j < cols; j += 64) {
    const float a = xr[j] - lx, b = tr[j] - lt;  // log p, log q
    const float q = __expf(b);
    acc += q * (b - a);
    if (dx) dx[(size_t)row * cols + j] = (__expf(a) - q) * gscale;
  }
  acc = wave_sum(acc);
  if (lane == 0) loss_rows[row] = acc;
}

}  // namespace clipfs

using namespace clipfs;

extern "C" int clipfs_vit_fill_special(float* x, const float* class_emb, const float* pos, const float* vpt, int batch,
                                       int tokens, int n_patch, int n_vpt, int width, void* stream) {
  CLIPFS_REQUIRE(x && class_emb && pos && (n_vpt == 0 || vpt), "vit_fill_special: null pointer");
  CLIPFS_REQUIRE(batch > 0 && width > 0 && n_vpt >= 0 && tokens == 1 + n_patch + n_vpt, "vit_fill_special: tokens %d != 1 + %d + %d", tokens, n_patch, n_vpt);
  const size_t total = (size_t)batch * (1 + n_vpt) * width;
  hipLaunchKernelGGL(vit_fill_special_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, class_emb, pos,
                     vpt, batch, tokens, n_patch, n_vpt, width);
  return launch_status();
}

extern "C" int clipfs_text_embed(const int64_t* ids, const float* table, const float* pos, const float* ctx, int n_ctx,
                                 float* x, int n, int seq, int width, void* stream) {
  CLIPFS_REQUIRE(ids && table && pos && x, "text_embed: null pointer");
  CLIPFS_REQUIRE(n > 0 && seq > 0 && width > 0 && (width & 3) == 0 && (!ctx || (n_ctx > 0 && n_ctx < seq)), "text_embed: bad dims");
  CLIPFS_REQUIRE(aligned16(table) && aligned16(pos) && aligned16(x) && (!ctx || aligned16(ctx)), "text_embed: misaligned pointer");
  const size_t total = (size_t)n * seq * (width / 4);
  hipLaunchKernelGGL(text_embed_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, ids, table, pos, ctx,
                     n_ctx, x, n, seq, width);
  return launch_status();
}

extern "C" int clipfs_token_rows_grad(const float* dx, float* dctx, int n, int seq, int width, int n_ctx, int first,
                                      void* stream) {
  CLIPFS_REQUIRE(dx && dctx && n > 0 && n_ctx > 0 && first >= 0 && first + n_ctx <= seq && width > 0, "token_rows_grad: bad args");
  hipLaunchKernelGGL(text_ctx_grad_kernel, dim3((n_ctx * width + 255) / 256), dim3(256), 0, (hipStream_t)stream, dx, dctx,
                     n, seq, width, n_ctx, first);
  return launch_status();
}

extern "C" int clipfs_matmul_small(const float* A, const float* B, float* C, int M, int N, int K, long sam, long sak,
                                   long sbk, long sbn, float alpha, void* stream) {
  CLIPFS_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "matmul_small: bad args");
  const size_t total = (size_t)M * N;
  hipLaunchKernelGGL(matmul_small_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A, B, C,
                     M, N, K, sam, sak, sbk, sbn, alpha);
  return launch_status();
}

extern "C" int clipfs_gather_eot(const float* x, const int64_t* ids, float* out, int32_t* idx_out, int n, int seq,
                                 int width, void* stream) {
  CLIPFS_REQUIRE(x && ids && out && n > 0 && seq > 0 && width > 0, "gather_eot: bad args");
  hipLaunchKernelGGL(gather_eot_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, x, ids, out, idx_out, seq, width);
  return launch_status();
}

extern "C" int clipfs_scatter_rows(const float* dy, const int32_t* idx, float* dx, int n, int seq, int width,
                                   void* stream) {
  CLIPFS_REQUIRE(dy && idx && dx && n > 0 && seq > 0 && width > 0, "scatter_rows: bad args");
  const size_t total = (size_t)n * seq * width;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, n, seq,
                     width);
  return launch_status();
}

extern "C" int clipfs_gather_seq_rows(const float* src, size_t ld, const int32_t* idx, float* out, int n, int seq, int width,
                                      void* stream) {
  CLIPFS_REQUIRE(src && idx && out && n > 0 && seq > 0 && width > 0 && ld >= (size_t)width, "gather_seq_rows: bad args");
  hipLaunchKernelGGL(gather_seq_rows_kernel, dim3(grid_for((size_t)n * width)), dim3(256), 0, (hipStream_t)stream, src, ld,
                     idx, out, n, seq, width);
  return launch_status();
}

extern "C" int clipfs_add_seq_rows(const float* src, const int32_t* idx, float* dx, int n, int seq, int width, void* stream) {
  CLIPFS_REQUIRE(src && idx && dx && n > 0 && seq > 0 && width > 0, "add_seq_rows: bad args");
  hipLaunchKernelGGL(add_seq_rows_kernel, dim3(grid_for((size_t)n * width)), dim3(256), 0, (hipStream_t)stream, src, idx, dx,
                     n, seq, width);
  return launch_status();
}

extern "C" int clipfs_put_seq_rows(const float* src, const int32_t* idx, float* dst, size_t ld, int n, int seq, int width,
                                   void* stream) {
  CLIPFS_REQUIRE(src && idx && dst && n > 0 && seq > 0 && width > 0 && ld >= (size_t)width, "put_seq_rows: bad args");
  hipLaunchKernelGGL(put_seq_rows_kernel, dim3(grid_for((size_t)n * width)), dim3(256), 0, (hipStream_t)stream, src, idx, dst,
                     ld, n, seq, width);
  return launch_status();
}

extern "C" int clipfs_gather_seq_rows_f16(const void* src_f16, size_t ld, const int32_t* idx, float* out, int n, int seq,
                                          int width, void* stream) {
  CLIPFS_REQUIRE(src_f16 && idx && out && n > 0 && seq > 0 && width > 0 && ld >= (size_t)width, "gather_seq_rows_f16: bad args");
  hipLaunchKernelGGL(gather_seq_rows_f16_kernel, dim3(grid_for((size_t)n * width)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const _Float16*>(src_f16), ld, idx, out, n, seq, width);
  return launch_status();
}

extern "C" int clipfs_put_seq_rows_f16(const float* src, const int32_t* idx, void* dst_f16, size_t ld, int n, int seq,
                                       int width, void* stream) {
  CLIPFS_REQUIRE(src && idx && dst_f16 && n > 0 && seq > 0 && width > 0 && ld >= (size_t)width, "put_seq_rows_f16: bad args");
  hipLaunchKernelGGL(put_seq_rows_f16_kernel, dim3(grid_for((size_t)n * width)), dim3(256), 0, (hipStream_t)stream, src, idx,
                     reinterpret_cast<_Float16*>(dst_f16), ld, n, seq, width);
  return launch_status();
}

extern "C" int clipfs_eot_index(const int64_t* ids, int32_t* idx, int n, int seq, void* stream) {
  CLIPFS_REQUIRE(ids && idx && n > 0 && seq > 0, "eot_index: bad args");
  hipLaunchKernelGGL(eot_index_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, ids, idx, n, seq);
  return launch_status();
}

extern "C" int clipfs_class_mean_fwd(const float* emb, float* out, int classes, int templates, int width, void* stream) {
  CLIPFS_REQUIRE(emb && out && classes > 0 && templates > 0 && width > 0 && width <= 2048, "class_mean_fwd: bad args");
  hipLaunchKernelGGL(class_mean_fwd_kernel, dim3(classes), dim3(64), 0, (hipStream_t)stream, emb, out, templates, width);
  return launch_status();
}

extern "C" int clipfs_class_mean_bwd(const float* emb, const float* dout, float* demb, int classes, int templates,
                                     int width, void* stream) {
  CLIPFS_REQUIRE(emb && dout && demb && classes > 0 && templates > 0 && width > 0 && width <= 2048, "class_mean_bwd: bad args");
  hipLaunchKernelGGL(class_mean_bwd_kernel, dim3(classes), dim3(64), 0, (hipStream_t)stream, emb, dout, demb, templates,
                     width);
  return launch_status();
}

extern "C" int clipfs_cross_entropy(const float* logits, const int64_t* target, float* dlogits, float* loss_rows,
                                    float* loss_sum, int32_t* correct, int rows, int classes, float grad_scale,
                                    void* stream) {
  CLIPFS_REQUIRE(logits && target && loss_rows && loss_sum && rows > 0 && classes > 0, "cross_entropy: bad args");
  // loss_rows is [2 * rows]: per-row losses, then per-row hit flags (int32) when correct != NULL
  int32_t* flags = correct ? reinterpret_cast<int32_t*>(loss_rows + rows) : nullptr;
  hipLaunchKernelGGL(cross_entropy_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, target,
                     dlogits, loss_rows, flags, rows, classes, grad_scale);
  CLIPFS_CHECK(launch_status());
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_rows, flags, loss_sum, correct,
                     rows);
  return launch_status();
}

extern "C" int clipfs_topk(const float* logits, int32_t* labels, int rows, int classes, int k, void* stream) {
  CLIPFS_REQUIRE(logits && labels && rows > 0 && classes > 0 && k > 0 && k <= classes, "topk: bad args");
  hipLaunchKernelGGL(topk_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, logits, labels, classes, k);
  return launch_status();
}

extern "C" int clipfs_channel_affine(const float* x, const float* scale1, const float* bias1, float* y, int rows,
                                     int width, void* stream) {
  CLIPFS_REQUIRE(x && scale1 && bias1 && y && rows > 0 && width > 0, "channel_affine: bad args");
  const size_t total = (size_t)rows * width;
  hipLaunchKernelGGL(channel_affine_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, scale1, bias1, y,
                     total, width);
  return launch_status();
}

extern "C" int clipfs_logit_normalize(const float* z, float* out, float* work, int rows, int classes, void* stream) {
  CLIPFS_REQUIRE(z && out && rows > 0 && classes > 0 && (size_t)rows * classes > 1, "logit_normalize: bad args");
  hipLaunchKernelGGL(logit_normalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, z, out, work, rows, classes);
  return launch_status();
}

extern "C" int clipfs_logit_normalize_bwd(const float* z, const float* dzn, float* dz, int rows, int classes,
                                          void* stream) {
  CLIPFS_REQUIRE(z && dzn && dz && rows > 0 && classes > 0 && (size_t)rows * classes > 1, "logit_normalize_bwd: bad args");
  hipLaunchKernelGGL(logit_normalize_bwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, z, dzn, dz, rows, classes);
  return launch_status();
}

extern "C" int clipfs_colsum(const float* x, const float* y, float* out, int rows, int cols, void* stream) {
  CLIPFS_REQUIRE(x && out && rows > 0 && cols > 0, "colsum: bad args");
  hipLaunchKernelGGL(colsum_kernel, dim3((cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, out, rows, cols);
  return launch_status();
}

extern "C" int clipfs_adamw(float* p, const float* g, float* m, float* v, size_t n, int step, float lr, float beta1,
                            float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
  CLIPFS_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw: bad args");
  const double bc1 = 1.0 - pow((double)beta1, step);
  const double bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr,
                     beta1, beta2, eps, weight_decay, (float)(1.0 / sqrt(bc2)), (float)((double)lr / bc1), grad_scale);
  return launch_status();
}

extern "C" int clipfs_l1_loss(const float* a, const float* b, size_t n, float* loss, float* da, float grad_scale,
                              void* stream) {
  CLIPFS_REQUIRE(a && b && loss && n > 0, "l1_loss: bad args");
  hipLaunchKernelGGL(l1_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, b, n, loss, da, grad_scale);
  return launch_status();
}

extern "C" int clipfs_kl_logits(const float* logits, const float* target_logits, float* loss_rows, float* dlogits,
                                int rows, int classes, float grad_scale, void* stream) {
  CLIPFS_REQUIRE(logits && target_logits && loss_rows && rows > 0 && classes > 0, "kl_logits: bad args");
  hipLaunchKernelGGL(kl_logits_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, target_logits,
                     loss_rows, dlogits, rows, classes, grad_scale);
  return launch_status();
}
