// MTA (robust mean-shift test-time augmentation): solve_mta of lora_train_vlp.py:742-811 /
// slow_pace.py:1363-1433 as ONE kernel, one 1024-thread workgroup per source image.
// The reference issues >= 55 tiny launches and a device->host sync per `while` test; here the
// data-dependent loop exits are workgroup-uniform branches on an LDS flag and nothing leaves the GPU.
// Latency-bound (~35 MFLOP per image at V = 65): images are independent, the grid is the batch.
#include "common.h"

namespace clipfs {

constexpr int MTA_THREADS = 1024;
constexpr int MTA_MAX_V = 1024;
constexpr int MTA_MAX_D = 1024;

__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < MTA_THREADS / 64; ++w) t += red[w];
  return t;
}

__device__ __forceinline__ float block_max(float v, float* red, int tid) {
  v = wave_max(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int w = 1; w < MTA_THREADS / 64; ++w) t = fmaxf(t, red[w]);
  return t;
}

__global__ __launch_bounds__(MTA_THREADS) void mta_kernel(const float* __restrict__ feats,
                                                          const float* __restrict__ text, float* __restrict__ mode_out,
                                                          float* __restrict__ logits_out, float* __restrict__ work, int V,
                                                          int d, int C) {
  __shared__ float s_y[MTA_MAX_V], s_den[MTA_MAX_V], s_bw[MTA_MAX_V], s_tmp[MTA_MAX_V];
  __shared__ float s_mode[MTA_MAX_D], s_old[MTA_MAX_D];
  __shared__ float red[MTA_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = MTA_THREADS / 64;
  const int img = blockIdx.x;
  const float* F = feats + (size_t)img * V * d;
  float* sm = work + (size_t)img * ((size_t)V * C + 2 * (size_t)V * V);  // [V][C] softmax(logits)
  float* aff = sm + (size_t)V * C;                                        // [V][V]
  float* dist = aff + (size_t)V * V;                                      // [V][V]

  // 1. logits = F T^T * 100, row softmax  (wave per (v, c) dot product over d)
  for (int v = wave; v < V; v += NW) {
    for (int c = 0; c < C; ++c) {
      float a = 0.f;
      for (int k = lane; k < d; k += 64) a = fmaf(F[(size_t)v * d + k], text[(size_t)c * d + k], a);
      a = wave_sum(a);
      if (lane == 0) sm[(size_t)v * C + c] = a * 100.f;
    }
  }
  __syncthreads();
  for (int v = wave; v < V; v += NW) {
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, sm[(size_t)v * C + c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += __expf(sm[(size_t)v * C + c] - m);
    s = wave_sum(s);
    for (int c = lane; c < C; c += 64) sm[(size_t)v * C + c] = __expf(sm[(size_t)v * C + c] - m) / s;
  }
  __syncthreads();
  // 2. affinity = sm sm^T ; 3. dist = sqrt(max(|fi|^2 - 2 fi.fj + |fj|^2, 0))
  for (int v = wave; v < V; v += NW) {  // squared norms -> s_tmp
    float a = 0.f;
    for (int k = lane; k < d; k += 64) a = fmaf(F[(size_t)v * d + k], F[(size_t)v * d + k], a);
    a = wave_sum(a);
    if (lane == 0) s_tmp[v] = a;
  }
  __syncthreads();
  for (int p = wave; p < V * V; p += NW) {
    const int i = p / V, j = p % V;
    float a = 0.f, g = 0.f;
    for (int c = lane; c < C; c += 64) a = fmaf(sm[(size_t)i * C + c], sm[(size_t)j * C + c], a);
    for (int k = lane; k < d; k += 64) g = fmaf(F[(size_t)i * d + k], F[(size_t)j * d + k], g);
    a = wave_sum(a);
    g = wave_sum(g);
    if (lane == 0) {
      aff[p] = a;
      dist[p] = sqrtf(fmaxf(s_tmp[i] - 2.f * g + s_tmp[j], 0.f));
    }
  }
  __syncthreads();
  // 4. bandwidth_i = sqrt(0.5 * mean of the k smallest squared distances, excluding rank 0 (itself))
  const int kn = (int)(0.3 * (double)(V - 1));
  for (int i = wave; i < V; i += NW) {
    const float* dr = dist + (size_t)i * V;
    float acc = 0.f;
    for (int j = lane; j < V; j += 64) {
      const float dj = dr[j];
      int rank = 0;
      for (int l = 0; l < V; ++l) {
        const float dl = dr[l];
        rank += (dl < dj || (dl == dj && l < j)) ? 1 : 0;
      }
      if (rank >= 1 && rank <= kn) acc = fmaf(dj, dj, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) s_bw[i] = sqrtf(0.5f * (acc / (float)kn));
  }
  for (int v = tid; v < V; v += MTA_THREADS) s_y[v] = 1.f / (float)V;
  for (int k = tid; k < d; k += MTA_THREADS) s_mode[k] = F[k];
  __syncthreads();

  auto compute_density = [&]() {  // s_den[v] = exp(-|F_v - mode|^2 / (2 bw_v^2))
    for (int v = wave; v < V; v += NW) {
      float a = 0.f;
      for (int k = lane; k < d; k += 64) {
        const float df = F[(size_t)v * d + k] - s_mode[k];
        a = fmaf(df, df, a);
      }
      a = wave_sum(a);
      if (lane == 0) {
        const float dn = sqrtf(a);  // the reference takes the norm, then squares it again (:734-735)
        s_den[v] = __expf(-(dn * dn) / (2.f * s_bw[v] * s_bw[v]));
      }
    }
    __syncthreads();
  };

  const int max_iter = 5;
  const float th = 1e-6f;
  for (int it = 0; it < max_iter; ++it) {
    compute_density();
    // inlierness step
    for (int i = 1; i <= max_iter; ++i) {
      for (int v = wave; v < V; v += NW) {
        float a = 0.f;
        for (int j = lane; j < V; j += 64) a = fmaf(aff[(size_t)v * V + j], s_y[j], a);
        a = wave_sum(a);
        if (lane == 0) s_tmp[v] = 5.f * (s_den[v] + 4.f * a);  // 1/lambda_y * (density + lambda_q * sum)
      }
      __syncthreads();
      float m = -INFINITY;
      for (int v = tid; v < V; v += MTA_THREADS) m = fmaxf(m, s_tmp[v]);
      m = block_max(m, red, tid);
      float s = 0.f;
      for (int v = tid; v < V; v += MTA_THREADS) s += __expf(s_tmp[v] - m);
      s = block_sum(s, red, tid);
      float dn = 0.f;
      for (int v = tid; v < V; v += MTA_THREADS) {
        const float ny = __expf(s_tmp[v] - m) / s;
        const float df = s_y[v] - ny;
        dn = fmaf(df, df, dn);
        s_tmp[v] = ny;
      }
      dn = block_sum(dn, red, tid);
      for (int v = tid; v < V; v += MTA_THREADS) s_y[v] = s_tmp[v];
      __syncthreads();
      if (sqrtf(dn) < th) break;  // uniform: dn is identical in every thread
    }
    // mode step
    for (int i = 1; i <= max_iter; ++i) {
      compute_density();
      float ws = 0.f;
      for (int v = tid; v < V; v += MTA_THREADS) ws += s_den[v] * s_y[v];
      ws = block_sum(ws, red, tid);
      float nm = 0.f;
      float mk[MTA_MAX_D / MTA_THREADS];
#pragma unroll
      for (int q = 0; q < MTA_MAX_D / MTA_THREADS; ++q) {
        const int k = tid + q * MTA_THREADS;
        mk[q] = 0.f;
        if (k < d) {
          float a = 0.f;
          for (int v = 0; v < V; ++v) a = fmaf(s_den[v] * s_y[v], F[(size_t)v * d + k], a);
          mk[q] = a / ws;
          nm = fmaf(mk[q], mk[q], nm);
        }
      }
      nm = sqrtf(block_sum(nm, red, tid));
      float dn = 0.f;
#pragma unroll
      for (int q = 0; q < MTA_MAX_D / MTA_THREADS; ++q) {
        const int k = tid + q * MTA_THREADS;
        if (k < d) {
          const float nv = mk[q] / nm;
          const float df = s_mode[k] - nv;
          dn = fmaf(df, df, dn);
          s_old[k] = nv;
        }
      }
      dn = block_sum(dn, red, tid);
      for (int k = tid; k < d; k += MTA_THREADS) s_mode[k] = s_old[k];
      __syncthreads();
      if (sqrtf(dn) < th) break;
    }
  }
  if (mode_out)
    for (int k = tid; k < d; k += MTA_THREADS) mode_out[(size_t)img * d + k] = s_mode[k];
  if (logits_out) {
    for (int c = wave; c < C; c += NW) {
      float a = 0.f;
      for (int k = lane; k < d; k += 64) a = fmaf(s_mode[k], text[(size_t)c * d + k], a);
      a = wave_sum(a);
      if (lane == 0) logits_out[(size_t)img * C + c] = a * 100.f;
    }
  }
}

}  // namespace clipfs

using namespace clipfs;

extern "C" size_t clipfs_mta_work_floats(int n_img, int views, int width, int classes) {
  (void)width;
  return (size_t)n_img * ((size_t)views * classes + 2 * (size_t)views * views);
}

extern "C" int clipfs_mta(const float* feats, const float* text, float* mode_out, float* logits_out, float* work,
                          int n_img, int views, int width, int classes, void* stream) {
  CLIPFS_REQUIRE(feats && text && work && (mode_out || logits_out), "mta: null pointer");
  CLIPFS_REQUIRE(n_img > 0 && views >= 5 && views <= MTA_MAX_V && width > 0 && width <= MTA_MAX_D && classes > 0,
                 "mta: n_img %d views %d width %d classes %d unsupported (views <= %d, width <= %d)", n_img, views, width,
                 classes, MTA_MAX_V, MTA_MAX_D);
  hipLaunchKernelGGL(mta_kernel, dim3(n_img), dim3(MTA_THREADS), 0, (hipStream_t)stream, feats, text, mode_out, logits_out,
                     work, views, width, classes);
  return launch_status();
}
