// Which fp32 MFMA shape does the chip hold a higher clock on?  (cdna_hip_programming.md 5.4 rule 28, MI355X_MICROARCH.md
// "DVFS give-back" item 7: measured there for bf16 only.)  Two loops with the SAME output tile per wave (32 x 64), the
// same LDS fragment reads (12 ds_read_b128 per 32-deep K-step, conflict-free images of random data) and the same FLOPs:
//   shape 0: v_mfma_f32_32x32x2_f32, 1 x 2 tiles, 32 MFMAs per K-step
//   shape 1: v_mfma_f32_16x16x4_f32, 2 x 4 tiles, 64 MFMAs per K-step
// 2 workgroups of 4 waves per CU (the GEMM's occupancy), no global loads, no barriers in the loop.  Reports TFLOP/s by
// wall time and the in-kernel clock (delta s_memtime / delta s_memrealtime * 100 MHz, median over workgroups).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probe/mfma_shape_probe scripts/probe/mfma_shape_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool READS>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, float* __restrict__ out, unsigned long long* stamps,
                                            int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (64 + 128) * 32];  // two stages of the 64 x 128 x 32 GEMM tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * (64 + 128) * 32; i += 256) lds[i] = src[(blockIdx.x * 97 + i) & 0xFFFFF];
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  unsigned long long t0 = 0, r0 = 0;
  if (tid == 0) {
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  float sink = 0.f;
  if (SHAPE == 0) {
    const int fr = lane & 31, fh = lane >> 5, swz = (fr >> 1) & 7;
    const int a_off = (wm * 32 + fr) * 32, b_off0 = 64 * 32 + (wn * 64 + fr) * 32, b_off1 = b_off0 + 32 * 32;
    f32x16 acc0 = {}, acc1 = {};
    f32x4 av = {1.f, 2.f, 3.f, 4.f}, bv0 = av, bv1 = av;
    for (int it = 0; it < iters; ++it) {
      const float* s = lds + (it & 1) * (64 + 128) * 32;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = ((2 * q + fh) ^ swz) << 2;
        if (READS) {
          av = *reinterpret_cast<const f32x4*>(s + a_off + ch);
          bv0 = *reinterpret_cast<const f32x4*>(s + b_off0 + ch);
          bv1 = *reinterpret_cast<const f32x4*>(s + b_off1 + ch);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv0[e], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv1[e], acc1, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sink += acc0[r] + acc1[r];
  } else {
    const int fr = lane & 15, g = lane >> 4;
    f32x4 acc[2][4] = {};
    f32x4 av[2] = {{1.f, 2.f, 3.f, 4.f}, {1.f, 2.f, 3.f, 4.f}}, bv[4] = {av[0], av[0], av[0], av[0]};
    int a_off[2], b_off[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) a_off[i] = (wm * 32 + i * 16 + fr) * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) b_off[j] = 64 * 32 + (wn * 64 + j * 16 + fr) * 32;
    int swz_a[2], swz_b[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) swz_a[i] = ((i * 16 + fr) >> 1) & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) swz_b[j] = ((j * 16 + fr) >> 1) & 7;
    for (int it = 0; it < iters; ++it) {
      const float* s = lds + (it & 1) * (64 + 128) * 32;
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {  // lane group g reads chunks g and g + 4 of the 8 in a 32-deep row
        if (READS) {
#pragma unroll
          for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const f32x4*>(s + a_off[i] + (((g + 4 * jh) ^ swz_a[i]) << 2));
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(s + b_off[j] + (((g + 4 * jh) ^ swz_b[j]) << 2));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sink += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  }
  if (tid == 0) {
    stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
    stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
  out[blockIdx.x * 256 + tid] = sink;
}

template <int SHAPE, bool READS>
static void run(const char* name, const float* src, float* out, unsigned long long* stamps, int wgs, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<SHAPE, READS>), dim3(wgs), dim3(256), 0, 0, src, out, stamps, iters);
  hipDeviceSynchronize();
  std::vector<double> tf, clk;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<SHAPE, READS>), dim3(wgs), dim3(256), 0, 0, src, out, stamps, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(wgs * 2);
    hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * wgs * 2, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int i = 0; i < wgs; ++i)
      if (h[2 * i + 1]) c.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
    std::sort(c.begin(), c.end());
    const double flops = (double)wgs * 4 /*waves*/ * iters * 32.0 * (2.0 * 32 * 32 * 2);  // per K-step: 32 x 64 x 32 MACs per wave
    tf.push_back(flops / (ms * 1e-3) / 1e12);
    clk.push_back(c[c.size() / 2]);
  }
  std::sort(tf.begin(), tf.end());
  std::sort(clk.begin(), clk.end());
  printf("%-34s %7.1f TFLOP/s (median of 5; min %6.1f max %6.1f)   in-kernel clock %5.3f GHz   -> %5.1f FLOP/clk/CU\n", name, tf[2],
         tf[0], tf[4], clk[2], tf[2] * 1e12 / (clk[2] * 1e9) / 256.0);
}


// ---- the same 32x32x2 loop with the GEMM's global -> LDS stream beside it: per K-step every wave issues 6
// global_load_lds_dwordx4 (24 KiB per workgroup, the 64 x 128 x 32 tile's operands), waits for them at the end of the
// K-step and passes a barrier.  The source window decides where the bytes come from: `span` bytes per XCD-group of
// workgroups, walked cyclically (small: every line an L2 hit; larger than 4 MiB per XCD: Infinity Cache; > 256 MiB: HBM).
template <int NDMA>
__global__ __launch_bounds__(256) void probe_dma(const float* __restrict__ src, float* __restrict__ out, unsigned long long* stamps,
                                                 int iters, size_t span_floats, int share) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (64 + 128) * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * (64 + 128) * 32; i += 256) lds[i] = src[(blockIdx.x * 97 + i) & 0xFFFFF];
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fh = lane >> 5, swz = (fr >> 1) & 7;
  const int a_off = (wm * 32 + fr) * 32, b_off0 = 64 * 32 + (wn * 64 + fr) * 32, b_off1 = b_off0 + 32 * 32;
  f32x16 acc0 = {}, acc1 = {};
  // share = 1: every workgroup of the launch walks the SAME window (operands shared chip-wide, like weights);
  // share = 0: each workgroup its own slice of the window
  const size_t wg_off = share ? 0 : ((size_t)blockIdx.x * 6151 * 256) & (span_floats - 1);
  unsigned long long t0 = 0, r0 = 0;
  if (tid == 0) {
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  // span_floats is a power of two: the walk wraps with a mask (no 64-bit division in the loop)
  const size_t mask = span_floats - 1;
  size_t pos = wg_off + (size_t)uw * NDMA * 256 + lane * 4;
  for (int it = 0; it < iters; ++it) {
    float* st = lds + ((it + 1) & 1) * (64 + 128) * 32;
#pragma unroll
    for (int i = 0; i < NDMA; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((pos + (size_t)i * 256) & mask)),
                                       (__attribute__((address_space(3))) void*)(st + (uw * 6 + i) * 256), 16, 0, 0);
    pos += 4 * NDMA * 256;
    const float* s = lds + (it & 1) * (64 + 128) * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = ((2 * q + fh) ^ swz) << 2;
      const f32x4 av = *reinterpret_cast<const f32x4*>(s + a_off + ch);
      const f32x4 bv0 = *reinterpret_cast<const f32x4*>(s + b_off0 + ch);
      const f32x4 bv1 = *reinterpret_cast<const f32x4*>(s + b_off1 + ch);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv0[e], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv1[e], acc1, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float sink = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) sink += acc0[r] + acc1[r];
  if (tid == 0) {
    stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
    stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
  out[blockIdx.x * 256 + tid] = sink;
}

template <int NDMA>
static void run_dma(const char* name, const float* src, float* out, unsigned long long* stamps, int wgs, int iters, size_t span_bytes,
                    int share) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w)
    hipLaunchKernelGGL(probe_dma<NDMA>, dim3(wgs), dim3(256), 0, 0, src, out, stamps, iters, span_bytes / 4, share);
  hipDeviceSynchronize();
  std::vector<double> tf, clk;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe_dma<NDMA>, dim3(wgs), dim3(256), 0, 0, src, out, stamps, iters, span_bytes / 4, share);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(wgs * 2);
    hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * wgs * 2, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int i = 0; i < wgs; ++i)
      if (h[2 * i + 1]) c.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
    std::sort(c.begin(), c.end());
    const double flops = (double)wgs * 4 * iters * 32.0 * (2.0 * 32 * 32 * 2);
    tf.push_back(flops / (ms * 1e-3) / 1e12);
    clk.push_back(c[c.size() / 2]);
  }
  std::sort(tf.begin(), tf.end());
  std::sort(clk.begin(), clk.end());
  const double tbs = (double)wgs * iters * (4096.0 * NDMA) / (((double)wgs * 4 * iters * 32.0 * 4096.0) / (tf[2] * 1e12)) / 1e12;
  printf("%-58s %7.1f TFLOP/s   clock %5.3f GHz   %5.1f FLOP/clk/CU   stream %5.2f TB/s\n", name, tf[2], clk[2],
         tf[2] * 1e12 / (clk[2] * 1e9) / 256.0, tbs);
}

int main() {
  const int wgs = 512, iters = 20000;
  float *src, *out;
  unsigned long long* stamps;
  const size_t src_floats = (size_t)1 << 29;  // 2 GiB: beyond the 256 MiB Infinity Cache
  hipMalloc(&src, sizeof(float) * src_floats);
  hipMalloc(&out, sizeof(float) * wgs * 256);
  hipMalloc(&stamps, sizeof(unsigned long long) * wgs * 2);
  std::vector<float> h(1 << 20);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (size_t o = 0; o < src_floats; o += h.size()) hipMemcpy(src + o, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice);
  // warm the chip for ~2 s so that the clock has settled under load
  for (int i = 0; i < 12; ++i) hipLaunchKernelGGL((probe<0, true>), dim3(wgs), dim3(256), 0, 0, src, out, stamps, iters);
  hipDeviceSynchronize();
  if (getenv("PROBE_DMA")) {
    const int it2 = 4000;
    run<0, true>("32x32x2 + LDS reads, no global stream", src, out, stamps, wgs, it2);
    run_dma<6>("+ 6 LDS-DMA / wave / K-step (64x128 tile), 1 MiB shared window (L2)", src, out, stamps, wgs, it2, (size_t)1 << 20, 1);
    run_dma<4>("+ 4 LDS-DMA (128x128-tile bytes per FLOP), 1 MiB shared (L2)", src, out, stamps, wgs, it2, (size_t)1 << 20, 1);
    run_dma<3>("+ 3 LDS-DMA (128x256-tile bytes per FLOP), 1 MiB shared (L2)", src, out, stamps, wgs, it2, (size_t)1 << 20, 1);
    run_dma<2>("+ 2 LDS-DMA (256x256-tile bytes per FLOP), 1 MiB shared (L2)", src, out, stamps, wgs, it2, (size_t)1 << 20, 1);
    run_dma<6>("+ 6 LDS-DMA, own slices of a 16 MiB window (L2 capacity)", src, out, stamps, wgs, it2, (size_t)16 << 20, 0);
    run_dma<6>("+ 6 LDS-DMA, own slices of a 128 MiB window (Infinity Cache)", src, out, stamps, wgs, it2, (size_t)128 << 20, 0);
    run_dma<3>("+ 3 LDS-DMA, own slices of a 128 MiB window (Infinity Cache)", src, out, stamps, wgs, it2, (size_t)128 << 20, 0);
    run_dma<6>("+ 6 LDS-DMA, own slices of a 2 GiB window (HBM)", src, out, stamps, wgs, it2, (size_t)2048 << 20, 0);
    run_dma<3>("+ 3 LDS-DMA, own slices of a 2 GiB window (HBM)", src, out, stamps, wgs, it2, (size_t)2048 << 20, 0);
    run<0, true>("32x32x2 + LDS reads, no global stream (again)", src, out, stamps, wgs, it2);
    return 0;
  }
  for (int round = 0; round < 2; ++round) {
    run<0, true>("32x32x2  + LDS fragment reads", src, out, stamps, wgs, iters);
    run<1, true>("16x16x4  + LDS fragment reads", src, out, stamps, wgs, iters);
    run<0, false>("32x32x2  registers only", src, out, stamps, wgs, iters);
    run<1, false>("16x16x4  registers only", src, out, stamps, wgs, iters);
  }
  return 0;
}
