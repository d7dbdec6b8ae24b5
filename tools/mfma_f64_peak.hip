// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (the fp64
// MFMA peak is not in the local guides; SURVEY.md §8d asks to measure it).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o tools/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_peak(double *out, int iters, double a0, double b0) {
  v4f64 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// v_fma_f64 vector peak for comparison
__global__ __launch_bounds__(256) void k_fma(double *out, int iters, double a0, double b0) {
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fma(a, acc[i], b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Do the fp64 MFMA and the fp64 VALU share a pipe?  Waves 0, 2 of a workgroup issue
// MFMAs, waves 1, 3 v_fma_f64 (with 2 workgroups per CU every SIMD holds one wave of
// each kind): separate pipes -> time = max of the two alone, one pipe -> their sum.
__global__ __launch_bounds__(256) void k_mixed(double *out, int it_mfma, int it_fma, double a0, double b0) {
  const int wv = threadIdx.x >> 6;
  double a = a0 + threadIdx.x * 1e-9, b = b0, s = 0;
  if (wv & 1) {
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = i;
    for (int it = 0; it < it_fma; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fma(a, acc[i], b);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
  } else {
    v4f64 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
double time_ms(F f) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f();  // warm
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) f();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  double *out; (void)hipMalloc(&out, sizeof(double) * 256 * 4096);
  const int iters = 4000;
  for (int wg_per_cu : {1, 2, 4}) {
    const int grid = 256 * wg_per_cu;
    double ms1 = time_ms([&] { hipLaunchKernelGGL(k_peak<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
    double ms4 = time_ms([&] { hipLaunchKernelGGL(k_peak<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
    double ms8 = time_ms([&] { hipLaunchKernelGGL(k_peak<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
    auto tf = [&](double ms, int nacc) { return 2048.0 * nacc * iters * (grid * 4.0) / (ms * 1e-3) / 1e12; };
    printf("mfma_f64_16x16x4: %d WG/CU  1acc %.1f TF  4acc %.1f TF  8acc %.1f TF\n", wg_per_cu, tf(ms1, 1), tf(ms4, 4), tf(ms8, 8));
    double msf = time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
    printf("v_fma_f64        : %d WG/CU  %.1f TF\n", wg_per_cu, 2.0 * 16 * iters * (grid * 256.0) / (msf * 1e-3) / 1e12);
  }
  {
    // 2 workgroups per CU; MFMA waves: 8 x it MFMAs, FMA waves: 16 x it_f FMAs, sized to take about the same time alone
    const int grid = 512, itm = 2000, itf = 16000;
    double m_only = time_ms([&] { hipLaunchKernelGGL(k_mixed, dim3(grid), dim3(256), 0, 0, out, itm, 0, 1.0, 1e-3); });
    double f_only = time_ms([&] { hipLaunchKernelGGL(k_mixed, dim3(grid), dim3(256), 0, 0, out, 0, itf, 1.0000001, 1e-9); });
    double both = time_ms([&] { hipLaunchKernelGGL(k_mixed, dim3(grid), dim3(256), 0, 0, out, itm, itf, 1.0000001, 1e-9); });
    printf("fp64 pipe sharing (1 MFMA wave + 1 FMA wave per SIMD): MFMA alone %.3f ms, FMA alone %.3f ms, together %.3f ms "
           "(sum %.3f, max %.3f)\n", m_only, f_only, both, m_only + f_only, m_only > f_only ? m_only : f_only);
    printf("  cycles per MFMA at 2.4 GHz (alone): %.1f;  per wave64 v_fma_f64: %.1f\n",
           m_only * 1e-3 * 2.4e9 / (8.0 * itm), f_only * 1e-3 * 2.4e9 / (16.0 * itf));
  }
  return 0;
}
