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

template <class F>
double time_ms(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();  // warm
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  double *out; hipMalloc(&out, sizeof(double) * 256 * 4096);
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
  return 0;
}
