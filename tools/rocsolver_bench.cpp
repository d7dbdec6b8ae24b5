// Developer tool (never on the product path): rocSOLVER dpotrf + dpotrs on a random
// SPD system of order n — the vendor comparator for the dense sweep of
// ba_dense.hip (tools/dense_bench.py measures ours on the same kind of matrix).
//   hipcc -O2 --offload-arch=gfx950 tools/rocsolver_bench.cpp -o tools/rocsolver_bench -lrocsolver -lrocblas
//   tools/rocsolver_bench [n = 5970] [reps = 5]
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHECK(x)                                                          \
  do {                                                                    \
    if ((x) != 0) {                                                       \
      std::printf("FAILED %s (line %d)\n", #x, __LINE__);                 \
      return 1;                                                           \
    }                                                                     \
  } while (0)

int main(int argc, char **argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 5970, reps = argc > 2 ? std::atoi(argv[2]) : 5;
  const int kq = 256;
  std::mt19937_64 gen(0);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::uniform_real_distribution<double> ud(0.0, 1.0);
  // A = Q Q^T + n I + diag(u),  Q : n x 256  (same law as tools/dense_bench.py)
  std::vector<double> Q((size_t)n * kq), A((size_t)n * n), b(n);
  for (double &v : Q) v = nd(gen);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0.0;
      for (int k = 0; k < kq; ++k) s += Q[(size_t)i * kq + k] * Q[(size_t)j * kq + k];
      A[(size_t)j * n + i] = A[(size_t)i * n + j] = s;
    }
  for (int i = 0; i < n; ++i) {
    A[(size_t)i * n + i] += n + ud(gen);
    b[i] = nd(gen);
  }
  double *dA0, *dA, *dB;
  int *dinfo;
  CHECK(hipMalloc(&dA0, sizeof(double) * n * n));
  CHECK(hipMalloc(&dA, sizeof(double) * n * n));
  CHECK(hipMalloc(&dB, sizeof(double) * n));
  CHECK(hipMalloc(&dinfo, sizeof(int)));
  CHECK(hipMemcpy(dA0, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
  rocblas_handle h;
  CHECK(rocblas_create_handle(&h));
  hipEvent_t e0, e1, e2;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventCreate(&e2));
  double best_f = 1e30, best_s = 1e30;
  std::vector<double> x(n);
  for (int r = 0; r < reps + 1; ++r) {  // first pass = warm-up
    CHECK(hipMemcpy(dA, dA0, sizeof(double) * n * n, hipMemcpyDeviceToDevice));
    CHECK(hipMemcpy(dB, b.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    CHECK(rocsolver_dpotrf(h, rocblas_fill_lower, n, dA, n, dinfo));
    CHECK(hipEventRecord(e1, 0));
    CHECK(rocsolver_dpotrs(h, rocblas_fill_lower, n, 1, dA, n, dB, n));
    CHECK(hipEventRecord(e2, 0));
    CHECK(hipEventSynchronize(e2));
    float tf = 0, ts = 0;
    CHECK(hipEventElapsedTime(&tf, e0, e1));
    CHECK(hipEventElapsedTime(&ts, e1, e2));
    if (r > 0) {
      best_f = std::min(best_f, (double)tf);
      best_s = std::min(best_s, (double)ts);
    }
  }
  CHECK(hipMemcpy(x.data(), dB, sizeof(double) * n, hipMemcpyDeviceToHost));
  double res = 0, bn = 0;
  for (int i = 0; i < n; ++i) {
    double s = -b[i];
    for (int j = 0; j < n; ++j) s += A[(size_t)i * n + j] * x[j];
    res = std::max(res, std::fabs(s));
    bn = std::max(bn, std::fabs(b[i]));
  }
  const double fl = (double)n * n * n / 3.0 + 4.0 * (double)n * n;
  std::printf("rocSOLVER n=%d: dpotrf %.3f ms + dpotrs %.3f ms = %.3f ms  -> %.2f TFLOP/s (factor+solve), residual %.2e\n",
              n, best_f, best_s, best_f + best_s, fl / (best_f + best_s) / 1e9, res / bn);
  return 0;
}
