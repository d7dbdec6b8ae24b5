// Host-only check of the level schedule of the reduced-system factorisation
// (csrc/ba_dense_sched.cpp): the schedule is EXECUTED with 1x1 "tiles" on a
// random SPD matrix of the given tile pattern and the result is compared with a
// plain dense Cholesky solve.  A missing fill tile, two adjacent tiles in one
// level, an incomplete source list or a wrong row list all show up as a wrong
// factor / solution.  Compiled with g++ by tests/test_plan_invariants.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "ba_dense_sched.h"

static int g_fail = 0;
#define CHECK(cond, ...)                                  \
  do {                                                    \
    if (!(cond)) {                                        \
      if (g_fail < 20) {                                  \
        std::printf("FAIL line %d: ", __LINE__);          \
        std::printf(__VA_ARGS__);                         \
        std::printf("\n");                                \
      }                                                   \
      ++g_fail;                                           \
    }                                                     \
  } while (0)

static void run(const std::string &name, int n, const std::vector<uint8_t> &adj, bool natural, int nb) {
  ba::DenseSchedule s;
  ba::build_dense_schedule(n, adj, natural, nb, s);
  CHECK(s.ncb == n && (int)s.pos_of_tile.size() == n && (int)s.tile_at_pos.size() == n, "%s: sizes", name.c_str());
  std::vector<int> seen(n, 0);
  for (int t = 0; t < n; ++t) {
    const int p = s.pos_of_tile[t];
    CHECK(p >= 0 && p < n && !seen[p] && s.tile_at_pos[p] == t, "%s: ordering is not a permutation", name.c_str());
    if (p >= 0 && p < n) seen[p] = 1;
  }
  CHECK(s.lev_ptr.front() == 0 && s.lev_ptr.back() == n && (int)s.lev_ptr.size() == s.nlev + 1, "%s: levels", name.c_str());
  // random SPD matrix with this pattern, in POSITION space, plus rhs
  std::mt19937_64 gen(7 + n);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  const int m = n + 1;  // row n = rhs
  std::vector<double> A((size_t)n * n, 0.0), b(n);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < i; ++j)
      if (adj[(size_t)s.tile_at_pos[i] * n + s.tile_at_pos[j]]) A[(size_t)i * n + j] = A[(size_t)j * n + i] = U(gen);
    b[i] = U(gen);
  }
  for (int i = 0; i < n; ++i) {
    double r = 1.0;
    for (int j = 0; j < n; ++j) r += std::fabs(A[(size_t)i * n + j]);
    A[(size_t)i * n + i] = r;  // diagonally dominant
  }
  // reference: dense Cholesky + solves
  std::vector<double> Lr = A, y(n), xr(n);
  for (int k = 0; k < n; ++k) {
    Lr[(size_t)k * n + k] = std::sqrt(Lr[(size_t)k * n + k]);
    for (int i = k + 1; i < n; ++i) Lr[(size_t)i * n + k] /= Lr[(size_t)k * n + k];
    for (int j = k + 1; j < n; ++j)
      for (int i = j; i < n; ++i) Lr[(size_t)i * n + j] -= Lr[(size_t)i * n + k] * Lr[(size_t)j * n + k];
  }
  for (int i = 0; i < n; ++i) {
    double v = b[i];
    for (int k = 0; k < i; ++k) v -= Lr[(size_t)i * n + k] * y[k];
    y[i] = v / Lr[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double v = y[i];
    for (int k = i + 1; k < n; ++k) v -= Lr[(size_t)k * n + i] * xr[k];
    xr[i] = v / Lr[(size_t)i * n + i];
  }
  // the schedule, executed with 1x1 tiles; M is (n+1) x n lower + rhs row
  std::vector<double> M((size_t)m * n, 0.0);
  std::vector<uint8_t> has((size_t)m * n, 0);  // tiles the schedule knows about
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) M[(size_t)i * n + j] = A[(size_t)i * n + j];
  for (int j = 0; j < n; ++j) M[(size_t)n * n + j] = b[j];
  for (int p = 0; p < n; ++p) {
    has[(size_t)p * n + p] = 1;
    int prev = p;
    for (int a = s.row_ptr[p]; a < s.row_ptr[p + 1]; ++a) {
      const int I = s.rows[a];
      CHECK(I > prev && I <= n, "%s: rows of %d not ascending / below", name.c_str(), p);
      prev = I;
      has[(size_t)I * n + p] = 1;
    }
    CHECK(s.row_ptr[p + 1] > s.row_ptr[p] && s.rows[s.row_ptr[p + 1] - 1] == n, "%s: rhs block must be the last row of %d",
          name.c_str(), p);
    const int *bd = &s.back_desc[8 * (size_t)p], *rd = &s.row_desc[16 * (size_t)p];
    CHECK(bd[0] == s.row_ptr[p + 1] - s.row_ptr[p] - 1 && bd[1] == s.row_ptr[p] && rd[0] == bd[0] + 1 && rd[1] == bd[1],
          "%s: inline records of %d", name.c_str(), p);
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j)
      if (A[(size_t)i * n + j] != 0.0) CHECK(has[(size_t)i * n + j], "%s: non-zero tile (%d,%d) missing", name.c_str(), i, j);
  for (int l = 0; l < s.nlev; ++l) {
    for (int p = s.lev_ptr[l]; p < s.lev_ptr[l + 1]; ++p) M[(size_t)p * n + p] = std::sqrt(M[(size_t)p * n + p]);
    for (int it = s.item_ptr[l]; it < s.item_ptr[l + 1]; ++it) {
      const int t = s.item_t[it], I = s.item_I[it];
      CHECK(t >= s.lev_ptr[l] && t < s.lev_ptr[l + 1] && has[(size_t)I * n + t], "%s: TRSM item outside its level", name.c_str());
      M[(size_t)I * n + t] /= M[(size_t)t * n + t];
    }
    for (int tg = s.tgt_ptr[l]; tg < s.tgt_ptr[l + 1]; ++tg) {
      const int I = s.tgt_I[tg], J = s.tgt_J[tg];
      CHECK(J < n && I >= J && has[(size_t)I * n + J], "%s: update target (%d,%d) is not a tile of the factor", name.c_str(), I, J);
      CHECK(J >= s.lev_ptr[l + 1], "%s: target column %d is eliminated in the same or an earlier level", name.c_str(), J);
      const int *td = &s.tgt_desc[8 * (size_t)tg];
      CHECK(td[0] == I && td[1] == J && td[2] == s.tgt_src_ptr[tg + 1] - s.tgt_src_ptr[tg] && td[3] == s.tgt_src_ptr[tg],
            "%s: target record", name.c_str());
      for (int q = s.tgt_src_ptr[tg]; q < s.tgt_src_ptr[tg + 1]; ++q) {
        const int t = s.src_t[q];
        CHECK(t >= s.lev_ptr[l] && t < s.lev_ptr[l + 1], "%s: source outside the level", name.c_str());
        M[(size_t)I * n + J] -= M[(size_t)I * n + t] * M[(size_t)J * n + t];
      }
    }
  }
  double eL = 0.0, ey = 0.0, ex = 0.0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      const double d = std::fabs(M[(size_t)i * n + j] - Lr[(size_t)i * n + j]);
      if (!has[(size_t)i * n + j]) CHECK(std::fabs(Lr[(size_t)i * n + j]) < 1e-13, "%s: fill at (%d,%d) not in the schedule", name.c_str(), i, j);
      eL = std::fmax(eL, d);
    }
  for (int j = 0; j < n; ++j) ey = std::fmax(ey, std::fabs(M[(size_t)n * n + j] - y[j]));
  std::vector<double> x(n);
  for (int l = s.nlev - 1; l >= 0; --l)
    for (int p = s.lev_ptr[l]; p < s.lev_ptr[l + 1]; ++p) {
      double v = M[(size_t)n * n + p];
      for (int a = s.row_ptr[p]; a < s.row_ptr[p + 1]; ++a)
        if (s.rows[a] < n) v -= M[(size_t)s.rows[a] * n + p] * x[s.rows[a]];
      x[p] = v / M[(size_t)p * n + p];
    }
  for (int i = 0; i < n; ++i) ex = std::fmax(ex, std::fabs(x[i] - xr[i]));
  CHECK(eL < 1e-10 && ey < 1e-10 && ex < 1e-10, "%s: factor %.2e  forward %.2e  solution %.2e", name.c_str(), eL, ey, ex);
  std::printf("%-28s n=%4d nb=%d levels=%3d fill=%.3f  |dL| %.1e |dx| %.1e\n", name.c_str(), n, nb, s.nlev, s.fill, eL, ex);
}

int main() {
  std::mt19937 gen(3);
  for (int nb : {32, 64}) {
    for (int n : {1, 2, 7, 64, 199}) {  // block tridiagonal (C3/C4) and wider bands (C2)
      for (int band : {1, 2, 5}) {
        std::vector<uint8_t> adj((size_t)n * n, 0);
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j)
            if (i != j && std::abs(i - j) <= band) adj[(size_t)i * n + j] = 1;
        run("band " + std::to_string(band), n, adj, false, nb);
      }
    }
    {  // dense
      const int n = 12;
      std::vector<uint8_t> adj((size_t)n * n, 1);
      for (int i = 0; i < n; ++i) adj[(size_t)i * n + i] = 0;
      run("dense", n, adj, false, nb);
      run("dense, natural order", n, adj, true, nb);
    }
    for (int rep = 0; rep < 4; ++rep) {  // random sparse + loop closures
      const int n = 40 + 30 * rep;
      std::vector<uint8_t> adj((size_t)n * n, 0);
      for (int i = 0; i + 1 < n; ++i) adj[(size_t)i * n + i + 1] = adj[(size_t)(i + 1) * n + i] = 1;
      for (int k = 0; k < n; ++k) {
        const int i = gen() % n, j = gen() % n;
        if (i != j) adj[(size_t)i * n + j] = adj[(size_t)j * n + i] = 1;
      }
      run("chain + random closures", n, adj, false, nb);
    }
    {  // disconnected components and an isolated tile
      const int n = 20;
      std::vector<uint8_t> adj((size_t)n * n, 0);
      for (int i = 0; i + 1 < 9; ++i) adj[(size_t)i * n + i + 1] = adj[(size_t)(i + 1) * n + i] = 1;
      for (int i = 10; i + 1 < n; ++i) adj[(size_t)i * n + i + 1] = adj[(size_t)(i + 1) * n + i] = 1;
      run("two chains + isolated tile", n, adj, false, nb);
    }
  }
  std::printf(g_fail ? "DENSE SCHEDULE CHECK FAILED (%d)\n" : "DENSE SCHEDULE CHECK OK\n", g_fail);
  return g_fail ? 1 : 0;
}
