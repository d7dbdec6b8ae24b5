// ba_dense_sched.cpp — see ba_dense_sched.h.  Host-only.
#include "ba_dense_sched.h"

#include <cstdlib>
#include <map>

#include <algorithm>
#include <tuple>

namespace ba {

namespace {

// One schedule.  `relaxed`: tiles of up to about twice the minimum degree may
// enter a level, instead of only the (nearly) minimum-degree ones.  For a block
// tridiagonal pattern both give odd-even cyclic reduction; for wider bands the
// strict rule only ever finds the two ends of the band (n/2 levels), the
// relaxed one eliminates every (band+1)-th tile at once (O(log n) levels, about
// twice the fill).
void build_one(int ncb, const std::vector<uint8_t> &adj_in, bool natural_order, int nb,
               bool relaxed, DenseSchedule &s) {
  const int n = ncb;
  s = DenseSchedule();
  s.nb = nb;
  s.ncb = n;
  std::vector<uint8_t> A(adj_in);
  auto at = [&](int a, int b) -> uint8_t & { return A[(size_t)a * n + b]; };
  for (int v = 0; v < n; ++v) at(v, v) = 0;
  std::vector<uint8_t> alive(n, 1), blocked(n, 0);
  std::vector<int> deg(n, 0);
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b)
      if (a != b && (at(a, b) || at(b, a))) {
        at(a, b) = at(b, a) = 1;
      }
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) deg[a] += at(a, b);
  std::vector<std::vector<int>> rows_of(n);
  s.pos_of_tile.assign(n, -1);
  s.tile_at_pos.clear();
  s.lev_ptr.assign(1, 0);
  int remaining = n, next = 0;
  std::vector<int> cand, level, nv;
  while (remaining > 0) {
    level.clear();
    if (natural_order) {
      for (int v = 0; v < n; ++v)
        if (alive[v]) {
          level.push_back(v);
          break;
        }
    } else {
      int mind = n + 1;
      for (int v = 0; v < n; ++v)
        if (alive[v]) mind = std::min(mind, deg[v]);
      const int thr = relaxed ? 2 * mind + 1 : mind + std::max(1, mind / 4);
      cand.clear();
      for (int v = 0; v < n; ++v)
        if (alive[v] && deg[v] <= thr) cand.push_back(v);
      std::stable_sort(cand.begin(), cand.end(),
                       [&](int a, int b) { return deg[a] < deg[b]; });
      std::fill(blocked.begin(), blocked.end(), 0);
      for (int v : cand) {
        if (blocked[v]) continue;
        level.push_back(v);
        blocked[v] = 1;
        for (int u = 0; u < n; ++u)
          if (alive[u] && at(v, u)) blocked[u] = 1;
      }
      std::sort(level.begin(), level.end());
    }
    // eliminate the independent set: neighbours become cliques (fill-in)
    for (int v : level) {
      nv.clear();
      for (int u = 0; u < n; ++u)
        if (alive[u] && u != v && at(v, u)) nv.push_back(u);
      rows_of[v] = nv;
      for (size_t a = 0; a < nv.size(); ++a)
        for (size_t b = a + 1; b < nv.size(); ++b)
          if (!at(nv[a], nv[b])) {
            at(nv[a], nv[b]) = at(nv[b], nv[a]) = 1;
            deg[nv[a]]++;
            deg[nv[b]]++;
          }
    }
    for (int v : level) {
      alive[v] = 0;
      for (int u : rows_of[v]) deg[u]--;
      s.pos_of_tile[v] = next++;
      s.tile_at_pos.push_back(v);
      --remaining;
    }
    s.lev_ptr.push_back(next);
  }
  s.nlev = (int)s.lev_ptr.size() - 1;

  // rows per position (ascending positions, rhs block last)
  s.row_ptr.assign(n + 1, 0);
  s.rows.clear();
  double nz_tiles = n;
  for (int p = 0; p < n; ++p) {
    std::vector<int> r;
    for (int u : rows_of[s.tile_at_pos[p]]) r.push_back(s.pos_of_tile[u]);
    std::sort(r.begin(), r.end());
    nz_tiles += (double)r.size();
    for (int I : r) s.rows.push_back(I);
    s.rows.push_back(n);
    s.row_ptr[p + 1] = (int)s.rows.size();
    const double m = (double)r.size() + 1.0;  // incl. rhs row block
    const double c3 = 64.0 * 64.0 * 64.0;
    s.flops += c3 / 3.0 + m * c3 + m * (m + 1.0) * c3;
  }
  s.fill = nz_tiles / ((double)n * (n + 1) / 2.0);

  // work lists per level
  s.item_ptr.assign(1, 0);
  s.tgt_ptr.assign(1, 0);
  s.tgt_first.clear();
  s.tgt_src_ptr.clear();
  std::vector<std::tuple<int, int, int>> trip;
  for (int l = 0; l < s.nlev; ++l) {
    trip.clear();
    for (int p = s.lev_ptr[l]; p < s.lev_ptr[l + 1]; ++p) {
      const int b = s.row_ptr[p], e = s.row_ptr[p + 1];
      for (int a = b; a < e; ++a) {
        s.item_t.push_back(p);
        s.item_I.push_back(s.rows[a]);
        for (int c = b; c <= a; ++c)
          if (s.rows[c] < n) trip.emplace_back(s.rows[a], s.rows[c], p);
      }
    }
    s.item_ptr.push_back((int)s.item_t.size());
    // target-centric: every (I,J) tile touched in this level, with the panels
    // that update it in ascending position order
    // (targets in a column that is eliminated in the NEXT level come first: with
    //  lookahead — dense patterns — they are updated before the rest, so that the next
    //  level's factorisation can start beside the bulk of this level's update)
    auto next_level = [&](int J) { return l + 1 < s.nlev && J >= s.lev_ptr[l + 1] && J < s.lev_ptr[l + 2]; };
    std::sort(trip.begin(), trip.end(), [&](const std::tuple<int, int, int> &x, const std::tuple<int, int, int> &y) {
      const int nx = next_level(std::get<1>(x)) ? 0 : 1, ny = next_level(std::get<1>(y)) ? 0 : 1;
      if (nx != ny) return nx < ny;
      return x < y;
    });
    int n_first = 0;
    for (size_t k = 0; k < trip.size(); ++k) {
      const bool is_new = k == 0 ||
                          std::get<0>(trip[k]) != std::get<0>(trip[k - 1]) ||
                          std::get<1>(trip[k]) != std::get<1>(trip[k - 1]);
      if (is_new) {
        s.tgt_I.push_back(std::get<0>(trip[k]));
        s.tgt_J.push_back(std::get<1>(trip[k]));
        s.tgt_src_ptr.push_back((int)s.src_t.size());
        n_first += next_level(std::get<1>(trip[k])) ? 1 : 0;
      }
      s.src_t.push_back(std::get<2>(trip[k]));
    }
    s.tgt_first.push_back(n_first);
    s.tgt_ptr.push_back((int)s.tgt_I.size());
  }
  s.tgt_src_ptr.push_back((int)s.src_t.size());
  s.tgt_desc.assign(8 * s.tgt_I.size(), -1);
  for (size_t tg = 0; tg < s.tgt_I.size(); ++tg) {
    int *q = &s.tgt_desc[8 * tg];
    q[0] = s.tgt_I[tg];
    q[1] = s.tgt_J[tg];
    q[2] = s.tgt_src_ptr[tg + 1] - s.tgt_src_ptr[tg];
    q[3] = s.tgt_src_ptr[tg];
    for (int k = 0; k < 4 && k < q[2]; ++k) q[4 + k] = s.src_t[q[3] + k];
  }
  // ---- fused schedule with lazily applied contributions ----
  {
    s.fused_ok = true;
    long long total = 0;
    for (int p = 0; p < n; ++p) {
      const long long nrow = s.row_ptr[p + 1] - s.row_ptr[p];
      if (nrow > dense_max_fused_rows(nb)) s.fused_ok = false;
      total += nrow * (nrow + 1) / 2 - 1;
    }
    if (total > kMaxContrib) s.fused_ok = false;
    if (s.fused_ok) {
      std::map<std::pair<int, int>, std::vector<int>> pending;
      s.f_desc.assign(16 * (size_t)n, -1);
      s.f_pend.clear();
      int next = 0;
      for (int p = 0; p < n; ++p) {  // positions are in elimination order
        int *q = &s.f_desc[16 * (size_t)p];
        const int b = s.row_ptr[p], e = s.row_ptr[p + 1];
        q[0] = e - b;
        q[1] = b;
        q[2] = (int)s.f_pend.size() / 2;
        auto take = [&](int I, int slot) {
          auto it = pending.find({I, p});
          if (it == pending.end()) return;
          for (int cid : it->second) {
            s.f_pend.push_back(slot);
            s.f_pend.push_back(cid);
          }
          pending.erase(it);
        };
        take(p, -1);
        for (int a = b; a < e; ++a) take(s.rows[a], a - b);
        q[3] = (int)s.f_pend.size() / 2 - q[2];
        q[4] = next;
        int k = 0;
        for (int a = b; a < e; ++a)
          for (int c = b; c <= a; ++c)
            if (s.rows[c] < n) {
              pending[{s.rows[a], s.rows[c]}].push_back(next + k);
              ++k;
            }
        q[5] = k;
        q[6] = q[7] = 0;
        for (int a = 0; a < 8 && b + a < e; ++a) q[8 + a] = s.rows[b + a];
        next += k;
      }
      s.n_contrib = next;
    }
  }
  s.row_desc.assign(16 * (size_t)n, -1);
  for (int p = 0; p < n; ++p) {
    int *q = &s.row_desc[16 * (size_t)p];
    q[0] = s.row_ptr[p + 1] - s.row_ptr[p];
    q[1] = s.row_ptr[p];
    s.max_rows = std::max(s.max_rows, q[0]);
    for (int k = 2; k < 8; ++k) q[k] = 0;
    for (int a = 0; a < 8 && a < q[0]; ++a) q[8 + a] = s.rows[q[1] + a];
  }
  s.back_desc.assign(8 * (size_t)n, -1);
  for (int p = 0; p < n; ++p) {
    int *q = &s.back_desc[8 * (size_t)p];
    int cnt = 0;
    for (int a = s.row_ptr[p]; a < s.row_ptr[p + 1] && s.rows[a] < n; ++a) ++cnt;
    q[0] = cnt;
    q[1] = s.row_ptr[p];
    for (int k = 0; k < 6 && k < cnt; ++k) q[2 + k] = s.rows[q[1] + k];
  }
}

}  // namespace

void build_dense_schedule(int ncb, const std::vector<uint8_t> &adj, bool natural_order, int nb,
                          DenseSchedule &s) {
  const char *force = getenv("BA_DENSE_ORDER");  // "strict" | "relaxed" (developer knob)
  if (natural_order || (force && force[0] == 's')) {
    build_one(ncb, adj, natural_order, nb, false, s);
    return;
  }
  if (force && force[0] == 'r') {
    build_one(ncb, adj, false, nb, true, s);
    return;
  }
  // the solve is a chain of dependent launches per level whose length grows
  // mildly with the row tiles a column carries: keep the shorter chain
  DenseSchedule relaxed;
  build_one(ncb, adj, false, nb, false, s);
  build_one(ncb, adj, false, nb, true, relaxed);
  auto chain = [](const DenseSchedule &q) { return q.nlev * (1.0 + 0.1 * q.max_rows); };
  if (chain(relaxed) < chain(s)) s = relaxed;
}

}  // namespace ba
