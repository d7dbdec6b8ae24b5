// ba_plan.cpp — see ba_plan.h.  Host-only, no HIP.
#include "ba_plan.h"

#include "ba_dense_sched.h"

#include <algorithm>
#include <numeric>

namespace ba {

namespace {

// Internal pose order: optimised poses in input order, then fixed poses.
void order_poses(const PlanInput &in, std::vector<int32_t> &int_of_user,
                 std::vector<int32_t> &user_of_int,
                 std::vector<int32_t> &jopt_of_user, int &N) {
  int_of_user.assign(in.n_pose, -1);
  user_of_int.clear();
  jopt_of_user.assign(in.n_pose, -1);
  N = 0;
  for (int p = 0; p < in.n_pose; ++p)
    if (!in.pose_fixed[p]) {
      jopt_of_user[p] = N;
      int_of_user[p] = N++;
      user_of_int.push_back(p);
    }
  int nxt = N;
  for (int p = 0; p < in.n_pose; ++p)
    if (in.pose_fixed[p]) {
      int_of_user[p] = nxt++;
      user_of_int.push_back(p);
    }
}

// Locality order of all points: (first observing internal pose, input index).
void locality_order(const PlanInput &in,
                    const std::vector<int32_t> &pose_int_of_user,
                    std::vector<int32_t> &order,
                    std::vector<int64_t> &obs_count) {
  std::vector<int32_t> first_pose(in.n_pt, in.n_pose);
  obs_count.assign(in.n_pt, 0);
  for (int64_t k = 0; k < in.n_obs; ++k) {
    const int q = in.obs_pt[k];
    const int j = pose_int_of_user[in.obs_pose[k]];
    if (j < first_pose[q]) first_pose[q] = j;
    obs_count[q]++;
  }
  // counting sort by first_pose keeps input index order inside a bucket
  std::vector<int64_t> bucket(in.n_pose + 2, 0);
  for (int q = 0; q < in.n_pt; ++q) bucket[first_pose[q] + 1]++;
  for (int b = 0; b <= in.n_pose; ++b) bucket[b + 1] += bucket[b];
  order.resize(in.n_pt);
  for (int q = 0; q < in.n_pt; ++q) order[bucket[first_pose[q]]++] = q;
}

void assign_owner(const PlanInput &in, const std::vector<int32_t> &order,
                  const std::vector<int64_t> &obs_count,
                  std::vector<int32_t> &owner) {
  owner.assign(in.n_pt, 0);
  if (in.world <= 1) return;
  // weight = observations + 1 so that unobserved points are spread too
  const long double total = (long double)in.n_obs + (long double)in.n_pt;
  long double cum = 0;
  for (int idx = 0; idx < in.n_pt; ++idx) {
    const int q = order[idx];
    int r = (int)((cum * in.world) / total);
    if (r >= in.world) r = in.world - 1;
    owner[q] = r;
    cum += (long double)obs_count[q] + 1;
  }
}

void make_chunks(const std::vector<int64_t> &ptr, int n_seg, int chunk,
                 std::vector<int32_t> &c_seg, std::vector<int64_t> &c_begin,
                 std::vector<int64_t> &c_end, std::vector<int32_t> &seg_cptr) {
  c_seg.clear();
  c_begin.clear();
  c_end.clear();
  seg_cptr.assign(n_seg + 1, 0);
  for (int s = 0; s < n_seg; ++s) {
    for (int64_t b = ptr[s]; b < ptr[s + 1]; b += chunk) {
      c_seg.push_back(s);
      c_begin.push_back(b);
      c_end.push_back(std::min<int64_t>(b + chunk, ptr[s + 1]));
    }
    seg_cptr[s + 1] = (int32_t)c_seg.size();
  }
}

}  // namespace

void partition_points(const PlanInput &in, std::vector<int32_t> &owner) {
  std::vector<int32_t> piu, pui, jou;
  int N = 0;
  order_poses(in, piu, pui, jou, N);
  std::vector<int32_t> order;
  std::vector<int64_t> cnt;
  locality_order(in, piu, order, cnt);
  assign_owner(in, order, cnt, owner);
}

std::string build_plan(const PlanInput &in, Plan &pl) {
  if (in.n_cam <= 0) return "no cameras";
  if (in.n_pose <= 0) return "no poses";
  if (in.n_pt <= 0) return "no points";
  if (in.world < 1 || in.rank < 0 || in.rank >= in.world)
    return "bad rank/world";
  for (int64_t k = 0; k < in.n_obs; ++k) {
    if (in.obs_cam[k] < 0 || in.obs_cam[k] >= in.n_cam)
      return "observation with invalid camera index";
    if (in.obs_pose[k] < 0 || in.obs_pose[k] >= in.n_pose)
      return "observation with invalid pose index";
    if (in.obs_pt[k] < 0 || in.obs_pt[k] >= in.n_pt)
      return "observation with invalid point index";
  }
  pl = Plan();
  pl.n_cam = in.n_cam;
  pl.n_pose = in.n_pose;
  pl.n_pt_global = in.n_pt;
  pl.n_obs_global = in.n_obs;
  order_poses(in, pl.pose_int_of_user, pl.pose_user_of_int, pl.jopt_of_user,
              pl.N);
  const int N = pl.N;

  // ---- points: global opt index (input order), locality order, owners ----
  pl.iopt_of_user.assign(in.n_pt, -1);
  pl.M_global = 0;
  for (int q = 0; q < in.n_pt; ++q)
    if (!in.pt_fixed[q]) pl.iopt_of_user[q] = pl.M_global++;
  std::vector<int32_t> order;
  std::vector<int64_t> cnt;
  locality_order(in, pl.pose_int_of_user, order, cnt);
  assign_owner(in, order, cnt, pl.owner);

  pl.pt_int_of_user.assign(in.n_pt, -1);
  pl.pt_user_of_int.clear();
  for (int idx = 0; idx < in.n_pt; ++idx) {
    const int q = order[idx];
    if (pl.owner[q] == in.rank && !in.pt_fixed[q]) {
      pl.pt_int_of_user[q] = (int32_t)pl.pt_user_of_int.size();
      pl.pt_user_of_int.push_back(q);
    }
  }
  pl.M = (int)pl.pt_user_of_int.size();
  for (int idx = 0; idx < in.n_pt; ++idx) {
    const int q = order[idx];
    if (pl.owner[q] == in.rank && in.pt_fixed[q]) {
      pl.pt_int_of_user[q] = (int32_t)pl.pt_user_of_int.size();
      pl.pt_user_of_int.push_back(q);
    }
  }
  pl.n_pt = (int)pl.pt_user_of_int.size();
  const int M = pl.M;

  // ---- landmark-major observation list ----
  std::vector<int64_t> sel;
  sel.reserve(in.n_obs / in.world + 16);
  for (int64_t k = 0; k < in.n_obs; ++k)
    if (pl.pt_int_of_user[in.obs_pt[k]] >= 0) sel.push_back(k);
  pl.n_obs = (int64_t)sel.size();
  auto key_lm = [&](int64_t k) -> uint64_t {
    return ((uint64_t)(uint32_t)pl.pt_int_of_user[in.obs_pt[k]] << 32) |
           (uint32_t)pl.pose_int_of_user[in.obs_pose[k]];
  };
  {
    std::vector<uint64_t> keys(sel.size());
    for (size_t s = 0; s < sel.size(); ++s) keys[s] = key_lm(sel[s]);
    std::vector<int64_t> perm(sel.size());
    std::iota(perm.begin(), perm.end(), (int64_t)0);
    std::stable_sort(perm.begin(), perm.end(),
                     [&](int64_t a, int64_t b) { return keys[a] < keys[b]; });
    std::vector<int64_t> sorted(sel.size());
    for (size_t s = 0; s < sel.size(); ++s) sorted[s] = sel[perm[s]];
    sel.swap(sorted);
  }
  pl.obs_idx.resize((size_t)pl.n_obs * 4);
  pl.obs_uv.resize((size_t)pl.n_obs * 2);
  pl.lm_obs_ptr.assign(M + 1, 0);
  pl.lm_pair_ptr.assign(M + 1, 0);
  pl.pair_pose.clear();
  pl.pair_lm.clear();
  for (int64_t s = 0; s < pl.n_obs; ++s) {
    const int64_t k = sel[s];
    const int32_t pi = pl.pt_int_of_user[in.obs_pt[k]];
    const int32_t ji = pl.pose_int_of_user[in.obs_pose[k]];
    pl.obs_idx[4 * s + 0] = in.obs_cam[k];
    pl.obs_idx[4 * s + 1] = ji;
    pl.obs_idx[4 * s + 2] = pi;
    pl.obs_idx[4 * s + 3] = -1;
    pl.obs_uv[2 * s + 0] = in.obs_uv[2 * k + 0];
    pl.obs_uv[2 * s + 1] = in.obs_uv[2 * k + 1];
    if (pi < M) {
      pl.lm_obs_ptr[pi + 1]++;
      if (ji < N) {
        const bool is_new = pl.pair_pose.empty() || pl.pair_lm.back() != pi ||
                            pl.pair_pose.back() != ji;
        if (is_new) {
          pl.pair_pose.push_back(ji);
          pl.pair_lm.push_back(pi);
          pl.lm_pair_ptr[pi + 1]++;
        } else {
          // same pair as the previous (adjacent) observation: the earlier
          // insertion loses its B_ji to this one (reference :826)
          pl.obs_idx[4 * (s - 1) + 3] = -1;
        }
        pl.obs_idx[4 * s + 3] = (int32_t)(pl.pair_pose.size() - 1);
      }
    }
  }
  for (int i = 0; i < M; ++i) {
    pl.lm_obs_ptr[i + 1] += pl.lm_obs_ptr[i];
    pl.lm_pair_ptr[i + 1] += pl.lm_pair_ptr[i];
  }
  pl.n_obs_opt = pl.lm_obs_ptr[M];
  pl.P = (int64_t)pl.pair_pose.size();
  if (pl.P >= (int64_t)INT32_MAX) return "too many pairs for int32 pair ids";

  // ---- pose-major observation list (optimisable poses) ----
  {
    std::vector<int64_t> psel;
    psel.reserve(pl.n_obs);
    for (int64_t s = 0; s < pl.n_obs; ++s)
      if (pl.obs_idx[4 * s + 1] < N) psel.push_back(s);
    // stable counting sort by pose keeps (point, insertion) order inside
    pl.pose_obs_ptr.assign(N + 1, 0);
    for (int64_t s : psel) pl.pose_obs_ptr[pl.obs_idx[4 * s + 1] + 1]++;
    for (int j = 0; j < N; ++j) pl.pose_obs_ptr[j + 1] += pl.pose_obs_ptr[j];
    pl.n_pobs = (int64_t)psel.size();
    pl.pobs_idx.resize((size_t)pl.n_pobs * 4);
    pl.pobs_uv.resize((size_t)pl.n_pobs * 2);
    std::vector<int64_t> cur(pl.pose_obs_ptr.begin(), pl.pose_obs_ptr.end() - 1);
    for (int64_t s : psel) {
      const int j = pl.obs_idx[4 * s + 1];
      const int64_t d = cur[j]++;
      pl.pobs_idx[4 * d + 0] = pl.obs_idx[4 * s + 0];
      pl.pobs_idx[4 * d + 1] = j;
      pl.pobs_idx[4 * d + 2] = pl.obs_idx[4 * s + 2];
      pl.pobs_idx[4 * d + 3] = 0;
      pl.pobs_uv[2 * d + 0] = pl.obs_uv[2 * s + 0];
      pl.pobs_uv[2 * d + 1] = pl.obs_uv[2 * s + 1];
    }
    make_chunks(pl.pose_obs_ptr, N, kPoseChunk, pl.achunk_pose,
                pl.achunk_begin, pl.achunk_end, pl.pose_achunk_ptr);
  }

  // ---- pose-major pair permutation ----
  {
    std::vector<int64_t> pptr(N + 1, 0);
    for (int64_t p = 0; p < pl.P; ++p) pptr[pl.pair_pose[p] + 1]++;
    for (int j = 0; j < N; ++j) pptr[j + 1] += pptr[j];
    pl.ppair.resize(pl.P);
    std::vector<int64_t> cur(pptr.begin(), pptr.end() - 1);
    for (int64_t p = 0; p < pl.P; ++p) pl.ppair[cur[pl.pair_pose[p]]++] = p;
    make_chunks(pptr, N, kRhsChunk, pl.rchunk_pose, pl.rchunk_begin,
                pl.rchunk_end, pl.pose_rchunk_ptr);
  }

  // ---- Schur triples sorted by (j, k, landmark) ----
  {
    int64_t T = 0;
    for (int i = 0; i < M; ++i) {
      const int64_t d = pl.lm_pair_ptr[i + 1] - pl.lm_pair_ptr[i];
      T += d * (d + 1) / 2;
    }
    pl.T = T;
    std::vector<int64_t> tp(T), tq(T);
    {
      int64_t t = 0;
      for (int i = 0; i < M; ++i)
        for (int64_t p = pl.lm_pair_ptr[i]; p < pl.lm_pair_ptr[i + 1]; ++p)
          for (int64_t q = p; q < pl.lm_pair_ptr[i + 1]; ++q) {
            tp[t] = p;
            tq[t] = q;
            ++t;
          }
    }
    // LSD radix: stable counting sort by k, then by j
    std::vector<int64_t> tp2(T), tq2(T);
    auto pass = [&](bool by_q, std::vector<int64_t> &sp,
                    std::vector<int64_t> &sq, std::vector<int64_t> &dp,
                    std::vector<int64_t> &dq) {
      std::vector<int64_t> cntv(N + 1, 0);
      for (int64_t t = 0; t < T; ++t)
        cntv[pl.pair_pose[by_q ? sq[t] : sp[t]] + 1]++;
      for (int j = 0; j < N; ++j) cntv[j + 1] += cntv[j];
      for (int64_t t = 0; t < T; ++t) {
        const int64_t d = cntv[pl.pair_pose[by_q ? sq[t] : sp[t]]]++;
        dp[d] = sp[t];
        dq[d] = sq[t];
      }
    };
    pass(true, tp, tq, tp2, tq2);
    pass(false, tp2, tq2, tp, tq);
    pl.tri_p.swap(tp);
    pl.tri_q.swap(tq);
    // blocks (every diagonal present, even without triples)
    pl.sblk_j.clear();
    pl.sblk_k.clear();
    pl.sblk_tri_ptr.clear();
    int64_t t = 0;
    for (int j = 0; j < N; ++j) {
      bool have_diag = false;
      while (t < T && pl.pair_pose[pl.tri_p[t]] == j) {
        const int k = pl.pair_pose[pl.tri_q[t]];
        have_diag = true;
        pl.sblk_j.push_back(j);
        pl.sblk_k.push_back(k);
        pl.sblk_tri_ptr.push_back(t);
        while (t < T && pl.pair_pose[pl.tri_p[t]] == j &&
               pl.pair_pose[pl.tri_q[t]] == k)
          ++t;
      }
      if (!have_diag) {
        pl.sblk_j.push_back(j);
        pl.sblk_k.push_back(j);
        pl.sblk_tri_ptr.push_back(t);
      }
    }
    pl.sblk_tri_ptr.push_back(T);
    pl.B = (int64_t)pl.sblk_j.size();
    make_chunks(pl.sblk_tri_ptr, (int)pl.B, kTriChunk, pl.tchunk_blk,
                pl.tchunk_begin, pl.tchunk_end, pl.sblk_tchunk_ptr);
  }

  // ---- tile pattern of S (global: every shard factors the same matrix) ----
  {
    // tiles = groups of kPosesPerTile consecutive optimised poses
    pl.ncb = std::max(1, (N + kPosesPerTile - 1) / kPosesPerTile);
    const int ncb = pl.ncb;
    pl.tile_nz.assign((size_t)ncb * ncb, 0);
    for (int t = 0; t < ncb; ++t) pl.tile_nz[(size_t)t * ncb + t] = 1;
    auto mark = [&](int j, int k) {  // optimised poses j, k are coupled
      const int a = j / kPosesPerTile, b = k / kPosesPerTile;
      pl.tile_nz[(size_t)a * ncb + b] = 1;
      pl.tile_nz[(size_t)b * ncb + a] = 1;
    };
    if (in.world == 1) {
      for (int64_t bk = 0; bk < pl.B; ++bk) mark(pl.sblk_j[bk], pl.sblk_k[bk]);
    } else {
      // all (landmark, pose) pairs of the FULL problem, both optimisable
      std::vector<uint64_t> keys;
      keys.reserve(in.n_obs);
      for (int64_t k = 0; k < in.n_obs; ++k) {
        const int j = pl.pose_int_of_user[in.obs_pose[k]];
        if (j < N && !in.pt_fixed[in.obs_pt[k]])
          keys.push_back(((uint64_t)(uint32_t)in.obs_pt[k] << 32) | (uint32_t)j);
      }
      std::sort(keys.begin(), keys.end());
      keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
      size_t a = 0;
      while (a < keys.size()) {
        size_t b = a;
        while (b < keys.size() && (keys[b] >> 32) == (keys[a] >> 32)) ++b;
        for (size_t u = a; u < b; ++u)
          for (size_t v = u; v < b; ++v)
            mark((int)(uint32_t)keys[u], (int)(uint32_t)keys[v]);
        a = b;
      }
    }
  }
  return std::string();
}

}  // namespace ba
