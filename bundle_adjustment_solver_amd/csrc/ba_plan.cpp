// ba_plan.cpp — see ba_plan.h.  Host-only, no HIP.
#include "ba_plan.h"

#include "ba_dense_sched.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <limits>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <thread>

#include <sys/mman.h>

namespace ba {

void *plan_big_alloc(size_t bytes) {
  void *p = nullptr;
  const size_t sz = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
  if (posix_memalign(&p, (size_t)2 << 20, sz) != 0 || !p) throw std::bad_alloc();
  (void)madvise(p, sz, MADV_HUGEPAGE);  // advisory: plain pages when THP is off
  return p;
}
void plan_big_free(void *p, size_t) { free(p); }

namespace {

// BA_PLAN_TIMES=1: wall time of the planner's phases on stderr (developer knob)
struct PhaseClock {
  const bool on = getenv("BA_PLAN_TIMES") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char *what) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[plan] %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};

// Static-chunk parallel loop over [0, n) on host threads (BA_PLAN_THREADS, default:
// the hardware's, at most 16).  Every use below writes disjoint outputs per index
// and all prefix sums stay sequential, so the plan is identical for every thread
// count (tests/cpp/plan_check.cpp compares it with the one-thread build).
int plan_threads() {
  static const int n = [] {
    const char *e = getenv("BA_PLAN_THREADS");
    int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(v, 16));
  }();
  return n;
}
// (grain: indices a thread should at least get — 4096 for per-landmark loops; loops over
//  runs / buckets of hundreds of landmarks pass a small one)
template <class F>
void parallel_for(int64_t n, const F &fn, int64_t grain = 4096) {
  const int nt = (int)std::min<int64_t>(plan_threads(), std::max<int64_t>(1, n / grain));
  if (nt <= 1) {
    fn((int64_t)0, n);
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t] { fn(n * t / nt, n * (t + 1) / nt); });
  for (auto &x : th) x.join();
}

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
  // (threaded by point range: every thread scans the whole list and keeps its own points)
  parallel_for(in.n_pt, [&](int64_t q0, int64_t q1) {
    for (int64_t k = 0; k < in.n_obs; ++k) {
      const int64_t q = in.obs_pt[k];
      if (q < q0 || q >= q1) continue;
      const int j = pose_int_of_user[in.obs_pose[k]];
      if (j < first_pose[q]) first_pose[q] = j;
      obs_count[q]++;
    }
  });
  // counting sort by first_pose keeps input index order inside a bucket
  std::vector<int64_t> bucket(in.n_pose + 2, 0);
  for (int q = 0; q < in.n_pt; ++q) bucket[first_pose[q] + 1]++;
  for (int b = 0; b <= in.n_pose; ++b) bucket[b + 1] += bucket[b];
  order.resize(in.n_pt);
  for (int q = 0; q < in.n_pt; ++q) order[bucket[first_pose[q]]++] = q;
}

// Landmarks per Schur super-run for a shard of M optimised landmarks: the kernel
// runs kSchurRunTarget workgroups at a time, so the runs are sized to fill a
// whole number of such rounds (C4: 500 k landmarks -> 2 rounds of 326 instead
// of 2.5 rounds of 256), at most kSchurSuperLandmarks and at least
// kSchurSuperMin each.
int schur_run_cap(int64_t M) {
  const int64_t rounds = std::max<int64_t>(
      1, (M + (int64_t)kSchurRunTarget * kSchurSuperLandmarks - 1) /
             ((int64_t)kSchurRunTarget * kSchurSuperLandmarks));
  int cap = (int)std::min<int64_t>(
      kSchurSuperLandmarks,
      std::max<int64_t>(kSchurSuperMin,
                        (M + kSchurRunTarget * rounds - 1) / (kSchurRunTarget * rounds)));
  if (const char *e = getenv("BA_SUP_CAP")) cap = std::max(1, atoi(e));  // tuning knob
  return cap;
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
    const int64_t len = ptr[s + 1] - ptr[s];
    if (len > 0) {
      // equal pieces (multiples of 64 so that a wave's last step is full)
      const int64_t nc = (len + chunk - 1) / chunk;
      int64_t piece = (len + nc - 1) / nc;
      piece = std::min<int64_t>((piece + 63) / 64 * 64, chunk);
      for (int64_t b = ptr[s]; b < ptr[s + 1]; b += piece) {
        c_seg.push_back(s);
        c_begin.push_back(b);
        c_end.push_back(std::min<int64_t>(b + piece, ptr[s + 1]));
      }
    }
    seg_cptr[s + 1] = (int32_t)c_seg.size();
  }
}

}  // namespace

// Lane table of one Schur super-run (k_schur_lds: 4 waves x 64 lanes).  Slot s
// gets an even number of lanes n_s in [2, 32], contiguous inside ONE wave, in
// proportion to its triple count; lane word = slot | h << 8 | sub2 << 9 |
// (n_s / 2) << 14, 0xff = idle lane.  Waves: longest-processing-time first on
// the triple counts; lanes inside a wave: repeatedly +2 to the slot with the
// most triples per lane.  Deterministic (ties by slot id).
void deal_lanes(const std::vector<int64_t> &tcount, std::vector<uint32_t> &out) {
  const int ns = (int)tcount.size();
  std::vector<int> ord(ns);
  std::iota(ord.begin(), ord.end(), 0);
  std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return tcount[a] > tcount[b]; });
  std::vector<int> wave_of(ns, 0), nl(ns, 2);
  int64_t load[4] = {0, 0, 0, 0};
  int cnt[4] = {0, 0, 0, 0};
  for (int s : ord) {
    int w = -1;
    for (int c = 0; c < 4; ++c)
      if (cnt[c] < 32 && (w < 0 || load[c] < load[w])) w = c;
    wave_of[s] = w;
    load[w] += tcount[s];
    cnt[w]++;
  }
  const size_t base = out.size();
  out.resize(base + 256, 0xffu);
  for (int w = 0; w < 4; ++w) {
    int free_lanes = 64 - 2 * cnt[w];
    while (free_lanes >= 2) {
      int best = -1;
      for (int s = 0; s < ns; ++s) {
        if (wave_of[s] != w || nl[s] >= 32) continue;
        // tcount[s] / nl[s] > tcount[best] / nl[best], in integers
        if (best < 0 || tcount[s] * nl[best] > tcount[best] * nl[s]) best = s;
      }
      if (best < 0 || tcount[best] == 0) break;
      nl[best] += 2;
      free_lanes -= 2;
    }
    int lane = 0;
    for (int s = 0; s < ns; ++s) {
      if (wave_of[s] != w) continue;
      for (int sub = 0; sub < nl[s]; ++sub)
        out[base + 64 * w + lane + sub] =
            (uint32_t)s | ((uint32_t)(sub & 1) << 8) | ((uint32_t)(sub >> 1) << 9) |
            ((uint32_t)(nl[s] >> 1) << 14);
      lane += nl[s];
    }
  }
}

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
  PhaseClock clk;
  {
    // first offending observation (the smallest index, whatever the thread count)
    std::vector<int64_t> bad((size_t)plan_threads() + 1, INT64_MAX);
    std::atomic<int> slot{0};
    parallel_for(in.n_obs, [&](int64_t k0, int64_t k1) {
      const int me = slot.fetch_add(1);
      for (int64_t k = k0; k < k1; ++k)
        if (in.obs_cam[k] < 0 || in.obs_cam[k] >= in.n_cam || in.obs_pose[k] < 0 || in.obs_pose[k] >= in.n_pose ||
            in.obs_pt[k] < 0 || in.obs_pt[k] >= in.n_pt) {
          bad[me] = k;
          break;
        }
    });
    const int64_t k = *std::min_element(bad.begin(), bad.end());
    if (k != INT64_MAX) {
      if (in.obs_cam[k] < 0 || in.obs_cam[k] >= in.n_cam) return "observation with invalid camera index";
      if (in.obs_pose[k] < 0 || in.obs_pose[k] >= in.n_pose) return "observation with invalid pose index";
      return "observation with invalid point index";
    }
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
  clk.lap("validate + order + owners");
  // ---- covisibility groups: landmarks with the identical OBSERVATION PATTERN —
  // the same sequence of (pose, camera) over their observations in landmark-major
  // order, fixed poses included — hence the identical set of optimisable poses
  // (1..kGrpMaxPoses of them), in groups of >= kGrpMinLandmarks come FIRST, group
  // after group (groups ordered by their patterns, i.e. still by first observing
  // pose; landmarks of a group in locality order).  k_lin_grp linearises a group
  // with one lane per pattern slot (pose and camera are lane constants: no
  // gathers, no index records, pose side in the same pass); k_schur_grp turns its
  // Schur contributions into one dense product.  Everything else keeps the
  // locality order and goes through the chunk / super-run kernels. ----
  // ---- observations of the OWNED points as a CSR over user point ids, every list in
  // (pose, insertion) order: built once, used by the covisibility grouping below and,
  // landmark by landmark in the final order, by the landmark-major list ----
  // (counting sort by point id, threaded by POINT RANGE: every thread scans the whole
  //  observation list — a sequential read — and files the observations of its own points,
  //  in input order: the result does not depend on the thread count)
  std::vector<int64_t> uo_ptr((size_t)in.n_pt + 1, 0);
  parallel_for(in.n_pt, [&](int64_t q0, int64_t q1) {
    for (int64_t k = 0; k < in.n_obs; ++k) {
      const int64_t q = in.obs_pt[k];
      if (q >= q0 && q < q1 && pl.owner[q] == in.rank) uo_ptr[q + 1]++;
    }
  });
  for (int q = 0; q < in.n_pt; ++q) uo_ptr[q + 1] += uo_ptr[q];
  // (packed records: every later pass reads a landmark's observations as ONE contiguous
  //  run instead of gathering four input arrays at random positions)
  struct ObsRecP {
    int32_t pose, cam;  // internal pose index, camera
    double u, v;
  };
  pvec<ObsRecP> uo_rec((size_t)uo_ptr[in.n_pt]);
  {
    pvec<int64_t> cur((size_t)in.n_pt);
    parallel_for(in.n_pt, [&](int64_t q0, int64_t q1) {
      for (int64_t q = q0; q < q1; ++q) cur[q] = uo_ptr[q];
      for (int64_t k = 0; k < in.n_obs; ++k) {
        const int64_t q = in.obs_pt[k];
        if (q < q0 || q >= q1 || pl.owner[q] != in.rank) continue;
        ObsRecP &r = uo_rec[cur[q]++];
        r.pose = pl.pose_int_of_user[in.obs_pose[k]];
        r.cam = in.obs_cam[k];
        r.u = in.obs_uv ? in.obs_uv[2 * k + 0] : 0.0;
        r.v = in.obs_uv ? in.obs_uv[2 * k + 1] : 0.0;
      }
    });
  }
  parallel_for(in.n_pt, [&](int64_t q0, int64_t q1) {
    for (int64_t q = q0; q < q1; ++q) {
      // stable insertion sort by pose (lists are short and nearly sorted: the usual
      // insertion order is pose-major)
      ObsRecP *a = uo_rec.data() + uo_ptr[q];
      const int64_t n = uo_ptr[q + 1] - uo_ptr[q];
      for (int64_t i = 1; i < n; ++i) {
        if (a[i - 1].pose <= a[i].pose) continue;
        const ObsRecP x = a[i];
        int64_t j = i;
        while (j > 0 && a[j - 1].pose > x.pose) {
          a[j] = a[j - 1];
          --j;
        }
        a[j] = x;
      }
    }
  });
  clk.lap("per-point observation lists");
  pl.M_grp = 0;
  pl.grp_range.clear();
  pl.lin_groups = !(getenv("BA_NO_LINGRP") && getenv("BA_NO_LINGRP")[0] == '1');
  if (pl.n_cam >= 65536) pl.lin_groups = false;  // (the pattern record packs the camera into 16 bits)
  if (pl.M > 0 && !(getenv("BA_NO_GROUPS") && getenv("BA_NO_GROUPS")[0] == '1')) {
    const int M0 = pl.M;
    std::vector<int64_t> kp(M0 + 1, 0);
    for (int i = 0; i < M0; ++i) {
      const int q = pl.pt_user_of_int[i];
      kp[i + 1] = kp[i] + (uo_ptr[q + 1] - uo_ptr[q]);
    }
    pvec<uint64_t> pat((size_t)kp[M0]);  // (pose << 32) | camera, per point in (pose, insertion) order
    parallel_for(M0, [&](int64_t i0, int64_t i1) {
      for (int64_t i = i0; i < i1; ++i) {
        const int q = pl.pt_user_of_int[i];
        int64_t w = kp[i];
        for (int64_t t = uo_ptr[q]; t < uo_ptr[q + 1]; ++t)
          pat[w++] = ((uint64_t)(uint32_t)uo_rec[t].pose << 32) | (uint32_t)uo_rec[t].cam;
      }
    });
    clk.lap("  groups: patterns");
    auto deg = [&](int i) { return (int)(kp[i + 1] - kp[i]); };
    auto dopt = [&](int i) {  // distinct optimisable poses
      int n = 0;
      int64_t last = -1;
      for (int64_t t = kp[i]; t < kp[i + 1]; ++t) {
        const int64_t ji = (int64_t)(pat[t] >> 32);
        if (ji < N && ji != last) ++n;
        last = ji;
      }
      return n;
    };
    auto less_sig = [&](int a, int b) {  // lexicographic on the patterns, shorter first on ties
      const int da = deg(a), db = deg(b);
      for (int t = 0; t < std::min(da, db); ++t)
        if (pat[kp[a] + t] != pat[kp[b] + t]) return pat[kp[a] + t] < pat[kp[b] + t];
      return da < db;
    };
    auto same_sig = [&](int a, int b) {
      return deg(a) == deg(b) && std::equal(pat.begin() + kp[a], pat.begin() + kp[a + 1], pat.begin() + kp[b]);
    };
    std::vector<int32_t> cand;
    std::vector<int32_t> dop(M0, 0);
    for (int i = 0; i < M0; ++i) {
      dop[i] = dopt(i);
      // (dop == 0: landmarks seen by fixed poses only — the first poses of a trajectory are
      //  usually fixed — have no pair and no Schur term, but they are linearised and
      //  back-substituted: groups of them keep those kernels' chunk launches empty)
      if (dop[i] <= kGrpMaxPoses && deg(i) >= 1 && deg(i) <= kGrpMaxObs) cand.push_back(i);
    }
    clk.lap("  groups: candidates");
    // stable sort by pattern (stable: locality order inside a group).  The candidates
    // arrive in locality order, i.e. already grouped by their first observing pose = the
    // first pattern word's pose: every run of equal first pose is sorted on its own, the
    // runs in parallel (the result is the stable sort of the whole list whenever the
    // first poses are non-decreasing, which the check below establishes).
    {
      std::vector<size_t> run0(1, 0);
      bool monotone = true;
      for (size_t t = 1; t < cand.size(); ++t) {
        const uint64_t a0 = pat[kp[cand[t - 1]]] >> 32, b0 = pat[kp[cand[t]]] >> 32;
        if (b0 < a0) monotone = false;
        if (b0 != a0) run0.push_back(t);
      }
      run0.push_back(cand.size());
      if (!monotone) {
        std::stable_sort(cand.begin(), cand.end(), less_sig);
      } else {
        parallel_for((int64_t)run0.size() - 1, [&](int64_t r0, int64_t r1) {
          for (int64_t r = r0; r < r1; ++r)  // (regular scenes: one pattern per run, nothing to sort)
            if (!std::is_sorted(cand.begin() + run0[r], cand.begin() + run0[r + 1], less_sig))
              std::stable_sort(cand.begin() + run0[r], cand.begin() + run0[r + 1], less_sig);
        }, 8);
      }
    }
    clk.lap("  groups: pattern sort");
    // ---- groups.  A run of candidates with the IDENTICAL pattern is an exact group (as
    // before).  SUPERSET groups (round 3) take what real visibility leaves of that —
    // occlusion, image borders, track loss make the patterns of one pose window differ
    // from landmark to landmark: all candidates with the same pose SPAN (first and last
    // observing pose) form one group whose pattern is the UNION of theirs; a member's
    // missing observations become padded slots (uv = NaN: weight 0, no cost) and its
    // missing pairs zero W records, so that the group kernels keep their one-lane-per-
    // (landmark, slot) layout.  Landmarks that lost their first or last pose altogether
    // join a group whose union contains their pattern.  (BA_NO_SUPERSET=1: exact groups
    // only; masks need k_lin_grp, so also off with BA_NO_LINGRP=1.)
    struct GroupBuild {
      std::vector<uint64_t> upat;     // union pattern, sorted
      std::vector<int32_t> members;   // preliminary landmark indices
      int d = 0;
      bool masked = false;
      int32_t first_pose = 0, last_pose = 0;  // span in user pose indices
    };
    std::vector<GroupBuild> groups;
    std::vector<int32_t> group_of(M0, -1);
    const bool no_superset_env = getenv("BA_NO_SUPERSET") && getenv("BA_NO_SUPERSET")[0] == '1';
    const bool superset = pl.lin_groups && !no_superset_env;
    // the pose SPAN of a landmark in USER pose indices (the registration order of the
    // poses is normally the trajectory; the internal order puts the fixed poses last,
    // which would throw every window that touches a fixed pose into one bucket)
    std::vector<int32_t> span_lo(M0, 0), span_hi(M0, 0);
    parallel_for(M0, [&](int64_t i0, int64_t i1) {
      for (int64_t i = i0; i < i1; ++i) {
        int32_t lo = INT32_MAX, hi = -1;
        for (int64_t t = kp[i]; t < kp[i + 1]; ++t) {
          const int32_t u = pl.pose_user_of_int[(size_t)(pat[t] >> 32)];
          lo = std::min(lo, u);
          hi = std::max(hi, u);
        }
        span_lo[i] = lo;
        span_hi[i] = hi;
      }
    });
    auto first_pose = [&](int i) { return span_lo[i]; };
    auto last_pose = [&](int i) { return span_hi[i]; };
    // a member of a superset group must list its (pose, camera) entries in strictly
    // increasing order: the slots of the union are in that order, and the pair's last
    // WRITER (reference :826: the last inserted observation of a pose) is then the last
    // valid slot of the pose.  (Duplicates and cameras inserted in descending order:
    // exact groups only.)
    auto has_dup = [&](int i) {
      for (int64_t t = kp[i] + 1; t < kp[i + 1]; ++t)
        if (pat[t] <= pat[t - 1]) return true;
      return false;
    };
    auto dopt_of = [&](const std::vector<uint64_t> &u) {
      int n = 0;
      int64_t last = -1;
      for (uint64_t w : u) {
        const int64_t ji = (int64_t)(w >> 32);
        if (ji < N && ji != last) ++n;
        last = ji;
      }
      return n;
    };
    // (the group builders append to `out`; the caller numbers the groups and files their
    //  members in group_of afterwards: buckets are processed on host threads)
    auto exact_runs = [&](size_t lo, size_t hi, std::vector<GroupBuild> &out) {
      size_t a = lo;
      while (a < hi) {
        size_t b = a + 1;
        while (b < hi && same_sig(cand[a], cand[b])) ++b;
        if ((int)(b - a) >= kGrpMinLandmarks) {
          GroupBuild gb;
          gb.upat.assign(pat.begin() + kp[cand[a]], pat.begin() + kp[cand[a] + 1]);
          gb.d = dop[cand[a]];
          gb.first_pose = first_pose(cand[a]);
          gb.last_pose = last_pose(cand[a]);
          for (size_t t = a; t < b; ++t) gb.members.push_back(cand[t]);
          out.push_back(std::move(gb));
        }
        a = b;
      }
    };
    auto adopt = [&](std::vector<GroupBuild> &from) {
      for (GroupBuild &gb : from) {
        for (int32_t i : gb.members) group_of[i] = (int32_t)groups.size();
        groups.push_back(std::move(gb));
      }
      from.clear();
    };
    if (!superset) {
      std::vector<GroupBuild> out;
      exact_runs(0, cand.size(), out);
      adopt(out);
    } else {
      // buckets of equal (first pose, last pose); inside a bucket the pattern order stays
      // (stable, by two counting passes — last pose, then first pose)
      {
        std::vector<int32_t> tmp(cand.size());
        auto pass = [&](const std::vector<int32_t> &key, const std::vector<int32_t> &src, std::vector<int32_t> &dst) {
          std::vector<int64_t> at((size_t)in.n_pose + 1, 0);
          for (int32_t i : src) at[(size_t)key[i] + 1]++;
          for (int b = 0; b < in.n_pose; ++b) at[b + 1] += at[b];
          for (int32_t i : src) dst[at[key[i]]++] = i;
        };
        pass(span_hi, cand, tmp);
        pass(span_lo, tmp, cand);
      }
      clk.lap("  groups: span sort");
      std::vector<size_t> bkt(1, 0);  // bucket boundaries
      for (size_t t = 1; t < cand.size(); ++t)
        if (first_pose(cand[t]) != first_pose(cand[t - 1]) || last_pose(cand[t]) != last_pose(cand[t - 1])) bkt.push_back(t);
      bkt.push_back(cand.size());
      const size_t nbkt = cand.empty() ? 0 : bkt.size() - 1;
      std::vector<std::vector<GroupBuild>> bout(nbkt);
      parallel_for((int64_t)nbkt, [&](int64_t k0, int64_t k1) {
        std::vector<uint64_t> uni;
        for (int64_t k = k0; k < k1; ++k) {
          const size_t a = bkt[k], b = bkt[k + 1];
          std::vector<GroupBuild> &out = bout[k];
          if (same_sig(cand[a], cand[b - 1])) {  // (sorted by pattern: first == last => all equal)
            exact_runs(a, b, out);
            continue;
          }
          // union of the duplicate-free members' patterns
          uni.clear();
          int64_t slots = 0;
          int n_el = 0;
          for (size_t t = a; t < b; ++t) {
            const int i = cand[t];
            if (has_dup(i)) continue;
            uni.insert(uni.end(), pat.begin() + kp[i], pat.begin() + kp[i + 1]);
            slots += deg(i);
            ++n_el;
          }
          std::sort(uni.begin(), uni.end());
          uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
          const int du = dopt_of(uni);
          if (n_el >= kGrpMinLandmarks && (int)uni.size() <= kGrpMaxObs && du <= kGrpMaxPoses &&
              (double)slots >= 0.55 * (double)n_el * (double)uni.size()) {
            GroupBuild gb;
            gb.upat = uni;
            gb.d = du;
            gb.first_pose = first_pose(cand[a]);
            gb.last_pose = last_pose(cand[a]);
            for (size_t t = a; t < b; ++t) {
              const int i = cand[t];
              if (has_dup(i)) continue;
              gb.members.push_back(i);
              if (deg(i) < (int)uni.size()) gb.masked = true;
            }
            out.push_back(std::move(gb));
            // (members with duplicate observations: exact runs among themselves)
          } else {
            exact_runs(a, b, out);
          }
        }
      }, 8);
      for (size_t k = 0; k < nbkt; ++k) adopt(bout[k]);
      clk.lap("  groups: buckets");
      // leftovers join a group whose union contains their pattern
      std::vector<std::vector<int32_t>> by_first(in.n_pose + 1);
      for (size_t gidx = 0; gidx < groups.size(); ++gidx)
        by_first[(size_t)groups[gidx].first_pose].push_back((int32_t)gidx);
      for (int i : cand) {
        if (group_of[i] >= 0 || has_dup(i)) continue;
        const int32_t f = first_pose(i), l = last_pose(i);
        bool joined = false;
        for (int32_t g0 = f; g0 >= 0 && g0 > f - kGrpMaxObs && !joined; --g0)
          for (int32_t gidx : by_first[g0]) {
            GroupBuild &gb = groups[gidx];
            if (gb.last_pose < l) continue;
            if (!std::includes(gb.upat.begin(), gb.upat.end(), pat.begin() + kp[i], pat.begin() + kp[i + 1])) continue;
            gb.members.push_back(i);
            group_of[i] = gidx;
            if (deg(i) < (int)gb.upat.size()) gb.masked = true;
            joined = true;
            break;
          }
      }
    }
    clk.lap("  groups: leftovers");
    std::vector<int32_t> neworder;
    neworder.reserve(M0);
    pl.grp_upat.clear();
    for (GroupBuild &gb : groups) {
      std::sort(gb.members.begin(), gb.members.end());  // locality order inside a group
      Plan::GrpRange gr;
      gr.l0 = (int32_t)neworder.size();
      gr.nl = (int32_t)gb.members.size();
      gr.d = gb.d;
      gr.no = (int32_t)gb.upat.size();
      gr.masked = gb.masked ? 1 : 0;
      gr.upat0 = (int64_t)pl.grp_upat.size();
      pl.grp_upat.insert(pl.grp_upat.end(), gb.upat.begin(), gb.upat.end());
      pl.grp_range.push_back(gr);
      for (int32_t i : gb.members) neworder.push_back(pl.pt_user_of_int[i]);
    }
    pl.M_grp = (int)neworder.size();
    for (int i = 0; i < M0; ++i)
      if (group_of[i] < 0) neworder.push_back(pl.pt_user_of_int[i]);
    pl.pt_user_of_int.swap(neworder);
    for (int k = 0; k < M0; ++k) pl.pt_int_of_user[pl.pt_user_of_int[k]] = k;
  }
  if (pl.M_grp == 0) pl.lin_groups = false;
  clk.lap("covisibility groups");
  // Interleave inside windows of one Schur super-run: position p of a window
  // takes the landmark p would have had in residue-class order (k = r, r + S,
  // r + 2S, ...), so that every chunk of consecutive landmarks samples the
  // WHOLE window and touches the run's S blocks in the run's proportions
  // (the lanes of the Schur kernel are dealt to the blocks in those proportions).
  if (!(getenv("BA_NO_INTERLEAVE") && getenv("BA_NO_INTERLEAVE")[0] == '1')) {
    const int W = schur_run_cap(pl.M - pl.M_grp);
    std::vector<int32_t> tmp;
    for (int base = pl.M_grp; base < pl.M; base += W) {
      const int n = std::min(W, pl.M - base);
      tmp.clear();
      for (int r = 0; r < kSchurInterleave; ++r)
        for (int k = r; k < n; k += kSchurInterleave) tmp.push_back(pl.pt_user_of_int[base + k]);
      for (int k = 0; k < n; ++k) pl.pt_user_of_int[base + k] = tmp[k];
    }
    for (int k = 0; k < pl.M; ++k) pl.pt_int_of_user[pl.pt_user_of_int[k]] = k;
  }
  for (int idx = 0; idx < in.n_pt; ++idx) {
    const int q = order[idx];
    if (pl.owner[q] == in.rank && in.pt_fixed[q]) {
      pl.pt_int_of_user[q] = (int32_t)pl.pt_user_of_int.size();
      pl.pt_user_of_int.push_back(q);
    }
  }
  pl.n_pt = (int)pl.pt_user_of_int.size();
  const int M = pl.M;

  clk.lap("interleave + fixed points");
  // ---- landmark-major observation list ----
  // = the per-point lists in the final landmark order (key: landmark, pose, insertion).
  // Per landmark: its observations and its (landmark, pose) pairs — one per distinct
  // optimisable pose, carried by the LAST inserted observation of that pose (reference
  // :826: B_ji is assigned, not accumulated).  Counts first, prefix sums, then every
  // landmark fills its own ranges: parallel over landmarks.
  {
    const int n_own = pl.n_pt;
    std::vector<int64_t> optr((size_t)n_own + 1, 0), pptr((size_t)n_own + 1, 0);
    // group of every grouped landmark (masked groups pad their members to the union)
    std::vector<int32_t> lm_grp((size_t)pl.M_grp, -1);
    for (size_t gidx = 0; gidx < pl.grp_range.size(); ++gidx)
      for (int l = pl.grp_range[gidx].l0; l < pl.grp_range[gidx].l0 + pl.grp_range[gidx].nl; ++l) lm_grp[l] = (int32_t)gidx;
    auto masked_group = [&](int64_t pi) -> const Plan::GrpRange * {
      if (pi >= pl.M_grp) return nullptr;
      const Plan::GrpRange &gr = pl.grp_range[lm_grp[pi]];
      return gr.masked ? &gr : nullptr;
    };
    parallel_for(n_own, [&](int64_t a0, int64_t a1) {
      for (int64_t pi = a0; pi < a1; ++pi) {
        const int q = pl.pt_user_of_int[pi];
        if (const Plan::GrpRange *gr = masked_group(pi)) {
          optr[pi + 1] = gr->no;
          pptr[pi + 1] = gr->d;
          continue;
        }
        optr[pi + 1] = uo_ptr[q + 1] - uo_ptr[q];
        if (pi < M) {
          int64_t np = 0;
          int32_t last = -1;
          for (int64_t t = uo_ptr[q]; t < uo_ptr[q + 1]; ++t) {
            const int32_t ji = uo_rec[t].pose;
            if (ji < N && ji != last) ++np;
            last = ji;
          }
          pptr[pi + 1] = np;
        }
      }
    });
    for (int pi = 0; pi < n_own; ++pi) {
      optr[pi + 1] += optr[pi];
      pptr[pi + 1] += pptr[pi];
    }
    pl.n_obs = optr[n_own];
    if (pptr[n_own] >= (int64_t)INT32_MAX) return "too many pairs for int32 pair ids";
    pl.obs_idx.resize((size_t)pl.n_obs * 4);
    pl.obs_uv.resize((size_t)pl.n_obs * 2);
    const bool slim = pl.n_cam < 65536 && pl.n_pose < 65536;
    pl.obs_cp.clear();
    if (slim) pl.obs_cp.resize((size_t)pl.n_obs * 2);
    pl.lm_obs_ptr.assign(M + 1, 0);
    pl.lm_pair_ptr.assign(M + 1, 0);
    pl.pair_pose.resize((size_t)pptr[n_own]);
    pl.pair_lm.resize((size_t)pptr[n_own]);
    pl.pair_pad.assign((size_t)pptr[n_own], 0);
    parallel_for(n_own, [&](int64_t a0, int64_t a1) {
      for (int64_t pi = a0; pi < a1; ++pi) {
        const int q = pl.pt_user_of_int[pi];
        if (const Plan::GrpRange *gr = masked_group(pi)) {
          // the union's slots in order; this landmark's observations are a subsequence
          int64_t s = optr[pi], pw = pptr[pi] - 1, t = uo_ptr[q];
          int32_t last = -1;
          bool pose_seen = false;
          for (int e = 0; e < gr->no; ++e, ++s) {
            const uint64_t w = pl.grp_upat[gr->upat0 + e];
            const int32_t ji = (int32_t)(w >> 32), cam = (int32_t)(uint32_t)w;
            const bool have = t < uo_ptr[q + 1] && uo_rec[t].pose == ji && uo_rec[t].cam == cam;
            pl.obs_idx[4 * s + 0] = cam;
            pl.obs_idx[4 * s + 1] = ji;
            pl.obs_idx[4 * s + 2] = (int32_t)pi;
            pl.obs_idx[4 * s + 3] = -1;
            if (slim) {
              pl.obs_cp[2 * s + 0] = cam | (ji << 16);
              pl.obs_cp[2 * s + 1] = (int32_t)pi;
            }
            pl.obs_uv[2 * s + 0] = have ? uo_rec[t].u : std::numeric_limits<double>::quiet_NaN();
            pl.obs_uv[2 * s + 1] = have ? uo_rec[t].v : std::numeric_limits<double>::quiet_NaN();
            if (ji < N) {
              if (ji != last) {
                ++pw;
                pl.pair_pose[pw] = ji;
                pl.pair_lm[pw] = (int32_t)pi;
                pose_seen = false;
              } else {
                pl.obs_idx[4 * (s - 1) + 3] = -1;  // (the pair id sits on the pose's LAST slot)
              }
              pl.obs_idx[4 * s + 3] = (int32_t)pw;
              pose_seen = pose_seen || have;
              pl.pair_pad[pw] = pose_seen ? 0 : 1;
            }
            last = ji < N ? ji : -1;
            if (have) ++t;
          }
          continue;
        }
        int64_t s = optr[pi], pw = pptr[pi] - 1;
        int32_t last = -1;
        for (int64_t t = uo_ptr[q]; t < uo_ptr[q + 1]; ++t, ++s) {
          const ObsRecP &r = uo_rec[t];
          const int32_t ji = r.pose;
          pl.obs_idx[4 * s + 0] = r.cam;
          pl.obs_idx[4 * s + 1] = ji;
          pl.obs_idx[4 * s + 2] = (int32_t)pi;
          pl.obs_idx[4 * s + 3] = -1;
          if (slim) {
            pl.obs_cp[2 * s + 0] = r.cam | (ji << 16);
            pl.obs_cp[2 * s + 1] = (int32_t)pi;
          }
          pl.obs_uv[2 * s + 0] = r.u;
          pl.obs_uv[2 * s + 1] = r.v;
          if (pi < M && ji < N) {
            if (ji != last) {
              ++pw;
              pl.pair_pose[pw] = ji;
              pl.pair_lm[pw] = (int32_t)pi;
            } else {
              // same pair as the previous (adjacent) observation: the earlier
              // insertion loses its B_ji to this one (reference :826)
              pl.obs_idx[4 * (s - 1) + 3] = -1;
            }
            pl.obs_idx[4 * s + 3] = (int32_t)pw;
          }
          last = (pi < M && ji < N) ? ji : -1;
        }
      }
    });
    for (int i = 0; i < M; ++i) {
      pl.lm_obs_ptr[i + 1] = optr[i + 1] - optr[i];
      pl.lm_pair_ptr[i + 1] = pptr[i + 1] - pptr[i];
    }
    pl.n_pair_pad = 0;
    for (uint8_t v : pl.pair_pad) pl.n_pair_pad += v;
  }
  for (int i = 0; i < M; ++i) {
    pl.lm_obs_ptr[i + 1] += pl.lm_obs_ptr[i];
    pl.lm_pair_ptr[i + 1] += pl.lm_pair_ptr[i];
  }
  pl.n_obs_opt = pl.lm_obs_ptr[M];
  pl.P = (int64_t)pl.pair_pose.size();

  // (the per-point lists are not needed any more: their 140 MB go back to the system on a
  //  side thread while the remaining phases run)
  struct Reaper {
    std::thread t;
    ~Reaper() {
      if (t.joinable()) t.join();
    }
  } reaper;
  reaper.t = std::thread([a = std::move(uo_rec), b = std::move(uo_ptr)]() mutable {
    a = decltype(a)();
    b = decltype(b)();
  });
  clk.lap("landmark-major list");
  // ---- pose-major observation list (optimisable poses) ----
  {
    std::vector<int64_t> psel;
    // (the landmark-major list is in landmark order: the grouped landmarks' observations,
    //  which never enter, are its first lm_obs_ptr[M_grp] entries)
    const int64_t s_first = pl.lin_groups ? pl.lm_obs_ptr[pl.M_grp] : 0;
    psel.reserve((size_t)(pl.n_obs - s_first));
    // (observations of landmarks in covisibility groups are linearised, pose side
    //  included, by k_lin_grp: they do not enter the pose-major list)
    for (int64_t s = s_first; s < pl.n_obs; ++s)
      if (pl.obs_idx[4 * s + 1] < N && !(pl.lin_groups && pl.obs_idx[4 * s + 2] < pl.M_grp)) psel.push_back(s);
    // stable counting sort by pose keeps (point, insertion) order inside
    pl.pose_obs_ptr.assign(N + 1, 0);
    for (int64_t s : psel) pl.pose_obs_ptr[pl.obs_idx[4 * s + 1] + 1]++;
    for (int j = 0; j < N; ++j) pl.pose_obs_ptr[j + 1] += pl.pose_obs_ptr[j];
    pl.n_pobs = (int64_t)psel.size();
    pl.pobs_idx.resize((size_t)pl.n_pobs * 2);
    pl.pobs_uv.resize((size_t)pl.n_pobs * 2);
    std::vector<int64_t> cur(pl.pose_obs_ptr.begin(), pl.pose_obs_ptr.end() - 1);
    for (int64_t s : psel) {
      const int j = pl.obs_idx[4 * s + 1];
      const int64_t d = cur[j]++;
      pl.pobs_idx[2 * d + 0] = pl.obs_idx[4 * s + 0];  // camera
      pl.pobs_idx[2 * d + 1] = pl.obs_idx[4 * s + 2];  // point (the pose is the list's own)
      pl.pobs_uv[2 * d + 0] = pl.obs_uv[2 * s + 0];
      pl.pobs_uv[2 * d + 1] = pl.obs_uv[2 * s + 1];
    }
    // one wave per chunk: aim at ~kPoseWaveTarget waves in total (one resident
    // round on the GPU), each pose's list split into equal chunks
    int64_t cap = (pl.n_pobs + kPoseWaveTarget - 1) / kPoseWaveTarget;
    cap = (cap + 63) / 64 * 64;
    cap = std::min<int64_t>(std::max<int64_t>(cap, kPoseChunkMin), kPoseChunkMax);
    make_chunks(pl.pose_obs_ptr, N, (int)cap, pl.achunk_pose,
                pl.achunk_begin, pl.achunk_end, pl.pose_achunk_ptr);
  }

  clk.lap("pose-major list");
  // ---- Schur complement structure ----
  // Non-zero upper blocks (j <= k) of S = union over landmarks of J(i) x J(i),
  // plus every diagonal.  Landmarks are processed LANDMARK-MAJOR by "Schur
  // workgroups": a workgroup stages the W blocks of a run of consecutive
  // landmarks (<= kSchurPairs pairs) in LDS and accumulates their
  // contributions into workgroup-local block SLOTS; a second kernel sums the
  // slots of each block in workgroup order (deterministic).  Landmarks seen by
  // more than kSchurPairs poses go through the global (j,k)-sorted triple list.
  {
    // block set as a CSR over pose rows.  It is the GLOBAL set (all shards use
    // the same block numbering: the packed S||rhs exchange buffer is indexed
    // by block), so with world > 1 it is built from the full observation list.
    std::vector<std::vector<int32_t>> rowk(N);
    for (int j = 0; j < N; ++j) rowk[j].push_back(j);
    if (in.world == 1) {
      // (the landmarks of a covisibility group share one pose set: its first landmark
      //  stands for all of them)
      auto add_landmark = [&](int i) {
        for (int64_t p = pl.lm_pair_ptr[i]; p < pl.lm_pair_ptr[i + 1]; ++p) {
          auto &rk = rowk[pl.pair_pose[p]];
          for (int64_t q = p; q < pl.lm_pair_ptr[i + 1]; ++q) rk.push_back(pl.pair_pose[q]);
        }
      };
      for (const Plan::GrpRange &gr : pl.grp_range) add_landmark(gr.l0);
      for (int i = pl.M_grp; i < M; ++i)
        for (int64_t p = pl.lm_pair_ptr[i]; p < pl.lm_pair_ptr[i + 1]; ++p) {
          auto &rk = rowk[pl.pair_pose[p]];
          for (int64_t q = p; q < pl.lm_pair_ptr[i + 1]; ++q) {
            const int32_t k = pl.pair_pose[q];
            if (rk.empty() || rk.back() != k) rk.push_back(k);
          }
        }
    } else {
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
        for (size_t u = a; u < b; ++u) {
          auto &rk = rowk[(int)(uint32_t)keys[u]];
          for (size_t v = u; v < b; ++v) {
            const int32_t k = (int32_t)(uint32_t)keys[v];
            if (rk.empty() || rk.back() != k) rk.push_back(k);
          }
        }
        a = b;
      }
    }
    std::vector<int64_t> blk_row_ptr(N + 1, 0);
    pl.sblk_j.clear();
    pl.sblk_k.clear();
    for (int j = 0; j < N; ++j) {
      auto &rk = rowk[j];
      std::sort(rk.begin(), rk.end());
      rk.erase(std::unique(rk.begin(), rk.end()), rk.end());
      for (int32_t k : rk) {
        pl.sblk_j.push_back(j);
        pl.sblk_k.push_back(k);
      }
      blk_row_ptr[j + 1] = (int64_t)pl.sblk_j.size();
    }
    pl.B = (int64_t)pl.sblk_j.size();
    pl.diag_blk.resize(N);
    for (int j = 0; j < N; ++j) pl.diag_blk[j] = (int32_t)blk_row_ptr[j];  // (j,j) leads row j
    auto block_of = [&](int32_t j, int32_t k) -> int32_t {
      const int32_t *b0 = &pl.sblk_k[blk_row_ptr[j]];
      const int32_t *e0 = &pl.sblk_k[blk_row_ptr[j + 1]];
      return (int32_t)(std::lower_bound(b0, e0, k) - &pl.sblk_k[0]);
    };

    // does not fit a super-run whole ...
    auto is_big = [&](int64_t d) {
      return d * (d + 1) / 2 > kSchurTri || d * (d + 1) / 2 > kSchurSlots ||
             d > kSchurPairs;
    };
    // ... and cannot be split into classes either (its pairs exceed the LDS staging): global list
    auto is_list = [&](int64_t d) {
      static const bool no_split = getenv("BA_NO_SPLIT") && getenv("BA_NO_SPLIT")[0] == '1';
      return is_big(d) && (d > kSchurPairs || no_split);
    };
    // (a) big landmarks -> global triple list sorted by (block, landmark)
    int64_t Tbig = 0, Tall = 0;
    for (int i = 0; i < M; ++i) {
      const int64_t d = pl.lm_pair_ptr[i + 1] - pl.lm_pair_ptr[i];
      Tall += d * (d + 1) / 2;
      if (i >= pl.M_grp && is_list(d)) Tbig += d * (d + 1) / 2;  // (grouped landmarks: k_schur_grp's)
    }
    pl.T = Tall;
    {
      std::vector<std::pair<int32_t, std::pair<int64_t, int64_t>>> big;
      big.reserve(Tbig);
      for (int i = 0; i < M; ++i) {
        const int64_t p0 = pl.lm_pair_ptr[i], p1 = pl.lm_pair_ptr[i + 1];
        if (i < pl.M_grp || !is_list(p1 - p0)) continue;
        for (int64_t p = p0; p < p1; ++p)
          for (int64_t q = p; q < p1; ++q)
            big.push_back({block_of(pl.pair_pose[p], pl.pair_pose[q]), {p, q}});
      }
      std::stable_sort(big.begin(), big.end(),
                       [](const auto &x, const auto &y) { return x.first < y.first; });
      pl.tri_p.resize(big.size());
      pl.tri_q.resize(big.size());
      pl.sblk_tri_ptr.assign(pl.B + 1, 0);
      for (size_t t = 0; t < big.size(); ++t) {
        pl.tri_p[t] = big[t].second.first;
        pl.tri_q[t] = big[t].second.second;
        pl.sblk_tri_ptr[big[t].first + 1]++;
      }
      for (int64_t bk = 0; bk < pl.B; ++bk)
        pl.sblk_tri_ptr[bk + 1] += pl.sblk_tri_ptr[bk];
      make_chunks(pl.sblk_tri_ptr, (int)pl.B, kTriChunk, pl.tchunk_blk,
                  pl.tchunk_begin, pl.tchunk_end, pl.sblk_tchunk_ptr);
    }

    // (b) SUPER-RUNS over the remaining landmarks: maximal runs of consecutive
    // landmarks (locality order) that touch at most kSchurSlots distinct
    // blocks; one workgroup per super-run keeps one 6x6 accumulator per
    // (slot, lane) in registers while it streams the run's W blocks through
    // LDS in CHUNKS (<= kSchurPairs pairs, <= kSchurTri triples).
    pl.sup_desc.clear();
    pl.chunk_desc.clear();
    pl.slot_blk.clear();
    pl.ltri.clear();
    pl.ltri.reserve((size_t)(Tall - Tbig));
    pl.chunk_sp.clear();
    std::vector<std::pair<int32_t, int32_t>> contrib;  // (block, flat slot)
    std::vector<int32_t> mark(pl.B, -1);               // block -> local slot
    std::vector<int32_t> sup_blocks;
    std::vector<std::pair<int32_t, uint32_t>> loc;
    std::vector<int64_t> tcount;  // triples per slot of the current run
    double sum_max = 0, sum_ideal = 0, sum_mean_active = 0;  // BA_PLAN_STATS
    long n_ch = 0;
    pl.sup_lane.clear();
    // (a') covisibility groups: one k_schur_grp workgroup per (piece of a) group,
    // d (d + 1) / 2 slots each, in (jj, kk) row-major order of the upper triangle
    pl.grp32.clear();
    pl.grp64.clear();
    pl.grp128.clear();
    pl.grp_pat.clear();
    pl.lin_desc.clear();
    pl.n_apart2 = 0;
    for (const Plan::GrpRange &gr : pl.grp_range) {
      const int64_t p0 = pl.lm_pair_ptr[gr.l0];
      for (int l = gr.l0; l < gr.l0 + gr.nl; ++l)
        if (pl.lm_pair_ptr[l + 1] - pl.lm_pair_ptr[l] != gr.d ||
            pl.lm_obs_ptr[l + 1] - pl.lm_obs_ptr[l] != gr.no)
          return "internal: covisibility group with a ragged landmark";
      // observation pattern of the group, from its first landmark
      const int32_t pat0 = (int32_t)(pl.grp_pat.size() / 2);
      {
        const int64_t ob = pl.lm_obs_ptr[gr.l0];
        int jj = -1;
        int32_t lastp = -1;
        for (int oo = 0; oo < gr.no; ++oo) {
          const int32_t ji = pl.obs_idx[4 * (ob + oo) + 1], cam = pl.obs_idx[4 * (ob + oo) + 0];
          const bool opt = ji < N;
          if (opt && ji != lastp) ++jj;
          lastp = ji;
          const bool lastw = pl.obs_idx[4 * (ob + oo) + 3] >= 0;
          pl.grp_pat.push_back(ji);
          pl.grp_pat.push_back(cam | ((opt ? jj : 0) << 16) | (opt ? 1 << 29 : 0) | (lastw ? 1 << 30 : 0));
        }
        if (jj + 1 != gr.d) return "internal: covisibility group pattern / pose count mismatch";
      }
      // k_lin_grp pieces: runs of `steps` wave steps (4 waves x nlw landmarks each), the
      // pieces of a group equal up to one step
      {
        const int steps_env = getenv("BA_LIN_STEPS") ? std::max(1, atoi(getenv("BA_LIN_STEPS"))) : kLinGrpSteps;
        const int per_step = 4 * lin_grp_nlw(gr.no);
        const int nstep = (gr.nl + per_step - 1) / per_step;
        const int npiece = (nstep + steps_env - 1) / steps_env;
        for (int c = 0; c < npiece; ++c) {
          const int s0 = (int)((int64_t)nstep * c / npiece), s1 = (int)((int64_t)nstep * (c + 1) / npiece);
          Plan::LinDesc ld;
          const int il0 = s0 * per_step, il1 = std::min(gr.nl, s1 * per_step);
          ld.l0 = gr.l0 + il0;
          ld.nl = il1 - il0;
          ld.d = gr.d;
          ld.no = gr.no;
          ld.p0 = p0 + (int64_t)gr.d * il0;
          ld.o0 = pl.lm_obs_ptr[ld.l0];
          ld.pat0 = pat0;
          ld.apart0 = (int32_t)pl.n_apart2;
          ld.cost_idx = 0;  // set below, after the chunks are known
          ld.pad_ = gr.masked;  // 1: padded slots (uv = NaN) and a dynamic last writer
          pl.n_apart2 += gr.d;
          if (ld.nl > 0) pl.lin_desc.push_back(ld);
        }
      }
      if (gr.d == 0) continue;  // no pair, no Schur contribution: no k_schur_grp workgroup
      static const int grp_max = getenv("BA_GRP_MAX") ? std::max(12, atoi(getenv("BA_GRP_MAX"))) : kGrpMaxLandmarks;
      // (wide groups, d > 10: k_schur_grp_wide is bound by the fp64 matrix pipe — 27 MFMAs per
      //  landmark — and a 20-pose window of a few hundred landmarks is one long workgroup:
      //  pieces of <= kGrpWidePiece landmarks spread them over the CUs)
      const int cap = gr.d > 10 ? std::min(grp_max, kGrpWidePiece) : grp_max;
      const int pieces = (gr.nl + cap - 1) / cap;
      const int per = (gr.nl + pieces - 1) / pieces;
      for (int c0 = 0; c0 < gr.nl; c0 += per) {
        Plan::GrpDesc gd = Plan::GrpDesc();
        gd.l0 = gr.l0 + c0;
        gd.nl = std::min(per, gr.nl - c0);
        gd.d = gr.d;
        gd.p0 = p0 + (int64_t)gr.d * c0;
        gd.s0 = (int32_t)pl.slot_blk.size();
        for (int t = 0; t < kGrpMaxPoses; ++t) gd.pose[t] = t < gr.d ? pl.pair_pose[p0 + t] : 0;
        for (int jj = 0; jj < gr.d; ++jj)
          for (int kk = jj; kk < gr.d; ++kk) {
            const int32_t bk = block_of(gd.pose[jj], gd.pose[kk]);
            contrib.push_back({bk, (int32_t)pl.slot_blk.size()});
            pl.slot_blk.push_back(bk);
          }
        (gr.d <= 5 ? pl.grp32 : gr.d <= 10 ? pl.grp64 : pl.grp128).push_back(gd);
      }
    }
    const int sup_cap = schur_run_cap(M - pl.M_grp);
    // Landmark kinds: 0 = fits a super-run whole; 1 = SPLIT: its pairs fit the LDS
    // staging (<= kSchurPairs) but its d (d + 1) / 2 blocks exceed the kSchurSlots
    // register slots of a run: the pose list is cut into g groups of <= kSplit
    // poses and the landmark is processed once per CLASS (a, b), a <= b < g — the
    // triples (p, q) with p in group a and q in group b: at most kSplit^2 <= 128
    // blocks — in a separate run; 2 = big (more pairs than the staging holds):
    // global triple list.
    constexpr int kSplit = 11;
    static_assert(kSplit * kSplit <= kSchurSlots && kSplit * kSplit <= kSchurTri, "class size");
    auto kind_of = [&](int l) {
      const int64_t d = pl.lm_pair_ptr[l + 1] - pl.lm_pair_ptr[l];
      if (!is_big(d)) return 0;
      return is_list(d) ? 2 : 1;
    };
    auto ngrp = [&](int l) { return (int)((pl.lm_pair_ptr[l + 1] - pl.lm_pair_ptr[l] + kSplit - 1) / kSplit); };
    // triples of landmark l in class (ca, cb) of split size sp (sp == 0: all)
    auto class_triples = [&](int l, int sp, int ca, int cb) -> int64_t {
      const int64_t d = pl.lm_pair_ptr[l + 1] - pl.lm_pair_ptr[l];
      if (sp == 0) return d * (d + 1) / 2;
      const int64_t na = std::min<int64_t>(sp, d - (int64_t)ca * sp), nb = std::min<int64_t>(sp, d - (int64_t)cb * sp);
      if (na <= 0 || nb <= 0) return 0;
      return ca == cb ? na * (na + 1) / 2 : na * nb;
    };
    auto in_class = [&](int64_t lp, int64_t lq, int sp, int ca, int cb) {
      return sp == 0 || (lp / sp == ca && lq / sp == cb);
    };
    std::string run_error;
    // super-runs over the landmarks [lo, hi) (all of one kind), restricted to one class
    auto build_runs = [&](const int lo, const int hi, const int sp, const int ca, const int cb) {
    int i = lo;
    while (i < hi && run_error.empty()) {
      // ---- grow a super-run ----
      const int i0 = i;
      sup_blocks.clear();
      // chunk count of the run so far (same greedy rule as the chunk loop below)
      int nchunk = 1, cnl = 0;
      int64_t cnp = 0, cnt = 0;
      while (i < hi && i - i0 < sup_cap) {
        const int64_t p0 = pl.lm_pair_ptr[i], p1 = pl.lm_pair_ptr[i + 1];
        {
          const int64_t dd = p1 - p0, dt = class_triples(i, sp, ca, cb);
          if (cnl > 0 && (cnp + dd > kSchurPairs || cnt + dt > kSchurTri ||
                          cnl >= kSchurLandmarks)) {
            if (nchunk == kSchurSuperChunks) break;  // descriptor table in LDS is full
            ++nchunk;
            cnp = cnt = 0;
            cnl = 0;
          }
          cnp += dd;
          cnt += dt;
          ++cnl;
        }
        // blocks this landmark would add
        const size_t before = sup_blocks.size();
        for (int64_t p = p0; p < p1; ++p)
          for (int64_t q = p; q < p1; ++q) {
            if (!in_class(p - p0, q - p0, sp, ca, cb)) continue;
            const int32_t bk = block_of(pl.pair_pose[p], pl.pair_pose[q]);
            if (mark[bk] < 0) {
              mark[bk] = (int32_t)sup_blocks.size();
              sup_blocks.push_back(bk);
            }
          }
        if ((int)sup_blocks.size() > kSchurSlots && i > i0) {
          for (size_t s2 = before; s2 < sup_blocks.size(); ++s2) mark[sup_blocks[s2]] = -1;
          sup_blocks.resize(before);
          break;
        }
        ++i;
      }
      const int i1 = i;
      const int ns = (int)sup_blocks.size();
      if (ns == 0) {  // landmarks without pairs
        continue;
      }
      if (ns > kSchurSlots) {
        for (int32_t bk : sup_blocks) mark[bk] = -1;
        run_error = "internal: landmark class exceeds kSchurSlots blocks";
        return;
      }
      // slots sorted by block id for a stable order
      std::vector<int32_t> order(sup_blocks);
      std::sort(order.begin(), order.end());
      for (int s2 = 0; s2 < ns; ++s2) mark[order[s2]] = s2;
      Plan::SupDesc sd;
      sd.s0 = (int32_t)pl.slot_blk.size();
      sd.ns = ns;
      sd.chunk_begin = (int32_t)pl.chunk_desc.size();
      for (int s2 = 0; s2 < ns; ++s2) {
        contrib.push_back({order[s2], sd.s0 + s2});
        pl.slot_blk.push_back(order[s2]);
      }
      // ---- chunks ----
      tcount.assign(ns, 0);
      int l = i0;
      while (l < i1) {
        const int c0 = l;
        const int64_t pbase = pl.lm_pair_ptr[c0];
        int64_t np = 0, nt = 0;
        while (l < i1) {
          const int64_t d = pl.lm_pair_ptr[l + 1] - pl.lm_pair_ptr[l];
          const int64_t dt = class_triples(l, sp, ca, cb);
          if (l > c0 && (np + d > kSchurPairs || nt + dt > kSchurTri ||
                         l - c0 >= kSchurLandmarks))
            break;
          np += d;
          nt += dt;
          ++l;
        }
        loc.clear();
        for (int m = c0; m < l; ++m) {
          const int64_t q0 = pl.lm_pair_ptr[m];
          for (int64_t p = q0; p < pl.lm_pair_ptr[m + 1]; ++p)
            for (int64_t q = p; q < pl.lm_pair_ptr[m + 1]; ++q)
              if (in_class(p - q0, q - q0, sp, ca, cb))
                loc.push_back({mark[block_of(pl.pair_pose[p], pl.pair_pose[q])],
                               (uint32_t)(((p - pbase) << 16) | ((uint32_t)(m - c0) << 8) |
                                          (q - pbase))});
        }
        if (loc.empty()) continue;  // (landmarks without pairs)
        std::stable_sort(loc.begin(), loc.end(),
                         [](const auto &x, const auto &y) { return x.first < y.first; });
        Plan::ChunkDesc cd;
        cd.p0 = pbase;
        while (pl.ltri.size() & 3) pl.ltri.push_back(0u);  // chunks start 16-byte aligned
        cd.tb = (int64_t)pl.ltri.size();
        cd.sp = (int64_t)pl.chunk_sp.size();
        cd.l0 = c0;
        cd.nl = l - c0;
        cd.np = (int32_t)np;
        cd.nt = (int32_t)loc.size();
        // per-slot offsets into this chunk's triple list
        size_t t = 0;
        for (int s2 = 0; s2 <= ns; ++s2) {
          while (t < loc.size() && loc[t].first < s2) ++t;
          pl.chunk_sp.push_back((uint16_t)t);
        }
        for (auto &e : loc) {
          pl.ltri.push_back(e.second);
          tcount[e.first]++;
        }
        pl.chunk_desc.push_back(cd);
      }
      sd.chunk_end = (int32_t)pl.chunk_desc.size();
      if (sd.chunk_end - sd.chunk_begin > kSchurSuperChunks) {
        run_error = "internal: super-run exceeds kSchurSuperChunks chunks";
        return;
      }
      if (sd.chunk_end == sd.chunk_begin) {  // nothing to do after all: drop the run's slots
        for (int s2 = 0; s2 < ns; ++s2) {
          contrib.pop_back();
          pl.slot_blk.pop_back();
        }
        for (int32_t bk : sup_blocks) mark[bk] = -1;
        continue;
      }
      pl.sup_desc.push_back(sd);
      deal_lanes(tcount, pl.sup_lane);
      if (getenv("BA_PLAN_STATS")) {  // balance of the triple loop (developer knob)
        const uint32_t *lw = &pl.sup_lane[pl.sup_lane.size() - 256];
        std::vector<int> tps2(ns, 1);
        for (int q = 0; q < 256; ++q)
          if ((lw[q] & 0xffu) < (uint32_t)ns) tps2[lw[q] & 0xffu] = (lw[q] >> 14) & 0x3f;
        for (int c = sd.chunk_begin; c < sd.chunk_end; ++c) {
          const uint16_t *sp2 = &pl.chunk_sp[pl.chunk_desc[c].sp];
          int mx = 0, tot = 0, act = 0;
          for (int q = 0; q < ns; ++q) {
            const int cnt2 = sp2[q + 1] - sp2[q];
            tot += cnt2;
            act += cnt2 > 0;
            mx = std::max(mx, (cnt2 + tps2[q] - 1) / tps2[q]);
          }
          sum_max += mx;
          sum_ideal += tot * 2.0 / 256.0;
          sum_mean_active += act;
          ++n_ch;
        }
      }
      for (int32_t bk : sup_blocks) mark[bk] = -1;
    }
    };
    {
      int i = pl.M_grp;
      while (i < M && run_error.empty()) {
        const int kd = kind_of(i);
        if (kd == 2) {  // handled by the global triple list
          ++i;
          continue;
        }
        int j = i + 1;
        while (j < M && kind_of(j) == kd && (kd == 0 || ngrp(j) == ngrp(i))) ++j;
        if (kd == 0) {
          build_runs(i, j, 0, 0, 0);
        } else {
          const int g = ngrp(i);
          for (int ca = 0; ca < g; ++ca)
            for (int cb = ca; cb < g; ++cb) build_runs(i, j, kSplit, ca, cb);
        }
        i = j;
      }
      if (!run_error.empty()) return run_error;
    }
    if (n_ch > 0)
      fprintf(stderr, "[plan] chunks %ld: triple-loop iterations per chunk: busiest lane %.2f, ideal %.2f; "
                      "active slots %.1f\n",
              n_ch, sum_max / n_ch, sum_ideal / n_ch, sum_mean_active / n_ch);
    for (int k = 0; k < 4; ++k) pl.ltri.push_back(0u);  // the last chunk's 16-byte loads stay inside
    // per-block contribution lists (ascending super-run = ascending slot id)
    pl.blk_contrib_ptr.assign(pl.B + 1, 0);
    for (auto &c2 : contrib) pl.blk_contrib_ptr[c2.first + 1]++;
    for (int64_t bk = 0; bk < pl.B; ++bk)
      pl.blk_contrib_ptr[bk + 1] += pl.blk_contrib_ptr[bk];
    pl.contrib_slot.resize(contrib.size());
    {
      std::vector<int64_t> cur(pl.blk_contrib_ptr.begin(), pl.blk_contrib_ptr.end() - 1);
      for (auto &c2 : contrib) pl.contrib_slot[cur[c2.first]++] = c2.second;
    }
  }

  clk.lap("Schur structure");
  // ---- back-substitution chunks: consecutive landmarks, <= kSchurPairs pairs
  // (a landmark with more pairs forms a chunk of its own) ----
  {
    // (a chunk does not straddle M_grp: with lin_groups k_lin_landmarks starts
    //  at chunk n_bchunk_grp, the grouped landmarks are k_lin_grp's)
    pl.bchunk_lm.assign(1, 0);
    pl.n_bchunk_grp = 0;
    int l = 0;
    while (l < M) {
      int64_t np = 0;
      const int c0 = l;
      const int lim = l < pl.M_grp ? pl.M_grp : M;
      while (l < lim && l - c0 < kSchurLandmarks) {
        const int64_t dd = pl.lm_pair_ptr[l + 1] - pl.lm_pair_ptr[l];
        if (l > c0 && np + dd > kSchurPairs) break;
        np += dd;
        ++l;
      }
      pl.bchunk_lm.push_back(l);
      if (l <= pl.M_grp) pl.n_bchunk_grp = (int)pl.bchunk_lm.size() - 1;
    }
  }
  // ---- k_lin_grp bookkeeping: cost-partial entry of each group piece (after the
  // chunks' entries), rows of Apart2 per pose ----
  {
    // plain pieces first, masked pieces behind them: one launch of each k_lin_grp form
    std::stable_partition(pl.lin_desc.begin(), pl.lin_desc.end(), [](const Plan::LinDesc &g) { return g.pad_ == 0; });
    pl.n_lin_plain = 0;
    for (auto &g : pl.lin_desc) pl.n_lin_plain += g.pad_ == 0;
    int32_t ci = (int32_t)pl.bchunk_lm.size() - 1;
    for (auto &g : pl.lin_desc) g.cost_idx = ci++;
    pl.pose_gpart_ptr.assign(N + 1, 0);
    if (!pl.lin_groups) {
      pl.n_apart2 = 0;
      pl.lin_desc.clear();
      pl.n_lin_plain = 0;
    }
    for (auto &g : pl.lin_desc)
      for (int t = 0; t < g.d; ++t) pl.pose_gpart_ptr[pl.pair_pose[g.p0 + t] + 1]++;
    for (int j = 0; j < N; ++j) pl.pose_gpart_ptr[j + 1] += pl.pose_gpart_ptr[j];
    pl.pose_gpart.assign((size_t)pl.pose_gpart_ptr[N], 0);
    std::vector<int32_t> cur(pl.pose_gpart_ptr.begin(), pl.pose_gpart_ptr.end() - 1);
    for (auto &g : pl.lin_desc)
      for (int t = 0; t < g.d; ++t) pl.pose_gpart[cur[pl.pair_pose[g.p0 + t]]++] = g.apart0 + t;
  }
  clk.lap("chunks + bookkeeping");

  return std::string();
}

// Tile pattern of S for tiles of `poses_per_tile` consecutive optimised poses
// (global: every shard factors the same matrix).
void tile_pattern(const Plan &pl, int poses_per_tile, int &ncb,
                  std::vector<uint8_t> &adj) {
  ncb = std::max(1, (pl.N + poses_per_tile - 1) / poses_per_tile);
  adj.assign((size_t)ncb * ncb, 0);
  for (int t = 0; t < ncb; ++t) adj[(size_t)t * ncb + t] = 1;
  for (int64_t bk = 0; bk < pl.B; ++bk) {
    const int a = pl.sblk_j[bk] / poses_per_tile, b = pl.sblk_k[bk] / poses_per_tile;
    adj[(size_t)a * ncb + b] = 1;
    adj[(size_t)b * ncb + a] = 1;
  }
}

}  // namespace ba
