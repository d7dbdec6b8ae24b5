// ba_api.hip — implementation of the C ABI declared in include/ba_hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ba_handle.h"

namespace {
thread_local std::string g_err;
}  // namespace
void ba::set_last_error(const std::string &m) { g_err = m; }
int ba::fail(const std::string &m) {
  g_err = m;
  return -1;
}
using ba::fail;

namespace {

int use_device(ba_handle *h) {
  HIP_TRY(hipSetDevice(h->device));
  return 0;
}

// one stage boundary: optional event record
inline void mark(ba_handle *h, int k) {
  if (h->timing) (void)hipEventRecord(h->ev[k], h->stream);
}

int xchg(ba_handle *h, int which) {
  if (!h->ar_fn) return 0;
  void *ptr = which == 0 ? (void *)h->d.Spk : (void *)h->d.scal;
  int rc = h->ar_fn(h->ar_user, which, ptr, h->xbuf_n[which], (void *)h->stream);
  if (rc != 0) return fail("all-reduce hook returned an error");
  return 0;
}

// The main stream waits for the side stream's outstanding work (if any).
void join_side(ba_handle *h) {
  h->tiles_ready = false;  // (every entry point but the LM iteration itself may dirty the factor tiles)
  if (!h->side_pending) return;
  (void)hipStreamWaitEvent(h->stream, h->ev_join, 0);
  h->side_pending = false;
}

// Linearisation at the `sel` parameters on the main stream (blocks of buffer
// lcur ^ sel, cost partials as a by-product).
void enqueue_linearize(ba_handle *h, int sel) {
  const ba::DevProblem &d = h->d;
  ba::launch_lin_landmarks(d, sel, h->stream);  // (k_lin_grp's pose-side rows precede k_pose_finalize)
  ba::launch_lin_poses(d, sel, h->stream);
  if (d.n_obs_lm < d.n_obs) ba::launch_cost(d, sel ? 2 : 0, d.n_obs_lm, h->stream);
}

// Enqueue one LM iteration (reference :709-1007) without host sync.
// On entry the block buffer ctrl->lcur holds the linearisation at the accepted
// parameters (made by ba_lm_begin or, as the trial-point linearisation, by the
// previous iteration).  The iteration damps and inverts with the current lambda,
// forms and solves the reduced system, back-substitutes, writes the trial
// parameters, LINEARISES AT THE TRIAL POINT into the other block buffer — the
// trial cost (reference :927) is the sum of the residual norms that pass computes
// anyway — and takes the trust-region decision, which flips parameter and block
// buffers together on acceptance.  Per iteration one pass over the observations
// less than "cost at the trial point, then linearise again" (reference
// :709-927), and a rejected step costs no second linearisation of the same point.
int enqueue_iteration(ba_handle *h) {
  const ba::DevProblem &d = h->d;
  hipStream_t s = h->stream;
  ba::g_ktimer = h->timing ? &h->kt : nullptr;
  // per-kernel / per-stage timing needs the serial order on one stream; a captured
  // graph cannot wait for an event of the previous replay
  const bool ov = h->overlap && !h->timing && !h->use_graph;
  const bool direct = !h->ar_fn;  // single GPU: S, rhs are placed in the dense matrix at once
  h->ddev.flow_ok = !h->use_graph;  // (the dataflow sweep's generation number is a kernel argument)
  mark(h, 0);
  ba::launch_damp_invert(d, s);
  const bool had_side = h->side_pending;
  if (!had_side && !h->tiles_ready)
    ba::launch_dense_init(d.L, d.ld, d.col_x, d.zt_I, d.zt_J, d.n_zt, d.nb, &d.ctrl->done, s);
  h->tiles_ready = false;
  // NO SIDE STREAM when nothing is left for it: single GPU, every landmark with a free
  // pose in a covisibility group (no pose-major pass), no cost pass — the reset of the
  // factor tiles rides in the back-substitution launch (the solve is over by then),
  // the pose-side sums in the k_scalars launch: no fork / join gaps (6 + 8 us at C4).
  static const bool no_side_env = !(getenv("BA_FORCE_SIDE") && getenv("BA_FORCE_SIDE")[0] == '1');
  const bool no_side = no_side_env && !h->ar_fn && d.lin_chunk0 > 0 && d.n_achunk == 0 && d.n_obs_lm == d.n_obs;
  ba::launch_schur_accumulate(d, s);
  join_side(h);  // A_j, a_j of this point and the reset factor tiles come from the side stream
  ba::launch_schur_final(d, direct, s);
  mark(h, 1);
  if (xchg(h, 0)) return -1;
  mark(h, 2);
  if (!direct) ba::launch_scatter(d, s);
  ba::launch_dense_solve(d, h->sched, h->ddev, s);
  mark(h, 3);
  // (single GPU, every free landmark grouped: back-substitution and trial-point linearisation
  //  as roles of ONE launch — k_backsub_lin.  OPT-IN, BA_FUSE_BL=1: measured on MI355X the
  //  interleaved roles do not fill each other's idle pipes — both kernels run two waves per
  //  SIMD and are bound by the latency those two waves cannot hide, mixing them adds no wave:
  //  C4 172.6 us with the roles one after the other (= 68.8 + 104, one launch gap saved: 0.4245
  //  vs 0.428 ms per iteration), 177 / 189 / 203 / 227 us with a lead of 640 / 320 / 160 / 64
  //  pieces.  Not under a captured graph: the generation number is a kernel argument)
  static const bool fuse_env = getenv("BA_FUSE_BL") && getenv("BA_FUSE_BL")[0] == '1';
  const bool fuse_bl = no_side && fuse_env && !h->use_graph && ba::can_fuse_backsub_lin(d) && h->ddev.bad_pivots;
  if (fuse_bl)
    ba::launch_backsub_lin(d, s, true, ++h->bl_gen, h->ddev.bad_pivots);
  else
    ba::launch_backsub_update(d, s, no_side);  // trial parameters, model terms, step norms
  mark(h, 4);
  if (fuse_bl) {
    h->tiles_ready = true;
  } else if (no_side) {
    h->tiles_ready = true;
    ba::launch_lin_landmarks(d, 1, s);
  } else if (ov) {
    // pose side of the trial-point linearisation and the reset of the factor
    // tiles: first needed by the NEXT iteration's k_schur_final.  BA_POSE_LATE=1
    // (default) starts them after k_lin_landmarks, beside the control step, the
    // damping kernel and the first part of k_schur_lds; 0 beside k_lin_landmarks.
    // (with covisibility groups linearised by k_lin_grp the pose-side sums of the
    //  side stream's k_pose_finalize read that kernel's rows: always "late")
    static const bool late_env = !(getenv("BA_POSE_LATE") && getenv("BA_POSE_LATE")[0] == '0');
    const bool late = late_env || d.lin_chunk0 > 0;
    if (late) {
      ba::launch_lin_landmarks(d, 1, s);
      if (d.n_obs_lm < d.n_obs) ba::launch_cost(d, 2, d.n_obs_lm, s);
    }
    (void)hipEventRecord(h->ev_fork, s);
    (void)hipStreamWaitEvent(h->side_stream, h->ev_fork, 0);
    ba::launch_dense_init(d.L, d.ld, d.col_x, d.zt_I, d.zt_J, d.n_zt, d.nb, &d.ctrl->done, h->side_stream);
    ba::launch_lin_poses(d, 1, h->side_stream);
    (void)hipEventRecord(h->ev_join, h->side_stream);
    h->side_pending = true;
    if (!late) {
      ba::launch_lin_landmarks(d, 1, s);
      if (d.n_obs_lm < d.n_obs) ba::launch_cost(d, 2, d.n_obs_lm, s);
    }
  } else {
    enqueue_linearize(h, 1);
  }
  mark(h, 5);
  if (h->ar_fn) {
    ba::launch_scalars(d, 1, s);
    mark(h, 6);
    if (xchg(h, 1)) return -1;
    mark(h, 7);
    ba::launch_control(d, s);
  } else {  // nothing to exchange: the reduction workgroup also takes the LM decision
    ba::launch_scalars_and_control(d, 1, s, no_side ? 1 : -1);
    mark(h, 6);
    mark(h, 7);
  }
  mark(h, 8);
  ba::g_ktimer = nullptr;
  if (h->timing) {
    HIP_TRY(hipStreamSynchronize(s));
    h->kt.collect();
    static const int stage_of[8] = {ST_SCHUR, ST_XCHG, ST_SOLVE, ST_BACKSUB,
                                    ST_BUILD, ST_CTRL, ST_XCHG, ST_CTRL};
    for (int k = 0; k < 8; ++k) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->ev[k], h->ev[k + 1]) == hipSuccess)
        h->stage_ms[stage_of[k]] += ms;
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int push_ctrl(ba_handle *h) {
  HIP_TRY(hipMemcpyAsync(h->d.ctrl, &h->hc, sizeof(ba::DevCtrl),
                         hipMemcpyHostToDevice, h->stream));
  return 0;
}
int pull_ctrl(ba_handle *h) {
  HIP_TRY(hipMemcpyAsync(&h->hc, h->d.ctrl, sizeof(ba::DevCtrl),
                         hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

template <class T>
int download(std::vector<T> &out, const T *dev, size_t n, hipStream_t s) {
  out.resize(n);
  if (n == 0) return 0;
  HIP_TRY(hipMemcpyAsync(out.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

}  // namespace

extern "C" {

const char *ba_last_error(void) { return g_err.c_str(); }

int ba_create(ba_handle **out, int device_id) {
  if (!out) return fail("ba_create: null out pointer");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail("ba_create: no HIP device available (the HIP path has no CPU "
                "fallback)");
  if (device_id < 0 || device_id >= ndev) return fail("ba_create: bad device id");
  ba_handle *h = new ba_handle();
  h->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess ||
      hipStreamCreate(&h->own_stream) != hipSuccess ||
      hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
    delete h;
    return fail("ba_create: cannot create stream");
  }
  h->stream = h->own_stream;
  if (const char *e2 = getenv("BA_NO_OVERLAP")) h->overlap = !(e2[0] == '1');
  if (const char *e3 = getenv("BA_GRAPH")) h->use_graph = (e3[0] == '1');
  std::memset(&h->d, 0, sizeof(h->d));
  std::memset(&h->hc, 0, sizeof(h->hc));
  *out = h;
  return 0;
}

void ba_destroy(ba_handle *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  h->free_device();
  for (hipEvent_t e : h->kt.pool) (void)hipEventDestroy(e);
  h->kt.pool.clear();
  if (h->ev_ok)
    for (int k = 0; k <= ST_N; ++k) (void)hipEventDestroy(h->ev[k]);
  h->drop_graph();
  if (h->side_stream) {
    (void)hipStreamSynchronize(h->side_stream);
    (void)hipStreamDestroy(h->side_stream);
  }
  for (int k = 0; k < 3; ++k)
    if (h->ev_look[k]) (void)hipEventDestroy(h->ev_look[k]);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

int ba_set_stream(ba_handle *h, void *hip_stream) {
  if (!h) return fail("null handle");
  h->drop_graph();
  // the given value is used as is: NULL is HIP's (legacy) default stream, which
  // is also what torch.cuda.current_stream().cuda_stream reports by default
  h->stream = (hipStream_t)hip_stream;
  return 0;
}

int ba_set_cameras(ba_handle *h, int n_cam, const double *intr4,
                   const double *T_cj12) {
  if (!h || n_cam <= 0 || !intr4 || !T_cj12) return fail("ba_set_cameras: bad argument");
  if (h->finalized) return fail("ba_set_cameras: already finalized");
  h->n_cam = n_cam;
  h->cam_intr.assign(intr4, intr4 + 4 * (size_t)n_cam);
  h->cam_T.assign(T_cj12, T_cj12 + 12 * (size_t)n_cam);
  return 0;
}

int ba_set_poses(ba_handle *h, int n_pose, const double *T_jw12,
                 const uint8_t *fixed) {
  if (!h || n_pose <= 0 || !T_jw12) return fail("ba_set_poses: bad argument");
  if (h->finalized) return fail("ba_set_poses: already finalized");
  h->n_pose = n_pose;
  h->pose_T.assign(T_jw12, T_jw12 + 12 * (size_t)n_pose);
  if (fixed)
    h->pose_fixed.assign(fixed, fixed + n_pose);
  else
    h->pose_fixed.assign(n_pose, 0);
  return 0;
}

int ba_set_points(ba_handle *h, int n_pt, const double *X3,
                  const uint8_t *fixed) {
  if (!h || n_pt <= 0 || !X3) return fail("ba_set_points: bad argument");
  if (h->finalized) return fail("ba_set_points: already finalized");
  h->n_pt = n_pt;
  h->pt_X.assign(X3, X3 + 3 * (size_t)n_pt);
  if (fixed)
    h->pt_fixed.assign(fixed, fixed + n_pt);
  else
    h->pt_fixed.assign(n_pt, 0);
  return 0;
}

int ba_set_observations(ba_handle *h, int64_t n_obs, const int32_t *cam,
                        const int32_t *pose, const int32_t *point,
                        const double *uv2) {
  if (!h || n_obs < 0 || (n_obs > 0 && (!cam || !pose || !point || !uv2)))
    return fail("ba_set_observations: bad argument");
  if (h->finalized) return fail("ba_set_observations: already finalized");
  h->n_obs = n_obs;
  h->obs_cam.assign(cam, cam + n_obs);
  h->obs_pose.assign(pose, pose + n_obs);
  h->obs_pt.assign(point, point + n_obs);
  h->obs_uv.assign(uv2, uv2 + 2 * n_obs);
  return 0;
}

int ba_set_shard(ba_handle *h, int rank, int world) {
  if (!h || world < 1 || rank < 0 || rank >= world) return fail("ba_set_shard: bad argument");
  if (h->finalized) return fail("ba_set_shard: already finalized");
  h->rank = rank;
  h->world = world;
  return 0;
}

int ba_partition_points(int n_pose, const uint8_t *pose_fixed, int n_pt,
                        const uint8_t *pt_fixed, int64_t n_obs,
                        const int32_t *obs_pose, const int32_t *obs_pt,
                        int world, int32_t *owner_out) {
  if (n_pose <= 0 || n_pt <= 0 || world < 1 || !owner_out)
    return fail("ba_partition_points: bad argument");
  std::vector<uint8_t> pf(n_pose, 0), qf(n_pt, 0);
  ba::PlanInput in;
  in.n_cam = 1;
  in.n_pose = n_pose;
  in.pose_fixed = pose_fixed ? pose_fixed : pf.data();
  in.n_pt = n_pt;
  in.pt_fixed = pt_fixed ? pt_fixed : qf.data();
  in.n_obs = n_obs;
  in.obs_pose = obs_pose;
  in.obs_pt = obs_pt;
  in.world = world;
  for (int64_t k = 0; k < n_obs; ++k)
    if (obs_pose[k] < 0 || obs_pose[k] >= n_pose || obs_pt[k] < 0 || obs_pt[k] >= n_pt)
      return fail("ba_partition_points: observation index out of range");
  std::vector<int32_t> owner;
  ba::partition_points(in, owner);
  std::memcpy(owner_out, owner.data(), sizeof(int32_t) * (size_t)n_pt);
  return 0;
}

int ba_finalize(ba_handle *h) {
  if (h) h->drop_graph();
  if (!h) return fail("null handle");
  if (h->finalized) return 0;  // idempotent (README of the reference calls it publicly)
  if (h->n_cam <= 0 || h->n_pose <= 0 || h->n_pt <= 0)
    return fail("ba_finalize: cameras, poses and points must be set first");
  if (use_device(h)) return -1;
  ba::PlanInput in;
  in.n_cam = h->n_cam;
  in.n_pose = h->n_pose;
  in.pose_fixed = h->pose_fixed.data();
  in.n_pt = h->n_pt;
  in.pt_fixed = h->pt_fixed.data();
  in.n_obs = h->n_obs;
  in.obs_cam = h->obs_cam.data();
  in.obs_pose = h->obs_pose.data();
  in.obs_pt = h->obs_pt.data();
  in.obs_uv = h->obs_uv.data();
  in.rank = h->rank;
  in.world = h->world;
  const bool times = getenv("BA_PLAN_TIMES") != nullptr;
  h->up_times = times;
  h->up_alloc_s = h->up_copy_s = 0;
  h->up_bytes = h->up_calls = 0;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!times) return;
    (void)hipDeviceSynchronize();
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[finalize] %-26s %7.1f ms   (uploads so far: %zu calls, %.1f MB, alloc %.1f ms, copy %.1f ms)\n", what,
            std::chrono::duration<double, std::milli>(n - t_last).count(), h->up_calls, h->up_bytes / 1e6,
            h->up_alloc_s * 1e3, h->up_copy_s * 1e3);
    t_last = n;
  };
  std::string err = ba::build_plan(in, h->plan);
  if (!err.empty()) return fail("ba_finalize: " + err);
  lap("host plan");
  const ba::Plan &pl = h->plan;
  ba::DevProblem &d = h->d;
  std::memset(&d, 0, sizeof(d));
  d.n_cam = pl.n_cam; d.n_pose = pl.n_pose; d.N = pl.N; d.n_pt = pl.n_pt;
  d.M = pl.M; d.M_global = pl.M_global; d.n_obs = pl.n_obs;
  d.n_obs_opt = pl.n_obs_opt; d.n_obs_global = pl.n_obs_global; d.P = pl.P;
  d.n_pobs = pl.n_pobs; d.T = pl.T; d.B = pl.B;
  d.n_achunk = (int)pl.achunk_pose.size();
  d.n_tchunk = (int)pl.tchunk_blk.size();

  // parameters
  std::vector<double> cams((size_t)pl.n_cam * 16);
  for (int c = 0; c < pl.n_cam; ++c) {
    std::memcpy(&cams[(size_t)c * 16], &h->cam_intr[(size_t)c * 4], 4 * sizeof(double));
    std::memcpy(&cams[(size_t)c * 16 + 4], &h->cam_T[(size_t)c * 12], 12 * sizeof(double));
  }
  // Allocation kinds (ba_handle.h): with a device arena (ba_stream.hip) the arrays
  // that scale with the landmark chunk live in it — structure / observations as
  // kind 1 (restored only), blocks / points / partial sums as kind 2 (saved and
  // restored); everything pose-sized stays resident (kind 0).  Without an arena
  // every kind is a plain hipMalloc.
  h->kind(0);
  if (h->upload(&d.cams, cams)) return -1;
  std::vector<double> poses((size_t)pl.n_pose * 12);
  for (int p = 0; p < pl.n_pose; ++p)
    std::memcpy(&poses[(size_t)p * 12], &h->pose_T[(size_t)pl.pose_user_of_int[p] * 12],
                12 * sizeof(double));
  if (h->upload(&d.poses[0], poses) || h->upload(&d.poses[1], poses)) return -1;
  std::vector<double> pts((size_t)pl.n_pt * 3);
  for (int q = 0; q < pl.n_pt; ++q)
    std::memcpy(&pts[(size_t)q * 3], &h->pt_X[(size_t)pl.pt_user_of_int[q] * 3],
                3 * sizeof(double));
  h->kind(2);
  if (h->upload(&d.pts[0], pts) || h->upload(&d.pts[1], pts)) return -1;
  h->kind(1);

  // structure
  static_assert(sizeof(int4) == 16 && sizeof(double2) == 16, "layout");
  if (h->dalloc(&d.obs_idx, (size_t)pl.n_obs)) return -1;
  if (h->dalloc(&d.obs_uv, (size_t)pl.n_obs)) return -1;
  if (pl.n_obs > 0) {
    HIP_TRY(hipMemcpy(d.obs_idx, pl.obs_idx.data(), (size_t)pl.n_obs * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d.obs_uv, pl.obs_uv.data(), (size_t)pl.n_obs * 16, hipMemcpyHostToDevice));
  }
  // slim landmark-major record for the cost kernel (no pair id, 8 bytes)
  d.obs_cp = nullptr;
  const char *wide = getenv("BA_COST_WIDE");  // test knob: keep k_cost on the 16-byte records
  if (!pl.obs_cp.empty() && pl.n_obs > 0 && !(wide && wide[0] == '1')) {  // (filled by the planner's threaded pass)
    static_assert(sizeof(int2) == 8, "layout");
    if (h->dalloc(&d.obs_cp, (size_t)pl.n_obs)) return -1;
    HIP_TRY(hipMemcpy(d.obs_cp, pl.obs_cp.data(), (size_t)pl.n_obs * 8, hipMemcpyHostToDevice));
  }
  if (h->dalloc(&d.pobs_idx, (size_t)pl.n_pobs)) return -1;
  if (h->dalloc(&d.pobs_uv, (size_t)pl.n_pobs)) return -1;
  if (pl.n_pobs > 0) {
    HIP_TRY(hipMemcpy(d.pobs_idx, pl.pobs_idx.data(), (size_t)pl.n_pobs * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d.pobs_uv, pl.pobs_uv.data(), (size_t)pl.n_pobs * 16, hipMemcpyHostToDevice));
  }
  if (h->upload(&d.lm_obs_ptr, pl.lm_obs_ptr) || h->upload(&d.lm_pair_ptr, pl.lm_pair_ptr) ||
      h->upload(&d.pair_pose, pl.pair_pose) || h->upload(&d.pair_lm, pl.pair_lm) ||
      h->upload(&d.achunk_pose, pl.achunk_pose) || h->upload(&d.achunk_begin, pl.achunk_begin) ||
      h->upload(&d.achunk_end, pl.achunk_end) || h->upload(&d.pose_achunk_ptr, pl.pose_achunk_ptr))
    return -1;
  // the block numbering of S is global (identical on every chunk) and k_scatter reads it
  // when no particular chunk is resident: never in the arena
  h->kind(0);
  if (h->upload(&d.sblk_j, pl.sblk_j) || h->upload(&d.diag_blk, pl.diag_blk) || h->upload(&d.sblk_k, pl.sblk_k))
    return -1;
  h->kind(1);
  if (h->upload(&d.tri_p, pl.tri_p) ||
      h->upload(&d.tri_q, pl.tri_q) || h->upload(&d.tchunk_blk, pl.tchunk_blk) ||
      h->upload(&d.tchunk_begin, pl.tchunk_begin) || h->upload(&d.tchunk_end, pl.tchunk_end) ||
      h->upload(&d.sblk_tchunk_ptr, pl.sblk_tchunk_ptr) ||
      h->upload(&d.ltri, pl.ltri) || h->upload(&d.chunk_sp, pl.chunk_sp) ||
      h->upload(&d.sup_lane, pl.sup_lane) ||
      h->upload(&d.blk_contrib_ptr, pl.blk_contrib_ptr) || h->upload(&d.contrib_slot, pl.contrib_slot))
    return -1;
  {
    std::vector<int32_t> bd((size_t)pl.B * 16, 0);
    for (int64_t bk = 0; bk < pl.B; ++bk) {
      int32_t *r = bd.data() + 16 * bk;
      const int64_t c0 = pl.blk_contrib_ptr[bk], c1 = pl.blk_contrib_ptr[bk + 1];
      r[0] = pl.sblk_j[bk];
      r[1] = pl.sblk_k[bk];
      r[2] = (int32_t)c0;
      r[3] = (int32_t)(c1 - c0);
      r[4] = pl.sblk_tchunk_ptr[bk];
      r[5] = pl.sblk_tchunk_ptr[bk + 1] - pl.sblk_tchunk_ptr[bk];
      for (int t = 0; t < 8; ++t) r[8 + t] = c0 + t < c1 ? pl.contrib_slot[c0 + t] : 0;
    }
    if (pl.contrib_slot.size() >= (size_t)INT32_MAX) return fail("too many slot contributions for int32 indices");
    if (h->upload(&d.blk_desc, bd)) return -1;
  }
  d.n_sup = (int)pl.sup_desc.size();
  d.n_bchunk = (int)pl.bchunk_lm.size() - 1;
  if (h->upload(&d.bchunk_lm, pl.bchunk_lm)) return -1;
  {
    std::vector<ba::DevProblem::LmChunk> lc((size_t)d.n_bchunk);
    for (int c = 0; c < d.n_bchunk; ++c) {
      const int l0 = pl.bchunk_lm[c], l1 = pl.bchunk_lm[c + 1];
      lc[c].pb = pl.lm_pair_ptr[l0];
      lc[c].ob = pl.lm_obs_ptr[l0];
      lc[c].l0 = l0;
      lc[c].nl = l1 - l0;
      lc[c].np = (int32_t)(pl.lm_pair_ptr[l1] - pl.lm_pair_ptr[l0]);
      lc[c].no = (int32_t)(pl.lm_obs_ptr[l1] - pl.lm_obs_ptr[l0]);
    }
    if (h->upload(&d.lm_chunk, lc)) return -1;
  }
  d.n_slot = (int)pl.slot_blk.size();
  {
    static_assert(sizeof(ba::Plan::SupDesc) == sizeof(ba::DevProblem::SupDesc), "desc layout");
    static_assert(sizeof(ba::Plan::ChunkDesc) == sizeof(ba::DevProblem::ChunkDesc), "desc layout");
    if (h->dalloc(&d.sup_desc, pl.sup_desc.size()) || h->dalloc(&d.chunk_desc, pl.chunk_desc.size()))
      return -1;
    if (!pl.sup_desc.empty())
      HIP_TRY(hipMemcpy(d.sup_desc, pl.sup_desc.data(), pl.sup_desc.size() * sizeof(ba::Plan::SupDesc),
                        hipMemcpyHostToDevice));
    if (!pl.chunk_desc.empty())
      HIP_TRY(hipMemcpy(d.chunk_desc, pl.chunk_desc.data(),
                        pl.chunk_desc.size() * sizeof(ba::Plan::ChunkDesc), hipMemcpyHostToDevice));
  }
  h->kind(2);
  if (h->dalloc(&d.spart2, (size_t)d.n_slot * ba::kSlotStride)) return -1;
  h->kind(1);
  {
    static_assert(sizeof(ba::Plan::GrpDesc) == sizeof(ba::DevProblem::GrpDesc) && sizeof(ba::Plan::GrpDesc) == 128,
                  "group descriptor layout");
    static_assert(sizeof(ba::Plan::LinDesc) == sizeof(ba::DevProblem::LinDesc) && sizeof(ba::Plan::LinDesc) == 48,
                  "group linearisation descriptor layout");
    d.n_grp32 = (int)pl.grp32.size();
    d.n_grp64 = (int)pl.grp64.size();
    d.n_grp128 = (int)pl.grp128.size();
    if (h->dalloc(&d.grp32, pl.grp32.size()) || h->dalloc(&d.grp64, pl.grp64.size()) ||
        h->dalloc(&d.grp128, pl.grp128.size()))
      return -1;
    if (d.n_grp128)
      HIP_TRY(hipMemcpy(d.grp128, pl.grp128.data(), pl.grp128.size() * sizeof(ba::Plan::GrpDesc), hipMemcpyHostToDevice));
    if (d.n_grp32)
      HIP_TRY(hipMemcpy(d.grp32, pl.grp32.data(), pl.grp32.size() * sizeof(ba::Plan::GrpDesc), hipMemcpyHostToDevice));
    if (d.n_grp64)
      HIP_TRY(hipMemcpy(d.grp64, pl.grp64.data(), pl.grp64.size() * sizeof(ba::Plan::GrpDesc), hipMemcpyHostToDevice));
    // k_lin_grp: observation patterns, pose-side partial sums of the group pieces
    d.lin_chunk0 = pl.lin_groups ? pl.n_bchunk_grp : 0;
    d.n_lin_desc = (int)pl.lin_desc.size();
    d.n_lin_plain = pl.n_lin_plain;
    d.n_bs_grp = d.n_lin_desc;
    d.n_lm_part = d.n_bs_grp + (d.n_bchunk - d.lin_chunk0 + ba::kBsChunks - 1) / ba::kBsChunks;
    d.n_lin_cost = d.n_bchunk + d.n_lin_desc;
    if (h->dalloc(&d.lin_desc, pl.lin_desc.size())) return -1;
    if (d.n_lin_desc)
      HIP_TRY(hipMemcpy(d.lin_desc, pl.lin_desc.data(), pl.lin_desc.size() * sizeof(ba::Plan::LinDesc), hipMemcpyHostToDevice));
    if (h->dalloc(&d.grp_pat, pl.grp_pat.size() / 2) || h->upload(&d.pose_gpart_ptr, pl.pose_gpart_ptr) ||
        h->upload(&d.pose_gpart, pl.pose_gpart))
      return -1;
    h->kind(2);
    if (h->dalloc(&d.Apart2, (size_t)pl.n_apart2 * 27) || h->dalloc(&d.lin_dump, (size_t)ba::kLinDump)) return -1;
    h->kind(0);
    if (h->dalloc(&d.bl_flag, pl.lin_desc.size() + 1) || h->dalloc(&d.pose_flag, (size_t)ba::kPoseGrid)) return -1;
    HIP_TRY(hipMemset(d.bl_flag, 0, (pl.lin_desc.size() + 1) * sizeof(int)));
    HIP_TRY(hipMemset(d.pose_flag, 0, (size_t)ba::kPoseGrid * sizeof(int)));
    h->bl_gen = 0;
    h->kind(2);
    if (!pl.grp_pat.empty())
      HIP_TRY(hipMemcpy(d.grp_pat, pl.grp_pat.data(), pl.grp_pat.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (pl.n_apart2 > 0) HIP_TRY(hipMemset(d.Apart2, 0, (size_t)pl.n_apart2 * 27 * sizeof(double)));
  }

  lap("structure uploads");
  // per-iteration storage
  for (int k = 0; k < 2; ++k) {
    h->kind(2);
    if (h->dalloc(&d.Cu[k], (size_t)pl.M * 6) || h->dalloc(&d.b[k], (size_t)pl.M * 3) ||
        h->dalloc(&d.W[k], std::max<size_t>(2, (size_t)pl.P * ba::kWStride)))
      return -1;
    h->kind(0);
    if (h->dalloc(&d.A[k], (size_t)pl.N * 36) || h->dalloc(&d.a[k], (size_t)pl.N * 6)) return -1;
    HIP_TRY(hipMemset(d.W[k], 0, std::max<size_t>(1, (size_t)pl.P * ba::kWStride) * sizeof(double)));
    HIP_TRY(hipMemset(d.A[k], 0, std::max<size_t>(1, (size_t)pl.N * 36) * sizeof(double)));
    HIP_TRY(hipMemset(d.a[k], 0, std::max<size_t>(1, (size_t)pl.N * 6) * sizeof(double)));
  }
  d.n_obs_lm = pl.M > 0 ? pl.lm_obs_ptr[pl.M] : 0;
  h->kind(2);
  if (h->dalloc(&d.Cd, (size_t)pl.M * 6) || h->dalloc(&d.Cinv, (size_t)pl.M * 6) ||
      h->dalloc(&d.lin_cost_part, (size_t)std::max(1, d.n_lin_cost)) ||
      h->dalloc(&d.Apart, (size_t)d.n_achunk * 27) ||
      h->dalloc(&d.spart, (size_t)d.n_tchunk * ba::kSlotStride) ||
      h->dalloc(&d.y, (size_t)pl.M * 3) || h->dalloc(&d.lm_part, (size_t)std::max(1, d.n_lm_part) * 2))
    return -1;
  h->kind(0);
  if (h->dalloc(&d.x, (size_t)pl.N * 6 + 64) || h->dalloc(&d.cost_part, (size_t)ba::kCostGrid) ||
      h->dalloc(&d.pose_part, (size_t)2 + 2 * ba::kPoseGrid) ||
      h->dalloc(&d.scal, (size_t)4) || h->dalloc(&d.ctrl, (size_t)1))
    return -1;
  HIP_TRY(hipMemset(d.lin_cost_part, 0, (size_t)std::max(1, d.n_lin_cost) * sizeof(double)));
  HIP_TRY(hipMemset(d.x, 0, ((size_t)pl.N * 6 + 64) * sizeof(double)));
  HIP_TRY(hipMemset(d.y, 0, std::max<size_t>(1, (size_t)pl.M * 3) * sizeof(double)));
  HIP_TRY(hipMemset(d.cost_part, 0, ba::kCostGrid * sizeof(double)));
  HIP_TRY(hipMemset(d.lm_part, 0, (size_t)std::max(1, d.n_lm_part) * 2 * sizeof(double)));
  HIP_TRY(hipMemset(d.pose_part, 0, (2 + 2 * ba::kPoseGrid) * sizeof(double)));
  HIP_TRY(hipMemset(d.scal, 0, 4 * sizeof(double)));
  d.log_cap = 4096;
  if (h->dalloc(&d.log, (size_t)d.log_cap)) return -1;

  lap("block storage");
  // dense reduced system: tiles of 5 poses (32 columns) or 10 poses (64
  // columns), eliminated in the order of the level schedule.  Both schedules are
  // built.  NARROW patterns (every column tile has at most five row tiles below it
  // besides the rhs block: windows of <= ~15 poses) are latency-bound chains of
  // dependent launches, one set per level, whose cost is a fixed latency plus a term
  // proportional to the tile order (measured on MI355X: 16 us per level at 32, 30 us
  // at 64): the cheaper chain wins.  Wider and dense patterns are bound by the work
  // per level (row tiles per column, MFMA tile size) and run 1.1-1.9x faster at 64
  // (W20: 1.08 vs 1.21 ms per iteration, DENSE1K: 10.0 vs 19.0).  BA_DENSE_NB=32|64
  // forces one.
  if (h->dense_owner) {
    // streaming (ba_stream.hip): the reduced system is scattered, factorised and solved
    // ONCE per iteration, by the owner; this handle aliases its dense image, schedule
    // and solution (the S-block numbering and the tile schedule are global: identical
    // on every landmark chunk) and keeps only its own packed partial S||rhs
    const ba_handle *o = h->dense_owner;
    if (!o->finalized || o->plan.N != pl.N || o->plan.B != pl.B)
      return fail("ba_finalize: the dense owner belongs to a different problem");
    h->sched = o->sched;
    h->ddev = o->ddev;
    h->pose_col_h = o->pose_col_h;
    d.nb = o->d.nb; d.npad = o->d.npad; d.ld = o->d.ld;
    d.L = o->d.L; d.Ldiag = o->d.Ldiag; d.pose_col = o->d.pose_col; d.col_x = o->d.col_x;
    d.zt_I = o->d.zt_I; d.zt_J = o->d.zt_J; d.n_zt = o->d.n_zt;
    d.x = o->d.x;
    h->xbuf_n[0] = pl.B * 36 + 6 * (int64_t)pl.N;
    h->xbuf_n[1] = 4;
    h->xbuf_n[2] = 3 * (int64_t)pl.n_pt_global;
    if (h->dalloc(&d.Spk, (size_t)h->xbuf_n[0])) return -1;
    HIP_TRY(hipMemset(d.Spk, 0, (size_t)h->xbuf_n[0] * sizeof(double)));
  } else {
    const char *nat = getenv("BA_DENSE_NATURAL");
    const char *full = getenv("BA_DENSE_FULL");
    const char *force = getenv("BA_DENSE_NB");
    ba::DenseSchedule cand[2];
    double cost[2];
    const int orders[2] = {32, 64};
    const double level_us[2] = {16.0, 30.0};
    for (int k = 0; k < 2; ++k) {
      int ncb_k = 0;
      std::vector<uint8_t> adj;
      ba::tile_pattern(pl, ba::dense_poses_per_tile(orders[k]), ncb_k, adj);
      if (full && atoi(full) != 0) std::fill(adj.begin(), adj.end(), 1);
      ba::build_dense_schedule(ncb_k, adj, nat && atoi(nat) != 0, orders[k], cand[k]);
      cost[k] = cand[k].nlev * level_us[k];
    }
    if (getenv("BA_PLAN_STATS"))
      fprintf(stderr, "dense schedules: nb32 %d tiles %d levels max_rows %d fill %.3f | nb64 %d tiles %d levels max_rows %d fill %.3f\n",
              cand[0].ncb, cand[0].nlev, cand[0].max_rows, cand[0].fill, cand[1].ncb, cand[1].nlev, cand[1].max_rows,
              cand[1].fill);
    const bool narrow = cand[0].max_rows <= 6;
    int pick = (narrow && cost[0] <= cost[1]) ? 0 : 1;
    if (force && atoi(force) == 32) pick = 0;
    if (force && atoi(force) == 64) pick = 1;
    h->sched = cand[pick];
    const int nb = h->sched.nb;
    const int ppt = ba::dense_poses_per_tile(nb);
    const int ncb = h->sched.ncb;
    d.nb = nb;
    d.npad = ncb * nb;
    d.ld = d.npad + nb;
    h->xbuf_n[0] = pl.B * 36 + 6 * (int64_t)pl.N;
    h->xbuf_n[1] = 4;
    h->xbuf_n[2] = 3 * (int64_t)pl.n_pt_global;
    if (h->dalloc(&d.Spk, (size_t)h->xbuf_n[0])) return -1;
    if (h->dalloc(&d.L, (size_t)d.npad * d.ld)) return -1;
    if (h->dalloc(&d.Ldiag, (size_t)ncb * ba::dense_ws_per_block(nb))) return -1;
    h->pose_col_h.assign(pl.N, 0);
    std::vector<int> col_x((size_t)d.npad, -1);
    for (int j = 0; j < pl.N; ++j) {
      const int c0 = h->sched.pos_of_tile[j / ppt] * nb + 6 * (j % ppt);
      h->pose_col_h[j] = c0;
      for (int r = 0; r < 6; ++r) col_x[c0 + r] = 6 * j + r;
    }
    const ba::DenseSchedule &sc = h->sched;
    ba::DenseDev &dd = h->ddev;
    if (h->upload(&d.pose_col, h->pose_col_h) || h->upload(&d.col_x, col_x) ||
        h->upload(&dd.row_ptr, sc.row_ptr) || h->upload(&dd.rows, sc.rows) ||
        h->upload(&dd.item_t, sc.item_t) || h->upload(&dd.item_I, sc.item_I) ||
        h->upload(&dd.tgt_I, sc.tgt_I) || h->upload(&dd.tgt_J, sc.tgt_J) ||
        h->upload(&dd.tgt_src_ptr, sc.tgt_src_ptr) || h->upload(&dd.src_t, sc.src_t) ||
        h->upload(&dd.tgt_desc, sc.tgt_desc) || h->upload(&dd.back_desc, sc.back_desc) ||
        h->upload(&dd.row_desc, sc.row_desc) ||
        h->dalloc(&dd.xc, (size_t)d.npad) || h->dalloc(&dd.bad_pivots, (size_t)1))
      return -1;
    HIP_TRY(hipMemset(dd.bad_pivots, 0, sizeof(int)));
    if (sc.fused_ok &&
        (h->upload(&dd.f_desc, sc.f_desc) || h->upload(&dd.f_pend, sc.f_pend) ||
         h->dalloc(&dd.cbuf, (size_t)std::max(1, sc.n_contrib) * nb * nb)))
      return -1;
    dd.col_x = d.col_x;
    dd.read_env();
    dd.aux_stream = h->side_stream;
    if (!h->ev_look[0]) {
      for (int k = 0; k < 3; ++k) HIP_TRY(hipEventCreateWithFlags(&h->ev_look[k], hipEventDisableTiming));
    }
    dd.ev_m = h->ev_look[0];
    dd.ev_x[0] = h->ev_look[1];
    dd.ev_x[1] = h->ev_look[2];
    {
      std::vector<int> order;
      dd.flow_tail_t0 = ba::dense_flow_order(sc, dd, order);
      dd.n_flow = (int)order.size();
      dd.flow_gen = 0;
      if (h->upload(&dd.flow_order, order) || h->dalloc(&dd.flow_flags, (size_t)std::max(1, ncb)) ||
          h->dalloc(&dd.flow_ticket, (size_t)1) || h->dalloc(&dd.fwd_flags, (size_t)std::max(1, ncb)) ||
          h->dalloc(&dd.fwd_ticket, (size_t)1))
        return -1;
      HIP_TRY(hipMemset(dd.flow_flags, 0, (size_t)std::max(1, ncb) * sizeof(int)));
      HIP_TRY(hipMemset(dd.flow_ticket, 0, sizeof(int)));
      HIP_TRY(hipMemset(dd.fwd_flags, 0, (size_t)std::max(1, ncb) * sizeof(int)));
      HIP_TRY(hipMemset(dd.fwd_ticket, 0, sizeof(int)));
      std::vector<int> items, pre, need;
      if (ba::dense_fwd_items(sc, dd, items, pre, need)) {
        dd.n_fwd_items = (int)items.size() / 2;
        dd.n_fwd_cnt = (int)need.size();
        if (h->upload(&dd.fwd_items, items) || h->upload(&dd.upd_pre, pre) || h->upload(&dd.col_need, need) ||
            h->dalloc(&dd.fwd_cnt, need.size()))
          return -1;
        HIP_TRY(hipMemset(dd.fwd_cnt, 0, need.size() * sizeof(int)));
      }
      std::vector<int> ntrsm, lneed;
      if (ba::dense_dag_items(sc, dd, items, pre, need, ntrsm, lneed)) {
        dd.n_dag_items = (int)items.size() / 2;
        dd.n_fwd_cnt = (int)need.size();
        if (times) fprintf(stderr, "[finalize] k_chol_dag: %d items\n", dd.n_dag_items);
        if (h->upload(&dd.dag_items, items) || h->upload(&dd.upd_pre, pre) || h->upload(&dd.col_need, need) ||
            h->upload(&dd.dag_ntrsm, ntrsm) || h->upload(&dd.look_need, lneed) || h->dalloc(&dd.fwd_cnt, need.size()) ||
            h->dalloc(&dd.dag_dflags, need.size()) || h->dalloc(&dd.dag_tcnt, need.size()))
          return -1;
        HIP_TRY(hipMemset(dd.fwd_cnt, 0, need.size() * sizeof(int)));
        HIP_TRY(hipMemset(dd.dag_dflags, 0, need.size() * sizeof(int)));
        HIP_TRY(hipMemset(dd.dag_tcnt, 0, need.size() * sizeof(int)));
      }
    }
    HIP_TRY(hipMemset(dd.xc, 0, (size_t)d.npad * sizeof(double)));
    // tiles (re)initialised per iteration: factor pattern + diagonal + rhs row
    std::vector<int> ztI, ztJ;
    for (int p = 0; p < ncb; ++p) {
      ztI.push_back(p);
      ztJ.push_back(p);
      for (int q = sc.row_ptr[p]; q < sc.row_ptr[p + 1]; ++q) {
        ztI.push_back(sc.rows[q]);  // includes the rhs row block (== ncb)
        ztJ.push_back(p);
      }
    }
    d.n_zt = (int)ztI.size();
    if (h->upload(&d.zt_I, ztI) || h->upload(&d.zt_J, ztJ)) return -1;
    HIP_TRY(hipMemset(d.L, 0, (size_t)d.npad * d.ld * sizeof(double)));
    HIP_TRY(hipMemset(d.Spk, 0, (size_t)h->xbuf_n[0] * sizeof(double)));
  }

  lap("dense schedule + image");
  std::memset(&h->hc, 0, sizeof(h->hc));
  h->hc.lambda = 100.0;
  h->hc.huber = 1.0;
  h->hc.max_iter = 1;
  HIP_TRY(hipMemcpy(d.ctrl, &h->hc, sizeof(ba::DevCtrl), hipMemcpyHostToDevice));
  HIP_TRY(hipDeviceSynchronize());
  h->finalized = true;
  return 0;
}

int ba_update_values(ba_handle *h, const double *T_jw12, const double *X3) {
  if (!h || !h->finalized) return fail("ba_update_values: not finalized");
  if (h->arena) return fail("ba_update_values: not available on a streamed chunk");
  if (use_device(h)) return -1;
  const ba::Plan &pl = h->plan;
  join_side(h);
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (T_jw12) {
    h->pose_T.assign(T_jw12, T_jw12 + 12 * (size_t)pl.n_pose);
    std::vector<double> poses((size_t)pl.n_pose * 12);
    for (int p = 0; p < pl.n_pose; ++p)
      std::memcpy(&poses[(size_t)p * 12], T_jw12 + (size_t)pl.pose_user_of_int[p] * 12, 12 * sizeof(double));
    for (int k = 0; k < 2; ++k)
      HIP_TRY(hipMemcpy(h->d.poses[k], poses.data(), poses.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  if (X3) {
    h->pt_X.assign(X3, X3 + 3 * (size_t)pl.n_pt_global);
    std::vector<double> pts((size_t)pl.n_pt * 3);
    for (int q = 0; q < pl.n_pt; ++q)
      std::memcpy(&pts[(size_t)q * 3], X3 + (size_t)pl.pt_user_of_int[q] * 3, 3 * sizeof(double));
    if (pl.n_pt > 0)
      for (int k = 0; k < 2; ++k)
        HIP_TRY(hipMemcpy(h->d.pts[k], pts.data(), pts.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  h->gathered_valid = false;
  h->lm_begun = false;   // the blocks on the device belong to the old values: ba_lm_begin linearises again
  h->tiles_ready = false;
  return 0;
}

int ba_set_allreduce(ba_handle *h, ba_allreduce_fn fn, void *user) {
  if (!h) return fail("null handle");
  h->ar_fn = fn;
  h->ar_user = user;
  return 0;
}

int64_t ba_reduce_buffer_size(ba_handle *h, int which) {
  if (!h || !h->finalized || which < 0 || which > 2) return -1;
  return h->xbuf_n[which];
}

int ba_bind_reduce_buffer(ba_handle *h, int which, void *dev_ptr, int64_t n) {
  if (!h || !h->finalized) return fail("ba_bind_reduce_buffer: not finalized");
  h->drop_graph();
  if (which < 0 || which > 2 || !dev_ptr || n < h->xbuf_n[which])
    return fail("ba_bind_reduce_buffer: bad argument");
  if (which == 0)
    h->d.Spk = (double *)dev_ptr;
  else if (which == 1)
    h->d.scal = (double *)dev_ptr;
  else {
    h->gbuf = (double *)dev_ptr;  // (a buffer of the library's own stays in `allocs`)
    h->gbuf_bound = true;
  }
  return 0;
}

// ---------------------------------------------------------------------------
}  // extern "C"

// Controller state for a new LM loop (reference :705-708): everything of
// ba_lm_begin but the first linearisation.  *done_after = the loop is over before it
// starts (max_num_iterations <= 0).  Shared with ba_stream.hip.
int ba::lm_prepare_ctrl(ba_handle *h, const ba_options *opt, int *done_after) {
  if (opt->max_num_iterations > h->d.log_cap) {
    // grow the device-side iteration log
    ba::DevIterRec *nl = nullptr;
    const int keep = h->alloc_kind;
    h->alloc_kind = 0;
    const int rc = h->dalloc(&nl, (size_t)opt->max_num_iterations);
    h->alloc_kind = keep;
    if (rc) return -1;
    h->drop_graph();  // the captured kernels hold the old pointer
    h->d.log = nl;
    h->d.log_cap = opt->max_num_iterations;
  }
  h->tiles_ready = false;
  h->gathered_valid = false;
  if (pull_ctrl(h)) return -1;  // keep `cur` and `lcur`
  ba::DevCtrl &c = h->hc;
  c.lambda = (double)opt->initial_lambda;
  c.huber = (double)opt->threshold_huber_loss;
  c.thr_step = (double)opt->threshold_step_size;
  c.thr_cost = (double)opt->threshold_cost_change;
  c.dec_ratio = (double)opt->decrease_ratio_lambda;
  c.inc_ratio = (double)opt->increase_ratio_lambda;
  c.max_iter = opt->max_num_iterations;
  c.gn = opt->gauss_newton ? 1 : 0;
  c.iter = 0;
  c.converged = 0;
  c.prev_cost = 0.0;
  *done_after = (opt->max_num_iterations <= 0) ? 1 : 0;
  c.done = 0;
  if (push_ctrl(h)) return -1;
  HIP_TRY(hipMemsetAsync(h->ddev.bad_pivots, 0, sizeof(int), h->stream));
  return 0;
}
int ba::ctrl_pull(ba_handle *h) { return pull_ctrl(h); }
int ba::ctrl_push(ba_handle *h) { return push_ctrl(h); }

extern "C" {

int ba_lm_begin(ba_handle *h, const ba_options *opt) {
  if (!h || !opt) return fail("ba_lm_begin: bad argument");
  if (!h->finalized && ba_finalize(h)) return -1;
  if (use_device(h)) return -1;
  join_side(h);
  int done_after = 0;
  if (ba::lm_prepare_ctrl(h, opt, &done_after)) return -1;
  // first linearisation, at the starting point; previous_cost =
  // EvaluateCurrentCost() (reference :707) is the sum of its residual norms
  enqueue_linearize(h, 0);
  ba::launch_scalars_cost_only(h->d, 1, h->stream);
  if (xchg(h, 1)) return -1;
  ba::launch_init_ctrl_cost(h->d, h->stream);
  if (done_after) {
    if (pull_ctrl(h)) return -1;
    h->hc.done = 1;
    if (push_ctrl(h)) return -1;
  }
  HIP_TRY(hipGetLastError());
  h->lm_begun = true;
  return 0;
}

int ba_lm_iterate(ba_handle *h, int n) {
  if (!h || !h->lm_begun) return fail("ba_lm_iterate: call ba_lm_begin first");
  if (use_device(h)) return -1;
  h->gathered_valid = false;
  // Graph replay: the kernels early-exit on the device-side `done` word and
  // take the whole problem by value, so one captured iteration is valid until
  // the problem, the stream or the exchange buffers change (drop_graph()).
  const bool graphable = h->use_graph && !h->timing && !h->ar_fn && h->stream != nullptr;
  if (graphable && !h->graph_exec && n > 0) {
    HIP_TRY(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_iteration(h);
    hipGraph_t g = nullptr;
    const hipError_t ec = hipStreamEndCapture(h->stream, &g);
    if (rc != 0 || ec != hipSuccess || !g) {
      if (g) (void)hipGraphDestroy(g);
      (void)hipGetLastError();
      h->use_graph = false;  // fall back to plain launches on this handle
    } else {
      h->graph = g;
      if (hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        h->drop_graph();
        h->use_graph = false;
      }
    }
  }
  for (int k = 0; k < n; ++k) {
    if (graphable && h->graph_exec) {
      HIP_TRY(hipGraphLaunch(h->graph_exec, h->stream));
    } else if (enqueue_iteration(h)) {
      return -1;
    }
  }
  return 0;
}

int ba_lm_sync(ba_handle *h, ba_iter_info *out, int cap, int *n_iter,
               int *converged) {
  if (!h || !h->lm_begun) return fail("ba_lm_sync: call ba_lm_begin first");
  if (use_device(h)) return -1;
  if (h->side_pending) (void)hipStreamWaitEvent(h->stream, h->ev_join, 0);  // (stays pending: the
  //   next iteration joins it again, which is harmless)
  if (pull_ctrl(h)) return -1;
  {
    // A hand-off of a dataflow sweep of the reduced solve that timed out (bounded
    // polls, ba_dense_tile.inc) left x partly unsolved: the device added kFlowTimeout
    // to the pivot counter.  That is an ERROR of the solve, not a dropped pivot.
    int bp = 0;
    HIP_TRY(hipMemcpy(&bp, h->ddev.bad_pivots, sizeof(int), hipMemcpyDeviceToHost));
    if (bp >= ba::kFlowTimeout)
      return fail("ba_lm_sync: a dataflow hand-off of the reduced solve timed out (" +
                  std::to_string(bp / ba::kFlowTimeout) + " polls gave up); the iterations since ba_lm_begin "
                  "are invalid. BA_DENSE_FLOW=0 selects the per-level launches");
  }
  const int n = h->hc.iter;
  if (n_iter) *n_iter = n;
  if (converged) *converged = h->hc.converged;
  if (out && cap > 0 && n > 0) {
    static_assert(sizeof(ba::DevIterRec) == sizeof(ba_iter_info), "iter layout");
    const int m = std::min(std::min(n, cap), h->d.log_cap);
    HIP_TRY(hipMemcpy(out, h->d.log, (size_t)m * sizeof(ba_iter_info), hipMemcpyDeviceToHost));
  }
  return h->hc.done ? 1 : 0;  // 1 = loop finished
}

int ba_solve(ba_handle *h, const ba_options *opt, ba_iter_info *out, int cap,
             int *n_iter, int *converged) {
  if (ba_lm_begin(h, opt)) return -1;
  int done = opt->max_num_iterations <= 0;
  int issued = 0;
  while (!done) {
    const int batch = std::min(4, opt->max_num_iterations - issued);
    if (batch <= 0) break;
    if (ba_lm_iterate(h, batch)) return -1;
    issued += batch;
    int rc = ba_lm_sync(h, nullptr, 0, nullptr, nullptr);
    if (rc < 0) return -1;
    done = rc;
  }
  int rc = ba_lm_sync(h, out, cap, n_iter, converged);
  return rc < 0 ? -1 : 0;
}

// ---------------------------------------------------------------------------
// stage API
int ba_stage_cost(ba_handle *h, double *cost) {
  if (!h || !h->finalized || !cost) return fail("ba_stage_cost: bad argument");
  if (use_device(h)) return -1;
  if (pull_ctrl(h)) return -1;
  h->hc.done = 0;
  if (push_ctrl(h)) return -1;
  join_side(h);
  ba::launch_cost(h->d, 0, 0, h->stream);
  ba::launch_scalars_cost_only(h->d, 0, h->stream);
  if (xchg(h, 1)) return -1;
  HIP_TRY(hipMemcpyAsync(cost, h->d.scal, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

int ba_stage_linearize(ba_handle *h, double lambda, double huber) {
  if (!h || !h->finalized) return fail("ba_stage_linearize: not finalized");
  if (use_device(h)) return -1;
  if (pull_ctrl(h)) return -1;
  h->hc.done = 0;
  h->hc.lambda = lambda;
  h->hc.huber = huber;
  if (push_ctrl(h)) return -1;
  join_side(h);
  enqueue_linearize(h, 0);
  ba::launch_damp_invert(h->d, h->stream);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipGetLastError());
  return 0;
}

int ba_stage_schur(ba_handle *h) {
  if (!h || !h->finalized) return fail("ba_stage_schur: not finalized");
  if (use_device(h)) return -1;
  join_side(h);
  ba::launch_schur(h->d, /*direct=*/false, /*with_init=*/true, h->stream);
  if (xchg(h, 0)) return -1;
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipGetLastError());
  return 0;
}

int ba_stage_solve_reduced(ba_handle *h) {
  if (!h || !h->finalized) return fail("ba_stage_solve_reduced: not finalized");
  if (use_device(h)) return -1;
  ba::launch_scatter(h->d, h->stream);  // packed (reduced) S||rhs -> dense
  ba::launch_dense_solve(h->d, h->sched, h->ddev, h->stream);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipGetLastError());
  return 0;
}

int ba_stage_backsub_update(ba_handle *h) {
  if (!h || !h->finalized) return fail("ba_stage_backsub_update: not finalized");
  if (use_device(h)) return -1;
  ba::launch_backsub_update(h->d, h->stream);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipGetLastError());
  return 0;
}

int ba_stage_scalars(ba_handle *h, double *trial_cost, double *model_change,
                     double *pose_step_sum, double *point_step_sum) {
  if (!h || !h->finalized) return fail("ba_stage_scalars: not finalized");
  if (use_device(h)) return -1;
  ba::launch_cost(h->d, 1, 0, h->stream);
  ba::launch_scalars(h->d, 0, h->stream);
  if (xchg(h, 1)) return -1;
  double sc[4], pp[2];
  HIP_TRY(hipMemcpyAsync(sc, h->d.scal, sizeof(sc), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(pp, h->d.pose_part, sizeof(pp), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (trial_cost) *trial_cost = sc[0];
  if (model_change) *model_change = -sc[1];
  if (point_step_sum) *point_step_sum = sc[2];
  if (pose_step_sum) *pose_step_sum = pp[1];
  return 0;
}

int ba_stage_commit(ba_handle *h, int accept) {
  if (!h || !h->finalized) return fail("ba_stage_commit: not finalized");
  if (use_device(h)) return -1;
  h->gathered_valid = false;
  if (pull_ctrl(h)) return -1;
  if (accept) h->hc.cur ^= 1;
  if (push_ctrl(h)) return -1;
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

int ba_enable_stage_timing(ba_handle *h, int on) {
  if (!h) return fail("null handle");
  if (use_device(h)) return -1;
  if (on && !h->ev_ok) {
    for (int k = 0; k <= ST_N; ++k) HIP_TRY(hipEventCreate(&h->ev[k]));
    h->ev_ok = true;
  }
  h->timing = on != 0;
  h->kt.on = h->timing;
  return 0;
}

int ba_get_stage_ms(ba_handle *h, double out8[8], int reset) {
  if (!h || !out8) return fail("ba_get_stage_ms: bad argument");
  for (int k = 0; k < 8; ++k) out8[k] = h->stage_ms[k];
  if (reset)
    for (int k = 0; k < 8; ++k) h->stage_ms[k] = 0.0;
  return 0;
}

// ---------------------------------------------------------------------------
// readers
int ba_num_opt_poses(ba_handle *h) { return (h && h->finalized) ? h->plan.N : -1; }
int ba_num_opt_points(ba_handle *h) { return (h && h->finalized) ? h->plan.M : -1; }
// (pairs that a masked covisibility group pads in — no observation, W = 0 — are an
//  internal device of the layout: the readers do not show them)
int64_t ba_num_pairs(ba_handle *h) { return (h && h->finalized) ? h->plan.P - h->plan.n_pair_pad : -1; }
int64_t ba_num_schur_blocks(ba_handle *h) { return (h && h->finalized) ? h->plan.B : -1; }
int64_t ba_num_schur_triples(ba_handle *h) { return (h && h->finalized) ? h->plan.T : -1; }

int ba_get_poses(ba_handle *h, double *T_jw12) {
  if (!h || !h->finalized || !T_jw12) return fail("ba_get_poses: bad argument");
  if (use_device(h) || pull_ctrl(h)) return -1;
  std::vector<double> buf;
  if (download(buf, h->d.poses[h->hc.cur], (size_t)h->plan.n_pose * 12, h->stream)) return -1;
  for (int p = 0; p < h->plan.n_pose; ++p)
    std::memcpy(T_jw12 + (size_t)h->plan.pose_user_of_int[p] * 12, &buf[(size_t)p * 12],
                12 * sizeof(double));
  return 0;
}

// rows of `src` (internal point order) to user order in `dst`
__global__ void k_points_to_user(const double *__restrict__ src, const int32_t *__restrict__ user_of_int, int n,
                                 double *__restrict__ dst) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const size_t u = (size_t)user_of_int[q] * 3;
  dst[u + 0] = src[(size_t)q * 3 + 0];
  dst[u + 1] = src[(size_t)q * 3 + 1];
  dst[u + 2] = src[(size_t)q * 3 + 2];
}

int ba_gather_points(ba_handle *h) {
  if (!h || !h->finalized) return fail("ba_gather_points: not finalized");
  if (use_device(h)) return -1;
  h->gathered_valid = false;
  if (!h->ar_fn || h->world <= 1) return 0;  // one shard owns every point: nothing to gather
  const ba::Plan &pl = h->plan;
  const size_t n3 = (size_t)pl.n_pt_global * 3;
  if (!h->gbuf && h->dalloc(&h->gbuf, n3)) return -1;
  if (!h->pt_user_dev && h->upload(&h->pt_user_dev, pl.pt_user_of_int)) return -1;
  if (pull_ctrl(h)) return -1;  // `cur` (synchronises the stream)
  join_side(h);
  HIP_TRY(hipMemsetAsync(h->gbuf, 0, n3 * sizeof(double), h->stream));
  if (pl.n_pt > 0)
    hipLaunchKernelGGL(k_points_to_user, dim3((pl.n_pt + 255) / 256), dim3(256), 0, h->stream,
                       (const double *)h->d.pts[h->hc.cur], (const int32_t *)h->pt_user_dev, pl.n_pt, h->gbuf);
  if (h->ar_fn(h->ar_user, 2, (void *)h->gbuf, (int64_t)n3, (void *)h->stream) != 0)
    return fail("ba_gather_points: all-reduce hook returned an error");
  if (download(h->gathered, h->gbuf, n3, h->stream)) return -1;
  HIP_TRY(hipGetLastError());
  h->gathered_valid = true;
  return 0;
}

int ba_get_points(ba_handle *h, double *X3, uint8_t *owned_mask) {
  if (!h || !h->finalized || !X3) return fail("ba_get_points: bad argument");
  if (h->gathered_valid) {  // after ba_gather_points: every point of the full problem
    std::memcpy(X3, h->gathered.data(), h->gathered.size() * sizeof(double));
    if (owned_mask) std::memset(owned_mask, 1, (size_t)h->plan.n_pt_global);
    return 0;
  }
  if (use_device(h) || pull_ctrl(h)) return -1;
  std::vector<double> buf;
  if (download(buf, h->d.pts[h->hc.cur], (size_t)h->plan.n_pt * 3, h->stream)) return -1;
  if (owned_mask) std::memset(owned_mask, 0, (size_t)h->plan.n_pt_global);
  for (int q = 0; q < h->plan.n_pt; ++q) {
    const int u = h->plan.pt_user_of_int[q];
    std::memcpy(X3 + (size_t)u * 3, &buf[(size_t)q * 3], 3 * sizeof(double));
    if (owned_mask) owned_mask[u] = 1;
  }
  return 0;
}

int ba_get_A(ba_handle *h, double *A36, double *a6) {
  if (!h || !h->finalized) return fail("ba_get_A: not finalized");
  if (use_device(h)) return -1;
  join_side(h);
  if (pull_ctrl(h)) return -1;  // synchronises the stream
  const int lb = h->hc.lcur;
  if (A36) {
    HIP_TRY(hipMemcpy(A36, h->d.A[lb], (size_t)h->plan.N * 36 * sizeof(double), hipMemcpyDeviceToHost));
    // the device keeps A_j undamped and scales the diagonal where it reads it
    // (reference :833-844): the same scaling here
    const double lp1 = 1.0 + h->hc.lambda;
    for (int j = 0; j < h->plan.N; ++j)
      for (int r = 0; r < 6; ++r) A36[(size_t)j * 36 + r * 7] *= lp1;
  }
  if (a6) HIP_TRY(hipMemcpy(a6, h->d.a[lb], (size_t)h->plan.N * 6 * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

static void expand_sym3(const double *s6, double *f9) {
  f9[0] = s6[0]; f9[1] = s6[1]; f9[2] = s6[2];
  f9[3] = s6[1]; f9[4] = s6[3]; f9[5] = s6[4];
  f9[6] = s6[2]; f9[7] = s6[4]; f9[8] = s6[5];
}

static int get_sym_vec(ba_handle *h, const double *dS6, const double *dV3,
                       double *out9, double *out3) {
  const ba::Plan &pl = h->plan;
  std::vector<double> s6, v3;
  if (download(s6, dS6, (size_t)pl.M * 6, h->stream)) return -1;
  if (download(v3, dV3, (size_t)pl.M * 3, h->stream)) return -1;
  for (int i = 0; i < pl.M; ++i) {
    const int g = pl.iopt_of_user[pl.pt_user_of_int[i]];
    if (out9) expand_sym3(&s6[(size_t)i * 6], out9 + (size_t)g * 9);
    if (out3) std::memcpy(out3 + (size_t)g * 3, &v3[(size_t)i * 3], 3 * sizeof(double));
  }
  return 0;
}

int ba_get_C(ba_handle *h, double *C9, double *b3) {
  if (!h || !h->finalized) return fail("ba_get_C: not finalized");
  if (use_device(h)) return -1;
  join_side(h);
  if (pull_ctrl(h)) return -1;
  ba::launch_damp_invert_export(h->d, h->stream);  // damped C_i of the current block buffer and lambda
  return get_sym_vec(h, h->d.Cd, h->d.b[h->hc.lcur], C9, b3);
}

int ba_get_Cinv(ba_handle *h, double *Cinv9, double *Cinvb3) {
  if (!h || !h->finalized) return fail("ba_get_Cinv: not finalized");
  if (use_device(h)) return -1;
  // Cinv_i b_i is not stored on the device (k_backsub_update forms it): same expression here
  const ba::Plan &pl = h->plan;
  std::vector<double> s6, b3;
  join_side(h);
  if (pull_ctrl(h)) return -1;
  ba::launch_damp_invert_export(h->d, h->stream);  // (the LM loop keeps Cinv in registers where it can)
  if (download(s6, h->d.Cinv, (size_t)pl.M * 6, h->stream)) return -1;
  if (download(b3, h->d.b[h->hc.lcur], (size_t)pl.M * 3, h->stream)) return -1;
  for (int i = 0; i < pl.M; ++i) {
    const int g = pl.iopt_of_user[pl.pt_user_of_int[i]];
    const double *ci = &s6[(size_t)i * 6], *b = &b3[(size_t)i * 3];
    if (Cinv9) expand_sym3(ci, Cinv9 + (size_t)g * 9);
    if (Cinvb3) {
      double *o = Cinvb3 + (size_t)g * 3;
      o[0] = ci[0] * b[0] + ci[1] * b[1] + ci[2] * b[2];
      o[1] = ci[1] * b[0] + ci[3] * b[1] + ci[4] * b[2];
      o[2] = ci[2] * b[0] + ci[4] * b[1] + ci[5] * b[2];
    }
  }
  return 0;
}

int ba_get_pairs(ba_handle *h, int32_t *pair_i, int32_t *pair_j, double *W18) {
  if (!h || !h->finalized) return fail("ba_get_pairs: not finalized");
  if (use_device(h)) return -1;
  const ba::Plan &pl = h->plan;
  auto padded = [&](int64_t p) { return !pl.pair_pad.empty() && pl.pair_pad[p]; };
  {
    int64_t o = 0;
    for (int64_t p = 0; p < pl.P; ++p) {
      if (padded(p)) continue;
      if (pair_i) pair_i[o] = pl.iopt_of_user[pl.pt_user_of_int[pl.pair_lm[p]]];
      if (pair_j) pair_j[o] = pl.pair_pose[p];
      ++o;
    }
  }
  if (W18) {
    if (pull_ctrl(h)) return -1;  // synchronises the stream
    // the device keeps B_ji compact ({K, X_ij}, ba_device.h kWStride): expand
    std::vector<double> w12((size_t)pl.P * ba::kWStride);
    if (pl.P > 0)
      HIP_TRY(hipMemcpy(w12.data(), h->d.W[h->hc.lcur], w12.size() * sizeof(double), hipMemcpyDeviceToHost));
    int64_t o = 0;
    for (int64_t p = 0; p < pl.P; ++p) {
      if (padded(p)) continue;
      const double *k = &w12[(size_t)p * ba::kWStride];
      double *W = W18 + (size_t)o * 18;
      for (int e = 0; e < 9; ++e) W[e] = k[e];
      for (int c = 0; c < 3; ++c) {
        W[9 + c] = k[10] * k[6 + c] - k[11] * k[3 + c];
        W[12 + c] = k[11] * k[c] - k[9] * k[6 + c];
        W[15 + c] = k[9] * k[3 + c] - k[10] * k[c];
      }
      ++o;
    }
  }
  return 0;
}

int ba_get_S(ba_handle *h, double *S, double *rhs) {
  if (!h || !h->finalized) return fail("ba_get_S: not finalized");
  if (use_device(h)) return -1;
  const ba::DevProblem &d = h->d;
  const int n6 = 6 * h->plan.N;
  std::vector<double> L;
  ba::launch_scatter(d, h->stream);  // from the packed exchange buffer
  if (download(L, d.L, (size_t)d.npad * d.ld, h->stream)) return -1;
  auto colof = [&](int e) { return h->pose_col_h[e / 6] + e % 6; };
  for (int c = 0; c < n6; ++c) {
    for (int r = c; r < n6; ++r) {
      int rr = colof(r), cc = colof(c);
      if (rr < cc) std::swap(rr, cc);
      const double v = L[(size_t)cc * d.ld + rr];
      if (S) {
        S[(size_t)r * n6 + c] = v;
        S[(size_t)c * n6 + r] = v;
      }
    }
    if (rhs) rhs[c] = L[(size_t)colof(c) * d.ld + d.npad];
  }
  return 0;
}

int ba_get_xy(ba_handle *h, double *x6, double *y3) {
  if (!h || !h->finalized) return fail("ba_get_xy: not finalized");
  if (use_device(h)) return -1;
  const ba::Plan &pl = h->plan;
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (x6 && pl.N > 0)
    HIP_TRY(hipMemcpy(x6, h->d.x, (size_t)pl.N * 6 * sizeof(double), hipMemcpyDeviceToHost));
  if (y3) {
    std::vector<double> y;
    if (download(y, h->d.y, (size_t)pl.M * 3, h->stream)) return -1;
    for (int i = 0; i < pl.M; ++i) {
      const int g = pl.iopt_of_user[pl.pt_user_of_int[i]];
      std::memcpy(y3 + (size_t)g * 3, &y[(size_t)i * 3], 3 * sizeof(double));
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------
int ba_kernel_count(void) { return ba::K_COUNT; }

const char *ba_kernel_name(int id) {
  static const char *names[ba::K_COUNT] = {
      "k_cost", "k_lin_landmarks", "k_lin_poses", "k_pose_finalize", "k_dense_init",
      "k_schur_lds", "k_schur_partial", "k_schur_final", "k_scatter",
      "k_chol_diag", "k_chol_trsm", "k_chol_update", "k_chol_back", "k_chol_level", "k_chol_diag_trsm", "k_chol_tail", "k_backsub_update",
      "k_pose_update", "k_scalars", "k_control", "k_damp_invert", "k_schur_grp", "k_lin_grp", "k_backsub_lin"};
  return (id >= 0 && id < ba::K_COUNT) ? names[id] : "";
}

int ba_get_kernel_ms(ba_handle *h, double *ms_out, int64_t *calls_out, int reset) {
  if (!h || !ms_out || !calls_out) return fail("ba_get_kernel_ms: bad argument");
  for (int k = 0; k < ba::K_COUNT; ++k) {
    ms_out[k] = h->kt.ms[k];
    calls_out[k] = h->kt.calls[k];
  }
  if (reset) h->kt.reset();
  return 0;
}

int ba_get_dense_info(ba_handle *h, double out4[4]) {
  if (!h || !h->finalized || !out4) return fail("ba_get_dense_info: bad argument");
  out4[0] = h->sched.fill;
  out4[1] = h->sched.flops;
  out4[2] = (double)h->sched.nlev;
  out4[3] = (double)h->d.npad;
  return 0;
}

int ba_get_schur_info(ba_handle *h, int64_t out8[8]) {
  if (!h || !h->finalized || !out8) return fail("ba_get_schur_info: bad argument");
  const ba::Plan &pl = h->plan;
  int64_t pairs = 0, triples = 0, mfma = 0;
  for (const auto *list : {&pl.grp32, &pl.grp64, &pl.grp128})
    for (const auto &g : *list) {
      pairs += (int64_t)g.nl * g.d;
      triples += (int64_t)g.nl * g.d * (g.d + 1) / 2;
      // v_mfma_f64_16x16x4 instructions of k_schur_grp: chunks of nlw landmarks,
      // ceil(3 nlc / 4) k steps each, NT (NT + 1) / 2 tiles per step
      const int nt = list == &pl.grp32 ? 2 : (list == &pl.grp64 ? 4 : 8), krw = nt == 2 ? 36 : (nt == 4 ? 20 : 8);
      const int nlw = std::min(64 / g.d, krw / 3);
      for (int c0 = 0; c0 < g.nl; c0 += nlw)
        mfma += (int64_t)((3 * std::min(nlw, g.nl - c0) + 3) / 4) * (nt * (nt + 1) / 2);
    }
  out8[0] = (int64_t)pl.grp32.size();
  out8[1] = (int64_t)(pl.grp64.size() + pl.grp128.size());  // (64- and 128-wide images)
  out8[2] = pl.M_grp;
  out8[3] = (int64_t)pl.sup_desc.size();
  out8[4] = pairs;
  out8[5] = triples;
  out8[6] = mfma;
  out8[7] = (int64_t)pl.tri_p.size();
  return 0;
}

int ba_get_lin_info(ba_handle *h, int64_t out4[4]) {
  if (!h || !h->finalized || !out4) return fail("ba_get_lin_info: bad argument");
  const ba::Plan &pl = h->plan;
  int64_t obs = 0;
  for (const auto &g : pl.lin_desc) obs += (int64_t)g.nl * g.no;
  out4[0] = (int64_t)pl.lin_desc.size();
  out4[1] = obs;
  out4[2] = (int64_t)pl.bchunk_lm.size() - 1 - (pl.lin_groups ? pl.n_bchunk_grp : 0);
  out4[3] = pl.n_pobs;
  return 0;
}

int ba_get_mask_info(ba_handle *h, int64_t out4[4]) {
  if (!h || !h->finalized || !out4) return fail("ba_get_mask_info: bad argument");
  const ba::Plan &pl = h->plan;
  int64_t lm = 0;
  for (const auto &g : pl.grp_range) lm += g.masked ? g.nl : 0;
  out4[0] = (int64_t)pl.lin_desc.size() - pl.n_lin_plain;
  out4[1] = lm;
  out4[2] = pl.n_obs - (pl.n_obs_global > 0 && h->world == 1 ? pl.n_obs_global : pl.n_obs);
  {  // (sharded: count the padded slots themselves)
    int64_t pad = 0;
    for (int64_t s2 = 0; s2 < pl.n_obs; ++s2) pad += !(pl.obs_uv[2 * s2] == pl.obs_uv[2 * s2]);
    out4[2] = pad;
  }
  out4[3] = pl.n_pair_pad;
  return 0;
}

int ba_get_dropped_pivots(ba_handle *h, int64_t *count, int reset) {
  if (!h || !h->finalized || !count) return fail("ba_get_dropped_pivots: bad argument");
  if (use_device(h)) return -1;
  int v = 0;
  HIP_TRY(hipMemcpyAsync(&v, h->ddev.bad_pivots, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (reset) HIP_TRY(hipMemsetAsync(h->ddev.bad_pivots, 0, sizeof(int), h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  *count = v % ba::kFlowTimeout;  // (the upper bits count timed-out hand-offs: ba_lm_sync reports those)
  return 0;
}

int ba_dense_spd_solve(ba_handle *h, int n, const double *A, const double *b,
                       double *x, double *ms) {
  if (!h || n <= 0 || !A || !b || !x) return fail("ba_dense_spd_solve: bad argument");
  if (use_device(h)) return -1;
  const int nb = 64;  // a dense matrix has one tile per level either way: fewer, larger tiles
  const int ncb = (n + nb - 1) / nb;
  const int npad = ncb * nb;
  const int ld = npad + nb;
  // tile pattern of A -> level schedule -> symmetric tile permutation
  std::vector<uint8_t> adj((size_t)ncb * ncb, 0);
  for (int I = 0; I < ncb; ++I)
    for (int J = 0; J < I; ++J) {
      bool any = false;
      for (int r = I * nb; r < std::min(n, (I + 1) * nb) && !any; ++r)
        for (int c = J * nb; c < (J + 1) * nb && !any; ++c) any = A[(size_t)r * n + c] != 0.0;
      adj[(size_t)I * ncb + J] = adj[(size_t)J * ncb + I] = any;
    }
  ba::DenseSchedule sc;
  const char *nat = getenv("BA_DENSE_NATURAL");
  ba::build_dense_schedule(ncb, adj, nat && atoi(nat) != 0, nb, sc);
  std::vector<int> colmap(npad), col_x(npad, -1);  // original column -> dense column
  for (int c = 0; c < npad; ++c) colmap[c] = sc.pos_of_tile[c / nb] * nb + c % nb;
  std::vector<double> L((size_t)npad * ld, 0.0);
  for (int c = 0; c < npad; ++c) {
    if (c < n) {
      col_x[colmap[c]] = c;
      for (int r = c; r < n; ++r) {
        int rr = colmap[r], cc = colmap[c];
        if (rr < cc) std::swap(rr, cc);
        L[(size_t)cc * ld + rr] = A[(size_t)r * n + c];
      }
      L[(size_t)colmap[c] * ld + npad] = b[c];
    } else {
      L[(size_t)colmap[c] * ld + colmap[c]] = 1.0;
    }
  }
  double *dL = nullptr, *dD = nullptr, *dx = nullptr;
  ba::DenseDev dd;
  dd.read_env();
  dd.aux_stream = h->side_stream;
  if (!h->ev_look[0]) {
    for (int k = 0; k < 3; ++k) HIP_TRY(hipEventCreateWithFlags(&h->ev_look[k], hipEventDisableTiming));
  }
  dd.ev_m = h->ev_look[0];
  dd.ev_x[0] = h->ev_look[1];
  dd.ev_x[1] = h->ev_look[2];
  auto up = [&](int **p, const std::vector<int> &v) -> int {
    HIP_TRY(hipMalloc((void **)p, std::max<size_t>(1, v.size()) * sizeof(int)));
    if (!v.empty()) HIP_TRY(hipMemcpy(*p, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
    return 0;
  };
  if (up(&dd.row_ptr, sc.row_ptr) || up(&dd.rows, sc.rows) || up(&dd.item_t, sc.item_t) ||
      up(&dd.item_I, sc.item_I) || up(&dd.tgt_I, sc.tgt_I) || up(&dd.tgt_J, sc.tgt_J) ||
      up(&dd.tgt_src_ptr, sc.tgt_src_ptr) || up(&dd.src_t, sc.src_t) || up(&dd.col_x, col_x) ||
      up(&dd.tgt_desc, sc.tgt_desc) || up(&dd.back_desc, sc.back_desc) || up(&dd.row_desc, sc.row_desc))
    return -1;
  if (sc.fused_ok) {
    if (up(&dd.f_desc, sc.f_desc) || up(&dd.f_pend, sc.f_pend)) return -1;
    HIP_TRY(hipMalloc((void **)&dd.cbuf, (size_t)std::max(1, sc.n_contrib) * nb * nb * sizeof(double)));
  }
  HIP_TRY(hipMalloc((void **)&dd.xc, (size_t)npad * sizeof(double)));
  {  // the dataflow backward sweep of the LM path (ordered form for dense patterns)
    std::vector<int> order;
    dd.flow_tail_t0 = ba::dense_flow_order(sc, dd, order);
    dd.n_flow = (int)order.size();
    dd.flow_gen = 0;
    if (up(&dd.flow_order, order)) return -1;
    HIP_TRY(hipMalloc((void **)&dd.flow_flags, (size_t)std::max(1, ncb) * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&dd.flow_ticket, sizeof(int)));
    HIP_TRY(hipMemset(dd.flow_flags, 0, (size_t)std::max(1, ncb) * sizeof(int)));
    HIP_TRY(hipMemset(dd.flow_ticket, 0, sizeof(int)));
    HIP_TRY(hipMalloc((void **)&dd.bad_pivots, sizeof(int)));
    HIP_TRY(hipMemset(dd.bad_pivots, 0, sizeof(int)));
    // the dataflow forward sweep (narrow patterns only: see dense_fwd_items)
    HIP_TRY(hipMalloc((void **)&dd.fwd_flags, (size_t)std::max(1, ncb) * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&dd.fwd_ticket, sizeof(int)));
    HIP_TRY(hipMemset(dd.fwd_flags, 0, (size_t)std::max(1, ncb) * sizeof(int)));
    HIP_TRY(hipMemset(dd.fwd_ticket, 0, sizeof(int)));
    std::vector<int> items, pre, need;
    if (ba::dense_fwd_items(sc, dd, items, pre, need)) {
      dd.n_fwd_items = (int)items.size() / 2;
      dd.n_fwd_cnt = (int)need.size();
      if (up(&dd.fwd_items, items) || up(&dd.upd_pre, pre) || up(&dd.col_need, need)) return -1;
      HIP_TRY(hipMalloc((void **)&dd.fwd_cnt, need.size() * sizeof(int)));
      HIP_TRY(hipMemset(dd.fwd_cnt, 0, need.size() * sizeof(int)));
    }
    std::vector<int> ntrsm, lneed;
    if (ba::dense_dag_items(sc, dd, items, pre, need, ntrsm, lneed)) {
      dd.n_dag_items = (int)items.size() / 2;
      dd.n_fwd_cnt = (int)need.size();
      if (up(&dd.dag_items, items) || up(&dd.upd_pre, pre) || up(&dd.col_need, need) || up(&dd.dag_ntrsm, ntrsm) ||
          up(&dd.look_need, lneed))
        return -1;
      for (int **q : {&dd.fwd_cnt, &dd.dag_dflags, &dd.dag_tcnt}) {
        HIP_TRY(hipMalloc((void **)q, need.size() * sizeof(int)));
        HIP_TRY(hipMemset(*q, 0, need.size() * sizeof(int)));
      }
    }
  }
  HIP_TRY(hipMalloc((void **)&dL, L.size() * sizeof(double)));
  HIP_TRY(hipMalloc((void **)&dD, (size_t)ncb * ba::dense_ws_per_block(nb) * sizeof(double)));
  HIP_TRY(hipMalloc((void **)&dx, (size_t)npad * sizeof(double)));
  HIP_TRY(hipMemcpy(dL, L.data(), L.size() * sizeof(double), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipEventRecord(e0, h->stream));
  ba::dense_factor_solve(dL, npad, ld, dD, dx, nullptr, sc, dd, h->stream);
  HIP_TRY(hipEventRecord(e1, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  float t = 0.f;
  (void)hipEventElapsedTime(&t, e0, e1);
  if (ms) *ms = t;
  HIP_TRY(hipMemcpy(x, dx, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  int bad_h = 0;
  HIP_TRY(hipMemcpy(&bad_h, dd.bad_pivots, sizeof(int), hipMemcpyDeviceToHost));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for (void *p : {(void *)dL, (void *)dD, (void *)dx, (void *)dd.xc, (void *)dd.row_ptr,
                  (void *)dd.rows, (void *)dd.item_t, (void *)dd.item_I, (void *)dd.tgt_I,
                  (void *)dd.tgt_J, (void *)dd.tgt_src_ptr, (void *)dd.src_t, (void *)dd.col_x,
                  (void *)dd.tgt_desc, (void *)dd.back_desc, (void *)dd.row_desc, (void *)dd.f_desc,
                  (void *)dd.f_pend, (void *)dd.cbuf, (void *)dd.flow_order, (void *)dd.flow_flags,
                  (void *)dd.flow_ticket, (void *)dd.bad_pivots, (void *)dd.fwd_flags,
                  (void *)dd.fwd_ticket, (void *)dd.fwd_items, (void *)dd.upd_pre, (void *)dd.col_need,
                  (void *)dd.fwd_cnt, (void *)dd.dag_items, (void *)dd.dag_ntrsm, (void *)dd.dag_dflags,
                  (void *)dd.dag_tcnt, (void *)dd.look_need})
    (void)hipFree(p);
  HIP_TRY(hipGetLastError());
  if (bad_h >= ba::kFlowTimeout) return fail("ba_dense_spd_solve: a dataflow hand-off timed out");
  return 0;
}

}  // extern "C" (reopened below)

// One call = one H2D copy, one kernel, one D2H copy, all through ONE pinned
// staging buffer that mirrors ONE device buffer (ten small pageable copies
// cost more than the kernel at 10 k points):
//   [X | uv | uv_right | right camera | T | mask | mask_right | barrier words ]  <- H2D
//                                     [ T | mask | mask_right | barrier words | meta | iters | debug poses ]  <- D2H
namespace {
struct PoLayout {
  size_t X, uv, uvr, camr, T, mask, maskr, gsw, h2d_end, meta, iters, dbg, end;
};
PoLayout po_layout(int n, int icap, bool stereo) {
  auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
  PoLayout L;
  size_t o = 0;
  L.X = o;      o = al(o + (size_t)n * 3 * sizeof(float));
  L.uv = o;     o = al(o + (size_t)n * 2 * sizeof(float));
  L.uvr = o;    o = al(o + (stereo ? (size_t)n * 2 * sizeof(float) : 0));
  L.camr = o;   o = al(o + 16 * sizeof(float));
  L.T = o;      o = al(o + 12 * sizeof(float));
  L.mask = o;   o = al(o + (size_t)n);
  L.maskr = o;  o = al(o + (stereo ? (size_t)n : 0));
  L.gsw = o;    o = al(o + sizeof(int) * (size_t)ba::pose_only_sync_ints());
  L.h2d_end = o;
  L.meta = o;   o = al(o + 4 * sizeof(int));
  L.iters = o;  o = al(o + (size_t)icap * sizeof(ba::PoIter));
  L.dbg = o;    o = al(o + (size_t)icap * 12 * sizeof(float));
  L.end = o;
  return L;
}

int po_run(ba_handle *h, bool stereo, const float *X3, const float *uv2, const float *uvr2, int n,
           float fx, float fy, float cx, float cy, const float *camr16, float *T12, uint8_t *mask,
           uint8_t *mask_r, const ba_options *opt, ba_po_iter *iters, int cap, int *n_iter,
           int *converged, float *debug_T12) {
  if (use_device(h)) return -1;
  static_assert(sizeof(ba::PoIter) == sizeof(ba_po_iter), "po iter layout");
  const int max_it = opt->max_num_iterations;
  const int icap = std::max(1, std::max(cap, max_it));
  const PoLayout L = po_layout(n, icap, stereo);
  if (L.end > h->po_cap) {  // grow the buffer pair (not in `allocs`: freed here and in free_device)
    if (h->po_dev) (void)hipFree(h->po_dev);
    if (h->po_host) (void)hipHostFree(h->po_host);
    h->po_dev = h->po_host = nullptr;
    h->po_cap = 0;
    HIP_TRY(hipMalloc((void **)&h->po_dev, L.end));
    HIP_TRY(hipHostMalloc((void **)&h->po_host, L.end, hipHostMallocDefault));
    h->po_cap = L.end;
  }
  if (!h->po_part && h->dalloc(&h->po_part, (size_t)ba::pose_only_partial_floats())) return -1;
  uint8_t *hb = h->po_host, *db = h->po_dev;
  std::memcpy(hb + L.X, X3, (size_t)n * 3 * sizeof(float));
  std::memcpy(hb + L.uv, uv2, (size_t)n * 2 * sizeof(float));
  if (stereo) {
    std::memcpy(hb + L.uvr, uvr2, (size_t)n * 2 * sizeof(float));
    std::memcpy(hb + L.camr, camr16, 16 * sizeof(float));
    std::memcpy(hb + L.maskr, mask_r, (size_t)n);
  }
  std::memcpy(hb + L.T, T12, 12 * sizeof(float));
  std::memcpy(hb + L.mask, mask, (size_t)n);
  std::memset(hb + L.gsw, 0, sizeof(int) * (size_t)ba::pose_only_sync_ints());
  hipStream_t s = h->stream;
  HIP_TRY(hipMemcpyAsync(db, hb, L.h2d_end, hipMemcpyHostToDevice, s));
  float *dT = (float *)(db + L.T), *ddbg = debug_T12 ? (float *)(db + L.dbg) : nullptr;
  int rc;
  if (stereo)
    rc = ba::pose_only_stereo6_device(
        (const float *)(db + L.X), (const float *)(db + L.uv), (const float *)(db + L.uvr), n, fx, fy,
        cx, cy, (const float *)(db + L.camr), dT, db + L.mask, db + L.maskr, opt->threshold_huber_loss,
        opt->threshold_step_size, opt->threshold_cost_change, opt->threshold_outlier_rejection, max_it,
        (ba::PoIter *)(db + L.iters), icap, (int *)(db + L.meta), ddbg, (int *)(db + L.gsw), h->po_part, s);
  else
    rc = ba::pose_only_mono6_device(
        (const float *)(db + L.X), (const float *)(db + L.uv), n, fx, fy, cx, cy, dT, db + L.mask,
        opt->threshold_huber_loss, opt->threshold_step_size, opt->threshold_cost_change,
        opt->threshold_outlier_rejection, max_it, (ba::PoIter *)(db + L.iters), icap,
        (int *)(db + L.meta), ddbg, (int *)(db + L.gsw), h->po_part, s);
  if (rc) return fail("pose-only kernel launch failed");
  HIP_TRY(hipMemcpyAsync(hb + L.T, db + L.T, L.end - L.T, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  int meta[4], gsw[2];
  std::memcpy(meta, hb + L.meta, sizeof(meta));
  std::memcpy(gsw, hb + L.gsw, sizeof(gsw));
  if (gsw[1]) return fail("pose-only kernel: grid barrier timed out");
  std::memcpy(T12, hb + L.T, 12 * sizeof(float));
  std::memcpy(mask, hb + L.mask, (size_t)n);
  if (stereo) std::memcpy(mask_r, hb + L.maskr, (size_t)n);
  if (max_it <= 0) { meta[0] = 0; meta[1] = 1; meta[2] = 0; meta[3] = 1; }
  const int rows = std::min(meta[2], cap);
  if (iters && rows > 0) std::memcpy(iters, hb + L.iters, (size_t)rows * sizeof(ba_po_iter));
  if (debug_T12 && meta[0] > 0)
    std::memcpy(debug_T12, hb + L.dbg, (size_t)std::min(meta[0], cap) * 12 * sizeof(float));
  if (n_iter) *n_iter = meta[0];
  if (converged) *converged = meta[1];
  return meta[3] ? 0 : 1;  // 1 = NaN pose, input left unchanged (reference :159-167)
}
}  // namespace

extern "C" {

int ba_pose_only_mono6(ba_handle *h, const float *X3, const float *uv2, int n,
                       float fx, float fy, float cx, float cy, float *T12,
                       uint8_t *mask, const ba_options *opt, ba_po_iter *iters,
                       int cap, int *n_iter, int *converged,
                       float *debug_T12) {
  if (!h || !X3 || !uv2 || n <= 0 || !T12 || !mask || !opt)
    return fail("ba_pose_only_mono6: bad argument");
  return po_run(h, false, X3, uv2, nullptr, n, fx, fy, cx, cy, nullptr, T12, mask, nullptr, opt,
                iters, cap, n_iter, converged, debug_T12);
}

int ba_pose_only_stereo6(ba_handle *h, const float *X3, const float *uv2,
                         const float *uvr2, int n, const float *intr_l4,
                         const float *intr_r4, const float *T_lr12, float *T12,
                         uint8_t *mask, uint8_t *mask_r, const ba_options *opt,
                         ba_po_iter *iters, int cap, int *n_iter, int *converged,
                         float *debug_T12) {
  if (!h || !X3 || !uv2 || !uvr2 || n <= 0 || !intr_l4 || !intr_r4 || !T_lr12 || !T12 ||
      !mask || !mask_r || !opt)
    return fail("ba_pose_only_stereo6: bad argument");
  // right camera record: fx fy cx cy, then pose_right_to_left = left_to_right^-1
  // (reference :226) as R (9, row-major) and t (3)
  float camr[16];
  for (int k = 0; k < 4; ++k) camr[k] = intr_r4[k];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) camr[4 + r * 3 + c] = T_lr12[c * 3 + r];
  for (int r = 0; r < 3; ++r)
    camr[13 + r] = -(camr[4 + r * 3 + 0] * T_lr12[9] + camr[4 + r * 3 + 1] * T_lr12[10] +
                     camr[4 + r * 3 + 2] * T_lr12[11]);
  return po_run(h, true, X3, uv2, uvr2, n, intr_l4[0], intr_l4[1], intr_l4[2], intr_l4[3], camr, T12,
                mask, mask_r, opt, iters, cap, n_iter, converged, debug_T12);
}

}  // extern "C"
