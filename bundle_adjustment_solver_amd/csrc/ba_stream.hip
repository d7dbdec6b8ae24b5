// ba_stream.hip — observation streaming: full BA of a problem whose landmark-side
// data (observations, W = B_ji, C_i, b_i, points) does not have to fit in device
// memory (include/ba_hip.h "observation streaming").
//
// The landmarks are cut into K chunks with the sharding rule of the multi-GPU path
// (ba_set_shard: contiguous ranges of the locality order, balanced by observation
// count); every chunk is a shard handle whose chunk-sized arrays are carved from a
// shared DEVICE ARENA instead of hipMalloc (ba_handle.h: kind 1 = structure /
// observations, restored only; kind 2 = blocks / points / partial sums, saved and
// restored).  Two arenas ping-pong: while the kernels of chunk k run in one, the
// copy stream saves the previous occupant of the other and restores chunk k + 1
// into it from its pinned host image.  Everything pose-sized (poses, A_j, a_j, the
// controller, the packed partial S||rhs) stays resident per chunk; the dense image,
// its schedule and x exist ONCE (the first chunk owns them: ba_handle::dense_owner).
//
// One LM iteration = the sharded iteration of ba_api.hip with the two all-reduces
// replaced by sums over the chunks visited one after the other:
//   A  for k = 0 .. K-1:   damp / invert, Schur accumulation, k_schur_final -> packed
//                          partial S||rhs of chunk k, added to chunk 0's (fixed order)
//      scatter, factorise, solve                                    (once)
//   B  for k = K-1 .. 0:   back-substitution + update, linearisation at the trial
//                          point, LM scalars of chunk k, added up (fixed order)
//   C  for every k:        control step with the summed scalars (replicated, tiny)
// B runs in reverse so that the two chunks resident at the end of A are used
// without a transfer, and so does the next A after B.  The trust-region decision
// needs the cost of ALL chunks before the next Schur accumulation can damp with the
// new lambda, so every chunk's data crosses PCIe twice per iteration: the loop is
// PCIe-bound by construction (DESIGN.md §6b gives the measured rate).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ba_handle.h"

using ba::fail;

struct ba_stream {
  int device = 0, K = 1;
  int64_t arena_bytes = 0;
  hipStream_t s_comp = nullptr, s_copy = nullptr;
  ba::DeviceArena arena[2];
  std::vector<ba_handle *> h;
  std::vector<char *> img;          // pinned host image of chunk k: [kind 1 | kind 2]
  std::vector<hipEvent_t> ev_in, ev_comp;
  std::vector<uint8_t> in_pending, dirty, computed;
  int resident[2] = {-1, -1};
  double *scal_sum = nullptr;
  bool finalized = false, lm_begun = false;
  int64_t bytes_h2d = 0, bytes_d2h = 0;
  // problem (host, until finalize)
  int n_cam = 0, n_pose = 0, n_pt = 0;
  int64_t n_obs = 0;
  std::vector<double> cam_intr, cam_T, pose_T, pt_X, obs_uv;
  std::vector<uint8_t> pose_fixed, pt_fixed;
  std::vector<int32_t> obs_cam, obs_pose, obs_pt;
};

namespace {

__global__ void k_add_into(double *__restrict__ dst, const double *__restrict__ src, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

char *mut_base(ba_stream *s, int k) {
  const ba::DeviceArena &a = s->arena[k & 1];
  return a.base + a.cap - s->h[k]->arena_mut;
}

// Save chunk j's mutable region (after its kernels) — copy stream.
int save(ba_stream *s, int j) {
  ba_handle *hj = s->h[j];
  if (s->computed[j]) HIP_TRY(hipStreamWaitEvent(s->s_copy, s->ev_comp[j], 0));
  if (s->dirty[j] && hj->arena_mut > 0) {
    HIP_TRY(hipMemcpyAsync(s->img[j] + hj->arena_imm, mut_base(s, j), hj->arena_mut, hipMemcpyDeviceToHost, s->s_copy));
    s->bytes_d2h += (int64_t)hj->arena_mut;
  }
  s->dirty[j] = 0;
  return 0;
}

// Make chunk k resident in its arena (copy stream); the compute stream is NOT made
// to wait here (prefetch) — acquire() does that.
int prefetch(ba_stream *s, int k) {
  if (k < 0 || k >= s->K) return 0;
  const int a = k & 1;
  if (s->resident[a] == k) return 0;
  const int j = s->resident[a];
  if (j >= 0 && save(s, j)) return -1;
  ba_handle *hk = s->h[k];
  if (hk->arena_imm > 0)
    HIP_TRY(hipMemcpyAsync(s->arena[a].base, s->img[k], hk->arena_imm, hipMemcpyHostToDevice, s->s_copy));
  if (hk->arena_mut > 0)
    HIP_TRY(hipMemcpyAsync(mut_base(s, k), s->img[k] + hk->arena_imm, hk->arena_mut, hipMemcpyHostToDevice, s->s_copy));
  s->bytes_h2d += (int64_t)(hk->arena_imm + hk->arena_mut);
  HIP_TRY(hipEventRecord(s->ev_in[k], s->s_copy));
  s->in_pending[k] = 1;
  s->resident[a] = k;
  return 0;
}

// BA_STREAM_SYNC=1 (developer knob): a device synchronisation after every transfer and
// every chunk's kernels — no overlap; separates ordering bugs from logic bugs
bool sync_mode() {
  static const bool on = getenv("BA_STREAM_SYNC") && getenv("BA_STREAM_SYNC")[0] == '1';
  return on;
}

int acquire(ba_stream *s, int k) {
  if (prefetch(s, k)) return -1;
  if (s->in_pending[k]) {
    HIP_TRY(hipStreamWaitEvent(s->s_comp, s->ev_in[k], 0));
    s->in_pending[k] = 0;
  }
  if (sync_mode()) HIP_TRY(hipDeviceSynchronize());
  return 0;
}

// the kernels of chunk k have been enqueued: its arena may be reused after them
int release(ba_stream *s, int k, bool wrote) {
  HIP_TRY(hipEventRecord(s->ev_comp[k], s->s_comp));
  s->computed[k] = 1;
  if (wrote) s->dirty[k] = 1;
  if (sync_mode()) HIP_TRY(hipDeviceSynchronize());
  return 0;
}

void linearize(ba_handle *h, int sel, hipStream_t st) {
  const ba::DevProblem &d = h->d;
  ba::launch_lin_landmarks(d, sel, st);
  ba::launch_lin_poses(d, sel, st);
  if (d.n_obs_lm < d.n_obs) ba::launch_cost(d, sel ? 2 : 0, d.n_obs_lm, st);
}

int add_into(double *dst, const double *src, int64_t n, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(k_add_into, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, src, n);
  return 0;
}

int sum_scalars_to_all(ba_stream *s) {
  for (int k = 0; k < s->K; ++k)
    HIP_TRY(hipMemcpyAsync(s->h[k]->d.scal, s->scal_sum, 4 * sizeof(double), hipMemcpyDeviceToDevice, s->s_comp));
  return 0;
}

int stream_begin(ba_stream *s, const ba_options *opt, int *done_after) {
  HIP_TRY(hipMemsetAsync(s->scal_sum, 0, 4 * sizeof(double), s->s_comp));
  for (int k = 0; k < s->K; ++k) {
    ba_handle *h = s->h[k];
    int da = 0;
    if (ba::lm_prepare_ctrl(h, opt, &da)) return -1;
    *done_after = da;
    if (acquire(s, k)) return -1;
    if (prefetch(s, k + 1)) return -1;
    linearize(h, 0, s->s_comp);
    ba::launch_scalars_cost_only(h->d, 1, s->s_comp);
    add_into(s->scal_sum, h->d.scal, 4, s->s_comp);
    if (release(s, k, true)) return -1;
  }
  if (sum_scalars_to_all(s)) return -1;
  for (int k = 0; k < s->K; ++k) ba::launch_init_ctrl_cost(s->h[k]->d, s->s_comp);
  HIP_TRY(hipGetLastError());
  return 0;
}

// direction of the chunk loops: the order in which the chunks were left resident
int stream_iteration(ba_stream *s) {
  const int K = s->K;
  ba_handle *own = s->h[0];
  // ---- A: partial reduced systems, ascending, summed into chunk 0's packed buffer ----
  ba::launch_dense_init(own->d.L, own->d.ld, own->d.col_x, own->d.zt_I, own->d.zt_J, own->d.n_zt, own->d.nb,
                        &own->d.ctrl->done, s->s_comp);
  for (int k = 0; k < K; ++k) {
    ba_handle *h = s->h[k];
    if (acquire(s, k)) return -1;
    if (prefetch(s, k + 1)) return -1;
    ba::launch_damp_invert(h->d, s->s_comp);
    ba::launch_schur_accumulate(h->d, s->s_comp);
    ba::launch_schur_final(h->d, /*direct=*/false, s->s_comp);
    if (k > 0) add_into(own->d.Spk, h->d.Spk, h->xbuf_n[0], s->s_comp);
    if (release(s, k, false)) return -1;  // (Cinv / slot partials are scratch: nothing to save)
  }
  ba::launch_scatter(own->d, s->s_comp);
  own->ddev.flow_ok = true;
  ba::launch_dense_solve(own->d, own->sched, own->ddev, s->s_comp);
  // ---- B: back-substitution, update, trial-point linearisation, LM scalars: descending ----
  HIP_TRY(hipMemsetAsync(s->scal_sum, 0, 4 * sizeof(double), s->s_comp));
  for (int k = K - 1; k >= 0; --k) {
    ba_handle *h = s->h[k];
    if (acquire(s, k)) return -1;
    if (prefetch(s, k - 1)) return -1;
    ba::launch_backsub_update(h->d, s->s_comp);
    linearize(h, 1, s->s_comp);
    ba::launch_scalars(h->d, 1, s->s_comp);
    add_into(s->scal_sum, h->d.scal, 4, s->s_comp);
    if (release(s, k, true)) return -1;
  }
  // ---- C: the trust-region decision, replicated on every chunk's controller ----
  if (sum_scalars_to_all(s)) return -1;
  for (int k = 0; k < K; ++k) ba::launch_control(s->h[k]->d, s->s_comp);
  HIP_TRY(hipGetLastError());
  return 0;
}

int flush(ba_stream *s) {  // every chunk's mutable data to its host image
  for (int a = 0; a < 2; ++a)
    if (s->resident[a] >= 0 && save(s, s->resident[a])) return -1;
  HIP_TRY(hipStreamSynchronize(s->s_copy));
  HIP_TRY(hipStreamSynchronize(s->s_comp));
  return 0;
}

}  // namespace

extern "C" {

int ba_stream_create(ba_stream **out, int device_id, int n_chunks, int64_t arena_bytes) {
  if (!out) return fail("ba_stream_create: null out pointer");
  *out = nullptr;
  if (n_chunks < 1 || arena_bytes < (1 << 20)) return fail("ba_stream_create: bad chunk count / arena size");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail("ba_stream_create: no HIP device available (the HIP path has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail("ba_stream_create: bad device id");
  ba_stream *s = new ba_stream();
  s->device = device_id;
  s->K = n_chunks;
  s->arena_bytes = (arena_bytes + 255) & ~(int64_t)255;
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&s->s_comp) != hipSuccess ||
      hipStreamCreateWithFlags(&s->s_copy, hipStreamNonBlocking) != hipSuccess) {
    delete s;
    return fail("ba_stream_create: cannot create streams");
  }
  *out = s;
  return 0;
}

void ba_stream_destroy(ba_stream *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->s_comp) (void)hipStreamSynchronize(s->s_comp);
  if (s->s_copy) (void)hipStreamSynchronize(s->s_copy);
  // the chunks alias the owner's dense image: destroy them first
  for (int k = (int)s->h.size() - 1; k >= 0; --k) ba_destroy(s->h[k]);
  for (char *p : s->img)
    if (p) (void)hipHostFree(p);
  for (hipEvent_t e : s->ev_in) (void)hipEventDestroy(e);
  for (hipEvent_t e : s->ev_comp) (void)hipEventDestroy(e);
  if (s->arena[0].base) (void)hipFree(s->arena[0].base);
  if (s->arena[1].base && s->arena[1].base != s->arena[0].base) (void)hipFree(s->arena[1].base);
  if (s->scal_sum) (void)hipFree(s->scal_sum);
  if (s->s_copy) (void)hipStreamDestroy(s->s_copy);
  if (s->s_comp) (void)hipStreamDestroy(s->s_comp);
  delete s;
}

int ba_stream_set_cameras(ba_stream *s, int n_cam, const double *intr4, const double *T_cj12) {
  if (!s || n_cam <= 0 || !intr4 || !T_cj12) return fail("ba_stream_set_cameras: bad argument");
  if (s->finalized) return fail("ba_stream_set_cameras: already finalized");
  s->n_cam = n_cam;
  s->cam_intr.assign(intr4, intr4 + 4 * (size_t)n_cam);
  s->cam_T.assign(T_cj12, T_cj12 + 12 * (size_t)n_cam);
  return 0;
}
int ba_stream_set_poses(ba_stream *s, int n_pose, const double *T_jw12, const uint8_t *fixed) {
  if (!s || n_pose <= 0 || !T_jw12) return fail("ba_stream_set_poses: bad argument");
  if (s->finalized) return fail("ba_stream_set_poses: already finalized");
  s->n_pose = n_pose;
  s->pose_T.assign(T_jw12, T_jw12 + 12 * (size_t)n_pose);
  if (fixed) s->pose_fixed.assign(fixed, fixed + n_pose); else s->pose_fixed.assign(n_pose, 0);
  return 0;
}
int ba_stream_set_points(ba_stream *s, int n_pt, const double *X3, const uint8_t *fixed) {
  if (!s || n_pt <= 0 || !X3) return fail("ba_stream_set_points: bad argument");
  if (s->finalized) return fail("ba_stream_set_points: already finalized");
  s->n_pt = n_pt;
  s->pt_X.assign(X3, X3 + 3 * (size_t)n_pt);
  if (fixed) s->pt_fixed.assign(fixed, fixed + n_pt); else s->pt_fixed.assign(n_pt, 0);
  return 0;
}
int ba_stream_set_observations(ba_stream *s, int64_t n_obs, const int32_t *cam, const int32_t *pose,
                               const int32_t *point, const double *uv2) {
  if (!s || n_obs < 0 || (n_obs > 0 && (!cam || !pose || !point || !uv2)))
    return fail("ba_stream_set_observations: bad argument");
  if (s->finalized) return fail("ba_stream_set_observations: already finalized");
  s->n_obs = n_obs;
  s->obs_cam.assign(cam, cam + n_obs);
  s->obs_pose.assign(pose, pose + n_obs);
  s->obs_pt.assign(point, point + n_obs);
  s->obs_uv.assign(uv2, uv2 + 2 * n_obs);
  return 0;
}

int ba_stream_finalize(ba_stream *s) {
  if (!s) return fail("null stream handle");
  if (s->finalized) return 0;
  if (s->n_cam <= 0 || s->n_pose <= 0 || s->n_pt <= 0)
    return fail("ba_stream_finalize: cameras, poses and points must be set first");
  HIP_TRY(hipSetDevice(s->device));
  const int narena = s->K > 1 ? 2 : 1;
  for (int a = 0; a < narena; ++a) {
    HIP_TRY(hipMalloc((void **)&s->arena[a].base, (size_t)s->arena_bytes));
    s->arena[a].cap = (size_t)s->arena_bytes;
  }
  if (narena == 1) s->arena[1] = s->arena[0];
  HIP_TRY(hipMalloc((void **)&s->scal_sum, 4 * sizeof(double)));
  s->img.assign(s->K, nullptr);
  s->ev_in.resize(s->K);
  s->ev_comp.resize(s->K);
  s->in_pending.assign(s->K, 0);
  s->dirty.assign(s->K, 0);
  s->computed.assign(s->K, 0);
  for (int k = 0; k < s->K; ++k) {
    HIP_TRY(hipEventCreateWithFlags(&s->ev_in[k], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_comp[k], hipEventDisableTiming));
  }
  for (int k = 0; k < s->K; ++k) {
    ba_handle *h = nullptr;
    if (ba_create(&h, s->device)) return -1;
    s->h.push_back(h);
    h->arena = &s->arena[k & 1];
    h->dense_owner = k > 0 ? s->h[0] : nullptr;
    h->overlap = false;  // (one compute stream: the chunk loops are the overlap)
    if (ba_set_stream(h, (void *)s->s_comp) ||
        ba_set_cameras(h, s->n_cam, s->cam_intr.data(), s->cam_T.data()) ||
        ba_set_poses(h, s->n_pose, s->pose_T.data(), s->pose_fixed.data()) ||
        ba_set_points(h, s->n_pt, s->pt_X.data(), s->pt_fixed.data()) ||
        ba_set_observations(h, s->n_obs, s->obs_cam.data(), s->obs_pose.data(), s->obs_pt.data(), s->obs_uv.data()) ||
        ba_set_shard(h, k, s->K))
      return -1;
    // the previous occupant of this arena was imaged before this finalize overwrites it
    if (ba_finalize(h)) return -1;
    // the handle's host copy of the full problem is not needed again (it never re-plans)
    std::vector<int32_t>().swap(h->obs_cam);
    std::vector<int32_t>().swap(h->obs_pose);
    std::vector<int32_t>().swap(h->obs_pt);
    std::vector<double>().swap(h->obs_uv);
    const size_t bytes = h->arena_imm + h->arena_mut;
    HIP_TRY(hipHostMalloc((void **)&s->img[k], std::max<size_t>(bytes, 256), hipHostMallocDefault));
    HIP_TRY(hipDeviceSynchronize());
    if (h->arena_imm) HIP_TRY(hipMemcpy(s->img[k], s->arena[k & 1].base, h->arena_imm, hipMemcpyDeviceToHost));
    if (h->arena_mut)
      HIP_TRY(hipMemcpy(s->img[k] + h->arena_imm, mut_base(s, k), h->arena_mut, hipMemcpyDeviceToHost));
    s->resident[k & 1] = k;
  }
  // the problem's host vectors are in the plans / images now
  std::vector<int32_t>().swap(s->obs_cam);
  std::vector<int32_t>().swap(s->obs_pose);
  std::vector<int32_t>().swap(s->obs_pt);
  std::vector<double>().swap(s->obs_uv);
  s->finalized = true;
  return 0;
}

int ba_stream_lm_begin(ba_stream *s, const ba_options *opt) {
  if (!s || !opt) return fail("ba_stream_lm_begin: bad argument");
  if (!s->finalized && ba_stream_finalize(s)) return -1;
  HIP_TRY(hipSetDevice(s->device));
  int done_after = 0;
  if (stream_begin(s, opt, &done_after)) return -1;
  if (done_after) {
    for (ba_handle *h : s->h) {
      if (ba::ctrl_pull(h)) return -1;
      h->hc.done = 1;
      if (ba::ctrl_push(h)) return -1;
    }
  }
  s->lm_begun = true;
  return 0;
}

int ba_stream_lm_iterate(ba_stream *s, int n) {
  if (!s || !s->lm_begun) return fail("ba_stream_lm_iterate: call ba_stream_lm_begin first");
  HIP_TRY(hipSetDevice(s->device));
  for (int k = 0; k < n; ++k)
    if (stream_iteration(s)) return -1;   // (iterations after convergence are device-side no-ops)
  return 0;
}

int ba_stream_lm_sync(ba_stream *s, ba_iter_info *out, int cap, int *n_iter, int *converged) {
  if (!s || !s->lm_begun) return fail("ba_stream_lm_sync: call ba_stream_lm_begin first");
  HIP_TRY(hipSetDevice(s->device));
  ba_handle *h0 = s->h[0];
  HIP_TRY(hipStreamSynchronize(s->s_copy));
  if (ba::ctrl_pull(h0)) return -1;  // (synchronises the compute stream)
  {
    int bp = 0;
    HIP_TRY(hipMemcpy(&bp, h0->ddev.bad_pivots, sizeof(int), hipMemcpyDeviceToHost));
    if (bp >= ba::kFlowTimeout) return fail("ba_stream_lm_sync: a dataflow hand-off of the reduced solve timed out");
  }
  const int n = h0->hc.iter;
  if (n_iter) *n_iter = n;
  if (converged) *converged = h0->hc.converged;
  if (out && cap > 0 && n > 0) {
    const int m = std::min(std::min(n, cap), h0->d.log_cap);
    HIP_TRY(hipMemcpy(out, h0->d.log, (size_t)m * sizeof(ba_iter_info), hipMemcpyDeviceToHost));
  }
  return h0->hc.done ? 1 : 0;
}

int ba_stream_solve(ba_stream *s, const ba_options *opt, ba_iter_info *out, int cap, int *n_iter, int *converged) {
  if (ba_stream_lm_begin(s, opt)) return -1;
  int done = opt->max_num_iterations <= 0;
  for (int it = 0; !done && it < opt->max_num_iterations; ++it) {
    if (ba_stream_lm_iterate(s, 1)) return -1;
    // (one synchronisation per iteration: the loop is PCIe-bound anyway)
    const int rc = ba_stream_lm_sync(s, nullptr, 0, nullptr, nullptr);
    if (rc < 0) return -1;
    done = rc;
  }
  return ba_stream_lm_sync(s, out, cap, n_iter, converged) < 0 ? -1 : 0;
}

int ba_stream_get_poses(ba_stream *s, double *T_jw12) {
  if (!s || !s->finalized || !T_jw12) return fail("ba_stream_get_poses: bad argument");
  return ba_get_poses(s->h[0], T_jw12);
}

int ba_stream_get_points(ba_stream *s, double *X3) {
  if (!s || !s->finalized || !X3) return fail("ba_stream_get_points: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  if (flush(s)) return -1;
  for (int k = 0; k < s->K; ++k) {
    ba_handle *h = s->h[k];
    if (ba::ctrl_pull(h)) return -1;
    // the accepted point buffer inside the chunk's host image
    const char *dev = (const char *)h->d.pts[h->hc.cur];
    const size_t off = h->arena_imm + (size_t)(dev - mut_base(s, k));
    const double *src = (const double *)(s->img[k] + off);
    for (int q = 0; q < h->plan.n_pt; ++q)
      std::memcpy(X3 + (size_t)h->plan.pt_user_of_int[q] * 3, src + (size_t)q * 3, 3 * sizeof(double));
  }
  return 0;
}

int ba_stream_info(ba_stream *s, int64_t out6[6]) {
  if (!s || !s->finalized || !out6) return fail("ba_stream_info: bad argument");
  size_t big = 0, total = 0;
  for (ba_handle *h : s->h) {
    big = std::max(big, h->arena_imm + h->arena_mut);
    total += h->arena_imm + h->arena_mut;
  }
  out6[0] = (int64_t)s->arena_bytes * (s->K > 1 ? 2 : 1);
  out6[1] = (int64_t)big;
  out6[2] = (int64_t)total;
  out6[3] = s->bytes_h2d;
  out6[4] = s->bytes_d2h;
  out6[5] = s->K;
  return 0;
}

}  // extern "C"
