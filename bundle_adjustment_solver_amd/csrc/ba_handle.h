// ba_handle.h — the handle behind the C ABI (include/ba_hip.h), shared by the
// translation units that implement it (ba_api.hip: the resident problem;
// ba_stream.hip: landmark chunks streamed through a device arena).  Internal.
#ifndef BA_HANDLE_H_
#define BA_HANDLE_H_

#include <hip/hip_runtime.h>

#include <cstdint>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/ba_hip.h"
#include "ba_dense_sched.h"
#include "ba_device.h"
#include "ba_plan.h"

namespace ba {
void launch_damp_invert_export(const DevProblem &d, hipStream_t s);
void set_last_error(const std::string &m);  // (also used by ba_rccl.cpp)
int fail(const std::string &m);             // sets the thread's error text, returns -1

// A device memory region that several handles carve their chunk-resident arrays
// from (ba_stream.hip): immutable arrays (structure, observations) grow from the
// bottom, mutable ones (blocks, points, partial sums) from the top, so that a
// chunk is swapped in with two copies and out with one.
struct DeviceArena {
  char *base = nullptr;
  size_t cap = 0;
};
}  // namespace ba

#define HIP_TRY(expr)                                                        \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess)                                                    \
      return ::ba::fail(std::string(#expr) + ": " + hipGetErrorString(e_));  \
  } while (0)

enum Stage { ST_BUILD = 0, ST_SCHUR, ST_SOLVE, ST_BACKSUB, ST_COST, ST_CTRL, ST_XCHG, ST_N = 8 };

struct ba_handle {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // second stream for work that is independent inside one LM iteration
  // (pose-side linearisation beside the landmark side + Schur accumulation;
  // pose update beside the back-substitution), joined with events
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_look[3] = {nullptr, nullptr, nullptr};  // lookahead of the dense three-kernel path
  bool overlap = true;
  // the side stream holds work of the last enqueued iteration (pose-side
  // linearisation at the trial point, reset of the factor tiles) that the main
  // stream has not joined yet
  bool side_pending = false;
  // the factor tiles were reset by the last k_backsub_update (its tile-reset role): the
  // next iteration needs no k_dense_init
  bool tiles_ready = false;
  // one LM iteration captured as a hipGraph (single GPU, no timing) and
  // replayed by ba_lm_iterate instead of ~50 separate launches.  Opt-in
  // (BA_GRAPH=1): on ROCm 7.2 / MI355X the replay measured 2.5 % SLOWER than
  // plain launches on C4 (993 vs 969 us per iteration), see DESIGN.md.
  bool use_graph = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  void drop_graph() {
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (graph) (void)hipGraphDestroy(graph);
    graph_exec = nullptr;
    graph = nullptr;
  }
  // host copies of the problem (scaled units)
  int n_cam = 0, n_pose = 0, n_pt = 0;
  int64_t n_obs = 0;
  std::vector<double> cam_intr, cam_T, pose_T, pt_X, obs_uv;
  std::vector<uint8_t> pose_fixed, pt_fixed;
  std::vector<int32_t> obs_cam, obs_pose, obs_pt;
  int rank = 0, world = 1;
  bool finalized = false;
  ba::Plan plan;
  ba::DevProblem d;
  std::vector<void *> allocs;
  ba_allreduce_fn ar_fn = nullptr;
  void *ar_user = nullptr;
  int64_t xbuf_n[3] = {0, 0, 0};
  // exchange buffer 2: every point of the full problem in user order (3 doubles each),
  // rows of points this shard does not own zero — ba_gather_points
  double *gbuf = nullptr;
  bool gbuf_bound = false;
  int32_t *pt_user_dev = nullptr;       // pt_user_of_int on the device (lazily)
  std::vector<double> gathered;         // result of the last ba_gather_points (host, user order)
  bool gathered_valid = false;
  ba::DevCtrl hc;  // host mirror for the stage API
  bool timing = false;
  hipEvent_t ev[ST_N + 1];
  bool ev_ok = false;
  double stage_ms[ST_N] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool lm_begun = false;
  int bl_gen = 0;            // generation number of the k_backsub_lin launches (flags are never reset)
  ba::KernelTimer kt;        // per-kernel event timing (diagnostic mode)
  ba::DenseSchedule sched;   // level schedule of the reduced-system Cholesky
  ba::DenseDev ddev;
  std::vector<int> pose_col_h;
  // pose-only scratch (grown on demand, reused across calls)
  // pose-only: one device buffer + its pinned host mirror (po_run), barrier scratch
  uint8_t *po_dev = nullptr, *po_host = nullptr;
  size_t po_cap = 0;
  float *po_part = nullptr;

  // ---- streaming (ba_stream.hip): chunk-resident arrays live in a shared arena ----
  ba::DeviceArena *arena = nullptr;
  size_t arena_imm = 0, arena_mut = 0;  // bytes used from the bottom / from the top
  int alloc_kind = 0;                   // 0 resident (hipMalloc), 1 arena immutable, 2 arena mutable
  // the handle whose dense image / schedule / solution this handle aliases (the
  // reduced system is solved ONCE per iteration, by the owner), or null
  ba_handle *dense_owner = nullptr;
  void kind(int k) { alloc_kind = arena ? k : 0; }

  template <class T>
  int dalloc(T **p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    if (arena && alloc_kind != 0) {
      const size_t sz = (n * sizeof(T) + 255) & ~(size_t)255;
      if (arena_imm + arena_mut + sz > arena->cap)
        return ::ba::fail("the landmark chunk does not fit the device arena (" + std::to_string(arena->cap >> 20) +
                          " MiB): use more chunks or a larger arena");
      if (alloc_kind == 1) {
        *p = (T *)(arena->base + arena_imm);
        arena_imm += sz;
      } else {
        arena_mut += sz;
        *p = (T *)(arena->base + arena->cap - arena_mut);
      }
      return 0;
    }
    hipError_t e = hipMalloc((void **)p, n * sizeof(T));
    if (e != hipSuccess)
      return ::ba::fail(std::string("hipMalloc: ") + hipGetErrorString(e));
    allocs.push_back((void *)*p);
    return 0;
  }
  // BA_PLAN_TIMES: bytes / seconds spent in upload() (allocation and copy apart)
  bool up_times = false;
  double up_alloc_s = 0, up_copy_s = 0;
  size_t up_bytes = 0, up_calls = 0;
  template <class T, class A>
  int upload(T **p, const std::vector<T, A> &v) {
    const auto t0 = up_times ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    if (dalloc(p, v.size())) return -1;
    const auto t1 = up_times ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    if (!v.empty()) {
      hipError_t e = hipMemcpy(*p, v.data(), v.size() * sizeof(T),
                               hipMemcpyHostToDevice);
      if (e != hipSuccess)
        return ::ba::fail(std::string("hipMemcpy H2D: ") + hipGetErrorString(e));
    }
    if (up_times) {
      const auto t2 = std::chrono::steady_clock::now();
      up_alloc_s += std::chrono::duration<double>(t1 - t0).count();
      up_copy_s += std::chrono::duration<double>(t2 - t1).count();
      up_bytes += v.size() * sizeof(T);
      ++up_calls;
    }
    return 0;
  }
  void free_device() {
    for (void *p : allocs) (void)hipFree(p);
    allocs.clear();
    if (po_dev) (void)hipFree(po_dev);
    if (po_host) (void)hipHostFree(po_host);
    po_dev = po_host = nullptr;
    po_cap = 0;
    po_part = nullptr;  // was in `allocs`
    gbuf = nullptr;     // (its own: in `allocs`; bound: the caller's)
    gbuf_bound = false;
    pt_user_dev = nullptr;
    gathered_valid = false;
    finalized = false;
  }
};

namespace ba {
int lm_prepare_ctrl(ba_handle *h, const ba_options *opt, int *done_after);
int ctrl_pull(ba_handle *h);  // device controller state -> h->hc (synchronises the stream)
int ctrl_push(ba_handle *h);
}  // namespace ba

#endif  // BA_HANDLE_H_
