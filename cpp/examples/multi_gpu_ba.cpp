// Multi-GPU full bundle adjustment through the C++ facade: one process per GPU,
// landmarks sharded across the ranks, RCCL sum-all-reduce of the reduced camera
// system (utility/rccl_allreduce.h).
//
//   multi_gpu_ba <rank> <world> <id_file> [device]
//
// Rank 0 creates the RCCL unique id and writes it to <id_file>; the other ranks
// wait for the file.  Every rank registers the SAME seeded stereo scene, owns
// the landmark shard ba_set_shard gives it, and ends with identical poses.
// With world == 1 (what a one-GPU box can run) the program also solves the
// problem without the hook and requires bit-identical results: the exchange
// path (packed S||rhs through ncclAllReduce on the solver's stream, separate
// control kernel) must not change a single bit when there is nothing to add.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <random>
#include <thread>
#include <vector>

#include "core/full_bundle_adjustment_solver.h"
#include "utility/rccl_allreduce.h"

using namespace visual_navigation::analytic_solver;
using visual_navigation::multi_gpu::RcclAllReduce;
using Pose = _BA_Pose;
using Point = _BA_Point;

struct Scene {
  std::vector<_BA_Camera> cameras;
  std::vector<Pose> poses;   // user storage the solver writes back into
  std::vector<Point> points;
  struct Obs { int cam, pose, point; _BA_Pixel px; };
  std::vector<Obs> obs;
  int num_fixed = 5;
};

// stereo rig moving along +x and looking along +z; every landmark is seen by a
// window of consecutive poses (the layout of the BASELINE configs, small)
static Scene MakeScene(int n_pose, int n_point, unsigned seed) {
  Scene s;
  std::mt19937 gen(seed);
  std::uniform_real_distribution<double> u01(0.0, 1.0);
  _BA_Camera left, right;
  left.fx = right.fx = left.fy = right.fy = 525.0;
  left.cx = right.cx = 320.0;
  left.cy = right.cy = 240.0;
  left.pose_this_to_cam0 = Pose::Identity();
  right.pose_this_to_cam0 = Pose::Identity();
  right.pose_this_to_cam0.translation() = Point(-0.12, 0.0, 0.0);
  s.cameras = {left, right};
  std::vector<Pose> truth(n_pose, Pose::Identity());
  for (int j = 0; j < n_pose; ++j) truth[j].translation() = Point(0.2 * j, 0.0, 0.0);
  s.poses = truth;
  for (int j = s.num_fixed; j < n_pose; ++j)
    s.poses[j].translation() = s.poses[j].translation() + Point(0.2 * (u01(gen) - 0.5), 0.2 * (u01(gen) - 0.5), 0.2 * (u01(gen) - 0.5));
  const int window = 5;
  while ((int)s.points.size() < n_point) {
    const int first = (int)(u01(gen) * (n_pose - window + 1));
    const double depth = 4.0 + 8.0 * u01(gen);
    const Point X(0.2 * (first + 2) + (u01(gen) - 0.5) * depth, (u01(gen) - 0.5) * 0.7 * depth, depth);
    std::vector<Scene::Obs> mine;
    bool ok = true;
    for (int w = 0; w < window && ok; ++w)
      for (int c = 0; c < 2 && ok; ++c) {
        const Point l = s.cameras[c].pose_this_to_cam0 * (truth[first + w].inverse() * X);
        const _BA_Pixel px(525.0 * l(0) / l(2) + 320.0, 525.0 * l(1) / l(2) + 240.0);
        ok = l(2) > 0 && px(0) > 0 && px(0) < 640 && px(1) > 0 && px(1) < 480;
        mine.push_back({c, first + w, (int)s.points.size(), px});
      }
    if (!ok) continue;
    s.points.push_back(X + Point(u01(gen) - 0.5, u01(gen) - 0.5, u01(gen) - 0.5));
    s.obs.insert(s.obs.end(), mine.begin(), mine.end());
  }
  return s;
}

static void Register(Scene &s, FullBundleAdjustmentSolver &ba) {
  ba.SetVerbose(false);
  for (size_t c = 0; c < s.cameras.size(); ++c) ba.AddCamera((int)c, s.cameras[c]);
  for (Pose &p : s.poses) ba.AddPose(&p);
  for (Point &x : s.points) ba.AddPoint(&x);
  for (int j = 0; j < s.num_fixed; ++j) ba.MakePoseFixed(&s.poses[j]);
  for (const Scene::Obs &o : s.obs) ba.AddObservation(o.cam, &s.poses[o.pose], &s.points[o.point], o.px);
}

// ---- "threads" mode: `world` shard facades on host threads of ONE process, all on
// device 0 (what a one-GPU box can run with more than one shard).  The hook stages
// the buffers through the host and meets the other shards at a barrier: the protocol
// of the library (two exchanges per iteration + the final landmark gather) with a
// real sum.  Every rank must write back EVERY pose and EVERY point (reference
// core/full_bundle_adjustment_solver.cpp:1011-1022), identical on all ranks and
// equal to the unsharded solve up to the permuted summation order.
namespace {
struct ThreadExchange {
  int world = 1;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  long generation = 0;
  std::vector<std::vector<double>> part;  // per rank
  std::vector<double> sum;
  void Barrier() {
    std::unique_lock<std::mutex> lk(m);
    const long g = generation;
    if (++arrived == world) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != g; });
    }
  }
};
struct ThreadRank {
  ThreadExchange *x;
  int rank;
  long calls = 0;
};
int ThreadHook(void *user, int /*which*/, void *dev_ptr, int64_t n, void *stream) {
  ThreadRank *r = static_cast<ThreadRank *>(user);
  ThreadExchange &x = *r->x;
  std::vector<double> &mine = x.part[r->rank];
  mine.resize((size_t)n);
  if (hipMemcpyAsync(mine.data(), dev_ptr, (size_t)n * 8, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
    return 1;
  x.Barrier();
  if (r->rank == 0) {
    x.sum.assign((size_t)n, 0.0);
    for (int k = 0; k < x.world; ++k)
      for (int64_t e = 0; e < n; ++e) x.sum[e] += x.part[k][e];
  }
  x.Barrier();
  if (hipMemcpyAsync(dev_ptr, x.sum.data(), (size_t)n * 8, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
    return 1;
  x.Barrier();  // nobody overwrites `sum` before everyone has copied it
  ++r->calls;
  return 0;
}

int RunThreads(int world) {
  Options options;
  options.iteration_handle.max_num_iterations = 12;
  options.convergence_handle.threshold_cost_change = 0.0f;
  options.convergence_handle.threshold_step_size = 0.0f;
  Scene plain_scene = MakeScene(40, 3000, 20240611u);
  {
    FullBundleAdjustmentSolver plain;
    Register(plain_scene, plain);
    plain.Solve(options, nullptr);
  }
  ThreadExchange x;
  x.world = world;
  x.part.resize(world);
  std::vector<Scene> scenes;
  for (int r = 0; r < world; ++r) scenes.push_back(MakeScene(40, 3000, 20240611u));
  std::vector<ThreadRank> ranks(world);
  std::vector<size_t> owned(world, 0);
  std::vector<std::thread> th;
  for (int r = 0; r < world; ++r) {
    ranks[r].x = &x;
    ranks[r].rank = r;
    th.emplace_back([&, r] {
      FullBundleAdjustmentSolver ba;
      Register(scenes[r], ba);
      ba.SetShard(r, world);
      ba.SetAllReduce(&ThreadHook, &ranks[r]);
      ba.Solve(options, nullptr);
      for (const Point &p : scenes[r].points) owned[r] += ba.OwnsPoint(&p);
    });
  }
  for (std::thread &t : th) t.join();
  int fail = 0;
  size_t owned_total = 0;
  for (int r = 0; r < world; ++r) {
    owned_total += owned[r];
    if (ranks[r].calls != 2 * 12 + 2) {  // begin + 2 per iteration + the landmark gather
      std::printf("rank %d: %ld hook calls, expected %d\n", r, ranks[r].calls, 2 * 12 + 2);
      ++fail;
    }
    double dmax = 0, pmax = 0;
    bool same_as_rank0 = true;
    for (size_t q = 0; q < scenes[r].points.size(); ++q)
      for (int c = 0; c < 3; ++c) {
        dmax = std::max(dmax, std::fabs(scenes[r].points[q](c) - plain_scene.points[q](c)));
        same_as_rank0 = same_as_rank0 && scenes[r].points[q](c) == scenes[0].points[q](c);
      }
    for (size_t j = 0; j < scenes[r].poses.size(); ++j)
      for (int c = 0; c < 3; ++c) {
        pmax = std::max(pmax, std::fabs(scenes[r].poses[j].translation()(c) - plain_scene.poses[j].translation()(c)));
        same_as_rank0 = same_as_rank0 && scenes[r].poses[j].translation()(c) == scenes[0].poses[j].translation()(c);
      }
    std::printf("rank %d/%d: owns %zu landmarks; written-back points differ from the unsharded solve by %.2e m, poses by %.2e m; %s rank 0\n",
                r, world, owned[r], dmax, pmax, same_as_rank0 ? "bit-identical to" : "DIFFERENT from");
    if (!(dmax < 1e-7) || !(pmax < 1e-7) || !same_as_rank0) ++fail;
  }
  if (owned_total != plain_scene.points.size()) {
    std::printf("the shards own %zu of %zu landmarks\n", owned_total, plain_scene.points.size());
    ++fail;
  }
  std::printf(fail ? "THREAD SHARD TEST FAILED\n" : "THREAD SHARD TEST PASSED: every rank wrote back every point\n");
  return fail ? 1 : 0;
}
}  // namespace

int main(int argc, char **argv) {
  if (argc == 3 && std::string(argv[1]) == "threads") return RunThreads(std::atoi(argv[2]));
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <rank> <world> <id_file> [device]\n", argv[0]);
    return 2;
  }
  const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
  const std::string id_file = argv[3];
  const int device = argc > 4 ? std::atoi(argv[4]) : rank;
  std::string why;
  if (!RcclAllReduce::Available(&why)) {
    std::printf("RCCL not available (%s)\n", why.c_str());
    return 3;
  }
  // ---- communicator id: rank 0 writes it, the others wait for it ----
  std::string id;
  if (rank == 0) {
    id = RcclAllReduce::NewUniqueId(&why);
    if (id.empty()) {
      std::printf("%s\n", why.c_str());
      return 1;
    }
    std::ofstream tmp(id_file + ".tmp", std::ios::binary);
    tmp.write(id.data(), (std::streamsize)id.size());
    tmp.close();
    std::rename((id_file + ".tmp").c_str(), id_file.c_str());
  } else {
    for (int tries = 0; tries < 600 && id.size() != 128; ++tries) {
      std::ifstream in(id_file, std::ios::binary);
      id.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
      if (id.size() != 128) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    if (id.size() != 128) {
      std::printf("rank %d: no unique id in %s\n", rank, id_file.c_str());
      return 1;
    }
  }
  Options options;
  options.iteration_handle.max_num_iterations = 15;
  options.convergence_handle.threshold_cost_change = 0.0f;
  options.convergence_handle.threshold_step_size = 0.0f;

  Scene scene = MakeScene(60, 4000, 20240607u);
  FullBundleAdjustmentSolver ba;
  ba.SetDevice(device);
  Register(scene, ba);
  ba.SetShard(rank, world);
  ba.FinalizeParameters();  // selects the device for this thread
  RcclAllReduce exchange(rank, world, id, device);
  if (!exchange.ok()) {
    std::printf("rank %d: %s\n", rank, exchange.error().c_str());
    return 1;
  }
  ba.SetAllReduce(&RcclAllReduce::Hook, &exchange);
  Summary summary;
  ba.Solve(options, &summary);
  const auto &rows = summary.GetOptimizationInfoList();
  size_t owned = 0;
  for (const Point &x : scene.points) owned += ba.OwnsPoint(&x);
  std::printf("rank %d/%d: %zu iterations, cost %.9e -> %.9e, %lld all-reduce calls, owns %zu of %zu landmarks\n", rank,
              world, rows.size(), rows.empty() ? 0.0 : rows.front().cost, rows.empty() ? 0.0 : rows.back().cost,
              (long long)exchange.calls(), owned, scene.points.size());
  int fail = 0;
  if (rows.size() != 15 || !(rows.back().cost < 0.5 * rows.front().cost)) {
    std::printf("rank %d: the sharded solve did not make progress\n", rank);
    ++fail;
  }
  if (exchange.calls() < 2 * 15 + 1) {
    std::printf("rank %d: expected two exchanges per iteration\n", rank);
    ++fail;
  }
  if (world == 1) {
    // the same problem without the hook: bit-identical trajectory and result
    Scene plain_scene = MakeScene(60, 4000, 20240607u);
    FullBundleAdjustmentSolver plain;
    plain.SetDevice(device);
    Register(plain_scene, plain);
    Summary plain_summary;
    plain.Solve(options, &plain_summary);
    const auto &prows = plain_summary.GetOptimizationInfoList();
    bool same = prows.size() == rows.size();
    for (size_t k = 0; same && k < rows.size(); ++k) same = rows[k].cost == prows[k].cost && rows[k].damping_term == prows[k].damping_term;
    for (size_t j = 0; same && j < scene.poses.size(); ++j)
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) same = same && scene.poses[j].linear()(r, c) == plain_scene.poses[j].linear()(r, c);
        same = same && scene.poses[j].translation()(r) == plain_scene.poses[j].translation()(r);
      }
    for (size_t i = 0; same && i < scene.points.size(); ++i)
      for (int r = 0; r < 3; ++r) same = same && scene.points[i](r) == plain_scene.points[i](r);
    if (!same) {
      std::printf("world 1: the RCCL hook changed the result\n");
      ++fail;
    } else {
      std::printf("world 1: RCCL hook path is bit-identical to the plain solve\n");
    }
  }
  std::printf(fail ? "MULTI GPU EXAMPLE FAILED\n" : "MULTI GPU EXAMPLE PASSED\n");
  return fail ? 1 : 0;
}
