// Multi-GPU exchange for the C++ facade: the sum-all-reduce that
// ba_set_allreduce (include/ba_hip.h) asks the caller to provide, implemented
// over RCCL — the C++ face of the ba_rccl_* entry points of libba_hip.so
// (include/ba_hip.h), which bind librccl.so with dlopen at run time, never at link
// time: a process that also holds PyTorch's bundled RCCL must not get a second
// copy, and machines without RCCL can still build and run the single-GPU path.
//
// One process per GPU.  Rank 0 creates the communicator id (NewUniqueId) and
// hands its 128 bytes to the other ranks by any side channel (a file, an
// environment variable, MPI, a socket); every rank then constructs
// RcclAllReduce(rank, world, id, device) and registers
// RcclAllReduce::Hook / this through FullBundleAdjustmentSolver::SetAllReduce.
// Per LM iteration the library calls the hook twice: the packed reduced camera
// system S||rhs (1.5 MB at BASELINE config C4) and four LM scalars, both
// in place, both ordered on the stream the kernels run on; once more at the end
// of Solve for the landmark rows (12 MB at C4), so that every rank writes back
// every point.
#ifndef BA_FACADE_RCCL_ALLREDUCE_H_
#define BA_FACADE_RCCL_ALLREDUCE_H_

#include <cstdint>
#include <string>

namespace visual_navigation {
namespace multi_gpu {

class RcclAllReduce {
 public:
  // true if an RCCL library can be loaded (BA_RCCL_LIB, else librccl.so[.1])
  static bool Available(std::string *why_not = nullptr);
  // 128 opaque bytes (ncclGetUniqueId); empty string on failure
  static std::string NewUniqueId(std::string *error = nullptr);

  RcclAllReduce(int rank, int world, const std::string &unique_id, int device);
  ~RcclAllReduce();
  RcclAllReduce(const RcclAllReduce &) = delete;
  RcclAllReduce &operator=(const RcclAllReduce &) = delete;

  bool ok() const { return comm_ != nullptr; }
  const std::string &error() const { return error_; }
  int64_t calls() const;  // all-reduces issued so far
  int size() const;       // ranks the communicator itself sees (ncclCommCount)

  // ba_allreduce_fn (include/ba_hip.h): user = RcclAllReduce*
  static int Hook(void *user, int which, void *dev_ptr, int64_t n_doubles, void *hip_stream);

 private:
  void *comm_ = nullptr;
  std::string error_;
};

}  // namespace multi_gpu
}  // namespace visual_navigation
#endif
