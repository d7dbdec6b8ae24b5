// See utility/rccl_allreduce.h.  A thin C++ face of the RCCL exchange that
// libba_hip.so itself exports (include/ba_hip.h "RCCL exchange",
// csrc/ba_rccl.cpp: librccl bound with dlopen at run time).
#include "utility/rccl_allreduce.h"

#include <cstring>

#include "ba_hip.h"

namespace visual_navigation {
namespace multi_gpu {

bool RcclAllReduce::Available(std::string *why_not) {
  const bool ok = ba_rccl_available() != 0;
  if (!ok && why_not) *why_not = ba_last_error();
  return ok;
}

std::string RcclAllReduce::NewUniqueId(std::string *error) {
  uint8_t id[128];
  if (ba_rccl_get_unique_id(id) != 0) {
    if (error) *error = ba_last_error();
    return std::string();
  }
  return std::string(reinterpret_cast<const char *>(id), sizeof(id));
}

RcclAllReduce::RcclAllReduce(int rank, int world, const std::string &unique_id, int device) {
  if (unique_id.size() != 128) {
    error_ = "RcclAllReduce: bad rank / world / unique id";
    return;
  }
  ba_rccl_comm *c = nullptr;
  if (ba_rccl_comm_create(&c, rank, world, reinterpret_cast<const uint8_t *>(unique_id.data()), device) != 0) {
    error_ = ba_last_error();
    return;
  }
  comm_ = c;
}

RcclAllReduce::~RcclAllReduce() {
  if (comm_) ba_rccl_comm_destroy(static_cast<ba_rccl_comm *>(comm_));
}

int64_t RcclAllReduce::calls() const { return comm_ ? ba_rccl_comm_calls(static_cast<ba_rccl_comm *>(comm_)) : 0; }

int RcclAllReduce::size() const { return comm_ ? ba_rccl_comm_size(static_cast<ba_rccl_comm *>(comm_)) : 0; }

int RcclAllReduce::Hook(void *user, int which, void *dev_ptr, int64_t n_doubles, void *hip_stream) {
  RcclAllReduce *self = static_cast<RcclAllReduce *>(user);
  if (!self || !self->comm_) return 1;
  const int rc = ba_rccl_allreduce_hook(self->comm_, which, dev_ptr, n_doubles, hip_stream);
  if (rc != 0) self->error_ = ba_last_error();
  return rc;
}

}  // namespace multi_gpu
}  // namespace visual_navigation
