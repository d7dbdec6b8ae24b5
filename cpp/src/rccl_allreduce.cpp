// See utility/rccl_allreduce.h.  Entry points used, by their documented C
// signatures (rccl.h): ncclGetUniqueId, ncclCommInitRank, ncclAllReduce,
// ncclCommDestroy, ncclGetErrorString; datatype ncclFloat64 = 8, op ncclSum = 0.
#include "utility/rccl_allreduce.h"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace visual_navigation {
namespace multi_gpu {

namespace {

struct UniqueId {
  char internal[128];
};
constexpr int kNcclFloat64 = 8, kNcclSum = 0;

struct Api {
  void *lib = nullptr;
  int (*GetUniqueId)(UniqueId *) = nullptr;
  int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, void *) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  int (*hipSetDevice)(int) = nullptr;
  std::string why;
};

Api &Load() {
  static Api api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *env = std::getenv("BA_RCCL_LIB");
    const char *names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) {
      if (!n || !*n) continue;
      api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
      api.why = dlerror();
    }
    if (!api.lib) return;
    auto sym = [](const char *name) { return dlsym(api.lib, name); };
    api.GetUniqueId = reinterpret_cast<int (*)(UniqueId *)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<int (*)(void **, int, UniqueId, int)>(sym("ncclCommInitRank"));
    api.AllReduce = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, void *, void *)>(sym("ncclAllReduce"));
    api.CommDestroy = reinterpret_cast<int (*)(void *)>(sym("ncclCommDestroy"));
    api.GetErrorString = reinterpret_cast<const char *(*)(int)>(sym("ncclGetErrorString"));
    // the HIP runtime is already in the process (libba_hip.so links it)
    api.hipSetDevice = reinterpret_cast<int (*)(int)>(dlsym(RTLD_DEFAULT, "hipSetDevice"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) {
      api.why = "the loaded library does not export the RCCL entry points";
      dlclose(api.lib);
      api.lib = nullptr;
    }
  });
  return api;
}

std::string Describe(Api &api, const char *what, int rc) {
  std::string s = std::string(what) + " failed";
  if (api.GetErrorString) s += std::string(": ") + api.GetErrorString(rc);
  return s;
}

}  // namespace

bool RcclAllReduce::Available(std::string *why_not) {
  Api &api = Load();
  if (!api.lib && why_not) *why_not = api.why.empty() ? "librccl.so not found" : api.why;
  return api.lib != nullptr;
}

std::string RcclAllReduce::NewUniqueId(std::string *error) {
  Api &api = Load();
  if (!api.lib) {
    if (error) *error = "RCCL is not available: " + api.why;
    return std::string();
  }
  UniqueId id;
  std::memset(&id, 0, sizeof(id));
  const int rc = api.GetUniqueId(&id);
  if (rc != 0) {
    if (error) *error = Describe(api, "ncclGetUniqueId", rc);
    return std::string();
  }
  return std::string(id.internal, sizeof(id.internal));
}

RcclAllReduce::RcclAllReduce(int rank, int world, const std::string &unique_id, int device) {
  Api &api = Load();
  if (!api.lib) {
    error_ = "RCCL is not available: " + api.why;
    return;
  }
  if (unique_id.size() != sizeof(UniqueId) || world < 1 || rank < 0 || rank >= world) {
    error_ = "RcclAllReduce: bad rank / world / unique id";
    return;
  }
  if (api.hipSetDevice && api.hipSetDevice(device) != 0) {
    error_ = "hipSetDevice failed";
    return;
  }
  UniqueId id;
  std::memcpy(id.internal, unique_id.data(), sizeof(id.internal));
  const int rc = api.CommInitRank(&comm_, world, id, rank);
  if (rc != 0) {
    comm_ = nullptr;
    error_ = Describe(api, "ncclCommInitRank", rc);
  }
}

RcclAllReduce::~RcclAllReduce() {
  if (comm_) (void)Load().CommDestroy(comm_);
}

int RcclAllReduce::Hook(void *user, int /*which*/, void *dev_ptr, int64_t n_doubles, void *hip_stream) {
  RcclAllReduce *self = static_cast<RcclAllReduce *>(user);
  if (!self || !self->comm_ || !dev_ptr || n_doubles < 0) return 1;
  Api &api = Load();
  const int rc = api.AllReduce(dev_ptr, dev_ptr, static_cast<size_t>(n_doubles), kNcclFloat64, kNcclSum, self->comm_,
                               hip_stream);
  if (rc != 0) {
    self->error_ = Describe(api, "ncclAllReduce", rc);
    return 1;
  }
  ++self->calls_;
  return 0;
}

}  // namespace multi_gpu
}  // namespace visual_navigation
