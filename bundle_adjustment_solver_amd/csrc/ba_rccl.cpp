// ba_rccl.cpp — the multi-GPU exchange of the BA hot path over RCCL, as plain C
// entry points (include/ba_hip.h, "RCCL exchange").  One process per GPU; the
// per-iteration collectives (packed S||rhs, LM scalars) and the final gather of
// the landmark shards are ncclAllReduce(sum, double) calls issued from the
// all-reduce hook of the library, i.e. from C++ on the stream the kernels run on:
// no interpreter inside the LM loop.
//
// librccl is bound at RUN time (dlopen), never at link time: a process that holds
// PyTorch already has RCCL in it ("librccl.so.1" then resolves to that copy by
// SONAME, so both share one library and one HIP runtime), and machines without
// RCCL still load libba_hip.so and run the single-GPU path.  Entry points used, by
// their documented C signatures (rccl.h): ncclGetUniqueId, ncclCommInitRank,
// ncclCommCount, ncclAllReduce, ncclCommDestroy, ncclGetErrorString;
// ncclFloat64 = 8, ncclSum = 0.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/ba_hip.h"

namespace ba {
void set_last_error(const std::string &m);  // ba_api.hip
}

namespace {

struct UniqueId {
  char internal[128];
};
constexpr int kNcclFloat64 = 8, kNcclSum = 0;

struct Api {
  void *lib = nullptr;
  int (*GetUniqueId)(UniqueId *) = nullptr;
  int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
  int (*CommCount)(void *, int *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, void *) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string why;
};

Api &load() {
  static Api api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *env = std::getenv("BA_RCCL_LIB");
    const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
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
    api.CommCount = reinterpret_cast<int (*)(void *, int *)>(sym("ncclCommCount"));
    api.AllReduce =
        reinterpret_cast<int (*)(const void *, void *, size_t, int, int, void *, void *)>(sym("ncclAllReduce"));
    api.CommDestroy = reinterpret_cast<int (*)(void *)>(sym("ncclCommDestroy"));
    api.GetErrorString = reinterpret_cast<const char *(*)(int)>(sym("ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommCount || !api.AllReduce || !api.CommDestroy) {
      api.why = "the loaded library does not export the RCCL entry points";
      dlclose(api.lib);
      api.lib = nullptr;
    }
  });
  return api;
}

int fail(const std::string &m) {
  ba::set_last_error(m);
  return -1;
}

std::string describe(Api &api, const char *what, int rc) {
  std::string s = std::string(what) + " failed";
  if (api.GetErrorString) s += std::string(": ") + api.GetErrorString(rc);
  return s;
}

}  // namespace

struct ba_rccl_comm {
  void *comm = nullptr;
  int rank = 0, world = 1, device = 0;
  int64_t calls = 0;
};

extern "C" {

int ba_rccl_available(void) {
  Api &api = load();
  if (!api.lib) ba::set_last_error("RCCL is not available: " + (api.why.empty() ? std::string("librccl.so not found") : api.why));
  return api.lib ? 1 : 0;
}

int ba_rccl_get_unique_id(uint8_t id128[128]) {
  if (!id128) return fail("ba_rccl_get_unique_id: null argument");
  Api &api = load();
  if (!api.lib) return fail("RCCL is not available: " + api.why);
  UniqueId id;
  std::memset(&id, 0, sizeof(id));
  const int rc = api.GetUniqueId(&id);
  if (rc != 0) return fail(describe(api, "ncclGetUniqueId", rc));
  std::memcpy(id128, id.internal, sizeof(id.internal));
  return 0;
}

int ba_rccl_comm_create(ba_rccl_comm **out, int rank, int world, const uint8_t id128[128], int device) {
  if (!out) return fail("ba_rccl_comm_create: null out pointer");
  *out = nullptr;
  if (!id128 || world < 1 || rank < 0 || rank >= world) return fail("ba_rccl_comm_create: bad rank / world / id");
  Api &api = load();
  if (!api.lib) return fail("RCCL is not available: " + api.why);
  if (hipSetDevice(device) != hipSuccess) return fail("ba_rccl_comm_create: hipSetDevice failed");
  UniqueId id;
  std::memcpy(id.internal, id128, sizeof(id.internal));
  ba_rccl_comm *c = new ba_rccl_comm();
  c->rank = rank;
  c->world = world;
  c->device = device;
  const int rc = api.CommInitRank(&c->comm, world, id, rank);
  if (rc != 0 || !c->comm) {
    delete c;
    return fail(describe(api, "ncclCommInitRank", rc));
  }
  *out = c;
  return 0;
}

int ba_rccl_comm_size(ba_rccl_comm *c) {
  if (!c || !c->comm) return fail("ba_rccl_comm_size: null communicator");
  int n = 0;
  const int rc = load().CommCount(c->comm, &n);
  if (rc != 0) return fail(describe(load(), "ncclCommCount", rc));
  return n;
}

int64_t ba_rccl_comm_calls(ba_rccl_comm *c) { return c ? c->calls : -1; }

void ba_rccl_comm_destroy(ba_rccl_comm *c) {
  if (!c) return;
  if (c->comm) {
    (void)hipSetDevice(c->device);
    (void)load().CommDestroy(c->comm);
  }
  delete c;
}

int ba_rccl_allreduce_hook(void *user, int /*which*/, void *dev_ptr, int64_t n_doubles, void *hip_stream) {
  ba_rccl_comm *c = static_cast<ba_rccl_comm *>(user);
  if (!c || !c->comm || !dev_ptr || n_doubles < 0) return 1;
  Api &api = load();
  const int rc =
      api.AllReduce(dev_ptr, dev_ptr, static_cast<size_t>(n_doubles), kNcclFloat64, kNcclSum, c->comm, hip_stream);
  if (rc != 0) {
    ba::set_last_error(describe(api, "ncclAllReduce", rc));
    return 1;
  }
  ++c->calls;
  return 0;
}

}  // extern "C"
