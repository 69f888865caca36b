// daco_api.cpp -- version / error plumbing of libdeepaco_hip.so (host only).
#include <cstdarg>
#include <cstdio>

#include "../../include/deepaco_hip.h"

namespace daco {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
}  // namespace daco

extern "C" int daco_version(void) { return DACO_VERSION; }

// ---------------------------------------------------------------------------------------------------------------
// daco_allreduce_delta_tau -- the one collective of the ant-sharded colony for binders without torch.distributed (SURVEY.md
// 8(b) / 8(e): each rank's deposits delta-tau [B][n][n] f32 summed in place over the ranks, then tau <- decay tau + delta on
// every rank).  RCCL is resolved at run time: the copy the process already holds (a Python host has torch's, whose
// communicators are that copy's), else librccl.so.1 from the loader path or /opt/rocm/lib -- the library has no link-time
// dependency on it, and a single-GPU user never loads it.
#include <dlfcn.h>
namespace {
typedef int (*nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, void *);
nccl_allreduce_fn resolve_allreduce() {
  static nccl_allreduce_fn fn = []() -> nccl_allreduce_fn {
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) if (!h) h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    for (const char *nm : names) if (!h) h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    return h ? (nccl_allreduce_fn)dlsym(h, "ncclAllReduce") : nullptr;
  }();
  return fn;
}
}  // namespace

extern "C" int daco_allreduce_delta_tau(void *comm, void *stream, float *delta, size_t count) {
  if (!comm || !delta || count == 0) { daco::set_error("daco_allreduce_delta_tau: bad argument"); return DACO_E_BADARG; }
  nccl_allreduce_fn ar = resolve_allreduce();
  if (!ar) { daco::set_error("daco_allreduce_delta_tau: librccl.so not found (%s)", dlerror() ? dlerror() : "no ncclAllReduce"); return DACO_E_HIP; }
  const int rc = ar(delta, delta, count, /* ncclFloat32 */ 7, /* ncclSum */ 0, comm, stream);
  if (rc != 0) { daco::set_error("daco_allreduce_delta_tau: ncclAllReduce returned %d", rc); return DACO_E_HIP; }
  return DACO_OK;
}
extern "C" const char *daco_last_error(void) { return daco::g_err; }
