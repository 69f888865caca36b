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
extern "C" const char *daco_last_error(void) { return daco::g_err; }
