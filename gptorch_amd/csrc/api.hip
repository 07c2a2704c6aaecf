// Library-level entry points of libgpnative (version, arch, error text).
#include <string>
#include "gpn_common.h"

namespace gpn {
static thread_local std::string g_last_error;
void set_hip_error(hipError_t e, const char* where) {
  g_last_error = std::string(hipGetErrorName(e)) + ": " + hipGetErrorString(e) + " at " + where;
}
}  // namespace gpn

extern "C" int gpn_version(void) { return GPN_VERSION; }
extern "C" const char* gpn_arch(void) { return "gfx950"; }
extern "C" const char* gpn_last_hip_error(void) { return gpn::g_last_error.c_str(); }
