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
// the --offload-arch this library was compiled for: build.sh passes it as -DGPN_ARCH=<arch> next to --offload-arch=<arch>
// (one variable), so a library built for another target says so instead of claiming gfx950
#ifndef GPN_ARCH
#error "build with -DGPN_ARCH=<the --offload-arch value> (gptorch_amd/csrc/build.sh does)"
#endif
#define GPN_STR2(x) #x
#define GPN_STR(x) GPN_STR2(x)
extern "C" const char* gpn_arch(void) { return GPN_STR(GPN_ARCH); }
extern "C" const char* gpn_last_hip_error(void) { return gpn::g_last_error.c_str(); }

extern "C" int gpn_fill_zero(void* stream, void* dst, int64_t bytes) {
  if (!dst) return -2;
  if (bytes < 0) return -3;
  if (bytes == 0) return GPN_OK;
  GPN_HIP_CHECK(hipMemsetAsync(dst, 0, (size_t)bytes, static_cast<hipStream_t>(stream)));
  return GPN_OK;
}
