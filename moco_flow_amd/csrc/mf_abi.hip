// mf_abi.hip -- version / error plumbing of the C ABI (include/mocoflow_hip.h).
#include "mf_host.hpp"

namespace mf {
char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace mf

extern "C" int32_t mf_version(void) { return MF_ABI_VERSION; }
extern "C" const char* mf_last_error(void) { return mf::last_error_buf(); }
