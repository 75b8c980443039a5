// mf_host.hpp -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "../../include/mocoflow_hip.h"

namespace mf {

char* last_error_buf();   // thread-local, 512 bytes (mf_abi.hip)

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return MF_OK;
}

}  // namespace mf
