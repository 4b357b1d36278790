// Internal helpers shared by the libm2h translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "m2h.h"

namespace m2h {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// Launch errors: sticky error is consumed so that a later call does not inherit it.
inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
  return 0;
}

#define M2H_REQUIRE(cond, ...)                     \
  do {                                             \
    if (!(cond)) return m2h::fail(-1, __VA_ARGS__); \
  } while (0)

inline hipStream_t as_stream(m2h_stream s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace m2h
