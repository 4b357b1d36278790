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

// log1p(max(m * (exp(x) - 1), 0)) (separator_cnn.py:77-79) on the hardware transcendental units: v_exp_f32 / v_log_f32 (~1 ulp)
// and log1p(z) = log(u) * z / (u - 1), u = fl(1 + z), which gives back the bits that rounding 1 + z loses (exact z when u == 1).
// The library expf + log1pf cost ~100 instructions per element and made the masked slice ALU-bound (230 us at the headline
// shape against 80 us for the unmasked one); relative error of this form ~3e-7, four orders inside the 1e-3 contract.
__device__ __forceinline__ float masked_log_mag(float x, float m) {
  const float z = fmaxf(m * (__expf(x) - 1.f), 0.f);
  const float u = 1.f + z;
  const float d = u - 1.f;
  return d == 0.f ? z : __logf(u) * __fdividef(z, d);
}

}  // namespace m2h
