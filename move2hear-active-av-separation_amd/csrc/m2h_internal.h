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

// Label of the calling thread's most recent kernel launch (the `what` of launch_status: every launch site names its kernel family):
// read back by m2h_last_kernel / m2h_unet_fwd_stage_kernel, so that benchmark tables name the kernel that really ran.
extern thread_local const char* tl_last_launch;

// Launch errors: sticky error is consumed so that a later call does not inherit it.
inline int launch_status(const char* what) {
  tl_last_launch = what;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
  return 0;
}

#define M2H_REQUIRE(cond, ...)                     \
  do {                                             \
    if (!(cond)) return m2h::fail(-1, __VA_ARGS__); \
  } while (0)

inline hipStream_t as_stream(m2h_stream s) { return reinterpret_cast<hipStream_t>(s); }

// log1p(max(m * (exp(x) - 1), 0)) (separator_cnn.py:77-79) on the hardware transcendental units, raw: v_exp_f32 / v_log_f32 /
// v_rcp_f32 (~1 ulp each) and log1p(z) = log(u) * z / (u - 1), u = fl(1 + z), which gives back the bits that rounding 1 + z loses
// (exact z when u == 1).  The operands are in the normal range by construction (x is a log-magnitude >= 0, u >= 1), so the
// denormal scaling, the extended-precision ln 2 and the IEEE division sequence that expf / logf / the `/` operator expand to --
// ~40 instructions and a branch per element, which made the fused first stage VALU-bound -- are not needed: ~12 instructions,
// relative error ~3e-7, four orders inside the 1e-3 contract.
__device__ __forceinline__ float masked_log_mag(float x, float m) {
  const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
  const float z = fmaxf(m * (e - 1.f), 0.f);
  const float u = 1.f + z;
  const float d = u - 1.f;
  const float r = (__builtin_amdgcn_logf(u) * 0.693147180559945309f) * (z * __builtin_amdgcn_rcpf(d));
  return d == 0.f ? z : r;
}

}  // namespace m2h
