// Internal helpers shared by the libm2h translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>

#include "m2h.h"
#include "m2h_tuning.h"

namespace m2h {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}


// Tuning / test knobs of the dispatch code: which of several kernels that compute the same values a launch takes (engine-vs-engine
// tests, A/B timing).  THREAD-LOCAL like the arithmetic mode -- the library holds no process-global mutable state: every
// launch reads the knobs of the host thread that makes it; m2h_tuning_set / _snapshot / _restore (include/m2h_tuning.h) are the only
// writers, and m2h.functional carries a forward pass's knobs into the autograd thread that runs its backward.  Index = knob number.
struct Tuning {
  int v[M2H_TUNING_KNOBS];
};
extern thread_local Tuning tl_tuning;
#define g_force_splitk (::m2h::tl_tuning.v[0])
#define g_force_stages (::m2h::tl_tuning.v[1])
#define g_wide_stages (::m2h::tl_tuning.v[2])
#define g_skinny (::m2h::tl_tuning.v[3])
#define g_narrow16 (::m2h::tl_tuning.v[4])
#define g_extra_lds (::m2h::tl_tuning.v[7])
#define g_phase_major (::m2h::tl_tuning.v[8])
#define g_fast_loader (::m2h::tl_tuning.v[9])
#define g_wgrad_blocks (::m2h::tl_tuning.v[11])
#define g_tapshare (::m2h::tl_tuning.v[15])
#define g_tap_bm (::m2h::tl_tuning.v[16])
#define g_tap_window (::m2h::tl_tuning.v[18])
#define g_wgrad_row3x3 (::m2h::tl_tuning.v[21])
#define g_row3x3 (::m2h::tl_tuning.v[22])
#define g_skinny_linear (::m2h::tl_tuning.v[23])
#define g_skinny_gather (::m2h::tl_tuning.v[24])
#define g_big_tile (::m2h::tl_tuning.v[26])
#define g_dma (::m2h::tl_tuning.v[27])
#define g_dma_shape (::m2h::tl_tuning.v[28])
#define g_quad (::m2h::tl_tuning.v[30])
#define g_dma_split2 (::m2h::tl_tuning.v[34])
#define g_strip (::m2h::tl_tuning.v[35])
#define g_patch (::m2h::tl_tuning.v[36])
#define g_skinny_mgb (::m2h::tl_tuning.v[38])
#define g_strip_rev (::m2h::tl_tuning.v[39])
#define g_skinny_tiny (::m2h::tl_tuning.v[33])
#define g_wgrad_small_m (::m2h::tl_tuning.v[25])
#define g_patch_grid (::m2h::tl_tuning.v[10])

// Label of the calling thread's most recent kernel launch (the `what` of launch_status: every launch site names its kernel family):
// read back by m2h_last_kernel / m2h_unet_fwd_stage_kernel, so that benchmark tables name the kernel that really ran.
extern thread_local const char* tl_last_launch;

// Every kernel launch of the library goes through M2H_LAUNCH: it counts (m2h_launch_count: the kernels this process enqueued or
// captured through libm2h -- a diagnostic for bench.py's per-phase launch figures, never read by the product path).
extern std::atomic<long long> g_launch_count;
#define M2H_LAUNCH(...)                                                      \
  do {                                                                       \
    ::m2h::g_launch_count.fetch_add(1, std::memory_order_relaxed);           \
    hipLaunchKernelGGL(__VA_ARGS__);                                         \
  } while (0)

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
