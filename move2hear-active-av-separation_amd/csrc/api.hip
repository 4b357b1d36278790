// extern "C" surface of libm2h.so (see include/m2h.h).  Argument adapters only; kernels live in
// conv_igemm.hip / layout.hip.
#include "m2h_internal.h"

namespace m2h {
thread_local char g_err[512] = {0};
int conv_igemm_f32(const m2h_conv_args& a, hipStream_t st);
size_t conv_igemm_workspace_bytes(const m2h_conv_args& a);
extern int g_force_splitk, g_force_stages, g_wide_stages;
}  // namespace m2h

using namespace m2h;

extern "C" {

int m2h_version(void) { return M2H_VERSION; }

const char* m2h_last_error(void) { return g_err; }

int m2h_conv_igemm_f32(const m2h_conv_args* args, m2h_stream stream) {
  M2H_REQUIRE(args != nullptr, "conv_igemm: null args");
  return conv_igemm_f32(*args, as_stream(stream));
}

size_t m2h_conv_igemm_workspace_bytes(const m2h_conv_args* args) {
  return args != nullptr ? conv_igemm_workspace_bytes(*args) : 0;
}

int m2h_debug_set(int knob, int value) {
  if (knob == 0) g_force_splitk = value;
  else if (knob == 1) g_force_stages = value;
  else if (knob == 2) g_wide_stages = value;
  else return fail(-1, "debug_set: unknown knob %d", knob);
  return 0;
}

static m2h_conv_args down_args(const float* x, const float* wp, const float* scale, const float* shift, const float* cls_table,
                               const float* cls_val, float* y, int B, int H, int W, int Ci, int Co) {
  m2h_conv_args a = {};
  a.src0 = x; a.src1 = nullptr; a.C0 = Ci; a.C1 = 0;
  a.B = B; a.Hi = H; a.Wi = W; a.Hq = H / 2; a.Wq = W / 2;
  a.stride = 2; a.nth = 4; a.ntw = 4; a.mulh = 1; a.offh = -1; a.mulw = 1; a.offw = -1;
  a.conv_transpose = 0; a.wp = wp; a.N = Co; a.scale = scale; a.shift = shift; a.slope = 0.2f;
  a.cls_table = cls_table; a.cls_val = cls_val;
  a.dst = y; a.Ho = H / 2; a.Wo = W / 2; a.os = 1; a.ph = 0; a.pw = 0; a.ldc = Co; a.out_mode = M2H_OUT_NHWC;
  return a;
}

// K3: Conv2d(4x4,s2,p1,no bias) + BN(eval) + LeakyReLU(0.2)     separator_cnn.py:5-12
int m2h_unet_down_fwd(const float* x, const float* wp, const float* scale, const float* shift, const float* cls_table,
                      const float* cls_val, float* y, int B, int H, int W, int Ci, int Co, void* workspace,
                      size_t workspace_bytes, m2h_stream stream) {
  M2H_REQUIRE(H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "unet_down: H, W must be even and >= 2 (got %d x %d)", H, W);
  m2h_conv_args a = down_args(x, wp, scale, shift, cls_table, cls_val, y, B, H, W, Ci, Co);
  a.workspace = workspace; a.workspace_bytes = workspace_bytes;
  return conv_igemm_f32(a, as_stream(stream));
}

size_t m2h_unet_down_workspace_bytes(int B, int H, int W, int Ci, int Co) {
  if (B <= 0 || H < 2 || W < 2 || Ci <= 0 || Co <= 0) return 0;
  return conv_igemm_workspace_bytes(down_args(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, H, W, Ci, Co));
}

static m2h_conv_args up_args(const float* x, const float* skip, const float* wp, const float* scale, const float* shift, float* y,
                             int B, int H, int W, int C0, int C1, int Co) {
  m2h_conv_args a = {};
  a.src0 = x; a.src1 = skip; a.C0 = C0; a.C1 = C1;
  a.B = B; a.Hi = H; a.Wi = W; a.Hq = H; a.Wq = W;
  a.stride = 1; a.nth = 2; a.ntw = 2; a.mulh = 0; a.offh = 0; a.mulw = 0; a.offw = 0;
  a.conv_transpose = 1; a.wp = wp; a.N = Co; a.scale = scale; a.shift = shift; a.slope = 0.f;
  a.dst = y; a.Ho = 2 * H; a.Wo = 2 * W; a.os = 2; a.ph = 0; a.pw = 0; a.ldc = Co; a.out_mode = M2H_OUT_NHWC;
  return a;
}

// K4: cat + ConvTranspose2d(4x4,s2,p1,no bias) + BN(eval) + ReLU    separator_cnn.py:15-24,156-161
int m2h_unet_up_fwd(const float* x, const float* skip, const float* wp, const float* scale, const float* shift, float* y,
                    int B, int H, int W, int C0, int C1, int Co, void* workspace, size_t workspace_bytes, m2h_stream stream) {
  m2h_conv_args a = up_args(x, skip, wp, scale, shift, y, B, H, W, C0, C1, Co);
  a.workspace = workspace; a.workspace_bytes = workspace_bytes;
  return conv_igemm_f32(a, as_stream(stream));
}

size_t m2h_unet_up_workspace_bytes(int B, int H, int W, int C0, int C1, int Co) {
  if (B <= 0 || H <= 0 || W <= 0 || C0 <= 0 || C1 < 0 || Co <= 0) return 0;
  return conv_igemm_workspace_bytes(up_args(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, H, W, C0, C1, Co));
}

// K5: Conv2d(1x1, bias) + de-slice + permute to BHWC    separator_cnn.py:134,163-168
int m2h_unet_head_fwd(const float* x, const float* wp, const float* bias, float* out, int B, int H, int W, int Ci, int Co,
                      m2h_stream stream) {
  m2h_conv_args a = {};
  a.src0 = x; a.src1 = nullptr; a.C0 = Ci; a.C1 = 0;
  a.B = B; a.Hi = H; a.Wi = W; a.Hq = H; a.Wq = W;
  a.stride = 1; a.nth = 1; a.ntw = 1; a.mulh = 0; a.offh = 0; a.mulw = 0; a.offw = 0;
  a.conv_transpose = 0; a.wp = wp; a.N = Co; a.scale = nullptr; a.shift = bias; a.slope = 1.f;
  a.dst = out; a.Ho = H; a.Wo = W; a.os = 1; a.ph = 0; a.pw = 0; a.ldc = Co; a.out_mode = M2H_OUT_DESLICE;
  return conv_igemm_f32(a, as_stream(stream));
}

}  // extern "C"
