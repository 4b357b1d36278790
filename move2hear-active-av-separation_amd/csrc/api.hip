// extern "C" surface of libm2h.so (see include/m2h.h).  Argument adapters only; kernels live in
// conv_igemm.hip / layout.hip.
#include "m2h_internal.h"

namespace m2h {
thread_local char g_err[512] = {0};
thread_local const char* tl_last_launch = "";
thread_local const char* tl_unet_stage[11] = {"", "", "", "", "", "", "", "", "", "", ""};
struct ConvL1;
int conv_igemm_f32(const m2h_conv_args& a, hipStream_t st, const ConvL1* l1 = nullptr);
int launch_strip_conv1(const float* mix, const float* masks, const void* wreg, const float* scale, const float* shift, const float* cls_table,
                       const float* cls_val, float* dst, int B, int T, float slope, hipStream_t st, int cls_kind);
int sep_slice_input_cls(const float* mix, const float* masks, float* out, int B, int F, int T, int split_out, const void* cls_raw, int cls_kind,
                        float* cls_out, hipStream_t st);
size_t conv_igemm_workspace_bytes(const m2h_conv_args& a);
thread_local Tuning tl_tuning = {};   // every knob 0 = automatic (m2h_internal.h)
extern thread_local int tl_math_mode;
extern thread_local int tl_hi_only;
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

const char* m2h_last_kernel(void) { return tl_last_launch; }

const char* m2h_unet_fwd_stage_kernel(int stage) { return (stage >= 0 && stage < 11) ? tl_unet_stage[stage] : ""; }

int m2h_set_math_mode(int mode) {
  M2H_REQUIRE(mode == M2H_MATH_FP32 || mode == M2H_MATH_BF16X3 || mode == M2H_MATH_BF16, "set_math_mode: mode must be M2H_MATH_FP32, M2H_MATH_BF16X3 or M2H_MATH_BF16");
  tl_math_mode = mode == M2H_MATH_FP32 ? 0 : 1;
  tl_hi_only = mode == M2H_MATH_BF16 ? 1 : 0;
  return 0;
}

long long m2h_launch_count(void) { return m2h::g_launch_count.load(std::memory_order_relaxed); }

int m2h_get_math_mode(void) { return tl_hi_only ? M2H_MATH_BF16 : tl_math_mode; }

int m2h_tuning_set(int knob, int value) {
  if (knob == 14) return m2h_set_math_mode(value);   // (kept for older callers; thread-local like the rest)
  M2H_REQUIRE(knob >= 0 && knob < M2H_TUNING_KNOBS, "tuning_set: unknown knob %d", knob);
  tl_tuning.v[knob] = value;   // numbers of experiments that were measured and removed are accepted and read by nothing
  return 0;
}

int m2h_tuning_snapshot(int* out, int n) {
  M2H_REQUIRE(out != nullptr && n == M2H_TUNING_KNOBS, "tuning_snapshot: need room for %d knobs", M2H_TUNING_KNOBS);
  for (int i = 0; i < M2H_TUNING_KNOBS; ++i) out[i] = tl_tuning.v[i];
  return 0;
}

int m2h_tuning_restore(const int* in, int n) {
  M2H_REQUIRE(in != nullptr && n == M2H_TUNING_KNOBS, "tuning_restore: need %d knobs", M2H_TUNING_KNOBS);
  for (int i = 0; i < M2H_TUNING_KNOBS; ++i) tl_tuning.v[i] = in[i];
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

int m2h_unet_up_head_fwd(const float* x, const float* skip, const float* wp, const float* scale, const float* shift, const float* head_w,
                         const float* head_b, float* out, int B, int H, int W, int C0, int C1, int Co, m2h_stream stream) {
  M2H_REQUIRE(head_w != nullptr && head_b != nullptr, "unet_up_head: null head");
  m2h_conv_args a = up_args(x, skip, wp, scale, shift, out, B, H, W, C0, C1, Co);
  a.out_mode = M2H_OUT_DESLICE;
  a.head_w = head_w;
  a.head_b = head_b;
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

// ---- whole-network runner -----------------------------------------------------------------------------------------
static const int kEnc[6] = {32, 64, 128, 256, 512, 512};

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

struct UnetLayout {
  size_t x0, e[5], d[5], splitk, cls, total;
};

static UnetLayout unet_layout(int B, int F, int T, int n_out) {
  UnetLayout L;
  const int H = F / 16;
  size_t off = 0;
  L.x0 = off; off += align256((size_t)B * H * T * 32 * 4);
  int h = H, w = T;
  for (int i = 0; i < 5; ++i) {
    h /= 2; w /= 2;
    L.e[i] = off; off += align256((size_t)B * h * w * kEnc[i + 1] * 4);
  }
  const int dco[5] = {512, 256, 128, 64, n_out};
  for (int i = 0; i < 5; ++i) {
    h *= 2; w *= 2;
    L.d[i] = off; off += align256((size_t)B * h * w * dco[i] * 4);
  }
  // split-K scratch: the largest request of any stage
  size_t sk = 0;
  h = H; w = T;
  for (int i = 0; i < 5; ++i) {
    size_t b = m2h_unet_down_workspace_bytes(B, h, w, kEnc[i], kEnc[i + 1]);
    if (b > sk) sk = b;
    h /= 2; w /= 2;
  }
  const int c0[5] = {512, 512, 256, 128, 64}, c1[5] = {0, 512, 256, 128, 64};
  for (int i = 0; i < 5; ++i) {
    size_t b = m2h_unet_up_workspace_bytes(B, h, w, c0[i], c1[i], dco[i]);
    if (b > sk) sk = b;
    h *= 2; w *= 2;
  }
  L.splitk = off; off += align256(sk);
  L.cls = off; off += align256((size_t)B * 4);   // the class plane's values when the caller hands over the raw target_class (cls_kind)
  L.total = off;
  return L;
}

size_t m2h_unet_fwd_workspace_bytes(int B, int F, int T) {
  if (B <= 0 || F <= 0 || T <= 0) return 0;
  return unet_layout(B, F, T, 32).total;
}

static int unet_fwd_impl(const m2h_unet_weights* wts, const float* mix, const float* masks, const float* cls_val, float* out, int B, int F,
                         int T, void* workspace, size_t workspace_bytes, void* const* events, m2h_stream stream) {
  M2H_REQUIRE(wts && mix && out && workspace, "unet_fwd: null pointer");
  M2H_REQUIRE(B > 0 && F == 512 && T > 0 && T % 32 == 0, "unet_fwd: F must be 512 and T a multiple of 32 (got %d x %d)", F, T);
  M2H_REQUIRE(wts->n_out == 32 || wts->n_out == 16, "unet_fwd: n_out must be 32 or 16");
  M2H_REQUIRE((wts->cls_table == nullptr) == (cls_val == nullptr), "unet_fwd: class table / value mismatch");
  M2H_REQUIRE(wts->math_mode >= 0 && wts->math_mode <= 2, "unet_fwd: math_mode must be 0 (thread's), 1 (fp32) or 2 (bf16x3)");
  const int math = wts->math_mode == 0 ? tl_math_mode : wts->math_mode - 1;
  M2H_REQUIRE(!wts->weights_split32 || math == 1, "unet_fwd: split32 weights need the bf16x3 math mode");
  const UnetLayout L = unet_layout(B, F, T, wts->n_out);
  M2H_REQUIRE(workspace_bytes >= L.total, "unet_fwd: workspace too small (%zu < %zu)", workspace_bytes, L.total);
  char* ws = static_cast<char*>(workspace);
  float* x0 = reinterpret_cast<float*>(ws + L.x0);
  float* e[5];
  float* d[5];
  for (int i = 0; i < 5; ++i) {
    e[i] = reinterpret_cast<float*>(ws + L.e[i]);
    d[i] = reinterpret_cast<float*>(ws + L.d[i]);
  }
  void* sk = ws + L.splitk;
  const size_t skb = L.cls - L.splitk;
  hipStream_t st = as_stream(stream);
  int ev = 0, stage_no = 0;
  auto mark = [&]() -> int {
    if (stage_no > 0 && stage_no <= 11) tl_unet_stage[stage_no - 1] = tl_last_launch;   // the stage that just went out
    ++stage_no;
    tl_last_launch = "(no launch: fused into the next stage)";
    if (events == nullptr) return 0;
    const hipError_t err = hipEventRecord(static_cast<hipEvent_t>(events[ev++]), st);
    return err == hipSuccess ? 0 : fail((int)err, "unet_fwd: hipEventRecord failed: %s", hipGetErrorString(err));
  };
  // with split32 weights every intermediate tensor lives in the split32 layout: producers write it, consumers copy it to LDS
  const int sp = wts->weights_split32 ? 1 : 0;
  const int fmt_math = math == 1 ? M2H_FMT_MATH_BF16X3 : M2H_FMT_MATH_FP32;   // every launch of this call pinned to its arithmetic
  const int fmt_mid = fmt_math | (sp ? (M2H_FMT_SRC_SPLIT | M2H_FMT_W_SPLIT | M2H_FMT_DST_SPLIT) : 0);
  const int fmt_last = fmt_math | (sp ? (M2H_FMT_SRC_SPLIT | M2H_FMT_W_SPLIT) : 0);
  int rc = mark();
  if (rc) return rc;
  // slice + first encoder stage as one strip-walker launch (csrc/conv_strip.hip) where its shape conditions hold
  // (the strip walkers index pixels with 32 bits: batches beyond that fall back to the tiled engines instead of failing)
  const bool strip_fits = (size_t)B * 512 * T * 2 < (1ull << 31);
  const bool strip0 = sp && wts->down0_strip != nullptr && T % 64 == 0 && g_strip >= 0 && strip_fits;
  const int cls_kind = cls_val != nullptr ? wts->cls_kind : 0;
  M2H_REQUIRE(cls_kind >= 0 && cls_kind <= 2, "unet_fwd: cls_kind must be 0, 1 or 2");
  if (!strip0) {
    if (cls_kind != 0) {   // the slice launch also makes the class plane's values from the raw target_class
      float* cls_out = reinterpret_cast<float*>(ws + L.cls);
      rc = sep_slice_input_cls(mix, masks, x0, B, F, T, sp, cls_val, cls_kind, cls_out, st);
      cls_val = cls_out;
    } else {
      rc = m2h_sep_slice_input_fmt(mix, masks, x0, B, F, T, 2, sp, stream);
    }
    if (rc) return rc;
  }
  if ((rc = mark())) return rc;
  int h = F / 16, w = T;
  const float* cur = x0;
  for (int i = 0; i < 5; ++i) {
    M2H_REQUIRE(h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0, "unet_fwd: stage %d input %d x %d", i, h, w);
    if (i == 0 && strip0) {
      M2H_REQUIRE(F == 512 && (wts->cls_table == nullptr) == (cls_val == nullptr), "unet_fwd: strip stage: F must be 512, class table / value mismatch");
      if ((rc = launch_strip_conv1(mix, masks, wts->down0_strip, wts->down_scale[0], wts->down_shift[0], wts->cls_table, cls_val, e[0], B, T, 0.2f, st,
                                   cls_kind)))
        return rc;
      if ((rc = mark())) return rc;
      cur = e[0];
      h /= 2; w /= 2;
      continue;
    }
    m2h_conv_args a = down_args(cur, wts->down_w[i], wts->down_scale[i], wts->down_shift[i], i == 0 ? wts->cls_table : nullptr,
                                i == 0 ? cls_val : nullptr, e[i], B, h, w, kEnc[i], kEnc[i + 1]);
    a.workspace = sk; a.workspace_bytes = skb; a.operand_format = fmt_mid;
    if ((rc = conv_igemm_f32(a, st))) return rc;
    if ((rc = mark())) return rc;
    cur = e[i];
    h /= 2; w /= 2;
  }
  const int c0[5] = {512, 512, 256, 128, 64}, c1[5] = {0, 512, 256, 128, 64};
  const int dco[5] = {512, 256, 128, 64, wts->n_out};
  for (int i = 0; i < 4; ++i) {
    const float* skip = i == 0 ? nullptr : e[4 - i];
    m2h_conv_args a = up_args(cur, skip, wts->up_w[i], wts->up_scale[i], wts->up_shift[i], d[i], B, h, w, c0[i], c1[i], dco[i]);
    a.workspace = sk; a.workspace_bytes = skb; a.operand_format = fmt_mid;
    if ((rc = conv_igemm_f32(a, st))) return rc;
    if ((rc = mark())) return rc;
    cur = d[i];
    h *= 2; w *= 2;
  }
  // last stage + 1x1 head + de-slice in one kernel: the strip walker where its shape conditions hold
  if (sp && T % 64 == 0 && g_strip >= 0 && wts->down0_strip != nullptr && strip_fits && (long)B * h * w * 64 < (1L << 31)) {
    M2H_REQUIRE(wts->head_w != nullptr && wts->head_b != nullptr, "unet_fwd: null head");
    if ((rc = m2h_strip_last_fwd(cur, e[0], wts->up_w[4], wts->up_scale[4], wts->up_shift[4], wts->head_w, wts->head_b, out, B, h, w, dco[4], stream)))
      return rc;
    return mark();
  }
  m2h_conv_args a = up_args(cur, e[0], wts->up_w[4], wts->up_scale[4], wts->up_shift[4], out, B, h, w, c0[4], c1[4], dco[4]);
  a.out_mode = M2H_OUT_DESLICE;
  a.head_w = wts->head_w;
  a.head_b = wts->head_b;
  a.operand_format = fmt_last;
  M2H_REQUIRE(a.head_w != nullptr && a.head_b != nullptr, "unet_fwd: null head");
  if ((rc = conv_igemm_f32(a, st))) return rc;
  return mark();
}

int m2h_unet_fwd(const m2h_unet_weights* wts, const float* mix, const float* masks, const float* cls_val, float* out, int B, int F,
                 int T, void* workspace, size_t workspace_bytes, m2h_stream stream) {
  return unet_fwd_impl(wts, mix, masks, cls_val, out, B, F, T, workspace, workspace_bytes, nullptr, stream);
}

int m2h_unet_fwd_events(const m2h_unet_weights* wts, const float* mix, const float* masks, const float* cls_val, float* out, int B, int F,
                        int T, void* workspace, size_t workspace_bytes, void* const* events, int n_events, m2h_stream stream) {
  M2H_REQUIRE(events != nullptr && n_events == M2H_UNET_FWD_EVENTS, "unet_fwd_events: need %d events", M2H_UNET_FWD_EVENTS);
  for (int i = 0; i < n_events; ++i) M2H_REQUIRE(events[i] != nullptr, "unet_fwd_events: null event %d", i);
  return unet_fwd_impl(wts, mix, masks, cls_val, out, B, F, T, workspace, workspace_bytes, events, stream);
}

}  // extern "C"
