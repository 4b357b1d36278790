// HBM-bound layout / pointwise kernels of the separator path (gfx950).
//   sep_slice_input  : K1 + K2 of SURVEY section 2.2 (separator_cnn.py:73-90)
//   pack_conv_weight / pack_convT_weight / unet_class_table / fold_bn : one-off weight preparation
#include "m2h_internal.h"

namespace m2h {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// out[b][h][t][c*16+s] = f(mix[b][s*Hs+h][t][c])   (Hs = F/16)
// One thread produces 4 consecutive output channels (same c, s0..s0+3) as one 16-byte store; lanes walk
// (channel-group, t) so a wave writes 8 x 128-byte pixel rows... reads are 4-byte gathers from 4 frequency
// rows (served by L2: every 32-byte sector fetched is fully consumed by neighbouring lanes).
__global__ __launch_bounds__(256) void sep_slice_input_kernel(const float* __restrict__ mix, const float* __restrict__ masks,
                                                              float* __restrict__ out, int B, int F, int T, int C) {
  const int Hs = F >> 4;
  const int CG = (16 * C) >> 2;  // channel groups of 4
  const size_t total = (size_t)B * Hs * T * CG;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % CG);
    size_t r = i / CG;
    const int t = (int)(r % T);
    r /= T;
    const int h = (int)(r % Hs);
    const int b = (int)(r / Hs);
    const int n0 = cg * 4;
    const int c = n0 >> 4, s0 = n0 & 15;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t off = (((size_t)b * F + (size_t)(s0 + j) * Hs + h) * T + t) * C + c;
      float x = mix[off];
      if (masks != nullptr) {
        // log1p(clamp(mask * (exp(mix) - 1), min=0))   (separator_cnn.py:77-79)
        const float e = expf(x) - 1.f;
        x = log1pf(fmaxf(masks[off] * e, 0.f));
      }
      v[j] = x;
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
  }
}

// Tiled version for C == 2 (binaural spectrograms): a block takes one (b, h) and 64 time frames.  The 16 frequency rows it
// needs (s*Hs + h) are each one contiguous 512-byte run -> coalesced 16-byte loads into an LDS tile [16][132]; the output is
// then gathered from LDS and written as whole 128-byte pixel rows (8 KB contiguous per block).  The direct version above
// reads with 4-byte gathers (64-byte pieces per wave instruction) and ran at 3.5 TB/s of the ~5.5 TB/s a copy reaches.
typedef __bf16 bf16x4_s __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void sep_slice_input_c2_kernel(const float* __restrict__ mix, const float* __restrict__ masks,
                                                                 float* __restrict__ out, int B, int F, int T, int split_out,
                                                                 const void* __restrict__ cls_raw = nullptr, int cls_kind = 0,
                                                                 float* __restrict__ cls_out = nullptr) {
  constexpr int TT = 64, LD = 132;
  if (cls_out != nullptr && blockIdx.x == 0)   // the class plane's value target_class.float() + 1 (separator_cnn.py:96) for the first conv's epilogue
    for (int b = threadIdx.x; b < B; b += 256)
      cls_out[b] = (cls_kind == 2 ? (float)static_cast<const long long*>(cls_raw)[b] : static_cast<const float*>(cls_raw)[b]) + 1.f;
  __shared__ __attribute__((aligned(16))) float tile[16 * LD];
  const int Hs = F >> 4;
  const int tiles_t = (T + TT - 1) / TT;
  int blk = blockIdx.x;
  const int tt = blk % tiles_t;
  blk /= tiles_t;
  const int h = blk % Hs;
  const int b = blk / Hs;
  const int t0 = tt * TT;
  const int tid = threadIdx.x;
  {
    const int s = tid >> 4, q = tid & 15;
    const size_t rowbase = (((size_t)b * F + (size_t)s * Hs + h) * T + t0) * 2;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f4 = q + 16 * u;              // 16-byte segment inside the 64-frame run: frames 2*f4, 2*f4+1
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (t0 + 2 * f4 + 1 < T) {              // T is even: a segment is entirely inside or outside
        v = *reinterpret_cast<const f32x4*>(mix + rowbase + f4 * 4);
        if (masks != nullptr) {
          const f32x4 m = *reinterpret_cast<const f32x4*>(masks + rowbase + f4 * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = masked_log_mag(v[j], m[j]);
        }
      }
      *reinterpret_cast<f32x4*>(&tile[s * LD + f4 * 4]) = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int idx = tid + 256 * u;
    const int cg = idx & 7, t = idx >> 3;     // 8 channel groups of 4 per pixel, 64 pixels
    if (t0 + t >= T) continue;
    const int n0 = cg * 4;
    const int c = n0 >> 4, s0 = n0 & 15;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = tile[(s0 + j) * LD + 2 * t + c];
    float* prow = out + (((size_t)b * Hs + h) * T + t0 + t) * 32;
    if (split_out) {   // split32 layout: the pixel's 32 channels are one group; this thread owns values n0 .. n0+3
      const bf16x4_s hi = __builtin_convertvector(v, bf16x4_s);
      const f32x4 hf = __builtin_convertvector(hi, f32x4);
      const bf16x4_s lo = __builtin_convertvector(v - hf, bf16x4_s);
      char* base = reinterpret_cast<char*>(prow) + cg * 8;
      *reinterpret_cast<bf16x4_s*>(base) = hi;
      *reinterpret_cast<bf16x4_s*>(base + 64) = lo;
    } else {
      *reinterpret_cast<f32x4*>(prow + n0) = v;
    }
  }
}

// Training-time variant: same slice, but each pixel row has ldo >= 16*C+1 channels: channel 16*C holds the (target_class+1)
// plane (separator_cnn.py:93-99 materialised so that its weight gradient falls out of the ordinary wgrad), the rest 0.
__global__ __launch_bounds__(256) void sep_slice_input_plane_kernel(const float* __restrict__ mix, const float* __restrict__ cls_val,
                                                                    float* __restrict__ out, int B, int F, int T, int C, int ldo) {
  const int Hs = F >> 4;
  const int CG = ldo >> 2;
  const int nc = 16 * C;
  const size_t total = (size_t)B * Hs * T * CG;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % CG);
    size_t r = i / CG;
    const int t = (int)(r % T);
    r /= T;
    const int h = (int)(r % Hs);
    const int b = (int)(r / Hs);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = cg * 4 + j;
      float x = 0.f;
      if (n < nc) {
        const int c = n >> 4, s = n & 15;
        x = mix[(((size_t)b * F + (size_t)s * Hs + h) * T + t) * C + c];
      } else if (n == nc) {
        x = cls_val[b];
      }
      v[j] = x;
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
  }
}

// [Co][Ci][KH][KW] -> [Co][KH][KW][cout]; channels >= cu are zero
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int KH, int KW, int cu, int cout) {
  const size_t total = (size_t)Co * KH * KW * cout;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cout);
    size_t r = i / cout;
    const int kw = (int)(r % KW);
    r /= KW;
    const int kh = (int)(r % KH);
    const int co = (int)(r / KH);
    wp[i] = ci < cu ? w[(((size_t)co * Ci + ci) * KH + kh) * KW + kw] : 0.f;
  }
}

// [Ci][Co][4][4] -> [phase = ph*2+pw][Co][th][tw][Ci];  kh = (ph ? 2 : 1) + th * (ph ? -2 : 2)
__global__ void pack_convT_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Ci, int Co) {
  const size_t total = (size_t)4 * Co * 4 * Ci;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ci);
    size_t r = i / Ci;
    const int tw = (int)(r & 1);
    const int th = (int)((r >> 1) & 1);
    r >>= 2;
    const int co = (int)(r % Co);
    const int phase = (int)(r / Co);
    const int ph = phase >> 1, pw = phase & 1;
    const int kh = (ph ? 2 : 1) + th * (ph ? -2 : 2);
    const int kw = (pw ? 2 : 1) + tw * (pw ? -2 : 2);
    wp[i] = w[(((size_t)ci * Co + co) * 4 + kh) * 4 + kw];
  }
}

// table[(ch*3+cw)*Co + co] = sum_{kh valid for ch} sum_{kw valid for cw} w[co][plane][kh][kw]
// 4x4 stride-2 pad-1: the first output row/col misses tap 0, the last misses tap 3.
__global__ void unet_class_table_kernel(const float* __restrict__ w, float* __restrict__ table, int Co, int Ci, int plane) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * Co) return;
  const int co = i % Co;
  const int cls = i / Co;
  const int ch = cls / 3, cw = cls % 3;
  const int h0 = (ch == 0) ? 1 : 0, h1 = (ch == 2) ? 3 : 4;
  const int w0 = (cw == 0) ? 1 : 0, w1 = (cw == 2) ? 3 : 4;
  float s = 0.f;
  for (int kh = h0; kh < h1; ++kh)
    for (int kw = w0; kw < w1; ++kw) s += w[(((size_t)co * Ci + plane) * 4 + kh) * 4 + kw];
  table[i] = s;
}

__global__ void fold_bn_kernel(const float* __restrict__ g, const float* __restrict__ b, const float* __restrict__ mean,
                               const float* __restrict__ var, float eps, float* __restrict__ scale, float* __restrict__ shift, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C) return;
  const float s = g[i] / sqrtf(var[i] + eps);
  scale[i] = s;
  shift[i] = b[i] - mean[i] * s;
}

static inline unsigned grid_for(size_t total, int block = 256, unsigned cap = 256 * 8) {
  size_t g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// fp32 -> split32 (include/m2h.h): thread = one 16-byte segment (4 values) of a 32-value group
typedef __bf16 bf16x4_l __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void split32_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t nseg) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nseg; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + i * 4);
    const bf16x4_l hi = __builtin_convertvector(v, bf16x4_l);
    const f32x4 hf = __builtin_convertvector(hi, f32x4);
    const bf16x4_l lo = __builtin_convertvector(v - hf, bf16x4_l);
    const size_t g = i >> 3, sg = i & 7;
    char* base = reinterpret_cast<char*>(dst + g * 32) + sg * 8;
    *reinterpret_cast<bf16x4_l*>(base) = hi;
    *reinterpret_cast<bf16x4_l*>(base + 64) = lo;
  }
}

}  // namespace m2h

using namespace m2h;

namespace m2h {
// the runner's slice launch (C == 2, even T) that also turns the raw target_class into the class plane's values (m2h_unet_weights.cls_kind)
int sep_slice_input_cls(const float* mix, const float* masks, float* out, int B, int F, int T, int split_out, const void* cls_raw, int cls_kind,
                        float* cls_out, hipStream_t st) {
  const long nblk = (long)B * (F / 16) * ((T + 63) / 64);
  M2H_REQUIRE(mix && out && cls_raw && cls_out && (cls_kind == 1 || cls_kind == 2) && F % 16 == 0 && T % 2 == 0 && nblk > 0 && nblk <= 0x7fffffffL,
              "sep_slice_input_cls: bad arguments");
  M2H_LAUNCH(sep_slice_input_c2_kernel, dim3((unsigned)nblk), dim3(256), 0, st, mix, masks, out, B, F, T, split_out, cls_raw, cls_kind, cls_out);
  return launch_status("sep_slice_input");
}
}  // namespace m2h

extern "C" {

int m2h_split32(const float* src, float* dst, size_t count, m2h_stream stream) {
  M2H_REQUIRE(src != nullptr && dst != nullptr && src != dst && count > 0 && count % 32 == 0, "split32: bad arguments (count %% 32, out of place)");
  M2H_LAUNCH(split32_kernel, dim3(grid_for(count / 4, 256, 256 * 16)), dim3(256), 0, as_stream(stream), src, dst, count / 4);
  return launch_status("split32");
}

int m2h_sep_slice_input_fmt(const float* mix, const float* masks, float* out, int B, int F, int T, int C, int split_out, m2h_stream stream) {
  M2H_REQUIRE(mix != nullptr && out != nullptr, "sep_slice_input: null pointer");
  M2H_REQUIRE(B > 0 && F > 0 && T > 0 && C > 0, "sep_slice_input: non-positive size");
  M2H_REQUIRE(F % 16 == 0, "sep_slice_input: F (%d) must be a multiple of 16", F);
  M2H_REQUIRE((16 * C) % 4 == 0, "sep_slice_input: 16*C must be a multiple of 4");
  const long nblk = (long)B * (F / 16) * ((T + 63) / 64);
  if (C == 2 && T % 2 == 0 && nblk <= 0x7fffffffL) {
    M2H_LAUNCH(sep_slice_input_c2_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), mix, masks, out, B, F, T, split_out);
    return launch_status("sep_slice_input");
  }
  M2H_REQUIRE(!split_out, "sep_slice_input: split32 output needs C == 2 and an even T");
  const size_t total = (size_t)B * (F / 16) * T * (16 * C / 4);
  M2H_LAUNCH(sep_slice_input_kernel, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, as_stream(stream), mix, masks, out, B, F, T, C);
  return launch_status("sep_slice_input");
}

int m2h_sep_slice_input(const float* mix, const float* masks, float* out, int B, int F, int T, int C, m2h_stream stream) {
  return m2h_sep_slice_input_fmt(mix, masks, out, B, F, T, C, 0, stream);
}

int m2h_sep_slice_input_plane(const float* mix, const float* cls_val, float* out, int B, int F, int T, int C, int ldo, m2h_stream stream) {
  M2H_REQUIRE(mix && cls_val && out && B > 0 && F > 0 && T > 0 && C > 0 && F % 16 == 0, "sep_slice_input_plane: bad arguments");
  M2H_REQUIRE(ldo >= 16 * C + 1 && ldo % 4 == 0, "sep_slice_input_plane: ldo must be a multiple of 4 and > 16*C");
  const size_t total = (size_t)B * (F / 16) * T * (ldo / 4);
  M2H_LAUNCH(sep_slice_input_plane_kernel, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, as_stream(stream), mix, cls_val, out, B, F, T, C, ldo);
  return launch_status("sep_slice_input_plane");
}

int m2h_pack_conv_weight_ex(const float* w, float* wp, int Co, int Ci, int KH, int KW, int ci_used, int ci_out, m2h_stream stream) {
  M2H_REQUIRE(w != nullptr && wp != nullptr, "pack_conv_weight: null pointer");
  M2H_REQUIRE(Co > 0 && Ci > 0 && KH > 0 && KW > 0 && ci_used > 0 && ci_used <= Ci && ci_out >= ci_used, "pack_conv_weight: bad sizes");
  const size_t total = (size_t)Co * KH * KW * ci_out;
  M2H_LAUNCH(pack_conv_weight_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), w, wp, Co, Ci, KH, KW, ci_used, ci_out);
  return launch_status("pack_conv_weight");
}

int m2h_pack_conv_weight(const float* w, float* wp, int Co, int Ci, int KH, int KW, int ci_used, m2h_stream stream) {
  return m2h_pack_conv_weight_ex(w, wp, Co, Ci, KH, KW, ci_used, ci_used, stream);
}

int m2h_pack_convT_weight(const float* w, float* wp, int Ci, int Co, m2h_stream stream) {
  M2H_REQUIRE(w != nullptr && wp != nullptr, "pack_convT_weight: null pointer");
  M2H_REQUIRE(Co > 0 && Ci > 0, "pack_convT_weight: bad sizes");
  const size_t total = (size_t)16 * Co * Ci;
  M2H_LAUNCH(pack_convT_weight_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), w, wp, Ci, Co);
  return launch_status("pack_convT_weight");
}

int m2h_unet_class_table(const float* w, float* table, int Co, int Ci, int plane, m2h_stream stream) {
  M2H_REQUIRE(w != nullptr && table != nullptr, "unet_class_table: null pointer");
  M2H_REQUIRE(Co > 0 && Ci > 0 && plane >= 0 && plane < Ci, "unet_class_table: bad sizes");
  M2H_LAUNCH(unet_class_table_kernel, dim3((9 * Co + 255) / 256), dim3(256), 0, as_stream(stream), w, table, Co, Ci, plane);
  return launch_status("unet_class_table");
}

int m2h_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale,
                float* shift, int C, m2h_stream stream) {
  M2H_REQUIRE(gamma && beta && mean && var && scale && shift, "fold_bn: null pointer");
  M2H_REQUIRE(C > 0, "fold_bn: bad size");
  M2H_LAUNCH(fold_bn_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), gamma, beta, mean, var, eps, scale, shift, C);
  return launch_status("fold_bn");
}

}  // extern "C"
