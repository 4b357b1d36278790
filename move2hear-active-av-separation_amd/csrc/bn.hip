// Train-mode BatchNorm2d over NHWC activations (gfx950, fp32), forward and backward, fused with the U-Net activations.
// Replaces nn.BatchNorm2d(train) + LeakyReLU/ReLU of audio_separation/rl/models/separator_cnn.py:5-24 during passive
// pre-training (pretrain/passive/passive_trainer.py:211-249, BN in train mode) and their autograd.
//
// Layout: z [M][C] = the conv output viewed as rows of C channels (C contiguous).  All reductions are two ordered stages
// (row splits -> per-channel combine), so results are bit-reproducible; batch statistics use Welford/Chan combination
// (no E[x^2]-E[x]^2 cancellation).  HBM-bound: forward reads z twice and writes y once; backward reads dy, y, z twice and
// writes dz once.
#include "m2h_internal.h"

namespace m2h {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Wf {
  float n, mean, m2;
};

__device__ __forceinline__ Wf wf_combine(Wf a, Wf b) {
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  Wf r;
  r.n = a.n + b.n;
  const float d = b.mean - a.mean;
  r.mean = a.mean + d * (b.n / r.n);
  r.m2 = a.m2 + b.m2 + d * d * (a.n * b.n / r.n);
  return r;
}

// stage 1: part[split][c] = (n, mean, M2) over the split's rows
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ z, float* __restrict__ part, int M, int C, int rows_per_split) {
  constexpr int NW = 4;   // waves = row lanes
  __shared__ Wf sh[NW][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const int m0 = blockIdx.y * rows_per_split, m1 = min(M, m0 + rows_per_split);
  Wf a = {0.f, 0.f, 0.f};
  if (c < C)
    for (int m = m0 + w; m < m1; m += 8 * NW) {   // eight rows' loads in flight, then their updates in row order (the order of a one-row loop)
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = m + NW * j < m1 ? z[(size_t)(m + NW * j) * C + c] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (m + NW * j < m1) {
          a.n += 1.f;
          const float d = x[j] - a.mean;
          a.mean += d / a.n;
          a.m2 += d * (x[j] - a.mean);
        }
    }
  sh[w][threadIdx.x & 63] = a;
  __syncthreads();
  if (w == 0 && c < C) {
    Wf r = wf_combine(wf_combine(sh[0][threadIdx.x], sh[1][threadIdx.x]), wf_combine(sh[2][threadIdx.x], sh[3][threadIdx.x]));
#pragma unroll
    for (int k = 4; k < NW; k += 4)
      r = wf_combine(r, wf_combine(wf_combine(sh[k][threadIdx.x], sh[k + 1][threadIdx.x]), wf_combine(sh[k + 2][threadIdx.x], sh[k + 3][threadIdx.x])));
    float* p = part + ((size_t)blockIdx.y * C + c) * 3;
    p[0] = r.n;
    p[1] = r.mean;
    p[2] = r.m2;
  }
}

// stage 2: mean, invstd = 1/sqrt(var_biased + eps); running stats update (momentum, unbiased variance) as torch does.
// One wave per channel: lane l folds splits l, l+64, ... in order, then the 64 lane results meet in a fixed shuffle tree
// (same order every run: bit-reproducible).  A thread per channel walking all the splits -- up to 1024 dependent Welford
// combines with two divisions each, on C threads -- cost 84 us per call, 21 % of a passive training step.
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* __restrict__ part, int splits, int C, float eps, float momentum,
                                                             float* __restrict__ mean, float* __restrict__ invstd,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;  // whole wave
  Wf r = {0.f, 0.f, 0.f};
  for (int s = lane; s < splits; s += 64) {
    const float* p = part + ((size_t)s * C + c) * 3;
    Wf b = {p[0], p[1], p[2]};
    r = wf_combine(r, b);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    Wf b;
    b.n = __shfl_down(r.n, off, 64);
    b.mean = __shfl_down(r.mean, off, 64);
    b.m2 = __shfl_down(r.m2, off, 64);
    r = wf_combine(r, b);
  }
  if (lane != 0) return;
  const float var_b = r.m2 / r.n;
  mean[c] = r.mean;
  invstd[c] = 1.f / sqrtf(var_b + eps);
  if (running_mean != nullptr) {
    const float var_u = r.n > 1.f ? r.m2 / (r.n - 1.f) : var_b;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * r.mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * var_u;
  }
}

// y = act((z - mean) * invstd * gamma + beta),  act(v) = v > 0 ? v : v*slope.  C % 4 == 0.
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
                                                       float* __restrict__ y, size_t n4, int C4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const f32x4 x = *reinterpret_cast<const f32x4*>(z + i * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = (x[j] - mean[c + j]) * invstd[c + j] * gamma[c + j] + beta[c + j];
      o[j] = v > 0.f ? v : v * slope;
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = o;
  }
}

// backward stage 1: with g = dy * act'(y) and xh = (z - mean) * invstd:  part[split][0][c] = sum g, part[split][1][c] = sum g*xh
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ z,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd, float slope,
                                                             float* __restrict__ part, int M, int C, int rows_per_split) {
  constexpr int NW = 4;
  __shared__ float sh[2][NW][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const int m0 = blockIdx.y * rows_per_split, m1 = min(M, m0 + rows_per_split);
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    const float mu = mean[c], is = invstd[c];
    for (int m = m0 + w; m < m1; m += 4 * NW) {   // four rows' loads in flight, summed in row order
      float vy[4], vd[4], vz[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = m + NW * j < m1;
        const size_t i = (size_t)(ok ? m + NW * j : m) * C + c;
        vy[j] = y[i];
        vd[j] = dy[i];
        vz[j] = z[i];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (m + NW * j < m1) {
          const float g = vy[j] > 0.f ? vd[j] : vd[j] * slope;
          s0 += g;
          s1 += g * ((vz[j] - mu) * is);
        }
    }
  }
  sh[0][w][threadIdx.x & 63] = s0;
  sh[1][w][threadIdx.x & 63] = s1;
  __syncthreads();
  if (w == 0 && c < C) {
    const int t = threadIdx.x;
    float r0 = sh[0][0][t] + sh[0][1][t] + sh[0][2][t] + sh[0][3][t];
    float r1 = sh[1][0][t] + sh[1][1][t] + sh[1][2][t] + sh[1][3][t];
#pragma unroll
    for (int k = 4; k < NW; ++k) {
      r0 += sh[0][k][t];
      r1 += sh[1][k][t];
    }
    part[((size_t)blockIdx.y * 2 + 0) * C + c] = r0;
    part[((size_t)blockIdx.y * 2 + 1) * C + c] = r1;
  }
}

// one wave per channel, as bn_stats_final_kernel
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* __restrict__ part, int splits, int C, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  float s0 = 0.f, s1 = 0.f;
  for (int s = lane; s < splits; s += 64) {
    s0 += part[((size_t)s * 2 + 0) * C + c];
    s1 += part[((size_t)s * 2 + 1) * C + c];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    s0 += __shfl_down(s0, off, 64);
    s1 += __shfl_down(s1, off, 64);
  }
  if (lane == 0) {
    dbeta[c] = s0;
    dgamma[c] = s1;
  }
}

// dz = gamma * invstd * (g - dbeta/M - xh * dgamma/M)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta, float slope, float invM, float* __restrict__ dz,
                                                           size_t n4, int C4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const f32x4 gy = *reinterpret_cast<const f32x4*>(dy + i * 4);
    const f32x4 yy = *reinterpret_cast<const f32x4*>(y + i * 4);
    const f32x4 zz = *reinterpret_cast<const f32x4*>(z + i * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g = yy[j] > 0.f ? gy[j] : gy[j] * slope;
      const float xh = (zz[j] - mean[c + j]) * invstd[c + j];
      o[j] = gamma[c + j] * invstd[c + j] * (g - dbeta[c + j] * invM - xh * dgamma[c + j] * invM);
    }
    *reinterpret_cast<f32x4*>(dz + i * 4) = o;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// One launch per direction for the layers whose channel slab fits a block's REGISTERS: a block of 256 threads owns four channels
// (one float4 per row), thread t holds rows t, t + 256, ... -- the slab is read ONCE, the statistics meet in a fixed order
// (thread-sequential, then a shuffle tree per wave, then the waves in order: bit-reproducible), and the block applies them to what it
// holds.  At the passive training batch (64 clips: 64 ... 65 536 rows of 512 ... 32 channels per layer) the three launches of the
// general path are 15 us forward + 17 us backward per layer, nearly all of it launch and dependency latency.  Which layers take this
// path: bn_small_rows below.  m2h_tuning_set(37, -1) forces the three-launch path (tests, A/B), > 0 sets the row limit.
// ---------------------------------------------------------------------------------------------------------------------------------
#define g_bn_small (::m2h::tl_tuning.v[37])
constexpr int BNS_NT = 256;    // threads per block
constexpr int BNS_RMAX = 16;   // rows per thread at most: layers of up to 4 096 rows

// sum of v over the block, the same value in every thread: shuffle tree per wave, then the four waves' partials in wave order
__device__ __forceinline__ void block_sum4(float (&v)[4], float (*sh)[4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    if (lane == 0) sh[wave][k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = (sh[0][k] + sh[1][k]) + (sh[2][k] + sh[3][k]);
}

// R rows per thread (1, 4 or 16: the launch picks the smallest that covers M).  Two passes over the REGISTERS: the mean, then the
// centred sum of squares (no E[x^2] - E[x]^2 cancellation, and no per-element division as in the streaming Welford update).
template <int R>
__global__ __launch_bounds__(BNS_NT) void bn_fwd_small_kernel(const float* __restrict__ z, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float eps, float momentum, float slope, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var, float* __restrict__ mean, float* __restrict__ invstd,
                                                              float* __restrict__ y, int M, int C) {
  __shared__ float sh[2][4][4];
  const int tid = threadIdx.x;
  const int c0 = blockIdx.x * 4;
  f32x4 x[R];
#pragma unroll
  for (int j = 0; j < R; ++j)               // unconditional loads (clamped row), the select afterwards: all of them in flight together
    x[j] = *reinterpret_cast<const f32x4*>(z + (size_t)min(tid + j * BNS_NT, M - 1) * C + c0);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < R; ++j)
    if (tid + j * BNS_NT < M) {
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] += x[j][k];
    }
  block_sum4(s, sh[0]);
  const float invM = 1.f / (float)M;
  float mu[4], q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) mu[k] = s[k] * invM;
#pragma unroll
  for (int j = 0; j < R; ++j)
    if (tid + j * BNS_NT < M) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = x[j][k] - mu[k];
        q[k] += d * d;
      }
    }
  block_sum4(q, sh[1]);
  float is[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) is[k] = 1.f / sqrtf(q[k] * invM + eps);
  if (tid < 4) {
    const int c = c0 + tid;
    const float m2 = tid == 0 ? q[0] : tid == 1 ? q[1] : tid == 2 ? q[2] : q[3];
    const float mk = tid == 0 ? mu[0] : tid == 1 ? mu[1] : tid == 2 ? mu[2] : mu[3];
    mean[c] = mk;
    invstd[c] = tid == 0 ? is[0] : tid == 1 ? is[1] : tid == 2 ? is[2] : is[3];
    if (running_mean != nullptr) {
      const float var_u = M > 1 ? m2 / (float)(M - 1) : m2 * invM;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mk;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * var_u;
    }
  }
#pragma unroll
  for (int j = 0; j < R; ++j)
    if (tid + j * BNS_NT < M) {
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = (x[j][k] - mu[k]) * is[k] * ga[k] + be[k];    // bn_apply_kernel's expression
        o[k] = v > 0.f ? v : v * slope;
      }
      *reinterpret_cast<f32x4*>(y + (size_t)(tid + j * BNS_NT) * C + c0) = o;
    }
}

template <int R>
__global__ __launch_bounds__(BNS_NT) void bn_bwd_small_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ z,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, float slope, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ dz, int M, int C) {
  __shared__ float sh[2][4][4];
  const int tid = threadIdx.x;
  const int c0 = blockIdx.x * 4;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c0), is = *reinterpret_cast<const f32x4*>(invstd + c0);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0);
  f32x4 g[R], xh[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const size_t o = (size_t)min(tid + j * BNS_NT, M - 1) * C + c0;
    const f32x4 vd = *reinterpret_cast<const f32x4*>(dy + o);
    const f32x4 vy = *reinterpret_cast<const f32x4*>(y + o);
    const f32x4 vz = *reinterpret_cast<const f32x4*>(z + o);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      g[j][k] = vy[k] > 0.f ? vd[k] : vd[k] * slope;
      xh[j][k] = (vz[k] - mu[k]) * is[k];
    }
  }
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < R; ++j)
    if (tid + j * BNS_NT < M) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s0[k] += g[j][k];
        s1[k] += g[j][k] * xh[j][k];
      }
    }
  block_sum4(s0, sh[0]);
  block_sum4(s1, sh[1]);
  if (tid < 4) {
    dbeta[c0 + tid] = tid == 0 ? s0[0] : tid == 1 ? s0[1] : tid == 2 ? s0[2] : s0[3];
    dgamma[c0 + tid] = tid == 0 ? s1[0] : tid == 1 ? s1[1] : tid == 2 ? s1[2] : s1[3];
  }
  const float invM = 1.f / (float)M;
#pragma unroll
  for (int j = 0; j < R; ++j)
    if (tid + j * BNS_NT < M) {
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = ga[k] * is[k] * (g[j][k] - s0[k] * invM - xh[j][k] * s1[k] * invM);   // bn_bwd_apply_kernel's expression
      *reinterpret_cast<f32x4*>(dz + (size_t)(tid + j * BNS_NT) * C + c0) = o;
    }
}

// rows per thread of the one-launch kernels for M rows (0: the general path).  The row limit: knob 37 > 0, else 256 -- measured on
// the passive training step (profiles/r06_bn_small_ab.txt): the layers of 64 and 256 rows gain, those of 1 024 rows are even, 4 096
// and 16 384 rows lose (a block's loads touch one 128-byte line per row for 16 bytes of it).
static int bn_small_rows(int M, int C) {
  if (g_bn_small < 0 || C % 4 != 0) return 0;
  const int lim = g_bn_small > 0 ? g_bn_small : 256;
  if (M > lim || M > BNS_RMAX * BNS_NT) return 0;
  return M <= BNS_NT ? 1 : M <= 4 * BNS_NT ? 4 : 16;
}

static int bn_splits(int M, int C) {
  const int colblocks = (C + 63) / 64;
  int splits = (1024 + colblocks - 1) / colblocks;
  if (splits > (M + 31) / 32) splits = (M + 31) / 32;
  if (splits < 1) splits = 1;
  return splits;
}

static unsigned ew_grid(size_t n) {
  size_t g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace m2h

using namespace m2h;

extern "C" {

size_t m2h_bn_workspace_bytes(int M, int C) {
  if (M <= 0 || C <= 0) return 0;
  return (size_t)bn_splits(M, C) * C * 3 * sizeof(float);
}

int m2h_bn_train_fwd(const float* z, const float* gamma, const float* beta, float eps, float momentum, float slope, float* running_mean,
                     float* running_var, float* mean, float* invstd, float* y, int M, int C, float* workspace, m2h_stream stream) {
  M2H_REQUIRE(z && gamma && beta && mean && invstd && y && workspace && M > 1 && C > 0 && C % 4 == 0, "bn_train_fwd: bad arguments (C %% 4, M > 1)");
  M2H_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_fwd: running stats mismatch");
  hipStream_t st = as_stream(stream);
  if (const int r = bn_small_rows(M, C)) {
#define M2H_BN_FWD(R) M2H_LAUNCH(bn_fwd_small_kernel<R>, dim3(C / 4), dim3(BNS_NT), 0, st, z, gamma, beta, eps, momentum, slope, running_mean, running_var, mean, invstd, y, M, C)
    if (r == 1) M2H_BN_FWD(1);
    else if (r == 4) M2H_BN_FWD(4);
    else M2H_BN_FWD(16);
#undef M2H_BN_FWD
    return launch_status("bn_train_fwd (one launch)");
  }
  const int splits = bn_splits(M, C);
  const int rps = (M + splits - 1) / splits;
  M2H_LAUNCH(bn_stats_partial_kernel, dim3((C + 63) / 64, splits), dim3(256), 0, st, z, workspace, M, C, rps);
  M2H_LAUNCH(bn_stats_final_kernel, dim3((C + 3) / 4), dim3(256), 0, st, workspace, splits, C, eps, momentum, mean, invstd,
                     running_mean, running_var);
  const size_t n4 = (size_t)M * C / 4;
  M2H_LAUNCH(bn_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, st, z, mean, invstd, gamma, beta, slope, y, n4, C / 4);
  return launch_status("bn_train_fwd");
}

int m2h_bn_train_bwd(const float* dy, const float* y, const float* z, const float* mean, const float* invstd, const float* gamma, float slope,
                     float* dgamma, float* dbeta, float* dz, int M, int C, float* workspace, m2h_stream stream) {
  M2H_REQUIRE(dy && y && z && mean && invstd && gamma && dgamma && dbeta && dz && workspace && M > 1 && C > 0 && C % 4 == 0,
              "bn_train_bwd: bad arguments");
  hipStream_t st = as_stream(stream);
  if (const int r = bn_small_rows(M, C)) {
#define M2H_BN_BWD(R) M2H_LAUNCH(bn_bwd_small_kernel<R>, dim3(C / 4), dim3(BNS_NT), 0, st, dy, y, z, mean, invstd, gamma, slope, dgamma, dbeta, dz, M, C)
    if (r == 1) M2H_BN_BWD(1);
    else if (r == 4) M2H_BN_BWD(4);
    else M2H_BN_BWD(16);
#undef M2H_BN_BWD
    return launch_status("bn_train_bwd (one launch)");
  }
  const int splits = bn_splits(M, C);
  const int rps = (M + splits - 1) / splits;
  M2H_LAUNCH(bn_bwd_partial_kernel, dim3((C + 63) / 64, splits), dim3(256), 0, st, dy, y, z, mean, invstd, slope, workspace, M, C, rps);
  M2H_LAUNCH(bn_bwd_final_kernel, dim3((C + 3) / 4), dim3(256), 0, st, workspace, splits, C, dgamma, dbeta);
  const size_t n4 = (size_t)M * C / 4;
  M2H_LAUNCH(bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, st, dy, y, z, mean, invstd, gamma, dgamma, dbeta, slope,
                     1.f / (float)M, dz, n4, C / 4);
  return launch_status("bn_train_bwd");
}

}  // extern "C"
