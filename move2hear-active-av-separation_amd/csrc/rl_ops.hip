// HBM-bound / wavefront-reduction kernels of the RL path (gfx950), fp32.
//   slice_concat_input : AcousticMem / AudioCNN input glue          (memory_nets.py:40-59, audio_cnn.py:117-133)
//   visual_input       : rgb/255 (+depth) -> NHWC padded to 4 ch     (visual_cnn.py:135-150)
//   gru_gates          : GRU cell pointwise part, hidden mask fused  (rnn_state_encoder.py:74-84)
//   policy_heads       : actor/critic linear heads + categorical     (common/utils.py:16-50, rl/ppo/policy.py:15-23)
//   gae_returns        : GAE / discounted-return scan                (common/rollout_storage.py:155-180)
//   normalize_adv      : advantage normalisation                     (rl/ppo/ppo.py:75-80; ddppo_utils.py:168-190)
//   ppo_loss_fwd       : clipped surrogate + clipped value loss      (rl/ppo/ppo.py:125-157)
//   sep_rewards        : -mse/mean(gt^2) per env (+variants)         (common/env_utils.py:690-713)
//   stft_l2            : STFT-L2 distance per env                    (common/eval_metrics.py:306-366)
#include "m2h_internal.h"

namespace m2h {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum (blockDim = 256); result valid in every thread.
__device__ __forceinline__ float block_sum(float v, float* sh /* >= 4 floats */) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// same for a 1024-thread block (16 waves), fixed summation order
__device__ __forceinline__ float block_sum16(float v, float* sh /* >= 16 floats */) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += sh[i];
  return t;
}

// out[b][h][t][c*16+s] = f(virtual[b][s*Hs+h][t][c]),  virtual channel c < Ca -> a[..][c] (times mul[..][c] inside op 1),
// else b[..][c-Ca] * bscale[batch].   op: 0 none | 1 log1p(max(0, mul*(exp(a)-1))) | 2 log1p(max(0, x))
__global__ __launch_bounds__(256) void slice_concat_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ bsrc, int Cb,
                                                           const float* __restrict__ mul, const float* __restrict__ bscale, int op,
                                                           float* __restrict__ out, int B, int F, int T) {
  const int C = Ca + Cb;
  const int Hs = F >> 4;
  const int CG = (16 * C) >> 2;
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
      const size_t pix = ((size_t)b * F + (size_t)(s0 + j) * Hs + h) * T + t;
      float x;
      if (c < Ca) {
        x = a[pix * Ca + c];
        if (op == 1) x = log1pf(fmaxf(mul[pix * Ca + c] * (expf(x) - 1.f), 0.f));
      } else {
        x = bsrc[pix * Cb + (c - Ca)];
        if (bscale != nullptr) x *= bscale[b];
      }
      if (op == 2) x = log1pf(fmaxf(x, 0.f));
      v[j] = x;
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
  }
}

// rgb [B,H,W,3] (0..255) (+ depth [B,H,W,1]) -> out [B,H,W,4] = (r/255, g/255, b/255, depth or 0)
__global__ __launch_bounds__(256) void visual_input_kernel(const float* __restrict__ rgb, const float* __restrict__ depth,
                                                           float* __restrict__ out, size_t npix) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 v;
    v[0] = rgb[i * 3 + 0] / 255.0f;
    v[1] = rgb[i * 3 + 1] / 255.0f;
    v[2] = rgb[i * 3 + 2] / 255.0f;
    v[3] = depth != nullptr ? depth[i] : 0.f;
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// GRU cell pointwise part.  gi = x W_ih^T + b_ih  [M,3H];  gh_raw = h W_hh^T (no bias, UNMASKED h)  [M,3H]
// mask m[row] in {0,1}: (h*m) W^T = m * (h W^T), so the hidden mask is applied here.  Gate order r,z,n.
__global__ __launch_bounds__(256) void gru_gates_kernel(const float* __restrict__ gi, const float* __restrict__ gh_raw,
                                                        const float* __restrict__ bhh, const float* __restrict__ hprev,
                                                        const float* __restrict__ mask, float* __restrict__ hout, int M, int H) {
  const int total = M * H;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / H, j = i - row * H;
    const float m = mask != nullptr ? mask[row] : 1.f;
    const float* gir = gi + (size_t)row * 3 * H;
    const float* ghr = gh_raw + (size_t)row * 3 * H;
    const float r = sigmoidf_(gir[j] + (m * ghr[j] + bhh[j]));
    const float z = sigmoidf_(gir[H + j] + (m * ghr[H + j] + bhh[H + j]));
    const float n = tanhf(gir[2 * H + j] + r * (m * ghr[2 * H + j] + bhh[2 * H + j]));
    const float hp = m * hprev[i];
    hout[i] = (1.f - z) * n + z * hp;
  }
}

typedef float f32x4_r __attribute__((ext_vector_type(4)));

// Skinny dot products on the matrix pipe: D[m][r] = sum_k X[m][k] * Wr[k] for 16 rows m of X and 16 weight rows r, both operands
// straight from global memory into v_mfma_f32_16x16x4_f32 registers (lane (row i, k-quarter kq) loads 16 bytes of its row = four
// consecutive MFMAs' worth; no LDS staging, no barrier in the k-loop).  The block's NW waves split the K/16 steps (a block's latency is
// its longest wave's chain of dependent loads: 16 waves quarter it against 4 for the long-K products); the caller
// sums their partial tiles through LDS in a fixed order (wave_tile_sum).  Returns acc[e] = D[m = kq*4 + e][r = i] over this wave's steps.
template <int NW = 4>
__device__ __forceinline__ f32x4_r skinny_dot16(const float* __restrict__ xrow, const float* __restrict__ wrow, int steps, int wave) {
  const int s0 = (steps * wave) / NW, s1 = (steps * (wave + 1)) / NW;   // the block's NW waves split the K / 16 steps
  f32x4_r acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int s = s0; s < s1; ++s) {
    const f32x4_r a = *reinterpret_cast<const f32x4_r*>(xrow + 16 * s);
    const f32x4_r b = *reinterpret_cast<const f32x4_r*>(wrow + 16 * s);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
  }
  return acc;
}

// sum over the NW waves' partial tiles, pairwise in a fixed tree (bit-reproducible)
template <int NW>
__device__ __forceinline__ float wave_tile_sum(const float (*R)[16][17], int m, int r) {
  float t[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) t[w] = R[w][m][r];
#pragma unroll
  for (int n = NW; n > 1; n >>= 1)
#pragma unroll
    for (int w = 0; w < n / 2; ++w) t[w] = t[2 * w] + t[2 * w + 1];
  return t[0];
}

// One GRU time step for a handful of rows (M <= 16: the rollout's 14 envs, one step of the update's sequence pass) in ONE
// launch: recurrent product gh_raw = h W_hh^T and the gate math of gru_gates_kernel.  At this size the product is a weight
// stream (3H x H floats, 3 MB at H = 512) against 14 rows: a tiled-GEMM launch + its split-K reduce + the gate kernel spend
// ~29 us on it, almost all launch and pipeline latency.  A block owns GRU_U hidden units = 3*GRU_U weight rows (r, z, n) and
// runs their dot products with the 16 state rows through skinny_dot16; the three gates of a unit meet through LDS.  gh_raw is
// written too (the backward pass reads it).  (First version: weights and state staged in LDS, one thread per (row, env) dot
// product with broadcast LDS reads -- 12 us, bound by LDS bandwidth.)
constexpr int GRU_U = 4;     // hidden units per block
constexpr int GRU_E = 16;    // env slots per block (M <= 16)
constexpr int GS_NW = 8;     // waves per block: the 32 steps of the H-long products are 4 per wave
__global__ __launch_bounds__(64 * GS_NW) void gru_step_kernel(const float* __restrict__ gi, const float* __restrict__ whh,
                                                       const float* __restrict__ bhh, const float* __restrict__ hprev,
                                                       const float* __restrict__ mask, float* __restrict__ gh_raw,
                                                       float* __restrict__ hout, int M, int H) {
  __shared__ float R[GS_NW][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int j0 = blockIdx.x * GRU_U;
  const int r = min(i, 3 * GRU_U - 1);                         // weight row of this lane: gate r / GRU_U, unit r % GRU_U
  const float* wrow = whh + ((size_t)(r / GRU_U) * H + j0 + (r % GRU_U)) * H + 4 * kq;
  const float* xrow = hprev + (size_t)min(i, M - 1) * H + 4 * kq;
  // gate inputs of this thread's (unit, env) pair, fetched before the dot products
  const int gu = (tid >> 4) & (GRU_U - 1), ge = min(tid & 15, M - 1), gj = j0 + gu;
  const float gmask = mask != nullptr ? mask[ge] : 1.f;
  const float gi_r = gi[(size_t)ge * 3 * H + gj], gi_z = gi[(size_t)ge * 3 * H + H + gj], gi_n = gi[(size_t)ge * 3 * H + 2 * H + gj];
  const float b_r = bhh[gj], b_z = bhh[H + gj], b_n = bhh[2 * H + gj];
  const float hp_raw = hprev[(size_t)ge * H + gj];
  const f32x4_r acc = skinny_dot16<GS_NW>(xrow, wrow, H >> 4, wave);
#pragma unroll
  for (int e = 0; e < 4; ++e) R[wave][kq * 4 + e][i] = acc[e];   // [m][weight row]
  __syncthreads();
  if (tid < 3 * GRU_U * GRU_E) {                                  // gh_raw: thread (weight row rr, env m)
    const int rr = tid >> 4, m = tid & 15;
    if (m < M) gh_raw[(size_t)m * 3 * H + (size_t)(rr / GRU_U) * H + j0 + (rr % GRU_U)] = wave_tile_sum<GS_NW>(R, m, rr);
  }
  if (tid < GRU_U * GRU_E && (tid & 15) < M) {
    const int u = tid >> 4, m = tid & 15;
    auto tot = [&](int rr) { return wave_tile_sum<GS_NW>(R, m, rr); };
    const float gr = tot(u), gz = tot(GRU_U + u), gn = tot(2 * GRU_U + u);
    const float rg = sigmoidf_(gi_r + (gmask * gr + b_r));
    const float z = sigmoidf_(gi_z + (gmask * gz + b_z));
    const float n = tanhf(gi_n + rg * (gmask * gn + b_n));
    hout[(size_t)m * H + j0 + u] = (1.f - z) * n + z * (gmask * hp_raw);
  }
}

// The whole GRU cell of a no-grad single step (rnn_state_encoder.py:74-84 at the rollout width, M <= 16 rows) in ONE launch: the input
// projection gi = x W_ih^T + b_ih is the same kind of weight stream as the recurrent product (3H x I floats, 9.4 MB at I = 1536),
// so the block that owns GRU_U hidden units runs the dot products of its 3*GRU_U rows of BOTH matrices through skinny_dot16 and
// applies the gates; nothing but the new state is written (no backward pass follows a no-grad step).  Replaces the projection GEMM +
// gru_step_kernel of the rollout step.
constexpr int GC_NW = 16;    // waves per block of gru_cell_kernel: the 96 + 32 steps of the two products are 8 per wave
__global__ __launch_bounds__(64 * GC_NW) void gru_cell_kernel(const float* __restrict__ x, const float* __restrict__ wih, const float* __restrict__ bih,
                                                       const float* __restrict__ whh, const float* __restrict__ bhh,
                                                       const float* __restrict__ hprev, const float* __restrict__ mask,
                                                       float* __restrict__ hout, int M, int I, int H) {
  __shared__ float R[2][GC_NW][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int j0 = blockIdx.x * GRU_U;
  const int r = min(i, 3 * GRU_U - 1);                         // weight row of this lane: gate r / GRU_U, unit r % GRU_U
  const size_t wr = (size_t)(r / GRU_U) * H + j0 + (r % GRU_U);
  const int row = min(i, M - 1);
  const int gu = (tid >> 4) & (GRU_U - 1), ge = min(tid & 15, M - 1), gj = j0 + gu;
  const float gmask = mask != nullptr ? mask[ge] : 1.f;
  const float bi_r = bih[gj], bi_z = bih[H + gj], bi_n = bih[2 * H + gj];
  const float b_r = bhh[gj], b_z = bhh[H + gj], b_n = bhh[2 * H + gj];
  const float hp_raw = hprev[(size_t)ge * H + gj];
  const f32x4_r ai = skinny_dot16<GC_NW>(x + (size_t)row * I + 4 * kq, wih + wr * I + 4 * kq, I >> 4, wave);
  const f32x4_r ah = skinny_dot16<GC_NW>(hprev + (size_t)row * H + 4 * kq, whh + wr * H + 4 * kq, H >> 4, wave);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    R[0][wave][kq * 4 + e][i] = ai[e];   // [m][weight row]
    R[1][wave][kq * 4 + e][i] = ah[e];
  }
  __syncthreads();
  if (tid < GRU_U * GRU_E && (tid & 15) < M) {
    const int u = tid >> 4, m = tid & 15;
    auto tot = [&](int w, int rr) { return wave_tile_sum<GC_NW>(R[w], m, rr); };
    const float gi_r = tot(0, u) + bi_r, gi_z = tot(0, GRU_U + u) + bi_z, gi_n = tot(0, 2 * GRU_U + u) + bi_n;
    const float gr = tot(1, u), gz = tot(1, GRU_U + u), gn = tot(1, 2 * GRU_U + u);
    const float rg = sigmoidf_(gi_r + (gmask * gr + b_r));
    const float z = sigmoidf_(gi_z + (gmask * gz + b_z));
    const float n = tanhf(gi_n + rg * (gmask * gn + b_n));
    hout[(size_t)m * H + j0 + u] = (1.f - z) * n + z * (gmask * hp_raw);
  }
}

// LSTM cell, pointwise part (the class's rnn_type = "LSTM" variant, rnn_state_encoder.py:10-34,49-69; nn.LSTM's gate order i, f, g, o):
// pre = gi + gh  [M][4H] (gi = x W_ih^T + b_ih, gh = (h m) W_hh^T + b_hh);  c' = sigma(f) (c m) + sigma(i) tanh(g);  h' = sigma(o) tanh(c').
// The reset mask m[row] multiplies the carried cell state here (the hidden state's mask is applied before its product: _mask_hidden, :63-69).
// gates_out [M][4H] keeps the ACTIVATED gates for the backward.  One thread per (row, unit).
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ cprev,
                                                        const float* __restrict__ mask, float* __restrict__ hout, float* __restrict__ cout,
                                                        float* __restrict__ gates_out, int M, int H) {
  const size_t total = (size_t)M * H;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(idx / H), j = (int)(idx - (size_t)row * H);
    const size_t b = (size_t)row * 4 * H;
    const float ig = sigmoidf_(gi[b + j] + gh[b + j]);
    const float fg = sigmoidf_(gi[b + H + j] + gh[b + H + j]);
    const float gg = tanhf(gi[b + 2 * H + j] + gh[b + 2 * H + j]);
    const float og = sigmoidf_(gi[b + 3 * H + j] + gh[b + 3 * H + j]);
    const float cm = cprev[idx] * mask[row];
    const float c = fg * cm + ig * gg;
    cout[idx] = c;
    hout[idx] = og * tanhf(c);
    if (gates_out != nullptr) {
      gates_out[b + j] = ig;
      gates_out[b + H + j] = fg;
      gates_out[b + 2 * H + j] = gg;
      gates_out[b + 3 * H + j] = og;
    }
  }
}

// Its backward: dh, dc = gradients of (h', c'); dpre [M][4H] = gradient of the pre-activations (of gi and gh alike), dcprev = gradient of the
// carried cell state (through the mask).  dh or dc may be NULL (zero).
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dc, const float* __restrict__ gates,
                                                            const float* __restrict__ cprev, const float* __restrict__ mask, const float* __restrict__ c,
                                                            float* __restrict__ dpre, float* __restrict__ dcprev, int M, int H) {
  const size_t total = (size_t)M * H;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(idx / H), j = (int)(idx - (size_t)row * H);
    const size_t b = (size_t)row * 4 * H;
    const float ig = gates[b + j], fg = gates[b + H + j], gg = gates[b + 2 * H + j], og = gates[b + 3 * H + j];
    const float m = mask[row];
    const float tc = tanhf(c[idx]);
    const float gh_ = dh != nullptr ? dh[idx] : 0.f;
    const float dct = (dc != nullptr ? dc[idx] : 0.f) + gh_ * og * (1.f - tc * tc);
    dpre[b + j] = dct * gg * ig * (1.f - ig);
    dpre[b + H + j] = dct * (cprev[idx] * m) * fg * (1.f - fg);
    dpre[b + 2 * H + j] = dct * ig * (1.f - gg * gg);
    dpre[b + 3 * H + j] = gh_ * tc * og * (1.f - og);
    dcprev[idx] = dct * fg * m;
  }
}

// Philox4x32-10 (Salmon et al., SC'11: the counter-based generator torch's device generator is built on), one 32-bit word of the
// block at `counter` under `seed`, as Exp(1) noise: -log(u), u = (23 random bits + 0.5) / 2^23 in (0, 1) -- 23 bits, so that the sum
// is exact in fp32 (with 24 bits 0xFFFFFF + 0.5 rounds to 2^24: u = 1, noise 0, probs / noise = inf and that action wins whatever its
// probability): u in [2^-24, 1 - 2^-24], noise in [6.0e-8, 16.7].
__device__ __forceinline__ float philox_exp1(unsigned long long seed, unsigned long long counter) {
  unsigned c0 = (unsigned)counter, c1 = (unsigned)(counter >> 32), c2 = 0x6d32685fu, c3 = 0u;   // (c2: a stream tag)
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float u = ((float)(c0 >> 9) + 0.5f) * (1.0f / 8388608.0f);
  return -logf(u);
}

// One wave per row: logits = feats Wa^T + ba (A <= 8 actions), value = feats Wc^T + bc; log-softmax, softmax, entropy,
// optional log-prob of a given action.  H multiple of 64.
__global__ __launch_bounds__(256) void policy_heads_kernel(const float* __restrict__ feats, const float* __restrict__ Wa,
                                                           const float* __restrict__ ba, const float* __restrict__ Wc,
                                                           const float* __restrict__ bc, const long long* __restrict__ actions,
                                                           float* __restrict__ value, float* __restrict__ logp_all,
                                                           float* __restrict__ probs, float* __restrict__ entropy,
                                                           float* __restrict__ logp_act, int M, int H, int A,
                                                           const float* __restrict__ noise = nullptr, long long* __restrict__ act_out = nullptr,
                                                           const unsigned long long* __restrict__ rng = nullptr,
                                                           float* __restrict__ noise_out = nullptr) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float acc[9];
#pragma unroll
  for (int a = 0; a < 9; ++a) acc[a] = 0.f;
  const float* f = feats + (size_t)row * H;
  if ((H & 255) == 0) {
    // (The heads' weights live back to back in FlatAdam's buffer -- the critic's row starts 12 bytes past a 16-byte boundary behind the
    // 3-float action bias: the 16-byte loads below are then dword-aligned only, which global loads accept on gfx950, as the GRU / skinny
    // kernels' loads of flat-buffer rows rely on too.  ONE path whatever the alignment: a second one would round differently, and the
    // same weights before and after they move into the flat buffer would give losses that differ in the last bit.)
    // 16 bytes per lane and load, every load of a pass issued before the first use (the scalar loop below is a chain of H / 64 dependent
    // load rounds: 10 us of a 14-row launch)
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int k = lane * 4; k < H; k += 256) {
      const f4 x = *reinterpret_cast<const f4*>(f + k);
      const f4 wc = *reinterpret_cast<const f4*>(Wc + k);
      f4 w[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) w[a] = *reinterpret_cast<const f4*>(Wa + (size_t)(a < A ? a : 0) * H + k);
#pragma unroll
      for (int a = 0; a < 8; ++a)
        if (a < A) acc[a] += (x[0] * w[a][0] + x[1] * w[a][1]) + (x[2] * w[a][2] + x[3] * w[a][3]);
      acc[8] += (x[0] * wc[0] + x[1] * wc[1]) + (x[2] * wc[2] + x[3] * wc[3]);
    }
  } else {
    for (int k = lane; k < H; k += 64) {
      const float x = f[k];
#pragma unroll
      for (int a = 0; a < 8; ++a)
        if (a < A) acc[a] += x * Wa[a * H + k];
      acc[8] += x * Wc[k];
    }
  }
#pragma unroll
  for (int a = 0; a < 9; ++a) acc[a] = wave_sum(acc[a]);
  if (lane == 0) {
    float mx = -INFINITY;
    for (int a = 0; a < A; ++a) {
      acc[a] += ba[a];
      mx = fmaxf(mx, acc[a]);
    }
    float se = 0.f;
    for (int a = 0; a < A; ++a) se += expf(acc[a] - mx);
    const float lse = mx + logf(se);
    float ent = 0.f;
    for (int a = 0; a < A; ++a) {
      const float lp = acc[a] - lse;
      const float p = expf(lp);
      logp_all[row * A + a] = lp;
      probs[row * A + a] = p;
      ent -= lp * p;
    }
    value[row] = acc[8] + bc[0];
    entropy[row] = ent;
    if (actions != nullptr && logp_act != nullptr) logp_act[row] = logp_all[row * A + (int)actions[row]];
    if (act_out != nullptr) {
      // Policy.act in the same launch (rl/ppo/policy.py:217-225): the action -- with noise the single draw of torch.multinomial,
      // argmax(probs / Exp(1) noise), exactly as sample_actions_kernel takes it (correctly rounded division, NaN is the maximum,
      // ties keep the lowest index); without noise the mode, argmax(probs) -- and its log-probability
      // rng = {seed, counter} on the device ("fused" sampling): the Exp(1) noise of element (row, a) is -log(u), u from Philox4x32-10
      // keyed by the seed at counter + row A + a -- no generator launch in the step; the counter is advanced by step_index_advance
      // noise_out (optional, with rng): the noise drawn, [M][A] -- what the parity tests hand the CPU oracle in place of its generator's draw
      const bool draw = noise != nullptr || rng != nullptr;
      auto nz = [&](int a) {
        if (rng == nullptr) return noise[row * A + a];
        const float q = philox_exp1(rng[0], rng[1] + (unsigned long long)(row * A + a));
        if (noise_out != nullptr) noise_out[row * A + a] = q;
        return q;
      };
      int arg = 0;
      float best = draw ? expf(acc[0] - lse) / nz(0) : expf(acc[0] - lse);
      for (int a = 1; a < A; ++a) {
        const float p = expf(acc[a] - lse);
        const float q = draw ? p / nz(a) : p;
        if (!(best != best) && (q > best || q != q)) { best = q; arg = a; }
      }
      act_out[row] = arg;
      if (logp_act != nullptr) logp_act[row] = acc[arg] - lse;
    }
  }
}

__global__ void gather_logp_kernel(const float* __restrict__ logp_all, const long long* __restrict__ actions,
                                   float* __restrict__ out, int M, int A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) out[i] = logp_all[i * A + (int)actions[i]];
}

// argmax(probs / noise) per row: the single-draw path of torch.multinomial with caller-supplied Exp(1) noise (m2h_sample_actions).
// Correctly rounded fp32 division (hipcc's default), NaN counts as the maximum and ties keep the lowest index, as ATen's argmax.
__global__ void sample_actions_kernel(const float* __restrict__ probs, const float* __restrict__ noise, long long* __restrict__ actions,
                                      int M, int A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  float best = probs[i * A] / noise[i * A];
  int arg = 0;
  for (int j = 1; j < A; ++j) {
    const float q = probs[i * A + j] / noise[i * A + j];
    if (!(best != best) && (q > best || q != q)) { best = q; arg = j; }
  }
  actions[i] = arg;
}

// GAE / discounted-return scan: one thread per env, T steps backwards.  rewards [T,N], value_preds [T+1,N] (row T is
// overwritten with next_value when use_gae), masks [T+1,N], returns [T+1,N].
__global__ void gae_returns_kernel(const float* __restrict__ rewards, float* __restrict__ value_preds, const float* __restrict__ masks,
                                   const float* __restrict__ next_value, float* __restrict__ returns, int T, int N, int use_gae,
                                   float gamma, float tau) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  if (use_gae) {
    value_preds[T * N + e] = next_value[e];
    float gae = 0.f;
    for (int s = T - 1; s >= 0; --s) {
      const float m = masks[(s + 1) * N + e];
      const float delta = rewards[s * N + e] + gamma * value_preds[(s + 1) * N + e] * m - value_preds[s * N + e];
      gae = delta + gamma * tau * m * gae;
      returns[s * N + e] = gae + value_preds[s * N + e];
    }
  } else {
    returns[T * N + e] = next_value[e];
    for (int s = T - 1; s >= 0; --s)
      returns[s * N + e] = returns[(s + 1) * N + e] * gamma * masks[(s + 1) * N + e] + rewards[s * N + e];
  }
}

// adv = returns[:-1] - value_preds[:-1]; stats[0] = sum(adv), stats[1] = sum((adv-mean)^2) for the local n elements.
// mode 0: adv only.  mode 1 (local): (adv-mean)/(std_unbiased+eps).  mode 2 (distributed step A): write raw adv + local mean
// to stats[0];  the host all-reduces and calls normalize_apply.  Single block (n = T*N is small).
__global__ __launch_bounds__(256) void advantages_kernel(const float* __restrict__ returns, const float* __restrict__ value_preds,
                                                         float* __restrict__ adv, float* __restrict__ stats, int n, int mode, float eps) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float a = returns[i] - value_preds[i];
    adv[i] = a;
    s += a;
  }
  const float mean = block_sum(s, sh) / (float)n;
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = adv[i] - mean;
    q += d * d;
  }
  const float ss = block_sum(q, sh);
  if (threadIdx.x == 0 && stats != nullptr) {
    stats[0] = mean;
    stats[1] = ss / (float)n;  // biased variance around the LOCAL mean (mode 2 recomputes around the global mean)
  }
  if (mode == 1) {
    const float sd = sqrtf(ss / (float)(n - 1));
    for (int i = threadIdx.x; i < n; i += 256) adv[i] = (adv[i] - mean) / (sd + eps);
  }
}

// distributed step B: sqdiff[0] = mean((adv - gmean)^2) locally;  step C: adv = (adv - gmean)/(sqrt(gvar)+eps)
__global__ __launch_bounds__(256) void adv_sqdiff_kernel(const float* __restrict__ adv, const float* __restrict__ gmean,
                                                         float* __restrict__ out, int n) {
  __shared__ float sh[4];
  const float m = gmean[0];
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = adv[i] - m;
    q += d * d;
  }
  const float ss = block_sum(q, sh);
  if (threadIdx.x == 0) out[0] = ss / (float)n;
}

__global__ void adv_apply_kernel(float* __restrict__ adv, const float* __restrict__ gmean, const float* __restrict__ gvar, int n, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) adv[i] = (adv[i] - gmean[0]) / (sqrtf(gvar[0]) + eps);
}

// PPO losses, forward + analytic gradients w.r.t. values, action_log_probs (and the entropy coefficient is applied by the
// caller).  out[0] = value_loss, out[1] = action_loss.  Single block.
__global__ __launch_bounds__(256) void ppo_loss_kernel(const float* __restrict__ values, const float* __restrict__ logp,
                                                       const float* __restrict__ old_values, const float* __restrict__ returns,
                                                       const float* __restrict__ adv, const float* __restrict__ old_logp,
                                                       float clip_host, const float* __restrict__ clip_dev, int use_clipped_value_loss,
                                                       float* __restrict__ out, float* __restrict__ g_values, float* __restrict__ g_logp,
                                                       float value_loss_coef, const float* __restrict__ entropy, float entropy_coef, int n) {
  __shared__ float sh[4];
  const float clip = clip_dev != nullptr ? clip_dev[0] : clip_host;  // device scalar: a replayed HIP graph follows the clip decay
  float sv = 0.f, sa = 0.f, se = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    if (entropy != nullptr) se += entropy[i];
    const float ratio = expf(logp[i] - old_logp[i]);
    const float a = adv[i];
    const float s1 = ratio * a;
    const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
    const float s2 = rc * a;
    sa += fminf(s1, s2);
    // d(-min(s1,s2))/dlogp: torch.min picks s1 when s1 <= s2 (ties -> first), clamp passes gradient inside [1-c, 1+c]
    float gl;
    if (s1 <= s2) gl = -a * ratio;
    else gl = (ratio >= 1.f - clip && ratio <= 1.f + clip) ? -a * ratio : 0.f;
    const float v = values[i], ov = old_values[i], R = returns[i];
    float vl, gv;
    if (use_clipped_value_loss) {
      const float dv = v - ov;
      const float vc = ov + fminf(fmaxf(dv, -clip), clip);
      const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
      if (l1 >= l2) {  // torch.max picks the first on ties
        vl = l1;
        gv = 2.f * (v - R);
      } else {
        vl = l2;
        gv = (dv >= -clip && dv <= clip) ? 2.f * (vc - R) : 0.f;
      }
    } else {
      vl = (R - v) * (R - v);
      gv = 2.f * (v - R);
    }
    sv += vl;
    if (g_logp != nullptr) g_logp[i] = gl / (float)n;
    if (g_values != nullptr) g_values[i] = 0.5f * gv / (float)n * value_loss_coef;
  }
  const float tv = block_sum(sv, sh);
  const float ta = block_sum(sa, sh);
  const float te = block_sum(se, sh);
  if (threadIdx.x == 0) {
    out[0] = 0.5f * tv / (float)n;
    out[1] = -ta / (float)n;
    out[2] = te / (float)n;                                                      // dist_entropy (mean over rows)
    out[3] = out[0] * value_loss_coef + out[1] - out[2] * entropy_coef;          // total_loss (ppo.py:150-154)
  }
}

// Per-env reductions over L = F*T*C elements (one 1024-thread block per env, four independent loads in flight per thread: the
// 256-thread serial loop was latency-bound at 24 us for 64 KB of input):
//   stats[e][0] = sum (p-g)^2, stats[e][1] = sum g^2  -> reward_util = -(s0/L)/(s1/L)
__global__ __launch_bounds__(1024) void sq_stats_kernel(const float* __restrict__ pred, const float* __restrict__ gt_comps, int gt_stride,
                                                        int gt_off, float* __restrict__ stats, int L) {
  __shared__ float sh[16];
  const int e = blockIdx.x;
  const float* p = pred + (size_t)e * L;
  const float* g = gt_comps + (size_t)e * L * gt_stride + gt_off;
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i0 = threadIdx.x; i0 < L; i0 += 4096) {
    float pv[4], gv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 1024 * u;
      const bool ok = i < L;
      pv[u] = ok ? p[i] : 0.f;
      gv[u] = ok ? g[(size_t)i * gt_stride] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float d = pv[u] - gv[u];
      s0[u] += d * d;
      s1[u] += gv[u] * gv[u];
    }
  }
  const float t0 = block_sum16((s0[0] + s0[1]) + (s0[2] + s0[3]), sh);
  const float t1 = block_sum16((s1[0] + s1[1]) + (s1[2] + s1[3]), sh);
  if (threadIdx.x == 0) {
    stats[e * 2 + 0] = t0;
    stats[e * 2 + 1] = t1;
  }
}

// rewards[e] per override_rewards:  done -> 0;  else r = -(next0/L)/(next1/L);  quality_improvement: r -= -(cur0/L)/(cur1/L);
// otherwise r *= mult.
__global__ void rewards_from_stats_kernel(const float* __restrict__ next_stats, const float* __restrict__ cur_stats,
                                          const float* __restrict__ not_done, float* __restrict__ rewards, int N, int L, int quality,
                                          float mult) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  float r = 0.f;
  if (not_done[e] != 0.f) {
    r = -(next_stats[2 * e] / (float)L) / (next_stats[2 * e + 1] / (float)L);
    if (quality) r -= -(cur_stats[2 * e] / (float)L) / (cur_stats[2 * e + 1] / (float)L);
    else r *= mult;
  }
  rewards[e] = r;
}

// STFT-L2 per env (one 1024-thread block per env).  Both sides use the GT phase (common/eval_metrics.py STFT_L2_distance), so
// with m_g, m_p the magnitudes the reference's mean over (re, im, F, T) of (m_g cos - m_p cos)^2 and (m_g sin - m_p sin)^2 is
// (m_g - m_p)^2 (cos^2 + sin^2); the kernel evaluates (m_g - m_p)^2 directly -- equal to the term-by-term form to fp32 rounding
// (|cos^2 + sin^2 - 1| <= 2^-23), without 2 transcendental calls per element.  The phase plane is not read.
// pred magnitude p = (exp(mix)-1)*mask for the binaural channels (use_mix=1) or the tensor itself (mono).
__global__ __launch_bounds__(1024) void stft_l2_kernel(const float* __restrict__ mix, const float* __restrict__ pred, int Cp,
                                                       const float* __restrict__ gt_comps, int Cg, int nch, int use_mix,
                                                       float* __restrict__ out, int L /* F*T */) {
  __shared__ float sh[16];
  const int e = blockIdx.x;
  float tot = 0.f;
  for (int ch = 0; ch < nch; ++ch) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = threadIdx.x; i0 < L; i0 += 4096) {
      float gm[4], pm[4], mx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 1024 * u;
        const bool ok = i < L;
        const size_t pix = (size_t)e * L + (ok ? i : 0);
        gm[u] = ok ? gt_comps[pix * Cg + 2 * ch] : 0.f;
        pm[u] = ok ? pred[pix * Cp + ch] : 0.f;
        mx[u] = (ok && use_mix) ? mix[pix * Cp + ch] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float p = use_mix ? (expf(mx[u]) - 1.f) * pm[u] : pm[u];
        const float d = gm[u] - p;
        s[u] += d * d;
      }
    }
    tot += block_sum16((s[0] + s[1]) + (s[2] + s[3]), sh) / (float)(2 * L);
  }
  if (threadIdx.x == 0) out[e] = tot;
}



// Per-episode statistics of the rollout step (ppo_trainer.py:407-478: the reference keeps them as python floats / numpy and
// pays a host sync per value): one thread per env does what the reference does after every env step -- accumulate the
// running sums of the current episode, fold them into the finished-episode totals where the env is done (nd = 1 - not_done),
// reset the running sums there.  Arithmetic order and rounding are those of the elementwise formulation
// (cur += x; total += nd * (cur / steps); cur *= not_done), without fused multiply-adds.
__global__ void episode_stats_kernel(m2h_episode_stats st, const float* __restrict__ rewards, const float* __restrict__ dist_probs,
                                     const float* __restrict__ bin_losses, const float* __restrict__ mono_losses,
                                     const float* __restrict__ mem_losses, const float* __restrict__ not_done,
                                     const float* __restrict__ ndgs, const float* __restrict__ dgs, int N, int A) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  const float m = not_done[e];
  const float nd = 1.f - m;
  if (ndgs) st.episode_ndgs[e] = __fadd_rn(st.episode_ndgs[e], __fmul_rn(nd, ndgs[e]));
  if (dgs) st.episode_dgs[e] = __fadd_rn(st.episode_dgs[e], __fmul_rn(nd, dgs[e]));
  const float cr = __fadd_rn(st.current_episode_reward[e], rewards[e]);
  const float cs = __fadd_rn(st.current_episode_step[e], 1.f);
  const float cb = __fadd_rn(st.current_episode_bin_losses[e], bin_losses[e]);
  const float cm = __fadd_rn(st.current_episode_mono_losses[e], mono_losses[e]);
  const float cf = __fadd_rn(st.current_episode_monoFromMem_losses[e], mem_losses[e]);
  st.episode_rewards[e] = __fadd_rn(st.episode_rewards[e], __fmul_rn(nd, cr));
  st.episode_steps[e] = __fadd_rn(st.episode_steps[e], __fmul_rn(nd, cs));
  st.episode_counts[e] = __fadd_rn(st.episode_counts[e], nd);
  for (int a = 0; a < A; ++a) {
    const float cp = __fadd_rn(st.current_episode_dist_probs[e * A + a], dist_probs[e * A + a]);
    st.episode_dist_probs[e * A + a] = __fadd_rn(st.episode_dist_probs[e * A + a], __fmul_rn(nd, __fdiv_rn(cp, cs)));
    st.current_episode_dist_probs[e * A + a] = __fmul_rn(cp, m);
  }
  st.episode_bin_losses_allSteps[e] = __fadd_rn(st.episode_bin_losses_allSteps[e], __fmul_rn(nd, __fdiv_rn(cb, cs)));
  st.episode_mono_losses_lastStep[e] = __fadd_rn(st.episode_mono_losses_lastStep[e], __fmul_rn(nd, mono_losses[e]));
  st.episode_mono_losses_allSteps[e] = __fadd_rn(st.episode_mono_losses_allSteps[e], __fmul_rn(nd, __fdiv_rn(cm, cs)));
  st.episode_monoFromMem_losses_lastStep[e] = __fadd_rn(st.episode_monoFromMem_losses_lastStep[e], __fmul_rn(nd, mem_losses[e]));
  st.episode_monoFromMem_losses_allSteps[e] = __fadd_rn(st.episode_monoFromMem_losses_allSteps[e], __fmul_rn(nd, __fdiv_rn(cf, cs)));
  st.current_episode_reward[e] = __fmul_rn(cr, m);
  st.current_episode_step[e] = __fmul_rn(cs, m);
  st.current_episode_bin_losses[e] = __fmul_rn(cb, m);
  st.current_episode_mono_losses[e] = __fmul_rn(cm, m);
  st.current_episode_monoFromMem_losses[e] = __fmul_rn(cf, m);
}

// Batched row copies with device-resident row indices (RolloutStoragePol/Sep.insert and the per-step row reads of the rollout,
// common/rollout_storage.py:68-96, 372-390, when the step is replayed from a HIP graph): item i copies `bytes` bytes from
// src + idx[src_slot] * bytes to dst + idx[dst_slot] * bytes (slot < 0: no offset).  grid.y = item, grid.x strides over it.
struct RowsCopyArgs {
  m2h_row_copy item[M2H_ROWS_COPY_MAX];
};
__global__ __launch_bounds__(256) void rows_copy_kernel(RowsCopyArgs a, const long long* __restrict__ idx) {
  const m2h_row_copy it = a.item[blockIdx.y];
  const char* src = static_cast<const char*>(it.src) + (it.src_slot >= 0 ? (size_t)idx[it.src_slot] * it.bytes : 0);
  char* dst = static_cast<char*>(it.dst) + (it.dst_slot >= 0 ? (size_t)idx[it.dst_slot] * it.bytes : 0);
  const size_t start = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  if (((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst) | it.bytes) & 15) == 0) {  // block-uniform
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (size_t i = start; i < it.bytes / 16; i += stride) d4[i] = s4[i];
  } else {
    const uint32_t* s1 = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d1 = reinterpret_cast<uint32_t*>(dst);
    for (size_t i = start; i < it.bytes / 4; i += stride) d1[i] = s1[i];
  }
}

// Device-resident step counters of the two rollout storages, idx = (pol_step, pol_step + 1, sep_step + 1), advanced at the end
// of a replayed rollout step (RolloutStoragePol/Sep.insert: step = (step + 1) % num_steps, common/rollout_storage.py:96,390).
__global__ void step_index_advance_kernel(long long* __restrict__ idx, int T_pol, int T_sep, unsigned long long* __restrict__ rng = nullptr,
                                          unsigned long long rng_inc = 0) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (rng != nullptr) rng[1] += rng_inc;   // the fused sampler's counter: past the step's draws
    const long long p = (idx[0] + 1) % T_pol;
    const long long s = idx[2] % T_sep;   // idx[2] holds sep_step + 1
    idx[0] = p;
    idx[1] = p + 1;
    idx[2] = s + 1;
  }
}

// Synthetic vector env (m2h/envs/synthetic_env.py; stands in for the simulator's pose update, habitat_audio/simulator_train.py):
// action 0 = MOVE_FORWARD (node + 1), 1 = TURN_LEFT (angle + 1), 2 = TURN_RIGHT (angle + 3), all modulo; one thread per env.
__global__ void synth_env_step_kernel(const long long* __restrict__ actions, long long* __restrict__ node, long long* __restrict__ angle,
                                      int n_nodes, int N) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  const long long a = actions[e];
  node[e] = (node[e] + (a == 0 ? 1 : 0)) % n_nodes;
  angle[e] = (angle[e] + (a == 1 ? 1 : 0) + (a == 2 ? 3 : 0)) % 4;
}

// Batched row gathers: item i copies, for every env e < N, row index[sel_i][e] * mul_i + add[e] * ... of src_i to row e of dst_i.
// Serves the synthetic env's observation lookup (cached frames indexed by node * 4 + angle, audio pool indexed per env): six
// index_select launches as one.  index: [2][N] int64 = (frame index parts, audio index); frame row = node * 4 + angle.
struct RowsGatherArgs {
  m2h_row_copy item[M2H_ROWS_COPY_MAX];   // src_slot: 0 = frame index (node * 4 + angle), 1 = audio index; dst_slot unused
};
__global__ __launch_bounds__(256) void rows_gather_kernel(RowsGatherArgs a, const long long* __restrict__ node,
                                                          const long long* __restrict__ angle, const long long* __restrict__ audio_idx) {
  const m2h_row_copy it = a.item[blockIdx.z];
  const int e = blockIdx.y;
  const long long row = it.src_slot == 0 ? node[e] * 4 + angle[e] : audio_idx[e];
  const char* src = static_cast<const char*>(it.src) + (size_t)row * it.bytes;
  char* dst = static_cast<char*>(it.dst) + (size_t)e * it.bytes;
  const size_t start = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  if (((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst) | it.bytes) & 15) == 0) {
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (size_t i = start; i < it.bytes / 16; i += stride) d4[i] = s4[i];
  } else {
    const uint32_t* s1 = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d1 = reinterpret_cast<uint32_t*>(dst);
    for (size_t i = start; i < it.bytes / 4; i += stride) d1[i] = s1[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// training-side kernels: GRU / heads backward, losses, grad-norm clipping, Adam
// ---------------------------------------------------------------------------------------------------------------

// Backward of gru_gates_kernel for one time step.  Inputs as the forward plus dh = dL/dh_out.  Outputs:
//   dgi  [M,3H] = dL/d(x W_ih^T + b_ih)           dpre [M,3H] = dL/d(m * gh_raw + b_hh)  (so dL/dgh_raw = m * dpre)
//   dhp  [M,H]  = dh * z  (direct path to the masked previous hidden state; caller applies the mask)
//   hpm  [M,H]  = m * hprev  (masked previous hidden state, the A operand of the W_hh weight gradient)
__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh_raw,
                                                            const float* __restrict__ bhh, const float* __restrict__ hprev,
                                                            const float* __restrict__ mask, const float* __restrict__ dh,
                                                            float* __restrict__ dgi, float* __restrict__ dpre, float* __restrict__ dhp,
                                                            float* __restrict__ hpm, int M, int H) {
  const int total = M * H;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / H, j = i - row * H;
    const float m = mask != nullptr ? mask[row] : 1.f;
    const size_t o = (size_t)row * 3 * H;
    const float r = sigmoidf_(gi[o + j] + (m * gh_raw[o + j] + bhh[j]));
    const float z = sigmoidf_(gi[o + H + j] + (m * gh_raw[o + H + j] + bhh[H + j]));
    const float hn = m * gh_raw[o + 2 * H + j] + bhh[2 * H + j];
    const float n = tanhf(gi[o + 2 * H + j] + r * hn);
    const float hp = m * hprev[i];
    const float g = dh[i];
    const float dn = g * (1.f - z);
    const float dz = g * (hp - n);
    const float dan = dn * (1.f - n * n);
    const float dr = dan * hn;
    const float daz = dz * z * (1.f - z);
    const float dar = dr * r * (1.f - r);
    dgi[o + j] = dar;
    dgi[o + H + j] = daz;
    dgi[o + 2 * H + j] = dan;
    dpre[o + j] = dar;
    dpre[o + H + j] = daz;
    dpre[o + 2 * H + j] = dan * r;
    dhp[i] = g * z;
    hpm[i] = hp;
  }
}

// out = a + mask_row * (b + c): total gradient of h_{t-1} = output-path gradient + masked recurrent-path gradient
__global__ void gru_bwd_combine_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                       const float* __restrict__ mask, float* __restrict__ out, int M, int H) {
  const int total = M * H;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const float m = mask != nullptr ? mask[i / H] : 1.f;
    out[i] = (a != nullptr ? a[i] : 0.f) + m * (b[i] + c[i]);
  }
}

// Recurrent part of one GRU backward step for M <= 16 rows in one launch: out = a + mask * (dpre W_hh + dhp), i.e. the
// [M,3H] x [3H,H] product (a 3 MB weight stream against 14 rows) and m2h_gru_bwd_combine.  A block owns GB_U hidden units = GB_U rows
// of W_hh^T and runs their 3H-long dot products with the 16 rows of dpre through skinny_dot16 (the 16-wide tile's other columns
// repeat the last row); the four waves' partial tiles meet through LDS in wave order.
constexpr int GB_U = 4, GB_E = 16;
// With the *_p arguments (gi_p != nullptr) the gate backward of the PREVIOUS time step (gru_gates_bwd_kernel over this block's
// columns: it is elementwise in (row, hidden unit), and its dh is exactly the out element the thread has just produced) runs as this
// kernel's epilogue: one launch per BPTT step instead of two.  dhp is read (this step's) and rewritten (the previous step's) by the
// same thread at the same element.
constexpr int GB_NW = 16;    // waves per block: the 96 steps of the 3H-long products are 6 per wave (four waves: 24 -- 9.7 us per BPTT step)
__global__ __launch_bounds__(64 * GB_NW) void gru_bwd_rec_kernel(const float* __restrict__ dpre, const float* __restrict__ whh_t,
                                                          const float* __restrict__ a, float* __restrict__ dhp,
                                                          const float* __restrict__ mask, float* __restrict__ out, int M, int H,
                                                          const float* __restrict__ gi_p = nullptr, const float* __restrict__ gh_p = nullptr,
                                                          const float* __restrict__ bhh = nullptr, const float* __restrict__ hprev_p = nullptr,
                                                          const float* __restrict__ mask_p = nullptr, float* __restrict__ dgi_p = nullptr,
                                                          float* __restrict__ dpre_p = nullptr, float* __restrict__ hpm_p = nullptr) {
  __shared__ float R[GB_NW][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int j0 = blockIdx.x * GB_U, K = 3 * H;
  const float* wrow = whh_t + (size_t)(j0 + min(i, GB_U - 1)) * K + 4 * kq;
  const float* xrow = dpre + (size_t)min(i, M - 1) * K + 4 * kq;
  // the epilogue's operands of this thread's (unit, row) pair, fetched before the dot products (as gru_step_kernel does): behind the
  // barrier they were a second dependent round trip in every one of the 20 BPTT steps
  const bool epi = tid < GB_U * GB_E && (tid & 15) < M;
  const int eu = (tid >> 4) & (GB_U - 1), em = min(tid & 15, M - 1), ej = j0 + eu;
  const size_t eo = (size_t)em * H + ej, eo3 = (size_t)em * 3 * H;
  const float e_mk = mask != nullptr ? mask[em] : 1.f;
  const float e_a = (epi && a != nullptr) ? a[eo] : 0.f;
  const float e_dhp = epi ? dhp[eo] : 0.f;
  float e_mp = 1.f, e_gir = 0.f, e_giz = 0.f, e_gin = 0.f, e_ghr = 0.f, e_ghz = 0.f, e_ghn = 0.f, e_br = 0.f, e_bz = 0.f, e_bn = 0.f, e_hp = 0.f;
  if (gi_p != nullptr && epi) {
    e_mp = mask_p != nullptr ? mask_p[em] : 1.f;
    e_gir = gi_p[eo3 + ej]; e_giz = gi_p[eo3 + H + ej]; e_gin = gi_p[eo3 + 2 * H + ej];
    e_ghr = gh_p[eo3 + ej]; e_ghz = gh_p[eo3 + H + ej]; e_ghn = gh_p[eo3 + 2 * H + ej];
    e_br = bhh[ej]; e_bz = bhh[H + ej]; e_bn = bhh[2 * H + ej];
    e_hp = hprev_p[eo];
  }
  const f32x4_r acc = skinny_dot16<GB_NW>(xrow, wrow, K >> 4, wave);
#pragma unroll
  for (int e = 0; e < 4; ++e) R[wave][kq * 4 + e][i] = acc[e];
  __syncthreads();
  if (epi) {
    const int u = tid >> 4, m = tid & 15;
    const float rec = wave_tile_sum<GB_NW>(R, m, u);
    const size_t o = (size_t)m * H + j0 + u;
    const float mk = e_mk;
    const float g = e_a + mk * (rec + e_dhp);
    out[o] = g;
    if (gi_p != nullptr) {                 // gru_gates_bwd_kernel of the previous step at (row m, unit j), dh = g
      const int j = j0 + u;
      const float mp = e_mp;
      const size_t o3 = (size_t)m * 3 * H;
      const float r = sigmoidf_(e_gir + (mp * e_ghr + e_br));
      const float z = sigmoidf_(e_giz + (mp * e_ghz + e_bz));
      const float hn = mp * e_ghn + e_bn;
      const float n = tanhf(e_gin + r * hn);
      const float hp = mp * e_hp;
      const float dn = g * (1.f - z);
      const float dz = g * (hp - n);
      const float dan = dn * (1.f - n * n);
      const float dr = dan * hn;
      const float daz = dz * z * (1.f - z);
      const float dar = dr * r * (1.f - r);
      dgi_p[o3 + j] = dar;
      dgi_p[o3 + H + j] = daz;
      dgi_p[o3 + 2 * H + j] = dan;
      dpre_p[o3 + j] = dar;
      dpre_p[o3 + H + j] = daz;
      dpre_p[o3 + 2 * H + j] = dan * r;
      dhp[o] = g * z;
      hpm_p[o] = hp;
    }
  }
}

// Parameter gradients of the two heads from dz [M][ZS] (policy_heads_bwd_kernel) and the features: dw[z][h] = sum_m dz[m][z] feats[m][h],
// db[z] = sum_m dz[m][z].  ZS x H = 8 x 512 outputs over a few hundred rows: the tiled weight-gradient engine spent 19 us + a 5 us reduce +
// a 4.5 us bias launch on it at the head of every policy epoch's backward.  A block owns 64 columns; its four waves take every fourth row
// (each lane: one column, ZS running sums; dz rows come through LDS), and meet in LDS in wave order (bit-reproducible).
template <int ZS>
__global__ __launch_bounds__(256) void policy_heads_wgrad_kernel(const float* __restrict__ feats, const float* __restrict__ dz,
                                                                 float* __restrict__ dw, float* __restrict__ db, int M, int H) {
  constexpr int RB = 64;                         // rows of dz staged per round
  __shared__ float zs[RB][ZS];
  __shared__ float part[4][ZS][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x * 64 + lane;
  float acc[ZS], bsum[ZS];
#pragma unroll
  for (int z = 0; z < ZS; ++z) acc[z] = bsum[z] = 0.f;
  for (int m0 = 0; m0 < M; m0 += RB) {
    __syncthreads();
    for (int i = tid; i < RB * ZS; i += 256) {
      const int r = i / ZS;
      zs[r][i - r * ZS] = m0 + r < M ? dz[(size_t)(m0 + r) * ZS + (i - r * ZS)] : 0.f;
    }
    __syncthreads();
    const int rn = min(RB, M - m0);
    float f[RB / 4];
#pragma unroll
    for (int j = 0; j < RB / 4; ++j) f[j] = feats[(size_t)min(m0 + wave + 4 * j, M - 1) * H + h];   // (all in flight; rows past the end are read clamped and not used)
#pragma unroll
    for (int j = 0; j < RB / 4; ++j) {
      const int r = wave + 4 * j;
      if (r < rn) {
#pragma unroll
        for (int z = 0; z < ZS; ++z) {
          acc[z] += zs[r][z] * f[j];
          bsum[z] += zs[r][z];
        }
      }
    }
  }
#pragma unroll
  for (int z = 0; z < ZS; ++z) part[wave][z][lane] = acc[z];
  __syncthreads();
  for (int i = tid; i < ZS * 64; i += 256) {
    const int z = i >> 6, l = i & 63;
    dw[(size_t)z * H + blockIdx.x * 64 + l] = (part[0][z][l] + part[1][z][l]) + (part[2][z][l] + part[3][z][l]);
  }
  if (blockIdx.x == 0) {       // the bias gradient: every lane of a wave holds the same row sums
    __syncthreads();
#pragma unroll
    for (int z = 0; z < ZS; ++z) part[wave][z][lane] = bsum[z];
    __syncthreads();
    if (tid < ZS) db[tid] = (part[0][tid][0] + part[1][tid][0]) + (part[2][tid][0] + part[3][tid][0]);
  }
}

// Backward of policy_heads_kernel: one wave per row.  g_value, g_logp, g_ent_rows = dL/d(value | logp_act | entropy) per row.
//   dz [M][ZS]: columns 0..A-1 = dL/dlogits, column A = dL/dvalue, rest 0   (ZS = A+1 rounded up to 4)
//   dfeats [M][H] = sum_a dlogit_a * Wa[a] + dvalue * Wc
__global__ __launch_bounds__(256) void policy_heads_bwd_kernel(const float* __restrict__ logp_all, const float* __restrict__ probs,
                                                               const long long* __restrict__ actions, const float* __restrict__ g_value,
                                                               const float* __restrict__ g_logp, const float* __restrict__ g_ent_rows, const float* __restrict__ Wa,
                                                               const float* __restrict__ Wc, float* __restrict__ dz,
                                                               float* __restrict__ dfeats, int M, int H, int A, int ZS) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float ent = 0.f;
  for (int a = 0; a < A; ++a) ent -= probs[row * A + a] * logp_all[row * A + a];
  const int act = actions != nullptr ? (int)actions[row] : -1;
  const float gl = g_logp != nullptr ? g_logp[row] : 0.f;
  const float gv = g_value != nullptr ? g_value[row] : 0.f;
  const float g_ent = g_ent_rows != nullptr ? g_ent_rows[row] : 0.f;
  float dl[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    dl[a] = 0.f;
    if (a < A) {
      const float p = probs[row * A + a], lp = logp_all[row * A + a];
      dl[a] = gl * ((a == act ? 1.f : 0.f) - p) + g_ent * (-p * (lp + ent));
    }
  }
  if (lane == 0) {
    for (int a = 0; a < ZS; ++a) dz[row * ZS + a] = a < A ? dl[a] : (a == A ? gv : 0.f);
  }
  for (int k = lane; k < H; k += 64) {
    float s = gv * Wc[k];
#pragma unroll
    for (int a = 0; a < 8; ++a)
      if (a < A) s += dl[a] * Wa[a * H + k];
    dfeats[(size_t)row * H + k] = s;
  }
}

// L1 loss against a strided ground truth (F.l1_loss(pred, gt_comps[..., off], reduction=mean); ppo.py:213,216,221):
// partial sums per block into part[blockIdx], grad[i] = sign(p - g) / n  (sign(0) = 0 like torch).
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int gt_stride, int gt_off,
                                                      float* __restrict__ part, float* __restrict__ grad, size_t n) {
  __shared__ float sh[4];
  float s = 0.f;
  const float inv = 1.f / (float)n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = pred[i] - gt[i * gt_stride + gt_off];
    s += fabsf(d);
    if (grad != nullptr) grad[i] = d > 0.f ? inv : (d < 0.f ? -inv : 0.f);
  }
  const float t = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// The same loss for a prediction that is still in the convolution's NHWC layout [B][32 rows][T][16 bands] (AcousticMem's last conv before its
// de-slice, memory_nets.py:62-67): pred_BHWC[b][band * 32 + row][t] = y[b][row][t][band].  One block per (b, row): the 16 band rows of the
// ground truth (T x gt_stride floats each, contiguous) and the T x 16 tile of y are read coalesced, |y - g| is summed and the gradient
// sign(y - g) / n leaves in the SAME NHWC layout -- the layout the conv's weight / input gradient kernels read.  Replaces the conv's
// de-sliced store + m2h_l1_loss + the gradient's re-slice (m2h_slice_concat_input): one pass over y instead of three tensors' round trips.
__global__ __launch_bounds__(256) void l1_nhwc16_kernel(const float* __restrict__ y, const float* __restrict__ gt, int gt_stride, int gt_off, int T,
                                                        float* __restrict__ part, float* __restrict__ dy, float inv, int nrows) {
  extern __shared__ float gts[];                       // [16 bands][T] (+1 pad per band row)
  __shared__ float sh[4];
  const int TP = T + 1;
  float s = 0.f;
  for (int br = blockIdx.x; br < nrows; br += gridDim.x) {   // (b, row) pairs of this block: at most 2048 partial sums whatever B
    const int b = br >> 5, row = br & 31;
    __syncthreads();                                   // (the previous pair's tile is consumed)
    for (int i = threadIdx.x; i < 16 * T; i += 256) {  // band-major, coalesced over t
      const int band = i / T, t = i - band * T;
      gts[band * TP + t] = gt[(((size_t)b * 512 + band * 32 + row) * T + t) * gt_stride + gt_off];
    }
    __syncthreads();
    const size_t base = ((size_t)b * 32 + row) * T * 16;
    for (int i = threadIdx.x; i < 16 * T; i += 256) {  // NHWC order: band fastest
      const int t = i >> 4, band = i & 15;
      const float d = y[base + i] - gts[band * TP + t];
      s += fabsf(d);
      if (dy != nullptr) dy[base + i] = d > 0.f ? inv : (d < 0.f ? -inv : 0.f);
    }
  }
  const float tot = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// partial sums of a long list (one block per 1024 entries is overkill here: a few tens of thousands of floats) -> out[0] = scale * sum, fixed order
__global__ __launch_bounds__(1024) void sum_partials_wide_kernel(const float* __restrict__ part, int n, float scale, float* __restrict__ out) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += sh[w];
    out[0] = t * scale;
  }
}

// bin loss for logging (ppo.py:219-221; passive_trainer.py:270-272): mean | (exp(mix)-1)*mask - gt_bin_comps[..., 2c] | over [.., c<2]
__global__ __launch_bounds__(256) void bin_l1_kernel(const float* __restrict__ mix, const float* __restrict__ masks, const float* __restrict__ gt,
                                                     int Cg, int cstep, float* __restrict__ part, float* __restrict__ grad_masks, size_t npix) {
  __shared__ float sh[4];
  float s = 0.f;
  const float inv = 1.f / (float)(2 * npix);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * npix; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i >> 1;
    const int c = (int)(i & 1);
    const float e = expf(mix[i]) - 1.f;
    const float d = e * masks[i] - gt[pix * Cg + cstep * c];
    s += fabsf(d);
    if (grad_masks != nullptr) grad_masks[i] = (d > 0.f ? inv : (d < 0.f ? -inv : 0.f)) * e;
  }
  const float t = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// out[0] = scale * sum(part[0..n))   (one wave, fixed lane-strided order + butterfly: deterministic)
__global__ void sum_partials_kernel(const float* __restrict__ part, int n, float scale, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += part[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = s * scale;
}

// sum of squares partials (grad-norm): part[blockIdx] = sum x^2 over the block's grid-stride share
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, size_t n, float* __restrict__ part) {
  __shared__ float sh[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += x[i] * x[i];
  const float t = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// clip_grad_norm_ coefficient (torch: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1): coef[0] = coefficient,
// coef[1] = total_norm.  part holds nparts sums of squares.  max_norm <= 0 -> coef 1.
__global__ void clip_coef_kernel(const float* __restrict__ part, int nparts, float max_norm, float* __restrict__ coef) {
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    const float norm = sqrtf(s);
    float c = 1.f;
    if (max_norm > 0.f) c = fminf(max_norm / (norm + 1e-6f), 1.f);
    coef[0] = c;
    coef[1] = norm;
  }
}

// torch.optim.Adam step (no amsgrad, no weight decay) over a flat buffer; the gradient is scaled by coef[0] (clip) * gscale
// (1/world_size after a sum all-reduce) and written back scaled (as clip_grad_norm_ does in place).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   size_t n, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   const float* __restrict__ coef, float gscale, const float* __restrict__ hyper) {
  const float c = (coef != nullptr ? coef[0] : 1.f) * gscale;
  if (hyper != nullptr) {   // m2h_adam_step_dev: the step-dependent scalars come from device memory (a replayed HIP graph holds no host scalars)
    lr = hyper[0];
    bc1 = hyper[1];
    bc2_sqrt = hyper[2];
  }
  const float step = lr / bc1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * c;
    g[i] = gi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= step * (mi / denom);
  }
}

__global__ void set3_kernel(float* __restrict__ dst, float a, float b, float c) {
  dst[0] = a;
  dst[1] = b;
  dst[2] = c;
}

// Minibatch gather of the recurrent generators (common/rollout_storage.py:182-298,392-457):
// dst[t][j][:] = src[t][perm[j]][:]  for t < T, j < Nsel; rows of L elements (bytes, copied as 16-byte or 4-byte words).
__global__ __launch_bounds__(256) void gather_envs_kernel(const uint32_t* __restrict__ src, const long long* __restrict__ perm,
                                                          uint32_t* __restrict__ dst, int T, int N, int Nsel, size_t Lw) {
  const size_t total = (size_t)T * Nsel * Lw;
  if ((Lw & 3) == 0) {
    const size_t L4 = Lw >> 2;
    const size_t tot4 = (size_t)T * Nsel * L4;
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot4; i += (size_t)gridDim.x * blockDim.x) {
      const size_t l = i % L4;
      const size_t r = i / L4;
      const int j = (int)(r % Nsel);
      const int t = (int)(r / Nsel);
      d4[i] = s4[((size_t)t * N + (size_t)perm[j]) * L4 + l];
    }
    return;
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t l = i % Lw;
    const size_t r = i / Lw;
    const int j = (int)(r % Nsel);
    const int t = (int)(r / Nsel);
    dst[i] = src[((size_t)t * N + (size_t)perm[j]) * Lw + l];
  }
}

static inline unsigned grid_for(size_t total, unsigned cap = 4096) {
  size_t g = (total + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace m2h

using namespace m2h;

extern "C" {

int m2h_slice_concat_input(const float* a, int Ca, const float* b, int Cb, const float* mul, const float* bscale, int op,
                           float* out, int B, int F, int T, m2h_stream stream) {
  M2H_REQUIRE(a != nullptr && out != nullptr && Ca > 0 && Cb >= 0, "slice_concat_input: bad arguments");
  M2H_REQUIRE((Cb == 0) == (b == nullptr), "slice_concat_input: b/Cb mismatch");
  M2H_REQUIRE(op >= 0 && op <= 2 && (op != 1 || mul != nullptr), "slice_concat_input: bad op");
  M2H_REQUIRE(B > 0 && F > 0 && T > 0 && F % 16 == 0, "slice_concat_input: bad sizes (F %% 16)");
  const size_t total = (size_t)B * (F / 16) * T * (16 * (Ca + Cb) / 4);
  M2H_LAUNCH(slice_concat_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), a, Ca, b, Cb, mul, bscale, op, out, B, F, T);
  return launch_status("slice_concat_input");
}

int m2h_visual_input(const float* rgb, const float* depth, float* out, int B, int H, int W, m2h_stream stream) {
  M2H_REQUIRE(rgb != nullptr && out != nullptr && B > 0 && H > 0 && W > 0, "visual_input: bad arguments");
  const size_t npix = (size_t)B * H * W;
  M2H_LAUNCH(visual_input_kernel, dim3(grid_for(npix)), dim3(256), 0, as_stream(stream), rgb, depth, out, npix);
  return launch_status("visual_input");
}

int m2h_gru_gates(const float* gi, const float* gh_raw, const float* bhh, const float* hprev, const float* mask, float* hout,
                  int M, int H, m2h_stream stream) {
  M2H_REQUIRE(gi && gh_raw && bhh && hprev && hout && M > 0 && H > 0, "gru_gates: bad arguments");
  M2H_LAUNCH(gru_gates_kernel, dim3(grid_for((size_t)M * H)), dim3(256), 0, as_stream(stream), gi, gh_raw, bhh, hprev, mask, hout, M, H);
  return launch_status("gru_gates");
}

int m2h_gru_step(const float* gi, const float* whh, const float* bhh, const float* hprev, const float* mask, float* gh_raw, float* hout,
                 int M, int H, m2h_stream stream) {
  M2H_REQUIRE(gi && whh && bhh && hprev && gh_raw && hout, "gru_step: null pointer");
  M2H_REQUIRE(M > 0 && M <= GRU_E && H > 0 && H % 16 == 0, "gru_step: needs 1 <= M <= %d rows and H %% 16 == 0 (got M=%d, H=%d)",
              GRU_E, M, H);
  M2H_LAUNCH(gru_step_kernel, dim3(H / GRU_U), dim3(64 * GS_NW), 0, as_stream(stream), gi, whh, bhh, hprev, mask, gh_raw, hout, M, H);
  return launch_status("gru_step");
}

int m2h_gru_cell(const float* x, const float* wih, const float* bih, const float* whh, const float* bhh, const float* hprev,
                 const float* mask, float* hout, int M, int I, int H, m2h_stream stream) {
  M2H_REQUIRE(x && wih && bih && whh && bhh && hprev && hout, "gru_cell: null pointer");
  M2H_REQUIRE(M > 0 && M <= GRU_E && H > 0 && H % 16 == 0 && I > 0 && I % 16 == 0,
              "gru_cell: needs 1 <= M <= %d rows, H %% 16 == 0 and I %% 16 == 0 (got M=%d, I=%d, H=%d)", GRU_E, M, I, H);
  M2H_LAUNCH(gru_cell_kernel, dim3(H / GRU_U), dim3(64 * GC_NW), 0, as_stream(stream), x, wih, bih, whh, bhh, hprev, mask, hout, M, I, H);
  return launch_status("gru_cell");
}

int m2h_lstm_cell(const float* gi, const float* gh, const float* c_prev, const float* mask, float* h_out, float* c_out, float* gates_out, int M, int H,
                  m2h_stream stream) {
  M2H_REQUIRE(gi && gh && c_prev && mask && h_out && c_out && M > 0 && H > 0, "lstm_cell: bad arguments");
  size_t g = ((size_t)M * H + 255) / 256;
  if (g > 4096) g = 4096;
  M2H_LAUNCH(lstm_cell_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), gi, gh, c_prev, mask, h_out, c_out, gates_out, M, H);
  return launch_status("lstm_cell");
}

int m2h_lstm_cell_bwd(const float* dh, const float* dc, const float* gates, const float* c_prev, const float* mask, const float* c, float* dpre,
                      float* dc_prev, int M, int H, m2h_stream stream) {
  M2H_REQUIRE(gates && c_prev && mask && c && dpre && dc_prev && M > 0 && H > 0, "lstm_cell_bwd: bad arguments");
  size_t g = ((size_t)M * H + 255) / 256;
  if (g > 4096) g = 4096;
  M2H_LAUNCH(lstm_cell_bwd_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), dh, dc, gates, c_prev, mask, c, dpre, dc_prev, M, H);
  return launch_status("lstm_cell_bwd");
}

int m2h_policy_heads(const float* feats, const float* Wa, const float* ba, const float* Wc, const float* bc,
                     const long long* actions, float* value, float* logp_all, float* probs, float* entropy, float* logp_act,
                     int M, int H, int A, m2h_stream stream) {
  M2H_REQUIRE(feats && Wa && ba && Wc && bc && value && logp_all && probs && entropy, "policy_heads: null pointer");
  M2H_REQUIRE(M > 0 && H > 0 && H % 64 == 0 && A > 0 && A <= 8, "policy_heads: bad sizes (H %% 64, A <= 8)");
  M2H_LAUNCH(policy_heads_kernel, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), feats, Wa, ba, Wc, bc, actions, value,
                     logp_all, probs, entropy, logp_act, M, H, A);
  return launch_status("policy_heads");
}

int m2h_policy_heads_act(const float* feats, const float* Wa, const float* ba, const float* Wc, const float* bc, const float* noise,
                         float* value, float* logp_all, float* probs, float* entropy, long long* actions, float* logp_act, int M, int H,
                         int A, m2h_stream stream) {
  M2H_REQUIRE(feats && Wa && ba && Wc && bc && value && logp_all && probs && entropy && actions && logp_act, "policy_heads_act: null pointer");
  M2H_REQUIRE(M > 0 && H > 0 && H % 64 == 0 && A > 0 && A <= 8, "policy_heads_act: bad sizes (H %% 64, A <= 8)");
  M2H_LAUNCH(policy_heads_kernel, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), feats, Wa, ba, Wc, bc,
                     static_cast<const long long*>(nullptr), value, logp_all, probs, entropy, logp_act, M, H, A, noise, actions);
  return launch_status("policy_heads_act");
}

int m2h_policy_heads_act_rng(const float* feats, const float* Wa, const float* ba, const float* Wc, const float* bc, const unsigned long long* rng_state,
                             float* value, float* logp_all, float* probs, float* entropy, long long* actions, float* logp_act, float* noise_out,
                             int M, int H, int A, m2h_stream stream) {
  M2H_REQUIRE(feats && Wa && ba && Wc && bc && rng_state && value && logp_all && probs && entropy && actions && logp_act, "policy_heads_act_rng: null pointer");
  M2H_REQUIRE(M > 0 && H > 0 && H % 64 == 0 && A > 0 && A <= 8, "policy_heads_act_rng: bad sizes (H %% 64, A <= 8)");
  M2H_LAUNCH(policy_heads_kernel, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), feats, Wa, ba, Wc, bc,
             static_cast<const long long*>(nullptr), value, logp_all, probs, entropy, logp_act, M, H, A, static_cast<const float*>(nullptr), actions, rng_state,
             noise_out);
  return launch_status("policy_heads_act (fused draw)");
}

int m2h_sample_actions(const float* probs, const float* noise, long long* actions, int M, int A, m2h_stream stream) {
  M2H_REQUIRE(probs && noise && actions && M > 0 && A > 0 && A <= 64, "sample_actions: bad arguments");
  M2H_LAUNCH(sample_actions_kernel, dim3((M + 255) / 256), dim3(256), 0, as_stream(stream), probs, noise, actions, M, A);
  return launch_status("sample_actions");
}

int m2h_gather_logp(const float* logp_all, const long long* actions, float* out, int M, int A, m2h_stream stream) {
  M2H_REQUIRE(logp_all && actions && out && M > 0 && A > 0, "gather_logp: bad arguments");
  M2H_LAUNCH(gather_logp_kernel, dim3((M + 255) / 256), dim3(256), 0, as_stream(stream), logp_all, actions, out, M, A);
  return launch_status("gather_logp");
}

int m2h_gae_returns(const float* rewards, float* value_preds, const float* masks, const float* next_value, float* returns, int T,
                    int N, int use_gae, float gamma, float tau, m2h_stream stream) {
  M2H_REQUIRE(rewards && value_preds && masks && next_value && returns && T > 0 && N > 0, "gae_returns: bad arguments");
  M2H_LAUNCH(gae_returns_kernel, dim3((N + 63) / 64), dim3(64), 0, as_stream(stream), rewards, value_preds, masks, next_value,
                     returns, T, N, use_gae, gamma, tau);
  return launch_status("gae_returns");
}

int m2h_advantages(const float* returns, const float* value_preds, float* adv, float* stats, int n, int mode, float eps,
                   m2h_stream stream) {
  M2H_REQUIRE(returns && value_preds && adv && n > 1 && mode >= 0 && mode <= 2, "advantages: bad arguments");
  M2H_LAUNCH(advantages_kernel, dim3(1), dim3(256), 0, as_stream(stream), returns, value_preds, adv, stats, n, mode, eps);
  return launch_status("advantages");
}

int m2h_adv_sqdiff(const float* adv, const float* gmean, float* out, int n, m2h_stream stream) {
  M2H_REQUIRE(adv && gmean && out && n > 0, "adv_sqdiff: bad arguments");
  M2H_LAUNCH(adv_sqdiff_kernel, dim3(1), dim3(256), 0, as_stream(stream), adv, gmean, out, n);
  return launch_status("adv_sqdiff");
}

int m2h_adv_apply(float* adv, const float* gmean, const float* gvar, int n, float eps, m2h_stream stream) {
  M2H_REQUIRE(adv && gmean && gvar && n > 0, "adv_apply: bad arguments");
  M2H_LAUNCH(adv_apply_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), adv, gmean, gvar, n, eps);
  return launch_status("adv_apply");
}

int m2h_ppo_loss(const float* values, const float* logp, const float* old_values, const float* returns, const float* adv,
                 const float* old_logp, const float* entropy, float clip, const float* clip_dev, int use_clipped_value_loss,
                 float value_loss_coef, float entropy_coef, float* out, float* g_values, float* g_logp, int n, m2h_stream stream) {
  M2H_REQUIRE(values && logp && old_values && returns && adv && old_logp && out && n > 0, "ppo_loss: bad arguments");
  M2H_LAUNCH(ppo_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), values, logp, old_values, returns, adv, old_logp, clip,
                     clip_dev, use_clipped_value_loss, out, g_values, g_logp, value_loss_coef, entropy, entropy_coef, n);
  return launch_status("ppo_loss");
}

int m2h_gru_gates_bwd(const float* gi, const float* gh_raw, const float* bhh, const float* hprev, const float* mask, const float* dh,
                      float* dgi, float* dpre, float* dhp, float* hpm, int M, int H, m2h_stream stream) {
  M2H_REQUIRE(gi && gh_raw && bhh && hprev && dh && dgi && dpre && dhp && hpm && M > 0 && H > 0, "gru_gates_bwd: bad arguments");
  M2H_LAUNCH(gru_gates_bwd_kernel, dim3(grid_for((size_t)M * H)), dim3(256), 0, as_stream(stream), gi, gh_raw, bhh, hprev, mask, dh,
                     dgi, dpre, dhp, hpm, M, H);
  return launch_status("gru_gates_bwd");
}

int m2h_gru_bwd_combine(const float* a, const float* b, const float* c, const float* mask, float* out, int M, int H, m2h_stream stream) {
  M2H_REQUIRE(b && c && out && M > 0 && H > 0, "gru_bwd_combine: bad arguments");
  M2H_LAUNCH(gru_bwd_combine_kernel, dim3(grid_for((size_t)M * H)), dim3(256), 0, as_stream(stream), a, b, c, mask, out, M, H);
  return launch_status("gru_bwd_combine");
}

int m2h_gru_bwd_rec(const float* dpre, const float* whh_t, const float* a, const float* dhp, const float* mask, float* out, int M, int H,
                    m2h_stream stream) {
  M2H_REQUIRE(dpre && whh_t && dhp && out, "gru_bwd_rec: null pointer");
  M2H_REQUIRE(M > 0 && M <= GB_E && H > 0 && H % 16 == 0, "gru_bwd_rec: needs 1 <= M <= %d rows and H %% 16 == 0 (got M=%d, H=%d)",
              GB_E, M, H);
  M2H_LAUNCH(gru_bwd_rec_kernel, dim3(H / GB_U), dim3(64 * GB_NW), 0, as_stream(stream), dpre, whh_t, a, const_cast<float*>(dhp), mask, out, M, H,
                     static_cast<const float*>(nullptr), static_cast<const float*>(nullptr), static_cast<const float*>(nullptr),
                     static_cast<const float*>(nullptr), static_cast<const float*>(nullptr), static_cast<float*>(nullptr),
                     static_cast<float*>(nullptr), static_cast<float*>(nullptr));
  return launch_status("gru_bwd_rec");
}

int m2h_gru_bwd_step(const float* dpre, const float* whh_t, const float* a, float* dhp, const float* mask, float* out, const float* gi_prev,
                     const float* gh_prev, const float* bhh, const float* hprev_prev, const float* mask_prev, float* dgi_prev, float* dpre_prev,
                     float* hpm_prev, int M, int H, m2h_stream stream) {
  M2H_REQUIRE(dpre && whh_t && dhp && out && gi_prev && gh_prev && bhh && hprev_prev && dgi_prev && dpre_prev && hpm_prev, "gru_bwd_step: null pointer");
  M2H_REQUIRE(M > 0 && M <= GB_E && H > 0 && H % 16 == 0, "gru_bwd_step: needs 1 <= M <= %d rows and H %% 16 == 0 (got M=%d, H=%d)",
              GB_E, M, H);
  M2H_REQUIRE(dpre_prev != dpre, "gru_bwd_step: the previous step's dpre must not alias this step's (every block reads all of it)");
  M2H_LAUNCH(gru_bwd_rec_kernel, dim3(H / GB_U), dim3(64 * GB_NW), 0, as_stream(stream), dpre, whh_t, a, dhp, mask, out, M, H, gi_prev, gh_prev, bhh,
                     hprev_prev, mask_prev, dgi_prev, dpre_prev, hpm_prev);
  return launch_status("gru_bwd_step");
}

int m2h_policy_heads_bwd(const float* logp_all, const float* probs, const long long* actions, const float* g_value, const float* g_logp,
                         const float* g_ent, const float* Wa, const float* Wc, float* dz, float* dfeats, int M, int H, int A, int ZS,
                         m2h_stream stream) {
  M2H_REQUIRE(logp_all && probs && Wa && Wc && dz && dfeats && M > 0 && H > 0 && A > 0 && A <= 8 && ZS >= A + 1 && ZS % 4 == 0,
              "policy_heads_bwd: bad arguments");
  M2H_LAUNCH(policy_heads_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), logp_all, probs, actions, g_value, g_logp,
                     g_ent, Wa, Wc, dz, dfeats, M, H, A, ZS);
  return launch_status("policy_heads_bwd");
}

int m2h_policy_heads_wgrad(const float* feats, const float* dz, float* dw, float* db, int M, int H, int ZS, m2h_stream stream) {
  M2H_REQUIRE(feats && dz && dw && db && M > 0 && H > 0 && H % 64 == 0 && (ZS == 4 || ZS == 8), "policy_heads_wgrad: bad arguments (H %% 64, ZS 4 | 8)");
  if (ZS == 8) M2H_LAUNCH(policy_heads_wgrad_kernel<8>, dim3(H / 64), dim3(256), 0, as_stream(stream), feats, dz, dw, db, M, H);
  else M2H_LAUNCH(policy_heads_wgrad_kernel<4>, dim3(H / 64), dim3(256), 0, as_stream(stream), feats, dz, dw, db, M, H);
  return launch_status("policy_heads_wgrad");
}

#define M2H_PARTS 1024
int m2h_l1_loss(const float* pred, const float* gt, int gt_stride, int gt_off, float* loss, float* grad, float* scratch, size_t n,
                m2h_stream stream) {
  M2H_REQUIRE(pred && gt && loss && scratch && n > 0 && gt_stride > 0 && gt_off >= 0 && gt_off < gt_stride, "l1_loss: bad arguments");
  const unsigned g = grid_for(n, M2H_PARTS);
  M2H_LAUNCH(l1_loss_kernel, dim3(g), dim3(256), 0, as_stream(stream), pred, gt, gt_stride, gt_off, scratch, grad, n);
  M2H_LAUNCH(sum_partials_kernel, dim3(1), dim3(64), 0, as_stream(stream), scratch, (int)g, 1.f / (float)n, loss);
  return launch_status("l1_loss");
}

int m2h_l1_loss_nhwc16(const float* y, const float* gt, int gt_stride, int gt_off, float* loss, float* dy, float* scratch, int B, int T,
                       m2h_stream stream) {
  M2H_REQUIRE(y && gt && loss && scratch && B > 0 && T > 0 && T <= 256 && gt_stride > 0 && gt_off >= 0 && gt_off < gt_stride, "l1_loss_nhwc16: bad arguments");
  const size_t n = (size_t)B * 512 * T;
  const int nrows = B * 32, blocks = nrows < 2048 ? nrows : 2048;
  M2H_LAUNCH(l1_nhwc16_kernel, dim3((unsigned)blocks), dim3(256), 16 * (T + 1) * sizeof(float), as_stream(stream), y, gt, gt_stride, gt_off, T,
                     scratch, dy, 1.f / (float)n, nrows);
  M2H_LAUNCH(sum_partials_wide_kernel, dim3(1), dim3(1024), 0, as_stream(stream), scratch, blocks, 1.f / (float)n, loss);
  return launch_status("l1_loss_nhwc16");
}

int m2h_bin_l1_loss(const float* mix, const float* masks, const float* gt_bin_comps, int Cg, int cstep, float* loss, float* grad_masks,
                    float* scratch, size_t npix, m2h_stream stream) {
  M2H_REQUIRE(mix && masks && gt_bin_comps && loss && scratch && npix > 0 && cstep >= 1 && Cg >= cstep + 1, "bin_l1_loss: bad arguments");
  const unsigned g = grid_for(2 * npix, M2H_PARTS);
  M2H_LAUNCH(bin_l1_kernel, dim3(g), dim3(256), 0, as_stream(stream), mix, masks, gt_bin_comps, Cg, cstep, scratch, grad_masks, npix);
  M2H_LAUNCH(sum_partials_kernel, dim3(1), dim3(64), 0, as_stream(stream), scratch, (int)g, 1.f / (float)(2 * npix), loss);
  return launch_status("bin_l1_loss");
}

int m2h_grad_clip_coef(const float* g, size_t n, float max_norm, float* coef, float* scratch, m2h_stream stream) {
  M2H_REQUIRE(g && coef && scratch && n > 0, "grad_clip_coef: bad arguments");
  const unsigned gr = grid_for(n, M2H_PARTS);
  M2H_LAUNCH(sumsq_kernel, dim3(gr), dim3(256), 0, as_stream(stream), g, n, scratch);
  M2H_LAUNCH(clip_coef_kernel, dim3(1), dim3(64), 0, as_stream(stream), scratch, (int)gr, max_norm, coef);
  return launch_status("grad_clip_coef");
}

int m2h_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step,
                  const float* coef, float gscale, m2h_stream stream) {
  M2H_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  M2H_LAUNCH(adam_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, as_stream(stream), p, g, m, v, n, lr, beta1, beta2, eps, bc1,
                     sqrtf(bc2), coef, gscale, (const float*)nullptr);
  return launch_status("adam_step");
}

int m2h_adam_hyper(float lr, float beta1, float beta2, int step, float* hyper, m2h_stream stream) {
  M2H_REQUIRE(hyper && step >= 1, "adam_hyper: bad arguments");
  // host arithmetic of m2h_adam_step (both entries take the same step); the values travel as kernel arguments: stream-ordered, no
  // host buffer to keep alive, no blocking copy
  M2H_LAUNCH(set3_kernel, dim3(1), dim3(1), 0, as_stream(stream), hyper, lr, 1.f - powf(beta1, (float)step),
                     sqrtf(1.f - powf(beta2, (float)step)));
  return launch_status("adam_hyper");
}

int m2h_adam_step_dev(float* p, float* g, float* m, float* v, size_t n, const float* hyper, float beta1, float beta2, float eps,
                      const float* coef, float gscale, m2h_stream stream) {
  M2H_REQUIRE(p && g && m && v && hyper && n > 0, "adam_step_dev: bad arguments");
  M2H_LAUNCH(adam_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, as_stream(stream), p, g, m, v, n, 0.f, beta1, beta2, eps, 1.f, 1.f, coef,
                     gscale, hyper);
  return launch_status("adam_step_dev");
}

int m2h_sq_stats(const float* pred, const float* gt_comps, int gt_stride, int gt_off, float* stats, int N, int L, m2h_stream stream) {
  M2H_REQUIRE(pred && gt_comps && stats && N > 0 && L > 0 && gt_stride > 0 && gt_off >= 0 && gt_off < gt_stride, "sq_stats: bad arguments");
  M2H_LAUNCH(sq_stats_kernel, dim3(N), dim3(1024), 0, as_stream(stream), pred, gt_comps, gt_stride, gt_off, stats, L);
  return launch_status("sq_stats");
}

int m2h_rewards_from_stats(const float* next_stats, const float* cur_stats, const float* not_done, float* rewards, int N, int L,
                           int quality_improvement, float mult, m2h_stream stream) {
  M2H_REQUIRE(next_stats && not_done && rewards && N > 0 && L > 0 && (!quality_improvement || cur_stats), "rewards_from_stats: bad arguments");
  M2H_LAUNCH(rewards_from_stats_kernel, dim3((N + 63) / 64), dim3(64), 0, as_stream(stream), next_stats, cur_stats, not_done,
                     rewards, N, L, quality_improvement, mult);
  return launch_status("rewards_from_stats");
}

int m2h_gather_envs(const void* src, const long long* perm, void* dst, int T, int N, int Nsel, size_t row_bytes, m2h_stream stream) {
  M2H_REQUIRE(src && perm && dst && T > 0 && N > 0 && Nsel > 0 && row_bytes > 0 && row_bytes % 4 == 0, "gather_envs: bad arguments (row_bytes %% 4)");
  const size_t Lw = row_bytes / 4;
  M2H_LAUNCH(gather_envs_kernel, dim3(grid_for((size_t)T * Nsel * ((Lw & 3) ? Lw : Lw / 4), 8192)), dim3(256), 0, as_stream(stream),
                     static_cast<const uint32_t*>(src), perm, static_cast<uint32_t*>(dst), T, N, Nsel, Lw);
  return launch_status("gather_envs");
}

int m2h_stft_l2(const float* mix, const float* pred, int Cp, const float* gt_comps, int Cg, int nch, int use_mix, float* out, int N,
                int L, m2h_stream stream) {
  M2H_REQUIRE(pred && gt_comps && out && N > 0 && L > 0 && nch > 0 && Cp >= nch && Cg >= 2 * nch && (!use_mix || mix), "stft_l2: bad arguments");
  M2H_LAUNCH(stft_l2_kernel, dim3(N), dim3(1024), 0, as_stream(stream), mix, pred, Cp, gt_comps, Cg, nch, use_mix, out, L);
  return launch_status("stft_l2");
}

int m2h_episode_stats_update(const m2h_episode_stats* st, const float* rewards, const float* dist_probs, const float* bin_losses,
                             const float* mono_losses, const float* monoFromMem_losses, const float* not_done, const float* ndgs,
                             const float* dgs, int N, int A, m2h_stream stream) {
  M2H_REQUIRE(st && rewards && dist_probs && bin_losses && mono_losses && monoFromMem_losses && not_done && N > 0 && A > 0,
              "episode_stats_update: bad arguments");
  const float* const* fields = reinterpret_cast<const float* const*>(st);
  for (size_t i = 0; i < sizeof(m2h_episode_stats) / sizeof(float*); ++i) M2H_REQUIRE(fields[i], "episode_stats_update: null statistics tensor");
  M2H_LAUNCH(episode_stats_kernel, dim3((N + 63) / 64), dim3(64), 0, as_stream(stream), *st, rewards, dist_probs, bin_losses,
                     mono_losses, monoFromMem_losses, not_done, ndgs, dgs, N, A);
  return launch_status("episode_stats_update");
}

int m2h_rows_copy(const m2h_row_copy* items, int n_items, const long long* idx, m2h_stream stream) {
  M2H_REQUIRE(items && n_items > 0 && n_items <= M2H_ROWS_COPY_MAX, "rows_copy: 1..%d items", M2H_ROWS_COPY_MAX);
  RowsCopyArgs a;
  size_t big = 0;
  for (int i = 0; i < n_items; ++i) {
    const m2h_row_copy& it = items[i];
    M2H_REQUIRE(it.src && it.dst && it.bytes > 0 && it.bytes % 4 == 0, "rows_copy: item %d: null pointer or size not a multiple of 4", i);
    M2H_REQUIRE((it.src_slot < 0 && it.dst_slot < 0) || idx, "rows_copy: item %d uses a row index but idx is NULL", i);
    a.item[i] = it;
    big = it.bytes > big ? it.bytes : big;
  }
  const int gx = (int)((big / 16 + 255) / 256 > 256 ? 256 : ((big / 16 + 255) / 256 < 1 ? 1 : (big / 16 + 255) / 256));
  M2H_LAUNCH(rows_copy_kernel, dim3(gx, n_items), dim3(256), 0, as_stream(stream), a, idx);
  return launch_status("rows_copy");
}

int m2h_step_index_advance(long long* idx, int T_pol, int T_sep, m2h_stream stream) {
  M2H_REQUIRE(idx && T_pol > 0 && T_sep > 0, "step_index_advance: bad arguments");
  M2H_LAUNCH(step_index_advance_kernel, dim3(1), dim3(64), 0, as_stream(stream), idx, T_pol, T_sep);
  return launch_status("step_index_advance");
}

int m2h_step_index_advance_rng(long long* idx, int T_pol, int T_sep, unsigned long long* rng_state, unsigned long long rng_inc, m2h_stream stream) {
  M2H_REQUIRE(idx && T_pol > 0 && T_sep > 0 && rng_state, "step_index_advance_rng: bad arguments");
  M2H_LAUNCH(step_index_advance_kernel, dim3(1), dim3(64), 0, as_stream(stream), idx, T_pol, T_sep, rng_state, rng_inc);
  return launch_status("step_index_advance");
}

int m2h_synth_env_step(const long long* actions, long long* node, long long* angle, int n_nodes, int N, m2h_stream stream) {
  M2H_REQUIRE(actions && node && angle && n_nodes > 0 && N > 0, "synth_env_step: bad arguments");
  M2H_LAUNCH(synth_env_step_kernel, dim3((N + 63) / 64), dim3(64), 0, as_stream(stream), actions, node, angle, n_nodes, N);
  return launch_status("synth_env_step");
}

int m2h_synth_env_observe(const m2h_row_copy* items, int n_items, const long long* node, const long long* angle, const long long* audio_idx,
                          int N, m2h_stream stream) {
  M2H_REQUIRE(items && n_items > 0 && n_items <= M2H_ROWS_COPY_MAX && node && angle && audio_idx && N > 0, "synth_env_observe: bad arguments");
  RowsGatherArgs a;
  size_t big = 0;
  for (int i = 0; i < n_items; ++i) {
    M2H_REQUIRE(items[i].src && items[i].dst && items[i].bytes > 0 && items[i].bytes % 4 == 0 && (items[i].src_slot == 0 || items[i].src_slot == 1),
                "synth_env_observe: item %d: null pointer, size not a multiple of 4, or index selector not in {0, 1}", i);
    a.item[i] = items[i];
    big = items[i].bytes > big ? items[i].bytes : big;
  }
  size_t gx = (big / 16 + 255) / 256;
  gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
  M2H_LAUNCH(rows_gather_kernel, dim3((unsigned)gx, (unsigned)N, (unsigned)n_items), dim3(256), 0, as_stream(stream), a, node, angle, audio_idx);
  return launch_status("synth_env_observe");
}

}  // extern "C"
