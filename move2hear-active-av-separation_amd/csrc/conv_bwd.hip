// Backward pieces of the implicit-GEMM convolution (gfx950, fp32 MFMA).
//
//   wgrad   dW[n][k] = sum_m dY[m][n] * A[m][k]      (reduction over output pixels m; A gathered exactly as in the forward)
//           = the weight gradient of Conv2d / Linear in the packed [N][K] layout, K = (tap, channel).
//   dgrad   runs on the FORWARD engine (conv_igemm.hip): the input gradient of a stride-s conv is s*s sub-pixel phase
//           convolutions of dY with the (ci <-> co)-transposed, tap-strided weights; m2h_pack_dgrad_weight lays those out.
//   act_bwd dY * (y > 0 ? 1 : slope)  for the fused ReLU / LeakyReLU epilogues;  bias_grad = column sums of dY.
//
// wgrad tiling: block = BNG (n) x 128*KT (k) output tile, 4 waves, fp32 v_mfma_f32_32x32x2; the reduction runs over 32-pixel
// chunks staged [m][n] / [m][k] in LDS (prefetched through registers); fragments are ds_read_b32 column reads
// (consecutive lanes -> consecutive addresses, conflict free).  The pixel range is split over grid.z; partial tiles go to a
// slab [split][N][Kpad] and an ordered reduce kernel sums them (deterministic, no atomics).
#include <type_traits>

#include "m2h_internal.h"

namespace m2h {

extern thread_local int tl_math_mode;   // conv_igemm.hip: the calling thread's arithmetic

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct WGradP {
  const float* src0;
  const float* src1;
  int C0, C1, Ctot;
  int B, Hi, Wi, Hq, Wq;
  int stride, ntw, ntap, mulh, offh, mulw, offw;
  const float* dy;  // row m -> output pixel (b, q*os+ph, r*os+pw) of an NHWC [B][Ho][Wo][ldy] tensor
  int ldy;
  int Ho, Wo, os, ph, pw;
  int direct;       // 1: pixel index == m (os 1, Ho x Wo == Hq x Wq)
  int N, K, Kpad;   // Kpad = K rounded up to 128
  int M;
  int S;            // splits over m (grid z)
  int chunks;       // ceil(M / 32)
  int ntiles, ktiles;  // output tiles along n and k
  float* ws;        // [S][N][Kpad]  (quad: [4 phases][S][N][Kpad])
  float* dw;        // [N][K]        (quad: the transposed conv's torch layout [Ci][N][4][4])
  int quad;         // 1: the four sub-pixel phases of a ConvTranspose2d(4,2,1) in one launch (grid y = phase: its taps' direction,
                    // its dy rows, its slabs); the reduce kernels write dwp, convT_wgrad_unpack_kernel scatters it into dw
  float* dwp;       // quad: [4 phases][N][K] packed gradients (behind the slabs in the workspace)
  const float* gate;  // optional (image-row 3x3 kernel): the forward output y of the layer, same layout as dy: dy is read as
  float gate_slope;   // dy * (y > 0 ? 1 : gate_slope) -- the backward of the layer's fused ReLU / LeakyReLU without a pass of its own
  int torch_ci;       // > 0: dw is nn.Conv2d's own layout [N][torch_ci][KH][KW] (channels torch_ci .. Ctot-1 of the packed k axis are input padding: dropped)
  // fused input gradient (wgrad3x3_row_dgrad_bf16x3_kernel): dy2 != nullptr -> `dy` is not read; the layer's output gradient is made in the
  // kernel, row by row, as the input gradient of the NEXT 3x3 conv: dy[r][px][c] = sum_{tap, n} dy2[r + 1 - ty][px + 1 - tx][n] w2p[n][tap][c]
  const float* dy2;   // [rows][32][16] NHWC gradient of the next conv's output
  const float* w2p;   // the next conv's packed weight [16][9 * 32] (m2h_pack_conv_weight_ex)
};

// phase (ph, pw) of a quad launch: taps step by 2 ph - 1 / 2 pw - 1 (separator_cnn.py:15-24 as four sub-pixel GEMMs)
struct WPhase {
  int ph, pw, mulh, mulw;
  size_t ws_off;
};
__device__ __forceinline__ WPhase wgrad_phase(const WGradP& p) {
  WPhase w{p.ph, p.pw, p.mulh, p.mulw, 0};
  if (p.quad) {
    const int phase = blockIdx.y;
    w.ph = phase >> 1;
    w.pw = phase & 1;
    w.mulh = 2 * w.ph - 1;
    w.mulw = 2 * w.pw - 1;
    w.ws_off = (size_t)phase * p.S * p.N * p.Kpad;
  }
  return w;
}
constexpr int WK = 128;  // k sub-tile (one 16-byte segment per thread of a 32-thread row group)
constexpr int WM = 32;   // pixels per reduction chunk

// BNG = n extent of the block (32 | 128); KT = number of 128-wide k sub-tiles of the block (k extent 128*KT).
// Narrow layers (N <= 32) would give a wave ONE 32x32 fragment per chunk (16 MFMAs beside ~300 other instructions: the first
// version ran issue-bound at 30 % matrix-pipe utilisation); with KT = 2 or 3 a wave owns KT fragments that share one dY
// operand, the input rows are fetched once per chunk instead of once per k-tile, and the row bookkeeping is amortised.
// NST = LDS stages (2: one barrier per chunk; 1: two barriers, for the wide-k blocks whose tile would not fit twice).
template <int BNG, int KT, int NST>
__global__ __launch_bounds__(256) void wgrad_kernel(const WGradP p) {
  constexpr int WKB = WK * KT;                    // k extent of the block
  constexpr int WN_ = (BNG == 128) ? 2 : 1;       // waves along n
  constexpr int WK_ = 4 / WN_;                    // waves along k
  constexpr int TN = BNG / WN_, TK = WKB / WK_;   // wave tile
  constexpr int FN = TN / 32, FK = TK / 32;
  constexpr int YSEG = BNG / 4;                   // 16-byte segments per dY row
  constexpr int YR = (WM * YSEG + 255) / 256;     // dY segments per thread
  constexpr int YSTEP = 256 / YSEG;
  static_assert(TK % 32 == 0 && FK >= 1, "wave k extent must be whole fragments");
  __shared__ __attribute__((aligned(16))) float Ys[NST][WM * BNG];
  __shared__ __attribute__((aligned(16))) float As[NST][WM * WKB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WK_, wk = wave % WK_;
  const WPhase wp_ = wgrad_phase(p);
  // 1-D grid, XCD-aware: the (n-tile, k-tile) blocks of one pixel split are consecutive blocks of ONE XCD (L % 8), so the
  // split's input rows and dY rows meet in that XCD's L2
  const int L = blockIdx.x;
  const int tiles = p.ntiles * p.ktiles;
  int tile, split;
  if (p.S >= 8) {
    const int idx = L >> 3;
    tile = idx % tiles;
    split = (idx / tiles) * 8 + (L & 7);
    if (split >= p.S) return;  // padding blocks of the XCD map (whole block, before any barrier)
  } else {  // few splits (short M): plain order, tiles spread over all XCDs
    tile = L % tiles;
    split = L / tiles;
  }
  const int n0 = (tile / p.ktiles) * BNG;
  const int k0 = (tile % p.ktiles) * WKB;
  const int c0 = (int)(((long)p.chunks * split) / p.S), c1 = (int)(((long)p.chunks * (split + 1)) / p.S);

  // this thread's fixed A columns (one per k sub-tile): decode (tap, channel) once
  const int aseg = tid & 31;  // 32 segments of 4 floats = 128 k
  const int arow = tid >> 5;  // 0..7, rows arow + 8*i
  bool kok[KT];
  int dh[KT], dw[KT], Cs[KT], cc[KT];
  const float* src[KT];
#pragma unroll
  for (int c = 0; c < KT; ++c) {
    const int k = k0 + c * WK + aseg * 4;
    kok[c] = k < p.K;
    int tap = 0, ci = k;
    if (p.ntap > 1) {
      tap = (unsigned)k / (unsigned)p.Ctot;
      ci = k - tap * p.Ctot;
    }
    const int th = (unsigned)tap / (unsigned)p.ntw, tw = tap - th * p.ntw;
    dh[c] = th * wp_.mulh + p.offh;
    dw[c] = tw * wp_.mulw + p.offw;
    src[c] = p.src0;
    Cs[c] = p.C0;
    cc[c] = ci;
    if (ci >= p.C0 && p.src1 != nullptr) {  // (padding columns k >= K of a single-source conv keep src0: their loads are masked, not skipped)
      src[c] = p.src1;
      Cs[c] = p.C1;
      cc[c] = ci - p.C0;
    }
  }
  const int yseg = tid % YSEG, yrow0 = tid / YSEG;  // dY: rows yrow0 + YSTEP*i

  f32x16 acc[FN][FK];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FK; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  f32x4 ra[4][KT], ry[YR];
  unsigned okm = 0;  // validity bits of the staged registers (A: bit i*KT+c, dY: bit 16+i); selects happen at the LDS write
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // Row state (b, q, r) of the rows this thread stages, advanced by 32 pixels per chunk WITHOUT divisions.
  // 32 = d_b * Hq*Wq + d_q * Wq + d_r  (uniform), so one conditional carry per digit suffices.
  const int d_r = WM % p.Wq, d_q = (WM / p.Wq) % p.Hq, d_b = WM / (p.Wq * p.Hq);
  struct Row { int m, b, q, r; };
  auto row_init = [&](int m) {
    Row w;
    w.m = m;
    w.r = m % p.Wq;
    const int t = m / p.Wq;
    w.q = t % p.Hq;
    w.b = t / p.Hq;
    return w;
  };
  auto row_next = [&](Row& w) {
    w.m += WM;
    w.r += d_r;
    const int c1_ = w.r >= p.Wq ? 1 : 0;
    w.r -= c1_ ? p.Wq : 0;
    w.q += d_q + c1_;
    const int c2_ = w.q >= p.Hq ? 1 : 0;
    w.q -= c2_ ? p.Hq : 0;
    w.b += d_b + c2_;
  };
  Row rowA[4], rowY[YR];
#pragma unroll
  for (int i = 0; i < 4; ++i) rowA[i] = row_init(c0 * WM + arow + 8 * i);
#pragma unroll
  for (int i = 0; i < YR; ++i) rowY[i] = row_init(c0 * WM + yrow0 + YSTEP * i);
  const int ny = n0 + yseg * 4;
  const bool yvec = ny + 3 < p.N && (p.ldy & 3) == 0;

  // loads the chunk the row state points at (unconditional loads from a clamped offset; no divergent branches), then advances
  auto load_chunk = [&]() {
    okm = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const Row& w = rowA[i];
      const int bpix = w.b * p.Hi * p.Wi, qs = w.q * p.stride, rs = w.r * p.stride;
      const bool rok = w.m < p.M;
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        const int ih = qs + dh[c], iw = rs + dw[c];
        const bool ok = kok[c] && rok && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
        const size_t off = ok ? ((size_t)(bpix + ih * p.Wi + iw)) * (size_t)Cs[c] + (size_t)cc[c] : (size_t)0;
        ra[i][c] = *reinterpret_cast<const f32x4*>(src[c] + off);
        okm |= ok ? (1u << (i * KT + c)) : 0u;
      }
      row_next(rowA[i]);
    }
#pragma unroll
    for (int i = 0; i < YR; ++i) {
      const Row& w = rowY[i];
      const int row = yrow0 + YSTEP * i;
      const bool ok = row < WM && w.m < p.M && ny < p.N;
      size_t pix = (size_t)w.m;
      if (!p.direct) pix = ((size_t)w.b * p.Ho + (size_t)(w.q * p.os + wp_.ph)) * p.Wo + (size_t)(w.r * p.os + wp_.pw);
      const float* yp = p.dy + (ok ? pix * p.ldy + ny : (size_t)0);
      if (yvec) {
        ry[i] = *reinterpret_cast<const f32x4*>(yp);
      } else {  // ragged N or unaligned rows (heads): scalar tail, block-uniform branch
        ry[i] = zero4;
        if (ok)
          for (int j = 0; j < 4; ++j)
            if (ny + j < p.N) ry[i][j] = yp[j];
      }
      okm |= ok ? (1u << (16 + i)) : 0u;
      row_next(rowY[i]);
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < KT; ++c)
        *reinterpret_cast<f32x4*>(&As[buf][(arow + 8 * i) * WKB + c * WK + aseg * 4]) = (okm & (1u << (i * KT + c))) ? ra[i][c] : zero4;
#pragma unroll
    for (int i = 0; i < YR; ++i) {
      const int row = yrow0 + YSTEP * i;
      if (row < WM) *reinterpret_cast<f32x4*>(&Ys[buf][row * BNG + yseg * 4]) = (okm & (1u << (16 + i))) ? ry[i] : zero4;
    }
  };
  const int fi = lane & 31, fh = lane >> 5;
  // The fragment reads of half a chunk are issued together and the MFMAs follow (the first version's read -> wait -> MFMA
  // chain exposed the LDS latency 16 times per chunk).
  auto compute = [&](int buf) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float av[WM / 4][FN], bv[WM / 4][FK];
#pragma unroll
      for (int j = 0; j < WM / 4; ++j) {
        const int m = 2 * (half * (WM / 4) + j) + fh;
#pragma unroll
        for (int x = 0; x < FN; ++x) av[j][x] = Ys[buf][m * BNG + wn * TN + x * 32 + fi];
#pragma unroll
        for (int x = 0; x < FK; ++x) bv[j][x] = As[buf][m * WKB + wk * TK + x * 32 + fi];
      }
#pragma unroll
      for (int j = 0; j < WM / 4; ++j)
#pragma unroll
        for (int x = 0; x < FN; ++x)
#pragma unroll
          for (int y = 0; y < FK; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][x], bv[j][y], acc[x][y], 0, 0, 0);
    }
  };

  if (c0 < c1) {
    load_chunk();
    store_chunk(0);
    __syncthreads();
    if constexpr (NST == 2) {
      int cur = 0;
      for (int c = c0; c + 1 < c1; ++c) {  // straight-line body; the last chunk is peeled
        load_chunk();
        compute(cur);
        store_chunk(cur ^ 1);
        __syncthreads();
        cur ^= 1;
      }
      compute(cur);
    } else {
      for (int c = c0; c + 1 < c1; ++c) {
        load_chunk();
        compute(0);
        __syncthreads();  // everyone is done reading the stage
        store_chunk(0);
        __syncthreads();
      }
      compute(0);
    }
  }

  // partial tile -> slab[split][n][k]
  float* slab = p.ws + wp_.ws_off + (size_t)split * p.N * p.Kpad;
  const int col = lane & 31, rhalf = (lane >> 5) * 4;
#pragma unroll
  for (int x = 0; x < FN; ++x)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = n0 + wn * TN + x * 32 + (e & 3) + 8 * (e >> 2) + rhalf;
      if (n >= p.N) continue;
#pragma unroll
      for (int y = 0; y < FK; ++y) {
        const int kk = k0 + wk * TK + y * 32 + col;
        if (kk < p.Kpad) slab[(size_t)n * p.Kpad + kk] = acc[x][y][e];
      }
    }
}

// Weight gradient of a 3x3 / stride 1 / pad 1 convolution over 32-channel, 32-pixel-wide images (both AcousticMem convs,
// rl/models/memory_nets.py:11-16, at 1.7 M pixels per update_sep epoch): the general kernel above gathers the nine taps of every
// pixel separately (1.15 KB per pixel through L2 -> LDS, 4.4 TB/s at 441 us) and pads N = 16 to a 32-wide fragment.  Here a
// reduction chunk is one IMAGE ROW: the three input rows it touches are staged once as a zero-padded 3 x 34-pixel patch (the
// nine taps are row / column shifts of that patch: 400 B per pixel), each wave owns 8 of the row's 32 pixels and ALL nine
// tap fragments of the output (no k padding: K = 288 exactly), and N <= 16 runs on v_mfma_f32_16x16x4_f32 (half the matrix
// work).  The four waves' partial tiles meet through LDS in wave order; splits over rows go to the usual slab + ordered reduce.
template <int FR>
__global__ __launch_bounds__(256) void wgrad3x3_row_kernel(const WGradP p) {
  constexpr int W = 32, C = 32, PW = W + 2;
  constexpr int CS = FR == 32 ? 32 : 48;            // patch pixel stride (floats): conflict-free fragment reads for both shapes
  constexpr int KH = 32 / FR;                       // channel halves per tap (16-wide fragments: 2)
  constexpr int KF = 9 * KH;                        // accumulator fragments per wave
  constexpr int MS = FR == 32 ? 2 : 4;              // pixels contracted per MFMA
  constexpr int STEPS = 8 / MS;                     // a wave owns 8 pixels of the row
  constexpr int NE = FR == 32 ? 16 : 4;
  constexpr int NPL = (3 * PW * 8 + 255) / 256;     // 16-byte patch loads per thread (816 in all)
  using AccT = typename std::conditional<FR == 32, f32x16, f32x4>::type;
  __shared__ __attribute__((aligned(16))) float Ps[2][3 * PW * CS];
  __shared__ __attribute__((aligned(16))) float Ys[2][W * FR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.x;
  const int c0 = (int)(((long)p.chunks * split) / p.S), c1 = (int)(((long)p.chunks * (split + 1)) / p.S);

  f32x4 rp[NPL], ry;
  unsigned okm = 0;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int yrow = tid / (FR / 4), yseg = tid % (FR / 4);      // dY: 32 rows x FR/4 segments (FR = 16: the first 128 threads)
  auto load_chunk = [&](int c) {
    const int b = c / p.Hq, q = c - b * p.Hq;
    okm = 0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int i = tid + 256 * j;
      const int l = i >> 3, seg = i & 7;
      const int pr = l / PW, pc = l - pr * PW;
      const int ih = q + pr - 1, iw = pc - 1;
      const bool ok = i < 3 * PW * 8 && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)W;
      const size_t off = ok ? ((size_t)(b * p.Hi + ih) * W + iw) * C + seg * 4 : (size_t)0;
      rp[j] = *reinterpret_cast<const f32x4*>(p.src0 + off);
      okm |= ok ? (1u << j) : 0u;
    }
    const bool yok = yrow < W && yseg * 4 < p.N;
    ry = *reinterpret_cast<const f32x4*>(p.dy + (yok ? ((size_t)c * W + yrow) * p.ldy + yseg * 4 : (size_t)0));
    if (p.gate != nullptr) {   // m2h_act_bwd folded into the load: same values, no 3-tensor pass of its own
      const f32x4 gy = *reinterpret_cast<const f32x4*>(p.gate + (yok ? ((size_t)c * W + yrow) * p.ldy + yseg * 4 : (size_t)0));
#pragma unroll
      for (int e = 0; e < 4; ++e) ry[e] = gy[e] > 0.f ? ry[e] : ry[e] * p.gate_slope;
    }
    okm |= yok ? (1u << 8) : 0u;
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int i = tid + 256 * j;
      if (i < 3 * PW * 8) *reinterpret_cast<f32x4*>(&Ps[buf][(i >> 3) * CS + (i & 7) * 4]) = (okm & (1u << j)) ? rp[j] : zero4;
    }
    if (yrow < W) *reinterpret_cast<f32x4*>(&Ys[buf][yrow * FR + yseg * 4]) = (okm & (1u << 8)) ? ry : zero4;
  };

  AccT acc[KF];
#pragma unroll
  for (int f = 0; f < KF; ++f)
#pragma unroll
    for (int e = 0; e < NE; ++e) acc[f][e] = 0.f;
  const int fi = lane & (FR - 1), fq = lane / FR;   // fragment row/column, pixel inside the MFMA's contraction
  const int m0 = wave * 8;
  auto compute = [&](int buf) {
    float av[STEPS];
#pragma unroll
    for (int st = 0; st < STEPS; ++st) av[st] = Ys[buf][(m0 + MS * st + fq) * FR + fi];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ty = t / 3, tx = t - 3 * ty;
#pragma unroll
      for (int h = 0; h < KH; ++h) {
        float bv[STEPS];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) bv[st] = Ps[buf][(ty * PW + m0 + MS * st + fq + tx) * CS + h * FR + fi];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
          if constexpr (FR == 32)
            acc[t * KH + h] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st], bv[st], acc[t * KH + h], 0, 0, 0);
          else
            acc[t * KH + h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st], bv[st], acc[t * KH + h], 0, 0, 0);
        }
      }
    }
  };

  if (c0 < c1) {
    load_chunk(c0);
    store_chunk(0);
    __syncthreads();
    int cur = 0;
    for (int c = c0; c + 1 < c1; ++c) {
      load_chunk(c + 1);
      compute(cur);
      store_chunk(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
    compute(cur);
  }
  __syncthreads();   // the stages become the cross-wave scratch

  // the four waves' partial tiles -> one tile, fragment by fragment: R[wave][n][k] in LDS, summed in wave order
  float* R = &Ps[0][0];                               // 4 x FR x FR floats <= 16 KB
  float* slab = p.ws + (size_t)split * p.N * p.Kpad;
#pragma unroll
  for (int f = 0; f < KF; ++f) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int n = FR == 32 ? (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) : (lane >> 4) * 4 + e;
      R[(wave * FR + n) * FR + fi] = acc[f][e];
    }
    __syncthreads();
    for (int i = tid; i < FR * FR; i += 256) {
      const int n = i / FR, kk = i - n * FR;
      const float v = (R[i] + R[FR * FR + i]) + (R[2 * FR * FR + i] + R[3 * FR * FR + i]);
      if (n < p.N) slab[(size_t)n * p.Kpad + (f / KH) * C + (f % KH) * FR + kk] = v;
    }
    __syncthreads();
  }
}

// The same weight gradient in bf16x3 arithmetic (M2H_MATH_BF16X3: lo*hi + hi*lo + hi*hi on the bf16 matrix pipe, fp32 accumulate).
// The reduction runs over PIXELS, so both operands of v_mfma_f32_16x16x32_bf16 (a lane holds eight consecutive k of its row) are
// needed pixel-contiguous: an image row of x is staged TRANSPOSED and split, XT[channel][32 pixels] as [hi | lo] bf16 (one MFMA
// contracts the whole 32-pixel row), and so is the row of dY, YT[n][32 pixels].  A tap's column shift is applied to dY instead of
// x -- dW[n][ty][tx][c] = sum_px' dY[px' - tx + 1][n] x[row + ty - 1][px'][c] -- and made in registers (a 16-byte fragment + the
// neighbouring dword, v_alignbit), so x rows are staged once, unshifted, in a ring of four (step c reads rows c - 1, c, c + 1 and
// row c + 2 arrives), and rows outside the image are skipped rather than staged as zeros.  Wave (nh, ch) owns the 16 x 16 tiles
// (n half, channel half) of all nine taps (N <= 16: channel half x taps 0-4 / 5-8): no cross-wave reduction.  One barrier per row.
// The fp32-MFMA kernel above is matrix-bound at 1.7 M pixels (324 us for the 32 x 288 gradient, 62 % of the fp32 peak); this one
// leaves the layer to its HBM stream (x + dY + gate: 660 MB).
constexpr int WRB_RS = 144;                         // row stride of the transposed stages, bytes: [hi 64 | lo 64 | 16]: 9 x 16 (odd)
template <int FR>
__global__ __launch_bounds__(256, 3) void wgrad3x3_row_bf16x3_kernel(const WGradP p) {
  constexpr int W = 32, C = 32;
  __shared__ __attribute__((aligned(16))) char XT[4][C * WRB_RS];      // ring over image rows (slot = row & 3)
  __shared__ __attribute__((aligned(16))) char YT[2][FR * WRB_RS];     // dY rows (slot = row & 1)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.x;
  const int c0 = (int)(((long)p.chunks * split) / p.S), c1 = (int)(((long)p.chunks * (split + 1)) / p.S);
  const int rows_total = p.B * p.Hq;
  const int fi = lane & 15, kq = lane >> 4;
  const int nh = FR == 32 ? (wave >> 1) : 0, ch = wave & 1;
  const int t_lo = FR == 32 ? 0 : ((wave >> 1) ? 5 : 0), t_hi = FR == 32 ? 9 : ((wave >> 1) ? 9 : 5);

  // staging: thread (pixel = tid / 8, quad = tid % 8) moves 16 bytes = 4 channels of one pixel
  const int spx = tid >> 3, sq = tid & 7;
  f32x4 rx, ry;
  bool okx = false, oky = false;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto load_x = [&](int r) {                         // image row r (global row index b * Hq + ih)
    okx = r >= 0 && r < rows_total;
    rx = *reinterpret_cast<const f32x4*>(p.src0 + (okx ? ((size_t)r * W + spx) * C + sq * 4 : (size_t)0));
  };
  auto load_y = [&](int r) {
    oky = r < rows_total && sq * 4 < p.N;
    const size_t off = oky ? ((size_t)r * W + spx) * p.ldy + sq * 4 : (size_t)0;
    ry = *reinterpret_cast<const f32x4*>(p.dy + off);
    if (p.gate != nullptr) {   // m2h_act_bwd folded into the load
      const f32x4 gy = *reinterpret_cast<const f32x4*>(p.gate + off);
#pragma unroll
      for (int e = 0; e < 4; ++e) ry[e] = gy[e] > 0.f ? ry[e] : ry[e] * p.gate_slope;
    }
  };
  auto store_t = [&](char* base, f32x4 v) {          // rows 4 sq .. 4 sq + 3 of a transposed stage, column spx
    const bf16x4 hi = __builtin_convertvector(v, bf16x4);
    const f32x4 hf = __builtin_convertvector(hi, f32x4);
    const bf16x4 lo = __builtin_convertvector(v - hf, bf16x4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      char* d = base + (sq * 4 + e) * WRB_RS + spx * 2;
      *reinterpret_cast<__bf16*>(d) = hi[e];
      *reinterpret_cast<__bf16*>(d + 64) = lo[e];
    }
  };
  auto store_x = [&](int r) { if (okx) store_t(XT[r & 3], rx); };
  auto store_y = [&](int r) { if (sq * 4 < FR) store_t(YT[r & 1], oky ? ry : zero4); };

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = zero4;
  auto mma = [&](const f32x4& a, const f32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto compute = [&](int c) {
    const int q = c % p.Hq;
    // dY fragments of the three column shifts (hi / lo): the unshifted 16 bytes + the dword before / after
    f32x4 ya[3][2];
    const char* yb = YT[c & 1] + (nh * 16 + fi) * WRB_RS + kq * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const char* yp = yb + h * 64;
      const u32x4 d = *reinterpret_cast<const u32x4*>(yp);
      const unsigned before = kq > 0 ? *reinterpret_cast<const unsigned*>(yp - 4) : 0u;
      const unsigned after = kq < 3 ? *reinterpret_cast<const unsigned*>(yp + 16) : 0u;
      u32x4 l, r;                                    // l: element k takes dY[k + 1] (tap column 0); r: dY[k - 1] (tap column 2)
      l[0] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
      l[1] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
      l[2] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
      l[3] = __builtin_amdgcn_alignbit(after, d[3], 16);
      r[0] = __builtin_amdgcn_alignbit(d[0], before, 16);
      r[1] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
      r[2] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
      r[3] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
      ya[0][h] = __builtin_bit_cast(f32x4, l);
      ya[1][h] = __builtin_bit_cast(f32x4, d);
      ya[2][h] = __builtin_bit_cast(f32x4, r);
    }
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      if (3 * ty + 3 <= t_lo || 3 * ty >= t_hi) continue;            // (wave-uniform: none of this wave's taps)
      const int ih = q + ty - 1;
      if ((unsigned)ih >= (unsigned)p.Hq) continue;                  // the row above / below the image: zeros
      const char* xb = XT[(c + ty - 1) & 3] + (ch * 16 + fi) * WRB_RS + kq * 16;
      const f32x4 bh = *reinterpret_cast<const f32x4*>(xb), bl = *reinterpret_cast<const f32x4*>(xb + 64);
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int t = ty * 3 + tx;
        if (t < t_lo || t >= t_hi) continue;
        mma(ya[tx][1], bh, acc[t]);
        mma(ya[tx][0], bl, acc[t]);
        mma(ya[tx][0], bh, acc[t]);
      }
    }
  };

  if (c0 < c1) {
    // prologue: rows c0 - 1, c0, c0 + 1 of x and row c0 of dY staged; rows c0 + 2 / c0 + 1 in registers
#pragma unroll 1
    for (int r = c0 - 1; r <= c0 + 1; ++r) {
      load_x(r);
      store_x(r);
    }
    load_y(c0);
    store_y(c0);
    load_x(c0 + 2);
    load_y(c0 + 1);
    __syncthreads();
#pragma unroll 1
    for (int c = c0; c < c1; ++c) {
      store_x(c + 2);                  // slot (c + 2) & 3 held row c - 2: last read in step c - 1, before that step's barrier
      store_y(c + 1);
      if (c + 1 < c1) {
        load_x(c + 3);
        load_y(c + 2);
      }
      compute(c);
      __syncthreads();
    }
  }

  // each wave owns its tiles: slab[split][n][t * 32 + ch * 16 + col]
  float* slab = p.ws + (size_t)split * p.N * p.Kpad;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if (t < t_lo || t >= t_hi) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = nh * 16 + kq * 4 + e;
      if (n < p.N) slab[(size_t)n * p.Kpad + t * C + ch * 16 + fi] = acc[t][e];
    }
  }
}

// The bf16x3 image-row weight gradient with the INPUT GRADIENT OF THE NEXT CONV fused in (update_sep's backward through AcousticMem,
// memory_nets.py:11-16: conv 32 -> 32, ReLU, conv 32 -> 16): the gradient this layer's weight gradient contracts with -- d loss / d h,
// h = ReLU(conv0(x)) -- is itself conv1's input gradient, a 3x3 convolution of d loss / d y (16 channels) with conv1's weights.  As two
// launches that tensor (220 MB at 1680 samples) is written by the one and read back, with the ReLU gate's 220 MB, by the other; here a
// block makes each image row of it on the matrix pipe from a ring of three staged rows of d loss / d y (110 MB in all) and conv1's
// weights held in registers as A fragments, gates it with h and writes it -- transposed and split, as the weight-gradient MFMAs want
// their pixel-contracted operand -- straight into the LDS stage the plain kernel fills from memory.  Per row: 15 more MFMAs per wave,
// no second barrier (five-slot rings: row c + 3 is staged while rows c - 1 .. c + 2 are read).
// D[c][px] = sum_k A[c][k] B[k][px], k = (tap, n): lane (row c = lane & 15, k-quarter kq) of k-step s holds tap 2 s + (kq >> 1),
// channels 8 (kq & 1) .. + 7 of d loss / d y at pixel (r + 1 - ty, px + 1 - tx) -- one 16-byte read of the ring ([hi 16 | lo 16] bf16 per pixel).
constexpr int WRD_PS = 80;                          // d loss / d y ring: pixel stride, bytes ([hi 32 | lo 32 | 16]: 5 x 16, odd)
constexpr int WRD_RS = 34 * WRD_PS;                 // ring row: 32 pixels + a zero pixel on either side
__global__ __launch_bounds__(256, 3) void wgrad3x3_row_dgrad_bf16x3_kernel(const WGradP p) {
  constexpr int W = 32, C = 32, FR = 32;
  __shared__ __attribute__((aligned(16))) char XT[5][C * WRB_RS];      // x rows, transposed + split (slot = row % 5)
  __shared__ __attribute__((aligned(16))) char YT[2][FR * WRB_RS];     // rows of the fused gradient (slot = row & 1)
  __shared__ __attribute__((aligned(16))) char DY[5][WRD_RS];          // d loss / d y rows, pixel-major + split (slot = row % 5)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.x;
  const int c0 = (int)(((long)p.chunks * split) / p.S), c1 = (int)(((long)p.chunks * (split + 1)) / p.S);
  const int rows_total = p.B * p.Hq;
  const int fi = lane & 15, kq = lane >> 4;
  const int nh = wave >> 1, ch = wave & 1;           // weight-gradient role: (output-channel half, x-channel half)
  const int dch = wave & 1, dpx = wave >> 1;         // input-gradient role: tile (channel half, pixel half) of the 32 x 32 row
  auto slot5 = [](int r) { return (r + 5) % 5; };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // staging roles: x -- thread (pixel tid / 8, quad tid % 8); d loss / d y -- threads 0..127 (pixel tid / 4, quad tid % 4)
  const int spx = tid >> 3, sq = tid & 7;
  const int ypx = tid >> 2, yq = tid & 3;
  f32x4 rx, rd, rg, rg_next;
  bool okx = false, okd = false;
  auto load_x = [&](int r) {
    okx = r >= 0 && r < rows_total;
    rx = *reinterpret_cast<const f32x4*>(p.src0 + (okx ? ((size_t)r * W + spx) * C + sq * 4 : (size_t)0));
  };
  auto load_d = [&](int r) {
    okd = tid < 128 && r >= 0 && r < rows_total;
    rd = *reinterpret_cast<const f32x4*>(p.dy2 + (okd ? ((size_t)r * W + ypx) * 16 + yq * 4 : (size_t)0));
  };
  auto load_g = [&](int r) {                         // the gate (this layer's forward output) at this lane's four accumulator elements of row r
    const bool ok = r >= 0 && r < rows_total;
    return *reinterpret_cast<const f32x4*>(p.gate + (ok ? ((size_t)r * W + dpx * 16 + fi) * C + dch * 16 + kq * 4 : (size_t)0));
  };
  auto split4 = [&](f32x4 v, bf16x4& hi, bf16x4& lo) {
    hi = __builtin_convertvector(v, bf16x4);
    const f32x4 hf = __builtin_convertvector(hi, f32x4);
    lo = __builtin_convertvector(v - hf, bf16x4);
  };
  auto store_x = [&](int r) {                        // rows 4 sq .. 4 sq + 3 of the transposed stage, column spx
    if (!okx) return;
    bf16x4 hi, lo;
    split4(rx, hi, lo);
    char* base = XT[slot5(r)];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      char* d = base + (sq * 4 + e) * WRB_RS + spx * 2;
      *reinterpret_cast<__bf16*>(d) = hi[e];
      *reinterpret_cast<__bf16*>(d + 64) = lo[e];
    }
  };
  auto store_d = [&](int r) {                        // pixel ypx + 1 of the ring row, channels 4 yq .. + 3
    if (!okd) return;
    bf16x4 hi, lo;
    split4(rd, hi, lo);
    char* d = DY[slot5(r)] + (ypx + 1) * WRD_PS + yq * 8;
    *reinterpret_cast<bf16x4*>(d) = hi;
    *reinterpret_cast<bf16x4*>(d + 32) = lo;
  };
  auto mma = [&](const f32x4& a, const f32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };

  // conv1's weights as the A fragments of the input gradient, once per block: row c = dch * 16 + fi, k-step s, this lane's eight k
  f32x4 wa[5][2];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = 2 * s + (kq >> 1), n0 = (kq & 1) * 8;
    f32x4 v0 = zero4, v1 = zero4;
    if (tap < 9) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v0[j] = p.w2p[(size_t)(n0 + j) * (9 * C) + tap * C + dch * 16 + fi];
        v1[j] = p.w2p[(size_t)(n0 + 4 + j) * (9 * C) + tap * C + dch * 16 + fi];
      }
    }
    bf16x4 h0, l0, h1, l1;
    split4(v0, h0, l0);
    split4(v1, h1, l1);
    bf16x8 hh, ll;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hh[j] = h0[j]; hh[4 + j] = h1[j];
      ll[j] = l0[j]; ll[4 + j] = l1[j];
    }
    wa[s][0] = __builtin_bit_cast(f32x4, hh);
    wa[s][1] = __builtin_bit_cast(f32x4, ll);
  }
  // the ring rows' zero pixels (columns -1 and 32): never written again
  for (int i = tid; i < 5 * 2 * (WRD_PS / 16); i += 256) {
    const int sl = i / (2 * (WRD_PS / 16)), rem = i - sl * 2 * (WRD_PS / 16);
    const int side = rem / (WRD_PS / 16), q16 = rem - side * (WRD_PS / 16);
    *reinterpret_cast<f32x4*>(DY[sl] + (side ? 33 : 0) * WRD_PS + q16 * 16) = zero4;
  }

  // image row r of the fused gradient -> YT[r & 1] (gate values of the row in g)
  auto dgrad_row = [&](int r, const f32x4& g) {
    const int q = r % p.Hq;
    f32x4 acc = zero4, acc_b = zero4;                 // two accumulation chains (even / odd k-steps): half the dependent MFMA latency per row
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int tap = 2 * s + (kq >> 1);
      const int ty = tap / 3, tx = tap - 3 * ty;
      const int qq = q + 1 - ty;
      const bool ok = tap < 9 && (unsigned)qq < (unsigned)p.Hq;
      const char* bp = DY[slot5(ok ? r + 1 - ty : r)] + (dpx * 16 + fi + 2 - tx) * WRD_PS + (kq & 1) * 16;
      f32x4 bh = *reinterpret_cast<const f32x4*>(bp), bl = *reinterpret_cast<const f32x4*>(bp + 32);
      bh = ok ? bh : zero4;
      bl = ok ? bl : zero4;
      f32x4& a_ = (s & 1) ? acc_b : acc;
      mma(wa[s][1], bh, a_);
      mma(wa[s][0], bl, a_);
      mma(wa[s][0], bh, a_);
    }
    acc += acc_b;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = g[e] > 0.f ? acc[e] : acc[e] * p.gate_slope;
    bf16x4 hi, lo;
    split4(v, hi, lo);
    char* yb = YT[r & 1] + (dch * 16 + kq * 4) * WRB_RS + (dpx * 16 + fi) * 2;   // rows = channels (kq * 4 + e), column = pixel
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      *reinterpret_cast<__bf16*>(yb + e * WRB_RS) = hi[e];
      *reinterpret_cast<__bf16*>(yb + e * WRB_RS + 64) = lo[e];
    }
  };

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = zero4;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto compute = [&](int c) {                        // the weight gradient's row step: wgrad3x3_row_bf16x3_kernel<32>::compute
    const int q = c % p.Hq;
    f32x4 ya[3][2];
    const char* yb = YT[c & 1] + (nh * 16 + fi) * WRB_RS + kq * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const char* yp = yb + h * 64;
      const u32x4 d = *reinterpret_cast<const u32x4*>(yp);
      const unsigned before = kq > 0 ? *reinterpret_cast<const unsigned*>(yp - 4) : 0u;
      const unsigned after = kq < 3 ? *reinterpret_cast<const unsigned*>(yp + 16) : 0u;
      u32x4 l, r;
      l[0] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
      l[1] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
      l[2] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
      l[3] = __builtin_amdgcn_alignbit(after, d[3], 16);
      r[0] = __builtin_amdgcn_alignbit(d[0], before, 16);
      r[1] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
      r[2] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
      r[3] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
      ya[0][h] = __builtin_bit_cast(f32x4, l);
      ya[1][h] = __builtin_bit_cast(f32x4, d);
      ya[2][h] = __builtin_bit_cast(f32x4, r);
    }
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      const int ih = q + ty - 1;
      if ((unsigned)ih >= (unsigned)p.Hq) continue;
      const char* xb = XT[slot5(c + ty - 1)] + (ch * 16 + fi) * WRB_RS + kq * 16;
      const f32x4 bh = *reinterpret_cast<const f32x4*>(xb), bl = *reinterpret_cast<const f32x4*>(xb + 64);
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int t = ty * 3 + tx;
        mma(ya[tx][1], bh, acc[t]);
        mma(ya[tx][0], bl, acc[t]);
        mma(ya[tx][0], bh, acc[t]);
      }
    }
  };

  if (c0 < c1) {
    // prologue: rows c0 - 1 .. c0 + 2 of x and of d loss / d y staged, the fused gradient's row c0 made; row c0 + 3 in registers
#pragma unroll 1
    for (int r = c0 - 1; r <= c0 + 2; ++r) {
      load_x(r);
      store_x(r);
      load_d(r);
      store_d(r);
    }
    rg = load_g(c0);
    __syncthreads();
    dgrad_row(c0, rg);
    rg = load_g(c0 + 1);
    load_x(c0 + 3);
    load_d(c0 + 3);
    __syncthreads();
#pragma unroll 1
    for (int c = c0; c < c1; ++c) {
      // slots of row c + 3 held row c - 2: last read in step c - 1 (x: its weight-gradient step read rows c - 2 .. c; d loss / d y: the
      // gradient row c was made from rows c - 1 .. c + 1 in step c - 1), before that step's barrier
      store_x(c + 3);
      store_d(c + 3);
      if (c + 1 < c1) {
        rg_next = load_g(c + 2);
        load_x(c + 4);
        load_d(c + 4);
        dgrad_row(c + 1, rg);      // reads ring rows c .. c + 2 (staged in earlier steps) -> YT[(c + 1) & 1], read after this step's barrier
        rg = rg_next;
      }
      compute(c);                  // reads YT[c & 1] (made in the previous step) and x rows c - 1 .. c + 1
      __syncthreads();
    }
  }

  float* slab = p.ws + (size_t)split * p.N * p.Kpad;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = nh * 16 + kq * 4 + e;
      if (n < p.N) slab[(size_t)n * p.Kpad + t * C + ch * 16 + fi] = acc[t][e];
    }
  }
}

// Sum of one slab element over the splits [z0, z1): eight running sums (eight loads in flight per lane), combined pairwise -- the ONE
// order of every many-split reduce below (wgrad_reduce_kernel and the fused re-layout kernels give the same bits).
__device__ __forceinline__ float wgrad_quarter_sum(const float* __restrict__ src, int z0, int z1, size_t zs) {
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int z = z0;
  for (; z + 7 < z1; z += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(z + j) * zs];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += v[j];
  }
  for (; z < z1; ++z) a[0] += src[(size_t)z * zs];
  return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

// dw[n][k] = sum over splits, fixed order.  A block owns 64 consecutive k of one row n; its four waves each sum a quarter of
// the splits (eight loads in flight per lane: wgrad_quarter_sum), then the quarters are combined in wave order.  (One thread per element walking
// all splits serially took 39 us for a 32 x 384 gradient with 512 splits: 36 blocks, one dependent load at a time.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WGradP p) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int kb = (p.K + 63) / 64;
  const int n = blockIdx.x / kb;
  const int k = (blockIdx.x - n * kb) * 64 + lane;
  const int z0 = (int)(((long)p.S * w) / 4), z1 = (int)(((long)p.S * (w + 1)) / 4);
  const size_t zs = (size_t)p.N * p.Kpad;
  const WPhase wp_ = wgrad_phase(p);
  float q = 0.f;
  if (k < p.K) q = wgrad_quarter_sum(p.ws + wp_.ws_off + (size_t)n * p.Kpad + k, z0, z1, zs);
  sh[w][lane] = q;
  __syncthreads();
  if (w == 0 && k < p.K) (p.quad ? p.dwp + (size_t)blockIdx.y * p.N * p.K : p.dw)[(size_t)n * p.K + k] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}

// few splits: one thread per element (the block-per-64-k form above would be tens of thousands of near-empty blocks)
__global__ __launch_bounds__(256) void wgrad_reduce_small_kernel(const WGradP p) {
  const size_t total = (size_t)p.N * p.K;
  const WPhase wp_ = wgrad_phase(p);
  const float* ws = p.ws + wp_.ws_off;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / p.K);
    const int k = (int)(i - (size_t)n * p.K);
    float s = 0.f;
    for (int z = 0; z < p.S; ++z) s += ws[((size_t)z * p.N + n) * p.Kpad + k];
    (p.quad ? p.dwp + (size_t)blockIdx.y * p.N * p.K : p.dw)[i] = s;
  }
}

// The split sum of a 16 x 16 tile of gradient elements, in the order (and so with the bits) of the reduce kernel the launch would
// otherwise take: S < 16 -> 256 threads, one running sum per element (wgrad_reduce_small_kernel); S >= 16 -> 1024 threads, thread
// (quarter, element) sums its quarter of the splits, the quarters meet in LDS as (q0 + q1) + (q2 + q3) (wgrad_reduce_kernel).
// src = this thread's element in split 0 (nullptr: outside the tensor).  Returns the sum in the threads of quarter 0.
template <bool Q4>
__device__ __forceinline__ float wgrad_tile_sum(const float* __restrict__ src, int S, size_t zs, float (*qs)[256]) {
  const int el = threadIdx.x & 255, w = threadIdx.x >> 8;
  if constexpr (!Q4) {
    float s = 0.f;
    if (src != nullptr)
      for (int z = 0; z < S; ++z) s += src[(size_t)z * zs];
    return s;
  } else {
    qs[w][el] = src != nullptr ? wgrad_quarter_sum(src, (int)(((long)S * w) / 4), (int)(((long)S * (w + 1)) / 4), zs) : 0.f;
    __syncthreads();
    return w == 0 ? (qs[0][el] + qs[1][el]) + (qs[2][el] + qs[3][el]) : 0.f;
  }
}

// Transposed-conv weight gradient: split sum AND the scatter to the torch layout dw[ci][co][kh][kw] in one launch (round 4: one node
// less per decoder layer on the training step's chain).  A block owns (co, 16 ci): element (e = kh * 4 + kw, ci) of phase (ph, pw),
// tap (th, tw) is summed over the splits straight from the slabs (16 consecutive ci = 64-byte runs), the tile is transposed through
// LDS and leaves as 16 runs of 64 bytes.
template <bool Q4>
__global__ __launch_bounds__(Q4 ? 1024 : 256) void convT_wgrad_reduce_unpack_kernel(const WGradP p) {
  __shared__ float tile[16][17];
  __shared__ float qs[Q4 ? 4 : 1][256];
  const int cb = (p.Ctot + 15) / 16;
  const int n = blockIdx.x / cb, ci0 = (blockIdx.x - n * cb) * 16;
  const int el = threadIdx.x & 255;
  {
    const int e = el >> 4, ci = ci0 + (el & 15);
    const int kh = e >> 2, kw = e & 3;
    const int ph = (kh & 1) ^ 1, th = (kh == 0 || kh == 3) ? 1 : 0, pw = (kw & 1) ^ 1, tw = (kw == 0 || kw == 3) ? 1 : 0;
    const size_t zs = (size_t)p.N * p.Kpad;
    const float* src = ci < p.Ctot ? p.ws + (size_t)(ph * 2 + pw) * p.S * zs + (size_t)n * p.Kpad + (size_t)(th * 2 + tw) * p.Ctot + ci : nullptr;
    const float v = wgrad_tile_sum<Q4>(src, p.S, zs, qs);
    if (threadIdx.x < 256) tile[el & 15][e] = v;
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    const int cl = el >> 4, e = el & 15;
    if (ci0 + cl < p.Ctot) p.dw[((size_t)(ci0 + cl) * p.N + n) * 16 + e] = tile[cl][e];
  }
}

// Conv2d weight gradient: split sum AND the re-layout packed [n][(tap, c)] -> torch [n][c][tap] in one launch (m2h_conv_wgrad_torch_f32: the
// permute(0, 3, 1, 2).contiguous() copy of the packed gradient was a launch per conv layer of every backward pass).  A block owns
// (n, 16 channels): its output is ONE run of 16 x ntap floats; taps go through the LDS tile sixteen at a time.
template <bool Q4>
__global__ __launch_bounds__(Q4 ? 1024 : 256) void conv_wgrad_reduce_torch_kernel(const WGradP p) {
  __shared__ float tile[16][17];
  __shared__ float qs[Q4 ? 4 : 1][256];
  const int Ci = p.torch_ci;
  const int cb = (Ci + 15) / 16;
  const int n = blockIdx.x / cb, ci0 = (blockIdx.x - n * cb) * 16;
  const size_t zs = (size_t)p.N * p.Kpad;
  const int el = threadIdx.x & 255;
  // gridDim.y > 1: a block takes every gridDim.y-th group of 16 taps (layers with few (n, 16-channel) blocks and many taps and splits --
  // VisualCNN's first conv: 32 blocks summing 64 taps x 500 splits took 37-41 us at the end of the policy epoch's longest branch)
  for (int t0 = 16 * blockIdx.y; t0 < p.ntap; t0 += 16 * gridDim.y) {
    {
      const int t = t0 + (el >> 4), ci = ci0 + (el & 15);
      const float* src = (t < p.ntap && ci < Ci) ? p.ws + (size_t)n * p.Kpad + (size_t)t * p.Ctot + ci : nullptr;
      const float v = wgrad_tile_sum<Q4>(src, p.S, zs, qs);
      if (threadIdx.x < 256) tile[el & 15][el >> 4] = v;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
      const int cl = el >> 4, t = t0 + (el & 15);
      if (ci0 + cl < Ci && t < p.ntap) p.dw[((size_t)n * Ci + ci0 + cl) * p.ntap + t] = tile[cl][el & 15];
    }
    __syncthreads();
  }
}

// (tuning knob g_wgrad_blocks: thread-local, m2h_internal.h) tuning knob (m2h_tuning_set 11): target block count of a weight-gradient launch
// (tuning knob g_wgrad_row3x3: thread-local, m2h_internal.h) -1: never use the image-row 3x3 kernel (m2h_tuning_set 21)

// block shape for (N, K): n extent, k sub-tiles per block, blocks along k
static void wgrad_cfg(int N, int K, int& bng, int& kt, int& ktiles, long M = 1L << 30) {
  const int kt128 = (K + WK - 1) / WK;
  bng = N > 64 ? 128 : (N > 32 ? 64 : 32);        // (64: round 4 -- a 64-channel layer on the 128-wide block spent half its MFMAs on padding)
  // a few hundred rows (the update batch's Linear layers: 280 x 1536 x 1536): the reduction is nine chunks long and a block's time is its
  // MFMAs -- 64-wide blocks, twice as many, each half as long: 33 -> 28, 21 -> 14, 31 -> 27 us per policy epoch (knob 25 = -1: the 128-wide blocks)
  if (M <= 1024 && N > 64 && K <= 2048 && g_wgrad_small_m >= 0) bng = 64;   // (K <= 2048: the 4608-deep full-spatial conv re-reads its input rows once per n-block: 34 -> 54 us)
  kt = bng == 32 ? (kt128 >= 3 ? 3 : kt128) : (bng == 64 ? (kt128 >= 2 && N <= 64 ? 2 : 1) : 1);  // narrow layers: up to three k sub-tiles per block share the dY operand
  // ... unless two sub-tiles per block leave fewer padding columns (K = 512: two blocks of 256 instead of two of 384 -- the last decoder
  // stage's weight gradient, 65 536 pixels x 512 x 16 | 32, spent a third of its MFMAs and input loads on columns beyond K; knob 12 = -1: the old rule)
  if (bng == 32 && kt == 3 && tl_tuning.v[12] >= 0 && ((kt128 + 1) / 2) * 2 < ((kt128 + 2) / 3) * 3) kt = 2;
  ktiles = (kt128 + kt - 1) / kt;
}

// the shapes the image-row 3x3 kernels take (geometry only: the launch adds its conditions on ldy and the knob)
static bool wgrad_row3x3_shape(const m2h_conv_args& a) {
  return a.nth == 3 && a.ntw == 3 && a.stride == 1 && a.mulh == 1 && a.mulw == 1 && a.offh == -1 && a.offw == -1 && a.C0 == 32 && a.C1 == 0 &&
         a.Wq == 32 && a.Wi == 32 && a.Hq == a.Hi && a.os == 1 && a.ph == 0 && a.pw == 0 && a.Ho == a.Hq && a.Wo == a.Wq && a.N <= 32 && a.N % 4 == 0;
}

static int wgrad_splits(long M, int N, int K, bool row3x3 = false) {
  int bng, kt, ktiles;
  wgrad_cfg(N, K, bng, kt, ktiles, M);
  const long tiles = ((N + bng - 1) / bng) * (long)ktiles;
  const long chunks = (M + WM - 1) / WM;
  // one wave front, no tail round: 3 resident blocks per CU for the one-sub-tile kernels (40 / 64 KB LDS, <= 176 VGPRs), 2 for
  // the wide-k ones (196-240 VGPRs); the image-row kernels (one block per split, 46 KB LDS) fill 3 per CU as well
  // (round 4: 512 for every tiled shape -- at 768 the one-sub-tile kernels' extra splits cost more in slabs and reduce than the third
  // resident block returned: the pre-training step 2.44 -> 2.41 ms)
  const long target = g_wgrad_blocks > 0 ? g_wgrad_blocks : (row3x3 ? 768 : 512);
  long S = (target + tiles - 1) / tiles;
  if (S > chunks / 4) S = chunks / 4;   // at least 4 chunks per split
  if (S > 1024) S = 1024;
  if (S < 1) S = 1;
  return (int)S;
}

size_t conv_wgrad_workspace_bytes(const m2h_conv_args& a) {
  const long M = (long)a.B * a.Hq * a.Wq;
  const int K = a.nth * a.ntw * (a.C0 + a.C1);
  const int Kpad = (K + WK - 1) / WK * WK;
  return (size_t)wgrad_splits(M, a.N, K, wgrad_row3x3_shape(a)) * a.N * Kpad * sizeof(float);
}

// quad: the four phases of a ConvTranspose2d(4,2,1) in one launch (a = the geometry of one phase: taps 2x2, stride 1, os 2,
// Ho = 2 Hi; its ph / pw / mulh / mulw are ignored), dw in the torch layout, workspace four times the single-phase size
// dy2 / w2p: the fused input gradient (m2h_conv_wgrad_dgrad_fused_f32): `dy` is then made inside the image-row kernel from the NEXT conv's
// output gradient dy2 [B][H][W][16] and packed weight w2p [16][9 * 32] (bf16x3 arithmetic, N = C0 = 32, gate required) and may be NULL
int conv_wgrad_f32(const m2h_conv_args& a, const float* dy, int ldy, float* dw, hipStream_t st, bool quad = false, const float* gate = nullptr,
                   float gate_slope = 1.f, int torch_ci = 0, const float* dy2 = nullptr, const float* w2p = nullptr) {
  M2H_REQUIRE(torch_ci >= 0 && torch_ci <= a.C0 + a.C1 && (!quad || torch_ci == 0), "conv_wgrad: torch_ci (%d) must lie in 1 .. C0 + C1", torch_ci);
  M2H_REQUIRE(a.src0 != nullptr && (dy != nullptr || dy2 != nullptr) && dw != nullptr, "conv_wgrad: null pointer");
  M2H_REQUIRE((dy2 == nullptr) == (w2p == nullptr), "conv_wgrad: fused input gradient needs both dy2 and w2p");
  M2H_REQUIRE(a.conv_transpose == 0, "conv_wgrad: describe a transposed conv by its phase geometry (m2h_convT_wgrad_f32)");
  M2H_REQUIRE(!quad || (a.nth == 2 && a.ntw == 2 && a.stride == 1 && a.os == 2 && a.offh == 0 && a.offw == 0 && a.Hq == a.Hi && a.Wq == a.Wi &&
                        a.Ho == 2 * a.Hi && a.Wo == 2 * a.Wi),
              "convT_wgrad: phase geometry of ConvTranspose2d(4,2,1) expected (taps 2x2, stride 1, os 2, Ho = 2 Hi)");
  M2H_REQUIRE(a.C0 > 0 && a.C0 % 4 == 0 && a.C1 >= 0 && a.C1 % 4 == 0, "conv_wgrad: C0/C1 must be multiples of 4");
  M2H_REQUIRE((a.C1 == 0) == (a.src1 == nullptr), "conv_wgrad: src1/C1 mismatch");
  M2H_REQUIRE(a.B > 0 && a.Hi > 0 && a.Wi > 0 && a.Hq > 0 && a.Wq > 0 && a.N > 0 && a.nth > 0 && a.ntw > 0 && a.stride > 0, "conv_wgrad: bad sizes");
  M2H_REQUIRE(ldy >= a.N || dy2 != nullptr, "conv_wgrad: ldy (%d) < N (%d)", ldy, a.N);
  const long M = (long)a.B * a.Hq * a.Wq;
  M2H_REQUIRE(M < (1L << 30) && (long)a.B * a.Hi * a.Wi < (1L << 30), "conv_wgrad: too many pixels");
  WGradP p;
  p.src0 = a.src0; p.src1 = a.src1; p.C0 = a.C0; p.C1 = a.C1; p.Ctot = a.C0 + a.C1;
  p.B = a.B; p.Hi = a.Hi; p.Wi = a.Wi; p.Hq = a.Hq; p.Wq = a.Wq;
  p.stride = a.stride; p.ntw = a.ntw; p.ntap = a.nth * a.ntw; p.mulh = a.mulh; p.offh = a.offh; p.mulw = a.mulw; p.offw = a.offw;
  p.Ho = a.Ho; p.Wo = a.Wo; p.os = a.os; p.ph = a.ph; p.pw = a.pw;
  p.quad = quad ? 1 : 0;
  const int phases = quad ? 4 : 1;
  p.direct = (a.os == 1 && a.ph == 0 && a.pw == 0 && a.Ho == a.Hq && a.Wo == a.Wq) ? 1 : 0;
  M2H_REQUIRE(a.os >= 1 && (a.Hq - 1) * a.os + a.ph < a.Ho && (a.Wq - 1) * a.os + a.pw < a.Wo, "conv_wgrad: output pixel grid exceeds Ho x Wo");
  p.dy = dy; p.ldy = ldy; p.N = a.N; p.K = p.ntap * p.Ctot; p.Kpad = (p.K + WK - 1) / WK * WK;
  p.M = (int)M; p.chunks = (int)((M + WM - 1) / WM);
  p.S = wgrad_splits(M, a.N, p.K, !quad && wgrad_row3x3_shape(a));
  const size_t slab_floats = (size_t)phases * p.S * p.N * p.Kpad, need = (slab_floats + (quad ? (size_t)4 * p.N * p.K : 0)) * sizeof(float);
  M2H_REQUIRE(a.workspace != nullptr && a.workspace_bytes >= need, "conv_wgrad: workspace too small (need %zu bytes)", need);
  p.ws = static_cast<float*>(a.workspace);
  p.dwp = quad ? p.ws + slab_floats : nullptr;
  p.dw = dw;
  p.gate = gate; p.gate_slope = gate_slope; p.torch_ci = torch_ci;
  p.dy2 = dy2; p.w2p = w2p;
  int bng, kt;
  wgrad_cfg(a.N, p.K, bng, kt, p.ktiles, M);
  p.ntiles = (a.N + bng - 1) / bng;
  const long nblk = (long)(p.S >= 8 ? (p.S + 7) / 8 * 8 : p.S) * p.ntiles * p.ktiles;
  M2H_REQUIRE(nblk < 0x7fffffffL, "conv_wgrad: grid too large");
  const dim3 grid((unsigned)nblk, (unsigned)phases), blk(256);
  // 3x3 / stride 1 / pad 1 over 32-channel, 32-pixel-wide images (AcousticMem): one image row per reduction chunk
  const bool row3x3 = !quad && g_wgrad_row3x3 >= 0 && a.nth == 3 && a.ntw == 3 && a.stride == 1 && a.mulh == 1 && a.mulw == 1 && a.offh == -1 &&
                      a.offw == -1 && a.C0 == 32 && a.C1 == 0 && a.Wq == 32 && a.Wi == 32 && a.Hq == a.Hi && p.direct && a.N <= 32 &&
                      a.N % 4 == 0 && ldy % 4 == 0 && p.ntiles * p.ktiles == 1;
  M2H_REQUIRE(gate == nullptr || row3x3, "conv_wgrad: the activation gate is built into the image-row 3x3 kernel only (3x3 / stride 1 / pad 1, 32 channels, 32-pixel rows)");
  M2H_REQUIRE(dy2 == nullptr || (row3x3 && tl_math_mode == 1 && a.N == 32 && gate != nullptr),
              "conv_wgrad: the fused input gradient is built into the bf16x3 image-row 3x3 kernel only (N = 32, with the activation gate)");
  if (row3x3) {
    p.chunks = a.B * a.Hq;   // image rows
    if (p.S > p.chunks) p.S = p.chunks;
    if (dy2 != nullptr) {
      M2H_LAUNCH(wgrad3x3_row_dgrad_bf16x3_kernel, dim3((unsigned)p.S), blk, 0, st, p);
    } else if (tl_math_mode == 1) {           // the calling thread computes in bf16x3 (update_sep with sep_update_math, the far-target leg)
      if (a.N <= 16) M2H_LAUNCH((wgrad3x3_row_bf16x3_kernel<16>), dim3((unsigned)p.S), blk, 0, st, p);
      else M2H_LAUNCH((wgrad3x3_row_bf16x3_kernel<32>), dim3((unsigned)p.S), blk, 0, st, p);
    } else if (a.N <= 16) M2H_LAUNCH((wgrad3x3_row_kernel<16>), dim3((unsigned)p.S), blk, 0, st, p);
    else M2H_LAUNCH((wgrad3x3_row_kernel<32>), dim3((unsigned)p.S), blk, 0, st, p);
  } else if (bng == 128) M2H_LAUNCH((wgrad_kernel<128, 1, 2>), grid, blk, 0, st, p);
  else if (bng == 64 && kt == 2) M2H_LAUNCH((wgrad_kernel<64, 2, 1>), grid, blk, 0, st, p);
  else if (bng == 64) M2H_LAUNCH((wgrad_kernel<64, 1, 2>), grid, blk, 0, st, p);
  else if (kt == 1) M2H_LAUNCH((wgrad_kernel<32, 1, 2>), grid, blk, 0, st, p);
  else if (kt == 2) M2H_LAUNCH((wgrad_kernel<32, 2, 1>), grid, blk, 0, st, p);
  else M2H_LAUNCH((wgrad_kernel<32, 3, 1>), grid, blk, 0, st, p);
  int rc = launch_status("conv_wgrad");
  if (rc) return rc;
  if (quad) {   // split sum + scatter to the torch layout in one launch
    const long gu = (long)p.N * ((p.Ctot + 15) / 16);
    M2H_REQUIRE(gu < 0x7fffffffL, "convT_wgrad: unpack grid too large");
    if (p.S >= 16) M2H_LAUNCH(convT_wgrad_reduce_unpack_kernel<true>, dim3((unsigned)gu), dim3(1024), 0, st, p);
    else M2H_LAUNCH(convT_wgrad_reduce_unpack_kernel<false>, dim3((unsigned)gu), dim3(256), 0, st, p);
    return launch_status("convT_wgrad reduce + unpack");
  }
  if (torch_ci > 0) {   // split sum + re-layout to [N][Ci][KH][KW] in one launch
    const long gt = (long)p.N * ((torch_ci + 15) / 16);
    M2H_REQUIRE(gt < 0x7fffffffL, "conv_wgrad: reduce grid too large");
    const int tgroups = (p.ntap + 15) / 16;
    const unsigned gy = (unsigned)(gt >= 512 || tgroups == 1 ? 1 : (tgroups < 8 ? tgroups : 8));   // enough blocks for the chip before the taps are spread
    if (p.S >= 16) M2H_LAUNCH(conv_wgrad_reduce_torch_kernel<true>, dim3((unsigned)gt, gy), dim3(1024), 0, st, p);
    else M2H_LAUNCH(conv_wgrad_reduce_torch_kernel<false>, dim3((unsigned)gt, gy), dim3(256), 0, st, p);
    return launch_status("conv_wgrad reduce (torch layout)");
  }
  if (p.S >= 16) {
    const long g = (long)p.N * ((p.K + 63) / 64);
    M2H_REQUIRE(g < 0x7fffffffL, "conv_wgrad: reduce grid too large");
    M2H_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)g, (unsigned)phases), dim3(256), 0, st, p);
  } else {
    size_t g = ((size_t)p.N * p.K + 255) / 256;
    if (g > 4096) g = 4096;
    M2H_LAUNCH(wgrad_reduce_small_kernel, dim3((unsigned)g, (unsigned)phases), dim3(256), 0, st, p);
  }
  return launch_status("conv_wgrad reduce");
}

// w [Co][Ci][KH][KW] -> per phase (ph,pw) of the stride: wp[phase][ci][th][tw][co] = w[co][ci][kh0(ph)+s*th][kw0(pw)+s*tw],
// kh0(ph) = (ph + pad) % s.  Requires KH % s == 0, KW % s == 0.  The matching launch: N = Ci, taps (KH/s, KW/s), mul = -1,
// off = (ph + pad - kh0)/s, stride 1, output step s, phase (ph,pw).
__global__ void pack_dgrad_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int KH, int KW, int s, int pad) {
  const int th_n = KH / s, tw_n = KW / s;
  const size_t per_phase = (size_t)Ci * th_n * tw_n * Co;
  const size_t total = per_phase * s * s;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % Co);
    size_t r = i / Co;
    const int tw = (int)(r % tw_n);
    r /= tw_n;
    const int th = (int)(r % th_n);
    r /= th_n;
    const int ci = (int)(r % Ci);
    const int phase = (int)(r / Ci);
    const int ph = phase / s, pw = phase % s;
    const int kh = (ph + pad) % s + s * th, kw = (pw + pad) % s + s * tw;
    wp[i] = w[(((size_t)co * Ci + ci) * KH + kh) * KW + kw];
  }
}

// inverse of pack_convT_weight for gradients: dw[ci][co][kh][kw] = dwp[phase][co][th][tw][ci], kh = (ph ? 2 : 1) + th*(ph ? -2 : 2)
__global__ void unpack_convT_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Ci, int Co) {
  const size_t total = (size_t)16 * Co * Ci;
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
    dw[(((size_t)ci * Co + co) * 4 + kh) * 4 + kw] = dwp[i];
  }
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float slope, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = y[i] > 0.f ? dy[i] : dy[i] * slope;
}

// db[n] = sum_m dy[m][n].  Two ordered stages (deterministic): grid (column blocks of 64, row splits) -> part[split][n],
// then one thread per column sums the splits.  Lanes walk columns (coalesced 256-byte rows), the 4 waves stride the rows.
// GATE: m2h_act_bwd_bias -- the element is first passed through the activation's backward (y > 0 ? dy : dy * slope) and written to `out`: the
// same partition and summation order, so db has the bits of m2h_act_bwd followed by m2h_bias_grad, from one pass over dy instead of two.
// NW = waves per block = row lanes: 16 for the one-stage form (a few hundred rows on N / 64 blocks: with 4 waves a wave walked 70 of the update
// batch's 280 rows, nine dependent batches of loads -- 24-37 us for the encoders' 512-wide Linear layers on 8 blocks)
template <bool GATE, int NW = 4>
__global__ __launch_bounds__(64 * NW) void bias_grad_partial_kernel(const float* __restrict__ dy, float* __restrict__ part, int M, int N, int rows_per_split,
                                                                const float* __restrict__ y = nullptr, float slope = 1.f, float* __restrict__ out = nullptr) {
  __shared__ float sh[NW][64];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const int m0 = blockIdx.y * rows_per_split;
  const int m1 = min(M, m0 + rows_per_split);
  float s = 0.f;
  if (n < N) {
    // U rows' loads in flight before the first add (the sum keeps its order, row by row: same bits as the one-load-at-a-time loop, which
    // was a chain of dependent memory round trips -- 47-51 us for the update batch's 280 rows x 512 columns on 8 blocks)
    constexpr int U = 8;
    for (int m = m0 + w; m < m1; m += NW * U) {
      float v[U], g[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int mm = m + NW * u;
        v[u] = mm < m1 ? dy[(size_t)mm * N + n] : 0.f;
        if constexpr (GATE) g[u] = mm < m1 ? y[(size_t)mm * N + n] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int mm = m + NW * u;
        if (mm < m1) {
          float x = v[u];
          if constexpr (GATE) {
            x = g[u] > 0.f ? x : x * slope;
            out[(size_t)mm * N + n] = x;
          }
          s += x;
        }
      }
    }
  }
  sh[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && n < N) {
    float r = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
#pragma unroll
    for (int k = 4; k < NW; ++k) r += sh[k][threadIdx.x];
    part[(size_t)blockIdx.y * N + n] = r;
  }
}

// The same partial sums for NARROW gradients (N in {1, 2, 4, 8, 16, 32}: the U-Net heads' 2 channels over a million pixels, the encoders'
// 32-channel convs): with lanes walking columns only N of 64 lanes work and a wave's load is an N-float run (the head's bias gradient
// took 82 us for 8 MB).  Here a wave reads 64 consecutive floats = 64 / N whole rows per step (lane l: row l / N, column l % N), four
// steps in flight, and the lanes of one column meet in a fixed xor butterfly; then the waves in order.
template <bool GATE>
__global__ __launch_bounds__(256) void bias_grad_partial_narrow_kernel(const float* __restrict__ dy, float* __restrict__ part, int M, int N,
                                                                       int rows_per_split, const float* __restrict__ y = nullptr, float slope = 1.f,
                                                                       float* __restrict__ out = nullptr) {
  __shared__ float sh[4][32];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int RW = 64 / N;                                   // rows per wave step
  const int m0 = blockIdx.y * rows_per_split;
  const int m1 = min(M, m0 + rows_per_split);
  const size_t end = (size_t)m1 * N;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  size_t i = ((size_t)m0 + (size_t)w * RW) * N + lane;      // this lane's element; a block step is 4 waves x 64 floats
  auto gated = [&](size_t j, float v) {
    if constexpr (GATE) {
      v = y[j] > 0.f ? v : v * slope;
      out[j] = v;
    }
    return v;
  };
  for (; i + 3 * 256 < end; i += 4 * 256) {
    float a = dy[i], b = dy[i + 256], c = dy[i + 512], d = dy[i + 768];
    a = gated(i, a); b = gated(i + 256, b); c = gated(i + 512, c); d = gated(i + 768, d);
    s0 += a; s1 += b; s2 += c; s3 += d;
  }
  for (; i < end; i += 256) s0 += gated(i, dy[i]);
  float s = (s0 + s1) + (s2 + s3);
  for (int o = 32; o >= N; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane < N) sh[w][lane] = s;
  __syncthreads();
  if (w == 0 && lane < N) part[(size_t)blockIdx.y * N + lane] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}

// one wave per column: lanes sum the splits strided by 64, then a fixed butterfly (deterministic)
__global__ __launch_bounds__(256) void bias_grad_final_kernel(const float* __restrict__ part, float* __restrict__ db, int N, int splits) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float s = 0.f;
  for (int z = threadIdx.x & 63; z < splits; z += 64) s += part[(size_t)z * N + n];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) db[n] = s;
}

static int bias_grad_splits(int M, int N) {
  const int colblocks = (N + 63) / 64;
  int splits = (1024 + colblocks - 1) / colblocks;  // ~4 blocks per CU
  if (splits > (M + 63) / 64) splits = (M + 63) / 64;  // at least 64 rows per split
  if (splits < 1) splits = 1;
  if (M <= 1024) splits = 1;   // a few hundred rows (the update batch's Linear / GRU layers): one stage, straight into db -- no second launch
  return splits;
}

template <bool GATE>
static int bias_grad_launch(const float* dy, const float* y, float slope, float* out, float* db, int M, int N, float* workspace, hipStream_t st) {
  const int splits = bias_grad_splits(M, N);
  const int rps = (M + splits - 1) / splits;
  if (splits == 1) {   // the one split's "partial" IS the column sum
    if (M > 64) M2H_LAUNCH((bias_grad_partial_kernel<GATE, 16>), dim3((N + 63) / 64, 1), dim3(1024), 0, st, dy, db, M, N, rps, y, slope, out);
    else M2H_LAUNCH((bias_grad_partial_kernel<GATE, 4>), dim3((N + 63) / 64, 1), dim3(256), 0, st, dy, db, M, N, rps, y, slope, out);
    return launch_status(GATE ? "act_bwd_bias" : "bias_grad");
  }
  if (N <= 32 && 64 % N == 0) {   // narrow: splits of whole wave steps (64 / N rows); trailing splits may be empty (their partial is 0)
    const int rw = 64 / N, rps_n = (rps + rw - 1) / rw * rw;
    M2H_LAUNCH(bias_grad_partial_narrow_kernel<GATE>, dim3(1, splits), dim3(256), 0, st, dy, workspace, M, N, rps_n, y, slope, out);
  } else
    M2H_LAUNCH((bias_grad_partial_kernel<GATE, 4>), dim3((N + 63) / 64, splits), dim3(256), 0, st, dy, workspace, M, N, rps, y, slope, out);
  M2H_LAUNCH(bias_grad_final_kernel, dim3((N + 3) / 4), dim3(256), 0, st, workspace, db, N, splits);
  return launch_status(GATE ? "act_bwd_bias" : "bias_grad");
}

}  // namespace m2h

using namespace m2h;

extern "C" {

size_t m2h_conv_wgrad_workspace_bytes(const m2h_conv_args* args) { return args ? conv_wgrad_workspace_bytes(*args) : 0; }

int m2h_conv_wgrad_f32(const m2h_conv_args* args, const float* dy, int ldy, float* dw, m2h_stream stream) {
  M2H_REQUIRE(args != nullptr, "conv_wgrad: null args");
  return conv_wgrad_f32(*args, dy, ldy, dw, as_stream(stream));
}

int m2h_conv_wgrad_gated_f32(const m2h_conv_args* args, const float* dy, int ldy, const float* y, float slope, float* dw, m2h_stream stream) {
  M2H_REQUIRE(args != nullptr && y != nullptr, "conv_wgrad_gated: null pointer");
  return conv_wgrad_f32(*args, dy, ldy, dw, as_stream(stream), false, y, slope);
}

int m2h_conv_wgrad_dgrad_fused_supported(const m2h_conv_args* args) {
  if (args == nullptr) return 0;
  const m2h_conv_args& a = *args;
  int bng, kt, ktiles;
  wgrad_cfg(a.N, a.nth * a.ntw * (a.C0 + a.C1), bng, kt, ktiles, (long)a.B * a.Hq * a.Wq);
  return (tl_math_mode == 1 && g_wgrad_row3x3 >= 0 && wgrad_row3x3_shape(a) && a.N == 32 && ((a.N + bng - 1) / bng) * ktiles == 1) ? 1 : 0;
}

int m2h_conv_wgrad_dgrad_fused_f32(const m2h_conv_args* args, const float* dy2, const float* w2_packed, const float* y, float slope, float* dw, int Ci,
                                   m2h_stream stream) {
  M2H_REQUIRE(args != nullptr && dy2 != nullptr && w2_packed != nullptr && y != nullptr, "conv_wgrad_dgrad_fused: null pointer");
  M2H_REQUIRE(m2h_conv_wgrad_dgrad_fused_supported(args), "conv_wgrad_dgrad_fused: needs the bf16x3 arithmetic and the image-row shape (3x3 / 1 / 1, 32 -> 32 "
              "channels over 32-pixel rows); use m2h_conv_igemm_f32 (input gradient) + m2h_conv_wgrad_torch_f32 otherwise");
  return conv_wgrad_f32(*args, nullptr, args->N, dw, as_stream(stream), false, y, slope, Ci, dy2, w2_packed);
}

int m2h_conv_wgrad_torch_f32(const m2h_conv_args* args, const float* dy, int ldy, const float* y, float slope, float* dw, int Ci, m2h_stream stream) {
  M2H_REQUIRE(args != nullptr && Ci > 0, "conv_wgrad_torch: null args / Ci <= 0");
  return conv_wgrad_f32(*args, dy, ldy, dw, as_stream(stream), false, y, slope, Ci);
}

size_t m2h_convT_wgrad_workspace_bytes(const m2h_conv_args* args) {   // four phases of slabs + the packed per-phase gradients
  return args ? 4 * conv_wgrad_workspace_bytes(*args) + (size_t)4 * args->N * args->nth * args->ntw * (args->C0 + args->C1) * sizeof(float) : 0;
}

int m2h_convT_wgrad_f32(const m2h_conv_args* args, const float* dy, int ldy, float* dw, m2h_stream stream) {
  M2H_REQUIRE(args != nullptr, "convT_wgrad: null args");
  return conv_wgrad_f32(*args, dy, ldy, dw, as_stream(stream), true);
}

int m2h_pack_dgrad_weight(const float* w, float* wp, int Co, int Ci, int KH, int KW, int stride, int pad, m2h_stream stream) {
  M2H_REQUIRE(w && wp && Co > 0 && Ci > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, "pack_dgrad_weight: bad arguments");
  M2H_REQUIRE(KH % stride == 0 && KW % stride == 0, "pack_dgrad_weight: kernel size must be a multiple of the stride");
  const size_t total = (size_t)Co * Ci * KH * KW;
  size_t g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  M2H_LAUNCH(pack_dgrad_weight_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), w, wp, Co, Ci, KH, KW, stride, pad);
  return launch_status("pack_dgrad_weight");
}

int m2h_unpack_convT_wgrad(const float* dwp, float* dw, int Ci, int Co, m2h_stream stream) {
  M2H_REQUIRE(dwp && dw && Ci > 0 && Co > 0, "unpack_convT_wgrad: bad arguments");
  size_t g = ((size_t)16 * Co * Ci + 255) / 256;
  if (g > 2048) g = 2048;
  M2H_LAUNCH(unpack_convT_wgrad_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), dwp, dw, Ci, Co);
  return launch_status("unpack_convT_wgrad");
}

int m2h_act_bwd(const float* dy, const float* y, float slope, float* out, size_t n, m2h_stream stream) {
  M2H_REQUIRE(dy && y && out && n > 0, "act_bwd: bad arguments");
  size_t g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  M2H_LAUNCH(act_bwd_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), dy, y, slope, out, n);
  return launch_status("act_bwd");
}

size_t m2h_bias_grad_workspace_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return (size_t)bias_grad_splits(M, N) * N * sizeof(float);
}

int m2h_bias_grad(const float* dy, float* db, int M, int N, float* workspace, m2h_stream stream) {
  M2H_REQUIRE(dy && db && workspace && M > 0 && N > 0, "bias_grad: bad arguments");
  return bias_grad_launch<false>(dy, nullptr, 1.f, nullptr, db, M, N, workspace, as_stream(stream));
}

int m2h_act_bwd_bias(const float* dy, const float* y, float slope, float* out, float* db, int M, int N, float* workspace, m2h_stream stream) {
  M2H_REQUIRE(dy && y && out && db && workspace && M > 0 && N > 0, "act_bwd_bias: bad arguments");
  M2H_REQUIRE((size_t)M * N < ((size_t)1 << 40), "act_bwd_bias: tensor too large");
  return bias_grad_launch<true>(dy, y, slope, out, db, M, N, workspace, as_stream(stream));
}

}  // extern "C"
