// Small-batch convolution engine (fp32 MFMA): the separator U-Nets and the policy's audio encoders at the ROLLOUT batch (14 envs:
// 14 ... 3 584 GEMM rows per layer against 0.1 ... 17 MB of weights; separator_cnn.py:46-52,128-135 at ppo_trainer.py:295-373).
//
// At this size a layer is 0.2 - 0.5 GFLOP (1.5 - 3 us at the fp32 matrix peak if all 256 CUs take part) and what the tiled
// engines pay for is (i) the im2col duplication of the activations (every input pixel fetched once per tap by every n-tile),
// (ii) too few workgroups (64 - 112) unless K is split over blocks, which then needs a reduce launch, and (iii) a serial
// k-tile chain per block.  Here
//   * a workgroup owns (a tile of <= 64 output pixels per phase) x (16 * NWN output channels) x (CG input channels, ALL taps):
//     the K dimension is split over workgroups by input-channel group, so the chip is filled by (tiles x n-tiles x groups);
//   * the tile's raw input patch -- (rows + halo) x (cols + halo) x CG channels -- is staged in LDS ONCE and every tap reads
//     its A fragments from a row / column shift of it (no im2col traffic; the halo is zero padding);
//   * the weights go straight from global memory into the B fragments of v_mfma_f32_16x16x4_f32 (a lane loads 16 bytes of its
//     output channel's row: four consecutive MFMAs' worth), each weight is read by exactly one wave of one workgroup, and the
//     loads of the first sixteen 16-channel chunks are in flight BEFORE the patch is staged (they do not depend on it);
//   * the four sub-pixel phases of a ConvTranspose2d(4, 2, 1) are four wave groups of one workgroup sharing the patch;
//   * partial sums of the channel groups go to S = Ctot / CG slabs and are NOT reduced by a launch of their own: the consumer's
//     patch stager sums the S slabs of each source in slab order (bit-reproducible) and applies the producer's epilogue
//     (folded BatchNorm scale / shift, LeakyReLU / ReLU) on the way into LDS.  A layer is ONE launch, with no epilogue pass and
//     no reduce kernel; `finish` makes a layer apply its own epilogue (one channel group) where a plain tensor must come out:
//     the first stage (class plane, separator_cnn.py:93-99) and the last stage + 1x1 head + de-slice (:134, :163-168).
// Same sums as the other engines up to fp32 association (tests/test_gpu_small.py: layers against torch on the CPU, the runner
// against the oracle and against the tiled engines).
#include "igemm_common.h"

// diagnostic builds only (tools/build_variant.sh NAME conv_small.hip -DM2H_SMALL_DBG=bits; tools/small_phases.sh): phases switched
// off to time the others by elimination -- 1 no patch staging, 2 no MFMA loop, 4 no output phase, 8 no weight loads.  Results are wrong.
#ifndef M2H_SMALL_DBG
#define M2H_SMALL_DBG 0
#endif

namespace m2h {

#if M2H_SMALL_DBG & 16
// bit 16: wall-clock stamps (100 MHz) of thread 0 of every block at the phase boundaries, read back by m2h_debug_small_stamps
__device__ unsigned long long g_small_stamps[4096][8];
#define M2H_STAMP(k) do { if (threadIdx.x == 0) g_small_stamps[blockIdx.x & 4095][k] = wall_clock64(); } while (0)
#else
#define M2H_STAMP(k) do { } while (0)
#endif

struct SmallSrcP {
  const float* p;
  const float* scale;
  const float* shift;
  float slope;
  int C, S;
  long slab;
};

struct SmallP {
  SmallSrcP s[2];
  const float* mix;     // first-stage input (BHWC [B][16*Hi][Wi][2]) instead of s[0]: 32 sliced channels c*16 + band
  const float* masks;   // with mix: bin2mono pre-op log1p(max(0, masks * (exp(mix) - 1)))
  int Ctot, B, Hi, Wi, Ho, Wo;
  int convT, stride, KWp;
  int th0, thn, tw0, twn;
  const float* w;
  int N, K;
  int IB, QR, CG, ccsh, ncg, nnt, row_tiles, rows_total;
  int ih_mul, ih_off;           // first input row of a tile's window: ih0 = q0 * ih_mul + ih_off (may lie above the image)
  int PRwin, pad;               // rows of the window; padding of the conv
  int PR, PC, pitch, Wt;        // patch rows (window clipped to the image) / cols (= Wi) per image, floats per patch pixel, pixels per tile row
  unsigned pc_magic, pr_magic, ncg_magic, nnt_magic, rt_magic, wt_magic, qr_magic, twn_magic;   // ceil(2^32 / d) of PC, PR, ncg, nnt, row_tiles, Wt, QR, twn
  int c0sh, c1sh;               // log2(C / 4) of the two sources (a finishing layer's one channel group covers both)
  float* dst;
  long dst_slab;
  int finish;
  const float* scale;
  const float* shift;
  float slope;
  const float* cls_table;
  const float* cls_val;
  const float* head_w;
  const float* head_b;
};

// exact n / d for the small indices of this file (n * d < 2^32) as one multiply-high: magic = ceil(2^32 / d), made on the host.
// (An integer division by a run-time value is ~40 dependent VALU instructions; the first version of this kernel spent more
// time in the ~20 of them per thread than in its MFMAs.)
__device__ __forceinline__ int fdiv(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }   // magic 0: d == 1

template <int PH, int NWN, int NWK, int MT>
__global__ __launch_bounds__(64 * PH * NWN * NWK) void conv_small_kernel(const SmallP p) {
  extern __shared__ __align__(16) float lds[];
  constexpr int NW = PH * NWN * NWK, NT = 64 * NW;
  constexpr int G = NW > 8 ? 4 : 8;          // weight chunks (16 channels of one tap each) per register buffer (128 VGPRs per lane at 16 waves)
  constexpr int U = NW > 8 ? 2 : 4;          // stager items per thread and pass (128 VGPRs per lane at 16 waves)
  constexpr bool EARLY_EPI = NW <= 8;        // epilogue constants loaded with the first slab (registers allow it below 16 waves)
  constexpr int SU = 4;                      // slabs per round trip of the stager
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int L = blockIdx.x;
  const int L1 = fdiv(L, p.ncg_magic), cg = L - L1 * p.ncg;
  const int tile = fdiv(L1, p.nnt_magic), nt = L1 - tile * p.nnt;
  const int it = fdiv(tile, p.rt_magic), rt = tile - it * p.row_tiles;
  const int img0 = it * p.IB, q0 = rt * p.QR;
  const int kp = wave % NWK, nj = (wave / NWK) % NWN, phase = wave / (NWK * NWN);
  const int ph = phase >> 1, pw = phase & 1;

  M2H_STAMP(0);
  // ---- 1. weights of this wave's chunks: the first two register buffers go out now, ahead of the patch
  const int CC = p.CG >> 4;
  const int nchunk = p.thn * p.twn * CC;
  const int c0 = (nchunk * kp) / NWK, c1 = (nchunk * (kp + 1)) / NWK;
  const int n = nt * (16 * NWN) + nj * 16 + i;
  const float* wrow = p.w + ((size_t)phase * p.N + min(n, p.N - 1)) * p.K + (size_t)cg * p.CG + 4 * kq;
  auto wofs = [&](int ch) -> int {           // float offset of chunk ch inside the weight row (wave-uniform)
    const int tapi = ch >> p.ccsh, cc = ch & (CC - 1);
    const int thi = fdiv(tapi, p.twn_magic), twi = tapi - thi * p.twn;
    return ((p.th0 + thi) * p.KWp + p.tw0 + twi) * p.Ctot + cc * 16;
  };
  f32x4 wb[2][G];
  auto wload = [&](f32x4 (&buf)[G], int base) {
#pragma unroll
    for (int j = 0; j < G; ++j)
      if (base + j < c1) buf[j] = *reinterpret_cast<const f32x4*>(wrow + wofs(base + j));
  };
  if (!(M2H_SMALL_DBG & 8)) {
    wload(wb[0], c0);
    wload(wb[1], c0 + G);
  } else {
#pragma unroll
    for (int j = 0; j < G; ++j) wb[0][j] = wb[1][j] = {1.f, 1.f, 1.f, 1.f};
  }

  M2H_STAMP(1);
  // ---- 2. patch: IB images x PR rows x PC cols x CG channels, slab sums + the producer's epilogue, zero outside the image.
  // U items (16-byte pieces) per thread and pass, all their loads issued before the first use; addresses of pieces outside
  // the image are clamped to the tensor's first element and the value replaced by zero (no divergent branches around loads).
  // A block's channel group lies in one source, or (a finishing layer over two sources) covers both: one pass per source.
  // The patch holds IN-IMAGE pixels only: input rows [ih_lo, ih_lo + prs) of the tile's window (clipped to the image) x all Wi
  // columns; what a tap reaches outside the image is masked per lane in the reduction (no zero halo: a whole-image tile of a deep
  // stage would spend half its LDS on padding).
  const int ih0 = q0 * p.ih_mul + p.ih_off;
  const int ih_lo = max(ih0, 0);
  const int prs = min(ih0 + p.PRwin, p.Hi) - ih_lo;      // rows this tile stores (<= p.PR)
  const int npix = p.IB * p.PR * p.PC;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto stage = [&](const SmallSrcP& s, int cl0, int lds_c0, int n4sh) {
    // channels cl0 ... cl0 + 4 * 2^n4sh - 1 of source s -> patch channels lds_c0 ...
    const int n4 = 1 << n4sh;
    const int total = npix << n4sh;
    for (int base = tid; base < total; base += U * NT) {
      f32x4 v[U];
      int dofs[U];
      bool ok[U];
      int cl[U];
      const float* q[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int item = min(base + u * NT, total - 1);
        const int px = item >> n4sh, piece = item & (n4 - 1);
        const int row = fdiv(px, p.pc_magic), pc = px - row * p.PC;
        const int ib = fdiv(row, p.pr_magic), pr = row - ib * p.PR;
        const int b = img0 + ib, ih = ih_lo + pr, iw = pc;
        ok[u] = b < p.B && pr < prs;
        dofs[u] = px * p.pitch + lds_c0 + piece * 4;
        cl[u] = cl0 + piece * 4;
        if (p.mix != nullptr) {
          // separator_cnn.py:73-90: channel c*16 + band of pixel (h, t) <- mix[b][band * Hi + h][t][c], optional bin2mono pre-op
          const int ci = cl[u] >> 4, s0 = cl[u] & 15;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const size_t off = ok[u] ? (((size_t)b * 16 * p.Hi + (size_t)(s0 + e) * p.Hi + ih) * p.Wi + iw) * 2 + ci : (size_t)0;
            float x = p.mix[off];
            if (p.masks != nullptr) x = log1pf(fmaxf(p.masks[off] * (expf(x) - 1.f), 0.f));
            v[u][e] = x;
          }
          q[u] = nullptr;
        } else {
          q[u] = s.p + (ok[u] ? ((size_t)(b * p.Hi + ih) * p.Wi + iw) * s.C + cl[u] : (size_t)0);
          v[u] = *reinterpret_cast<const f32x4*>(q[u]);
        }
      }
      if (p.mix == nullptr) {
        // the epilogue constants go out with the first slab's loads (one round trip instead of two)
        f32x4 sc[U], sf[U];
        if (EARLY_EPI && s.scale != nullptr) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            sc[u] = *reinterpret_cast<const f32x4*>(s.scale + cl[u]);
            sf[u] = *reinterpret_cast<const f32x4*>(s.shift + cl[u]);
          }
        }
        // slabs 1 ... S-1 in slab order, SU of them (x U items) per round trip
        int k = 1;
        for (; k + SU <= s.S; k += SU) {
          f32x4 t[SU][U];
#pragma unroll
          for (int j = 0; j < SU; ++j)
#pragma unroll
            for (int u = 0; u < U; ++u) t[j][u] = *reinterpret_cast<const f32x4*>(q[u] + (size_t)(k + j) * s.slab);
#pragma unroll
          for (int j = 0; j < SU; ++j)
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] += t[j][u];
        }
        if (k < s.S) {          // the last 1 ... SU-1 slabs: one more round trip
          f32x4 t[SU - 1][U];
#pragma unroll
          for (int j = 0; j < SU - 1; ++j)
#pragma unroll
            for (int u = 0; u < U; ++u) t[j][u] = *reinterpret_cast<const f32x4*>(q[u] + (size_t)min(k + j, s.S - 1) * s.slab);
#pragma unroll
          for (int j = 0; j < SU - 1; ++j)
#pragma unroll
            for (int u = 0; u < U; ++u)
              if (k + j < s.S) v[u] += t[j][u];
        }
        if (s.scale != nullptr) {
          if (!EARLY_EPI) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
              sc[u] = *reinterpret_cast<const f32x4*>(s.scale + cl[u]);
              sf[u] = *reinterpret_cast<const f32x4*>(s.shift + cl[u]);
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) v[u] = v[u] * sc[u] + sf[u];
        }
        if (s.slope != 1.f) {
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[u][e] = v[u][e] > 0.f ? v[u][e] : v[u][e] * s.slope;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (base + u * NT < total) *reinterpret_cast<f32x4*>(lds + dofs[u]) = ok[u] ? v[u] : zero4;
    }
  };
  if (!(M2H_SMALL_DBG & 1)) {
    const int cb = cg * p.CG;                          // first concatenated channel of this block
    if (p.mix != nullptr || cb + p.CG <= p.s[0].C) {
      stage(p.s[0], cb, 0, p.ccsh + 2);
    } else if (cb >= p.s[0].C) {
      stage(p.s[1], cb - p.s[0].C, 0, p.ccsh + 2);
    } else {                                           // one group over both sources (host: cb == 0, both channel counts powers of two)
      stage(p.s[0], 0, 0, p.c0sh);
      stage(p.s[1], 0, p.s[0].C, p.c1sh);
    }
  }

  M2H_STAMP(2);
  // ---- 3. this lane's GEMM rows: pixel (ib, r, x) of the tile -> patch pixel index at tap (0, 0)
  int pixb[MT], row0[MT], col0[MT];       // patch pixel index of image pixel (0, 0) of the row's image; input position at tap offset (0, 0)
  bool okm[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 16 + i;
    const int t = fdiv(m, p.wt_magic), x = m - t * p.Wt;
    const int ib = fdiv(t, p.qr_magic), r = t - ib * p.QR;
    okm[mt] = ib < p.IB && img0 + ib < p.B && q0 + r < p.rows_total;
    row0[mt] = p.convT ? q0 + r : (q0 + r) * p.stride - p.pad;
    col0[mt] = p.convT ? x : x * p.stride - p.pad;
    pixb[mt] = (ib * p.PR - ih_lo) * p.PC;
  }
  const int sh = p.convT ? (2 * ph - 1) : 1, sw = p.convT ? (2 * pw - 1) : 1;
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  M2H_STAMP(3);
  // ---- 4. the reduction: per chunk one B fragment (registers) against MT A fragments read at the tap's shift of the patch
  auto compute = [&](f32x4 (&buf)[G], int base) {
    // straight-line code over the buffer's G chunks (no branch per chunk: the LDS reads of all of them can be issued ahead of
    // the MFMAs); chunks past this wave's range multiply zero weights against the last chunk's pixels
    int dr[G], dc[G], coff[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int ch = min(base + j, c1 - 1);
      const int tapi = ch >> p.ccsh, cc = ch & (CC - 1);
      const int thi = fdiv(tapi, p.twn_magic), twi = tapi - thi * p.twn;
      dr[j] = p.convT ? thi * sh : p.th0 + thi;
      dc[j] = p.convT ? twi * sw : p.tw0 + twi;
      coff[j] = cc * 16 + 4 * kq;
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const f32x4 bw = base + j < c1 ? buf[j] : zero4;
      f32x4 a[MT];
      bool oka[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int rr = row0[mt] + dr[j], cl = col0[mt] + dc[j];
        oka[mt] = okm[mt] && (unsigned)rr < (unsigned)p.Hi && (unsigned)cl < (unsigned)p.Wi;
        const int ad = oka[mt] ? (pixb[mt] + rr * p.PC + cl) * p.pitch + coff[j] : coff[j];   // outside the image: any valid address, value dropped
        a[mt] = *reinterpret_cast<const f32x4*>(lds + ad);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 av = oka[mt] ? a[mt] : zero4;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bw[e], acc[mt], 0, 0, 0);
      }
    }
  };
  for (int base = c0; base < c1 && !(M2H_SMALL_DBG & 2); base += 2 * G) {
    compute(wb[0], base);
    if (!(M2H_SMALL_DBG & 8)) wload(wb[0], base + 2 * G);
    if (base + G < c1) {
      compute(wb[1], base + G);
      if (!(M2H_SMALL_DBG & 8)) wload(wb[1], base + 3 * G);
    }
  }
  M2H_STAMP(4);
  __syncthreads();   // every wave is done with the patch
  M2H_STAMP(5);

  if (M2H_SMALL_DBG & 4) {
    if (acc[0][0] == 12345.678f) p.dst[0] = acc[0][0];   // (keeps the accumulators alive)
    return;
  }
  // ---- 5. the K parts of a (phase, n-subtile) meet through LDS in wave order; store
  float* R = lds;    // [wave][mt][16][17]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) R[((wave * MT + mt) * 16 + kq * 4 + e) * 17 + i] = acc[mt][e];   // D[row kq*4 + e][column i]
  __syncthreads();
  constexpr int GROUPS = PH * NWN * MT;
  const int NB = 16 * NWN;                          // columns of this block
  float* Y = lds + NW * MT * 16 * 17;               // finish == 2: activated tile [PH * MT * 16 pixels][NB + 1]
  for (int o = tid; o < GROUPS * 256; o += NT) {
    const int g = o >> 8, r16 = (o >> 4) & 15, c16 = o & 15;
    const int mt = g % MT, nj2 = (g / MT) % NWN, ph2 = g / (MT * NWN);
    float x = 0.f;
#pragma unroll
    for (int k = 0; k < NWK; ++k) x += R[((((ph2 * NWN + nj2) * NWK + k) * MT + mt) * 16 + r16) * 17 + c16];
    const int m = mt * 16 + r16;
    const int t = fdiv(m, p.wt_magic), xx = m - t * p.Wt;
    const int ib = fdiv(t, p.qr_magic), r = t - ib * p.QR;
    const int b = img0 + ib;
    const bool ok = ib < p.IB && b < p.B && q0 + r < p.rows_total;
    const int n2 = nt * NB + nj2 * 16 + c16;
    const int oy = p.convT ? 2 * (q0 + r) + (ph2 >> 1) : q0 + r, ox = p.convT ? 2 * xx + (ph2 & 1) : xx;
    if (p.finish == 0) {
      if (ok && n2 < p.N) p.dst[(size_t)cg * p.dst_slab + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * p.N + n2] = x;
      continue;
    }
    if (ok && n2 < p.N) {
      if (p.cls_table != nullptr) {
        const int ch = (oy == 0) ? 0 : ((oy == p.Ho - 1) ? 2 : 1), cw = (ox == 0) ? 0 : ((ox == p.Wo - 1) ? 2 : 1);
        x += p.cls_val[b] * p.cls_table[(size_t)(ch * 3 + cw) * p.N + n2];
      }
      const float sc = p.scale != nullptr ? p.scale[n2] : 1.f, sf = p.shift != nullptr ? p.shift[n2] : 0.f;
      x = x * sc + sf;
      x = x > 0.f ? x : x * p.slope;
    }
    if (p.finish == 1) {
      if (ok && n2 < p.N) p.dst[((size_t)(b * p.Ho + oy) * p.Wo + ox) * p.N + n2] = x;
    } else {
      Y[((ph2 * MT + mt) * 16 + r16) * (NB + 1) + nj2 * 16 + c16] = x;
    }
  }
  M2H_STAMP(6);
  if (p.finish == 2) {
    // 1x1 head (separator_cnn.py:134) over the block's NB = N channels, bias, de-sliced store (:163-168): n = c*16 + band
    __syncthreads();
    const size_t plane = (size_t)p.Ho * p.Wo;
    const int Cc = p.N >> 4;
    for (int o = tid; o < PH * MT * 16 * NB; o += NT) {
      const int n2 = o & (NB - 1), pxl = o / NB;                // pxl = (ph2 * MT + mt) * 16 + r16  (NB: 16, 32 or 64)
      const int r16 = pxl & 15, mt = (pxl >> 4) % MT, ph2 = (pxl >> 4) / MT;
      const int m = mt * 16 + r16;
      const int t = fdiv(m, p.wt_magic), xx = m - t * p.Wt;
      const int ib = fdiv(t, p.qr_magic), r = t - ib * p.QR;
      const int b = img0 + ib;
      if (!(ib < p.IB && b < p.B && q0 + r < p.rows_total) || n2 >= p.N) continue;
      const float* y = Y + pxl * (NB + 1);
      const float* hw = p.head_w + (size_t)n2 * p.N;
      float v = 0.f;
      for (int k = 0; k < p.N; ++k) v += hw[k] * y[k];
      v += p.head_b[n2];
      const int oy = p.convT ? 2 * (q0 + r) + (ph2 >> 1) : q0 + r, ox = p.convT ? 2 * xx + (ph2 & 1) : xx;
      p.dst[((size_t)b * 16 * plane + (size_t)(n2 & 15) * plane + (size_t)oy * p.Wo + ox) * Cc + (n2 >> 4)] = v;
    }
  }
  M2H_STAMP(7);
}

namespace {
template <int PH, int NWN, int NWK, int MT>
int launch_small(const SmallP& p, unsigned blocks, size_t lds_bytes, hipStream_t st) {
  auto k = conv_small_kernel<PH, NWN, NWK, MT>;
  // dynamic LDS above 64 KB needs the attribute once per instantiation AND DEVICE (hipFuncSetAttribute acts on the current device's
  // copy of the function); the flags are process-wide and only ever go 0 -> 1 (a racing second set is harmless)
  static bool granted[64] = {};
  if (lds_bytes > 65536) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev < 0 || !granted[dev]) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return fail((int)e, "conv_small: hipFuncSetAttribute: %s", hipGetErrorString(e));
      if (dev >= 0) granted[dev] = true;
    }
  }
  M2H_LAUNCH(k, dim3(blocks), dim3(64 * PH * NWN * NWK), lds_bytes, st, p);
  return launch_status("conv_small");
}
}  // namespace

int conv_small_fwd(const m2h_small_conv_args& a, hipStream_t st) {
  M2H_REQUIRE(a.wp != nullptr && a.dst != nullptr, "conv_small: null pointer");
  M2H_REQUIRE(a.B > 0 && a.Hi > 0 && a.Wi > 0 && a.N > 0 && a.N % 16 == 0, "conv_small: bad sizes (B %d, %d x %d, N %d)", a.B, a.Hi, a.Wi, a.N);
  SmallP p = {};
  const bool first = a.mix != nullptr;
  for (int k = 0; k < 2; ++k) {
    const m2h_small_src& s = a.src[k];
    p.s[k] = {s.p, s.scale, s.shift, s.slope, s.C, s.S < 1 ? 1 : s.S, (long)s.slab};
    M2H_REQUIRE((s.scale == nullptr) == (s.shift == nullptr), "conv_small: source %d: scale and shift come together", k);
  }
  if (first) {
    p.s[0].C = 32;
    p.s[1].C = 0;
  }
  M2H_REQUIRE(first || (p.s[0].p != nullptr && p.s[0].C > 0), "conv_small: no source");
  M2H_REQUIRE(p.s[1].C == 0 || p.s[1].p != nullptr, "conv_small: null second source");
  p.mix = a.mix;
  p.masks = a.masks;
  p.Ctot = p.s[0].C + p.s[1].C;
  p.B = a.B; p.Hi = a.Hi; p.Wi = a.Wi;
  p.convT = a.conv_transpose ? 1 : 0;
  const int CG = a.channels_per_block, IB = a.images_per_tile, QR = a.rows_per_tile, NWN = a.cols_per_block / 16, NWK = a.k_waves;
  const bool both = p.s[1].C > 0 && CG == p.Ctot;     // one channel group over both sources
  auto pow2 = [](int v) { return v >= 4 && (v & (v - 1)) == 0; };
  M2H_REQUIRE(CG >= 16 && (CG & (CG - 1)) == 0 && p.Ctot % CG == 0 && (both ? (pow2(p.s[0].C) && pow2(p.s[1].C)) : p.s[0].C % CG == 0),
              "conv_small: channels_per_block %d must be a power of two >= 16 dividing both sources (%d + %d) or covering both", CG, p.s[0].C, p.s[1].C);
  p.c0sh = 0;
  while ((4 << p.c0sh) < p.s[0].C) ++p.c0sh;
  p.c1sh = 0;
  while ((4 << p.c1sh) < p.s[1].C) ++p.c1sh;
  M2H_REQUIRE(IB >= 1 && QR >= 1 && a.cols_per_block % 16 == 0 && NWN >= 1 && NWK >= 1, "conv_small: bad tiling");
  int rows;   // rows the tiles walk: output rows (conv) or input rows (transposed conv)
  if (p.convT) {
    p.Ho = 2 * a.Hi; p.Wo = 2 * a.Wi; p.stride = 1; p.KWp = 2;
    p.th0 = 0; p.tw0 = 0; p.thn = a.Hi == 1 ? 1 : 2; p.twn = a.Wi == 1 ? 1 : 2;     // a 1-row input: the second tap of every phase is padding
    p.K = 4 * p.Ctot;
    rows = a.Hi;
    p.Wt = a.Wi;
    p.ih_mul = 1; p.ih_off = -1; p.pad = 0;
    p.PRwin = QR + 2; p.PC = a.Wi;
  } else {
    M2H_REQUIRE(a.KH > 0 && a.KW > 0 && a.stride > 0 && a.pad >= 0, "conv_small: bad kernel geometry");
    p.Ho = (a.Hi + 2 * a.pad - a.KH) / a.stride + 1;
    p.Wo = (a.Wi + 2 * a.pad - a.KW) / a.stride + 1;
    M2H_REQUIRE(p.Ho > 0 && p.Wo > 0, "conv_small: empty output");
    p.stride = a.stride; p.KWp = a.KW;
    // tap window: kernel rows / columns that lie in the padding for EVERY output pixel are not walked (exact zeros)
    p.th0 = 0; p.thn = a.KH;
    while (p.thn > 1 && -a.pad + p.thn - 1 >= a.Hi) --p.thn;                                   // below the image even for the first output row
    while (p.thn > 1 && (p.Ho - 1) * a.stride - a.pad + p.th0 < 0) { ++p.th0; --p.thn; }      // above it even for the last one
    p.tw0 = 0; p.twn = a.KW;
    while (p.twn > 1 && -a.pad + p.twn - 1 >= a.Wi) --p.twn;
    while (p.twn > 1 && (p.Wo - 1) * a.stride - a.pad + p.tw0 < 0) { ++p.tw0; --p.twn; }
    p.K = a.KH * a.KW * p.Ctot;
    rows = p.Ho;
    p.Wt = p.Wo;
    p.ih_mul = a.stride; p.ih_off = -a.pad + p.th0; p.pad = a.pad;
    p.PRwin = (QR - 1) * a.stride + p.thn; p.PC = a.Wi;
  }
  p.PR = p.PRwin < a.Hi ? p.PRwin : a.Hi;
  p.rows_total = rows;
  p.w = a.wp; p.N = a.N;
  p.IB = IB; p.QR = QR; p.CG = CG; p.ncg = p.Ctot / CG;
  p.ccsh = 0;
  while ((16 << p.ccsh) < CG) ++p.ccsh;
  p.nnt = (a.N + a.cols_per_block - 1) / a.cols_per_block;
  p.row_tiles = (rows + QR - 1) / QR;
  p.pitch = CG + 4;
  auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned long long)d - 1) / (unsigned long long)d); };
  p.pc_magic = magic(p.PC); p.pr_magic = magic(p.PR); p.ncg_magic = magic(p.ncg); p.nnt_magic = magic(p.nnt); p.rt_magic = magic(p.row_tiles);
  p.wt_magic = magic(p.Wt); p.qr_magic = magic(QR); p.twn_magic = magic(p.twn);
  const int PH = p.convT ? 4 : 1;
  const int MTn = (IB * QR * p.Wt + 15) / 16;
  M2H_REQUIRE(MTn <= 4, "conv_small: a tile of %d x %d x %d pixels exceeds 64 GEMM rows", IB, QR, p.Wt);
  const int MT = MTn <= 1 ? 1 : (MTn <= 2 ? 2 : 4);
  p.dst = a.dst; p.dst_slab = (long)a.dst_slab;
  p.finish = a.finish;
  p.scale = a.scale; p.shift = a.shift; p.slope = a.slope; p.cls_table = a.cls_table; p.cls_val = a.cls_val;
  p.head_w = a.head_w; p.head_b = a.head_b;
  M2H_REQUIRE(a.finish >= 0 && a.finish <= 2, "conv_small: finish must be 0, 1 or 2");
  M2H_REQUIRE(a.finish == 0 || p.ncg == 1, "conv_small: a finishing layer holds all input channels in one block (channels_per_block = %d)", p.Ctot);
  M2H_REQUIRE(a.finish == 0 || (a.scale == nullptr) == (a.shift == nullptr), "conv_small: scale and shift come together");
  M2H_REQUIRE((a.cls_table == nullptr) == (a.cls_val == nullptr), "conv_small: class table / value mismatch");
  M2H_REQUIRE(a.finish != 2 || (a.head_w != nullptr && a.head_b != nullptr && a.cols_per_block == a.N), "conv_small: the head needs all %d channels in one block", a.N);
  const int NW = PH * NWN * NWK;
  M2H_REQUIRE(NW <= 16, "conv_small: %d waves per block", NW);
  const size_t patch = (size_t)IB * p.PR * p.PC * p.pitch * 4;
  size_t red = (size_t)NW * MT * 16 * 17 * 4;
  if (a.finish == 2) red += (size_t)PH * MT * 16 * (16 * NWN + 1) * 4;
  const size_t lds_bytes = patch > red ? patch : red;
  M2H_REQUIRE(lds_bytes <= 160 * 1024, "conv_small: %zu bytes of LDS (patch %d x %d x %d pixels x %d channels)", lds_bytes, IB, p.PR, p.PC, CG);
  const long tiles = (long)((a.B + IB - 1) / IB) * p.row_tiles;
  const long blocks = tiles * p.nnt * p.ncg;
  M2H_REQUIRE(blocks > 0 && blocks < (1L << 30), "conv_small: grid");
#define M2H_SMALL_CASE(PH_, NWN_, NWK_, MT_) \
  if (PH == PH_ && NWN == NWN_ && NWK == NWK_ && MT == MT_) return launch_small<PH_, NWN_, NWK_, MT_>(p, (unsigned)blocks, lds_bytes, st);
#define M2H_SMALL_MT(PH_, NWN_, NWK_) M2H_SMALL_CASE(PH_, NWN_, NWK_, 1) M2H_SMALL_CASE(PH_, NWN_, NWK_, 2) M2H_SMALL_CASE(PH_, NWN_, NWK_, 4)
  M2H_SMALL_MT(1, 1, 4) M2H_SMALL_MT(1, 2, 2) M2H_SMALL_MT(1, 4, 1) M2H_SMALL_MT(1, 1, 8) M2H_SMALL_MT(1, 2, 4) M2H_SMALL_MT(1, 4, 2)
  M2H_SMALL_MT(1, 1, 16) M2H_SMALL_MT(1, 2, 8) M2H_SMALL_MT(1, 4, 4)
  M2H_SMALL_MT(4, 1, 1) M2H_SMALL_MT(4, 1, 2) M2H_SMALL_MT(4, 2, 1) M2H_SMALL_MT(4, 1, 4) M2H_SMALL_MT(4, 2, 2) M2H_SMALL_MT(4, 4, 1)
#undef M2H_SMALL_MT
#undef M2H_SMALL_CASE
  return fail(-1, "conv_small: no kernel for %d phase(s) x %d column groups x %d K parts", PH, NWN, NWK);
}

}  // namespace m2h

#if M2H_SMALL_DBG & 16
extern "C" int m2h_debug_small_stamps(unsigned long long* host_dst /* [4096][8] */) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(m2h::g_small_stamps), sizeof(unsigned long long) * 4096 * 8);
}
#endif

extern "C" int m2h_conv_small_fwd(const m2h_small_conv_args* args, m2h_stream stream) {
  M2H_REQUIRE(args != nullptr, "conv_small: null args");
  return m2h::conv_small_fwd(*args, m2h::as_stream(stream));
}
