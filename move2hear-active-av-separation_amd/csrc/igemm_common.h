// Declarations shared by the implicit-GEMM translation units (conv_igemm.hip: the register-staged engine and its dispatch;
// conv_dma.hip: the LDS-DMA engine for split32 operands): launch parameters, row decode and the fused epilogue.
#pragma once
#include <type_traits>

#include "m2h_internal.h"

namespace m2h {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct IGemmP {
  const float* src0;
  const float* src1;
  int C0, C1, Ctot;
  int B, Hi, Wi;
  int Hq, Wq;
  int stride;
  int ntw, ntap;
  int mulh, offh, mulw, offw;
  int convT;
  const float* w;
  int N, K;
  const float* scale;
  const float* shift;
  float slope;
  const float* cls_table;
  const float* cls_val;
  float* dst;
  int Ho, Wo, os, ph, pw, ldc, out_mode;
  int M, MT, NT;
  const float* head_w;  // fused 1x1 head (N <= 32, one n-tile): [N][N] weights applied to the activated tile, + head_b, de-sliced store
  const float* head_b;
  int fast_ok; // scalar-decode loader applicable (host check)
  int math;      // arithmetic of this launch: 0 fp32 MFMA, 1 bf16x3 split products (args' M2H_FMT_MATH_* or the calling thread's mode)
  int hi_only;   // M2H_MATH_BF16 (the calling thread's mode): the split32 engines of the benchmark batch multiply the bf16 hi halves only (one product of the three)
  int presplit;  // both operands arrive in the split32 layout
  int dst_split; // epilogue writes dst in the split32 layout (bf16x3 math, NHWC, N % 32 == 0)
  int wq_sh, hq_sh;   // log2 of Wq / Hq when they are powers of two, else -1 (decode_row)
  int pmaj;    // transposed conv: phase is folded into grid x (fastest) instead of grid z
  // tap window of the scalar-decode loader: taps th0..th0+thn-1 x tw0..tw0+twn-1 are walked, the others lie in the zero
  // padding for EVERY output pixel of this launch (tiny images: a 2-row input under a 4x4/s2/p1 conv, a 1-row input under a
  // transposed conv) and are skipped: their products are exact zeros.  Kw = thn * twn * Ctot is the walked reduction length.
  int th0, thn, tw0, twn, Kw;
  int S;       // split-K factor (grid y); S > 1: raw partial sums go to `ws`, the epilogue runs in splitk_epilogue_kernel
  float* ws;   // [phase][S][M][N] fp32 partial slabs (caller-owned workspace)
  // fused L1 loss of the image-row 3x3 kernels (N == 16, NHWC; m2h_conv3x3_l1_nhwc16): l1_gt != nullptr -> the epilogue compares the conv's
  // output with the target plane [B][16 * Ho][Wo] (band n of pixel row q = plane row n * Ho + q), adds |y - g| to the block's partial
  // sum (l1_part[block]) and stores sign(y - g) * l1_inv -- d loss / d y -- in place of y
  const float* l1_gt;
  float* l1_part;
  float l1_inv;
};

// the fused-loss arguments of an image-row launch (conv_igemm_f32's optional last parameter)
struct ConvL1 {
  const float* gt;
  float* partials;   // >= 1024 floats
  float* loss;       // 1 float: inv * sum of the partials, fixed order
  float inv;         // 1 / number of elements
};



typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32;   // k-tile depth (floats)
constexpr int LDK = 36;  // padded LDS row (floats): 144 B, keeps 16-B alignment, conflict-free b128 reads

// Row bookkeeping shared by the main kernel and the split-K epilogue: output pixel offset and class id of GEMM row m.
__device__ __forceinline__ void decode_row(const IGemmP& p, int m, int ph, int pw, int& q, int& rr, int& b, int& out, int& bc) {
  if (p.wq_sh >= 0 && p.hq_sh >= 0) {   // power-of-two pixel grid (every U-Net stage): shifts instead of three integer divisions
    rr = m & (p.Wq - 1);
    const int t = m >> p.wq_sh;
    q = t & (p.Hq - 1);
    b = t >> p.hq_sh;
  } else {
    rr = m % p.Wq;
    const int t = m / p.Wq;
    q = t % p.Hq;
    b = t / p.Hq;
  }
  const int oh = q * p.os + ph, ow = rr * p.os + pw;
  if (p.out_mode == M2H_OUT_NHWC)
    out = (b * p.Ho + oh) * p.Wo + ow;
  else
    out = b * 16 * p.Ho * p.Wo + oh * p.Wo + ow;
  const int ch = (oh == 0) ? 0 : ((oh == p.Ho - 1) ? 2 : 1);
  const int cw = (ow == 0) ? 0 : ((ow == p.Wo - 1) ? 2 : 1);
  bc = b * 16 + ch * 3 + cw;
}

// Coalesced store of an activated NHWC tile (fp32 rows, or the split32 layout with dst_split): the accumulator layout of the MFMAs gives every
// lane single 4-byte words scattered over rows (32-byte runs per row and store instruction), which cost 14-15 us per 256 x 128
// tile (in-kernel stamps, tools/clock_diag_dma.py) -- a quarter of a short-K layer.  Here the tile goes through LDS: each lane
// writes its words into a [rows][BN * 4 + 16 bytes] image (RPASS rows per pass, as many as the scratch holds), then the block
// copies whole rows out with 16 bytes per lane: full 128-byte lines, 4 x fewer store instructions.  Values are those of the
// element-wise path bit for bit.  scratch: SCRATCH bytes of LDS the main loop no longer needs.
template <int BM, int BN, int WM, int WN, int FR, int SCRATCH, typename AccT>
__device__ __forceinline__ void nhwc_tile_store(const IGemmP& p, AccT (&acc)[BM / WM / FR][BN / WN / FR], char* scratch, const int* ri_out,
                                                   const int* ri_bc, int n0, int tid) {
  constexpr int NTH = 64 * WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / FR, FN = TN / FR;
  constexpr int NE = FR == 32 ? 16 : 4;
  constexpr int RP = BN * 4 + 16;                     // row pitch (bytes): 16-byte aligned for the b128 copy-out reads
  constexpr int RMAX = SCRATCH / RP;
  constexpr int RPASS = RMAX >= BM ? BM : (RMAX >= BM / 2 ? BM / 2 : (RMAX >= BM / 4 ? BM / 4 : BM / 8));
  constexpr int NPASS = BM / RPASS;
  constexpr int PIECES = BN / 4;                      // 16-byte pieces per row
  static_assert(RPASS * RP <= SCRATCH && RPASS % 32 == 0 && BN % 32 == 0, "scratch too small for the tile store");
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int col = lane & (FR - 1);
  auto row_of = [&](int e) { return FR == 32 ? (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) : (lane >> 4) * 4 + e; };
  float sc[FN], sh[FN];
  int nn[FN], woff[FN];
#pragma unroll
  for (int ni = 0; ni < FN; ++ni) {
    const int nl = wn * TN + ni * FR + col;           // column inside the tile
    const int n = n0 + nl;
    nn[ni] = n;
    sc[ni] = (p.scale != nullptr && n < p.N) ? p.scale[n] : 1.f;
    sh[ni] = (p.shift != nullptr && n < p.N) ? p.shift[n] : 0.f;
    woff[ni] = (nl & ~31) * 4 + ((nl & 1) ? 64 : 0) + ((nl & 31) >> 1) * 4;   // this lane's word: even n the hi pair, odd n the lo pair
  }
  // class plane (first encoder stage): the 9 border classes' table entries of this lane's columns, once, instead of a global
  // load per element (the stage ran 225 us with the plane against 168 us without it)
  float ct[FN <= 2 ? 9 : 1][FN <= 2 ? FN : 1];
  if constexpr (FN <= 2) {
    if (p.cls_table != nullptr) {
#pragma unroll
      for (int c = 0; c < 9; ++c)
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) ct[c][ni] = nn[ni] < p.N ? p.cls_table[(size_t)c * p.N + nn[ni]] : 0.f;
    }
  }
  __syncthreads();   // every wave is done with the main loop's LDS
  // one pass of element work; SPLIT (split32 words vs fp32 rows) and CLS (class plane) are compile-time so that the inner loop has no branch
  auto fill = [&](int pass, auto split_c, auto cls_c) {
    constexpr bool SPLIT = decltype(split_c)::value, CLS = decltype(cls_c)::value;
#pragma unroll
    for (int mi = 0; mi < FM; ++mi) {
      if ((wm * TM + mi * FR) / RPASS != pass) continue;   // wave-uniform: a fragment's rows lie in one pass (RPASS % 32 == 0)
      // the rows' class values first, as one batch of loads (a load per row inside the element loop serialises LDS -> global -> use)
      float cvs[NE];
      int clss[NE];
      if constexpr (CLS) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int bc = ri_bc[wm * TM + mi * FR + row_of(e)];
          clss[e] = bc & 15;
          cvs[e] = p.cls_val[bc >> 4];
        }
      }
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int lrow = wm * TM + mi * FR + row_of(e);
        char* rowp = scratch + (lrow - pass * RPASS) * RP;
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
          const int n = nn[ni];
          float v = acc[mi][ni][e];
          if constexpr (CLS) {
            if constexpr (FN <= 2) {
              float tv = ct[0][ni];
#pragma unroll
              for (int c = 1; c < 9; ++c) tv = clss[e] == c ? ct[c][ni] : tv;
              v += cvs[e] * tv;
            } else {
              if (n < p.N) v += cvs[e] * p.cls_table[(size_t)clss[e] * p.N + n];
            }
          }
          v = v * sc[ni] + sh[ni];
          v = v > 0.f ? v : v * p.slope;
          if constexpr (!SPLIT) {   // plain fp32 rows
            *reinterpret_cast<float*>(rowp + (wn * TN + ni * FR + col) * 4) = v;
          } else {
            const __bf16 hb = (__bf16)v;
            const __bf16 lb = (__bf16)(v - (float)hb);
            const unsigned h16 = __builtin_bit_cast(unsigned short, hb), l16 = __builtin_bit_cast(unsigned short, lb);
            // the neighbour lane's (n ^ 1) halves by a DPP quad permute [1,0,3,2]: no LDS round trip (__shfl_xor is a ds_bpermute)
            const unsigned both = h16 | (l16 << 16);
            const unsigned other = (unsigned)__builtin_amdgcn_mov_dpp((int)both, 0xB1, 0xF, 0xF, true);
            *reinterpret_cast<unsigned*>(rowp + woff[ni]) = (n & 1) ? ((other >> 16) | (l16 << 16)) : (h16 | (other << 16));
          }
        }
      }
    }
  };
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    if (p.dst_split) {
      if (p.cls_table != nullptr) fill(pass, std::true_type{}, std::true_type{});
      else fill(pass, std::true_type{}, std::false_type{});
    } else {
      if (p.cls_table != nullptr) fill(pass, std::false_type{}, std::true_type{});
      else fill(pass, std::false_type{}, std::false_type{});
    }
    __syncthreads();
    for (int idx = tid; idx < RPASS * PIECES; idx += NTH) {
      const int r = idx / PIECES, pc = idx - r * PIECES;
      const int out = ri_out[pass * RPASS + r];
      if (out >= 0 && n0 + pc * 4 < p.N)   // N % 4 == 0: a piece is inside the row or outside it
        *reinterpret_cast<f32x4*>(p.dst + (size_t)out * p.ldc + n0 + pc * 4) = *reinterpret_cast<const f32x4*>(scratch + r * RP + pc * 16);
    }
    if (pass + 1 < NPASS) __syncthreads();
  }
}

// The same store for TRANSPOSED accumulators: kernels that feed the weights as the MFMA's A operand and the pixels as its B operand
// (conv_dma.hip with 16x16x32 fragments, convt_quad.hip without the fused head; the strip-walker kernels do the same) hold per
// lane runs of FOUR CONSECUTIVE CHANNELS of ONE pixel -- 16x16: acc[mi][ni] = channels 16 ni + 4 (lane >> 4) + j of pixel 16 mi +
// (lane & 15); 32x32: acc[mi][ni][4 g + j] = channels 32 ni + 8 g + 4 (lane >> 5) + j of pixel 32 mi + (lane & 31) -- so BN /
// activation / hi-lo split are vector operations and one 8- or 16-byte LDS write per run, instead of a DPP exchange, ~20 scalar
// VALU and a 4-byte write per VALUE.  Same values, same LDS image, same copy-out.  No class plane (the first encoder stage never
// runs here).
template <int BM, int BN, int WM, int WN, int FR, int SCRATCH, typename AccT>
__device__ __forceinline__ void nhwc_tile_store_T(const IGemmP& p, AccT (&acc)[BM / WM / FR][BN / WN / FR], char* scratch, const int* ri_out,
                                                     int n0, int tid) {
  constexpr int NTH = 64 * WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / FR, FN = TN / FR;
  constexpr int NG = FR == 32 ? 4 : 1;          // runs of four channels per fragment and lane
  constexpr int RP = BN * 4 + 16;
  constexpr int RMAX = SCRATCH / RP;
  constexpr int RPASS = RMAX >= BM ? BM : (RMAX >= BM / 2 ? BM / 2 : (RMAX >= BM / 4 ? BM / 4 : BM / 8));
  constexpr int NPASS = BM / RPASS;
  constexpr int PIECES = BN / 4;
  static_assert(RPASS * RP <= SCRATCH && RPASS % FR == 0 && BN % 32 == 0, "scratch too small for the tile store");
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int px = lane & (FR - 1), cq = lane / FR;
  f32x4 sc[FN][NG], sh[FN][NG];
  int nl[FN][NG];
#pragma unroll
  for (int ni = 0; ni < FN; ++ni)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      nl[ni][g] = wn * TN + ni * FR + (FR == 32 ? 8 * g : 0) + 4 * cq;   // first of the run's four columns inside the tile (N % 4 == 0)
      const int n = n0 + nl[ni][g];
      const bool in = n < p.N;
      sc[ni][g] = (p.scale != nullptr && in) ? *reinterpret_cast<const f32x4*>(p.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
      sh[ni][g] = (p.shift != nullptr && in) ? *reinterpret_cast<const f32x4*>(p.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  __syncthreads();   // every wave is done with the main loop's LDS
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
    for (int mi = 0; mi < FM; ++mi) {
      if ((wm * TM + mi * FR) / RPASS != pass) continue;   // wave-uniform
      char* rowp = scratch + (wm * TM + mi * FR + px - pass * RPASS) * RP;
#pragma unroll
      for (int ni = 0; ni < FN; ++ni)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][4 * g + j];
          v = v * sc[ni][g] + sh[ni][g];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * p.slope;
          if (p.dst_split) {
            const bf16x4 hi = __builtin_convertvector(v, bf16x4);
            const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
            char* w = rowp + (nl[ni][g] & ~31) * 4 + (nl[ni][g] & 31) * 2;
            *reinterpret_cast<bf16x4*>(w) = hi;
            *reinterpret_cast<bf16x4*>(w + 64) = lo;
          } else {
            *reinterpret_cast<f32x4*>(rowp + nl[ni][g] * 4) = v;
          }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < RPASS * PIECES; idx += NTH) {
      const int r = idx / PIECES, pc = idx - r * PIECES;
      const int out = ri_out[pass * RPASS + r];
      if (out >= 0 && n0 + pc * 4 < p.N)
        *reinterpret_cast<f32x4*>(p.dst + (size_t)out * p.ldc + n0 + pc * 4) = *reinterpret_cast<const f32x4*>(scratch + r * RP + pc * 16);
    }
    if (pass + 1 < NPASS) __syncthreads();
  }
}

// Fused epilogue shared by the LDS-staged kernel and the tap-sharing transposed-conv kernel: class-plane bias, BN scale/shift
// or bias, LeakyReLU/ReLU and the NHWC / de-sliced store; with head_w, the last decoder stage's 1x1 head on the on-chip tile.
// As0 / Bs0: LDS scratch of at least BM*LDK and max(BN,32)*LDK floats (the main loop's tiles, free by now).
template <int BM, int BN, int WM, int WN, int FR, typename AccT, int SCRATCH = 0>   // SCRATCH: bytes of LDS at As0 (0: unknown)
__device__ __forceinline__ void fused_epilogue(const IGemmP& p, AccT (&acc)[BM / WM / FR][BN / WN / FR], float* As0, float* Bs0,
                                               const int* ri_out, const int* ri_bc, int n0, int tid, bool stage_head = true) {
  // stage_head = false: the head matrix already sits at Bs0 (a caller running several tiles of one launch through this epilogue)
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / FR, FN = TN / FR;
  constexpr int GK = FR == 32 ? 8 : 16;
  constexpr int NE = FR == 32 ? 16 : 4;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & (FR - 1);
  const int fk = (lane / FR) * 4;
  auto row_of = [&](int e) { return FR == 32 ? (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) : (lane >> 4) * 4 + e; };
  const int col = lane & (FR - 1);
  constexpr int NTE = 64 * WM * WN;   // threads of the calling block
  if constexpr (BN <= 32 && WN == 1) {
    if (p.head_w != nullptr) {
      // Last decoder stage + head in one kernel (separator_cnn.py:133-134,163-168): y = ReLU(BN(convT)) stays on chip, a second
      // small MFMA pass applies the 1x1 conv, the result is transposed through LDS and stored de-sliced with one contiguous
      // (s, pixel-run) segment per wave instruction instead of 4-byte scatters.
      constexpr int LDT = BM + 1;    // [n'][m] staging stride: conflict-free column writes
      constexpr int NGH = FR / GK;   // fragment groups of the FR-deep head contraction
      float* Y = As0;
      float* Wh = Bs0;
      // this lane's per-channel constants first: their load latency runs under the barrier instead of behind it (the loads cannot
      // move across __syncthreads by themselves; at four epilogue passes per workgroup they were ~1 us of each)
      const float scn = (p.scale != nullptr && col < p.N) ? p.scale[col] : 1.f;
      const float shn = (p.shift != nullptr && col < p.N) ? p.shift[col] : 0.f;
      const float hb = col < p.N ? p.head_b[col] : 0.f;
      __syncthreads();  // every wave is done with the main loop's LDS tiles
      {
        const int n = col;
#pragma unroll
        for (int mi = 0; mi < FM; ++mi)
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            const int lrow = wm * TM + mi * FR + row_of(e);
            float v = 0.f;
            if (n < p.N) {
              v = acc[mi][0][e] * scn + shn;
              v = v > 0.f ? v : v * p.slope;
            }
            Y[lrow * LDK + n] = v;
          }
        if (stage_head)
          for (int idx = tid; idx < FR * FR; idx += NTE) {  // FR x FR head matrix, zero padded
            const int n2 = idx / FR, k = idx % FR;
            Wh[n2 * LDK + k] = (n2 < p.N && k < p.N) ? p.head_w[n2 * p.N + k] : 0.f;
          }
      }
      __syncthreads();
      AccT acc2[FM];
#pragma unroll
      for (int mi = 0; mi < FM; ++mi)
#pragma unroll
        for (int e = 0; e < NE; ++e) acc2[mi][e] = 0.f;
#pragma unroll
      for (int g = 0; g < NGH; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(&Wh[frow * LDK + g * GK + fk]);
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(&Y[(wm * TM + mi * FR + frow) * LDK + g * GK + fk]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if constexpr (FR == 32)
              acc2[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc2[mi], 0, 0, 0);
            else
              acc2[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc2[mi], 0, 0, 0);
          }
        }
      }
      __syncthreads();  // Y fully consumed before it is overwritten by the transposed staging
      float* Tt = Y;
#pragma unroll
      for (int mi = 0; mi < FM; ++mi)
#pragma unroll
        for (int e = 0; e < NE; ++e) Tt[col * LDT + wm * TM + mi * FR + row_of(e)] = acc2[mi][e] + hb;
      __syncthreads();
      const size_t plane2 = (size_t)p.Ho * p.Wo;
      const int Cc2 = p.N >> 4;
#pragma unroll
      for (int it = 0; it < 16 * BM / NTE; ++it) {
        const int i = tid + NTE * it;
        const int s = i / BM, m = i % BM;
        const int out = ri_out[m];
        if (out < 0) continue;
        float* dptr = p.dst + ((size_t)out + (size_t)s * plane2) * Cc2;
        if (Cc2 == 2) {
          float2 v2;
          v2.x = Tt[s * LDT + m];
          v2.y = Tt[(16 + s) * LDT + m];
          *reinterpret_cast<float2*>(dptr) = v2;
        } else {
          dptr[0] = Tt[s * LDT + m];
        }
      }
      return;
    }
  }
  if constexpr (SCRATCH >= 32 * (BN * 4 + 16) && BN % 32 == 0 && BM % 32 == 0) {
    if (p.out_mode == M2H_OUT_NHWC && p.N % 4 == 0 && p.ldc % 4 == 0 && (reinterpret_cast<size_t>(p.dst) & 15) == 0) {   // NHWC tile: whole rows through LDS
      nhwc_tile_store<BM, BN, WM, WN, FR, SCRATCH, AccT>(p, acc, reinterpret_cast<char*>(As0), ri_out, ri_bc, n0, tid);
      return;
    }
  }
  float sc[FN], sh[FN];
  int nn[FN];
#pragma unroll
  for (int ni = 0; ni < FN; ++ni) {
    const int n = n0 + wn * TN + ni * FR + col;
    nn[ni] = n;
    sc[ni] = (p.scale != nullptr && n < p.N) ? p.scale[n] : 1.f;
    sh[ni] = (p.shift != nullptr && n < p.N) ? p.shift[n] : 0.f;
  }
  const size_t plane = (size_t)p.Ho * p.Wo;
  const int Cc = p.N >> 4;
#pragma unroll
  for (int mi = 0; mi < FM; ++mi) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int lrow = wm * TM + mi * FR + row_of(e);
      const int out = ri_out[lrow];
      if (out < 0) continue;
      const int bc = ri_bc[lrow];
      float cv = 0.f;
      const float* ctab = nullptr;
      if (p.cls_table != nullptr) {
        cv = p.cls_val[bc >> 4];
        ctab = p.cls_table + (size_t)(bc & 15) * p.N;
      }
#pragma unroll
      for (int ni = 0; ni < FN; ++ni) {
        const int n = nn[ni];
        if (n >= p.N) continue;
        float v = acc[mi][ni][e];
        if (ctab != nullptr) v += cv * ctab[n];
        v = v * sc[ni] + sh[ni];
        v = v > 0.f ? v : v * p.slope;
        if (p.out_mode == M2H_OUT_NHWC) {
          if (p.dst_split) {
            // split32 store: lanes (n even, n odd) pair up; the even lane writes both hi halves, the odd lane both lo halves
            const __bf16 hb = (__bf16)v;
            const __bf16 lb = (__bf16)(v - (float)hb);
            const unsigned h16 = __builtin_bit_cast(unsigned short, hb), l16 = __builtin_bit_cast(unsigned short, lb);
            const unsigned ph_ = __shfl_xor(h16, 1, 64), pl_ = __shfl_xor(l16, 1, 64);
            const unsigned word = (n & 1) ? (pl_ | (l16 << 16)) : (h16 | (ph_ << 16));
            unsigned* drow = reinterpret_cast<unsigned*>(p.dst + (size_t)out * p.ldc + (n & ~31));
            drow[((n & 1) ? 16 : 0) + ((n & 31) >> 1)] = word;
          } else {
            p.dst[(size_t)out * p.ldc + n] = v;
          }
        } else {
          const int c = n >> 4, s = n & 15;
          p.dst[((size_t)out + (size_t)s * plane) * Cc + c] = v;
        }
      }
    }
  }
}

// conv_igemm.hip: split-K factor of a launch on BM x BN tiles (the one rule of both engines: equal factors keep their results bit-identical)
int choose_splitk(const IGemmP& p, int BM, int BN, size_t ws_bytes);

// conv_dma.hip: LDS-DMA engine; returns -2 when the launch is not one of its shapes (the caller falls through), 0 / error otherwise
int launch_igemm_dma(IGemmP& p, size_t ws_bytes, hipStream_t st);

// conv_patch.hip: shared-patch LDS-DMA engine (4x4/s2 convs and transposed-conv phases, N % 128 == 0); -2 when not one of its shapes
int launch_igemm_patch(IGemmP& p, size_t ws_bytes, hipStream_t st);   // p.S == 2 on return: the caller runs splitk_epilogue_kernel

// convt_quad.hip: four-phase transposed-conv kernel (split32 operands, N <= 64); -2 when the launch is not one of its shapes
int launch_convT_quad(IGemmP& p, hipStream_t st);

// conv_dma.hip: shape rule of the engine's two-way split-K launch (the fourth encoder stage at the benchmark batch)
bool dma_split2_rule(long M, int N, int Kw, int phases, bool ws_present, size_t ws_bytes);
int dma_deep_split(long M, int N, int Kw, int phases);   // K-parts of the engine's launch on a layer of 16 .. 223 tiles (1 = none)

}  // namespace m2h
