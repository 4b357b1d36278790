// Implicit-GEMM convolution for gfx950 (MI355X), fp32 in / fp32 accumulate on the matrix cores.
//
// Replaces, in the reference, every Conv2d / ConvTranspose2d (+BatchNorm2d(eval) + activation) of the
// separator U-Nets (audio_separation/rl/models/separator_cnn.py:5-24,46-52,128-135) and, through the same
// engine, the 3x3 convs of AcousticMem (rl/models/memory_nets.py:11-16).
//
// GEMM view:  D[m][n] = sum_k A[m][k] * W[n][k]
//   m = (b, q, r) over the output pixel grid of one sub-pixel phase, n = output channel,
//   k = (th, tw, ci): kernel tap x input channel, input channels being the concatenation of two NHWC
//   sources (the U-Net skip concat is read in place, never materialised).
// A is gathered on the fly from NHWC activations (channel-contiguous 16-byte segments, zero outside the
// image); W is pre-packed [n][k] (m2h_pack_conv*_weight).  Both are staged global -> VGPR -> LDS with a
// register double buffer (one barrier per 32-deep k-tile), LDS rows padded 32 -> 36 floats so that the
// ds_read_b128 fragment reads are bank-conflict free.  Each wave owns a (TM x TN) sub-tile as FM x FN
// v_mfma_f32_32x32x2_f32 accumulators: one 16-byte LDS read feeds four MFMAs (lane (i, h) holds
// k = 8g + 4h + {0..3} of row i; MFMA j of the group contracts k = 8g + j and 8g + 4 + j).  The f32 MFMA
// is a k-ordered fp32 FMA chain, so results match an fp32 CPU convolution to rounding.
//
// Epilogue (fused): optional target-class plane (border-aware bias), BatchNorm(eval) scale/shift,
// LeakyReLU/ReLU, and the store either NHWC or de-sliced straight into the reference's BHWC layout.
//
// Block -> tile map: n-tiles of one m-tile are consecutive on one XCD (blocks b and b+8 share an XCD's
// L2), so the gathered A panel is fetched from HBM once and re-read from L2 by its sibling n-tiles.
#include <string>

#include "igemm_common.h"

#ifndef M2H_SCHED
#define M2H_SCHED 0  // instruction-interleave experiment selector for the k-loop (0 = compiler default)
#endif

namespace m2h {

#ifdef M2H_CLOCK_DIAG
// Diagnostic build only (tools/clock_diag.py): shader-clock vs 100 MHz real-time stamps around the k-loop of each block, to read
// the clock the chip holds under this kernel (MI355X_MICROARCH.md, DVFS give-back item 6).  Never compiled into libm2h.so.
__device__ unsigned long long g_clock_dbg[8192][6];
#endif

// FR = MFMA fragment edge: 32 (v_mfma_f32_32x32x2_f32, 8 k per 16-byte LDS read) or 16 (v_mfma_f32_16x16x4_f32, 16 k per read;
// used for N <= 16 so that a 16-channel layer does not pay for a half-empty 32-wide tile).  Same FLOP rate per cycle.
//
// FAST = 1 (both sources' channel counts multiples of the 32-deep k-tile, operands < 4 GiB): every k-tile lies inside one
// (tap, source) segment, so the whole k decode is SCALAR (SGPR) work and a load's address is `uniform base + per-lane 32-bit
// offset` with the per-lane part recomputed only when the segment changes (every C/32 tiles).  The generic path decodes k per
// lane (any C % 4 == 0) and costs ~250 vector instructions per k-tile, which made the k-loop issue-bound beside 64 MFMAs.
//
// SPLIT = 1 | 2 ("bf16x3" math; 2 = both operands already arrive in the split32 layout, no conversion in the loop): fp32 operands are split on their way into LDS into bf16 high and low parts (x = hi + lo to 2^-17
// relative) and each product a*b is formed as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on the bf16 matrix pipe with fp32 accumulation
// (the dropped a_lo*b_lo term is 2^-16 of the dropped precision again): three bf16 MFMAs replace sixteen (32x32) or eight
// (16x16) fp32 MFMAs per fragment and 32-deep k-tile -- the bf16 pipe is 16x the fp32 one -- at ~16 mantissa bits per product
// instead of 24.  Tensors in HBM stay fp32; the LDS row keeps its 144-byte stride ([hi 64 B | lo 64 B | pad]) and the fragment
// reads are byte-for-byte those of the fp32 path.  Measured end to end on the U-Net pair: rel-L1 1e-5 vs the fp32 reference
// (plain bf16 operands: 4e-3..6e-3, outside the 1e-3 contract).
template <int BM, int BN, int WM, int WN, int NSTAGE, int FR = 32, int FAST = 0, int SPLIT = 0>
__global__ __launch_bounds__(64 * WM * WN, (SPLIT && WM * WN == 4) ? 2 : 1) void igemm_f32_kernel(const IGemmP p) {
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per block");
  constexpr int NT = 64 * WM * WN;           // threads per block
  constexpr int RPP = NT / 8;                // tile rows staged per pass (a row = 8 threads x 16 bytes)
  constexpr int TM = BM / WM, TN = BN / WN;  // wave tile
  constexpr int FM = TM / FR, FN = TN / FR;  // MFMA fragments per wave
  constexpr int BNS = BN < RPP ? RPP : BN;   // staged weight rows (a pass stages RPP rows)
  constexpr int AR = BM / RPP, BR = BNS / RPP; // staged rows per thread
  constexpr int GK = FR == 32 ? 8 : 16;      // k covered by one fragment group (one 16-byte read per lane)
  constexpr int NG = BK / GK;                // fragment groups per k-tile
  constexpr int NE = FR == 32 ? 16 : 4;      // accumulator elements per lane
  using AccT = typename std::conditional<FR == 32, f32x16, f32x4>::type;
  static_assert(FM >= 1 && FN >= 1, "wave tile must hold at least one fragment");

  __shared__ __attribute__((aligned(16))) float As[NSTAGE][BM * LDK];
  __shared__ __attribute__((aligned(16))) float Bs[NSTAGE][BNS * LDK];
  __shared__ int ri_qh[BM], ri_rw[BM], ri_bpix[BM], ri_out[BM], ri_bc[BM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int seg = tid & 7;    // 16-byte segment inside the 128-byte k-tile row
  const int srow = tid >> 3;  // 0..RPP-1

  // ---- block -> (m-tile, n-tile): siblings (same m-tile) run back to back on one XCD ----
  const int L = blockIdx.x;
  const int xcd = L & 7;
  int idx = L >> 3;
  // transposed conv: the four sub-pixel phases of one m-tile read the same 3x3 input neighbourhood; with pmaj they are
  // consecutive blocks of one XCD (input tile served by that XCD's L2) instead of four passes over the whole input
  int phase = 0;
  if (p.convT) {
    if (p.pmaj) {
      phase = idx & 3;
      idx >>= 2;
    } else {
      phase = blockIdx.z;
    }
  }
  const int mt = (idx / p.NT) * 8 + xcd;
  const int nt = idx - (idx / p.NT) * p.NT;
  if (mt >= p.MT) return;  // whole block leaves before any barrier
  const int m0 = mt * BM;
  const int n0 = nt * BN;

  int mulh = p.mulh, offh = p.offh, mulw = p.mulw, offw = p.offw, ph = p.ph, pw = p.pw;
  const float* wbase = p.w;
  if (p.convT) {
    ph = phase >> 1;
    pw = phase & 1;
    mulh = 2 * ph - 1;
    mulw = 2 * pw - 1;
    offh = 0;
    offw = 0;
    wbase += (size_t)phase * p.N * p.K;
  }

  // ---- per-row (output pixel) bookkeeping, once per block ----
  for (int r = threadIdx.x; r < BM; r += NT) {
    const int m = m0 + r;
    int qh = -(1 << 24), rw = -(1 << 24), bpix = 0, out = -1, bc = 0;
    if (m < p.M) {
      int q, rr, b;
      decode_row(p, m, ph, pw, q, rr, b, out, bc);
      qh = q * p.stride + offh;
      rw = rr * p.stride + offw;
      bpix = b * p.Hi * p.Wi;
    }
    ri_qh[r] = qh;
    ri_rw[r] = rw;
    ri_bpix[r] = bpix;
    ri_out[r] = out;
    ri_bc[r] = bc;
  }
  __syncthreads();

  int a_qh[AR], a_rw[AR], a_bpix[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    a_qh[i] = ri_qh[srow + RPP * i];
    a_rw[i] = ri_rw[srow + RPP * i];
    a_bpix[i] = ri_bpix[srow + RPP * i];
  }

  AccT acc[FM][FN];
#pragma unroll
  for (int mi = 0; mi < FM; ++mi)
#pragma unroll
    for (int ni = 0; ni < FN; ++ni)
#pragma unroll
      for (int e = 0; e < NE; ++e) acc[mi][ni][e] = 0.f;
  // accumulator element e of this lane -> fragment row (C/D layout of the two MFMA shapes); column = lane & (FR-1)
  auto row_of = [&](int e) { return FR == 32 ? (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) : (lane >> 4) * 4 + e; };

  // Two register sets for the staged tile: the loads of k-tile t+2 are issued at the top of tile t's MFMA phase and are not
  // consumed (LDS write) until the end of tile t+1, so a loaded-chip HBM/L2 round trip (several microseconds) has a whole
  // tile of MFMAs (>= 4096 cycles on the 128x128 tile) to land instead of the tail of the current one.
  f32x4 ra[2][AR], rb[2][BR];
  unsigned okmask[2] = {0u, 0u};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  const int nk_all = ((FAST ? p.Kw : p.K) + BK - 1) / BK;
  const int split = blockIdx.y;
  const int kt0 = (int)(((long)nk_all * split) / p.S);
  const int kt1 = (int)(((long)nk_all * (split + 1)) / p.S);
  const int k_end = min(p.K, kt1 * BK);  // loads past this block's k-range are masked (they re-read element 0: a cache hit)

  // ---- loader state for the tile being loaded (advanced incrementally, one k-tile at a time) ----
  // generic: per-lane k = kt*BK + seg*4 -> (th, tw, ci)
  int ld_k, ld_th, ld_tw, ld_ci;
  bool t_kok;
  int t_dh, t_dw, t_Cs, t_c;
  const float* t_src;
  // FAST: uniform tile index / tap / channel offset; per-lane byte offsets of the staged rows inside the current segment
  int u_kt, u_th, u_tw, u_ci;
  unsigned voffA[AR], voffB[BR], okA = 0;

  auto seek_generic = [&](int kt) {
    ld_k = kt * BK + seg * 4;
    int tap = 0;
    ld_ci = ld_k;
    if (p.ntap > 1) {
      tap = (unsigned)ld_k / (unsigned)p.Ctot;
      ld_ci = ld_k - tap * p.Ctot;
    }
    ld_th = (unsigned)tap / (unsigned)p.ntw;
    ld_tw = tap - ld_th * p.ntw;
  };
  // FAST: per-lane row offsets for the current (tap, source) segment
  auto segment_rows = [&]() {
    const int dh = u_th * mulh, dw = u_tw * mulw;
    const int Cs = (u_ci >= p.C0 && p.src1 != nullptr) ? p.C1 : p.C0;
    okA = 0;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int ih = a_qh[i] + dh, iw = a_rw[i] + dw;
      const bool ok = (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
      voffA[i] = ok ? ((unsigned)(a_bpix[i] + ih * p.Wi + iw) * (unsigned)Cs + (unsigned)(seg * 4)) * 4u : 0u;
      okA |= ok ? (1u << i) : 0u;
    }
  };
  auto loader_seek = [&](int kt) {
    if constexpr (FAST) {
      u_kt = kt;
      const int k0 = kt * BK;
      const int tap = p.ntap > 1 ? k0 / p.Ctot : 0;   // index inside the tap window
      u_ci = k0 - tap * p.Ctot;
      u_th = p.th0 + tap / p.twn;
      u_tw = p.tw0 + tap % p.twn;
      segment_rows();
#pragma unroll
      for (int j = 0; j < BR; ++j)  // rows past N re-read row N-1 (their products are never stored): no mask on the weight side
        voffB[j] = ((unsigned)min(n0 + srow + RPP * j, p.N - 1) * (unsigned)p.K + (unsigned)(seg * 4)) * 4u;
    } else {
      seek_generic(kt);
    }
  };
  auto loader_next = [&]() {
    if constexpr (FAST) {
      if (u_kt + 1 < kt1) {  // past this block's k-range the loader stays on the last tile (a harmless cached re-read)
        ++u_kt;
        u_ci += BK;
        bool reseg = u_ci == p.C0 && p.src1 != nullptr;
        if (u_ci == p.Ctot) {
          u_ci = 0;
          reseg = true;
          if (++u_tw == p.tw0 + p.twn) {
            u_tw = p.tw0;
            ++u_th;
          }
        }
        if (reseg) segment_rows();  // wave-uniform branch
      }
    } else {
      if (p.Ctot >= BK) {  // at most one wrap per 32-deep step; selects, not branches (the wrap differs per lane)
        ld_k += BK;
        ld_ci += BK;
        const bool wrap = ld_ci >= p.Ctot;
        ld_ci -= wrap ? p.Ctot : 0;
        ld_tw += wrap ? 1 : 0;
        const bool wrap2 = ld_tw == p.ntw;
        ld_tw = wrap2 ? 0 : ld_tw;
        ld_th += wrap2 ? 1 : 0;
      } else {
        seek_generic(ld_k / BK + 1);
      }
    }
  };
  // Generic loads are unconditional (clamped to element 0 of the source when masked) and zeroed by a select afterwards: no
  // divergent branches in the k-loop, so the loads can be scheduled into the MFMA shadows.
  // The loaded value is NOT touched until store_tile (a select right after the load would force a vmcnt(0) wait there);
  // the validity bits travel in a mask.
  auto load_a = [&](int set, int i) {
    const int ih = a_qh[i] + t_dh, iw = a_rw[i] + t_dw;
    const bool ok = t_kok && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
    const unsigned off = ok ? (unsigned)(a_bpix[i] + ih * p.Wi + iw) * (unsigned)t_Cs + (unsigned)t_c : 0u;
    ra[set][i] = *reinterpret_cast<const f32x4*>(t_src + off);
    okmask[set] = ok ? (okmask[set] | (1u << i)) : (okmask[set] & ~(1u << i));
  };
  auto load_b = [&](int set, int j) {
    const int n = n0 + srow + RPP * j;
    const bool ok = t_kok && n < p.N;
    const unsigned off = ok ? (unsigned)n * (unsigned)p.K + (unsigned)ld_k : 0u;
    rb[set][j] = *reinterpret_cast<const f32x4*>(wbase + off);
    okmask[set] = ok ? (okmask[set] | (1u << (8 + j))) : (okmask[set] & ~(1u << (8 + j)));
  };
  auto load_tile = [&](int set) {
    if constexpr (FAST) {
      const bool second = u_ci >= p.C0 && p.src1 != nullptr;
      const char* baseA = reinterpret_cast<const char*>(second ? p.src1 + (u_ci - p.C0) : p.src0 + u_ci);  // uniform
      const char* baseB = reinterpret_cast<const char*>(wbase + (size_t)(u_th * p.ntw + u_tw) * p.Ctot + u_ci);  // uniform
#pragma unroll
      for (int i = 0; i < AR; ++i) ra[set][i] = *reinterpret_cast<const f32x4*>(baseA + voffA[i]);
#pragma unroll
      for (int j = 0; j < BR; ++j) rb[set][j] = *reinterpret_cast<const f32x4*>(baseB + voffB[j]);
      okmask[set] = okA;
    } else {
      t_kok = ld_k < k_end;
      t_dh = ld_th * mulh;
      t_dw = ld_tw * mulw;
      t_src = p.src0;
      t_Cs = p.C0;
      t_c = ld_ci;
      if (ld_ci >= p.C0 && p.src1 != nullptr) {  // second source; beyond-K padding tiles of a single-source conv keep src0
        t_src = p.src1;
        t_Cs = p.C1;
        t_c = ld_ci - p.C0;
      }
#pragma unroll
      for (int i = 0; i < AR; ++i) load_a(set, i);
#pragma unroll
      for (int j = 0; j < BR; ++j) load_b(set, j);
    }
  };
  // SPLIT: [hi bf16 x 32 | lo bf16 x 32] per row; this thread's four k-values land at byte seg*8 of each half
  auto store_split = [&](float* rowp, f32x4 v) {
    const bf16x4 hi = __builtin_convertvector(v, bf16x4);
    const f32x4 hf = __builtin_convertvector(hi, f32x4);
    const bf16x4 lo = __builtin_convertvector(v - hf, bf16x4);
    char* base = reinterpret_cast<char*>(rowp) + seg * 8;
    *reinterpret_cast<bf16x4*>(base) = hi;
    *reinterpret_cast<bf16x4*>(base + 64) = lo;
  };
  auto store_tile = [&](int set, int buf) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const f32x4 v = (okmask[set] & (1u << i)) ? ra[set][i] : zero4;
      if constexpr (SPLIT == 1)
        store_split(&As[buf][(srow + RPP * i) * LDK], v);
      else
        *reinterpret_cast<f32x4*>(&As[buf][(srow + RPP * i) * LDK + seg * 4]) = v;
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      const f32x4 v = (FAST || (okmask[set] & (1u << (8 + j)))) ? rb[set][j] : zero4;
      if constexpr (SPLIT == 1)
        store_split(&Bs[buf][(srow + RPP * j) * LDK], v);
      else
        *reinterpret_cast<f32x4*>(&Bs[buf][(srow + RPP * j) * LDK + seg * 4]) = v;
    }
  };

  const int frow = lane & (FR - 1);  // fragment row (A: pixel, B: channel)
  const int fk = (lane / FR) * 4;    // k offset of this lane group inside a GK-deep fragment group

  f32x4 fa[2][FM], fb[2][FN];  // fragment double buffer: group g+1 is read from LDS while group g's MFMAs run
  auto read_frags = [&](int buf, int g, int slot) {
#pragma unroll
    for (int mi = 0; mi < FM; ++mi)
      fa[slot][mi] = *reinterpret_cast<const f32x4*>(&As[buf][(wm * TM + mi * FR + frow) * LDK + g * GK + fk]);
#pragma unroll
    for (int ni = 0; ni < FN; ++ni)
      fb[slot][ni] = *reinterpret_cast<const f32x4*>(&Bs[buf][(wn * TN + ni * FR + frow) * LDK + g * GK + fk]);
  };
  auto mfma_group = [&](int slot) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mi = 0; mi < FM; ++mi)
#pragma unroll
        for (int ni = 0; ni < FN; ++ni)
          if constexpr (FR == 32)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][mi][j], fb[slot][ni][j], acc[mi][ni], 0, 0, 0);
          else
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[slot][mi][j], fb[slot][ni][j], acc[mi][ni], 0, 0, 0);
  };
  // SPLIT: the 16-byte fragment reads are the same four (FR 32) / two (FR 16) groups; group g < NG/2 holds the hi parts of
  // k-step g, group g + NG/2 the lo parts.  Per step: hi*hi + hi*lo + lo*hi (small terms first).
  auto mfma_bf16 = [&](const f32x4& a, const f32x4& b, AccT& c) {
    if constexpr (FR == 32)
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };
  auto mfma_tile = [&](int buf) {
    if constexpr (SPLIT) {
      constexpr int NSTEP = NG / 2;
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        read_frags(buf, st, 0);            // hi
        read_frags(buf, st + NSTEP, 1);    // lo
#pragma unroll
        for (int mi = 0; mi < FM; ++mi)
#pragma unroll
          for (int ni = 0; ni < FN; ++ni) {
            mfma_bf16(fa[1][mi], fb[0][ni], acc[mi][ni]);
            mfma_bf16(fa[0][mi], fb[1][ni], acc[mi][ni]);
            mfma_bf16(fa[0][mi], fb[0][ni], acc[mi][ni]);
          }
      }
    } else {
      read_frags(buf, 0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) read_frags(buf, g + 1, (g + 1) & 1);
        mfma_group(g & 1);
      }
    }
  };
  // One k-tile t (straight-line, no conditionals): issue the loads of tile t+2 into register set `par`, run tile t's MFMAs
  // from LDS stage `rd`, then write tile t+1 (register set par^1, loaded during tile t-1) to stage `wr`.
  auto tile_step = [&](int par, int rd, int wr) {
    loader_next();
    load_tile(par);
    mfma_tile(rd);
    if constexpr (NSTAGE == 1) __syncthreads();  // single stage: everyone is done reading before it is overwritten
    store_tile(par ^ 1, wr);
    __syncthreads();
  };

  // prologue: tile 0 -> LDS stage 0; tile 1 in flight in register set 1
  loader_seek(kt0);
  load_tile(0);
  store_tile(0, 0);
  if (kt1 - kt0 > 1) {
    loader_next();
    load_tile(1);
  }
  __syncthreads();
#ifdef M2H_CLOCK_DIAG
  const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  {
    // all but the last tile; unrolled by two so that register sets and LDS stages are compile-time
    const int nfull = kt1 - kt0 - 1;
    int t = 0;
    for (; t + 1 < nfull; t += 2) {
      tile_step(0, 0, NSTAGE == 2 ? 1 : 0);
      tile_step(1, NSTAGE == 2 ? 1 : 0, 0);
    }
    if (t < nfull) {
      tile_step(0, 0, NSTAGE == 2 ? 1 : 0);
      mfma_tile(NSTAGE == 2 ? 1 : 0);
    } else {
      mfma_tile(0);
    }
  }
#ifdef M2H_CLOCK_DIAG
  if (tid == 0 && blockIdx.y == 0) {
    const unsigned bi = blockIdx.x + gridDim.x * blockIdx.z;
    if (bi < 8192) {
      g_clock_dbg[bi][0] = __builtin_amdgcn_s_memtime() - dbg_t0;
      const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
      g_clock_dbg[bi][1] = r1 - dbg_r0;
      g_clock_dbg[bi][2] = dbg_r0;
      g_clock_dbg[bi][3] = r1;
      unsigned hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_clock_dbg[bi][4] = hwid;
      g_clock_dbg[bi][5] = xcc;
    }
  }
#endif

  if (p.S > 1) {
    // split-K: raw partial sums to the slab [phase][split][M][N]; BN/activation/store happen in the reduce kernel
    float* slab = p.ws + ((size_t)(phase * p.S + split) * p.M) * p.N;
    const int col = lane & (FR - 1);
#pragma unroll
    for (int mi = 0; mi < FM; ++mi)
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int m = m0 + wm * TM + mi * FR + row_of(e);
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
          const int n = n0 + wn * TN + ni * FR + col;
          if (n < p.N) slab[(size_t)m * p.N + n] = acc[mi][ni][e];
        }
      }
    return;
  }

  // ---- fused epilogue ----
  fused_epilogue<BM, BN, WM, WN, FR, AccT, NSTAGE * BM * LDK * 4>(p, acc, &As[0][0], &Bs[0][0], ri_out, ri_bc, n0, tid);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Tap-sharing transposed-conv kernel (bf16x3 math, N <= 64): one sub-pixel phase of ConvTranspose2d(4, 2, 1) is a 2x2-tap
// stride-1 conv, and its four taps read the SAME input pixels shifted by one row / one column.  The LDS-staged engine above
// treats each tap as its own k-tile and fetches the 128-pixel operand tile four times; with the cheap bf16 products that
// re-fetch (L1/TA traffic, splits, LDS writes, barriers) is what the narrow late decoder stages spend their time on.  Here a
// k-step is a 32-CHANNEL chunk: the block stages the (R+1) x (Wq+1) input pixels its 128 output pixels touch ONCE per chunk
// (R = 128 / Wq image rows), plus the four taps' weight rows, and runs the four taps' MFMAs from row-shifted windows of that
// one LDS image.  Per thread the global offsets are fixed for the whole kernel (only a uniform channel base advances).
// Requires: conv_transpose, FAST channels, 128 % Wq == 0, Wq >= 32, Hq % (128 / Wq) == 0.  Tile, accumulators and epilogue
// (incl. the fused head) are those of igemm_f32_kernel<128, BN, 4, 1, *, FR, 1, 1>.
template <int BN, int FR, int PRE = 0, int BM = 128, int WM = 4>   // PRE: operands already in the split32 layout (plain copies into LDS)
__global__ __launch_bounds__(64 * WM, WM == 4 ? 2 : 1) void convT_tap_kernel(const IGemmP p) {
  // BM = 256 (two image rows of 128, ...): the staged image grows by one row instead of doubling and the weight rows are
  // shared by twice the outputs -- the kernel is bound by L2 -> LDS traffic (PMC: 49 % of wave cycles parked on waits,
  // matrix pipe 24 % busy), so bytes per output are what counts.
  // WM = 8 (512 threads, one block per CU, twice the outputs per block): the same wave tiles, but the staged image has one halo
  // row per 2 x as many rows and the weight rows serve 2 x the outputs: ~25 % fewer L2 -> LDS bytes per output.
  constexpr int WN = 1, NT = 64 * WM, RPP = NT / 8;
  constexpr int TM = BM / WM;                    // rows per wave
  constexpr int FM = TM / FR, FN = BN / FR;
  constexpr int GK = FR == 32 ? 8 : 16;
  constexpr int NG = BK / GK, NSTEP = NG / 2;
  constexpr int NE = FR == 32 ? 16 : 4;
  using AccT = typename std::conditional<FR == 32, f32x16, f32x4>::type;
  constexpr int PMAX = (BM / 128 + 1) * 129;     // staged input pixels: (R+1)*(Wq+1) <= this for Wq in {32, 64, 128}
  constexpr int AR = (PMAX * 8 + NT - 1) / NT;   // 16-byte loads per thread for the input image
  constexpr int BROWS = 4 * BN;                  // weight rows per chunk (4 taps x BN channels)
  constexpr int BRL = BROWS * 8 / NT;            // loads per thread for them
  static_assert(BROWS * 8 % NT == 0 && FM >= 1 && FN >= 1, "tile shape");
  __shared__ __attribute__((aligned(16))) float As[PMAX * LDK];
  __shared__ __attribute__((aligned(16))) float Bs[BROWS * LDK];
  __shared__ int ri_out[BM], ri_bc[BM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & (FR - 1);
  const int fk = (lane / FR) * 4;
  const int seg = tid & 7, srow = tid >> 3;

  // ---- block -> (m-tile, phase): phase fastest, m-tiles round-robin over the XCDs ----
  const int L = blockIdx.x;
  const int xcd = L & 7;
  int idx = L >> 3;
  const int phase = idx & 3;
  idx >>= 2;
  const int mt = idx * 8 + xcd;
  if (mt >= p.MT) return;
  const int m0 = mt * BM;
  const int ph = phase >> 1, pw = phase & 1;
  const int dh = 2 * ph - 1, dw = 2 * pw - 1;
  const int hoff = dh < 0 ? dh : 0, woff = dw < 0 ? dw : 0;
  const float* wbase = p.w + (size_t)phase * p.N * p.K;
  const int Wq = p.Wq, W1 = Wq + 1;
  const int R = BM / Wq;
  const int P = (R + 1) * W1;
  const int b0 = m0 / (p.Hq * Wq);
  const int q0 = (m0 / Wq) % p.Hq;

  for (int r = tid; r < BM; r += NT) {
    const int m = m0 + r;
    int out = -1, bc = 0;
    if (m < p.M) {
      int q, rr, b;
      decode_row(p, m, ph, pw, q, rr, b, out, bc);
    }
    ri_out[r] = out;
    ri_bc[r] = bc;
  }

  // ---- fixed per-thread geometry of the staged input image and weight rows ----
  int pixA[AR];        // input pixel index (b, ih, iw) of staged row l = srow + 32 i, or -1
  unsigned voffA[AR], voffB[BRL];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int l = srow + RPP * i;
    const int qi = l / W1, rr = l - qi * W1;
    const int ih = q0 + qi + hoff, iw = rr + woff;
    const bool ok = l < P && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi && b0 < p.B;
    pixA[i] = ok ? (b0 * p.Hi + ih) * p.Wi + iw : -1;
  }
#pragma unroll
  for (int j = 0; j < BRL; ++j) {
    const int row = srow + RPP * j;               // tap * BN + n
    const int tap = row / BN, n = min(row - tap * BN, p.N - 1);   // rows past N re-read row N-1 (never stored)
    voffB[j] = ((unsigned)n * (unsigned)p.K + (unsigned)(tap * p.Ctot + seg * 4)) * 4u;
  }
  auto set_source = [&](int second) {
    const int Cs = second ? p.C1 : p.C0;
#pragma unroll
    for (int i = 0; i < AR; ++i) voffA[i] = pixA[i] >= 0 ? ((unsigned)pixA[i] * (unsigned)Cs + (unsigned)(seg * 4)) * 4u : 0u;
  };

  AccT acc[FM][FN];
#pragma unroll
  for (int mi = 0; mi < FM; ++mi)
#pragma unroll
    for (int ni = 0; ni < FN; ++ni)
#pragma unroll
      for (int e = 0; e < NE; ++e) acc[mi][ni][e] = 0.f;

  // LDS rows of this lane's fragments for tap (0,0)-relative addressing: row(qi, r) = qi*W1 + r, tap adds (a*W1 + b)
  int fragrow[FM];
#pragma unroll
  for (int mi = 0; mi < FM; ++mi) {
    const int ml = wave * TM + mi * FR;          // first tile row of the fragment; FR <= Wq keeps it inside one image row
    fragrow[mi] = (ml / Wq) * W1 + (ml % Wq) + frow;
  }
  int tapoff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) tapoff[t] = ((t >> 1) * dh - hoff) * W1 + ((t & 1) * dw - woff);

  f32x4 ra[AR], rb[BRL];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  int c_ci = 0, c_second = 0, c_k = 0;           // channel offset inside the source, source, chunk index (uniform)
  auto load_chunk = [&]() {
    const char* baseA = reinterpret_cast<const char*>((c_second ? p.src1 : p.src0) + c_ci);
    const char* baseB = reinterpret_cast<const char*>(wbase + (c_second ? p.C0 : 0) + c_ci);
#pragma unroll
    for (int i = 0; i < AR; ++i) ra[i] = *reinterpret_cast<const f32x4*>(baseA + voffA[i]);
#pragma unroll
    for (int j = 0; j < BRL; ++j) rb[j] = *reinterpret_cast<const f32x4*>(baseB + voffB[j]);
  };
  auto next_chunk = [&]() {
    ++c_k;
    c_ci += BK;
    if (c_ci == (c_second ? p.C1 : p.C0) && !c_second && p.src1 != nullptr) {
      c_second = 1;
      c_ci = 0;
      set_source(1);
    }
  };
  auto store_split = [&](float* rowp, f32x4 v) {
    const bf16x4 hi = __builtin_convertvector(v, bf16x4);
    const f32x4 hf = __builtin_convertvector(hi, f32x4);
    const bf16x4 lo = __builtin_convertvector(v - hf, bf16x4);
    char* base = reinterpret_cast<char*>(rowp) + seg * 8;
    *reinterpret_cast<bf16x4*>(base) = hi;
    *reinterpret_cast<bf16x4*>(base + 64) = lo;
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int l = srow + RPP * i;
      if (l < PMAX) {
        const f32x4 v = pixA[i] >= 0 ? ra[i] : zero4;
        if constexpr (PRE)
          *reinterpret_cast<f32x4*>(&As[l * LDK + seg * 4]) = v;
        else
          store_split(&As[l * LDK], v);
      }
    }
#pragma unroll
    for (int j = 0; j < BRL; ++j) {
      if constexpr (PRE)
        *reinterpret_cast<f32x4*>(&Bs[(srow + RPP * j) * LDK + seg * 4]) = rb[j];
      else
        store_split(&Bs[(srow + RPP * j) * LDK], rb[j]);
    }
  };
  auto mfma_bf16 = [&](const f32x4& a, const f32x4& b, AccT& c) {
    if constexpr (FR == 32)
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };
  auto compute_chunk = [&]() {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        f32x4 ah[FM], al[FM], bh[FN], bl[FN];
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) {
          const float* rp = &As[(fragrow[mi] + tapoff[t]) * LDK + fk];
          ah[mi] = *reinterpret_cast<const f32x4*>(rp + st * GK);
          al[mi] = *reinterpret_cast<const f32x4*>(rp + (st + NSTEP) * GK);
        }
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
          const float* rp = &Bs[(t * BN + ni * FR + frow) * LDK + fk];
          bh[ni] = *reinterpret_cast<const f32x4*>(rp + st * GK);
          bl[ni] = *reinterpret_cast<const f32x4*>(rp + (st + NSTEP) * GK);
        }
#pragma unroll
        for (int mi = 0; mi < FM; ++mi)
#pragma unroll
          for (int ni = 0; ni < FN; ++ni) {
            mfma_bf16(al[mi], bh[ni], acc[mi][ni]);
            mfma_bf16(ah[mi], bl[ni], acc[mi][ni]);
            mfma_bf16(ah[mi], bh[ni], acc[mi][ni]);
          }
      }
    }
  };

  const int nch = p.Ctot / BK;
  set_source(0);
  load_chunk();
  store_chunk();
  __syncthreads();
  for (int c = 0; c + 1 < nch; ++c) {
    next_chunk();
    load_chunk();
    compute_chunk();
    __syncthreads();   // everyone is done reading the stage
    store_chunk();
    __syncthreads();
  }
  compute_chunk();
  __syncthreads();     // the staged image becomes the epilogue's scratch

  fused_epilogue<BM, BN, WM, WN, FR, AccT, PMAX * LDK * 4>(p, acc, As, Bs, ri_out, ri_bc, 0, tid);
}

// Fused L1 epilogue of the image-row kernels' 16-channel instantiations (IGemmP::l1_gt): lane (band n = lane & 15, pixel group lane >> 4)
// holds four consecutive time frames of band n per 16-pixel fragment -- 16 contiguous bytes of the target plane -- so the loss costs one
// 16-byte load per fragment; the gradient sign(y - g) / n leaves in the conv's own NHWC layout, y itself is never stored (update_sep,
// ppo.py:206-216 with memory_nets.py:16,62-67: 110 MB written and read back per epoch otherwise, and one launch).  Returns the lane's |y - g| sum.
template <int FM, typename AccT>
__device__ __forceinline__ float l1_row_epilogue(const IGemmP& p, const AccT (&acc)[FM], int b, int q, int lane, float sh) {
  const int n = lane & 15;
  float s = 0.f;
#pragma unroll
  for (int mi = 0; mi < FM; ++mi) {
    const int x0 = mi * 16 + (lane >> 4) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(p.l1_gt + ((size_t)(b * 16 + n) * p.Ho + q) * p.Wo + x0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = (acc[mi][e] + sh) - g[e];
      s += fabsf(d);
      p.dst[((size_t)(b * p.Ho + q) * p.Wo + x0 + e) * p.ldc + n] = d > 0.f ? p.l1_inv : (d < 0.f ? -p.l1_inv : 0.f);
    }
  }
  return s;
}

// the block's partial sum of the fused loss: lanes -> wave (shuffles) -> the four waves in wave order, one float per block
__device__ __forceinline__ void l1_block_partial(const IGemmP& p, float s, float* scratch4, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __syncthreads();                                   // (scratch4 aliases the main loop's LDS: every wave is done with it)
  if ((tid & 63) == 0) scratch4[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) p.l1_part[blockIdx.x] = (scratch4[0] + scratch4[1]) + (scratch4[2] + scratch4[3]);
}

// loss = inv * sum of the blocks' partials, fixed order (one block of 256 threads; n <= 1024)
__global__ __launch_bounds__(256) void l1_partials_sum_kernel(const float* __restrict__ part, int n, float inv, float* __restrict__ loss) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = ((sh[0] + sh[1]) + (sh[2] + sh[3])) * inv;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Image-row 3x3 convolution (fp32 MFMA): Conv2d(3x3, stride 1, pad 1) over 16- or 32-channel, 32-pixel-wide images with N <= 32
// output channels -- AcousticMem's two convs (rl/models/memory_nets.py:11-16) and the input gradient of the second one, at the
// 1.7 M pixels of an update_sep epoch.  The general engine fetches each pixel's nine taps separately (and its scalar loader
// needs C % 32 == 0, so the 16-channel input gradient ran on the per-lane decode); here a block keeps the whole weight matrix
// in LDS and walks chunks of FOUR image rows: the six input rows they touch are staged once as a zero-padded 6 x 34-pixel
// patch, every tap is a row / column shift of it, each wave owns one image row (32 pixels) x all output channels, and N <= 16
// runs on v_mfma_f32_16x16x4_f32.  Fragment reads are the 16-byte reads of the engine above (rows padded to C + 4 floats).
// Tap t = (th, tw) reads the input at (q + offh + th*mulh, r + offw + tw*mulw): forward (mul 1, off -1) and input gradient
// (mul -1, off 1) alike.  Epilogue: optional bias, ReLU / LeakyReLU, NHWC or de-sliced store.
template <int FR, int C>
__global__ __launch_bounds__(256, 2) void conv3x3_row_kernel(const IGemmP p) {
  constexpr int W = 32, PW = W + 2, ROWS = 4, PR = ROWS + 2;
  constexpr int CP = C + 4;                         // patch pixel stride (floats)
  constexpr int K = 9 * C, KP = K + 4;              // weight row stride: an odd multiple of 4 floats mod 64, like CP (conflict-free 16-byte reads)
  constexpr int GK = FR == 32 ? 8 : 16;             // k per fragment group (one 16-byte read per lane)
  constexpr int NG = C / GK;                        // groups per tap
  constexpr int FM = 32 / FR;                       // pixel fragments per wave (one image row)
  constexpr int NE = FR == 32 ? 16 : 4;
  constexpr int SEG = C / 4;                        // 16-byte segments per pixel
  constexpr int NPL = (PR * PW * SEG + 255) / 256;  // patch loads per thread
  using AccT = typename std::conditional<FR == 32, f32x16, f32x4>::type;
  static_assert(NG >= 1 && (KP % 64) % 8 == 4 && (CP % 64) % 8 == 4, "tile shape");
  __shared__ __attribute__((aligned(16))) float Wl[FR * KP];
  __shared__ __attribute__((aligned(16))) float Pl[PR * PW * CP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & (FR - 1), fk = (lane / FR) * 4;
  const int chunks = p.B * (p.Hq / ROWS);

  // weights [N][K] -> LDS rows (rows past N: zeros), once per block
  for (int i = tid; i < FR * (K / 4); i += 256) {
    const int n = i / (K / 4), s4 = i - n * (K / 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < p.N) v = *reinterpret_cast<const f32x4*>(p.w + (size_t)n * K + s4 * 4);
    *reinterpret_cast<f32x4*>(&Wl[n * KP + s4 * 4]) = v;
  }
  int shift[9];                                     // patch offset of tap t relative to the output pixel's own patch position
#pragma unroll
  for (int t = 0; t < 9; ++t) shift[t] = (p.offh + (t / 3) * p.mulh) * PW + (p.offw + (t % 3) * p.mulw);

  f32x4 rp[NPL];
  unsigned okm = 0;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto load_chunk = [&](int c) {
    const int b = c / (p.Hq / ROWS), q0 = (c - b * (p.Hq / ROWS)) * ROWS;
    okm = 0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int i = tid + 256 * j;
      const int l = i / SEG, seg = i - l * SEG;
      const int pr = l / PW, pc = l - pr * PW;
      const int ih = q0 + pr - 1, iw = pc - 1;
      const bool ok = i < PR * PW * SEG && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)W;
      const size_t off = ok ? ((size_t)(b * p.Hi + ih) * W + iw) * C + seg * 4 : (size_t)0;
      rp[j] = *reinterpret_cast<const f32x4*>(p.src0 + off);
      okm |= ok ? (1u << j) : 0u;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int i = tid + 256 * j;
      if (i < PR * PW * SEG) *reinterpret_cast<f32x4*>(&Pl[(i / SEG) * CP + (i % SEG) * 4]) = (okm & (1u << j)) ? rp[j] : zero4;
    }
  };
  const size_t plane = (size_t)p.Ho * p.Wo;
  const int Cc = p.N >> 4;
  float l1_sum = 0.f;
  for (int c = blockIdx.x; c < chunks; c += gridDim.x) {
    if (c == (int)blockIdx.x) load_chunk(c);
    __syncthreads();              // the previous chunk's fragment reads (and the weight stores) are done
    store_chunk();
    __syncthreads();
    if (c + (int)gridDim.x < chunks) load_chunk(c + gridDim.x);   // next chunk's loads fly under this chunk's MFMAs
    AccT acc[FM];
#pragma unroll
    for (int mi = 0; mi < FM; ++mi)
#pragma unroll
      for (int e = 0; e < NE; ++e) acc[mi][e] = 0.f;
    const int prow0 = (wave + 1) * PW + 1;          // this wave's image row inside the patch, column 0
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const f32x4 bw = *reinterpret_cast<const f32x4*>(&Wl[frow * KP + t * C + g * GK + fk]);
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(&Pl[(prow0 + mi * FR + frow + shift[t]) * CP + g * GK + fk]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if constexpr (FR == 32)
              acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bw[j], acc[mi], 0, 0, 0);
            else
              acc[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bw[j], acc[mi], 0, 0, 0);
          }
        }
      }
    }
    // epilogue: rows = pixels of image row (q0 + wave), columns = output channels
    const int b = c / (p.Hq / ROWS), q = (c - b * (p.Hq / ROWS)) * ROWS + wave;
    const int n = lane & (FR - 1);
    const float sh = (p.shift != nullptr && n < p.N) ? p.shift[n] : 0.f;
    if constexpr (FR == 16) {
      if (p.l1_gt != nullptr) {       // (N == 16, NHWC, slope 1: host rule) the loss instead of the store
        l1_sum += l1_row_epilogue<FM>(p, acc, b, q, lane, sh);
        continue;
      }
    }
    if (n < p.N) {
#pragma unroll
      for (int mi = 0; mi < FM; ++mi)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int x = mi * FR + (FR == 32 ? (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) : (lane >> 4) * 4 + e);
          float v = acc[mi][e] + sh;
          v = v > 0.f ? v : v * p.slope;
          if (p.out_mode == M2H_OUT_NHWC) {
            p.dst[((size_t)(b * p.Ho + q) * p.Wo + x) * p.ldc + n] = v;
          } else {
            const size_t out = (size_t)b * 16 * plane + (size_t)q * p.Wo + x;
            p.dst[(out + (size_t)(n & 15) * plane) * Cc + (n >> 4)] = v;
          }
        }
    }
  }
  if constexpr (FR == 16) {
    if (p.l1_gt != nullptr) l1_block_partial(p, l1_sum, Pl, tid);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The image-row 3x3 convolution in bf16x3 arithmetic (three bf16 MFMAs per fp32 product: lo*hi + hi*lo + hi*hi, fp32 accumulate --
// the engine's split-product mode, M2H_MATH_BF16X3).  Same walk as conv3x3_row_kernel: a block keeps the weight matrix in LDS and
// stages four image rows + halo as a zero-padded 6 x 34-pixel patch; here both are SPLIT on the way into LDS -- a pixel's (a
// weight row's) k-values as [hi bf16 x K | lo bf16 x K] -- so a 16-byte fragment read is eight consecutive channels of one pixel:
// one operand of v_mfma_f32_32x32x16_bf16 (N <= 32) / v_mfma_f32_16x16x32_bf16 (N <= 16).  At the 1.7 M pixels of an update_sep
// epoch the fp32-MFMA kernel is matrix-bound (31.7 GFLOP of 16-pass fp32 MFMAs: 285 us at 71 % of the 157 TFLOP/s peak); the
// three bf16 MFMAs cost 3/16 of that, which leaves the layer to its HBM stream (220 MB in + 220 MB out).
// De-sliced store with N = 16: a lane's four accumulator values are four consecutive time frames of one band: one 16-byte store.
// blocks per CU by LDS: 32 -> 32 channels 66.8 KB (2), 32 -> 16 48.1 KB (3), 16 -> 32 35.3 KB (4, held at 3: the register budget of three)
template <int FR, int C>
__global__ __launch_bounds__(256, (FR == 32 && C == 32) ? 2 : 3) void conv3x3_row_bf16x3_kernel(const IGemmP p) {
  constexpr int W = 32, PW = W + 2, ROWS = 4, PR = ROWS + 2;
  constexpr int K = 9 * C;
  constexpr int PS = 4 * C + 16;                    // patch pixel stride, bytes ([hi C | lo C] + 16: an odd count of 16-byte units)
  constexpr int WS = 4 * K + 16;                    // weight row stride, bytes
  constexpr int KI = FR == 32 ? 16 : 32;            // k per MFMA
  constexpr int NG = C / KI;                        // MFMAs (x3) per tap
  constexpr int FM = 32 / FR;                       // pixel fragments per wave (one image row)
  constexpr int NE = FR == 32 ? 16 : 4;
  constexpr int SEG = C / 4;                        // 16-byte fp32 segments per pixel
  constexpr int NPL = (PR * PW * SEG + 255) / 256;  // patch loads per thread
  using AccT = typename std::conditional<FR == 32, f32x16, f32x4>::type;
  static_assert(NG >= 1 && C % KI == 0 && (PS / 16) % 2 == 1 && (WS / 16) % 2 == 1, "tile shape");
  __shared__ __attribute__((aligned(16))) char Wl[FR * WS];
  __shared__ __attribute__((aligned(16))) char Pl[PR * PW * PS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & (FR - 1), kq = lane / FR;
  const int chunks = p.B * (p.Hq / ROWS);

  auto split_store = [&](char* dst_hi, int lo_off, f32x4 v) {
    const bf16x4 hi = __builtin_convertvector(v, bf16x4);
    const f32x4 hf = __builtin_convertvector(hi, f32x4);
    const bf16x4 lo = __builtin_convertvector(v - hf, bf16x4);
    *reinterpret_cast<bf16x4*>(dst_hi) = hi;
    *reinterpret_cast<bf16x4*>(dst_hi + lo_off) = lo;
  };
  // weights [N][K] fp32 -> split LDS rows (rows past N: zeros), once per block
  for (int i = tid; i < FR * (K / 4); i += 256) {
    const int n = i / (K / 4), s4 = i - n * (K / 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < p.N) v = *reinterpret_cast<const f32x4*>(p.w + (size_t)n * K + s4 * 4);
    split_store(Wl + n * WS + s4 * 8, 2 * K, v);
  }
  int shift[9];                                     // patch offset of tap t relative to the output pixel's own patch position
#pragma unroll
  for (int t = 0; t < 9; ++t) shift[t] = (p.offh + (t / 3) * p.mulh) * PW + (p.offw + (t % 3) * p.mulw);

  f32x4 rp[NPL];
  unsigned okm = 0;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto load_chunk = [&](int c) {
    const int b = c / (p.Hq / ROWS), q0 = (c - b * (p.Hq / ROWS)) * ROWS;
    okm = 0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int i = tid + 256 * j;
      const int l = i / SEG, seg = i - l * SEG;
      const int pr = l / PW, pc = l - pr * PW;
      const int ih = q0 + pr - 1, iw = pc - 1;
      const bool ok = i < PR * PW * SEG && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)W;
      const size_t off = ok ? ((size_t)(b * p.Hi + ih) * W + iw) * C + seg * 4 : (size_t)0;
      rp[j] = *reinterpret_cast<const f32x4*>(p.src0 + off);
      okm |= ok ? (1u << j) : 0u;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int i = tid + 256 * j;
      if (i < PR * PW * SEG) split_store(Pl + (i / SEG) * PS + (i % SEG) * 8, 2 * C, (okm & (1u << j)) ? rp[j] : zero4);
    }
  };
  auto mma = [&](const f32x4& a, const f32x4& b, AccT& c) {
    if constexpr (FR == 32)
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };
  const size_t plane = (size_t)p.Ho * p.Wo;
  const int Cc = p.N >> 4;
  float l1_sum = 0.f;
  for (int c = blockIdx.x; c < chunks; c += gridDim.x) {
    if (c == (int)blockIdx.x) load_chunk(c);
    __syncthreads();              // the previous chunk's fragment reads (and the weight stores) are done
    store_chunk();
    __syncthreads();
    if (c + (int)gridDim.x < chunks) load_chunk(c + gridDim.x);   // next chunk's loads fly under this chunk's MFMAs
    AccT acc[FM];
#pragma unroll
    for (int mi = 0; mi < FM; ++mi)
#pragma unroll
      for (int e = 0; e < NE; ++e) acc[mi][e] = 0.f;
    const int prow0 = (wave + 1) * PW + 1;          // this wave's image row inside the patch, column 0
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const char* wp = Wl + frow * WS + (t * C + g * KI) * 2 + kq * 16;
        const f32x4 bh = *reinterpret_cast<const f32x4*>(wp), bl = *reinterpret_cast<const f32x4*>(wp + 2 * K);
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) {
          const char* ap = Pl + (prow0 + mi * FR + frow + shift[t]) * PS + g * KI * 2 + kq * 16;
          const f32x4 ah = *reinterpret_cast<const f32x4*>(ap), al = *reinterpret_cast<const f32x4*>(ap + 2 * C);
          mma(al, bh, acc[mi]);
          mma(ah, bl, acc[mi]);
          mma(ah, bh, acc[mi]);
        }
      }
    }
    // epilogue: rows = pixels of image row (q0 + wave), columns = output channels
    const int b = c / (p.Hq / ROWS), q = (c - b * (p.Hq / ROWS)) * ROWS + wave;
    const int n = lane & (FR - 1);
    const float sh = (p.shift != nullptr && n < p.N) ? p.shift[n] : 0.f;
    if constexpr (FR == 16) {
      if (p.l1_gt != nullptr) {       // (N == 16, NHWC, slope 1: host rule) the loss instead of the store
        l1_sum += l1_row_epilogue<FM>(p, acc, b, q, lane, sh);
        continue;
      }
    }
    if (n < p.N) {
#pragma unroll
      for (int mi = 0; mi < FM; ++mi) {
        if constexpr (FR == 16) {
          if (p.out_mode != M2H_OUT_NHWC && Cc == 1) {   // four consecutive frames of band n
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = acc[mi][e] + sh;
              v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
            }
            *reinterpret_cast<f32x4*>(p.dst + (size_t)b * 16 * plane + (size_t)n * plane + (size_t)q * p.Wo + mi * FR + (lane >> 4) * 4) = v;
            continue;
          }
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int x = mi * FR + (FR == 32 ? (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) : (lane >> 4) * 4 + e);
          float v = acc[mi][e] + sh;
          v = v > 0.f ? v : v * p.slope;
          if (p.out_mode == M2H_OUT_NHWC) {
            p.dst[((size_t)(b * p.Ho + q) * p.Wo + x) * p.ldc + n] = v;
          } else {
            const size_t out = (size_t)b * 16 * plane + (size_t)q * p.Wo + x;
            p.dst[(out + (size_t)(n & 15) * plane) * Cc + (n >> 4)] = v;
          }
        }
      }
    }
  }
  if constexpr (FR == 16) {
    if (p.l1_gt != nullptr) l1_block_partial(p, l1_sum, reinterpret_cast<float*>(Pl), tid);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Skinny dense GEMM, M <= 16 rows (fp32 MFMA): D[m][n] = act(scale[n] * sum_k X[m][k] W[n][k] + shift[n]) where every GEMM row is
// one contiguous run of floats -- nn.Linear at the rollout width (the GRU's input projection, 1536 x 1536), the full-spatial
// "conv as Linear" of VisualCNN / AudioCNN (visual_cnn.py:140-141: 4608 -> 512), and the two U-Net stages around the 1 x 1
// bottleneck at the rollout batch: the deepest encoder conv (its tap window covers the whole 2 x 2 input: the sample IS the row)
// and the first transposed conv (one tap per sub-pixel phase).  These are weight streams (4-17 MB against 14 rows): the tiled
// engine needs split-K slabs and a reduce launch to occupy the chip (20-28 us per layer).  Here a block owns COLS output
// channels of one phase; BOTH operands go straight from global memory into v_mfma_f32_16x16x4_f32 registers (lane (row,
// k-quarter) loads 16 bytes of its row: four consecutive MFMAs' worth; X is a few hundred KB and stays in L2), the four waves
// split K, and their partial tiles meet through 4 KB of LDS in wave order.  No LDS staging, no barrier in the k-loop.
// COLS = 16 fills the MFMA tile; COLS = 4 (the other columns repeat the last row) quadruples the block count for N <= 512.
// K is walked as thn segments of twn*Ctot floats: X contiguous, W at tap (th0 + seg, tw0) of its (nth x ntw x Ctot) row.
// NW = waves per block (4, 8 or 16): they split the walked reduction, so a long K over few blocks is a short chain per wave.
template <int COLS, int NW>
__global__ __launch_bounds__(64 * NW) void skinny_rows_kernel(const IGemmP p) {
  __shared__ float R[NW][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int nblk = (p.N + COLS - 1) / COLS;
  const int phase = blockIdx.x / nblk;
  const int n0 = (blockIdx.x - phase * nblk) * COLS;
  const int L = p.twn * p.Ctot;                               // floats per segment
  const int sps = L >> 4;                                     // 16-float steps per segment
  const int steps = p.thn * sps;
  const int s0 = (steps * wave) / NW, s1 = (steps * (wave + 1)) / NW;
  const float* xr = p.src0 + (size_t)min(i, p.M - 1) * ((size_t)p.thn * L) + 4 * kq;   // rows past M re-read row M-1 (never stored)
  const float* wr = p.w + ((size_t)phase * p.N + min(n0 + min(i, COLS - 1), p.N - 1)) * p.K + (size_t)p.tw0 * p.Ctot + 4 * kq;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int seg = s0 / sps, s = s0 - seg * sps;
#pragma unroll 8
  for (int t = s0; t < s1; ++t) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(xr + (size_t)seg * L + 16 * s);
    const f32x4 b = *reinterpret_cast<const f32x4*>(wr + (size_t)(p.th0 + seg) * p.ntw * p.Ctot + 16 * s);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
    if (++s == sps) {
      s = 0;
      ++seg;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) R[wave][kq * 4 + e][i] = acc[e];   // D[m = kq*4 + e][column i]
  __syncthreads();
  const int m = tid >> 4, c = tid & 15, n = n0 + c;
  if (tid < 256 && m < p.M && c < COLS && n < p.N) {
    float x = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) x += R[w][m][c];              // wave order
    const float sc = p.scale != nullptr ? p.scale[n] : 1.f;
    const float sh = p.shift != nullptr ? p.shift[n] : 0.f;
    x = x * sc + sh;
    const size_t pix = p.convT ? ((size_t)m * p.Ho + (phase >> 1)) * p.Wo + (phase & 1) : (size_t)m;
    p.dst[pix * p.ldc + n] = x > 0.f ? x : x * p.slope;
  }
}

// Skinny implicit-GEMM conv for small pixel counts (M <= 1024 rows per phase: the U-Net's deeper stages at the rollout batch, the
// policy's Linear layers over a 280-sample update batch), fp32
// MFMA, no LDS staging: a block computes a 16*MGB (pixels) x 16 (channels) tile of one phase; lane (row i, k-quarter) loads 16 bytes
// of its weight row and of each of its MGB pixel rows (gathered per tap exactly as the engine above does, zero outside the image;
// the activations are a few hundred KB and stay in L1 / L2) straight into v_mfma_f32_16x16x4_f32 registers; the four waves split
// the walked reduction (tap window x both sources x channels) and meet through LDS in wave order; BN scale / shift, activation
// and the NHWC store follow.  The tiled engine occupies the chip at these sizes only through split-K (slabs + a reduce launch,
// 24-45 us per layer against 2-17 MB of weights); here the weights are streamed MG/MGB times and the activations N/16 times.
// NCG = 16-column groups per block (1; 2 for the update batch's wide Linear layers: the activations' share of the L2 -> CU stream, one pass per
// column block, halves)
template <int MGB, int NW, int NCG = 1>
__global__ __launch_bounds__(64 * NW) void skinny_gather_kernel(const IGemmP p) {
  __shared__ float R[NW][MGB * NCG][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int NB = (p.N + 16 * NCG - 1) / (16 * NCG), MS = (p.MT + MGB - 1) / MGB;     // p.MT = 16-row groups of M
  int L = blockIdx.x;
  const int ms = L % MS;
  L /= MS;
  const int nb = L % NB, phase = L / NB;
  int mulh = p.mulh, offh = p.offh, mulw = p.mulw, offw = p.offw, ph = p.ph, pw = p.pw;
  const float* wbase = p.w;
  if (p.convT) {
    ph = phase >> 1;
    pw = phase & 1;
    mulh = 2 * ph - 1;
    mulw = 2 * pw - 1;
    offh = 0;
    offw = 0;
    wbase += (size_t)phase * p.N * p.K;
  }
  int qh[MGB], rw[MGB], bpix[MGB];
#pragma unroll
  for (int g = 0; g < MGB; ++g) {
    const int m = (ms * MGB + g) * 16 + i;
    qh[g] = rw[g] = -(1 << 24);
    bpix[g] = 0;
    if (m < p.M) {
      int q, rr, b, out, bc;
      decode_row(p, m, ph, pw, q, rr, b, out, bc);
      qh[g] = q * p.stride + offh;
      rw[g] = rr * p.stride + offw;
      bpix[g] = b * p.Hi * p.Wi;
    }
  }
  const float* wrow[NCG];
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg) wrow[cg] = wbase + (size_t)min((nb * NCG + cg) * 16 + i, p.N - 1) * p.K + 4 * kq;
  const int spt = p.Ctot >> 4;                                   // 16-float steps per tap
  const int steps = p.thn * p.twn * spt;
  const int s0 = (steps * wave) / NW, s1 = (steps * (wave + 1)) / NW;
  f32x4 acc[MGB][NCG];
#pragma unroll
  for (int g = 0; g < MGB; ++g)
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) acc[g][cg] = {0.f, 0.f, 0.f, 0.f};
  int tap = s0 / spt, ci = (s0 - tap * spt) * 16;
  int th = p.th0 + tap / p.twn, tw = p.tw0 + tap % p.twn;
  unsigned offA[MGB];                                            // float offset of the row's pixel at the current tap, per source stride
  bool okA[MGB];
  auto at_tap = [&]() {
#pragma unroll
    for (int g = 0; g < MGB; ++g) {
      const int ih = qh[g] + th * mulh, iw = rw[g] + tw * mulw;
      okA[g] = (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
      offA[g] = okA[g] ? (unsigned)(bpix[g] + ih * p.Wi + iw) : 0u;
    }
  };
  at_tap();
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int t = s0; t < s1;) {
    // one run of steps inside the current (tap, source): addresses advance by 16 floats, no branches -> the loads of the next
    // steps are issued under the MFMAs of the current ones
    const bool second = ci >= p.C0;
    const int seg_end = second ? p.Ctot : p.C0;                  // end of this source's channels
    const int nrun = min(s1 - t, (seg_end - ci) >> 4);
    const float* src = second ? p.src1 : p.src0;
    const unsigned Cs = second ? p.C1 : p.C0, c = (second ? ci - p.C0 : ci) + 4 * kq;
    const size_t wofs = (size_t)(th * p.ntw + tw) * p.Ctot + ci;
    const float* ap[MGB];
#pragma unroll
    for (int g = 0; g < MGB; ++g) ap[g] = src + (size_t)offA[g] * Cs + c;   // (rows outside the image: pixel 0, masked below)
#pragma unroll 4
    for (int k = 0; k < nrun; ++k) {
      f32x4 b[NCG];
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) b[cg] = *reinterpret_cast<const f32x4*>(wrow[cg] + wofs + 16 * k);
      f32x4 a[MGB];
#pragma unroll
      for (int g = 0; g < MGB; ++g) a[g] = *reinterpret_cast<const f32x4*>(ap[g] + 16 * k);
#pragma unroll
      for (int g = 0; g < MGB; ++g) {
        const f32x4 av = okA[g] ? a[g] : zero4;
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[g][cg] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], b[cg][j], acc[g][cg], 0, 0, 0);
      }
    }
    t += nrun;
    ci += 16 * nrun;
    if (ci == p.Ctot) {
      ci = 0;
      if (++tw == p.tw0 + p.twn) {
        tw = p.tw0;
        ++th;
      }
      at_tap();
    }
  }
#pragma unroll
  for (int g = 0; g < MGB; ++g)
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
      for (int e = 0; e < 4; ++e) R[wave][g * NCG + cg][kq * 4 + e][i] = acc[g][cg][e];    // D[row kq*4 + e][channel i]
  __syncthreads();
#pragma unroll
  for (int gc = 0; gc < MGB * NCG; ++gc) {
    const int g = gc / NCG, cg = gc % NCG;
    const int r16 = tid >> 4, c16 = tid & 15;
    const int m = (ms * MGB + g) * 16 + r16, n = (nb * NCG + cg) * 16 + c16;
    if (tid < 256 && m < p.M && n < p.N) {
      float x = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) x += R[w][gc][r16][c16];       // wave order
      int q, rr, b, out, bc;
      decode_row(p, m, ph, pw, q, rr, b, out, bc);
      if (p.cls_table != nullptr) x += p.cls_val[bc >> 4] * p.cls_table[(size_t)(bc & 15) * p.N + n];   // the class plane (as the tiled engine's epilogues)
      const float sc = p.scale != nullptr ? p.scale[n] : 1.f;
      const float sh = p.shift != nullptr ? p.shift[n] : 0.f;
      x = x * sc + sh;
      p.dst[(size_t)out * p.ldc + n] = x > 0.f ? x : x * p.slope;
    }
  }
}

// Split-K epilogue: sums the S partial slabs of one output element in a fixed order (deterministic) and applies the
// same fused epilogue as the main kernel.  One thread = one GEMM row x 4 consecutive channels.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const IGemmP p) {
  const int N4 = p.N >> 2;
  const long total = (long)p.M * N4;
  const int phase = blockIdx.y;
  const int ph = p.convT ? (phase >> 1) : p.ph, pw = p.convT ? (phase & 1) : p.pw;
  const size_t plane = (size_t)p.Ho * p.Wo;
  const int Cc = p.N >> 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / N4);
    const int n = (int)(i - (long)m * N4) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.S; ++s) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(p.ws + ((size_t)(phase * p.S + s) * p.M + m) * p.N + n);
      v += t;
    }
    int q, rr, b, out, bc;
    decode_row(p, m, ph, pw, q, rr, b, out, bc);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = v[j];
      if (p.cls_table != nullptr) x += p.cls_val[bc >> 4] * p.cls_table[(size_t)(bc & 15) * p.N + n + j];
      const float sc = p.scale != nullptr ? p.scale[n + j] : 1.f;
      const float sh = p.shift != nullptr ? p.shift[n + j] : 0.f;
      x = x * sc + sh;
      v[j] = x > 0.f ? x : x * p.slope;
    }
    if (p.out_mode == M2H_OUT_NHWC) {
      if (p.dst_split) {
        const bf16x4 hi = __builtin_convertvector(v, bf16x4);
        const f32x4 hf = __builtin_convertvector(hi, f32x4);
        const bf16x4 lo = __builtin_convertvector(v - hf, bf16x4);
        char* base = reinterpret_cast<char*>(p.dst + (size_t)out * p.ldc + (n & ~31)) + ((n & 31) >> 2) * 8;
        *reinterpret_cast<bf16x4*>(base) = hi;
        *reinterpret_cast<bf16x4*>(base + 64) = lo;
      } else {
        *reinterpret_cast<f32x4*>(p.dst + (size_t)out * p.ldc + n) = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = (n + j) >> 4, s = (n + j) & 15;
        p.dst[((size_t)out + (size_t)s * plane) * Cc + c] = v[j];
      }
    }
  }
}

// Debug/tuning knobs (m2h_tuning_set): 0 = automatic.
// (tuning knob g_force_splitk: thread-local, m2h_internal.h) >0: force this split-K factor (when workspace allows), -1: never split
// (tuning knob g_force_stages: thread-local, m2h_internal.h) 1 | 2: force the LDS stage count of the narrow-N configs
// (tuning knob g_wide_stages: thread-local, m2h_internal.h) 1 | 2: LDS stage count of the 128x128 config (0 = 2)
// (tuning knob g_skinny: thread-local, m2h_internal.h) -1: never use the 32/64-row tiles
// (tuning knob g_extra_lds: thread-local, m2h_internal.h) tuning experiment: dynamic LDS bytes added to every launch (lowers blocks/CU)
// (tuning knob g_phase_major: thread-local, m2h_internal.h) -1: transposed-conv phases as grid z (four passes over the input) instead of interleaved
static constexpr int M2H_FMT_LAYOUT_BITS = M2H_FMT_SRC_SPLIT | M2H_FMT_W_SPLIT | M2H_FMT_DST_SPLIT;   // operand_format minus the M2H_FMT_MATH_* bits
thread_local int tl_math_mode = 0;   // m2h_set_math_mode: the calling thread's arithmetic (0 fp32 MFMA, 1 bf16x3 split products)
std::atomic<long long> g_launch_count{0};   // M2H_LAUNCH (m2h_internal.h)
thread_local int tl_hi_only = 0;     // m2h_set_math_mode(M2H_MATH_BF16): tl_math_mode = 1 for every dispatch decision, and the engines listed in m2h.h drop the two cross products
// (tuning knob g_tapshare: thread-local, m2h_internal.h) -1: never use the tap-sharing transposed-conv kernel
// (tuning knob g_tap_bm: thread-local, m2h_internal.h) 128: 128-output tiles only in the tap-sharing kernel
// (tuning knob g_fast_loader: thread-local, m2h_internal.h) -1: always use the generic (per-lane k decode) loader
// (tuning knob g_narrow16: thread-local, m2h_internal.h) -1: never use the 16-wide (v_mfma_f32_16x16x4_f32) tile for N <= 16
// (tuning knob g_skinny_linear: thread-local, m2h_internal.h) -1: never use the skinny dense kernel for M <= 16
// (tuning knob g_skinny_gather: thread-local, m2h_internal.h) -1: never use the skinny gather kernel for 16 < M <= 256
// (tuning knob g_row3x3: thread-local, m2h_internal.h) -1: never use the image-row 3x3 kernel
// (tuning knob g_tap_window: thread-local, m2h_internal.h) -1: walk every tap even where a whole kernel row / column lies in the zero padding

// Tap window (see IGemmP): the contiguous range of kernel rows / columns that reach inside the image for at least one output
// pixel.  conv: ih = q*stride + off + t*mul, q in [0, Q); transposed conv: both sub-pixel phases (mul = -1, +1, off = 0) must agree.
static void tap_range(int ntaps, int Q, int stride, int off, int mul, int extent, int& t0, int& tn) {
  int lo = ntaps, hi = -1;
  for (int t = 0; t < ntaps; ++t) {
    bool any = false;
    for (int q = 0; q < Q && !any; ++q) {
      const int i = q * stride + off + t * mul;
      any = i >= 0 && i < extent;
    }
    if (any) {
      lo = t < lo ? t : lo;
      hi = t > hi ? t : hi;
    }
  }
  if (hi < 0) { t0 = 0; tn = ntaps; return; }   // nothing reaches the image: keep the full walk (all-zero result either way)
  t0 = lo;
  tn = hi - lo + 1;
}

static void tap_window(const m2h_conv_args& a, int& th0, int& thn, int& tw0, int& twn) {
  th0 = 0; thn = a.nth; tw0 = 0; twn = a.ntw;
  if (a.Hq > 64 || a.Wq > 4096 || g_tap_window < 0) return;   // large images: every tap is reached, skip the scan
  if (a.conv_transpose) {
    int a0, an, b0, bn;
    tap_range(a.nth, a.Hq, 1, 0, -1, a.Hi, a0, an);
    tap_range(a.nth, a.Hq, 1, 0, +1, a.Hi, b0, bn);
    if (a0 == b0 && an == bn) { th0 = a0; thn = an; }
    tap_range(a.ntw, a.Wq, 1, 0, -1, a.Wi, a0, an);
    tap_range(a.ntw, a.Wq, 1, 0, +1, a.Wi, b0, bn);
    if (a0 == b0 && an == bn) { tw0 = a0; twn = an; }
  } else {
    tap_range(a.nth, a.Hq, a.stride, a.offh, a.mulh, a.Hi, th0, thn);
    tap_range(a.ntw, a.Wq, a.stride, a.offw, a.mulw, a.Wi, tw0, twn);
  }
}

// Reduction length the launch will walk: the tap window applies to the scalar-decode loader only.
static int walked_K(const m2h_conv_args& a) {
  const int Ctot = a.C0 + a.C1;
  const bool fast = g_fast_loader >= 0 && a.C0 % BK == 0 && a.C1 % BK == 0 && a.C0 > 0;
  if (!fast) return a.nth * a.ntw * Ctot;
  int th0, thn, tw0, twn;
  tap_window(a, th0, thn, tw0, twn);
  return thn * twn * Ctot;
}

// waves per block of the skinny kernels: keep a wave's chain of 16-float steps at about 16
// waves per block of the skinny kernels: they split the walked reduction, and a wave's share is a chain of dependent load rounds
// (runs of at most Ctot / 16 steps between tap changes), so short shares win: more than 32 steps -> 16 waves, more than 16 -> 8
// (A/B on one box, tools/train_ab.sh: rollout 74.2 -> 71.6 ms per cycle against the round-2 thresholds 160 / 80)
static int skinny_waves(int steps) { return steps > 32 ? 16 : (steps > 16 ? 8 : 4); }

// Tile choice: N picks the width; skinny M (rollout batches, GRU steps: weight-streaming bound, nothing to re-use along M)
// gets 32- or 64-row tiles so that four times as many blocks stream the weights.
static void pick_tile(long M, int N, int& BM, int& BN) {
  BN = N > 64 ? 128 : (N > 32 ? 64 : (N > 16 ? 32 : 16));
  BM = 128;
  if (BN == 128 && g_skinny >= 0) {
    if (M <= 32) BM = 32;
    else if (M <= 64) BM = 64;
  }
}

static int splitk_for(long M, int N, int K, int phases, int BM, int BN) {
  if ((N & 3) != 0) return 1;
  const int nk = (K + BK - 1) / BK;
  const long mt = (M + BM - 1) / BM, ntl = (N + BN - 1) / BN;
  const long blocks = mt * ntl * phases;  // working blocks (padding blocks of the XCD map exit at once)
  long S = 1;
  // Fewer blocks than CUs: split K, aiming at two resident blocks per CU.  From one block per CU up the launch is left whole (round 4): at
  // 256-511 tiles a two-way split bought occupancy the layer did not need and paid a slab round trip plus a reduce launch for it
  // (the policy encoders at the 280-sample update batch, tools/enc_fwd_bench.py: conv 4x4/2 32 -> 64 forward 66 -> 50 us, the 3x3
  // conv's input gradient 49 -> 34 us).
  if (blocks < 256) S = (512 + blocks - 1) / blocks;
  // ... but keep enough k-tiles per split to amortise a block's fixed cost (row decode, cold first loads, slab write): measured
  // optimum at the rollout shapes (layer_bench --batch 14 --tm 32) is ~4 tiles for the MFMA-paced 128-row tile and ~16 for the
  // weight-streaming 32/64-row tiles (deeper splits made the 14-env U-Net pass 25 % slower)
  const long tmin = BM < 128 ? 16 : 4;
  if (S > nk / tmin) S = nk / tmin;
  return S < 1 ? 1 : (int)S;
}

int choose_splitk(const IGemmP& p, int BM, int BN, size_t ws_bytes) {
  if (p.ws == nullptr || g_force_splitk < 0 || (p.N & 3) != 0) return 1;
  const int phases = p.convT ? 4 : 1;
  const int Kw = (g_fast_loader >= 0 && p.fast_ok) ? p.Kw : p.K;
  int S = splitk_for(p.M, p.N, Kw, phases, BM, BN);
  if (g_force_splitk > 0) {
    S = g_force_splitk;
    const int nk = (Kw + BK - 1) / BK;
    if (S > nk / 2) S = nk / 2;
  }
  while (S > 1 && (size_t)phases * S * p.M * p.N * sizeof(float) > ws_bytes) --S;
  return S < 1 ? 1 : S;
}

// (tuning knob g_big_tile: thread-local, m2h_internal.h) -1: never use the 256 x 128 eight-wave tile; > 0: minimum tile count for it (m2h_tuning_set 26)

// 256 x BN tile, 8 waves, two LDS stages, bf16x3 math on scalar-loader shapes only (no split-K: chosen when the tiles fill the chip)
template <int BN>
static int launch_big(IGemmP& p, size_t ws_bytes, hipStream_t st) {
  constexpr int BM = 256;
  (void)ws_bytes;
  p.MT = (p.M + BM - 1) / BM;
  p.NT = (p.N + BN - 1) / BN;
  p.S = 1;
  const long mtpad = ((long)p.MT + 7) / 8 * 8;
  const long nblk = mtpad * p.NT;
  if (nblk * 4 > 0x7fffffffL) return fail(-1, "conv_igemm: grid too large (%ld blocks)", nblk);
  const int phases = p.convT ? 4 : 1;
  p.pmaj = (p.convT && g_phase_major >= 0) ? 1 : 0;
  dim3 grid((unsigned)(p.pmaj ? nblk * 4 : nblk), 1, p.pmaj ? 1 : phases);
  if (p.presplit)
    M2H_LAUNCH((igemm_f32_kernel<BM, BN, 4, 2, 2, 32, 1, 2>), grid, dim3(512), 0, st, p);
  else
    M2H_LAUNCH((igemm_f32_kernel<BM, BN, 4, 2, 2, 32, 1, 1>), grid, dim3(512), 0, st, p);
  return launch_status("igemm_f32<256,128> (eight waves)");
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int FR = 32>
static int launch_cfg(IGemmP& p, size_t ws_bytes, hipStream_t st) {
  const bool fast = g_fast_loader >= 0 && p.fast_ok;
  p.MT = (p.M + BM - 1) / BM;
  p.NT = (p.N + BN - 1) / BN;
  p.S = choose_splitk(p, BM, BN, ws_bytes);
  const long mtpad = ((long)p.MT + 7) / 8 * 8;
  const long nblk = mtpad * p.NT;
  if (nblk > 0x7fffffffL) return fail(-1, "conv_igemm: grid too large (%ld blocks)", nblk);
  const int phases = p.convT ? 4 : 1;
  p.pmaj = (p.convT && g_phase_major >= 0 && nblk * 4 <= 0x7fffffffL) ? 1 : 0;
  dim3 grid((unsigned)(p.pmaj ? nblk * 4 : nblk), (unsigned)p.S, p.pmaj ? 1 : phases);
  const dim3 blk(64 * WM * WN);
  if (fast && p.math == 1 && p.presplit)
    M2H_LAUNCH((igemm_f32_kernel<BM, BN, WM, WN, NSTAGE, FR, 1, 2>), grid, blk, (size_t)g_extra_lds, st, p);
  else if (fast && p.math == 1)
    M2H_LAUNCH((igemm_f32_kernel<BM, BN, WM, WN, NSTAGE, FR, 1, 1>), grid, blk, (size_t)g_extra_lds, st, p);
  else if (fast)
    M2H_LAUNCH((igemm_f32_kernel<BM, BN, WM, WN, NSTAGE, FR, 1>), grid, blk, (size_t)g_extra_lds, st, p);
  else
    M2H_LAUNCH((igemm_f32_kernel<BM, BN, WM, WN, NSTAGE, FR, 0>), grid, blk, (size_t)g_extra_lds, st, p);
  static const std::string label = "igemm_f32<" + std::to_string(BM) + "," + std::to_string(BN) + ">";   // one per instantiation
  int rc = launch_status(label.c_str());
  if (rc != 0 || p.S == 1) return rc;
  const long total = (long)p.M * (p.N >> 2);
  long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  M2H_LAUNCH(splitk_epilogue_kernel, dim3((unsigned)g, phases), dim3(256), 0, st, p);
  static const std::string label_sk = label + " + split-K reduce";
  rc = launch_status("conv_igemm_f32 split-K epilogue");
  tl_last_launch = label_sk.c_str();
  return rc;
}

#ifdef M2H_CLOCK_DIAG
extern "C" int m2h_diag_read_clocks(unsigned long long* host_out, int nblocks) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_clock_dbg), (size_t)nblocks * 6 * sizeof(unsigned long long));
}
#endif

// the ordered reduce + epilogue over the split-K slabs of an LDS-DMA / shared-patch launch (p.S slabs per phase)
static int launch_splitk_reduce(const IGemmP& p, hipStream_t st) {
  const long total = (long)p.M * (p.N >> 2);
  long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  M2H_LAUNCH(splitk_epilogue_kernel, dim3((unsigned)g, p.convT ? 4 : 1), dim3(256), 0, st, p);
  return launch_status("conv_igemm_f32 split-K epilogue");
}

size_t conv_igemm_workspace_bytes(const m2h_conv_args& a) {
  // the exact split-K scratch of the automatic choice for these arguments: phases * S * M * N floats
  const long M = (long)a.B * a.Hq * a.Wq;
  const int phases = a.conv_transpose ? 4 : 1;
  const int K = walked_K(a);
  int BM, BN;
  pick_tile(M, a.N, BM, BN);
  const int S = splitk_for(M, a.N, K, phases, BM, BN);
  size_t bytes = S <= 1 ? 0 : (size_t)phases * S * M * a.N * sizeof(float);
  // split32 operands in bf16x3 math: the LDS-DMA engine's two-K-halves launch of the 256 x 128 tile (conv_dma.hip) takes its slabs
  // from this workspace too -- report them, so that a caller who sizes the workspace by this function gets the same kernel (and the
  // same fp32 summation order) as the whole-network runner, whose scratch is the maximum over its stages
  const int both = M2H_FMT_SRC_SPLIT | M2H_FMT_W_SPLIT;
  const int math = (a.operand_format & M2H_FMT_MATH_BF16X3) ? 1 : (a.operand_format & M2H_FMT_MATH_FP32) ? 0 : tl_math_mode;
  if (math == 1 && (a.operand_format & both) == both && a.head_w == nullptr && a.C0 % BK == 0 && a.C1 % BK == 0 && M > 64 && g_force_splitk <= 0 &&
      g_fast_loader >= 0) {
    const size_t need = (size_t)phases * 2 * M * a.N * sizeof(float);
    if (dma_split2_rule(M, a.N, K, phases, true, need)) {
      if (need > bytes) bytes = need;
    } else {
      const int Sd = dma_deep_split(M, a.N, K, phases);   // ... and its S K-parts launch on the deepest stages
      const size_t needd = (size_t)phases * Sd * M * a.N * sizeof(float);
      if (Sd > 1 && needd > bytes) bytes = needd;
    }
  }
  return bytes;
}

// l1: optional fused L1 loss (m2h_conv3x3_l1_nhwc16): honoured by the image-row 3x3 kernels' 16-channel instantiations only -- any other
// dispatch is an error, never a silent plain conv
int conv_igemm_f32(const m2h_conv_args& a, hipStream_t st, const ConvL1* l1) {
  M2H_REQUIRE(a.src0 != nullptr && a.wp != nullptr && a.dst != nullptr, "conv_igemm: null pointer");
  M2H_REQUIRE(a.B > 0 && a.Hi > 0 && a.Wi > 0 && a.Hq > 0 && a.Wq > 0 && a.N > 0, "conv_igemm: non-positive size");
  M2H_REQUIRE(a.C0 > 0 && a.C0 % 4 == 0 && a.C1 >= 0 && a.C1 % 4 == 0, "conv_igemm: C0/C1 must be multiples of 4 (got %d, %d)", a.C0, a.C1);
  M2H_REQUIRE((a.C1 == 0) == (a.src1 == nullptr), "conv_igemm: src1/C1 mismatch");
  M2H_REQUIRE(a.nth > 0 && a.ntw > 0 && a.stride > 0 && a.os > 0, "conv_igemm: bad taps/stride");
  M2H_REQUIRE(a.Ho > 0 && a.Wo > 0, "conv_igemm: bad output size");
  if (a.conv_transpose) {
    M2H_REQUIRE(a.nth == 2 && a.ntw == 2 && a.stride == 1 && a.os == 2, "conv_igemm: transposed conv is 4x4/s2/p1 (2x2 taps per phase)");
    M2H_REQUIRE(a.Hq == a.Hi && a.Wq == a.Wi && a.Ho == 2 * a.Hi && a.Wo == 2 * a.Wi, "conv_igemm: transposed conv geometry");
  } else {
    M2H_REQUIRE((a.Hq - 1) * a.os + a.ph < a.Ho && (a.Wq - 1) * a.os + a.pw < a.Wo, "conv_igemm: output pixel grid exceeds Ho x Wo");
  }
  const long M = (long)a.B * a.Hq * a.Wq;
  M2H_REQUIRE(M < (1L << 30), "conv_igemm: M too large");
  M2H_REQUIRE((long)a.B * a.Hi * a.Wi < (1L << 30), "conv_igemm: input pixel count too large");
  M2H_REQUIRE((long)a.B * 16 * a.Ho * a.Wo < (1L << 31), "conv_igemm: output pixel count too large");
  M2H_REQUIRE((long)a.B * a.Hi * a.Wi * (a.C0 > a.C1 ? a.C0 : a.C1) < (1L << 32), "conv_igemm: source tensor exceeds 32-bit element offsets");
  M2H_REQUIRE((long)a.N * a.nth * a.ntw * (a.C0 + a.C1) < (1L << 32), "conv_igemm: weight matrix exceeds 32-bit element offsets");
  if (a.out_mode == M2H_OUT_DESLICE) {
    M2H_REQUIRE(a.N % 16 == 0, "conv_igemm: de-slice needs N %% 16 == 0");
  } else {
    M2H_REQUIRE(a.out_mode == M2H_OUT_NHWC && a.ldc >= a.N, "conv_igemm: bad out_mode/ldc");
  }
  M2H_REQUIRE((a.cls_table == nullptr) == (a.cls_val == nullptr), "conv_igemm: cls_table/cls_val mismatch");

  IGemmP p;
  p.src0 = a.src0; p.src1 = a.src1; p.C0 = a.C0; p.C1 = a.C1; p.Ctot = a.C0 + a.C1;
  p.B = a.B; p.Hi = a.Hi; p.Wi = a.Wi; p.Hq = a.Hq; p.Wq = a.Wq; p.stride = a.stride;
  p.wq_sh = (a.Wq & (a.Wq - 1)) == 0 ? __builtin_ctz((unsigned)a.Wq) : -1;
  p.hq_sh = (a.Hq & (a.Hq - 1)) == 0 ? __builtin_ctz((unsigned)a.Hq) : -1;
  p.ntw = a.ntw; p.ntap = a.nth * a.ntw;
  p.mulh = a.mulh; p.offh = a.offh; p.mulw = a.mulw; p.offw = a.offw; p.convT = a.conv_transpose ? 1 : 0;
  p.w = a.wp; p.N = a.N; p.K = p.ntap * p.Ctot;
  p.scale = a.scale; p.shift = a.shift; p.slope = a.slope; p.cls_table = a.cls_table; p.cls_val = a.cls_val;
  p.head_w = a.head_w; p.head_b = a.head_b;
  {
    const int fmt = a.operand_format;
    M2H_REQUIRE((fmt & (M2H_FMT_MATH_BF16X3 | M2H_FMT_MATH_FP32)) != (M2H_FMT_MATH_BF16X3 | M2H_FMT_MATH_FP32),
                "conv_igemm: operand_format names both arithmetic modes");
    p.math = (fmt & M2H_FMT_MATH_BF16X3) ? 1 : (fmt & M2H_FMT_MATH_FP32) ? 0 : tl_math_mode;
    p.hi_only = (p.math == 1 && tl_hi_only) ? 1 : 0;
    M2H_REQUIRE((fmt & (M2H_FMT_SRC_SPLIT | M2H_FMT_W_SPLIT | M2H_FMT_DST_SPLIT)) == 0 || p.math == 1,
                "conv_igemm: split32 operands need the bf16x3 math mode");
    const int both = M2H_FMT_SRC_SPLIT | M2H_FMT_W_SPLIT;
    M2H_REQUIRE((fmt & both) == 0 || (fmt & both) == both, "conv_igemm: sources and weights must be split32 together");
    p.presplit = (fmt & both) == both ? 1 : 0;
    p.dst_split = (fmt & M2H_FMT_DST_SPLIT) ? 1 : 0;
    M2H_REQUIRE(!p.dst_split || (a.out_mode == M2H_OUT_NHWC && a.N % 32 == 0 && a.ldc % 32 == 0 && a.head_w == nullptr),
                "conv_igemm: split32 output needs NHWC, N %% 32 == 0, ldc %% 32 == 0, no fused head");
  }
  if (a.head_w != nullptr) {
    M2H_REQUIRE(a.head_b != nullptr && (a.N == 32 || a.N == 16) && a.out_mode == M2H_OUT_DESLICE && a.workspace == nullptr && a.cls_table == nullptr,
                "conv_igemm: fused head needs N in {16,32}, de-sliced output, no split-K workspace, no class plane");
  }
  p.dst = a.dst; p.Ho = a.Ho; p.Wo = a.Wo; p.os = a.os; p.ph = a.ph; p.pw = a.pw; p.ldc = a.ldc; p.out_mode = a.out_mode;
  p.l1_gt = nullptr; p.l1_part = nullptr; p.l1_inv = 0.f;
  if (l1 != nullptr) {
    M2H_REQUIRE(l1->gt && l1->partials && l1->loss && a.N == 16 && a.ldc == 16 && a.out_mode == M2H_OUT_NHWC && a.slope == 1.f && a.scale == nullptr,
                "conv_igemm: the fused L1 loss needs N = 16 NHWC output without scale or activation");
    p.l1_gt = l1->gt; p.l1_part = l1->partials; p.l1_inv = l1->inv;
  }
  p.M = (int)M;
  {
    const size_t pix = (size_t)a.B * a.Hi * a.Wi;
    const size_t lim = (size_t)1 << 32;
    p.fast_ok = (a.C0 % BK == 0 && a.C1 % BK == 0 && a.C0 > 0 && pix * a.C0 * 4 < lim && pix * (size_t)a.C1 * 4 < lim &&
                 (size_t)a.N * p.K * 4 < lim) ? 1 : 0;
  }
  M2H_REQUIRE(p.K % 4 == 0, "conv_igemm: K must be a multiple of 4");
  tap_window(a, p.th0, p.thn, p.tw0, p.twn);
  p.Kw = p.thn * p.twn * p.Ctot;
  M2H_REQUIRE(!p.presplit || p.fast_ok, "conv_igemm: split32 operands need channel counts that are multiples of 32");

  p.ws = static_cast<float*>(a.workspace);
  const size_t wsb = a.workspace != nullptr ? a.workspace_bytes : 0;
  // split32 operands, 4x4/s2 conv or transposed conv, N a multiple of 64, a chip's worth of tiles: the shared-patch engine (conv_patch.hip)
  if (p.presplit && g_fast_loader >= 0 && g_force_splitk <= 0 && g_phase_major >= 0) {
    const int rc = launch_igemm_patch(p, wsb, st);
    if (rc != -2) {
      if (rc != 0 || p.S == 1) return rc;
      const int rc2 = launch_splitk_reduce(p, st);
      tl_last_launch = "igemm_patch<256,128> + split-K reduce";
      return rc2;
    }
  }
  // narrow transposed convs on split32 operands: all four phases from one staged patch (convt_quad.hip)
  if (p.convT && p.presplit && g_force_splitk <= 0 && g_phase_major >= 0) {
    const int rc = launch_convT_quad(p, st);
    if (rc != -2) return rc;
  }
  // narrow transposed convs in bf16x3 math: the four taps of a phase share one staged input image (convT_tap_kernel)
  if (p.convT && p.math == 1 && g_tapshare >= 0 && p.fast_ok && p.N <= (g_tapshare == 2 ? 32 : 64) && a.Wq >= 32 && 128 % a.Wq == 0 &&
      a.Hq % (128 / a.Wq) == 0 && g_force_splitk <= 0 && g_phase_major >= 0 && M >= 128L * 256) {
    // 256-output tiles when the image geometry and the block count allow (bytes per output: see the kernel)
    // eight-wave blocks (one per CU): 256-output tiles for N = 64 by default (pair_ab, headline pair: 3.392 -> 3.364 ms); the
    // 512-output tiles for N <= 32 measured no gain (3.388 / 3.388) and stay behind m2h_tuning_set 16 = 512 (128 / 256 = the
    // four-wave tiles only)
    const int bm8 = p.N <= 32 ? 512 : 256;
    const bool wave8 = (g_tap_bm == 512 || (g_tap_bm == 0 && p.N > 32)) && bm8 / a.Wq >= 1 && a.Hq % (bm8 / a.Wq) == 0 && M >= (long)bm8 * 512;
    const bool big = !wave8 && g_tap_bm != 128 && p.N <= 32 && a.Hq % (256 / a.Wq) == 0 && M >= 256L * 512;   // N = 64, 4 waves: 93 KB LDS, one block per CU
    const int bm = wave8 ? bm8 : big ? 256 : 128;
    p.MT = (int)((M + bm - 1) / bm);
    p.NT = 1;
    p.S = 1;
    const long nblk = ((long)p.MT + 7) / 8 * 8 * 4;
    // measured (layer_bench, B=256, 512x256, 128-output tiles): N=16 368 -> 308 us, N=64 277 -> 249 us; N=32 no change, so the
    // 32-wide stage uses this kernel only with split32 operands (runner) or when forced
    const int w = (p.N <= 16 && g_narrow16 >= 0) ? 16 : (p.N <= 32 ? 32 : 64);
    if (nblk <= 0x7fffffffL && (w != 32 || g_tapshare > 0 || p.presplit)) {
      const dim3 grid((unsigned)nblk), blk(wave8 ? 512 : 256);
#define M2H_TAP_P(BN_, FR_, PRE_)                                                                                  \
  do {                                                                                                             \
    if (wave8) M2H_LAUNCH((convT_tap_kernel<BN_, FR_, PRE_, (BN_ <= 32 ? 512 : 256), 8>), grid, blk, 0, st, p); \
    else if (big) M2H_LAUNCH((convT_tap_kernel<BN_, FR_, PRE_, 256>), grid, blk, 0, st, p);               \
    else M2H_LAUNCH((convT_tap_kernel<BN_, FR_, PRE_, 128>), grid, blk, 0, st, p);                        \
  } while (0)
#define M2H_TAP(BN_, FR_)                       \
  do {                                          \
    if (p.presplit) M2H_TAP_P(BN_, FR_, 1);     \
    else M2H_TAP_P(BN_, FR_, 0);                \
  } while (0)
      if (w == 16) M2H_TAP(16, 16);
      else if (w == 32) M2H_TAP(32, 32);
      else M2H_TAP(64, 32);
#undef M2H_TAP
#undef M2H_TAP_P
      return launch_status(w == 16 ? "igemm_convT_tap<16>" : (w == 32 ? "igemm_convT_tap<32>" : "igemm_convT_tap<64>"));
    }
  }
  // M <= 16 rows that are each one contiguous run of floats: Linear; a conv whose tap window covers the whole image and gives
  // one output pixel; a transposed conv over a 1 x 1 image (one tap per phase).  Weight streaming on the skinny kernel.
  if (p.math == 0 && g_skinny_linear >= 0 && g_fast_loader >= 0 && p.fast_ok && M <= 16 && a.C1 == 0 && a.Hq == 1 && a.Wq == 1 && a.os >= 1 && a.N % 4 == 0 &&
      a.out_mode == M2H_OUT_NHWC && a.cls_table == nullptr && a.head_w == nullptr && (a.operand_format & M2H_FMT_LAYOUT_BITS) == 0 &&
      (size_t)a.N * p.Kw * (p.convT ? 4 : 1) >= ((size_t)1 << (g_skinny_tiny >= 0 ? 14 : 18))) {   // (round 5: from 16 K weights, was 256 K: the fused audio pair's
    // third conv and Linear at the rollout batch took a tiled launch + split-K reduce / a 32-row tile for 14 rows; knob 33 = -1: the old limit)
    bool dense;
    if (p.convT) dense = a.Hi == 1 && a.Wi == 1 && p.thn == 1 && p.twn == 1 && p.th0 == 0 && p.tw0 == 0 && a.Ho == 2 && a.Wo == 2;
    else dense = a.Ho == 1 && a.Wo == 1 && a.ph == 0 && a.pw == 0 && p.thn == a.Hi && p.twn == a.Wi && a.mulh == 1 && a.mulw == 1 &&
                 a.offh + p.th0 == 0 && a.offw + p.tw0 == 0;
    if (dense) {
      const int phases = p.convT ? 4 : 1;
      const int nw = skinny_waves(p.Kw / 16);
      const bool wide = a.N * phases >= 64 * 16;
      // two columns per block where four would leave most CUs without a block (N = 512 of one phase: 128 blocks): as the skinny gather
      // kernel's 16-row blocks, the weights stream at a per-CU rate.  Same values (a column's sum does not depend on its neighbours).
      const bool two = !wide && g_skinny_mgb >= 0 && a.N % 2 == 0 && (long)phases * ((a.N + 3) / 4) < 192;
      const dim3 grid((unsigned)(phases * (wide ? (a.N + 15) / 16 : two ? (a.N + 1) / 2 : (a.N + 3) / 4))), blk(64 * nw);
#define M2H_SKINNY_ROWS(NW_)                                                                       \
  do {                                                                                             \
    if (wide) M2H_LAUNCH((skinny_rows_kernel<16, NW_>), grid, blk, 0, st, p);              \
    else if (two) M2H_LAUNCH((skinny_rows_kernel<2, NW_>), grid, blk, 0, st, p);           \
    else M2H_LAUNCH((skinny_rows_kernel<4, NW_>), grid, blk, 0, st, p);                    \
  } while (0)
      if (nw == 4) M2H_SKINNY_ROWS(4);
      else if (nw == 8) M2H_SKINNY_ROWS(8);
      else M2H_SKINNY_ROWS(16);
#undef M2H_SKINNY_ROWS
      return launch_status("conv_igemm_f32 (skinny rows)");
    }
  }
  // small pixel counts per phase (<= 1024; knob 24 > 0 overrides the limit): 32 x 16 tiles without LDS staging or split-K.
  // Also 1024 < M <= 4096 pixels against TINY weights (< 64 K elements: the rollout batch's first encoder stage, 3584 pixels x 512 x 64, and the
  // visual encoder's second and third convs): the tiled engine fills the chip there only through split-K slabs + a reduce launch (10 + 5 us
  // for 0.2 GFLOP); knob 33 = -1: off
  const bool tiny_w = (size_t)a.N * p.Kw * (p.convT ? 4 : 1) < ((size_t)1 << 16);
  const long skinny_lim = g_skinny_gather > 0 ? g_skinny_gather : (tiny_w && g_skinny_tiny >= 0 ? 4096 : 1024);
  if (p.math == 0 && g_skinny_gather >= 0 && g_fast_loader >= 0 && p.fast_ok && M > 16 && M <= skinny_lim &&
      a.N % 16 == 0 && a.out_mode == M2H_OUT_NHWC && a.head_w == nullptr && (a.operand_format & M2H_FMT_LAYOUT_BITS) == 0 &&
      p.Ctot % 16 == 0 && (tiny_w ? (M > 1024 && g_skinny_tiny >= 0) : a.cls_table == nullptr)) {
    const int phases = p.convT ? 4 : 1;
    p.MT = (int)((M + 15) / 16);
    const long blocks2 = (long)phases * (a.N / 16) * ((p.MT + 1) / 2);
    const int nw = skinny_waves(p.Kw / 16);
    // 32 pixel rows per block where that fills the chip; 16 where it would leave most CUs without a block (the deep U-Net stages at the
    // rollout batch: 56 rows x 512 channels = 64 blocks of 32 rows): the weights stream at a per-CU rate, so twice the blocks stream them
    // twice as fast, and their second read comes out of L2.  Same values: a row's sum does not depend on the rows beside it.
    const bool one = g_skinny_mgb >= 0 && p.MT >= 2 && (g_skinny_mgb == 1 || blocks2 < (g_skinny_mgb > 1 ? g_skinny_mgb : 192));   // (knob 38 > 1: the block-count threshold, A/B)
    // 64 rows x 32 columns per block (eight waves) where that still gives the chip a block per CU (the update batch's 280-row Linear layers
    // against 1536 / 4608 columns: 864 / 2592 blocks of 32 x 16): such a launch is bound by the L2 -> CU operand stream -- the weights
    // pass once per row block, the activations once per column block -- and both shares halve
    const long blocks4 = (long)phases * ((a.N + 31) / 32) * ((p.MT + 3) / 4);   // 64 rows x 32 columns
    // (both wide forms for ONE-pixel outputs only -- nn.Linear and the encoders' full-spatial convs over the update batch, where they were measured:
    // 38 -> 25, 35 -> 31, 37 -> 26, 55 -> 41 us per policy epoch; on the passive step's U-Net stages of 256-1024 pixels they measured 2 % slower)
    const bool dense = a.Hq == 1 && a.Wq == 1 && !p.convT;
    const bool four = dense && !one && g_skinny_mgb == 0 && p.MT >= 8 && blocks4 >= 240;
    const long blocks2w = (long)phases * ((a.N + 31) / 32) * ((p.MT + 1) / 2);   // 32 rows x 32 columns (the 280-row layers against 512 columns)
    const bool wide2 = dense && !one && !four && g_skinny_mgb == 0 && p.MT >= 8 && blocks2w >= 128;
    const long blocks = four ? blocks4 : (wide2 ? blocks2w : (one ? (long)phases * (a.N / 16) * p.MT : blocks2));
#define M2H_SKINNY_GATHER(NW_)                                                                              \
  do {                                                                                                      \
    if (four) M2H_LAUNCH((skinny_gather_kernel<4, (NW_ > 8 ? 8 : NW_), 2>), dim3((unsigned)blocks), dim3(64 * (NW_ > 8 ? 8 : NW_)), 0, st, p); \
    else if (wide2) M2H_LAUNCH((skinny_gather_kernel<2, NW_, 2>), dim3((unsigned)blocks), dim3(64 * NW_), 0, st, p); \
    else if (one) M2H_LAUNCH((skinny_gather_kernel<1, NW_>), dim3((unsigned)blocks), dim3(64 * NW_), 0, st, p);  \
    else M2H_LAUNCH((skinny_gather_kernel<2, NW_>), dim3((unsigned)blocks), dim3(64 * NW_), 0, st, p);      \
  } while (0)
    if (nw == 4) M2H_SKINNY_GATHER(4);
    else if (nw == 8) M2H_SKINNY_GATHER(8);
    else M2H_SKINNY_GATHER(16);
#undef M2H_SKINNY_GATHER
    return launch_status("conv_igemm_f32 (skinny gather)");
  }
  // 3x3 / stride 1 / pad 1 over 16- or 32-channel, 32-pixel-wide images in fp32 math, many rows (AcousticMem in update_sep)
  if (p.math == 0 && g_row3x3 >= 0 && !p.convT && a.nth == 3 && a.ntw == 3 && a.stride == 1 && a.os == 1 && a.ph == 0 && a.pw == 0 &&
      (a.mulh == 1 || a.mulh == -1) && a.offh == -a.mulh && a.mulw == a.mulh && a.offw == a.offh && a.C1 == 0 && (a.C0 == 16 || a.C0 == 32) &&
      a.Wq == 32 && a.Wi == 32 && a.Hq == a.Hi && a.Ho == a.Hq && a.Wo == a.Wq && a.Hq % 4 == 0 && a.N <= 32 && a.N % 4 == 0 &&
      a.scale == nullptr && a.cls_table == nullptr && a.head_w == nullptr && (a.operand_format & M2H_FMT_LAYOUT_BITS) == 0 && (long)a.B * (a.Hq / 4) >= 512 &&
      (a.out_mode == M2H_OUT_NHWC || a.N % 16 == 0)) {
    const long chunks = (long)a.B * (a.Hq / 4);
    const dim3 grid((unsigned)(chunks < 512 ? chunks : 512)), blk(256);
    if (a.N <= 16 && a.C0 == 32) M2H_LAUNCH((conv3x3_row_kernel<16, 32>), grid, blk, 0, st, p);
    else if (a.N <= 16) M2H_LAUNCH((conv3x3_row_kernel<16, 16>), grid, blk, 0, st, p);
    else if (a.C0 == 32) M2H_LAUNCH((conv3x3_row_kernel<32, 32>), grid, blk, 0, st, p);
    else M2H_LAUNCH((conv3x3_row_kernel<32, 16>), grid, blk, 0, st, p);
    if (l1 != nullptr) {
      M2H_LAUNCH(l1_partials_sum_kernel, dim3(1), dim3(256), 0, st, l1->partials, (int)grid.x, l1->inv, l1->loss);
      return launch_status("conv_igemm_f32 (image-row 3x3 + L1 loss)");
    }
    return launch_status("conv_igemm_f32 (image-row 3x3)");
  }
  // the same image-row shapes in bf16x3 math (update_sep with sep_update_math / the far-target leg): split operands in LDS, bf16 MFMAs
  if (p.math == 1 && !p.presplit && !p.dst_split && g_row3x3 >= 0 && !p.convT && a.nth == 3 && a.ntw == 3 && a.stride == 1 && a.os == 1 && a.ph == 0 &&
      a.pw == 0 && (a.mulh == 1 || a.mulh == -1) && a.offh == -a.mulh && a.mulw == a.mulh && a.offw == a.offh && a.C1 == 0 &&
      ((a.C0 == 32 && a.N <= 32) || (a.C0 == 16 && a.N > 16 && a.N <= 32)) && a.Wq == 32 && a.Wi == 32 && a.Hq == a.Hi && a.Ho == a.Hq && a.Wo == a.Wq &&
      a.Hq % 4 == 0 && a.N % 4 == 0 && a.scale == nullptr && a.cls_table == nullptr && a.head_w == nullptr &&
      (a.operand_format & M2H_FMT_LAYOUT_BITS) == 0 && (long)a.B * (a.Hq / 4) >= 512 && (a.out_mode == M2H_OUT_NHWC || a.N % 16 == 0)) {
    const long chunks = (long)a.B * (a.Hq / 4);
    const long cap = (a.N > 16 && a.C0 == 32) ? 512 : 768;      // resident blocks: two / three per CU (LDS)
    const dim3 grid((unsigned)(chunks < cap ? chunks : cap)), blk(256);
    if (a.N <= 16) M2H_LAUNCH((conv3x3_row_bf16x3_kernel<16, 32>), grid, blk, 0, st, p);
    else if (a.C0 == 32) M2H_LAUNCH((conv3x3_row_bf16x3_kernel<32, 32>), grid, blk, 0, st, p);
    else M2H_LAUNCH((conv3x3_row_bf16x3_kernel<32, 16>), grid, blk, 0, st, p);
    if (l1 != nullptr) {
      M2H_LAUNCH(l1_partials_sum_kernel, dim3(1), dim3(256), 0, st, l1->partials, (int)grid.x, l1->inv, l1->loss);
      return launch_status("conv_igemm_bf16x3 (image-row 3x3 + L1 loss)");
    }
    return launch_status("conv_igemm_bf16x3 (image-row 3x3)");
  }
  M2H_REQUIRE(l1 == nullptr, "conv_igemm: the fused L1 loss is built into the image-row 3x3 kernels only (3x3 / 1 / 1 over 32-channel, 32-pixel-wide images, "
                            "N = 16, B x H / 4 >= 512)");
  // bf16x3 math, wide N, enough work for one 256 x 128 tile per CU: eight waves (4 x 2 wave tiles of 64 x 64) share one staged
  // pair of operand tiles.  The 128 x 128 kernel at two blocks per CU is bound by the chip's aggregate L2 -> LDS operand stream
  // (PMC: ~8.5 TB/s of L2 reads with the matrix pipe 36 % and the LDS 39 % busy; one block per CU is only 7 % slower than two);
  // the larger tile reads 384 operand rows per 256 x 128 outputs instead of 512.  (A 256 x 64 tile for the 64-wide first encoder
  // stage measured slower: 252 vs 235 us.)
  if (g_fast_loader >= 0 && g_force_splitk <= 0) {   // split32 operands, wide N: the LDS-DMA engine (conv_dma.hip)
    const int rc = launch_igemm_dma(p, wsb, st);
    if (rc != -2) {
      if (rc != 0 || p.S == 1) return rc;
      const int rc2 = launch_splitk_reduce(p, st);
      tl_last_launch = "igemm_dma<256,128> + split-K reduce";
      return rc2;
    }
  }
  if (p.math == 1 && g_big_tile >= 0 && g_fast_loader >= 0 && p.fast_ok && p.N % 128 == 0 && g_force_splitk <= 0) {
    const long tiles = ((M + 255) / 256) * (p.N / 128) * (p.convT ? 4 : 1);
    if (tiles >= (g_big_tile > 0 ? g_big_tile : 224)) return launch_big<128>(p, wsb, st);
  }
  int BM, BN;
  pick_tile(M, p.N, BM, BN);
  if (BM == 32) return launch_cfg<32, 128, 1, 4, 2>(p, wsb, st);
  if (BM == 64) return launch_cfg<64, 128, 2, 2, 2>(p, wsb, st);
  if (p.N > 64) {
    return g_wide_stages == 1 ? launch_cfg<128, 128, 2, 2, 1>(p, wsb, st) : launch_cfg<128, 128, 2, 2, 2>(p, wsb, st);
  }
  // narrow-N tiles: one LDS stage doubles the resident blocks; measured better on every 64- and 32-wide layer once the loader
  // became scalar (layer_bench.py: down0 345 vs 382 us, up3 588 vs 618 us)
  if (p.N > 32) {
    const bool one_stage = g_force_stages != 2;
    return one_stage ? launch_cfg<128, 64, 2, 2, 1>(p, wsb, st) : launch_cfg<128, 64, 2, 2, 2>(p, wsb, st);
  }
  const bool one_stage = g_force_stages != 2;
  if (p.N <= 16 && g_narrow16 >= 0) return launch_cfg<128, 16, 4, 1, 1, 16>(p, wsb, st);  // 16-wide MFMA: no half-empty tile
  return one_stage ? launch_cfg<128, 32, 4, 1, 1>(p, wsb, st) : launch_cfg<128, 32, 4, 1, 2>(p, wsb, st);
}

}  // namespace m2h

using namespace m2h;

extern "C" {

// (1 iff the launch below would take an image-row 3x3 kernel for this shape: the rule of conv_igemm_f32's dispatch)
int m2h_conv3x3_l1_nhwc16_supported(int B, int H, int T, int C) {
  return (g_row3x3 >= 0 && B > 0 && H == 32 && T == 32 && C == 32 && (long)B * (H / 4) >= 512) ? 1 : 0;
}

int m2h_conv3x3_l1_nhwc16(const float* h, const float* wp, const float* gt_plane, float* dy, float* loss, float* partials, int B, int H, int T, int C,
                          m2h_stream stream) {
  M2H_REQUIRE(h && wp && gt_plane && dy && loss && partials, "conv3x3_l1_nhwc16: null pointer");
  M2H_REQUIRE(m2h_conv3x3_l1_nhwc16_supported(B, H, T, C), "conv3x3_l1_nhwc16: needs 32-channel, 32 x 32-pixel images and B >= 64 (the image-row kernels' shapes); "
              "use m2h_conv_igemm_f32 + m2h_l1_loss_nhwc16 otherwise");
  m2h_conv_args a = {};
  a.src0 = h; a.C0 = C; a.B = B; a.Hi = H; a.Wi = T; a.Hq = H; a.Wq = T;
  a.stride = 1; a.nth = 3; a.ntw = 3; a.mulh = 1; a.offh = -1; a.mulw = 1; a.offw = -1;
  a.wp = wp; a.N = 16; a.slope = 1.f;
  a.dst = dy; a.Ho = H; a.Wo = T; a.os = 1; a.ldc = 16; a.out_mode = M2H_OUT_NHWC;
  ConvL1 l1 = {gt_plane, partials, loss, 1.f / ((float)B * 16.f * (float)H * (float)T)};
  return conv_igemm_f32(a, as_stream(stream), &l1);
}

}  // extern "C"

