// LDS-DMA implicit-GEMM engine (gfx950, bf16x3 math, operands in the split32 layout): the kernel of the separator U-Nets' wide
// layers (separator_cnn.py:5-24,46-52,128-135) when the whole chain runs on split32 tensors (m2h_unet_fwd with split32 weights).
//
// Why a second engine.  The register-staged engine (conv_igemm.hip) moves every operand row global -> VGPR -> LDS.  With the
// 3-MFMA bf16 products it is bound by the operand stream into the CUs, not by the matrix pipe (PMC, round 2: matrix pipe 36 %
// busy, ~15 B/clk/CU of L2 reads against the ~28-30 B/clk/CU an L2-resident stream can deliver, MI355X_MICROARCH.md "Indexed
// rows: gather into LDS"): its loads are in flight only between their issue and the LDS write one tile later, and the staging
// registers (two sets) cap the tile.  In the split32 layout an operand row of a k-tile is 128 contiguous bytes that need no
// conversion, so here they go global -> LDS by `global_load_lds_dwordx4` (no VGPRs, no ds_write, no VALU): a ring of NST LDS
// stages keeps NST-1 k-tiles of loads in flight across the barriers the whole time, with counted `s_waitcnt vmcnt(N)` and one
// raw `s_barrier` per k-tile (cdna_hip_programming.md, section 5 "Pipelining across barriers" and rule 21).
//
// LDS image of a stage: [A rows BM][128 B] then [B rows BN][128 B], rows UNPADDED (one DMA wave-instruction writes 1 KiB =
// 8 rows x 128 B lane-linearly).  Bank conflicts of the 16-byte fragment reads are avoided by a permutation of the eight
// 16-byte pieces inside each row, applied on the per-lane SOURCE address of the DMA and again on the read:
//     LDS piece j of row r  holds  split32 piece  j ^ ((r >> 1) & 7)
// (a 16-lane group of a ds_read_b128 reads 16 consecutive rows at one logical piece: row parity selects the upper / lower 128 B
// of the 256-B bank span and (r >> 1) & 7 spreads the eight rows of each parity over its eight pieces: conflict-free).
// Rows outside the image (zero padding, rows past M) read a zeroed page instead.
//
// Everything else -- GEMM view, block -> tile map (siblings and transposed-conv phases consecutive on one XCD), the bf16x3
// product order (lo*hi, hi*lo, hi*hi), split-K slabs and the fused epilogue -- is that of igemm_f32_kernel<.., SPLIT = 2>: the
// two engines give bit-identical results on the same launch (tests/test_gpu_unet.py::test_dma_engine_matches_register_engine).
#include "igemm_common.h"
#include "lds_dma.h"

namespace m2h {

// (tuning knob g_dma_split2: thread-local, m2h_internal.h) m2h_tuning_set 34: -1 = no two-way split-K on the 256 x 128 tile
// (tuning knob g_dma_shape: thread-local, m2h_internal.h) m2h_tuning_set 28: 32 = v_mfma_f32_32x32x16_bf16 fragments instead of 16x16x32
// (tuning knob g_dma: thread-local, m2h_internal.h) m2h_tuning_set 27: -1 never use this engine; 2 = also below its tile-count threshold (tests)

__device__ __attribute__((aligned(128))) float g_zero_page[2048 + 32];   // 8 KiB + one row: source of padding rows at any channel offset

#ifdef M2H_CLOCK_DIAG
// Diagnostic build only (tools/clock_diag_dma.py): shader-clock vs 100 MHz real-time stamps around the k-loop of each block.
__device__ unsigned long long g_clock_dbg_dma[8192][8];   // [0] k-loop shader clocks, [1] k-loop real time, [2..6] real-time milestones
#endif

template <int BM, int BN, int WM, int WN, int NST, int FR = 32, int DBG = 0>   // FR: MFMA shape 32x32x16 / 16x16x32; DBG (diagnostic builds only): 3 / 6 operands from one cached page, 4 no MFMAs, 5 no loads
__global__ __launch_bounds__(64 * WM * WN, 1) void igemm_dma_kernel(const IGemmP p) {
  constexpr int NW = WM * WN, NT = 64 * NW;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / FR, FN = TN / FR;
  constexpr int NE = FR == 32 ? 16 : 4;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int AG = BM / (8 * NW), BG = BN / (8 * NW);   // 8-row DMA groups per wave and stage
  constexpr int LPT = AG + BG;                             // DMA instructions per wave and k-tile
  static_assert(AG >= 1 && BG >= 1 && AG * 8 * NW == BM && BG * 8 * NW == BN && FM >= 1 && FN >= 1, "tile shape");
  static_assert(NST == 3, "ring depth (the waits below count at most two younger tiles)");
  static_assert(LPT * 2 <= 63, "vmcnt range");
  using AccT = typename std::conditional<FR == 32, f32x16, f32x4>::type;

  __shared__ __attribute__((aligned(1024))) char smem[NST * ST_BYTES];
  __shared__ int ri_out[BM], ri_bc[BM];
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA destinations and wave-tile offsets are SGPR arithmetic
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = lane >> 3;                                     // row of this lane inside a DMA group
  const int frow = lane & (FR - 1), half = lane / FR;             // fragment row / k group (8 bf16 each) of the MFMA operand layout

  // ---- block -> (m-tile, n-tile, phase): as igemm_f32_kernel ----
  const int L = blockIdx.x;
  const int xcd = L & 7;
  int idx = L >> 3;
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
  if (mt >= p.MT) return;
  const int m0 = mt * BM, n0 = nt * BN;
#ifdef M2H_CLOCK_DIAG
  const unsigned long long dbg_s0 = __builtin_amdgcn_s_memrealtime();
#endif
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

  // ---- the rows this lane feeds (fixed for the whole kernel) ----
  int a_qh[AG], a_rw[AG], a_bpix[AG];
#pragma unroll
  for (int i = 0; i < AG; ++i) {
    const int r = (wave + NW * i) * 8 + lrow;
    const int m = m0 + r;
    int qh = -(1 << 24), rw = -(1 << 24), bpix = 0, out = -1, bc = 0;
    if (m < p.M) {
      int q, rr, b;
      decode_row(p, m, ph, pw, q, rr, b, out, bc);
      qh = q * p.stride + offh;
      rw = rr * p.stride + offw;
      bpix = b * p.Hi * p.Wi;
    }
    a_qh[i] = qh;
    a_rw[i] = rw;
    a_bpix[i] = bpix;
    if ((lane & 7) == 0) {
      ri_out[r] = out;
      ri_bc[r] = bc;
    }
  }
  // LDS piece (lane & 7) of row r = 8 g + lrow holds split32 piece (lane & 7) ^ ((r >> 1) & 7), and (r >> 1) & 7 = 4 (g & 1) + (lrow >> 1)
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  const char* ptrA[AG];
  const char* ptrB[BG];
  int pieceA[AG];
#pragma unroll
  for (int i = 0; i < AG; ++i) pieceA[i] = ((lane & 7) ^ ((((wave + NW * i) & 1) << 2) | (lrow >> 1))) * 16;
#pragma unroll
  for (int j = 0; j < BG; ++j) {
    const int g = wave + NW * j;
    const int r = g * 8 + lrow;
    const int piece = (lane & 7) ^ (((g & 1) << 2) | (lrow >> 1));
    ptrB[j] = reinterpret_cast<const char*>(wbase) + ((size_t)min(n0 + r, p.N - 1) * p.K) * 4 + piece * 16;   // rows past N re-read row N-1 (never stored)
  }

  // ---- k-tile walk (uniform): tile -> (tap of the window, source, 32-channel chunk) ----
  const int nk_all = p.Kw / BK;
  const int split = blockIdx.y;
  const int kt0 = (int)(((long)nk_all * split) / p.S);
  const int kt1 = (int)(((long)nk_all * (split + 1)) / p.S);
  const int nk = kt1 - kt0;
  // k-tiles run in (tap, chunk) order, the register engine's: increments only, and the staged rows' pointers (bounds checks + a
  // 64-bit address per row) are rebuilt only when the tap or the source changes.  (Orders that keep the re-reads of an input line
  // within a few tiles -- (chunk, tap), and (parity class, chunk, tap in class) for the stride-2 convs -- were built and measured
  // in round 2: the locality was worth 0.5 % of the step, the per-tile pointer rebuild cost 3 %; removed.)
  int u_th, u_tw, u_ci;
  auto rebuild_rows = [&]() {   // row pointers of the current (tap, source)
    const bool second = u_ci >= p.C0 && p.src1 != nullptr;
    const int dh = u_th * mulh, dw = u_tw * mulw;
    const int Cs = second ? p.C1 : p.C0;
    const char* base = reinterpret_cast<const char*>(second ? p.src1 : p.src0);
#pragma unroll
    for (int i = 0; i < AG; ++i) {
      const int ih = a_qh[i] + dh, iw = a_rw[i] + dw;
      const bool ok = (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
      const size_t off = (size_t)(unsigned)(a_bpix[i] + ih * p.Wi + iw) * (unsigned)Cs * 4u;
      ptrA[i] = ((ok && DBG != 3 && DBG != 6) ? base + off : zero) + pieceA[i];
    }
  };
  {
    const int nch = p.Ctot / BK;
    const int w_a = kt0 / nch;
    u_ci = (kt0 - w_a * nch) * BK;
    u_th = p.th0 + w_a / p.twn;
    u_tw = p.tw0 + w_a % p.twn;
    rebuild_rows();
  }
  int issued = 0;   // tiles issued so far (the next one goes to stage issued % NST)
  int istage = 0;
  auto issue_tile = [&]() {
    const bool second = u_ci >= p.C0 && p.src1 != nullptr;
    const unsigned cofs = (unsigned)(second ? u_ci - p.C0 : u_ci) * 4u;
    const unsigned kofs = DBG == 6 ? 0u : (unsigned)((u_th * p.ntw + u_tw) * p.Ctot + u_ci) * 4u;
    const unsigned sbase = lds0 + (unsigned)istage * ST_BYTES + (unsigned)wave * 1024u;
    if constexpr (DBG != 5) {
      const char* sa[AG];
      const char* sb[BG];
#pragma unroll
      for (int i = 0; i < AG; ++i) sa[i] = ptrA[i] + cofs;
#pragma unroll
      for (int j = 0; j < BG; ++j) sb[j] = ptrB[j] + kofs;
      if (!((DBG == 7 || DBG == 8) && (issued & 3) != 0)) glds16_run<AG>(sa, sbase, NW * 1024u);   // 7 / 8: the pixel rows of one tile in four (timing of a shared patch)
      glds16_run<BG>(sb, sbase + A_BYTES, NW * 1024u);
    }
    ++issued;
    istage = istage + 1 == NST ? 0 : istage + 1;
    // advance to the next tile: the row pointers are rebuilt when the tap or the source changes
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
    if (reseg && issued < nk) rebuild_rows();
  };

  AccT acc[FM][FN];
#pragma unroll
  for (int mi = 0; mi < FM; ++mi)
#pragma unroll
    for (int ni = 0; ni < FN; ++ni)
#pragma unroll
      for (int e = 0; e < NE; ++e) acc[mi][ni][e] = 0.f;

  // fragment addresses: row base + permuted piece.  32x32x16: steps st = 0, 1 read hi pieces 2 st + half and lo pieces
  // 4 + 2 st + half; 16x16x32: one MFMA spans the tile's 32 channels, hi piece = half (0..3), lo piece = 4 + half.
  const int fx = (frow >> 1) & 7;   // (r >> 1) & 7 of every fragment row of this lane (fragments start at multiples of 16 rows)
  int offH[2], offL[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    offH[st] = (((FR == 32 ? 2 * st : 0) + half) ^ fx) * 16;
    offL[st] = ((4 + (FR == 32 ? 2 * st : 0) + half) ^ fx) * 16;
  }
  const int a_row = (wm * TM + frow) * 128, b_row = A_BYTES + (wn * TN + frow) * 128;
  auto mfma = [&](const f32x4& a, const f32x4& b, AccT& c) {
    if constexpr (FR == 32)
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };
  auto wait_and_barrier = [&](auto younger) {   // younger: tiles issued after the one being waited for (compile-time)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(decltype(younger)::value * (DBG == 8 ? BG : LPT)) : "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  using Y0 = std::integral_constant<int, 0>;
  using Y1 = std::integral_constant<int, 1>;
  using Y2 = std::integral_constant<int, 2>;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  // ---- pipeline ----
  // The ring holds NST k-tiles; the fragment reads run half a tile ahead of the MFMAs and the barrier sits in the MIDDLE of a
  // tile, behind the first half's MFMAs, so the matrix pipe has queued work while the waves meet:
  //   iteration t:  read half 1 of tile t | MFMAs of half 0 | wait: own DMA of tile t+1 landed, own LDS reads done | barrier |
  //                 read half 0 of tile t+1 | MFMAs of half 1, the DMA of tile t+NST -> the stage of tile t issued among them
  // At the barrier of iteration t every wave has finished ALL its reads of tile t (half 0 was read in iteration t-1, half 1 is
  // waited for with lgkmcnt(0)), so that stage is free for tile t+NST, and every wave's share of tile t+1 has landed (each wave
  // waits for its own DMA, leaving the younger tile in flight).  NST-1 tiles of loads are in flight all the time.
  // Halves: 32x32x16 -- the two 16-channel k-steps; 16x16x32 -- the lower / upper pixel fragments, the channel fragments (B)
  // being read once per tile into one of two register sets (tile parity).
#pragma unroll
  for (int d = 0; d < NST; ++d)
    if (d < nk) issue_tile();
  if (nk >= 3) wait_and_barrier(Y2{});   // tile 0 has landed
  else if (nk == 2) wait_and_barrier(Y1{});
  else wait_and_barrier(Y0{});
#ifdef M2H_CLOCK_DIAG
  const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  int cs = 0;   // stage of tile t
  if constexpr (FR == 32) {
    // fragments of one k-step (16 of the tile's 32 channels): hi and lo halves of FM pixel and FN channel fragments
    struct Frags {
      f32x4 ah[FM], al[FM], bh[FN], bl[FN];
    };
    auto load_frags = [&](int stage, int st, Frags& f) {
      const char* sa = smem + stage * ST_BYTES + a_row;
      const char* sb = smem + stage * ST_BYTES + b_row;
#pragma unroll
      for (int mi = 0; mi < FM; ++mi) {
        f.ah[mi] = *reinterpret_cast<const f32x4*>(sa + mi * FR * 128 + offH[st]);
        f.al[mi] = *reinterpret_cast<const f32x4*>(sa + mi * FR * 128 + offL[st]);
      }
#pragma unroll
      for (int ni = 0; ni < FN; ++ni) {
        f.bh[ni] = *reinterpret_cast<const f32x4*>(sb + ni * FR * 128 + offH[st]);
        f.bl[ni] = *reinterpret_cast<const f32x4*>(sb + ni * FR * 128 + offL[st]);
      }
    };
    auto mfma_rows = [&](const Frags& f, int mi0, int mi1) {   // pixel fragments mi0 .. mi1-1 of a k-step
#pragma unroll
      for (int mi = mi0; mi < mi1; ++mi)
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
          if constexpr (DBG != 9) {
            mfma(f.al[mi], f.bh[ni], acc[mi][ni]);
            mfma(f.ah[mi], f.bl[ni], acc[mi][ni]);
          }
          mfma(f.ah[mi], f.bh[ni], acc[mi][ni]);
        }
    };
    Frags f0, f1;
    load_frags(0, 0, f0);
    // one iteration with a successor tile; ISSUE: tile t+NST exists and goes to the stage tile t leaves
    auto body = [&](auto younger, auto issue) {
      const int ns = cs + 1 == NST ? 0 : cs + 1;
      load_frags(cs, 1, f1);
      __builtin_amdgcn_sched_barrier(0);   // the reads stay ahead of the MFMAs (the scheduler would sink them to save registers)
      if constexpr (DBG != 4) mfma_rows(f0, 0, FM);
      wait_and_barrier(younger);
      load_frags(ns, 0, f0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DBG != 4) mfma_rows(f1, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (decltype(issue)::value) issue_tile();   // in the shadow of the MFMAs just queued
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DBG != 4) mfma_rows(f1, 1, FM);
      __builtin_amdgcn_sched_barrier(0);
      cs = ns;
    };
    int t = 0;
    for (; t + NST < nk; ++t) body(Y1{}, std::true_type{});    // outstanding at the wait: tiles t+1, t+2
    if (t + 2 < nk) {                                          // t = nk-3: tiles nk-2, nk-1 outstanding, nothing left to issue
      body(Y1{}, std::false_type{});
      ++t;
    }
    if (t + 1 < nk) {                                          // t = nk-2: only the last tile outstanding
      body(Y0{}, std::false_type{});
      ++t;
    }
    load_frags(cs, 1, f1);                                     // the last tile
    if constexpr (DBG != 4) {
      mfma_rows(f0, 0, FM);
      mfma_rows(f1, 0, FM);
    }
  } else {
    // Halves = the lower / upper pixel fragments; the channel fragments (B) of tile t+1 replace those of tile t one by one
    // behind the last MFMAs that read them (ni-major order in the second half), so one register set holds them.
    constexpr int HM = FM / 2;
    static_assert(FM % 2 == 0, "16x16x32: an even number of pixel fragments per wave");
    f32x4 ah[FM], al[FM], bh[FN], bl[FN];
    auto load_a = [&](int stage, auto lo, auto hi) {
      const char* sa = smem + stage * ST_BYTES + a_row;
#pragma unroll
      for (int mi = decltype(lo)::value; mi < decltype(hi)::value; ++mi) {
        ah[mi] = *reinterpret_cast<const f32x4*>(sa + mi * FR * 128 + offH[0]);
        al[mi] = *reinterpret_cast<const f32x4*>(sa + mi * FR * 128 + offL[0]);
      }
    };
    auto load_b = [&](int stage, auto nic) {
      constexpr int ni = decltype(nic)::value;
      const char* sb = smem + stage * ST_BYTES + b_row;
      bh[ni] = *reinterpret_cast<const f32x4*>(sb + ni * FR * 128 + offH[0]);
      bl[ni] = *reinterpret_cast<const f32x4*>(sb + ni * FR * 128 + offL[0]);
    };
    auto mfma_col = [&](auto lo, auto hi, auto nic) {   // pixel fragments lo .. hi-1 against channel fragment ni
      constexpr int ni = decltype(nic)::value;
#pragma unroll
      for (int mi = decltype(lo)::value; mi < decltype(hi)::value; ++mi) {
        // the weights as the A operand (rows = channels), the pixels as B (columns): a lane ends up with four consecutive
        // channels of one pixel (igemm_common.h nhwc_tile_store_T); same products, same k order, same sums
        if constexpr (DBG != 9) {
          mfma(bh[ni], al[mi], acc[mi][ni]);
          mfma(bl[ni], ah[mi], acc[mi][ni]);
        }
        mfma(bh[ni], ah[mi], acc[mi][ni]);
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using IH = std::integral_constant<int, HM>;
    using IF = std::integral_constant<int, FM>;
    auto for_ni = [&](auto&& fn) {   // fn(integral_constant ni) for ni = 0 .. FN-1, unrolled at compile time
      auto go = [&](auto self, auto nic) {
        if constexpr (decltype(nic)::value < FN) {
          fn(nic);
          self(self, std::integral_constant<int, decltype(nic)::value + 1>{});
        }
      };
      go(go, I0{});
    };
    load_a(0, I0{}, IH{});
    for_ni([&](auto nic) { load_b(0, nic); });
    auto body = [&](auto younger, auto issue) {
      const int ns = cs + 1 == NST ? 0 : cs + 1;
      load_a(cs, IH{}, IF{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DBG != 4) for_ni([&](auto nic) { mfma_col(I0{}, IH{}, nic); });
      wait_and_barrier(younger);
      load_a(ns, I0{}, IH{});
      __builtin_amdgcn_sched_barrier(0);
      for_ni([&](auto nic) {
        if constexpr (DBG != 4) mfma_col(IH{}, IF{}, nic);
        __builtin_amdgcn_sched_barrier(0);
        load_b(ns, nic);
        if constexpr (decltype(nic)::value == 0 && decltype(issue)::value) issue_tile();
        __builtin_amdgcn_sched_barrier(0);
      });
      cs = ns;
    };
    int t = 0;
    for (; t + NST < nk; ++t) body(Y1{}, std::true_type{});
    if (t + 2 < nk) {
      body(Y1{}, std::false_type{});
      ++t;
    }
    if (t + 1 < nk) {
      body(Y0{}, std::false_type{});
      ++t;
    }
    load_a(cs, IH{}, IF{});
    if constexpr (DBG != 4) for_ni([&](auto nic) { mfma_col(I0{}, IF{}, nic); });
  }

#ifdef M2H_CLOCK_DIAG
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 8192) {
    g_clock_dbg_dma[blockIdx.x][0] = __builtin_amdgcn_s_memtime() - dbg_t0;
    g_clock_dbg_dma[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    g_clock_dbg_dma[blockIdx.x][2] = dbg_s0;
    g_clock_dbg_dma[blockIdx.x][3] = dbg_r0;
    g_clock_dbg_dma[blockIdx.x][4] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  const auto row_of = [&](int e) { return FR == 32 ? (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) : (lane >> 4) * 4 + e; };
  if constexpr (FR == 16) {
    // transposed accumulators: acc[mi][ni] = channels n0 + wn TN + 16 ni + 4 (lane >> 4) + {0..3} of pixel m0 + wm TM + 16 mi + (lane & 15)
    if (p.S > 1) {   // split-K: raw partial sums to the slab [phase][split][M][N], 16 bytes per lane
      float* slab = p.ws + ((size_t)(phase * p.S + split) * p.M) * p.N;
#pragma unroll
      for (int mi = 0; mi < FM; ++mi) {
        const int m = m0 + wm * TM + mi * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
          const int n = n0 + wn * TN + ni * 16 + 4 * (lane >> 4);
          if (n < p.N) *reinterpret_cast<f32x4*>(slab + (size_t)m * p.N + n) = acc[mi][ni];
        }
      }
      return;
    }
    __syncthreads();
    nhwc_tile_store_T<BM, BN, WM, WN, 16, NST * ST_BYTES, AccT>(p, acc, smem, ri_out, n0, tid);
#ifdef M2H_CLOCK_DIAG
    if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 8192) g_clock_dbg_dma[blockIdx.x][5] = __builtin_amdgcn_s_memrealtime();
#endif
    return;
  }
  if (p.S > 1) {
    // split-K: raw partial sums to the slab [phase][split][M][N]; BN / activation / store happen in splitk_epilogue_kernel
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
  __syncthreads();   // row bookkeeping visible (already ordered by the k-loop's barriers when nk >= 1; kept for nk == 0)
  fused_epilogue<BM, BN, WM, WN, FR, AccT, NST * ST_BYTES>(p, acc, reinterpret_cast<float*>(smem), reinterpret_cast<float*>(smem) + BM * LDK, ri_out, ri_bc, n0, tid);
#ifdef M2H_CLOCK_DIAG
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 8192) g_clock_dbg_dma[blockIdx.x][5] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int BM, int BN, int WM, int WN, int NST, int FR>
static int launch_dma_cfg(IGemmP& p, int S, hipStream_t st) {
  p.MT = (p.M + BM - 1) / BM;
  p.NT = (p.N + BN - 1) / BN;
  p.S = S;
  const long mtpad = ((long)p.MT + 7) / 8 * 8;
  const long nblk = mtpad * p.NT;
  const int phases = p.convT ? 4 : 1;
  if (nblk * phases > 0x7fffffffL) return -2;
  p.pmaj = p.convT ? 1 : 0;
  const dim3 grid((unsigned)(nblk * phases), (unsigned)S, 1);
#ifdef M2H_CLOCK_DIAG
  if (g_dma == 3) M2H_LAUNCH((igemm_dma_kernel<BM, BN, WM, WN, NST, FR, 3>), grid, dim3(64 * WM * WN), 0, st, p);
  else if (g_dma == 4) M2H_LAUNCH((igemm_dma_kernel<BM, BN, WM, WN, NST, FR, 4>), grid, dim3(64 * WM * WN), 0, st, p);
  else if (g_dma == 5) M2H_LAUNCH((igemm_dma_kernel<BM, BN, WM, WN, NST, FR, 5>), grid, dim3(64 * WM * WN), 0, st, p);
  else if (g_dma == 6) M2H_LAUNCH((igemm_dma_kernel<BM, BN, WM, WN, NST, FR, 6>), grid, dim3(64 * WM * WN), 0, st, p);
  else if (g_dma == 7) M2H_LAUNCH((igemm_dma_kernel<BM, BN, WM, WN, NST, FR, 7>), grid, dim3(64 * WM * WN), 0, st, p);
  else if (g_dma == 8) M2H_LAUNCH((igemm_dma_kernel<BM, BN, WM, WN, NST, FR, 8>), grid, dim3(64 * WM * WN), 0, st, p);
  else
#endif
  if (p.hi_only) M2H_LAUNCH((igemm_dma_kernel<BM, BN, WM, WN, NST, FR, 9>), grid, dim3(64 * WM * WN), 0, st, p);   // M2H_MATH_BF16: hi x hi only
  else M2H_LAUNCH((igemm_dma_kernel<BM, BN, WM, WN, NST, FR>), grid, dim3(64 * WM * WN), 0, st, p);
  return launch_status(BM == 256 && BN == 128 ? "igemm_dma<256,128>" : "igemm_dma");
}

// the two-way split-K launch of the 256 x 128 tile (see launch_igemm_dma): shape rule shared with conv_igemm_workspace_bytes
bool dma_split2_rule(long M, int N, int Kw, int phases, bool ws_present, size_t ws_bytes) {
  const long t256 = ((M + 255) / 256) * (N / 128) * phases;
  return g_dma == 0 && g_big_tile >= 0 && g_dma_split2 >= 0 && N % 128 == 0 && t256 < 224 && t256 * 2 >= 224 && Kw / BK >= 64 && ws_present &&
         (size_t)phases * 2 * M * N * sizeof(float) <= ws_bytes;
}
// split-K factor of the 256 x 128 tile for layers of 16 .. 223 tiles that the two-halves rule does not take: enough K-parts for one
// block per CU, at most 8, at least 8 k-tiles each (1 = not this engine's shape; fewer than 16 tiles: the weight-streaming tiles
// of the register engine)
int dma_deep_split(long M, int N, int Kw, int phases) {
  if (g_dma != 0 || g_big_tile < 0 || g_dma_split2 < 0 || N % 128 != 0) return 1;
  const long t256 = ((M + 255) / 256) * (N / 128) * phases;
  if (t256 >= 224 || t256 < 16) return 1;
  long S = (256 + t256 - 1) / t256;
  if (S < 2) S = 2;
  if (S > 8) S = 8;
  while (S > 1 && Kw / BK < 8 * S) --S;
  return (int)S;
}
static bool dma_split2_applies(const IGemmP& p, size_t ws_bytes) {
  return dma_split2_rule(p.M, p.N, p.Kw, p.convT ? 4 : 1, p.ws != nullptr, ws_bytes);
}

// Shapes of this engine: bf16x3 math on split32 operands, scalar k decode (channel counts multiples of 32), N a multiple of 128,
// no fused head, enough 256 x 128 tiles to fill the chip (or half of it with a long reduction: two K-halves per tile).  Everything
// else -- fewer tiles, 64-wide layers -- is faster on the register engine's two blocks per CU (round 2: down3 125 vs 147 us on this
// engine's 128 x 128 tile, the first encoder stage 218 vs 251 us on its 256 x 64 tile; both tiles removed).
int launch_igemm_dma(IGemmP& p, size_t ws_bytes, hipStream_t st) {
  if (g_dma < 0 || p.math != 1 || !p.presplit || !p.fast_ok || p.head_w != nullptr || p.N % 128 != 0 || p.Kw % BK != 0) return -2;
  if (p.out_mode != M2H_OUT_NHWC || p.cls_table != nullptr || p.ldc % 4 != 0 || (reinterpret_cast<size_t>(p.dst) & 15) != 0) return -2;   // whole rows through LDS (nhwc_tile_store_T)
  if ((size_t)(p.C0 > p.C1 ? p.C0 : p.C1) * 4 > 8192) return -2;   // zero page covers one pixel's channels
  if (p.M <= 64) return -2;        // skinny M: the 32- / 64-row weight-streaming tiles of the register engine
  const int phases = p.convT ? 4 : 1;
  // tile and split-K factor: the register engine's own rules (conv_igemm_f32), so that the two engines agree bit for bit
  const long t256 = (((long)p.M + 255) / 256) * (p.N / 128) * phases;
  if (g_big_tile >= 0 && (g_dma == 2 || t256 >= (g_big_tile > 0 ? g_big_tile : 224)))
    return g_dma_shape == 32 ? launch_dma_cfg<256, 128, 4, 2, 3, 32>(p, 1, st) : launch_dma_cfg<256, 128, 4, 2, 3, 16>(p, 1, st);
  // half the chip's worth of 256 x 128 tiles and a long reduction (the fourth encoder stage at the benchmark batch: 128 tiles,
  // K = 4096): two K-halves per tile into split-K slabs + the ordered reduce kernel
  if (dma_split2_applies(p, ws_bytes))
    return g_dma_shape == 32 ? launch_dma_cfg<256, 128, 4, 2, 3, 32>(p, 2, st) : launch_dma_cfg<256, 128, 4, 2, 3, 16>(p, 2, st);
  // a small fraction of a chip's worth of tiles (the two deepest stages of each U-Net at the benchmark batch: 32 / 128 tiles): S K-parts
  // per tile, one block per CU (pair_ab: down4 47.8 -> 43.0 us, up0 51.5 -> 46.9 against the register engine's 128 x 128 tiles at S = 8 / 2)
  const int S = dma_deep_split(p.M, p.N, p.Kw, phases);
  if (S > 1 && p.ws != nullptr && (size_t)phases * S * p.M * p.N * sizeof(float) <= ws_bytes)
    return g_dma_shape == 32 ? launch_dma_cfg<256, 128, 4, 2, 3, 32>(p, S, st) : launch_dma_cfg<256, 128, 4, 2, 3, 16>(p, S, st);
  return -2;
}

#ifdef M2H_CLOCK_DIAG
extern "C" int m2h_diag_read_clocks_dma(unsigned long long* host_out, int nblocks) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_clock_dbg_dma), (size_t)nblocks * 8 * sizeof(unsigned long long));
}
#endif

}  // namespace m2h
