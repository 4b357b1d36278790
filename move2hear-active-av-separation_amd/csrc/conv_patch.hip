// Shared-patch LDS-DMA engine (gfx950, bf16x3 math, split32 operands): the eight-wave tile of conv_dma.hip for the two layer
// shapes of the separator U-Nets' wide stages -- Conv2d(4, 2, 1) (separator_cnn.py:5-13,46-52) and one phase of
// ConvTranspose2d(4, 2, 1) (separator_cnn.py:15-24,128-135) -- with the pixel operand staged ONCE per four taps.
//
// Why.  In the GEMM view of conv_dma.hip every (tap, 32-channel chunk) k-tile stages its own 256 pixel rows (32 KB) beside the
// 128 weight rows (16 KB).  But the four taps of one parity class of the 4x4/s2 window -- (kh, kw), (kh, kw+2), (kh+2, kw),
// (kh+2, kw+2) -- read the SAME input pixels one output column / row apart, and so do the four taps of a transposed-conv phase:
//     conv   tap (2a + gh, 2b + gw) of output (oh, ow)  reads  P[oh + a][ow + b],  P[i][j] = in(2i + gh - 1, 2j + gw - 1)
//     convT  tap (th, tw) of phase (ph, pw), pixel (q, r) reads  P[q + a][r + b],    P[i][j] = in(i + ph - 1, j + pw - 1),
//                                                              a = ph ? th : 1 - th,  b = pw ? tw : 1 - tw
// So a block stages the patch P of its output pixels once per (class, chunk) and the four taps read their fragments from it at a
// row shift.  Measured on the old engine with three of four pixel-row DMAs dropped (tools/clock_diag_dma.py, knob 7): the
// operand stream into the CU, not the matrix pipe, was what the k-tile waited for (2 390 -> 2 236 cycles per k-tile and a 5 %
// higher clock under the lighter memory traffic); every re-read of an input line now comes from LDS instead of from beyond L2.
//
// Two patch forms:
//   WHOLE = 1  the tile holds whole images (H W divides the tile's pixel count; every decoder stage and the deeper encoder
//              stages).  A tile row spans the image's width and its rows span the image's height, so every halo pixel of P is
//              zero padding: only the data pixels are staged -- patch row m = the input pixel under output pixel m: in(k, l) for a
//              transposed conv, in(2k + 1 - gh, 2l + 1 - gw) for class (gh, gw) of a conv -- and a fragment row that falls off
//              its image reads a row of zeros instead (per-lane edge flags against the tap's direction).
//   WHOLE = 0  a tile is some rows of a larger image (down1 at 256 frames: 4 of 8 rows): the (rows + 1) x (W + 1) patch with
//              its halo row (data of the neighbouring tile, or padding) and halo column staged as rows of their own, 297 to 340
//              rows in a 384-row buffer (rows past the patch copy zeros).
// Tiles: 256 x 128 (4 x 2 waves) for N % 128 == 0, 512 x 64 (8 x 1 waves) for the 64-wide decoder stage; wave tile 64 x 64.
//
// LDS: two patch buffers + a ring of three weight stages (+ the zero row), rows unpadded.  Weight rows carry the piece permutation
// of conv_dma.hip (LDS piece j of row r holds split32 piece j ^ ((r >> 1) & 7): conflict-free for fragments that start at
// multiples of 16 rows).  Patch fragments start at ANY row (shifts 0, 1, W + 1, W + 2 on lines of W + 1 rows), where that map is
// two-way conflicted (PMC: SQ_LDS_BANK_CONFLICT 28 % of SQ_LDS_IDX_ACTIVE); patch rows use a rotation instead: LDS piece
// (p + (r & 6)) & 7 holds split32 piece p.  A 16-lane group of a ds_read_b128 is fragment rows {0-3, 12-15} at one piece and
// {4-11} at its neighbour (p ^ 1): per row parity the row pairs r >> 1 are eight consecutive numbers, a cyclic block of four with
// piece p and the complementary block with p ^ 1, so the rotation sends the first to the pieces of p's parity and the second to
// the others: 16 distinct 16-byte slots at every start row (tests/test_kernel_model.py).  (WHOLE = 0: fragments start at
// multiples of 16 output columns and W is a multiple of 16, so a fragment never crosses the end of a patch line.)
//
// Pipeline: as conv_dma.hip's 16x16x32 path (fragment reads half a tile ahead, one barrier in the middle of each k-tile,
// counted vmcnt waits, straight-line steady state), unrolled over the four taps of a patch: the weights of tile t+3 are issued in
// tile t, the patch of class/chunk s+1 in the first tile of s (behind that tile's weights, so the counts are compile-time:
// BG, BG + AG, BG + AG, BG loads may stay in flight at the four barriers).
// k order: (class, chunk, tap) -- another summation order than the (tap, chunk) of the other engines (rel-L1 ~1e-6 between them).
#include "igemm_common.h"
#include "lds_dma.h"

namespace m2h {

// (tuning knob g_patch: thread-local, m2h_internal.h) m2h_tuning_set 36: -1 never use this engine; 2 = also below its tile-count threshold (tests); 3 = as 2, and the whole-image patch wherever it fits

__device__ __attribute__((aligned(128))) float g_zero_page_patch[2048 + 32];   // 8 KiB + one row: source of padding rows at any channel offset

#ifdef M2H_CLOCK_DIAG
// Diagnostic build only (tools/clock_diag_dma.py patch ...): shader-clock vs 100 MHz real-time stamps around the k-loop of each block.
__device__ unsigned long long g_clock_dbg_patch[8192][8];   // [0] k-loop shader clocks, [1] k-loop real time, [2..5] real-time milestones
#endif

namespace {

constexpr int PNW = 8;          // waves per block
constexpr int PHALO_ROWS = 384;  // WHOLE = 0: rows of a patch buffer
constexpr int PNSTB = 3;        // weight stages

struct PatchGeo {
  int W1;         // patch row length: W + 1 (WHOLE = 0) / W
  int seg_rows;   // patch rows of one segment: (rows + 1) (W + 1) / rows W
  int nseg;       // segments (images) per tile
  int w_sh;       // log2 W
  int seg_sh;     // log2 of the output pixels per segment
  int rows;       // output rows per segment
};

template <int WM, int WN, int WHOLE>
struct PatchCfg {
  static constexpr int BM = 64 * WM, BN = 64 * WN;
  static constexpr int A_ROWS = WHOLE ? BM : PHALO_ROWS;
  static constexpr int A_BYTES = A_ROWS * 128, B_BYTES = BN * 128;
  static constexpr int A_STRIDE = A_BYTES + (WHOLE ? 256 : 0);   // WHOLE: two zero rows (one per row parity: a lane sent there keeps its bank slot) behind each patch buffer
  static constexpr int PIPE = 2 * A_STRIDE + PNSTB * B_BYTES;    // bytes of the main loop's buffers
  static constexpr int STORE = BM * (BN * 4 + 16);               // the epilogue's row image (nhwc_tile_store_T, one pass)
  static constexpr int SCRATCH = PIPE > STORE ? PIPE : STORE;
  static constexpr int SMEM = SCRATCH;
};

}  // namespace

template <int WM, int WN, int WHOLE, int CONVT, int DBG>   // CONVT: a transposed-conv phase (one class for the whole kernel); DBG (diagnostic builds only): 4 no MFMAs, 5 no loads, 6 no loads and no k-loop barrier, 7 no k-loop barrier
__global__ __launch_bounds__(64 * PNW, 1) void igemm_patch_kernel(const IGemmP p, const PatchGeo g) {
  using Cfg = PatchCfg<WM, WN, WHOLE>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, A_BYTES = Cfg::A_BYTES, B_BYTES = Cfg::B_BYTES;
  constexpr int FM = 4, FN = 4, AG = Cfg::A_ROWS / (8 * PNW), BG = BN / (8 * PNW);
  constexpr int A_STRIDE = Cfg::A_STRIDE, B_OFF = 2 * A_STRIDE;
  static_assert(WM * WN == PNW && (WHOLE || BM == 256), "tile shape");
  static_assert(AG == 4 || AG == 6 || AG == 8, "patch DMA groups per wave");
  static_assert(BG == 1 || BG == 2, "weight DMA groups per wave");
  __shared__ __attribute__((aligned(1024))) char smem[Cfg::SMEM];
  __shared__ int ri_out[BM];
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = lane >> 3;
  const int frow = lane & 15, half = lane >> 4;

  // ---- block -> (m-tile, n-tile, phase): as igemm_dma_kernel ----
  const int L = blockIdx.x;
  const int xcd = L & 7;
  int idx = L >> 3;
  int phase = 0;
  if (CONVT) {
    phase = idx & 3;
    idx >>= 2;
  }
  const int mt = (idx / p.NT) * 8 + xcd;
  const int nt = idx - (idx / p.NT) * p.NT;
  if (mt >= p.MT) return;
  const int m0 = mt * BM, n0 = nt * BN;
#ifdef M2H_CLOCK_DIAG
  const unsigned long long dbg_s0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int ph = CONVT ? phase >> 1 : 0, pw = CONVT ? phase & 1 : 0;
  const float* wbase = p.w + (CONVT ? (size_t)phase * p.N * p.K : 0);

  for (int r = tid; r < BM; r += 64 * PNW) {
    const int m = m0 + r;
    int q, rr, b, out = -1, bc;
    if (m < p.M) decode_row(p, m, ph, pw, q, rr, b, out, bc);
    ri_out[r] = out;
  }
  if (WHOLE && tid < 128) reinterpret_cast<float*>(smem + (tid >> 6) * A_STRIDE + A_BYTES)[tid & 63] = 0.f;

  // ---- the patch rows this lane feeds (fixed for the whole kernel) ----
  const int b0 = m0 >> (g.w_sh + p.hq_sh);
  const int q0 = WHOLE ? 0 : (m0 >> g.w_sh) & (p.Hq - 1);   // first output row of the tile inside its image
  const int sm = CONVT ? 1 : 2;
  int a_hw[AG], a_pix[AG];   // (input row, column) at class offset 0, packed; image base pixel (-1: no such row)
#pragma unroll
  for (int i = 0; i < AG; ++i) {
    const int pr = (wave + PNW * i) * 8 + lrow;
    const int seg = pr / g.seg_rows;
    const int w = pr - seg * g.seg_rows;
    const int ii = w / g.W1, jj = w - ii * g.W1;
    const int b = b0 + seg;
    a_hw[i] = (((q0 + ii) * sm) << 16) | (jj * sm);
    a_pix[i] = (seg < g.nseg && b < p.B) ? b * p.Hi * p.Wi : -1;
  }
  // weight rows: LDS piece (lane & 7) of row r = 8 grp + lrow holds split32 piece (lane & 7) ^ ((r >> 1) & 7) (grp = wave + 8 j has
  // wave's parity); patch rows: LDS piece j of row r holds split32 piece (j - (r & 6)) & 7, and r & 6 = lrow & 6
  const int piece_ofs = ((lane & 7) ^ (((wave & 1) << 2) | (lrow >> 1))) * 16;
  const int piece_ofs_a = (((lane & 7) - (lrow & 6)) & 7) * 16;
  const char* zero = reinterpret_cast<const char*>(g_zero_page_patch);
  const char* ptrA[AG];
  const char* ptrB[BG];
#pragma unroll
  for (int j = 0; j < BG; ++j) {
    const int r = (wave + PNW * j) * 8 + lrow;
    ptrB[j] = reinterpret_cast<const char*>(wbase) + ((size_t)min(n0 + r, p.N - 1) * p.K) * 4 + piece_ofs;   // rows past N re-read row N-1 (never stored)
  }

  // ---- the two operand streams (uniform state) ----
  // patches: s = 0 .. NS-1 in (class, chunk) order (conv: class = (gh, gw) of the window; transposed conv: one class, the chunks
  // of both concatenated sources); weights: k-tiles t = 4 s + tt, tt = 2 a + b
  // split-K (S = 2, convs only): grid y = the half of the window's classes (gh = 0 / 1), raw partial sums to the slabs
  const int nch = p.Ctot / BK;
  const int split = blockIdx.y;
  const int cls0 = p.S == 2 ? 2 * split : 0;
  const int NS = (CONVT ? 1 : (p.S == 2 ? 2 : 4)) * nch;
  int a_cls = cls0, a_ci = 0, a_issued = 0, a_buf = 0;
  auto rebuild_rows = [&]() {
    const bool second = a_ci >= p.C0 && p.src1 != nullptr;
    const int Cs = second ? p.C1 : p.C0;
    const char* base = reinterpret_cast<const char*>(second ? p.src1 : p.src0);
    const int gh = CONVT ? ph : a_cls >> 1, gw = CONVT ? pw : a_cls & 1;
    const int dh = WHOLE ? (CONVT ? 0 : 1 - gh) : gh - 1, dw = WHOLE ? (CONVT ? 0 : 1 - gw) : gw - 1;
#pragma unroll
    for (int i = 0; i < AG; ++i) {
      const int ih = (a_hw[i] >> 16) + dh, iw = (a_hw[i] & 0xffff) + dw;
      const bool ok = a_pix[i] >= 0 && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
      const size_t off = (size_t)(unsigned)(a_pix[i] + ih * p.Wi + iw) * (unsigned)Cs * 4u;
      ptrA[i] = (ok ? base + off : zero) + piece_ofs_a;
    }
  };
  rebuild_rows();
  auto issue_patch = [&]() {
    const bool second = a_ci >= p.C0 && p.src1 != nullptr;
    const unsigned cofs = (unsigned)(second ? a_ci - p.C0 : a_ci) * 4u;
    const unsigned dst = lds0 + (unsigned)a_buf * A_STRIDE + (unsigned)wave * 1024u;
    const char* sa[AG];
#pragma unroll
    for (int i = 0; i < AG; ++i) sa[i] = ptrA[i] + cofs;
    if constexpr (DBG != 5 && DBG != 6) {
      glds16_run<4>(sa, dst, PNW * 1024u);
      if constexpr (AG == 6) glds16_run<2>(sa + 4, dst + 4u * PNW * 1024u, PNW * 1024u);
      if constexpr (AG == 8) glds16_run<4>(sa + 4, dst + 4u * PNW * 1024u, PNW * 1024u);
    }
    a_buf ^= 1;
    ++a_issued;
    a_ci += BK;
    bool reseg = a_ci == p.C0 && p.src1 != nullptr;
    if (a_ci == p.Ctot) {
      a_ci = 0;
      ++a_cls;
      reseg = true;
    }
    if (reseg && a_issued < NS) rebuild_rows();
  };
  int b_cls = cls0, b_ci = 0, b_tap = 0, b_stage = 0;
  auto issue_weights = [&]() {
    const int a = b_tap >> 1, b = b_tap & 1;
    const int th = CONVT ? (ph ? a : 1 - a) : 2 * a + (b_cls >> 1);
    const int tw = CONVT ? (pw ? b : 1 - b) : 2 * b + (b_cls & 1);
    const unsigned kofs = (unsigned)((th * p.ntw + tw) * p.Ctot + b_ci) * 4u;
    const unsigned dst = lds0 + (unsigned)B_OFF + (unsigned)b_stage * B_BYTES + (unsigned)wave * 1024u;
    const char* sb[BG];
#pragma unroll
    for (int j = 0; j < BG; ++j) sb[j] = ptrB[j] + kofs;
    if constexpr (DBG != 5 && DBG != 6) glds16_run<BG>(sb, dst, PNW * 1024u);
    b_stage = b_stage + 1 == PNSTB ? 0 : b_stage + 1;
    if (++b_tap == 4) {
      b_tap = 0;
      b_ci += BK;
      if (b_ci == p.Ctot) {
        b_ci = 0;
        ++b_cls;
      }
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int mi = 0; mi < FM; ++mi)
#pragma unroll
    for (int ni = 0; ni < FN; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addresses ----
  // pixels: patch row of output pixel (wm 64 + 16 mi + frow) at shift 0, and (WHOLE) which image edges it lies on: bit 0 top,
  // 1 bottom, 2 left, 3 right; weights: as igemm_dma_kernel's 16x16x32 path
  const int fx = (frow >> 1) & 7;
  const int offH = (half ^ fx) * 16, offL = ((4 + half) ^ fx) * 16;
  const int b_row = B_OFF + (wn * 64 + frow) * 128;
  f32x4 ah[FM], al[FM], bh[FN], bl[FN];
  // byte offsets (inside a patch buffer) of this lane's hi pieces for the four taps of the current class; lo = ^ 64.  Rebuilt only
  // when the class changes (never, for a transposed conv): the k-loop adds the buffer base and reads.  (Computed per read, the
  // whole-image form's edge tests were 60 VALU instructions per k-tile against 18 of the halo form.)
  int atab[4][FM];
  const int W1 = g.W1;
  auto build_atab = [&](int cls) {
    const int gh = CONVT ? ph : cls >> 1, gw = CONVT ? pw : cls & 1;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int a = tt >> 1, b = tt & 1;
      // WHOLE: rows / columns -1, 0 (gh = 0) or 0, +1 (gh = 1) of the pixel's own, zeros past the image's edges
      const int shift = WHOLE ? (a - 1 + gh) * W1 + (b - 1 + gw) : a * W1 + b;
      const int kill = WHOLE ? ((a == 0 && gh == 0) ? 1 : 0) | ((a == 1 && gh == 1) ? 2 : 0) | ((b == 0 && gw == 0) ? 4 : 0) | ((b == 1 && gw == 1) ? 8 : 0) : 0;
#pragma unroll
      for (int mi = 0; mi < FM; ++mi) {
        // patch row of output pixel (wm 64 + 16 mi + frow) at shift 0, and which image edges it lies on: bit 0 top, 1 bottom, 2 left, 3 right
        const int ml = wm * 64 + mi * 16 + frow;
        const int seg = ml >> g.seg_sh;
        const int rem = ml & ((1 << g.seg_sh) - 1);
        const int il = rem >> g.w_sh, jl = rem & ((1 << g.w_sh) - 1);
        const int edge = (il == 0 ? 1 : 0) | (il == g.rows - 1 ? 2 : 0) | (jl == 0 ? 4 : 0) | (jl == (1 << g.w_sh) - 1 ? 8 : 0);
        const int row = seg * g.seg_rows + il * g.W1 + jl + shift;
        const int ad = (row << 7) | (((half + (row & 6)) & 7) << 4);
        atab[tt][mi] = (edge & kill) ? A_BYTES + (ad & 255) : ad;   // zeros at the bank slot of the row they replace: the lane group stays conflict-free
      }
    }
  };
  auto load_a = [&](int buf, auto ttc, auto lo, auto hi) {
#pragma unroll
    for (int mi = decltype(lo)::value; mi < decltype(hi)::value; ++mi) {
      const int ad = buf * A_STRIDE + atab[decltype(ttc)::value][mi];
      ah[mi] = *reinterpret_cast<const f32x4*>(smem + ad);
      al[mi] = *reinterpret_cast<const f32x4*>(smem + (ad ^ 64));
    }
  };
  auto load_b = [&](int stage, auto nic) {
    constexpr int ni = decltype(nic)::value;
    const char* sb = smem + stage * B_BYTES + b_row;
    bh[ni] = *reinterpret_cast<const f32x4*>(sb + ni * 16 * 128 + offH);
    bl[ni] = *reinterpret_cast<const f32x4*>(sb + ni * 16 * 128 + offL);
  };
  auto mfma = [&](const f32x4& a, const f32x4& b, f32x4& c) {
    if constexpr (DBG != 4) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };
  auto mfma_col = [&](auto lo, auto hi, auto nic) {   // weights as the A operand: a lane ends up with four consecutive channels of one pixel
    constexpr int ni = decltype(nic)::value;
#pragma unroll
    for (int mi = decltype(lo)::value; mi < decltype(hi)::value; ++mi) {
      mfma(bh[ni], al[mi], acc[mi][ni]);
      mfma(bl[ni], ah[mi], acc[mi][ni]);
      mfma(bh[ni], ah[mi], acc[mi][ni]);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using IH = std::integral_constant<int, FM / 2>;
  using IF = std::integral_constant<int, FM>;
  auto for_ni = [&](auto&& fn) {
    auto go = [&](auto self, auto nic) {
      if constexpr (decltype(nic)::value < FN) {
        fn(nic);
        self(self, std::integral_constant<int, decltype(nic)::value + 1>{});
      }
    };
    go(go, I0{});
  };
  auto wait_and_barrier = [&](auto cnt) {   // cnt: this wave's DMA instructions that may stay in flight (compile-time)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DBG == 6 || DBG == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // diagnostic: no barrier in the k-loop (wrong results)
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(decltype(cnt)::value) : "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- pipeline ----
  issue_patch();
  issue_weights();
  issue_weights();
  issue_weights();
  wait_and_barrier(std::integral_constant<int, 2 * BG>{});   // patch 0 and the weights of tile 0 have landed (also orders ri_out, the zero row)
  int cs = 0, ab = 0;        // weight stage / patch buffer of the current tile
  int c_cls = cls0, c_ci = 0;   // class / chunk of the current patch
#ifdef M2H_CLOCK_DIAG
  const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  build_atab(cls0);
  load_a(0, std::integral_constant<int, 0>{}, I0{}, IH{});
  for_ni([&](auto nic) { load_b(0, nic); });
  // tile tt of a patch: upper pixel fragments | MFMAs of the lower ones | wait + barrier | lower fragments of the next tile |
  // MFMAs of the upper ones, the next tile's weight fragments replacing this tile's one by one, the DMA issues among them
  auto body = [&](auto ttc, auto cnt, auto issue_w, auto issue_p) {
    constexpr int TT = decltype(ttc)::value;
    const int ns = cs + 1 == PNSTB ? 0 : cs + 1;
    load_a(ab, ttc, IH{}, IF{});
    __builtin_amdgcn_sched_barrier(0);
    for_ni([&](auto nic) { mfma_col(I0{}, IH{}, nic); });
    wait_and_barrier(cnt);
    int nab = ab;
    if constexpr (TT == 3) {   // the next tile opens the next patch
      nab = ab ^ 1;
      if constexpr (!CONVT) {
        if (++c_ci == nch) {     // ... of the next class (convs: four classes per window)
          c_ci = 0;
          build_atab(++c_cls);
        }
      }
    }
    load_a(nab, std::integral_constant<int, (TT + 1) & 3>{}, I0{}, IH{});
    __builtin_amdgcn_sched_barrier(0);
    for_ni([&](auto nic) {
      mfma_col(IH{}, IF{}, nic);
      __builtin_amdgcn_sched_barrier(0);
      load_b(ns, nic);
      if constexpr (decltype(nic)::value == 0 && decltype(issue_w)::value) issue_weights();
      if constexpr (decltype(nic)::value == 1 && decltype(issue_p)::value) issue_patch();
      __builtin_amdgcn_sched_barrier(0);
    });
    cs = ns;
    ab = nab;
  };
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;
  using T3 = std::integral_constant<int, 3>;
  using C0 = std::integral_constant<int, 0>;
  using CW = std::integral_constant<int, BG>;
  using CP = std::integral_constant<int, BG + AG>;
  using Y = std::true_type;
  using N = std::false_type;
  for (int s = 0; s + 1 < NS; ++s) {
    body(T0{}, CW{}, Y{}, Y{});   // in flight at the barrier: the weights of t+2
    body(T1{}, CP{}, Y{}, N{});   // the weights of t+2 and the next patch
    body(T2{}, CP{}, Y{}, N{});   // the next patch and the weights of t+2 (the patch is the older: both stay)
    body(T3{}, CW{}, Y{}, N{});   // the weights of t+2; the next patch (older than the awaited weights) has landed
  }
  body(T0{}, CW{}, Y{}, N{});     // last patch: the last weight tile is issued here
  body(T1{}, CW{}, N{}, N{});
  body(T2{}, C0{}, N{}, N{});
  load_a(ab, std::integral_constant<int, 3>{}, IH{}, IF{});
  for_ni([&](auto nic) { mfma_col(I0{}, IF{}, nic); });

#ifdef M2H_CLOCK_DIAG
  if (tid == 0 && blockIdx.x < 8192) {
    g_clock_dbg_patch[blockIdx.x][0] = __builtin_amdgcn_s_memtime() - dbg_t0;
    g_clock_dbg_patch[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    g_clock_dbg_patch[blockIdx.x][2] = dbg_s0;
    g_clock_dbg_patch[blockIdx.x][3] = dbg_r0;
    g_clock_dbg_patch[blockIdx.x][4] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  if (p.S > 1) {   // raw partial sums to the slab [split][M][N], 16 bytes per lane; BN / activation / store in splitk_epilogue_kernel
    float* slab = p.ws + ((size_t)split * p.M) * p.N;
#pragma unroll
    for (int mi = 0; mi < FM; ++mi) {
      const int m = m0 + wm * 64 + mi * 16 + frow;
      if (m >= p.M) continue;
#pragma unroll
      for (int ni = 0; ni < FN; ++ni) {
        const int n = n0 + wn * 64 + ni * 16 + 4 * half;
        if (n < p.N) *reinterpret_cast<f32x4*>(slab + (size_t)m * p.N + n) = acc[mi][ni];
      }
    }
    return;
  }
  __syncthreads();
  nhwc_tile_store_T<BM, BN, WM, WN, 16, Cfg::SCRATCH, f32x4>(p, acc, smem, ri_out, n0, tid);
#ifdef M2H_CLOCK_DIAG
  if (tid == 0 && blockIdx.x < 8192) g_clock_dbg_patch[blockIdx.x][5] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int WM, int WN, int WHOLE>
static int launch_patch_cfg(IGemmP& p, const PatchGeo& g, int S, hipStream_t st) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  p.MT = (p.M + BM - 1) / BM;
  p.NT = p.N / BN;
  p.S = S;
  p.pmaj = p.convT ? 1 : 0;
  const long nblk = ((long)p.MT + 7) / 8 * 8 * p.NT * (p.convT ? 4 : 1);
  if (nblk > 0x7fffffffL) return -2;
  const dim3 grid((unsigned)nblk, (unsigned)S), blk(64 * PNW);
#ifdef M2H_CLOCK_DIAG
  if (p.convT) hipLaunchKernelGGL((igemm_patch_kernel<WM, WN, WHOLE, 1, 0>), grid, blk, 0, st, p, g);   // (the diagnostic variants are built for convs)
  else if (g_patch == 4) hipLaunchKernelGGL((igemm_patch_kernel<WM, WN, WHOLE, 0, 4>), grid, blk, 0, st, p, g);
  else if (g_patch == 5) hipLaunchKernelGGL((igemm_patch_kernel<WM, WN, WHOLE, 0, 5>), grid, blk, 0, st, p, g);
  else if (g_patch == 6) hipLaunchKernelGGL((igemm_patch_kernel<WM, WN, WHOLE, 0, 6>), grid, blk, 0, st, p, g);
  else if (g_patch == 7) hipLaunchKernelGGL((igemm_patch_kernel<WM, WN, WHOLE, 0, 7>), grid, blk, 0, st, p, g);
  else
#endif
  if (p.convT) hipLaunchKernelGGL((igemm_patch_kernel<WM, WN, WHOLE, 1, 0>), grid, blk, 0, st, p, g);
  else hipLaunchKernelGGL((igemm_patch_kernel<WM, WN, WHOLE, 0, 0>), grid, blk, 0, st, p, g);
  return launch_status(BN == 64 ? "igemm_patch<512,64>" : S == 2 ? "igemm_patch<256,128> (two K-halves)" : "igemm_patch<256,128>");
}

// Shapes: what launch_igemm_dma takes at split-K 1 (plus 64-wide layers), restricted to the two window geometries above with the
// whole window reached and power-of-two pixel grids: whole images per tile (any width), or -- 256 x 128 tile -- some rows of an
// image 16 / 32 / 64 pixels wide with a patch of at most 384 rows.  Returns -2 otherwise (the caller falls through).
int launch_igemm_patch(IGemmP& p, size_t ws_bytes, hipStream_t st) {
  if (g_patch < 0 || p.math != 1 || !p.presplit || !p.fast_ok || p.head_w != nullptr || p.N % 64 != 0 || p.Kw != p.K) return -2;
  if (p.out_mode != M2H_OUT_NHWC || p.cls_table != nullptr || p.ldc % 4 != 0 || (reinterpret_cast<size_t>(p.dst) & 15) != 0) return -2;
  if ((size_t)(p.C0 > p.C1 ? p.C0 : p.C1) * 4 > 8192 || p.M <= 64 || p.wq_sh < 0 || p.hq_sh < 0 || p.Hi >= 32768 || p.Wi >= 32768) return -2;
  if (p.convT) {
    if (p.ntap != 4 || p.ntw != 2 || p.thn != 2 || p.twn != 2 || p.Hq != p.Hi || p.Wq != p.Wi) return -2;
  } else {
    if (p.ntap != 16 || p.ntw != 4 || p.thn != 4 || p.twn != 4 || p.stride != 2 || p.mulh != 1 || p.mulw != 1 || p.offh != -1 || p.offw != -1 ||
        p.os != 1 || p.ph != 0 || p.pw != 0 || p.Hi != 2 * p.Hq || p.Wi != 2 * p.Wq || p.src1 != nullptr)
      return -2;
  }
  const int phases = p.convT ? 4 : 1;
  const bool wide = p.N % 128 == 0;
  const int BM = wide ? 256 : 512;
  const long tiles = (((long)p.M + BM - 1) / BM) * (p.N / (wide ? 128 : 64)) * phases;
  // half a chip's worth of 256 x 128 tiles and a long reduction (the fourth encoder stage at the benchmark batch): the two class
  // halves of the window as split-K slabs + the ordered reduce kernel (the shape rule of the LDS-DMA engine's two-K-halves launch)
  int S = 1;
  if (g_patch < 2 && tiles < 224) {
    if (!wide || p.convT || !dma_split2_rule(p.M, p.N, p.Kw, 1, p.ws != nullptr, ws_bytes)) return -2;
    S = 2;
  }
  PatchGeo g;
  const int img = p.Hq * p.Wq;
  g.w_sh = p.wq_sh;
  const int hrows = img >= BM ? BM / p.Wq : p.Hq;   // halo patch: output rows per segment
  const bool halo_ok = wide && (p.Wq == 16 || p.Wq == 32 || p.Wq == 64) && (img >= BM ? 1 : BM / img) * (hrows + 1) * (p.Wq + 1) <= PHALO_ROWS;
  if (img <= BM && (g_patch == 3 || !halo_ok)) {   // whole images per tile (where both forms fit, the halo patch measured 3-4 % faster: pair_ab, down2 81.7 vs 85.0 us, up2 153 vs 157)
    g.nseg = BM / img;
    g.rows = p.Hq;
    g.W1 = p.Wq;
    g.seg_rows = img;
    g.seg_sh = p.hq_sh + p.wq_sh;
    return wide ? launch_patch_cfg<4, 2, 1>(p, g, S, st) : launch_patch_cfg<8, 1, 1>(p, g, S, st);
  }
  if (!halo_ok) return -2;
  g.nseg = img >= BM ? 1 : BM / img;
  g.rows = hrows;
  g.W1 = p.Wq + 1;
  g.seg_rows = (g.rows + 1) * g.W1;
  g.seg_sh = __builtin_ctz((unsigned)(g.rows * p.Wq));
  return launch_patch_cfg<4, 2, 0>(p, g, S, st);
}

#ifdef M2H_CLOCK_DIAG
extern "C" int m2h_diag_read_clocks_patch(unsigned long long* host_out, int nblocks) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_clock_dbg_patch), (size_t)nblocks * 8 * sizeof(unsigned long long));
}
#endif

}  // namespace m2h
