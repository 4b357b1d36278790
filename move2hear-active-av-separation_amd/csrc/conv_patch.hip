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

#ifndef M2H_PATCH_W_AT
#define M2H_PATCH_W_AT 0   // weight-fragment step of a k-tile's second half behind which the weight DMAs are issued (A/B builds)
#endif
#ifndef M2H_PATCH_P_AT
#define M2H_PATCH_P_AT 1   // ... the patch DMAs
#endif
#ifndef M2H_PATCH_WT
#define M2H_PATCH_WT 2     // 2: a workgroup's LAST tile leaves write-through (sc1 stores), so the end-of-kernel write-back of the XCD L2s finds a round
                           // of tiles less dirty data: pair minima 2.1016 / 2.1051 / 2.1040 against 2.1130 / 2.1160 / 2.1201 ms, medians -0.15 % (A/B builds,
                           // profiles/r05_lib_ab_wt2.txt); 1: every tile (measured 1 % SLOWER); 0: none
#endif
#ifndef M2H_PATCH_PRIO
#define M2H_PATCH_PRIO 0   // 1: s_setprio 1 for waves 4-7 (the younger wave of every SIMD) for the whole kernel (A/B builds)
#endif

namespace m2h {

// (tuning knob g_patch: thread-local, m2h_internal.h) m2h_tuning_set 36: -1 never use this engine; 2 = also below its tile-count threshold (tests); 3 = as 2, and the whole-image patch wherever it fits

__device__ __attribute__((aligned(128))) float g_zero_page_patch[2048 + 32];   // 8 KiB + one row: source of padding rows at any channel offset

#ifdef M2H_CLOCK_DIAG
// Diagnostic build only (tools/clock_diag_dma.py patch ...): shader-clock vs 100 MHz real-time stamps around the k-loop of each block.
__device__ unsigned long long g_clock_dbg_patch[8192][16];   // [0] k-loop shader clocks, [1] k-loop real time, [2..5] real-time milestones, [6] tiles, [8..14] shader-clock stamps around the first tile boundary
#endif

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// one 16-byte store per lane; write-through in M2H_PATCH_WT builds (`sc1`: the bytes go to the memory side at once and the line is not kept
// dirty in the XCD's L2 -- MI355X_MICROARCH.md, stores of each flavour).  The s_nop keeps the data registers until the store has read
// them (cdna_hip_programming.md 5.7: an asm store of 12 / 16 bytes).  Counted in vmcnt like any store.
template <bool WT>
static __device__ __forceinline__ void store16(char* p, u32x4_t v) {
  if constexpr (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else *reinterpret_cast<u32x4_t*>(p) = v;
}

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
  static constexpr int MAX_N = WN == 1 ? 448 : 1024;             // output channels of a layer on this tile: its scale | shift table sits in LDS
  static constexpr int SC_OFF = PIPE;                            // (MAX_N floats each; the epilogue reads them with ds_read: a global load there would make the compiler wait for every DMA in flight)
  static constexpr int SMEM = PIPE + 2 * MAX_N * 4;
  static_assert(SMEM <= 160 * 1024, "LDS");
};

}  // namespace

// One workgroup per CU walks the tiles L = blockIdx.x, + gridDim.x, ... of the launch as ONE stream of k-tiles: the two DMA streams
// (patches, weight tiles) run ahead across tile boundaries, so the ring never drains between the tiles of a CU, and a finished tile
// leaves from the accumulator registers while the next tile's operands are already landing (no LDS image, no barrier: the weights
// are the MFMA's A operand, so a lane holds four consecutive channels of one pixel; v_permlane16_swap pairs two fragments into
// 16-byte pieces of the split32 row).  Stores count in vmcnt in issue order with the DMAs: the two waits that follow an epilogue
// are for DMAs OLDER than its stores and leave the stores in flight too (+ PST); the third wait is for a DMA issued after them.
template <int WM, int WN, int WHOLE, int CONVT, int DBG>   // CONVT: a transposed-conv phase (one class per tile); DBG: 9 = the bf16 hi halves only (M2H_MATH_BF16); diagnostic builds only: 4 no MFMAs, 5 no loads, 6 no loads and no k-loop barrier, 7 no k-loop barrier
__global__ __launch_bounds__(64 * PNW, 1) void igemm_patch_kernel(const IGemmP p, const PatchGeo g, const int ntiles) {
  using Cfg = PatchCfg<WM, WN, WHOLE>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, A_BYTES = Cfg::A_BYTES, B_BYTES = Cfg::B_BYTES;
  constexpr int FM = 4, FN = 4, AG = Cfg::A_ROWS / (8 * PNW), BG = BN / (8 * PNW);
  constexpr int A_STRIDE = Cfg::A_STRIDE, B_OFF = 2 * A_STRIDE;
  constexpr int PST = FM * FN;   // store instructions of one epilogue (per wave)
  static_assert(WM * WN == PNW && (WHOLE || BM == 256), "tile shape");
  static_assert(AG == 4 || AG == 6 || AG == 8, "patch DMA groups per wave");
  static_assert(BG == 1 || BG == 2, "weight DMA groups per wave");
  static_assert(BG + AG + PST < 64, "vmcnt field");
  __shared__ __attribute__((aligned(1024))) char smem[Cfg::SMEM];
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = lane >> 3;
  const int frow = lane & 15, half = lane >> 4;

  // ---- tile L -> (m-tile, n-tile, phase): as igemm_dma_kernel (the eight XCDs take every eighth m-tile) ----
  const int G = (int)gridDim.x;
  auto tile_of = [&](int L, int& m0, int& n0, int& phase) {
    const int xcd = L & 7;
    int idx = L >> 3;
    phase = 0;
    if (CONVT) {
      phase = idx & 3;
      idx >>= 2;
    }
    const int mq = idx / p.NT;
    m0 = (mq * 8 + xcd) * BM;
    n0 = (idx - mq * p.NT) * BN;
  };
  int c_m0, c_n0, c_phase;   // the tile being computed
  tile_of((int)blockIdx.x, c_m0, c_n0, c_phase);
  if (c_m0 >= p.MT * BM) return;   // (a grid of one tile per workgroup when MT is not a multiple of 8: the host never walks such a grid persistently)
#ifdef M2H_CLOCK_DIAG
  const unsigned long long dbg_s0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int my_tiles = (ntiles - 1 - (int)blockIdx.x) / G + 1;

  // per-channel scale / shift of the layer: loaded here, written to LDS behind the address set-up below (their latency runs under it)
  constexpr int SCN = (Cfg::MAX_N + 64 * PNW - 1) / (64 * PNW);
  float sc_r[SCN], sh_r[SCN];
#pragma unroll
  for (int k = 0; k < SCN; ++k) {
    const int n = tid + 64 * PNW * k;
    sc_r[k] = (p.scale != nullptr && n < p.N) ? p.scale[n] : 1.f;
    sh_r[k] = (p.shift != nullptr && n < p.N) ? p.shift[n] : 0.f;
  }
  if (WHOLE && tid < 128) reinterpret_cast<float*>(smem + (tid >> 6) * A_STRIDE + A_BYTES)[tid & 63] = 0.f;

  // ---- the patch rows this lane feeds: (segment, row, column) inside the tile, fixed for the whole kernel ----
  const int sm = CONVT ? 1 : 2;
  // (WHOLE: segments and lines are powers of two: recomputed from the lane's row number wherever needed, no registers held)
  auto patch_row_of = [&](int i, int lrow_) {
    const int pr = (wave + PNW * i) * 8 + lrow_;
    if constexpr (WHOLE) {
      const int seg = pr >> g.seg_sh, w = pr & ((1 << g.seg_sh) - 1);
      return seg >= g.nseg ? -1 : (seg << 24) | ((w >> g.w_sh) << 12) | (w & ((1 << g.w_sh) - 1));
    } else {
      const int seg = pr / g.seg_rows;
      const int w = pr - seg * g.seg_rows;
      const int ii = w / g.W1, jj = w - ii * g.W1;
      return seg >= g.nseg ? -1 : (seg << 24) | (ii << 12) | jj;
    }
  };
  int a_pk[WHOLE ? 1 : AG];
  if constexpr (!WHOLE) {
#pragma unroll
    for (int i = 0; i < AG; ++i) a_pk[i] = patch_row_of(i, lrow);
  }
  // weight rows: LDS piece (lane & 7) of row r = 8 grp + lrow holds split32 piece (lane & 7) ^ ((r >> 1) & 7) (grp = wave + 8 j has
  // wave's parity); patch rows: LDS piece j of row r holds split32 piece (j - (r & 6)) & 7, and r & 6 = lrow & 6
  const int piece_ofs = ((lane & 7) ^ (((wave & 1) << 2) | (lrow >> 1))) * 16;
  const int piece_ofs_a = (((lane & 7) - (lrow & 6)) & 7) * 16;
  const char* zero = reinterpret_cast<const char*>(g_zero_page_patch);
  const char* ptrA[AG];
  const char* ptrB[BG];

  // ---- the two operand streams (uniform state), running ahead of the MFMAs across tile boundaries ----
  // patches: per tile s = 0 .. NS-1 in (class, chunk) order (conv: class = (gh, gw) of the window; transposed conv: one class, the
  // chunks of both concatenated sources); weights: k-tiles t = 4 s + tt, tt = 2 a + b
  // split-K (S = 2, convs only): grid y = the half of the window's classes (gh = 0 / 1), raw partial sums to the slabs
  const int nch = p.Ctot / BK;
  const int split = blockIdx.y;
  const int cls0 = p.S == 2 ? 2 * split : 0;
  const int ncls = CONVT ? 1 : (p.S == 2 ? 2 : 4);
  const int NS = ncls * nch;
  int a_L = (int)blockIdx.x, a_b0, a_q0, a_phase;   // the tile whose patches are being issued
  int a_cls = cls0, a_ci = 0, a_buf = 0;
  auto open_a_tile = [&]() {
    int m0, n0;
    tile_of(a_L, m0, n0, a_phase);
    a_b0 = m0 >> (g.w_sh + p.hq_sh);
    a_q0 = WHOLE ? 0 : (m0 >> g.w_sh) & (p.Hq - 1);   // first output row of the tile inside its image
  };
  auto rebuild_rows = [&]() {
    const bool second = a_ci >= p.C0 && p.src1 != nullptr;
    const int Cs = second ? p.C1 : p.C0;
    const char* base = reinterpret_cast<const char*>(second ? p.src1 : p.src0);
    const int gh = CONVT ? a_phase >> 1 : a_cls >> 1, gw = CONVT ? a_phase & 1 : a_cls & 1;
    const int dh = WHOLE ? (CONVT ? 0 : 1 - gh) : gh - 1, dw = WHOLE ? (CONVT ? 0 : 1 - gw) : gw - 1;
    int lrow_ = lrow;
    asm volatile("" : "+v"(lrow_));   // (opaque: keeps the rebuild's lane-invariant terms out of the k-loop's registers)
#pragma unroll
    for (int i = 0; i < AG; ++i) {
      const int pk = WHOLE ? patch_row_of(i, lrow_) : a_pk[WHOLE ? 0 : i];
      const int b = a_b0 + (pk >> 24);
      const int ih = (a_q0 + ((pk >> 12) & 0xfff)) * sm + dh, iw = (pk & 0xfff) * sm + dw;
      const bool ok = pk >= 0 && b < p.B && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
      const size_t off = (size_t)(unsigned)((b * p.Hi + ih) * p.Wi + iw) * (unsigned)Cs * 4u;
      ptrA[i] = (ok ? base + off : zero) + piece_ofs_a;
    }
  };
  // the patch leaves in three parts, in the last k-tile of the patch two before it and the first two k-tiles of the patch before it
  // (all AG instructions in one k-tile made that k-tile ~1 100-1 400 cycles longer than the other three; in two halves still ~300:
  // in-kernel stamps, tools/clock_diag_dma.py patch).  Part 0 goes into the buffer the k-tile's first half has just finished with.
  auto issue_patch = [&](auto partc) {
    constexpr int PART = decltype(partc)::value;
    constexpr int P0N = AG == 8 ? 3 : 2, P1N = AG == 4 ? 1 : P0N, P2N = AG - P0N - P1N;
    constexpr int FIRST = PART == 0 ? 0 : (PART == 1 ? P0N : P0N + P1N), CNT = PART == 0 ? P0N : (PART == 1 ? P1N : P2N);
    static_assert(P2N >= 1 && P2N <= 2, "patch parts");
    const bool second = a_ci >= p.C0 && p.src1 != nullptr;
    const unsigned cofs = (unsigned)(second ? a_ci - p.C0 : a_ci) * 4u;
    const unsigned dst = lds0 + (unsigned)a_buf * A_STRIDE + (unsigned)(wave + PNW * FIRST) * 1024u;
    const char* sa[CNT];
#pragma unroll
    for (int i = 0; i < CNT; ++i) sa[i] = ptrA[FIRST + i] + cofs;
    if constexpr (DBG != 5 && DBG != 6) {
      if constexpr (CNT >= 2) glds16_run<2>(sa, dst, PNW * 1024u);
      if constexpr (CNT == 1) glds16_run<1>(sa, dst, PNW * 1024u);
      if constexpr (CNT == 3) glds16_run<1>(sa + 2, dst + 2u * PNW * 1024u, PNW * 1024u);
    }
    if constexpr (PART != 2) return;
    a_buf ^= 1;
    a_ci += BK;
    bool reseg = a_ci == p.C0 && p.src1 != nullptr;
    if (a_ci == p.Ctot) {
      a_ci = 0;
      reseg = true;
      if (++a_cls == cls0 + ncls) {   // the next patch opens the next tile of this workgroup (if there is none, nothing is issued from these rows)
        a_cls = cls0;
        a_L += G;
        if (a_L < ntiles) open_a_tile();
        else reseg = false;
      }
    }
    if (reseg) rebuild_rows();
  };
  int b_L = (int)blockIdx.x, b_phase = 0;   // the tile whose weights are being issued
  int b_cls = cls0, b_ci = 0, b_tap = 0, b_stage = 0;
  auto open_b_tile = [&]() {
    int m0, n0;
    tile_of(b_L, m0, n0, b_phase);
    const float* wbase = p.w + (CONVT ? (size_t)b_phase * p.N * p.K : 0);
#pragma unroll
    for (int j = 0; j < BG; ++j) {
      const int r = (wave + PNW * j) * 8 + lrow;
      ptrB[j] = reinterpret_cast<const char*>(wbase) + ((size_t)min(n0 + r, p.N - 1) * p.K) * 4 + piece_ofs;   // rows past N re-read row N-1 (never stored)
    }
  };
  auto issue_weights = [&]() {
    const int a = b_tap >> 1, b = b_tap & 1;
    const int bph = b_phase >> 1, bpw = b_phase & 1;
    const int th = CONVT ? (bph ? a : 1 - a) : 2 * a + (b_cls >> 1);
    const int tw = CONVT ? (bpw ? b : 1 - b) : 2 * b + (b_cls & 1);
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
        if (++b_cls == cls0 + ncls) {
          b_cls = cls0;
          b_L += G;
          if (b_L < ntiles) open_b_tile();
        }
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
  // when the class changes (a transposed conv: when the next tile is another phase): the k-loop adds the buffer base and reads.
  // (Computed per read, the whole-image form's edge tests were 60 VALU instructions per k-tile against 18 of the halo form.)
  int atab[4][FM];
  const int W1 = g.W1;
  auto build_atab = [&](int gh, int gw) {
    // (the lane's numbers through an opaque copy: otherwise the compiler hoists every lane-invariant subexpression of this rare
    // rebuild out of the k-loop and keeps ~50 registers alive for it)
    int frow_ = frow, half_ = half;
    asm volatile("" : "+v"(frow_), "+v"(half_));
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int a = tt >> 1, b = tt & 1;
      // WHOLE: rows / columns -1, 0 (gh = 0) or 0, +1 (gh = 1) of the pixel's own, zeros past the image's edges
      const int shift = WHOLE ? (a - 1 + gh) * W1 + (b - 1 + gw) : a * W1 + b;
      const int kill = WHOLE ? ((a == 0 && gh == 0) ? 1 : 0) | ((a == 1 && gh == 1) ? 2 : 0) | ((b == 0 && gw == 0) ? 4 : 0) | ((b == 1 && gw == 1) ? 8 : 0) : 0;
#pragma unroll
      for (int mi = 0; mi < FM; ++mi) {
        // patch row of output pixel (wm 64 + 16 mi + frow) at shift 0, and which image edges it lies on: bit 0 top, 1 bottom, 2 left, 3 right
        const int ml = wm * 64 + mi * 16 + frow_;
        const int seg = ml >> g.seg_sh;
        const int rem = ml & ((1 << g.seg_sh) - 1);
        const int il = rem >> g.w_sh, jl = rem & ((1 << g.w_sh) - 1);
        const int edge = (il == 0 ? 1 : 0) | (il == g.rows - 1 ? 2 : 0) | (jl == 0 ? 4 : 0) | (jl == (1 << g.w_sh) - 1 ? 8 : 0);
        const int row = seg * g.seg_rows + il * g.W1 + jl + shift;
        const int ad = (row << 7) | (((half_ + (row & 6)) & 7) << 4);
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
      if constexpr (DBG != 9) {   // (9: M2H_MATH_BF16, the hi halves only)
        mfma(bh[ni], al[mi], acc[mi][ni]);
        mfma(bl[ni], ah[mi], acc[mi][ni]);
      }
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
  // cnt: this wave's DMA instructions that may stay in flight (compile-time); stores: the PST stores of the previous tile's
  // epilogue are younger than the awaited DMA and may stay in flight as well (uniform)
  auto wait_and_barrier = [&](auto cnt, bool stores) {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DBG == 6 || DBG == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // diagnostic: no barrier in the k-loop (wrong results)
    else if (stores) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(decltype(cnt)::value + PST) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(decltype(cnt)::value) : "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- a finished tile leaves from the registers ----
  // acc[mi][ni] = channels 16 ni + 4 half + j of pixel 16 mi + frow.  split32 rows: fragments 2 q and 2 q + 1 are one 32-channel
  // chunk; after v_permlane16_swap (odd 16-lane rows of the first operand <-> even rows of the second) a lane of row `half` holds
  // the eight consecutive channels 16 (half & 1) + 8 (half >> 1) .. + 7 of the chunk: one 16-byte store of hi halves, one of lo
  // halves; the four lanes of a pixel cover its 64-byte hi run, then its lo run (one 128-byte line per pixel and chunk).
  auto store_tile = [&](int m0, int n0, int phase, auto lastc) {   // lastc: the workgroup's last tile (M2H_PATCH_WT 2: only that one leaves write-through)
    constexpr bool WT = M2H_PATCH_WT == 1 || (M2H_PATCH_WT == 2 && decltype(lastc)::value);
    int frow_ = frow, half_ = half;
    asm volatile("" : "+v"(frow_), "+v"(half_));   // (opaque, as in build_atab)
    if (p.S > 1) {   // raw partial sums to the slab [split][M][N], 16 bytes per lane; BN / activation / store in splitk_epilogue_kernel
      float* slab = p.ws + ((size_t)split * p.M) * p.N;
#pragma unroll
      for (int mi = 0; mi < FM; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + frow_;
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
          const int n = n0 + wn * 64 + ni * 16 + 4 * half_;
          if (m < p.M && n < p.N) *reinterpret_cast<f32x4*>(slab + (size_t)m * p.N + n) = acc[mi][ni];
          acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      return;
    }
    const int ph = phase >> 1, pw = phase & 1;
    const float* lsc = reinterpret_cast<const float*>(smem + Cfg::SC_OFF) + n0 + wn * 64 + 4 * half_;
#pragma unroll
    for (int mi = 0; mi < FM; ++mi) {
      const int m = m0 + wm * 64 + mi * 16 + frow_;
      int qq, rr, b, out = 0, bc;
      decode_row(p, m, ph, pw, qq, rr, b, out, bc);
      char* rowp = reinterpret_cast<char*>(p.dst + (size_t)out * p.ldc + n0 + wn * 64);
#pragma unroll
      for (int q = 0; q < FN / 2; ++q) {
        const bool ok = m < p.M && n0 + wn * 64 + 32 * q < p.N;   // (N % 64 == 0: a chunk is inside the row or outside it)
        f32x4 va = acc[mi][2 * q] * *reinterpret_cast<const f32x4*>(lsc + 32 * q) + *reinterpret_cast<const f32x4*>(lsc + Cfg::MAX_N + 32 * q);
        f32x4 vb = acc[mi][2 * q + 1] * *reinterpret_cast<const f32x4*>(lsc + 32 * q + 16) + *reinterpret_cast<const f32x4*>(lsc + Cfg::MAX_N + 32 * q + 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // LeakyReLU / ReLU / none, 0 <= slope <= 1 (host check): max(v, slope v), the same values as the select
          va[j] = fmaxf(va[j], va[j] * p.slope);
          vb[j] = fmaxf(vb[j], vb[j] * p.slope);
        }
        if (p.dst_split) {
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const bf16x4 ha = __builtin_convertvector(va, bf16x4), hb = __builtin_convertvector(vb, bf16x4);
          const bf16x4 la = __builtin_convertvector(va - __builtin_convertvector(ha, f32x4), bf16x4);
          const bf16x4 lb = __builtin_convertvector(vb - __builtin_convertvector(hb, f32x4), bf16x4);
          const u32x2 uha = __builtin_bit_cast(u32x2, ha), uhb = __builtin_bit_cast(u32x2, hb), ula = __builtin_bit_cast(u32x2, la), ulb = __builtin_bit_cast(u32x2, lb);
          const auto h0 = __builtin_amdgcn_permlane16_swap(uha[0], uhb[0], false, false);
          const auto h1 = __builtin_amdgcn_permlane16_swap(uha[1], uhb[1], false, false);
          const auto l0 = __builtin_amdgcn_permlane16_swap(ula[0], ulb[0], false, false);
          const auto l1 = __builtin_amdgcn_permlane16_swap(ula[1], ulb[1], false, false);
          const int cb = 128 * q + (16 * (half_ & 1) + 8 * (half_ >> 1)) * 2;   // byte offset of the lane's eight channels in the chunk's hi run
          if (ok) {
            store16<WT>(rowp + cb, u32x4{h0[0], h1[0], h0[1], h1[1]});
            store16<WT>(rowp + 64 + cb, u32x4{l0[0], l1[0], l0[1], l1[1]});
          }
        } else if (ok) {
          store16<WT>(rowp + 128 * q + 16 * half_, __builtin_bit_cast(u32x4_t, va));
          store16<WT>(rowp + 128 * q + 64 + 16 * half_, __builtin_bit_cast(u32x4_t, vb));
        }
        acc[mi][2 * q] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[mi][2 * q + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  // ---- pipeline ----
  // the first three weight tiles leave before anything else (their addresses need the tile's n-tile and phase only), under the
  // patch-row pointer set-up; then patch 0 and, when there is one, the first part of patch 1 (the steady loop's first wait counts it)
  if (M2H_PATCH_PRIO && wave >= PNW / 2) __builtin_amdgcn_s_setprio(1);
  open_b_tile();
  issue_weights();
  issue_weights();
  issue_weights();
  open_a_tile();
  rebuild_rows();
  issue_patch(I0{});
  issue_patch(std::integral_constant<int, 1>{});
  issue_patch(std::integral_constant<int, 2>{});
  constexpr int PA = AG == 8 ? 3 : 2, PB = AG == 4 ? 1 : PA;   // instructions of a patch's first / second part
  if (my_tiles * NS > 1) issue_patch(I0{});
  // (the scale / shift table last: the compiler waits for its global loads with vmcnt(0), i.e. for every DMA issued so far as well)
#pragma unroll
  for (int k = 0; k < SCN; ++k) {
    const int n = tid + 64 * PNW * k;
    if (n < Cfg::MAX_N) {
      reinterpret_cast<float*>(smem + Cfg::SC_OFF)[n] = sc_r[k];
      reinterpret_cast<float*>(smem + Cfg::SC_OFF)[Cfg::MAX_N + n] = sh_r[k];
    }
  }
  build_atab(CONVT ? c_phase >> 1 : cls0 >> 1, CONVT ? c_phase & 1 : cls0 & 1);   // (under the DMAs' latency)
  if (my_tiles * NS > 1) {
    wait_and_barrier(std::integral_constant<int, PA>{}, false);   // patch 0 and (older) the three weight tiles have landed; also orders the scale / shift table, the zero rows
  } else {
    wait_and_barrier(std::integral_constant<int, 0>{}, false);
  }
  int cs = 0, ab = 0;        // weight stage / patch buffer of the current tile
  int c_cls = cls0, c_ci = 0;   // class / chunk of the current patch
  int n_m0 = 0, n_n0 = 0, n_phase = c_phase;   // the workgroup's next tile
  if ((int)blockIdx.x + G < ntiles) tile_of((int)blockIdx.x + G, n_m0, n_n0, n_phase);
  int c_L = (int)blockIdx.x;
#ifdef M2H_CLOCK_DIAG
  const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  load_a(0, std::integral_constant<int, 0>{}, I0{}, IH{});
  for_ni([&](auto nic) { load_b(0, nic); });
  // tile tt of a patch: upper pixel fragments | MFMAs of the lower ones | wait + barrier | lower fragments of the next tile |
  // MFMAs of the upper ones, the next tile's weight fragments replacing this tile's one by one, the DMA issues among them
  // (Tried and dropped, round 5: waves 4-7 taking the k-tile's barrier at its start instead of in its middle -- half a k-tile behind
  // waves 0-3, so that one wave of a SIMD is in its MFMA-only half while its partner issues DMAs -- measured 2-5 % SLOWER on the
  // decoder stages: pair_ab --layers, up1 134 -> 141, up2 142 -> 146, up3 161 -> 165 us.)
  bool more_patches = false;   // the patch after the next one exists (its first part is issued in this patch's last k-tile)
  auto body = [&](auto ttc, auto cnt, bool stores, auto issue_w, auto issue_p) {
    constexpr int TT = decltype(ttc)::value;
    const int ns = cs + 1 == PNSTB ? 0 : cs + 1;
    load_a(ab, ttc, IH{}, IF{});
    __builtin_amdgcn_sched_barrier(0);
    for_ni([&](auto nic) { mfma_col(I0{}, IH{}, nic); });
    wait_and_barrier(cnt, stores);
    int nab = ab;
    if constexpr (TT == 3) {   // the next k-tile opens the next patch
      nab = ab ^ 1;
      if (++c_ci == nch) {     // ... of the next class (convs: four classes per window), or of the workgroup's next tile
        c_ci = 0;
        if constexpr (CONVT) {
          if (n_phase != c_phase) build_atab(n_phase >> 1, n_phase & 1);
        } else {
          if (++c_cls == cls0 + ncls) c_cls = cls0;
          build_atab(c_cls >> 1, c_cls & 1);
        }
      }
    }
    load_a(nab, std::integral_constant<int, (TT + 1) & 3>{}, I0{}, IH{});
    __builtin_amdgcn_sched_barrier(0);
    for_ni([&](auto nic) {
      mfma_col(IH{}, IF{}, nic);
      __builtin_amdgcn_sched_barrier(0);
      load_b(ns, nic);
      if constexpr (decltype(nic)::value == M2H_PATCH_W_AT && decltype(issue_w)::value) issue_weights();
      if constexpr (decltype(nic)::value == M2H_PATCH_P_AT && decltype(issue_p)::value == 1) issue_patch(std::integral_constant<int, 1>{});
      if constexpr (decltype(nic)::value == M2H_PATCH_P_AT && decltype(issue_p)::value == 2) issue_patch(std::integral_constant<int, 2>{});
      if constexpr (decltype(nic)::value == M2H_PATCH_P_AT && decltype(issue_p)::value == 3) {
        if (more_patches) issue_patch(I0{});
      }
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
  using CA = std::integral_constant<int, BG + PA>;            // the weights of t+2 and the next patch's first part
  using CAB = std::integral_constant<int, BG + PA + PB>;     // ... and its second part
  using CBC = std::integral_constant<int, BG + AG - PA>;     // the next patch's second and third parts, the weights of t+2
  using Y = std::true_type;
  using N = std::false_type;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;   // the next patch's second part is issued in this k-tile
  using P2 = std::integral_constant<int, 2>;   // ... its third part
  using P3 = std::integral_constant<int, 3>;   // the first part of the patch after the next one (if there is one)
#ifdef M2H_CLOCK_DIAG
  unsigned long long dbg_st[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
  const int total = my_tiles * NS;   // patches of this workgroup
  bool st = false;                   // the previous tile's stores are in flight, all PST of them issued by every lane
  int s_in = 0;                      // patch inside the tile
  for (int gs = 0; gs + 1 < total; ++gs) {
    more_patches = gs + 2 < total;
#ifdef M2H_CLOCK_DIAG
    const bool dbg_b = gs == NS;   // the first patch behind the first tile boundary
    if (dbg_b) dbg_st[2] = __builtin_amdgcn_s_memtime();
#endif
    body(T0{}, CA{}, st, Y{}, P1{});    // in flight at the barrier: the weights of t+2, the next patch's first part
#ifdef M2H_CLOCK_DIAG
    if (dbg_b) dbg_st[3] = __builtin_amdgcn_s_memtime();
#endif
    body(T1{}, CAB{}, st, Y{}, P2{});   // ... the weights of t+2, the first and the second part
#ifdef M2H_CLOCK_DIAG
    if (dbg_b) dbg_st[4] = __builtin_amdgcn_s_memtime();
#endif
    st = false;
    body(T2{}, CBC{}, false, Y{}, P0{});   // the second part, the weights of t+2, the third part
#ifdef M2H_CLOCK_DIAG
    if (dbg_b) dbg_st[5] = __builtin_amdgcn_s_memtime();
#endif
    body(T3{}, CW{}, false, Y{}, P3{});   // the weights of t+2; the whole next patch (older than the awaited weights) has landed
#ifdef M2H_CLOCK_DIAG
    if (dbg_b) dbg_st[6] = __builtin_amdgcn_s_memtime();
#endif
    if (++s_in == NS) {   // the tile is complete (and another one follows)
      s_in = 0;
#ifdef M2H_CLOCK_DIAG
      if (gs + 1 == NS) dbg_st[0] = __builtin_amdgcn_s_memtime();
#endif
      store_tile(c_m0, c_n0, c_phase, std::false_type{});
#ifdef M2H_CLOCK_DIAG
      if (gs + 1 == NS) dbg_st[1] = __builtin_amdgcn_s_memtime();
#endif
      st = p.S == 1 && c_m0 + BM <= p.M;   // (a ragged tile: some lanes skip their stores, the count is unknown -> the plain waits)
      c_L += G;
      c_m0 = n_m0;
      c_n0 = n_n0;
      c_phase = n_phase;
      if (c_L + G < ntiles) tile_of(c_L + G, n_m0, n_n0, n_phase);
    }
  }
  body(T0{}, CW{}, false, Y{}, P0{});     // the workgroup's last patch: the last weight tile is issued here (stores of the tile before: waited for)
  body(T1{}, CW{}, false, N{}, P0{});
  body(T2{}, C0{}, false, N{}, P0{});
  load_a(ab, std::integral_constant<int, 3>{}, IH{}, IF{});
  for_ni([&](auto nic) { mfma_col(I0{}, IF{}, nic); });
  store_tile(c_m0, c_n0, c_phase, std::true_type{});

#ifdef M2H_CLOCK_DIAG
  if (tid == 0 && blockIdx.x < 8192) {
    g_clock_dbg_patch[blockIdx.x][0] = __builtin_amdgcn_s_memtime() - dbg_t0;
    g_clock_dbg_patch[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    g_clock_dbg_patch[blockIdx.x][2] = dbg_s0;
    g_clock_dbg_patch[blockIdx.x][3] = dbg_r0;
    g_clock_dbg_patch[blockIdx.x][4] = __builtin_amdgcn_s_memrealtime();
    g_clock_dbg_patch[blockIdx.x][5] = __builtin_amdgcn_s_memrealtime();
    g_clock_dbg_patch[blockIdx.x][6] = (unsigned long long)my_tiles;
    for (int i = 0; i < 7; ++i) g_clock_dbg_patch[blockIdx.x][8 + i] = dbg_st[i];
  }
#endif
}

// workgroups of a persistent launch: one per CU (LDS: one workgroup of this kernel per CU), a multiple of 8 so that a workgroup's
// tiles stay on its XCD's m-tiles
static int patch_grid_limit() {
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    cus[dev] = n / 8 * 8;
  }
  return cus[dev];
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
  // every tile index is a tile (MT a multiple of 8): a workgroup per CU walks them; otherwise one workgroup per index
  // m2h_tuning_set 36 = 8: one workgroup per tile (A/B); m2h_tuning_set 10 = n >= 8: n workgroups (a multiple of 8; tests: other tile sequences per
  // workgroup than this chip's CU count gives -- phase and n-tile changing from one tile of a workgroup to its next, many tiles per workgroup)
  const long lim = g_patch == 8 ? nblk : (g_patch_grid >= 8 ? g_patch_grid / 8 * 8 : patch_grid_limit());
  const long gx = (p.MT % 8 == 0 && nblk > lim) ? lim : nblk;
  const dim3 grid((unsigned)gx, (unsigned)S), blk(64 * PNW);
  const int ntiles = (int)nblk;
#ifdef M2H_CLOCK_DIAG
  if (p.convT) M2H_LAUNCH((igemm_patch_kernel<WM, WN, WHOLE, 1, 0>), grid, blk, 0, st, p, g, ntiles);   // (the diagnostic variants are built for convs)
  else if (g_patch == 4) M2H_LAUNCH((igemm_patch_kernel<WM, WN, WHOLE, 0, 4>), grid, blk, 0, st, p, g, ntiles);
  else if (g_patch == 5) M2H_LAUNCH((igemm_patch_kernel<WM, WN, WHOLE, 0, 5>), grid, blk, 0, st, p, g, ntiles);
  else if (g_patch == 6) M2H_LAUNCH((igemm_patch_kernel<WM, WN, WHOLE, 0, 6>), grid, blk, 0, st, p, g, ntiles);
  else if (g_patch == 7) M2H_LAUNCH((igemm_patch_kernel<WM, WN, WHOLE, 0, 7>), grid, blk, 0, st, p, g, ntiles);
  else
#endif
  if (p.hi_only && p.convT) M2H_LAUNCH((igemm_patch_kernel<WM, WN, WHOLE, 1, 9>), grid, blk, 0, st, p, g, ntiles);
  else if (p.hi_only) M2H_LAUNCH((igemm_patch_kernel<WM, WN, WHOLE, 0, 9>), grid, blk, 0, st, p, g, ntiles);
  else if (p.convT) M2H_LAUNCH((igemm_patch_kernel<WM, WN, WHOLE, 1, 0>), grid, blk, 0, st, p, g, ntiles);
  else M2H_LAUNCH((igemm_patch_kernel<WM, WN, WHOLE, 0, 0>), grid, blk, 0, st, p, g, ntiles);
  return launch_status(BN == 64 ? "igemm_patch<512,64>" : S == 2 ? "igemm_patch<256,128> (two K-halves)" : "igemm_patch<256,128>");
}

// Shapes: what launch_igemm_dma takes at split-K 1 (plus 64-wide layers), restricted to the two window geometries above with the
// whole window reached and power-of-two pixel grids: whole images per tile (any width), or -- 256 x 128 tile -- some rows of an
// image 16 / 32 / 64 pixels wide with a patch of at most 384 rows.  Returns -2 otherwise (the caller falls through).
int launch_igemm_patch(IGemmP& p, size_t ws_bytes, hipStream_t st) {
  if (g_patch < 0 || p.math != 1 || !p.presplit || !p.fast_ok || p.head_w != nullptr || p.N % 64 != 0 || p.N > (p.N % 128 == 0 ? 1024 : 448) || p.Kw != p.K || !(p.slope >= 0.f && p.slope <= 1.f)) return -2;
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
  if ((g_patch < 2 || g_patch >= 8) && tiles < 224) {
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
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_clock_dbg_patch), (size_t)nblocks * 16 * sizeof(unsigned long long));
}
#endif

}  // namespace m2h
