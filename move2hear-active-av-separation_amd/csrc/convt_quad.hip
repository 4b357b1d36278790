// Four-phase transposed-conv kernel (gfx950, bf16x3 math, split32 operands): the narrow late decoder stages of the separator
// U-Nets -- ConvTranspose2d(4, 2, 1) with 64, 32 or 16 output channels (separator_cnn.py:46-52,128-135) -- which are half of the
// benchmark step's time.
//
// A transposed 4x4 / stride-2 conv is four 2x2-tap stride-1 convs (the sub-pixel phases), and all sixteen (phase, tap) pairs
// read the SAME input pixels: pixel (q + dy, r + dx), dy, dx in {-1, 0, 1}, of the 3x3 neighbourhood of output block (q, r).
// The per-phase kernels (convT_tap_kernel, igemm_f32_kernel) stage that neighbourhood once per phase -- four times per output
// block -- and are bound by that operand stream (PMC: matrix pipe 24 % busy).  Here a workgroup owns 256 positions (R = 256 / Wq
// image rows x Wq columns) for ALL FOUR phases (4 x the accumulators, which the LDS-DMA staging leaves room for):
//   * per 16-channel half-chunk the (R+2) x (Wq+2)-pixel patch is staged ONCE (64-byte rows: [hi 16 | lo 16] bf16), by
//     global_load_lds_dwordx4 into one of three patch buffers, two half-chunks ahead of its use;
//   * the weights stream through a three-stage ring of 16 KB pieces (256 rows x 64 B = 64 / BN phases x 4 taps x BN channels),
//     one piece = one k-tile of 24 v_mfma_f32_32x32x16_bf16 per wave;
//   * every (phase, tap) reads its A fragments from a row / column shift of the patch.
// L2 -> LDS bytes per output block and half-chunk: 33 KB of patch + 64 KB of weights for 4 x 256 x BN outputs, against
// 4 x (25 + 16) KB for the same outputs in the per-phase kernel at BN = 32.
// Pipeline, waits and the bank-conflict-free row permutation follow conv_dma.hip (pieces of a 64-byte row r sit at
// piece ^ ((r >> 2) & 3)); the epilogue is fused_epilogue (incl. the 1x1 head and the de-sliced store), run once per phase.
// Measured (B = 256): N = 64 243 -> 224 us (two 32-wide n-tiles), N = 32 + head 334 -> 305 us, N = 16 + head (padded to 32
// columns) 254 -> 290 us: used for N in (16, 64].
// Requires: conv_transpose, split32 operands, C0 / C1 multiples of 32, 256 % Wq == 0, 32 <= Wq <= 128, Hq % (256 / Wq) == 0,
// N <= 64.
#include "igemm_common.h"

namespace m2h {

// (tuning knob g_quad: thread-local, m2h_internal.h) m2h_tuning_set 30: -1 never use this kernel; 1 = wherever its shape conditions hold (also N <= 16 and few output blocks)

extern __device__ float g_zero_page_quad[];
__device__ __attribute__((aligned(128))) float g_zero_page_quad[2048 + 32];

#ifndef M2H_QUAD_DBG
#define M2H_QUAD_DBG 0   // diagnostic builds: 1 no MFMAs, 2 no DMA waits, 3 no fragment reads
#endif
#ifdef M2H_CLOCK_DIAG
// Diagnostic build only (tools/clock_diag_quad.py): 100 MHz real-time stamps at the block's milestones.
__device__ unsigned long long g_clock_dbg_quad[4096][8];
#define QSTAMP(i) do { if (tid == 0 && blockIdx.x < 4096) g_clock_dbg_quad[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define QSTAMP(i) do { } while (0)
#endif

namespace {

template <int CNT>
__device__ __forceinline__ void glds16_seq(const char* const* src, const unsigned* dst) {
  // CNT LDS-DMA loads of 16 bytes per lane with independent LDS destinations (wave-uniform); M0 saved / restored inside
  unsigned keep;
  static_assert(CNT >= 1 && CNT <= 3, "load count");
  if constexpr (CNT == 1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src[0]), "s"(dst[0])
                 : "memory");
  else if constexpr (CNT == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src[0]), "v"(src[1]), "s"(dst[0]), "s"(dst[1])
                 : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src[0]), "v"(src[1]), "v"(src[2]), "s"(dst[0]), "s"(dst[1]), "s"(dst[2])
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (M2H_QUAD_DBG == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace

template <int BN, bool HEAD>   // HEAD: the fused 1x1 head's epilogue (pixels as MFMA rows); without it the accumulators are transposed
__global__ __launch_bounds__(512, 1) void convT_quad_kernel(const IGemmP p) {
  constexpr int BM = 256, NW = 8;
  constexpr int WN = BN / 32, WM = NW / WN;      // 64 wide: 4 x 2 waves of 64 x 32; 32 wide: 8 x 1 waves of 32 x 32
  constexpr int TM = BM / WM, FM = TM / 32;
  constexpr int PPT = 64 / BN;                   // phases per k-tile (a weight piece is 256 rows = PPT x 4 taps x BN channels)
  constexpr int TPH = 4 / PPT;                   // k-tiles per half-chunk
  constexpr int NST = 3;                         // ring depth (weights) and patch buffers
  constexpr int PROWS = 528;                     // patch rows incl. DMA padding: (R+2)(Wq+2) <= 520 for Wq in {32, 64, 128}
  constexpr int PATCH_BYTES = PROWS * 64, B_BYTES = 256 * 64;
  constexpr int AMAX = (33 + 8 * TPH - 1) / (8 * TPH);   // patch DMA instructions per wave and k-tile (33 x 16 rows at most)
  using AccT = f32x16;
  static_assert(BN == 32 || BN == 64, "tile width");

  __shared__ __attribute__((aligned(1024))) char s_patch[NST * PATCH_BYTES];
  __shared__ __attribute__((aligned(1024))) char s_w[NST * B_BYTES];
  __shared__ __attribute__((aligned(1024))) char s_dummy[1024];
  __shared__ int ri_out[BM], ri_bc[BM];
  const unsigned lds_dummy = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)s_dummy;
  const unsigned lds_patch = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)s_patch;
  const unsigned lds_w = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)s_w;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 31, half = lane >> 5;

  // ---- block -> output block rows: contiguous runs of m-tiles per XCD (neighbouring tiles share their halo rows in one L2) ----
  // and the n-tiles of one output block consecutive on that XCD (they stage the same patch)
  const int per = (p.MT + 7) >> 3;
  const int idx = blockIdx.x >> 3;
  const int nt = idx % p.NT;
  const int mt = (blockIdx.x & 7) * per + idx / p.NT;
  if (idx / p.NT >= per || mt >= p.MT) return;
  const int m0 = mt * BM, n0 = nt * BN;
  QSTAMP(0);
  const int Wq = p.Wq, PW = Wq + 2;
  const int R = BM / Wq;
  const int P = (R + 2) * PW, TP = (P + 15) >> 4;
  const int b0 = m0 / (p.Hq * Wq);
  const int q0 = (m0 / Wq) % p.Hq;

  // ---- patch DMA slots of this wave: slot (i, a) = instruction x = (a TPH + i) 8 + wave, rows 16 x .. 16 x + 15 ----
  int slot_pix[TPH * AMAX];     // input pixel index (b, ih, iw) of this lane's row, -1 outside the image / past the patch
  const char* zero = reinterpret_cast<const char*>(g_zero_page_quad);
#pragma unroll
  for (int i = 0; i < TPH; ++i)
#pragma unroll
    for (int a = 0; a < AMAX; ++a) {
      const int x = (a * TPH + i) * 8 + wave;
      const int l = 16 * x + (lane >> 2);
      const int pr = l / PW, pc = l - pr * PW;
      const int ih = q0 - 1 + pr, iw = pc - 1;
      const bool ok = l < P && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
      slot_pix[i * AMAX + a] = ok ? (b0 * p.Hi + ih) * p.Wi + iw : -1;
    }
  // byte of this lane's piece inside a half-chunk's [hi | lo] pair (without the half's 32 h): LDS piece (lane & 3) of row l holds
  // logical piece (lane & 3) ^ ((l >> 2) & 3), and (l >> 2) & 3 = (lane >> 4) & 3 for every instruction (l = 16 x + (lane >> 2))
  const int slot_byte = (((lane & 3) ^ ((lane >> 4) & 3)) & 1) * 16 + ((((lane & 3) ^ ((lane >> 4) & 3)) >> 1) * 64);
  // ---- weight DMA: instruction jb of a k-tile covers piece rows (2 wave + jb) 16 .. + 15; row = (pl 4 + tap) BN + n ----
  unsigned w_off[2];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    const int rb = (wave * 2 + jb) * 16 + (lane >> 2);
    const int item = rb / BN, n = min(n0 + rb - item * BN, p.N - 1);   // rows past N re-read row N-1 (never stored)
    const int pl = item >> 2, tap = item & 3;
    const int pz = (lane & 3) ^ ((rb >> 2) & 3);
    w_off[jb] = ((unsigned)(pl * p.N + n) * (unsigned)p.K + (unsigned)(tap * p.Ctot)) * 4u + (unsigned)((pz & 1) * 16 + (pz >> 1) * 64);
  }

  const int NHC = 2 * (p.Ctot / 32);     // half-chunks
  const int nk = NHC * TPH;              // k-tiles
  // issue the patch part `slot i` of half-chunk hc into patch buffer hc % 3; returns the DMA instructions issued (wave-uniform)
  // Patch part `slot i` of half-chunk hc -> patch buffer hc % 3: always AMAX instructions, so that the DMA count per k-tile is a
  // compile-time constant for the counted waits below; an instruction past the patch (x >= TP) copies zeros into a dummy KiB.
  auto issue_patch = [&](int hc, auto ic) {
    constexpr int i = decltype(ic)::value;
    const int chunk = hc >> 1, h = hc & 1;
    const bool second = chunk * 32 >= p.C0 && p.src1 != nullptr;
    const int Cs = second ? p.C1 : p.C0;
    const char* base = reinterpret_cast<const char*>(second ? p.src1 : p.src0) + (size_t)((second ? chunk * 32 - p.C0 : chunk * 32) * 4 + h * 32);
    const unsigned dbase = lds_patch + (unsigned)(hc % NST) * PATCH_BYTES;
    const char* src[AMAX];
    unsigned dst[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; ++a) {
      const int x = (a * TPH + i) * 8 + wave;
      const int s = i * AMAX + a;
      src[a] = (slot_pix[s] >= 0 ? base + (size_t)(unsigned)slot_pix[s] * (unsigned)Cs * 4u : zero) + slot_byte;
      dst[a] = x < TP ? dbase + (unsigned)x * 1024u : lds_dummy;
    }
    glds16_seq<AMAX>(src, dst);
  };
  auto issue_weights = [&](int t) {   // k-tile t -> ring stage t % 3
    const int hc = t / TPH, jt = t - hc * TPH;
    const size_t uni = ((size_t)(jt * PPT) * p.N * p.K) * 4 + (size_t)((hc >> 1) * 128 + (hc & 1) * 32);
    const char* src[2] = {reinterpret_cast<const char*>(p.w) + uni + w_off[0], reinterpret_cast<const char*>(p.w) + uni + w_off[1]};
    const unsigned d0 = lds_w + (unsigned)(t % NST) * B_BYTES + (unsigned)wave * 2048u;
    const unsigned dst[2] = {d0, d0 + 1024u};
    glds16_seq<2>(src, dst);
  };

  // ---- fragment addresses ----
  // A: patch row l = (qi + 1 + dy) PW + (r + 1 + dx) of position (qi, r); bytes l 64 + ((piece ^ ((l >> 2) & 3)) << 4), piece =
  // half (hi) / 2 + half (lo: the hi address ^ 32).  Nine shifts (dy, dx) serve the sixteen (phase, tap) pairs.
  int a_addr[FM][9];
#pragma unroll
  for (int mi = 0; mi < FM; ++mi) {
    const int ml = wm * TM + mi * 32;
    const int qi = ml / Wq, r0 = ml - qi * Wq;
#pragma unroll
    for (int s9 = 0; s9 < 9; ++s9) {
      const int l = (qi + 1 + (s9 / 3 - 1)) * PW + (r0 + frow + 1 + (s9 % 3 - 1));
      a_addr[mi][s9] = l * 64 + ((half ^ ((l >> 2) & 3)) << 4);
    }
  }
  const int b_addr = (wn * 32 + frow) * 64 + ((half ^ ((frow >> 2) & 3)) << 4);   // + item BN 64 (items start at multiples of 32 rows)
  const int b_addr_lo = b_addr ^ 32;

  AccT acc[4][FM][1];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int mi = 0; mi < FM; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ph][mi][0][e] = 0.f;

  auto mfma = [&](const f32x4& a, const f32x4& b, AccT& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };
  // A granule = 6 MFMAs per wave = tap g of BOTH phases of the k-tile (phases 2 jt and 2 jt + 1; jt and g compile-time): the two
  // accumulation chains alternate, so no MFMA waits for the one issued just before it.  (Two taps of ONE phase per granule,
  // twelve dependent MFMAs in a row, ran the k-loop at 2400 instead of ~1600 cycles per tile even without any LDS read.)
  static_assert(FM == 1 && PPT == 2, "granules are written for the 32-wide tile");
  struct Gran {
    f32x4 ah[2], al[2], bh[2], bl[2];   // index: phase of the pair
  };
  auto load_gran = [&](int pbuf, int wstage, auto jtc, auto gc, Gran& f) {
    if constexpr (M2H_QUAD_DBG == 3) return;
    constexpr int jt = decltype(jtc)::value, tap = decltype(gc)::value;
    const char* sp = s_patch + pbuf * PATCH_BYTES;
    const char* sw = s_w + wstage * B_BYTES;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const int phase = jt * PPT + pl;
      const int dy = (tap >> 1) * (2 * (phase >> 1) - 1), dx = (tap & 1) * (2 * (phase & 1) - 1);
      const int s9 = (dy + 1) * 3 + (dx + 1);
      const int item = pl * 4 + tap;
      f.bh[pl] = *reinterpret_cast<const f32x4*>(sw + item * BN * 64 + b_addr);
      f.bl[pl] = *reinterpret_cast<const f32x4*>(sw + item * BN * 64 + b_addr_lo);
      f.ah[pl] = *reinterpret_cast<const f32x4*>(sp + a_addr[0][s9]);
      f.al[pl] = *reinterpret_cast<const f32x4*>(sp + (a_addr[0][s9] ^ 32));
    }
  };
  // MFMAs lo .. hi-1 (of 6) of a granule: products lo*hi, hi*lo, hi*hi of phase 0 / phase 1 alternately
  auto mfma_gran = [&](auto jtc, auto gc, const Gran& f, auto loc, auto hic) {
    constexpr int jt = decltype(jtc)::value;
#pragma unroll
    for (int u = decltype(loc)::value; u < decltype(hic)::value; ++u) {
      const int pl = u & 1, prod = u >> 1;
      AccT& c = acc[jt * PPT + pl][0][0];
      if constexpr (M2H_QUAD_DBG == 1) continue;
      if constexpr (HEAD) {
        if (prod == 0) mfma(f.al[pl], f.bh[pl], c);
        else if (prod == 1) mfma(f.ah[pl], f.bl[pl], c);
        else mfma(f.ah[pl], f.bh[pl], c);
      } else {   // the weights as the A operand: a lane holds runs of four channels of one pixel (igemm_common.h nhwc_tile_store_T)
        if (prod == 0) mfma(f.bh[pl], f.al[pl], c);
        else if (prod == 1) mfma(f.bl[pl], f.ah[pl], c);
        else mfma(f.bh[pl], f.ah[pl], c);
      }
    }
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>;
  using C3 = std::integral_constant<int, 3>;
  using C6 = std::integral_constant<int, 6>;
  // ---- pipeline (see conv_dma.hip): fragment reads one granule ahead; per k-tile one wait + barrier before the last granule ----
  // Iteration t issues, behind its barrier, the weights of k-tile t + 3 (into the stage of tile t, whose reads are complete at the
  // barrier) and, on the first tile of a half-chunk, the whole patch of half-chunk hc + 2 (its buffer was last read in half-chunk
  // hc - 1).  DMA completes in issue order, so the weights go first: the wait of iteration t needs tile t+1's weights (iteration
  // t-2) and leaves everything younger in flight -- a patch and one tile's weights, whichever of the two iterations issued the
  // patch; the patch of half-chunk hc(t+1) is at least three iterations old and so older than those weights.  Every flag of a
  // tile is a compile-time constant: the loop over the half-chunks that still issue both runs straight-line code, the last two
  // half-chunks are peeled.
  auto patch_all = [&](int hc) {
    auto parts = [&](auto self, auto ic) -> void {
      if constexpr (decltype(ic)::value < TPH) {
        issue_patch(hc, ic);
        self(self, std::integral_constant<int, decltype(ic)::value + 1>{});
      }
    };
    parts(parts, C0{});
  };
  patch_all(0);
  patch_all(1);                       // NHC >= 2
#pragma unroll
  for (int d = 0; d < NST; ++d)
    if (d < nk) issue_weights(d);
  QSTAMP(1);
  wait_vm_barrier<0>();
  QSTAMP(2);
  Gran fA, fB;
  load_gran(0, 0, C0{}, C0{}, fA);
  int t = 0, pbuf = 0;
  // one k-tile: jt (position in its half-chunk), WAITN (DMA instructions of the previous iteration that may stay in flight; -1: last
  // tile, no successor), ISSUE_P / ISSUE_W (this iteration issues a patch part / weights)
  auto tile = [&](auto jtc, auto waitn, auto issue_p, auto issue_w) {
    constexpr int jt = decltype(jtc)::value, WAITN = decltype(waitn)::value;
    using JN = std::integral_constant<int, (jt + 1) % TPH>;
    const int ws = t % NST;
    const int hc = t / TPH;
    load_gran(pbuf, ws, jtc, C1{}, fB);
    __builtin_amdgcn_sched_barrier(0);
    mfma_gran(jtc, C0{}, fA, C0{}, C6{});
    __builtin_amdgcn_sched_barrier(0);
    load_gran(pbuf, ws, jtc, C2{}, fA);
    __builtin_amdgcn_sched_barrier(0);
    mfma_gran(jtc, C1{}, fB, C0{}, C6{});
    __builtin_amdgcn_sched_barrier(0);
    load_gran(pbuf, ws, jtc, C3{}, fB);
    __builtin_amdgcn_sched_barrier(0);
    mfma_gran(jtc, C2{}, fA, C0{}, C6{});
    if constexpr (WAITN >= 0) {
      wait_vm_barrier<WAITN>();
      const int npbuf = jt + 1 == TPH ? (pbuf + 1 == NST ? 0 : pbuf + 1) : pbuf;
      load_gran(npbuf, ws + 1 == NST ? 0 : ws + 1, JN{}, C0{}, fA);
      __builtin_amdgcn_sched_barrier(0);
      mfma_gran(jtc, C3{}, fB, C0{}, C3{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (decltype(issue_w)::value) issue_weights(t + NST);          // weights first: their wait must not queue behind
      if constexpr (decltype(issue_p)::value) patch_all(hc + 2);               // the patch's first-touch (HBM) loads
      __builtin_amdgcn_sched_barrier(0);
      mfma_gran(jtc, C3{}, fB, C3{}, C6{});
      __builtin_amdgcn_sched_barrier(0);
      pbuf = npbuf;
    } else {
      __builtin_amdgcn_sched_barrier(0);
      mfma_gran(jtc, C3{}, fB, C0{}, C6{});
    }
    ++t;
  };
  using T = std::true_type;
  using F = std::false_type;
  constexpr int NFULL = TPH * AMAX + 2;   // behind the weights waited for: one whole patch and the next tile's weights
  using WF = std::integral_constant<int, NFULL>;
  using W2 = std::integral_constant<int, 2>;
  using WL = std::integral_constant<int, -1>;
  static_assert(TPH == 2, "the peeled tail below is written for two k-tiles per half-chunk");
  for (int hc = 0; hc + 2 < NHC; ++hc) {
    tile(C0{}, WF{}, T{}, T{});   // the whole patch of half-chunk hc + 2 behind the first tile: three tiles to land
    tile(C1{}, WF{}, F{}, T{});
  }
  tile(C0{}, WF{}, F{}, T{});   // t = nk-4: its weights (tile nk-1) are the last DMA
  tile(C1{}, W2{}, F{}, F{});   // t = nk-3: only those two weight loads may be in flight
  tile(C0{}, C0{}, F{}, F{});   // t = nk-2: everything has landed
  tile(C1{}, WL{}, F{}, F{});   // t = nk-1

  QSTAMP(3);
  // ---- epilogue: one pass of the fused epilogue per phase (row bookkeeping of that phase; the patch buffers are its scratch) ----
  // row bookkeeping once per block (phase (0, 0)); phase (ph, pw) writes the pixel ph rows / pw columns further (decode_row's
  // out = ... + (2 q + ph) Wo + 2 r + pw in both output layouts; the border class is not used by transposed stages)
  int ri_base[1];
  __syncthreads();
  if (tid < BM) {
    int q, rr, b, out, bc;
    decode_row(p, m0 + tid, 0, 0, q, rr, b, out, bc);
    ri_base[0] = out;
    ri_bc[tid] = bc;
  }
  auto phase_out = [&](auto self, auto phc) -> void {
    constexpr int ph = decltype(phc)::value;
    if constexpr (ph < 4) {
      if constexpr (ph > 0) __syncthreads();     // the previous phase is done with the row table
      if (tid < BM) ri_out[tid] = ri_base[0] + (ph >> 1) * p.Wo + (ph & 1);
      __syncthreads();
      if constexpr (HEAD)
        fused_epilogue<BM, BN, WM, WN, 32, AccT, NST * PATCH_BYTES>(p, acc[ph], reinterpret_cast<float*>(s_patch), reinterpret_cast<float*>(s_patch) + BM * LDK + 64,
                                                   ri_out, ri_bc, n0, tid, ph == 0);
      else
        nhwc_tile_store_T<BM, BN, WM, WN, 32, NST * PATCH_BYTES, AccT>(p, acc[ph], s_patch, ri_out, n0, tid);
      QSTAMP(4 + ph);
      self(self, std::integral_constant<int, ph + 1>{});
    }
  };
  phase_out(phase_out, C0{});
}

int launch_convT_quad(IGemmP& p, hipStream_t st) {
  if (g_quad < 0 || !p.convT || p.math != 1 || !p.presplit || !p.fast_ok || p.N > 64) return -2;
  if (p.Wq < 32 || p.Wq > 128 || 256 % p.Wq != 0 || p.Hq % (256 / p.Wq) != 0 || p.Ctot % 32 != 0 || p.M % 256 != 0) return -2;
  if (p.head_w != nullptr && p.N > 32) return -2;      // the fused head lives on the 32-wide tile
  if (g_quad == 0) {
    if (p.N <= 16) return -2;                                            // padded to 32 columns: 290 vs 254 us on the tap-sharing kernel
    if ((long)p.M * ((p.N + 31) / 32) < 256L * 224) return -2;           // too few output blocks to fill the chip
  }
  p.MT = p.M / 256;
  p.NT = (p.N + 31) / 32;   // 64 wide: two 32-wide n-tiles per output block (four phases x 64 x 32 accumulators per wave do not fit)
  p.S = 1;
  const int per = (p.MT + 7) / 8;
  const dim3 grid((unsigned)(per * 8 * p.NT)), blk(512);
  // without a head the tile leaves through nhwc_tile_store_T: whole NHWC rows, 16-byte pieces
  const bool plain = p.head_w == nullptr && p.out_mode == M2H_OUT_NHWC && p.N % 4 == 0 && p.ldc % 4 == 0 && (reinterpret_cast<size_t>(p.dst) & 15) == 0;
  if (plain) M2H_LAUNCH((convT_quad_kernel<32, false>), grid, blk, 0, st, p);
  else M2H_LAUNCH((convT_quad_kernel<32, true>), grid, blk, 0, st, p);
  return launch_status(p.N > 32 ? "igemm_convT_quad<64>" : "igemm_convT_quad<32>");
}

#ifdef M2H_CLOCK_DIAG
extern "C" int m2h_diag_read_clocks_quad(unsigned long long* host_out, int nblocks) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_clock_dbg_quad), (size_t)nblocks * 8 * sizeof(unsigned long long));
}
#endif

}  // namespace m2h
