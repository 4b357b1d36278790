// Shared-patch LDS-DMA engine (gfx950, bf16x3 math, split32 operands): the 256 x 128 tile of conv_dma.hip for the two layer
// shapes of the separator U-Nets' wide stages -- Conv2d(4, 2, 1) (separator_cnn.py:5-13,46-52) and one phase of
// ConvTranspose2d(4, 2, 1) (separator_cnn.py:15-24,128-135) -- with the pixel operand staged ONCE per four taps.
//
// Why.  In the GEMM view of conv_dma.hip every (tap, 32-channel chunk) k-tile stages its own 256 pixel rows (32 KB) beside the
// 128 weight rows (16 KB).  But the four taps of one parity class of the 4x4/s2 window -- (kh, kw), (kh, kw+2), (kh+2, kw),
// (kh+2, kw+2) -- read the SAME input pixels one output column / row apart, and so do the four taps of a transposed-conv phase:
//     conv   tap (2a + gh, 2b + gw) of output (oh, ow)  reads  P[oh + a][ow + b],  P[i][j] = in(2i + gh - 1, 2j + gw - 1)
//     convT  tap (th, tw) of phase (ph, pw), pixel (q, r) reads  P[q + a][r + b],    P[i][j] = in(i + ph - 1, j + pw - 1),
//                                                              a = ph ? th : 1 - th,  b = pw ? tw : 1 - tw
// So a block stages the (rows + 1) x (W + 1) patch P of its 256 output pixels once per (class, chunk) -- 297 to 340 rows of
// 128 B instead of 4 x 256 -- and the four taps read their fragments from it at a row shift a (W + 1) + b.  Measured on the
// old engine with three of four pixel-row DMAs dropped (tools/clock_diag_dma.py, knob 7): the operand stream into the CU, not the
// matrix pipe, was what the k-tile waited for (2 390 -> 2 236 cycles per k-tile and a 5 % higher clock under the lighter
// memory traffic); every re-read of an input line now comes from LDS instead of from beyond L2.
//
// LDS: two patch buffers of 48 KB (384 rows: six 8-row DMA groups per wave, rows past the patch copy zeros) + a ring of three
// 16 KB weight stages = 144 KB; rows unpadded with the piece permutation of conv_dma.hip (LDS piece j of row r holds split32
// piece j ^ ((r >> 1) & 7)): a 16-lane group of a ds_read_b128 reads 16 CONSECUTIVE patch rows at any shift (fragments start at
// multiples of 16 output columns and W is a multiple of 16, so a fragment never crosses the end of a patch row), which is the
// conflict-free case of that permutation.
//
// Pipeline: as conv_dma.hip's 16x16x32 path (fragment reads half a tile ahead, one barrier in the middle of each k-tile,
// counted vmcnt waits, straight-line steady state), unrolled over the four taps of a patch: the weights of tile t+3 are issued in
// tile t, the patch of class/chunk s+1 in the first tile of s (behind that tile's weights, so the counts are compile-time:
// 2, 8, 8, 2 loads may stay in flight at the four barriers).
// k order: (class, chunk, tap) -- another summation order than the (tap, chunk) of the other engines (rel-L1 ~1e-6 between them).
#include "igemm_common.h"
#include "lds_dma.h"

namespace m2h {

int g_patch = 0;   // m2h_debug_set 36: -1 never use this engine; 2 = also below its tile-count threshold (tests)

__device__ __attribute__((aligned(128))) float g_zero_page_patch[2048 + 32];   // 8 KiB + one row: source of padding rows at any channel offset

namespace {

constexpr int PBM = 256, PBN = 128, PWM = 4, PWN = 2, PNW = PWM * PWN;
constexpr int PA_ROWS = 384, PA_BYTES = PA_ROWS * 128, PB_BYTES = PBN * 128, PNSTB = 3;
constexpr int PSMEM = 2 * PA_BYTES + PNSTB * PB_BYTES;

struct PatchGeo {
  int W1;         // patch row length W + 1
  int seg_rows;   // patch rows of one segment: (rows + 1) (W + 1)
  int nseg;       // segments (images) per tile
  int w_sh;       // log2 W
  int seg_sh;     // log2 of the output pixels per segment
};

}  // namespace

__global__ __launch_bounds__(64 * PNW, 1) void igemm_patch_kernel(const IGemmP p, const PatchGeo g) {
  constexpr int FM = 4, FN = 4, AG = PA_ROWS / (8 * PNW), BG = PBN / (8 * PNW);
  static_assert(AG == 6 && BG == 2, "DMA groups per wave");
  __shared__ __attribute__((aligned(1024))) char smem[PSMEM];
  __shared__ int ri_out[PBM];
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / PWN, wn = wave % PWN;
  const int lrow = lane >> 3;
  const int frow = lane & 15, half = lane >> 4;

  // ---- block -> (m-tile, n-tile, phase): as igemm_dma_kernel ----
  const int L = blockIdx.x;
  const int xcd = L & 7;
  int idx = L >> 3;
  int phase = 0;
  if (p.convT) {
    phase = idx & 3;
    idx >>= 2;
  }
  const int mt = (idx / p.NT) * 8 + xcd;
  const int nt = idx - (idx / p.NT) * p.NT;
  if (mt >= p.MT) return;
  const int m0 = mt * PBM, n0 = nt * PBN;
  const int ph = p.convT ? phase >> 1 : 0, pw = p.convT ? phase & 1 : 0;
  const float* wbase = p.w + (p.convT ? (size_t)phase * p.N * p.K : 0);

  if (tid < PBM) {
    const int m = m0 + tid;
    int q, rr, b, out = -1, bc;
    if (m < p.M) decode_row(p, m, ph, pw, q, rr, b, out, bc);
    ri_out[tid] = out;
  }

  // ---- the patch rows this lane feeds (fixed for the whole kernel) ----
  const int b0 = m0 >> (g.w_sh + p.hq_sh);
  const int q0 = (m0 >> g.w_sh) & (p.Hq - 1);   // 0 when a tile holds whole images
  const int sm = p.convT ? 1 : 2;
  int a_ih[AG], a_iw[AG], a_pix[AG], pieceA[AG];
#pragma unroll
  for (int i = 0; i < AG; ++i) {
    const int grp = wave + PNW * i;
    const int pr = grp * 8 + lrow;
    const int seg = pr / g.seg_rows;
    const int w = pr - seg * g.seg_rows;
    const int ii = w / g.W1, jj = w - ii * g.W1;
    const int b = b0 + seg;
    a_ih[i] = (q0 + ii) * sm;
    a_iw[i] = jj * sm;
    a_pix[i] = (seg < g.nseg && b < p.B) ? b * p.Hi * p.Wi : -1;
    pieceA[i] = ((lane & 7) ^ (((grp & 1) << 2) | (lrow >> 1))) * 16;
  }
  const char* zero = reinterpret_cast<const char*>(g_zero_page_patch);
  const char* ptrA[AG];
  const char* ptrB[BG];
#pragma unroll
  for (int j = 0; j < BG; ++j) {
    const int grp = wave + PNW * j;
    const int r = grp * 8 + lrow;
    const int piece = (lane & 7) ^ (((grp & 1) << 2) | (lrow >> 1));
    ptrB[j] = reinterpret_cast<const char*>(wbase) + ((size_t)min(n0 + r, p.N - 1) * p.K) * 4 + piece * 16;
  }

  // ---- the two operand streams (uniform state) ----
  // patches: s = 0 .. NS-1 in (class, chunk) order (conv: class = (gh, gw) of the window; transposed conv: one class, the chunks
  // of both concatenated sources); weights: k-tiles t = 4 s + tt, tt = 2 a + b
  const int nch = p.Ctot / BK;
  const int NS = (p.convT ? 1 : 4) * nch;
  int a_cls = 0, a_ci = 0, a_issued = 0, a_buf = 0;
  auto rebuild_rows = [&]() {
    const bool second = a_ci >= p.C0 && p.src1 != nullptr;
    const int Cs = second ? p.C1 : p.C0;
    const char* base = reinterpret_cast<const char*>(second ? p.src1 : p.src0);
    const int dh = p.convT ? ph - 1 : (a_cls >> 1) - 1, dw = p.convT ? pw - 1 : (a_cls & 1) - 1;
#pragma unroll
    for (int i = 0; i < AG; ++i) {
      const int ih = a_ih[i] + dh, iw = a_iw[i] + dw;
      const bool ok = a_pix[i] >= 0 && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
      const size_t off = (size_t)(unsigned)(a_pix[i] + ih * p.Wi + iw) * (unsigned)Cs * 4u;
      ptrA[i] = (ok ? base + off : zero) + pieceA[i];
    }
  };
  rebuild_rows();
  auto issue_patch = [&]() {
    const bool second = a_ci >= p.C0 && p.src1 != nullptr;
    const unsigned cofs = (unsigned)(second ? a_ci - p.C0 : a_ci) * 4u;
    const unsigned dst = lds0 + (unsigned)a_buf * PA_BYTES + (unsigned)wave * 1024u;
    const char* sa[AG];
#pragma unroll
    for (int i = 0; i < AG; ++i) sa[i] = ptrA[i] + cofs;
    glds16_run<4>(sa, dst, PNW * 1024u);
    glds16_run<2>(sa + 4, dst + 4u * PNW * 1024u, PNW * 1024u);
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
  int b_cls = 0, b_ci = 0, b_tap = 0, b_stage = 0;
  auto issue_weights = [&]() {
    const int a = b_tap >> 1, b = b_tap & 1;
    const int th = p.convT ? (ph ? a : 1 - a) : 2 * a + (b_cls >> 1);
    const int tw = p.convT ? (pw ? b : 1 - b) : 2 * b + (b_cls & 1);
    const unsigned kofs = (unsigned)((th * p.ntw + tw) * p.Ctot + b_ci) * 4u;
    const unsigned dst = lds0 + 2u * PA_BYTES + (unsigned)b_stage * PB_BYTES + (unsigned)wave * 1024u;
    const char* sb[BG];
#pragma unroll
    for (int j = 0; j < BG; ++j) sb[j] = ptrB[j] + kofs;
    glds16_run<BG>(sb, dst, PNW * 1024u);
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
  // pixels: patch row of output pixel (wm 64 + 16 mi + frow) at shift 0; weights: as igemm_dma_kernel's 16x16x32 path
  int prow0[FM];
#pragma unroll
  for (int mi = 0; mi < FM; ++mi) {
    const int ml = wm * 64 + mi * 16 + frow;
    const int seg = ml >> g.seg_sh;
    const int rem = ml & ((1 << g.seg_sh) - 1);
    prow0[mi] = seg * g.seg_rows + (rem >> g.w_sh) * g.W1 + (rem & ((1 << g.w_sh) - 1));
  }
  const int fx = (frow >> 1) & 7;
  const int offH = (half ^ fx) * 16, offL = ((4 + half) ^ fx) * 16;
  const int b_row = 2 * PA_BYTES + (wn * 64 + frow) * 128;
  f32x4 ah[FM], al[FM], bh[FN], bl[FN];
  auto load_a = [&](int buf, int shift, auto lo, auto hi) {
    const char* sa = smem + buf * PA_BYTES;
#pragma unroll
    for (int mi = decltype(lo)::value; mi < decltype(hi)::value; ++mi) {
      const int row = prow0[mi] + shift;
      const int ad = (row << 7) | (((half ^ (row >> 1)) & 7) << 4);
      ah[mi] = *reinterpret_cast<const f32x4*>(sa + ad);
      al[mi] = *reinterpret_cast<const f32x4*>(sa + (ad ^ 64));
    }
  };
  auto load_b = [&](int stage, auto nic) {
    constexpr int ni = decltype(nic)::value;
    const char* sb = smem + stage * PB_BYTES + b_row;
    bh[ni] = *reinterpret_cast<const f32x4*>(sb + ni * 16 * 128 + offH);
    bl[ni] = *reinterpret_cast<const f32x4*>(sb + ni * 16 * 128 + offL);
  };
  auto mfma = [&](const f32x4& a, const f32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
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
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(decltype(cnt)::value) : "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  const int W1 = g.W1;
  auto shift_of = [&](int tt) { return (tt >> 1) * W1 + (tt & 1); };

  // ---- pipeline ----
  issue_patch();
  issue_weights();
  issue_weights();
  issue_weights();
  wait_and_barrier(std::integral_constant<int, 2 * BG>{});   // patch 0 and the weights of tile 0 have landed (also orders ri_out)
  int cs = 0, ab = 0;   // weight stage / patch buffer of the current tile
  load_a(0, 0, I0{}, IH{});
  for_ni([&](auto nic) { load_b(0, nic); });
  // tile tt of a patch: upper pixel fragments | MFMAs of the lower ones | wait + barrier | lower fragments of the next tile |
  // MFMAs of the upper ones, the next tile's weight fragments replacing this tile's one by one, the DMA issues among them
  auto body = [&](auto ttc, auto cnt, auto issue_w, auto issue_p) {
    constexpr int TT = decltype(ttc)::value;
    const int ns = cs + 1 == PNSTB ? 0 : cs + 1;
    load_a(ab, shift_of(TT), IH{}, IF{});
    __builtin_amdgcn_sched_barrier(0);
    for_ni([&](auto nic) { mfma_col(I0{}, IH{}, nic); });
    wait_and_barrier(cnt);
    const int nab = TT == 3 ? ab ^ 1 : ab;
    load_a(nab, shift_of((TT + 1) & 3), I0{}, IH{});
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
  using C2 = std::integral_constant<int, BG>;
  using C8 = std::integral_constant<int, BG + AG>;
  using Y = std::true_type;
  using N = std::false_type;
  for (int s = 0; s + 1 < NS; ++s) {
    body(T0{}, C2{}, Y{}, Y{});   // in flight at the barrier: the weights of t+2
    body(T1{}, C8{}, Y{}, N{});   // the weights of t+2 and the next patch
    body(T2{}, C8{}, Y{}, N{});   // the next patch and the weights of t+2 (the patch is the older: both stay)
    body(T3{}, C2{}, Y{}, N{});   // the weights of t+2; the next patch (older than the awaited weights) has landed
  }
  body(T0{}, C2{}, Y{}, N{});     // last patch: the last weight tile is issued here
  body(T1{}, C2{}, N{}, N{});
  body(T2{}, C0{}, N{}, N{});
  load_a(ab, shift_of(3), IH{}, IF{});
  for_ni([&](auto nic) { mfma_col(I0{}, IF{}, nic); });

  __syncthreads();
  nhwc_tile_store_T<PBM, PBN, PWM, PWN, 16, PSMEM, f32x4>(p, acc, smem, ri_out, n0, tid);
}

// Shapes: what launch_igemm_dma takes at split-K 1, restricted to the two window geometries above with the whole window reached,
// power-of-two pixel grids of width 16 / 32 / 64 and a patch of at most 384 rows.  Returns -2 otherwise (the caller falls through).
int launch_igemm_patch(IGemmP& p, hipStream_t st) {
  if (g_patch < 0 || p.math != 1 || !p.presplit || !p.fast_ok || p.head_w != nullptr || p.N % 128 != 0 || p.Kw != p.K) return -2;
  if (p.out_mode != M2H_OUT_NHWC || p.cls_table != nullptr || p.ldc % 4 != 0 || (reinterpret_cast<size_t>(p.dst) & 15) != 0) return -2;
  if ((size_t)(p.C0 > p.C1 ? p.C0 : p.C1) * 4 > 8192) return -2;
  if (p.wq_sh < 0 || p.hq_sh < 0 || (p.Wq != 16 && p.Wq != 32 && p.Wq != 64)) return -2;
  if (p.convT) {
    if (p.ntap != 4 || p.ntw != 2 || p.thn != 2 || p.twn != 2 || p.Hq != p.Hi || p.Wq != p.Wi) return -2;
  } else {
    if (p.ntap != 16 || p.ntw != 4 || p.thn != 4 || p.twn != 4 || p.stride != 2 || p.mulh != 1 || p.mulw != 1 || p.offh != -1 || p.offw != -1 ||
        p.os != 1 || p.ph != 0 || p.pw != 0 || p.Hi != 2 * p.Hq || p.Wi != 2 * p.Wq || p.src1 != nullptr)
      return -2;
  }
  const int phases = p.convT ? 4 : 1;
  const long t256 = (((long)p.M + 255) / 256) * (p.N / 128) * phases;
  if (g_patch != 2 && t256 < 224) return -2;
  PatchGeo g;
  const int img = p.Hq * p.Wq;
  const int rows = img >= PBM ? PBM / p.Wq : p.Hq;   // output rows per segment
  if (img >= PBM ? (p.Hq % rows != 0) : (PBM % img != 0)) return -2;
  g.nseg = img >= PBM ? 1 : PBM / img;
  g.W1 = p.Wq + 1;
  g.seg_rows = (rows + 1) * g.W1;
  g.w_sh = p.wq_sh;
  g.seg_sh = __builtin_ctz((unsigned)(rows * p.Wq));
  if (g.nseg * g.seg_rows > PA_ROWS) return -2;
  p.MT = (p.M + PBM - 1) / PBM;
  p.NT = p.N / PBN;
  p.S = 1;
  p.pmaj = p.convT ? 1 : 0;
  const long nblk = ((long)p.MT + 7) / 8 * 8 * p.NT * phases;
  if (nblk > 0x7fffffffL) return -2;
  hipLaunchKernelGGL(igemm_patch_kernel, dim3((unsigned)nblk), dim3(64 * PNW), 0, st, p, g);
  return launch_status("igemm_patch<256,128>");
}

}  // namespace m2h
