// First encoder stage of the separator U-Nets with its weights held in REGISTERS (gfx950, bf16x3 math, split32 operands):
// Conv2d(32 -> 64, 4x4, stride 2, pad 1) + BatchNorm(eval) + LeakyReLU (separator_cnn.py:5-12,101-105) over the sliced
// spectrogram -- M = B x 16 x Tm/2 output pixels (524 288 at the benchmark shape), K = 16 taps x 32 channels, N = 64.
//
// On the tiled engines this stage is bound by the operand stream, not the matrix pipe: a 128 x 64 tile fetches 16 KB of image
// rows AND 8 KB of weight rows per k-tile, 1.6 GB per launch, a third of it the same 128 KB weight matrix re-read by 4096 tiles.
// Here a wave OWNS 16 of the 64 output channels for the whole kernel: its B fragments for all 16 taps (16 x (hi + lo) x 16 bytes
// per lane = 128 VGPRs) are loaded once, and a persistent workgroup walks a run of 128-pixel m-tiles streaming only
// the A operand -- im2col rows by LDS-DMA into a four-stage ring.  Measured (B = 256): 187 / 162 us (with / without the class plane) against 191 / 163 us
// on the register engine once its epilogue batched the class-plane loads -- no gain, so the kernel is OFF by default
// (m2h_debug_set 32 = 1); an eight-stage ring and two taps per stage (150 KB of LDS, 256 VGPRs with spills) were slower still.
// The loop runs at ~1 450 cycles per tap for 384 of MFMA: one barrier, one DMA issue and eight fragment reads per 12 MFMAs of a wave (image, waits and row permutation of conv_dma.hip) -- so the
// weights cost neither L2 nor LDS traffic.  Eight waves: wave (row half, channel slice) reads its 64 rows of a k-tile (4 x 2
// ds_read_b128) for 12 v_mfma_f32_16x16x32_bf16; one workgroup per CU (the two row halves hold the same weights).  The k-tiles (= taps, C = 32) run in parity-class order (conv_dma.hip), the
// loop over the 16 taps is unrolled so that every register index is a compile-time constant, and the DMA of the next tile's
// first taps is issued under the last taps of the current one (the ring never drains between tiles).  Epilogue: nhwc_tile_store
// (class plane, BN, LeakyReLU, split32 or fp32 rows) through a scratch of its own, while the ring already holds the next tile.
// Requires: plain conv 4x4 / stride 2 / pad 1, C0 = 32 (one source), N = 64, split32 operands, M % 128 == 0.
#include "igemm_common.h"

namespace m2h {

int g_bres = 0;   // m2h_debug_set 32: 1 = use this kernel (off by default: see the measurements above)

__device__ __attribute__((aligned(128))) float g_zero_page_bres[64];

namespace {

__device__ __forceinline__ void glds16_x2(const char* const* src, unsigned dst, unsigned step) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
               "s_add_u32 m0, m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src[0]), "v"(src[1]), "s"(dst), "s"(step)
               : "memory", "scc");
}

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// tap (th, tw) of position o in the parity-class order: class = o >> 2 = (th & 1, tw & 1), inside a class (th >> 1, tw >> 1)
constexpr int tap_th(int o) { return ((o >> 2) >> 1) + 2 * ((o & 3) >> 1); }
constexpr int tap_tw(int o) { return ((o >> 2) & 1) + 2 * (o & 1); }

}  // namespace

__global__ __launch_bounds__(512, 1) void conv_s2_bres_kernel(const IGemmP p) {
  constexpr int BM = 128, BN = 64, NW = 8, WN = 4, WM = 2, NST = 4, LPT = 2, D = NST - 1;
  constexpr int ST_BYTES = BM * 128;                     // one tap of A rows: 128 rows x 128 B
  constexpr int EPI_BYTES = BM * (BN * 4 + 16);          // epilogue image: the whole tile in one pass
  constexpr int NSTORE = (BM * (BN / 4)) / (64 * NW);    // global stores per thread and tile in nhwc_tile_store
  using AccT = f32x4;

  __shared__ __attribute__((aligned(1024))) char s_ring[NST * ST_BYTES];
  __shared__ __attribute__((aligned(16))) char s_epi[EPI_BYTES];
  __shared__ int ri_out[2][BM], ri_bc[2][BM];
  const unsigned lds_ring = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)s_ring;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & (WN - 1), wm = wave >> 2;        // channel slice / row half of this wave
  const int lrow = lane >> 3;
  const int frow = lane & 15, kq = lane >> 4;

  // ---- this workgroup's run of m-tiles ----
  const int tiles_per = (p.MT + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = blockIdx.x * tiles_per;
  const int t_end = min(p.MT, t_begin + tiles_per);
  if (t_begin >= t_end) return;

  // ---- row bookkeeping of one m-tile for the DMA (this lane's two rows) and the epilogue (LDS, by tile parity) ----
  int a_qh[2], a_rw[2], a_bpix[2];
  auto tile_rows = [&](int mt, int (&qh)[2], int (&rw)[2], int (&bpix)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wave + NW * i) * 8 + lrow;
      int q, rr, b, out, bc;
      decode_row(p, mt * BM + r, 0, 0, q, rr, b, out, bc);
      qh[i] = q * 2 - 1;
      rw[i] = rr * 2 - 1;
      bpix[i] = b * p.Hi * p.Wi;
      if ((lane & 7) == 0) {
        ri_out[mt & 1][r] = out;
        ri_bc[mt & 1][r] = bc;
      }
    }
  };
  const char* zero = reinterpret_cast<const char*>(g_zero_page_bres);
  // LDS piece (lane & 7) of row r = 8 g + lrow holds split32 piece (lane & 7) ^ ((r >> 1) & 7); g = wave + 8 i has the parity of wave
  const int pieceA = ((lane & 7) ^ (((wave & 1) << 2) | (lrow >> 1))) * 16;
  // DMA of tap `o` of the tile whose rows are (qh, rw, bpix) into ring stage `stage`
  auto issue_step = [&](auto oc, int stage, const int (&qh)[2], const int (&rw)[2], const int (&bpix)[2]) {
    constexpr int o = decltype(oc)::value;
    constexpr int th = tap_th(o), tw = tap_tw(o);
    const char* src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ih = qh[i] + th, iw = rw[i] + tw;
      const bool ok = (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
      src[i] = (ok ? reinterpret_cast<const char*>(p.src0) + (size_t)(unsigned)(bpix[i] + ih * p.Wi + iw) * 128u : zero) + pieceA;
    }
    glds16_x2(src, lds_ring + (unsigned)stage * ST_BYTES + (unsigned)wave * 1024u, NW * 1024u);
  };

  using O0 = std::integral_constant<int, 0>;
  tile_rows(t_begin, a_qh, a_rw, a_bpix);
  {
    auto pro = [&](auto self, auto oc) -> void {
      if constexpr (decltype(oc)::value < D) {
        issue_step(oc, decltype(oc)::value, a_qh, a_rw, a_bpix);
        self(self, std::integral_constant<int, decltype(oc)::value + 1>{});
      }
    };
    pro(pro, O0{});
  }

  // ---- this wave's weights: B fragments of its 16 channels for all 16 taps (after the first DMAs: in-order completion makes the
  //      first tile's counted waits trivially true once these have landed) ----
  f32x4 bh[16], bl[16];
  {
    const char* wrow = reinterpret_cast<const char*>(p.w) + (size_t)(wn * 16 + frow) * p.K * 4 + kq * 16;
#pragma unroll
    for (int t = 0; t < 16; ++t) {   // t = th * 4 + tw: the k-tile of that tap in the packed weight row
      bh[t] = *reinterpret_cast<const f32x4*>(wrow + t * 128);
      bl[t] = *reinterpret_cast<const f32x4*>(wrow + t * 128 + 64);
    }
  }

  // fragment addresses: LDS piece j of row r holds split32 piece j ^ ((r >> 1) & 7) (conv_dma.hip)
  const int fx = (frow >> 1) & 7;
  const int a_hi = (wm * 64 + frow) * 128 + ((kq ^ fx) << 4), a_lo = (wm * 64 + frow) * 128 + (((4 + kq) ^ fx) << 4);

  AccT acc[4][1];
  auto mfma = [&](const f32x4& a, const f32x4& b, AccT& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };

  int stage = 0;   // ring stage of the k-tile being computed
  int a2_qh[2], a2_rw[2], a2_bpix[2];
  for (int mt = t_begin; mt < t_end; ++mt) {
    const bool has_next = mt + 1 < t_end;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mi][0][e] = 0.f;
    auto step = [&](auto self, auto sc) -> void {   // one tap
      constexpr int o = decltype(sc)::value;
      if constexpr (o < 16) {
        // wait: this tap's DMA has landed; the DMA of the D-1 taps behind it -- and, on the first D taps behind an epilogue, that
        // epilogue's stores (NSTORE per thread, issued between the next tile's first D taps and tap D) -- may stay in flight
        if constexpr (o < D) wait_vm_barrier<(D - 1) * LPT + NSTORE>();
        else if constexpr (o >= 16 - (D - 1)) {
          if (has_next) wait_vm_barrier<(D - 1) * LPT>();
          else wait_vm_barrier<(15 - o) * LPT>();
        } else wait_vm_barrier<(D - 1) * LPT>();
        // issue the tap D ahead into the stage the previous tap left; past this tile's last tap: the next tile's first taps
        const int istage = stage >= 1 ? stage - 1 : NST - 1;
        if constexpr (o + D < 16) {
          issue_step(std::integral_constant<int, o + D>{}, istage, a_qh, a_rw, a_bpix);
        } else {
          if (has_next) issue_step(std::integral_constant<int, o + D - 16>{}, istage, a2_qh, a2_rw, a2_bpix);
        }
        if constexpr (o == 16 - D - 2) {
          if (has_next) tile_rows(mt + 1, a2_qh, a2_rw, a2_bpix);   // VALU in the shadow of this tap's MFMAs
        }
        constexpr int t = tap_th(o) * 4 + tap_tw(o);
        const char* sa = s_ring + stage * ST_BYTES;
        f32x4 ah[4], al[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          ah[mi] = *reinterpret_cast<const f32x4*>(sa + mi * 16 * 128 + a_hi);
          al[mi] = *reinterpret_cast<const f32x4*>(sa + mi * 16 * 128 + a_lo);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          mfma(al[mi], bh[t], acc[mi][0]);
          mfma(ah[mi], bl[t], acc[mi][0]);
          mfma(ah[mi], bh[t], acc[mi][0]);
        }
        stage = stage + 1 == NST ? 0 : stage + 1;
        self(self, std::integral_constant<int, o + 1>{});
      }
    };
    step(step, O0{});
    // epilogue of this tile (its own scratch: the ring already holds the next tile's first taps)
    nhwc_tile_store<BM, BN, WM, WN, 16, EPI_BYTES, AccT>(p, acc, s_epi, ri_out[mt & 1], ri_bc[mt & 1], 0, tid);
    if (has_next) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a_qh[i] = a2_qh[i];
        a_rw[i] = a2_rw[i];
        a_bpix[i] = a2_bpix[i];
      }
    }
  }
}

int launch_conv_bres(IGemmP& p, hipStream_t st) {
  if (g_bres <= 0 || p.convT || p.math != 1 || !p.presplit || !p.fast_ok || p.head_w != nullptr) return -2;
  if (p.N != 64 || p.C0 != 32 || p.C1 != 0 || p.stride != 2 || p.ntap != 16 || p.ntw != 4 || p.mulh != 1 || p.mulw != 1 || p.offh != -1 ||
      p.offw != -1 || p.thn != 4 || p.twn != 4 || p.os != 1 || p.ph != 0 || p.pw != 0 || p.out_mode != M2H_OUT_NHWC || p.M % 128 != 0)
    return -2;
  if ((p.ldc & 3) != 0 || (reinterpret_cast<size_t>(p.dst) & 15) != 0) return -2;   // the tile store's 16-byte rows
  p.MT = p.M / 128;
  p.NT = 1;
  p.S = 1;
  const int blocks = p.MT < 256 ? p.MT : 256;   // one persistent workgroup per CU
  hipLaunchKernelGGL(conv_s2_bres_kernel, dim3((unsigned)blocks), dim3(512), 0, st, p);
  return launch_status("conv_igemm_f32 (weights-in-registers first encoder stage)");
}

}  // namespace m2h
