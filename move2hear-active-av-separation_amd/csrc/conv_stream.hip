// Weight-streaming implicit-GEMM conv for at most 64 GEMM rows per phase (gfx950, fp32 MFMA) -- the layers of the rollout step whose
// weights dwarf their activations: the two U-Net stages on either side of the 1 x 1 bottleneck and their neighbours at 14 environments
// (separator_cnn.py:46-52,128-135: down4 56 rows x 8.4 MB, down5 14 x 4.2 MB of live taps, up1 14 x 4.2 MB, up2 56 x 16.8 MB) and the
// 14-row Linear layers of the policy (visual_cnn.py:140-141: 4608 -> 512, 9.4 MB).  Such a layer is a stream of N x K weights met by a
// few KB of activations: what it takes is (a) every byte of the weights fetched ONCE, (b) a block on every CU, (c) many 16-byte loads in
// flight per lane.  skinny_gather_kernel (conv_igemm.hip) gives (b) by cutting the rows into 16-row blocks -- the weights then stream
// once per row block (4 x for 56 rows) -- or leaves half the chip without a block (down4: 128 blocks), and its waves walk chains of
// 16 dependent load rounds: 0.5-1 TB/s (profiles/r05_cycle_nodes.txt).
//
// Here a block owns ALL rows (MG <= 4 groups of 16) x 16 output channels of one sub-pixel phase x ONE SLICE of the reduction: the grid is
// (phases x N/16 tiles) x S slices ~ one or two blocks per CU whatever N is, every weight element is read by exactly one lane of one
// block, and a wave's share of the reduction is a handful of steps whose loads are all issued before the first MFMA waits.  Operands go
// from global memory straight into v_mfma_f32_16x16x4_f32 registers (lane (row i, k-quarter kq) loads 16 bytes of weight row i and of
// each of its MG pixel rows, gathered per tap exactly as the engines of conv_igemm.hip do, zero outside the image; the activations are
// <= 1 MB and stay in L2); the waves of a block meet through LDS in wave order.
//
// The S partial tiles of a (phase, channel group) meet through caller-owned scratch in SLICE order (bit-reproducible: the order never
// depends on arrival): plain slab stores -> every wave's vmcnt(0) -> block barrier -> lane 0: agent-scope release fence, vmcnt(0),
// relaxed agent-scope ticket; the block that draws the last ticket: agent-scope acquire fence, then plain loads of the S slabs, BN scale /
// shift, activation, NHWC store (cdna_hip_programming.md Guideline 16, counter form; the form rollout_fused.hip uses).  No reduce launch,
// no spin: a block never waits for another.  tickets: one zeroed word per tile, left zero (the whole-network runner zeroes them in its
// first kernel; the per-layer entry points by a memset node in front of the launch).
#include "igemm_common.h"

namespace m2h {

#ifdef M2H_STREAM_DIAG
// Diagnostic build only (tools/stream_diag.py): 100 MHz real-time stamps of each block's segments.  Never compiled into libm2h.so.
__device__ unsigned long long g_stream_dbg[4096][8];
#define M2H_STAMP(k) do { if (tid == 0 && blockIdx.x < 4096) g_stream_dbg[blockIdx.x][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define M2H_STAMP(k) do { } while (0)
#endif

// NS = most steps (16 floats of the reduction each) a wave may be handed: all of a wave's operand fragments are in flight at once.
template <int MG, int NW, int NS>
__global__ __launch_bounds__(64 * NW) void stream_splitk_kernel(const IGemmP p, unsigned* __restrict__ tickets) {
  __shared__ float R[NW][MG][16][17];
  __shared__ int is_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  M2H_STAMP(0);
  const int i = lane & 15, kq = lane >> 4;
  const int NB = p.N >> 4, S = p.S;
  const int ntiles = NB * (p.convT ? 4 : 1);
  // block -> (tile, slice).  The S slices of a tile get equal blockIdx % 8 (one XCD under the round-robin placement the dispatcher is
  // observed to use: the last arriver then reads its tile's slabs out of its own L2) -- speed only, nothing below depends on it
  int tile, ks;
  {
    const int id = blockIdx.x;
    if ((ntiles & 7) == 0) {
      const int j = id >> 3;
      tile = (id & 7) + 8 * (j / S);
      ks = j % S;
    } else {
      tile = id / S;
      ks = id - tile * S;
    }
  }
  const int nb = tile % NB, phase = tile / NB;
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
  const int spt = p.Ctot >> 4;                                     // 16-float steps per tap
  const int steps = p.thn * p.twn * spt;
  const int b0 = (int)(((long)steps * ks) / S), b1 = (int)(((long)steps * (ks + 1)) / S);
  const int s0 = b0 + ((b1 - b0) * wave) / NW, s1 = b0 + ((b1 - b0) * (wave + 1)) / NW;   // s1 - s0 <= NS (host rule)
  // ---- 1. the weight fragments of every step of this wave: their addresses need nothing but the block's place in the grid, so the
  // stream from memory starts before the rows are decoded
  const float* wrow = wbase + (size_t)(nb * 16 + i) * p.K + 4 * kq;
  const int sa = min(s0, steps - 1);                                // (a wave without a step of its own addresses the last one: never past the row)
  const int tap0 = sa / spt;
  int th = p.th0 + tap0 / p.twn, tw = p.tw0 + tap0 % p.twn, ci = (sa - tap0 * spt) * 16;
  f32x4 wf[NS];
  int s_th[NS], s_tw[NS], s_ci[NS];                                // (wave-uniform: scalar registers)
  // every load below is UNCONDITIONAL (steps past the wave's share re-read its last step, pixels outside the image read pixel 0) and the
  // unwanted values are replaced by zeros afterwards: a load inside a branch makes hipcc wait vmcnt(0) behind it (cdna_hip_programming.md
  // section 5, item 4c), which turned this prologue into a chain of dependent round trips
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    s_th[j] = th; s_tw[j] = tw; s_ci[j] = ci;
    wf[j] = *reinterpret_cast<const f32x4*>(wrow + (size_t)(th * p.ntw + tw) * p.Ctot + ci);
    if (s0 + j + 1 < s1) {                                         // (scalar state only: no load under this branch)
      ci += 16;
      if (ci == p.Ctot) {
        ci = 0;
        if (++tw == p.tw0 + p.twn) {
          tw = p.tw0;
          ++th;
        }
      }
    }
  }
  // ---- 2. the rows (output pixels) and 3. their activation fragments, gathered per tap as the engines of conv_igemm.hip gather them
  int qh[MG], rw[MG], bpix[MG];
#pragma unroll
  for (int g = 0; g < MG; ++g) {
    const int m = g * 16 + i;
    qh[g] = rw[g] = -(1 << 24);                                   // rows past M: outside the image at every tap
    bpix[g] = 0;
    if (m < p.M) {
      int q, rr, b, out, bc;
      decode_row(p, m, ph, pw, q, rr, b, out, bc);
      qh[g] = q * p.stride + offh;
      rw[g] = rr * p.stride + offw;
      bpix[g] = b * p.Hi * p.Wi;
    }
  }
  M2H_STAMP(1);
  f32x4 af[NS][MG];
  bool okA[NS][MG];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const bool second = s_ci[j] >= p.C0;
    const float* src = second ? p.src1 : p.src0;
    const unsigned Cs = second ? p.C1 : p.C0, c = (second ? s_ci[j] - p.C0 : s_ci[j]) + 4 * kq;
#pragma unroll
    for (int g = 0; g < MG; ++g) {
      const int ih = qh[g] + s_th[j] * mulh, iw = rw[g] + s_tw[j] * mulw;
      okA[j][g] = s0 + j < s1 && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
      const unsigned pix = okA[j][g] ? (unsigned)(bpix[g] + ih * p.Wi + iw) : 0u;
      af[j][g] = *reinterpret_cast<const f32x4*>(src + (size_t)pix * Cs + c);
    }
  }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[MG];
#pragma unroll
  for (int g = 0; g < MG; ++g) acc[g] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NS; ++j)
#pragma unroll
    for (int g = 0; g < MG; ++g) {
      const f32x4 av = okA[j][g] ? af[j][g] : zero4;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], wf[j][e], acc[g], 0, 0, 0);
    }
#pragma unroll
  for (int g = 0; g < MG; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) R[wave][g][kq * 4 + e][i] = acc[g][e];    // D[row kq*4 + e][channel i]
  __syncthreads();
  M2H_STAMP(2);
  const int r16 = (tid >> 4) & 15, c16 = tid & 15;
  const int n = nb * 16 + c16;
  f32x4 x = {0.f, 0.f, 0.f, 0.f};                                  // x[g]: the tile's row g*16 + r16, channel c16
  if (tid < 256) {
#pragma unroll
    for (int g = 0; g < MG; ++g) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += R[w][g][r16][c16];          // wave order
      x[g] = v;
    }
  }
  if (S > 1) {
    // slab of this (tile, slice): [256 threads][4 row groups], one 16-byte WRITE-THROUGH store per thread (sc1: the bytes go to the
    // memory side at once, no release fence needed), drained by every storing wave before the barrier in front of the ticket
    float* slabs = p.ws + (size_t)tile * S * 1024;
    if (tid < 256) {
      float* sp = slabs + (size_t)ks * 1024 + tid * 4;
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(sp), "v"(x) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    M2H_STAMP(3);
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(&tickets[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      is_last = old == (unsigned)(S - 1);
      M2H_STAMP(4);
      if (is_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // this CU's L1 may hold lines of an earlier launch's slabs
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    M2H_STAMP(5);
    if (!is_last) return;
    if (tid == 0) tickets[tile] = 0u;                               // zero again for the next launch (ordered by the kernel boundary)
    if (tid < 256) {
      f32x4 part[M2H_STREAM_MAX_SLICES];
#pragma unroll
      for (int sl = 0; sl < M2H_STREAM_MAX_SLICES; ++sl) {          // every slab's load issued before the first add waits
        part[sl] = {0.f, 0.f, 0.f, 0.f};
        if (sl < S) part[sl] = *reinterpret_cast<const f32x4*>(slabs + (size_t)sl * 1024 + tid * 4);
      }
      x = part[0];
#pragma unroll
      for (int sl = 1; sl < M2H_STREAM_MAX_SLICES; ++sl)
        if (sl < S) x += part[sl];                                  // slice order
    }
    asm volatile("" ::"v"(x));
    M2H_STAMP(6);
  }
  if (tid < 256) {
    const float sc = p.scale != nullptr ? p.scale[n] : 1.f;
    const float sh = p.shift != nullptr ? p.shift[n] : 0.f;
#pragma unroll
    for (int g = 0; g < MG; ++g) {
      const int m = g * 16 + r16;
      if (m < p.M) {
        int q, rr, b, out, bc;
        decode_row(p, m, ph, pw, q, rr, b, out, bc);
        const float v = x[g] * sc + sh;
        p.dst[(size_t)out * p.ldc + n] = v > 0.f ? v : v * p.slope;
      }
    }
  }
#ifdef M2H_STREAM_DIAG
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  M2H_STAMP(7);
#endif
}

#ifdef M2H_STREAM_DIAG
extern "C" int m2h_diag_read_stream(unsigned long long* host_out, int nblocks) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stream_dbg), (size_t)nblocks * 8 * sizeof(unsigned long long));
}
#endif

// Geometry of a launch: S slices per tile and NW waves per block such that a wave walks at most NS steps and the grid is about `target`
// blocks (default 256: one per CU).
constexpr int M2H_STREAM_NS = 6;
struct StreamGeom {
  int S, NW;
};
static StreamGeom stream_geom(int tiles, int steps, int MG) {
  const int target = g_stream_blocks > 0 ? g_stream_blocks : 256;
  int S = (target + tiles - 1) / tiles;
  if (S > steps / 4) S = steps / 4;                       // at least four steps per block
  if (S < 1) S = 1;
  const int nwmax = MG <= 2 ? 16 : 8;                     // (LDS: NW x MG partial tiles of 1088 bytes)
  while ((steps + S - 1) / S > nwmax * M2H_STREAM_NS && S < M2H_STREAM_MAX_SLICES) ++S;
  const int per = (steps + S - 1) / S;
  int NW = 4;
  while (NW < nwmax && (per + NW - 1) / NW > 4) NW *= 2;   // four steps per wave where the block may have the waves for it
  StreamGeom g = {S, NW};
  return g;
}

// Host-side rule (shared by the dispatch and by m2h_conv_igemm_workspace_bytes): the shapes this kernel takes.
// fast: channel counts multiples of 32 and 32-bit offsets (IGemmP::fast_ok); Kw = walked reduction length (tap window x channels).
bool stream_splitk_applicable(int math, bool fast, long M, int N, int Ctot, int Kw, int phases, int out_mode, bool plain_operands) {
  if (g_stream < 0 || math != 0 || g_fast_loader < 0 || !fast || !plain_operands) return false;
  if (M > 64 || (N & 15) != 0 || out_mode != M2H_OUT_NHWC || (Ctot & 15) != 0) return false;
  // a weight stream: at least 1 MB of walked weights (below that the 16-row kernels' single pass is as short)
  if ((size_t)N * Kw * phases * sizeof(float) < ((size_t)1 << 20)) return false;
  const int tiles = phases * (N >> 4), MG = (int)((M + 15) / 16);
  if (tiles > M2H_STREAM_MAX_TILES) return false;
  const StreamGeom g = stream_geom(tiles, Kw >> 4, MG);
  return g.S <= M2H_STREAM_MAX_SLICES && ((Kw >> 4) + g.S - 1) / g.S <= g.NW * M2H_STREAM_NS;
}

size_t stream_splitk_workspace_bytes(long M, int N, int Kw, int phases) {
  const int tiles = phases * (N >> 4), MG = (int)((M + 15) / 16);
  const int S = stream_geom(tiles, Kw >> 4, MG).S;
  return S > 1 ? (size_t)tiles * S * 1024 * sizeof(float) + (size_t)M2H_STREAM_MAX_TILES * sizeof(unsigned) : 0;   // [slabs][tickets]
}

// tickets: `tiles` zeroed words (the runner's); nullptr = the words behind the slabs in the workspace, zeroed here by a memset node.
// Returns -2 when the workspace is too small (the caller then takes the 16-row kernels).
int launch_stream_splitk(IGemmP& p, size_t ws_bytes, unsigned* tickets, hipStream_t st) {
  const int phases = p.convT ? 4 : 1;
  const int tiles = phases * (p.N >> 4), MG = (p.M + 15) / 16;
  const int steps = p.thn * p.twn * (p.Ctot >> 4);
  const StreamGeom g = stream_geom(tiles, steps, MG);
  const int S = g.S;
  const size_t slabs = S > 1 ? (size_t)tiles * S * 1024 * sizeof(float) : 0;
  if (S > 1) {
    if (p.ws == nullptr || ws_bytes < slabs + (tickets == nullptr ? (size_t)M2H_STREAM_MAX_TILES * sizeof(unsigned) : 0)) return -2;
    if (tickets == nullptr) {
      tickets = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(p.ws) + slabs);
      const hipError_t e = hipMemsetAsync(tickets, 0, (size_t)M2H_STREAM_MAX_TILES * sizeof(unsigned), st);
      if (e != hipSuccess) return fail((int)e, "conv_igemm (stream split-K): hipMemsetAsync failed: %s", hipGetErrorString(e));
    }
  }
  p.S = S;
  p.MT = MG;
  const dim3 grid((unsigned)(tiles * S));
#define M2H_STREAM_NW(MG_, NW_) M2H_LAUNCH((stream_splitk_kernel<MG_, NW_, M2H_STREAM_NS>), grid, dim3(64 * NW_), 0, st, p, tickets)
#define M2H_STREAM(MG_)                                      \
  do {                                                       \
    if (g.NW == 4) M2H_STREAM_NW(MG_, 4);                    \
    else if (g.NW == 8) M2H_STREAM_NW(MG_, 8);               \
    else M2H_STREAM_NW((MG_ <= 2 ? MG_ : 1), 16);            \
  } while (0)
  if (MG == 1) M2H_STREAM(1);
  else if (MG == 2) M2H_STREAM(2);
  else if (MG == 3) M2H_STREAM(3);
  else M2H_STREAM(4);
#undef M2H_STREAM
#undef M2H_STREAM_NW
  return launch_status("conv_igemm_f32 (stream split-K)");
}

}  // namespace m2h
