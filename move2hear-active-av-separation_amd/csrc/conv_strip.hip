// Strip-walker kernels (gfx950, bf16x3 math, split32 tensors): the two HBM-bound ends of the separator U-Nets at batch --
//   * conv1_strip_kernel      the 16-way frequency slice (+ the bin2mono pre-op) FUSED with the first encoder stage
//                             Conv2d(32(+class plane) -> 64, 4x4, s2, p1) + BN(eval) + LeakyReLU   (separator_cnn.py:73-105)
//   * convT_last_strip_kernel the last decoder stage ConvTranspose2d(64 + 64 skip -> 32 | 16, 4x4, s2, p1) + BN(eval) + ReLU with
//                             the 1x1 head and the de-sliced BHWC store                            (separator_cnn.py:128-135,153-168)
// Both layers have a short reduction (K = 512) against a large image, so they are bound by bytes, not by the matrix pipe: at the
// benchmark shape (B = 256, 512 x 256) the first stage reads 268 MB (536 MB with masks) and writes 134 MB for 34 GFLOP, the last
// reads 268 MB and writes 268 / 134 MB for 69 / 34 GFLOP.  The tiled engines re-stage every input pixel once per tap that touches
// it (4 x) and re-read the weights per tile; the slice was a separate 268 MB round trip through HBM.
//
// Structure shared by both kernels.  A workgroup owns a STRIP of 32 output columns of one image and walks down its rows with a
// rolling window of input rows in LDS, so every input byte is fetched from global memory once per workgroup (+ a 2-column halo):
//   * the next step's input rows are loaded into REGISTERS at the top of a step (plain 16-byte global loads, in flight under the
//     step's MFMAs) and written to the LDS ring at its end -- converted on the way for the first stage (fp32 BHWC -> bf16 hi / lo
//     in the kernel's own channel order: that IS the slice);
//   * the layer's whole weight matrix lives in REGISTERS for the kernel's lifetime: 128 KB (first stage) / 256 | 128 KB (last
//     stage) spread over the workgroup's waves as 16 B-fragments of v_mfma_f32_16x16x32_bf16 (hi + lo = 128 VGPRs per wave); a
//     wave owns 16 output channels (first stage) or one sub-pixel phase x 16 channels (last stage) and multiplies EVERY pixel of
//     the strip row against them.  No weight traffic in the loop at all;
//   * A fragments come from the LDS patch: per pixel a 64-byte hi record and, PART bytes further, a 64-byte lo record (32
//     channels as bf16 each, four 16-byte pieces), consecutive pixels of a plane 64 B apart, read with ds_read_b128 at any pixel
//     shift without bank conflicts: piece j of pixel i sits at slot (j + (i >> 1)) & 3 -- with the lane groups a ds_read_b128 is
//     served in ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) every group then covers the sixteen 16-byte bank slots
//     exactly once (tests/test_kernel_model.py checks the map for every shift); the lo fragment's address is the hi one plus a
//     constant, so a fragment pair costs one address register.  Fragment reads run two (tap, pixel-fragment) groups ahead of the
//     MFMAs that consume them (three register sets, compiler-counted lgkmcnt waits);
//   * the stride-2 first stage keeps even and odd input columns in two planes, so a tap's pixels are consecutive records;
//   * results leave through an LDS image as whole contiguous runs (16 bytes per lane).
// Jobs (image, strip) are dealt to a persistent grid of two workgroups per CU (the second covers the first's barriers).
#include <type_traits>

#include "igemm_common.h"

namespace m2h {

extern thread_local int tl_hi_only;   // conv_igemm.hip: M2H_MATH_BF16

// (tuning knob g_strip: thread-local, m2h_internal.h) m2h_tuning_set 35: -1 = the runner never takes the strip-walker kernels (A/B against the tiled engines)

#ifndef M2H_STRIP_DEPTH
#define M2H_STRIP_DEPTH 2   // fragment groups read ahead of their MFMAs
#endif
#ifndef M2H_STRIP_DBG
#define M2H_STRIP_DBG 0   // diagnostic builds: 1 no MFMAs in the main loops, 2 no fragment reads (one read per step, then registers)
#endif
#ifdef M2H_CLOCK_DIAG
// Diagnostic build only (tools/clock_diag_strip.py): shader-clock cycles every wave spends in each part of its steps, summed
// over the kernel, per workgroup and wave: [0] load issue, [1] MFMA loop, [2] epilogue, [3] first barrier, [4] head / copy-out,
// [5] ring store, [6] second barrier, [7] copy-out behind it; [8] = prologues, [9] = kernel total.
__device__ unsigned long long g_clock_dbg_strip[1024][8][10];
#define SDIAG_DECL unsigned long long sd_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long sd_t = __builtin_amdgcn_s_memtime(); const unsigned long long sd_t0 = sd_t
#define SDIAG(i) do { const unsigned long long sd_n = __builtin_amdgcn_s_memtime(); sd_acc[i] += sd_n - sd_t; sd_t = sd_n; } while (0)
#define SDIAG_END do { sd_acc[9] = __builtin_amdgcn_s_memtime() - sd_t0; if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) for (int i_ = 0; i_ < 10; ++i_) g_clock_dbg_strip[blockIdx.x][threadIdx.x >> 6][i_] = sd_acc[i_]; } while (0)
#else
#define SDIAG_DECL do { } while (0)
#define SDIAG(i) do { } while (0)
#define SDIAG_END do { } while (0)
#endif

namespace {

constexpr int SW = 32;            // output columns of a strip
constexpr int PX = SW + 2;        // staged pixel records per plane (one halo record each side)
constexpr int PART = PX * 64;     // bytes of the hi (or lo) records of a plane
constexpr int PLANE = 2 * PART;   // a plane: hi records, then lo records

// byte offset of 16-byte piece j (8 channels) of pixel record i inside the hi part of a plane (lo: + PART)
__device__ __forceinline__ int px_addr(int i, int j) { return i * 64 + (((j + (i >> 1)) & 3) << 4); }

struct Frag {
  f32x4 h, l;
};

// Job (image, strip) of a persistent workgroup's iteration: the strips of one image go to workgroups with equal blockIdx % 8 --
// one XCD under the round-robin placement (speed only, never correctness: MI355X_MICROARCH.md) -- at consecutive slots, so the
// halo columns two neighbouring strips both read are served by one L2 instead of being fetched from HBM twice (PMC, first version:
// FETCH_SIZE 1.38 x the input bytes on the first stage with image-major order over all XCDs).  Needs gridDim.x % 8 == 0.
__device__ __forceinline__ int strip_job(int linear, int strips, int jobs) {
  const int xcd = linear & 7, y = linear >> 3;
  const int img = (y / strips) * 8 + xcd;
  const int job = img * strips + (y % strips);
  return job < jobs ? job : -1;
}

__device__ __forceinline__ f32x4 mfma16(const f32x4& a, const f32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// fp32 -> (bf16 hi, bf16 lo) with x = hi + lo to 2^-17 relative (the split32 convention of the engines)
__device__ __forceinline__ void split4(const f32x4& v, bf16x4& hi, bf16x4& lo) {
  hi = __builtin_convertvector(v, bf16x4);
  lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// First encoder stage fused with the slice.
//   input   mix / masks [B][512][T][2] fp32 (BHWC as the reference hands it over); network pixel (b, h, w) has the 32 channels
//           (c, s) = mix[b][s * 32 + h][w][c]  (separator_cnn.py:85-90)
//   output  split32 NHWC [B][16][T/2][64]
// K order inside a tap: k' = kg * 8 + sb * 4 + t * 2 + c  <->  s = 4 kg + 2 t + sb, channel c * 16 + s of the reference: what a
// loader lane holds after two 16-byte loads of two frequency rows is then one aligned 8-byte piece of a pixel record.  The weights
// are packed to match (m2h_pack_strip_conv1).
// ---------------------------------------------------------------------------------------------------------------------------
struct StripConv1P {
  const float* mix;
  const float* masks;
  const f32x4* wreg;        // [wave 4][tap 16][hi, lo][lane 64] 16-byte B fragments
  const float* scale;
  const float* shift;
  const float* cls_table;   // [9][64] or null
  const float* cls_val;     // [B] (cls_kind 0: target_class + 1 as floats; 1: raw target_class, float32; 2: raw, int64)
  int cls_kind;
  float* dst;
  int B, T, Wq, strips, jobs, jobs_padded;   // jobs_padded: linear job slots incl. the images a partial group of 8 leaves empty
  int rev;                                   // walk the job slots from the last image to the first (what the producer wrote last is read first)
  float slope;
};

template <bool MASKED, bool HI = false>   // HI: the bf16 hi halves only (M2H_MATH_BF16)
__global__ __launch_bounds__(256, 2) void conv1_strip_kernel(const StripConv1P p) {
  constexpr int ROW = 2 * PLANE;   // a row slot: even-column plane, odd-column plane
  __shared__ __attribute__((aligned(1024))) char s_ring[4 * ROW];
  constexpr int OPX = 256 + 16;    // out image: a pixel's 256-byte split32 record + 16 B (the 16 lanes of a store step two bank slots each, not one)
  __shared__ __attribute__((aligned(16))) char s_out[SW * OPX];
  __shared__ __attribute__((aligned(16))) float s_cls[9 * 64];   // the class plane's border table (read per step: no global load in the loop)
  __shared__ __attribute__((aligned(16))) float s_bn[2 * 64];    // folded BatchNorm scale | shift (read in the epilogue: not held across the MFMA loop)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kg = lane >> 4;
  if (p.cls_table != nullptr)
    for (int i = tid; i < 9 * 64; i += 256) s_cls[i] = p.cls_table[i];
  if (tid < 128) s_bn[tid] = tid < 64 ? p.scale[tid] : p.shift[tid - 64];

  // ---- the layer's weights: this wave's 16 output channels, all 16 taps ----
  f32x4 Bh[16], Bl[16];
  {
    const f32x4* w = p.wreg + (size_t)wave * (16 * 2 * 64) + lane;
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
      Bh[tap] = w[(tap * 2 + 0) * 64];
      Bl[tap] = w[(tap * 2 + 1) * 64];
    }
  }
  // ---- A fragment addresses inside a row slot: input column 2 r + tw - 1 = patch column 2 (r - r0) + tw + 1: plane (tw + 1) & 1,
  // record (r - r0) + ((tw + 1) >> 1) ----
  int a_ad[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int tw = 0; tw < 4; ++tw) a_ad[mt][tw] = ((tw + 1) & 1) * PLANE + px_addr(16 * mt + m + ((tw + 1) >> 1), kg);
  // ---- epilogue constants.  The weights are the MFMA's A operand (rows = channels) and the pixels its B operand (columns), so a
  // lane's four accumulator values are channels n0 .. n0 + 3 of ONE pixel (16 mt + m): an aligned 8-byte run of its split32 record
  const int n0 = wave * 16 + 4 * kg;
  const int o_byte = m * OPX + (n0 >> 5) * 128 + (n0 & 31) * 2;   // hi run of pixel m's record in the out image (lo: + 64; mt: + 16 * OPX)
  // ---- loader lanes: wave = the piece (8 channels) it writes; lane = (sb, pixel pair ii); halo lanes 0..15 = (h, t, sb, side) ----
  const int sb = lane >> 5, ii = lane & 31;
  const int hh = (lane >> 3) & 1, ht = (lane >> 2) & 1, hsb = (lane >> 1) & 1, hside = lane & 1;
  const int w_ad = px_addr(ii + 1, wave) + sb * 8;
  const int hw_ad = px_addr(hside ? PX - 1 : 0, wave) + hsb * 8 + ht * 4;

  struct Pre {               // a prefetched pair of rows: [h][t] main float4s and the halo float4 (+ the masks beside them)
    f32x4 v[2][2], vh;
    f32x4 k[MASKED ? 2 : 1][MASKED ? 2 : 1], kh;
  };

  SDIAG_DECL;
  for (int lin = blockIdx.x; lin < p.jobs_padded; lin += gridDim.x) {
    const int job = strip_job(p.rev ? p.jobs_padded - 8 - (lin & ~7) + (lin & 7) : lin, p.strips, p.jobs);
    if (job < 0) continue;   // uniform
    const int b = job / p.strips;
    const int r0 = (job - b * p.strips) * SW;
    const float* mixb = p.mix + (size_t)b * 512 * p.T * 2;     // this image (uniform); lane offsets below are 32-bit
    const float* maskb = MASKED ? p.masks + (size_t)b * 512 * p.T * 2 : nullptr;
    const bool has_l = r0 > 0, has_r = r0 + SW < p.Wq;
    const float cv = p.cls_table == nullptr ? 0.f
                     : p.cls_kind == 0 ? p.cls_val[b]
                     : p.cls_kind == 1 ? p.cls_val[b] + 1.f
                                       : (float)reinterpret_cast<const long long*>(p.cls_val)[b] + 1.f;   // separator_cnn.py:96
    int l_off[2];    // [t]: frequency row s * 32 (s = 4 wave + 2 t + sb), column pair ii of the strip
#pragma unroll
    for (int t = 0; t < 2; ++t) l_off[t] = ((4 * wave + 2 * t + sb) * 32 * p.T + 2 * r0 + 2 * ii) * 2;
    const int lh_off = ((4 * wave + 2 * ht + hsb) * 32 * p.T + 2 * r0 + (hside ? 2 * SW : -2)) * 2;
    const bool lh_ok = lane < 16 && (hside ? has_r : has_l);
    // border class of this lane's pixel (16 mt + m) along the row: 0 first column of the image, 2 last, 1 inside
    const int cw0 = (m == 0 && !has_l) ? 0 : 1, cw1 = (m == 15 && !has_r) ? 2 : 1;
    SDIAG(9);

    auto load_pair = [&](int k, Pre& pr) {   // rows 2k-1, 2k of the sliced image -> registers
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int hi_ = 2 * k - 1 + h;
        const bool okr = (unsigned)hi_ < 32u;      // uniform
        const int rowterm = hi_ * p.T * 2;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          pr.v[h][t] = okr ? *reinterpret_cast<const f32x4*>(mixb + (l_off[t] + rowterm)) : f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (MASKED) pr.k[h][t] = okr ? *reinterpret_cast<const f32x4*>(maskb + (l_off[t] + rowterm)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      pr.vh = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (MASKED) pr.kh = pr.vh;
      {
        const int hi_ = 2 * k - 1 + hh;
        if (lh_ok && (unsigned)hi_ < 32u) {
          pr.vh = *reinterpret_cast<const f32x4*>(mixb + (lh_off + hi_ * p.T * 2));
          if constexpr (MASKED) pr.kh = *reinterpret_cast<const f32x4*>(maskb + (lh_off + hi_ * p.T * 2));
        }
      }
    };
    auto store_pair = [&](int k, Pre& pr) {   // registers -> row slots (2k) & 3, (2k + 1) & 3, converted to bf16 hi / lo records
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        char* row = s_ring + ((2 * k + h) & 3) * ROW;
        f32x4 v0 = pr.v[h][0], v1 = pr.v[h][1];
        if constexpr (MASKED) {   // (a padding row holds zeros in both: the pre-op of (0, 0) is 0)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v0[e] = masked_log_mag(v0[e], pr.k[h][0][e]);
            v1[e] = masked_log_mag(v1[e], pr.k[h][1][e]);
          }
        }
#pragma unroll
        for (int px = 0; px < 2; ++px) {   // the pair's even column -> plane 0, odd column -> plane 1
          const f32x4 v = {v0[2 * px], v0[2 * px + 1], v1[2 * px], v1[2 * px + 1]};
          bf16x4 hi, lo;
          split4(v, hi, lo);
          *reinterpret_cast<bf16x4*>(row + px * PLANE + w_ad) = hi;
          *reinterpret_cast<bf16x4*>(row + px * PLANE + PART + w_ad) = lo;
        }
      }
      if (lane < 16) {
        char* row = s_ring + ((2 * k + hh) & 3) * ROW;
        f32x4 v = pr.vh;
        if constexpr (MASKED) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = masked_log_mag(v[e], pr.kh[e]);
        }
        bf16x4 hi, lo;
        split4(v, hi, lo);
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
          *reinterpret_cast<bf16x2_t*>(row + px * PLANE + hw_ad) = bf16x2_t{hi[2 * px], hi[2 * px + 1]};
          *reinterpret_cast<bf16x2_t*>(row + px * PLANE + PART + hw_ad) = bf16x2_t{lo[2 * px], lo[2 * px + 1]};
        }
      }
    };

    Pre pre;
    {   // ---- prologue: rows -1 .. 2, both pairs' loads in flight together ----
      Pre pre1;
      load_pair(0, pre);
      load_pair(1, pre1);
      store_pair(0, pre);
      store_pair(1, pre1);
    }
    __syncthreads();
    SDIAG(8);

    // one output row q (QP = q & 1 fixes the ring slots at compile time): rows 2q-1 .. 2q+2 = slots (2 QP + th) & 3
    auto step = [&](int q, auto qpc) {
      constexpr int QP = decltype(qpc)::value;
      if (q + 2 <= 16) load_pair(q + 2, pre);   // consumed behind this step's MFMAs
      SDIAG(0);
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      {
        // 32 groups g = (th, tw, mt) of one fragment pair and three MFMAs; the reads run DEPTH groups (4 x 48 MFMA cycles) ahead:
        // with two, every group waited ~40 cycles for its fragments (2 260 cycles per step against 1 536 of MFMA issue)
        constexpr int G = 32, DEPTH = M2H_STRIP_DEPTH;
        Frag f[DEPTH + 1];
#pragma unroll
        for (int g = 0; g < G + DEPTH; ++g) {
          if (g < G) {
            const int th = g >> 3, tw = (g >> 1) & 3, mt = g & 1;
            const char* src = s_ring + ((2 * QP + th) & 3) * ROW + a_ad[mt][tw];
            if (M2H_STRIP_DBG != 2 || g < DEPTH + 1) {
              f[g % (DEPTH + 1)].h = *reinterpret_cast<const f32x4*>(src);
              f[g % (DEPTH + 1)].l = *reinterpret_cast<const f32x4*>(src + PART);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (g >= DEPTH) {
            const int c = g - DEPTH;
            const int tap = c >> 1, mt = c & 1;
            const Frag& a = f[c % (DEPTH + 1)];
            if constexpr (M2H_STRIP_DBG == 1) {
              acc[mt] += a.l + a.h + Bh[tap] + Bl[tap];
            } else {
              if constexpr (!HI) {
                acc[mt] = mfma16(Bh[tap], a.l, acc[mt]);   // rows = this wave's channels, columns = the fragment's pixels
                acc[mt] = mfma16(Bl[tap], a.h, acc[mt]);
              }
              acc[mt] = mfma16(Bh[tap], a.h, acc[mt]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      SDIAG(1);
      // ---- epilogue: class plane, BN, LeakyReLU, split32 record into the out image ----
      const f32x4 sc4 = *reinterpret_cast<const f32x4*>(s_bn + n0), sh4 = *reinterpret_cast<const f32x4*>(s_bn + 64 + n0);
      f32x4 shc[2] = {sh4, sh4};   // shift with the class plane folded in: (acc + cv t) sc + sh = acc sc + (cv t sc + sh)
      if (p.cls_table != nullptr) {
        const int ch = q == 0 ? 0 : (q == 15 ? 2 : 1);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(s_cls + (ch * 3 + cw0) * 64 + n0);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(s_cls + (ch * 3 + cw1) * 64 + n0);
        shc[0] = (cv * t0) * sc4 + sh4;
        shc[1] = (cv * t1) * sc4 + sh4;
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 v = acc[mt] * sc4 + shc[mt];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], v[j] * p.slope);   // LeakyReLU, 0 <= slope <= 1
        bf16x4 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<bf16x4*>(s_out + mt * (16 * OPX) + o_byte) = hi;
        *reinterpret_cast<bf16x4*>(s_out + mt * (16 * OPX) + o_byte + 64) = lo;
      }
      SDIAG(2);
      __syncthreads();   // every wave is done with rows 2q-1, 2q; the out image is complete
      SDIAG(3);
      {
        float* drow = p.dst + ((size_t)(b * 16 + q) * p.Wq + r0) * 64;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int idx = tid + 256 * u;
          *reinterpret_cast<f32x4*>(drow + idx * 4) = *reinterpret_cast<const f32x4*>(s_out + (idx >> 4) * OPX + (idx & 15) * 16);
        }
      }
      SDIAG(4);
      if (q + 2 <= 16) store_pair(q + 2, pre);   // into the slots rows 2q-1, 2q leave
      SDIAG(5);
      __syncthreads();
      SDIAG(6);
    };
    for (int q = 0; q < 16; q += 2) {
      step(q, std::integral_constant<int, 0>{});
      step(q + 1, std::integral_constant<int, 1>{});
    }
  }
  SDIAG_END;
}

// weights [Co = 64][Ci >= 32][4][4] fp32 (torch Conv2d layout; input channel c * 16 + s) -> the kernel's register image
__global__ void pack_strip_conv1_kernel(const float* __restrict__ w, int Ci, f32x4* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // (wave, tap, part, lane)
  if (idx >= 4 * 16 * 2 * 64) return;
  const int lane = idx & 63, part = (idx >> 6) & 1, tap = (idx >> 7) & 15, wave = idx >> 11;
  const int n = wave * 16 + (lane & 15), kg = lane >> 4;
  const int kh = tap >> 2, kw = tap & 3;
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int sb = e >> 2, t = (e >> 1) & 1, c = e & 1;
    const int s = 4 * kg + 2 * t + sb;
    const float x = w[(((size_t)n * Ci + (c * 16 + s)) * 4 + kh) * 4 + kw];
    const __bf16 hi = (__bf16)x;
    o[e] = part == 0 ? hi : (__bf16)(x - (float)hi);
  }
  out[idx] = __builtin_bit_cast(f32x4, o);
}

int launch_strip_conv1(const float* mix, const float* masks, const void* wreg, const float* scale, const float* shift, const float* cls_table,
                       const float* cls_val, float* dst, int B, int T, float slope, hipStream_t st, int cls_kind) {
  StripConv1P p;
  p.cls_kind = cls_kind;
  p.mix = mix; p.masks = masks; p.wreg = static_cast<const f32x4*>(wreg); p.scale = scale; p.shift = shift;
  p.cls_table = cls_table; p.cls_val = cls_val; p.dst = dst;
  p.B = B; p.T = T; p.Wq = T / 2; p.strips = p.Wq / SW; p.jobs = B * p.strips; p.slope = slope;
  p.jobs_padded = (B + 7) / 8 * 8 * p.strips;
  p.rev = masks != nullptr ? (g_strip_rev & 1) : ((g_strip_rev >> 2) & 1);
  const int grid = p.jobs_padded < 512 ? p.jobs_padded : 512;   // a multiple of 8 (strip_job)
  if (tl_hi_only && masks != nullptr) M2H_LAUNCH((conv1_strip_kernel<true, true>), dim3(grid), dim3(256), 0, st, p);
  else if (tl_hi_only) M2H_LAUNCH((conv1_strip_kernel<false, true>), dim3(grid), dim3(256), 0, st, p);
  else if (masks != nullptr) M2H_LAUNCH((conv1_strip_kernel<true>), dim3(grid), dim3(256), 0, st, p);
  else M2H_LAUNCH((conv1_strip_kernel<false>), dim3(grid), dim3(256), 0, st, p);
  return launch_status(masks != nullptr ? "strip_conv1<masked>" : "strip_conv1");
}


// ---------------------------------------------------------------------------------------------------------------------------
// Last decoder stage + head.
//   inputs  src0 (the previous decoder stage) and src1 (the first encoder stage's skip), split32 NHWC [B][Hq][Wq][64] each: the
//           concatenation (separator_cnn.py:160-161) is the order of the four 32-channel chunks
//   weights the engines' split32 transposed-conv pack [phase 4][N][tap 4][128] (m2h_pack_convT_weight + m2h_split32), read once
//           into registers: wave (phase, n-tile) holds its 4 taps x 4 chunks
//   output  BHWC [B][16 * 2 Hq][2 Wq][N / 16] fp32 after BN(eval) + ReLU, the 1x1 head (:134) and the de-slice (:163-168)
// Sub-pixel phase (ph, pw) of output block (q, r) is pixel (2q + ph, 2r + pw); its tap (th, tw) reads input pixel
// (q + th (2 ph - 1), r + tw (2 pw - 1)) (conv_igemm.hip).  A wave therefore reads two row slots and two column shifts only.
// The head runs on the matrix pipe too: the activated tile goes to LDS as bf16 hi / lo records (all N channels of a pixel side by
// side, whichever wave made them), wave (phase, 16-position half) multiplies its records with the head matrix (bf16x3 like the
// rest) and writes the de-sliced image Z[s][row][column][c]; the workgroup copies Z out as 512- / 256-byte runs.
// ---------------------------------------------------------------------------------------------------------------------------
struct StripLastP {
  const float* src0;
  const float* src1;
  const float* w;
  const float* scale;
  const float* shift;
  const float* head_w;   // [N][N] fp32
  const float* head_b;   // [N]
  float* dst;
  int B, Hq, Wq, strips, jobs, jobs_padded, rev;
  float slope;
};

template <int N, bool HI = false>   // HI: the bf16 hi halves only (M2H_MATH_BF16), head included
__global__ __launch_bounds__(N * 16, 2) void convT_last_strip_kernel(const StripLastP p) {
  constexpr int NT = N / 16, NW = 4 * NT, NTH = 64 * NW;
  constexpr int Cc = N / 16;                 // output channels after the de-slice
  constexpr int ROWB = 4 * PLANE;            // a row slot: four 32-channel chunk planes
  constexpr int NF4 = 2 * PX * 16;           // 16-byte pieces of one input row of the strip (both sources)
  constexpr int LPT = (NF4 + NTH - 1) / NTH; // loads per thread and row
  constexpr int ZROW = 64 * Cc;              // floats of one de-sliced output run
  // (A three-stage pipeline with ONE barrier per step for the one-workgroup-per-CU N = 32 kernel -- compute of row i, head of row
  // i-1, copy-out of row i-2 per iteration, Y and Z double-buffered, four ring slots, the two waves of a SIMD going through the
  // stages in opposite order -- was built and A/B-measured on one box: 247 vs 248 us.  The step is bound by the matrix pipe that
  // the two waves of a SIMD share at the clock the chip holds, not by the barriers; removed.)
  constexpr int RING = 3, NBUF = 1;
  constexpr int YPART = SW * 64;             // Y[buffer][phase]: 32 hi records, then 32 lo records
  constexpr int YBUF = 4 * 2 * YPART, ZBUF = 32 * ZROW;
  __shared__ __attribute__((aligned(1024))) char s_ring[RING * ROWB];
  __shared__ __attribute__((aligned(1024))) char s_y[NBUF * YBUF];
  __shared__ __attribute__((aligned(16))) float s_z[NBUF * ZBUF];
  __shared__ __attribute__((aligned(16))) f32x4 s_wh[NT * 2 * 64];
  __shared__ __attribute__((aligned(16))) float s_cst[3 * N];   // folded BatchNorm scale | shift | head bias (read where used, not held across the MFMA loop)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int phase = wave / NT, nt = wave % NT;     // main loop role
  if (tid < 3 * N) s_cst[tid] = tid < N ? p.scale[tid] : (tid < 2 * N ? p.shift[tid - N] : p.head_b[tid - 2 * N]);
  const int ph = phase >> 1, pw = phase & 1;
  const int sy = 2 * ph - 1, sx = 2 * pw - 1;
  const int m = lane & 15, kg = lane >> 4;

  // ---- weights of (phase, n-tile): 4 taps x 4 chunks, hi and lo ----
  f32x4 Bh[16], Bl[16];
  {
    const char* wrow = reinterpret_cast<const char*>(p.w) + ((size_t)(phase * N + nt * 16 + m) * 512) * 4 + kg * 16;
#pragma unroll
    for (int t = 0; t < 16; ++t) {   // t = tap * 4 + chunk: 128 bytes apart
      Bh[t] = *reinterpret_cast<const f32x4*>(wrow + t * 128);
      Bl[t] = *reinterpret_cast<const f32x4*>(wrow + t * 128 + 64);
    }
  }
  // ---- head matrix as B fragments in LDS (lane-linear), Y cleared once (N = 16 never writes its upper channel pieces) ----
  for (int i = tid; i < NT * 64; i += NTH) {
    const int hn = i >> 6, l = i & 63;
    const int n2 = hn * 16 + (l & 15), k0 = (l >> 4) * 8;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (k0 + e) < N ? p.head_w[n2 * N + k0 + e] : 0.f;
      hi[e] = (__bf16)x;
      lo[e] = (__bf16)(x - (float)hi[e]);
    }
    s_wh[(hn * 2 + 0) * 64 + l] = __builtin_bit_cast(f32x4, hi);
    s_wh[(hn * 2 + 1) * 64 + l] = __builtin_bit_cast(f32x4, lo);
  }
  for (int i = tid; i < NBUF * YBUF / 16; i += NTH) reinterpret_cast<f32x4*>(s_y)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- A fragment offsets inside a row slot (column shifts 0 and sx), without the chunk plane ----
  int a_off[2][2];   // [mt][tw]
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int tw = 0; tw < 2; ++tw) a_off[mt][tw] = px_addr(16 * mt + m + 1 + tw * sx, kg);
  // ---- epilogue constants.  The weights are the MFMA's A operand (rows = channels), the positions its B operand (columns): a
  // lane's four accumulator values are channels n0 .. n0 + 3 of ONE position (16 mt + m): an aligned 8-byte run of its Y record
  const int n0 = nt * 16 + 4 * kg;
  int y_ad[2];                     // hi run of this lane's channels in the record of position 16 mt + m (lo: + YPART)
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) y_ad[mt] = phase * (2 * YPART) + px_addr(16 * mt + m, n0 >> 3) + (n0 & 7) * 2;
  // head role: wave -> (phase, 16-position half) for N = 32, (phase, both halves) for N = 16
  const int h_phase = N == 32 ? (wave >> 1) : wave;
  const int h_ph = h_phase >> 1, h_pw = h_phase & 1;
  // ---- loader: piece u of this thread of an input row (decoded where used: a handful of integer instructions per row) ----
  auto piece_of = [&](int u, int& src, int& px, int& pc) {
    const int f = tid + NTH * u;
    src = f >= PX * 16 ? 1 : 0;
    const int rem = f - src * (PX * 16);
    px = f < NF4 ? (rem >> 4) : -1;
    pc = rem & 15;
  };

  SDIAG_DECL;
  for (int lin = blockIdx.x; lin < p.jobs_padded; lin += gridDim.x) {
    const int job = strip_job(p.rev ? p.jobs_padded - 8 - (lin & ~7) + (lin & 7) : lin, p.strips, p.jobs);
    if (job < 0) continue;   // uniform
    const int b = job / p.strips;
    const int r0 = (job - b * p.strips) * SW;
    const bool has_l = r0 > 0, has_r = r0 + SW < p.Wq;
    SDIAG(9);

    f32x4 pre[LPT];
    auto load_row = [&](int h) {
      const bool okr = (unsigned)h < (unsigned)p.Hq;
      const long rowoff = ((long)(b * p.Hq + h) * p.Wq + (r0 - 1)) * 64;
#pragma unroll
      for (int u = 0; u < LPT; ++u) {
        int src, px, pc;
        piece_of(u, src, px, pc);
        const bool ok = okr && px >= 0 && (px > 0 || has_l) && (px < PX - 1 || has_r);
        const float* base = src ? p.src1 : p.src0;
        pre[u] = ok ? *reinterpret_cast<const f32x4*>(base + rowoff + (px * 64 + pc * 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    auto store_row = [&](int h) {
      char* row = s_ring + ((h + 2 * RING) % RING) * ROWB;
#pragma unroll
      for (int u = 0; u < LPT; ++u) {
        int src, px, pc;
        piece_of(u, src, px, pc);
        const int chunk = src * 2 + (pc >> 3), piece = pc & 7;
        if (px >= 0) *reinterpret_cast<f32x4*>(row + chunk * PLANE + (piece < 4 ? 0 : PART) + px_addr(px, piece & 3)) = pre[u];
      }
    };

    // ---- stage C: output rows 2q, 2q+1 of this wave's (phase, 16 channels): 96 MFMAs, then BN + ReLU and the activated tile as
    // bf16 hi / lo records in Y[q & (NBUF-1)][phase][position] ----
    auto stage_c = [&](int q) {
      const int base0 = ((q + 2 * RING) % RING) * ROWB, basey = ((q + sy + 2 * RING) % RING) * ROWB;
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      {
        // 32 groups g = (tap, chunk, mt) of one fragment pair and three MFMAs; the reads run DEPTH groups (4 x 48 MFMA cycles) ahead
        constexpr int G = 32, DEPTH = M2H_STRIP_DEPTH;
        int ad[2][2][2];   // [th][tw][mt]: row slot + column shift
#pragma unroll
        for (int th = 0; th < 2; ++th)
#pragma unroll
          for (int tw = 0; tw < 2; ++tw)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) ad[th][tw][mt] = (th ? basey : base0) + a_off[mt][tw];
        Frag f[DEPTH + 1];
#pragma unroll
        for (int g = 0; g < G + DEPTH; ++g) {
          if (g < G) {
            const int tap = g >> 3, kc = (g >> 1) & 3, mt = g & 1;
            const char* src = s_ring + kc * PLANE + ad[tap >> 1][tap & 1][mt];
            f[g % (DEPTH + 1)].h = *reinterpret_cast<const f32x4*>(src);
            f[g % (DEPTH + 1)].l = *reinterpret_cast<const f32x4*>(src + PART);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (g >= DEPTH) {
            const int c = g - DEPTH;
            const int t = c >> 1, mt = c & 1;
            const Frag& a = f[c % (DEPTH + 1)];
            if constexpr (!HI) {
              acc[mt] = mfma16(Bh[t], a.l, acc[mt]);   // rows = this wave's channels, columns = the fragment's positions
              acc[mt] = mfma16(Bl[t], a.h, acc[mt]);
            }
            acc[mt] = mfma16(Bh[t], a.h, acc[mt]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      SDIAG(1);
      char* yb = s_y + (q & (NBUF - 1)) * YBUF;
      const f32x4 sc4 = *reinterpret_cast<const f32x4*>(s_cst + n0), sh4 = *reinterpret_cast<const f32x4*>(s_cst + N + n0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 v = acc[mt] * sc4 + sh4;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], v[j] * p.slope);   // ReLU (slope 0)
        bf16x4 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<bf16x4*>(yb + y_ad[mt]) = hi;
        *reinterpret_cast<bf16x4*>(yb + y_ad[mt] + YPART) = lo;
      }
    };
    // ---- stage H: z = Wh y + hb for this wave's (phase, positions) of row q, de-sliced into Z[q & (NBUF-1)] ----
    auto stage_h = [&](int q) {
      const char* yp = s_y + (q & (NBUF - 1)) * YBUF + h_phase * (2 * YPART);
      float* zb = s_z + (q & (NBUF - 1)) * ZBUF;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if (N == 32 && mt != (wave & 1)) continue;   // wave-uniform
        const int pos = 16 * mt + m;
        const f32x4 ah = *reinterpret_cast<const f32x4*>(yp + px_addr(pos, kg));
        const f32x4 al = *reinterpret_cast<const f32x4*>(yp + YPART + px_addr(pos, kg));
        f32x4 z[NT];
#pragma unroll
        for (int hn = 0; hn < NT; ++hn) {
          const f32x4 bh = s_wh[(hn * 2 + 0) * 64 + lane], bl = s_wh[(hn * 2 + 1) * 64 + lane];
          z[hn] = *reinterpret_cast<const f32x4*>(s_cst + 2 * N + hn * 16 + 4 * kg);   // head bias of rows c * 16 + s, s = 4 kg + j
          if constexpr (!HI) {
            z[hn] = mfma16(bh, al, z[hn]);   // rows = head outputs c * 16 + s (c = hn), columns = positions
            z[hn] = mfma16(bl, ah, z[hn]);
          }
          z[hn] = mfma16(bh, ah, z[hn]);
        }
        // lane: position 16 mt + m -> output column 2 position + pw; rows s = 4 kg + j of run (s, ph)
        const int ow = 2 * (16 * mt + m) + h_pw;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float* zr = zb + ((4 * kg + j) * 2 + h_ph) * ZROW + ow * Cc;
          if constexpr (N == 32) *reinterpret_cast<float2*>(zr) = float2{z[0][j], z[1][j]};
          else *zr = z[0][j];
        }
      }
    };
    // ---- stage O: Z of row q -> 32 runs (s, row parity) of 64 output columns: ZROW floats = 512 / 256 contiguous bytes each ----
    auto stage_o = [&](int q) {
      const float* zb = s_z + (q & (NBUF - 1)) * ZBUF;
      constexpr int F2 = 32 * ZROW / 2;   // float2 pieces
#pragma unroll
      for (int u = 0; u < F2 / NTH; ++u) {
        const int idx = tid + NTH * u;
        const int rs = idx / (ZROW / 2), c2 = idx - rs * (ZROW / 2);
        const int s = rs >> 1, oh = 2 * q + (rs & 1);
        const float2 v = *reinterpret_cast<const float2*>(zb + rs * ZROW + 2 * c2);
        float* d = p.dst + (((size_t)b * (32 * p.Hq) + (size_t)s * (2 * p.Hq) + oh) * (2 * p.Wq) + 2 * r0) * Cc + 2 * c2;
        *reinterpret_cast<float2*>(d) = v;
      }
    };

    // ---- prologue: rows -1, 0, 1 ----
    load_row(-1);
    store_row(-1);
    load_row(0);
    store_row(0);
    load_row(1);
    store_row(1);
    __syncthreads();
    SDIAG(8);

    {
      for (int q = 0; q < p.Hq; ++q) {
        load_row(q + 2);   // consumed behind this step's MFMAs (a row past the image is zeros)
        SDIAG(0);
        stage_c(q);
        SDIAG(2);
        __syncthreads();   // Y complete; every wave is done with row q - 1
        SDIAG(3);
        stage_h(q);
        SDIAG(4);
        store_row(q + 2);  // into the slot row q - 1 leaves
        SDIAG(5);
        __syncthreads();   // Z complete, ring updated
        SDIAG(6);
        stage_o(q);
        SDIAG(7);
      }
      __syncthreads();     // the last copy-out has read Z / the ring is free for the next job's prologue
    }
  }
  SDIAG_END;
}

int launch_strip_last(const StripLastP& p, int N, hipStream_t st) {
  if (tl_hi_only && N == 32) M2H_LAUNCH((convT_last_strip_kernel<32, true>), dim3(p.jobs_padded < 256 ? p.jobs_padded : 256), dim3(512), 0, st, p);
  else if (tl_hi_only) M2H_LAUNCH((convT_last_strip_kernel<16, true>), dim3(p.jobs_padded < 512 ? p.jobs_padded : 512), dim3(256), 0, st, p);
  else if (N == 32) M2H_LAUNCH((convT_last_strip_kernel<32>), dim3(p.jobs_padded < 256 ? p.jobs_padded : 256), dim3(512), 0, st, p);
  else M2H_LAUNCH((convT_last_strip_kernel<16>), dim3(p.jobs_padded < 512 ? p.jobs_padded : 512), dim3(256), 0, st, p);
  return launch_status(N == 32 ? "strip_convT_last<32>" : "strip_convT_last<16>");
}

#ifdef M2H_CLOCK_DIAG
extern "C" int m2h_diag_read_clocks_strip(unsigned long long* host_out, int nblocks) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_clock_dbg_strip), (size_t)nblocks * 80 * sizeof(unsigned long long));
}
#endif

}  // namespace m2h

using namespace m2h;

extern "C" {

size_t m2h_strip_conv1_weight_bytes(void) { return (size_t)4 * 16 * 2 * 64 * 16; }

int m2h_pack_strip_conv1(const float* w, int Ci, void* out, m2h_stream stream) {
  M2H_REQUIRE(w && out && Ci >= 32, "pack_strip_conv1: bad arguments");
  M2H_LAUNCH(pack_strip_conv1_kernel, dim3(32), dim3(256), 0, as_stream(stream), w, Ci, static_cast<f32x4*>(out));
  return launch_status("pack_strip_conv1");
}

int m2h_strip_conv1_fwd(const float* mix, const float* masks, const void* wreg, const float* scale, const float* shift,
                        const float* cls_table, const float* cls_val, float* dst, int B, int F, int T, float slope, m2h_stream stream) {
  M2H_REQUIRE(mix && wreg && scale && shift && dst, "strip_conv1: null pointer");
  M2H_REQUIRE(B > 0 && F == 512 && T >= 64 && T % 64 == 0, "strip_conv1: F must be 512 and T a multiple of 64 (got %d x %d)", F, T);
  M2H_REQUIRE((cls_table == nullptr) == (cls_val == nullptr), "strip_conv1: class table / value mismatch");
  M2H_REQUIRE((size_t)B * 512 * T * 2 < (1ull << 31), "strip_conv1: input too large for 32-bit pixel arithmetic");
  return launch_strip_conv1(mix, masks, wreg, scale, shift, cls_table, cls_val, dst, B, T, slope, as_stream(stream), 0);
}

int m2h_strip_last_fwd(const float* x, const float* skip, const float* wp_split32, const float* scale, const float* shift, const float* head_w,
                       const float* head_b, float* out, int B, int H, int W, int Co, m2h_stream stream) {
  M2H_REQUIRE(x && skip && wp_split32 && scale && shift && head_w && head_b && out, "strip_last: null pointer");
  M2H_REQUIRE(B > 0 && H >= 1 && W >= 32 && W % 32 == 0 && (Co == 32 || Co == 16), "strip_last: W must be a multiple of 32 and Co 32 or 16 (got %d x %d, %d)", H, W, Co);
  M2H_REQUIRE((long)B * H * W * 64 < (1L << 31), "strip_last: input too large for 32-bit pixel arithmetic");
  StripLastP p;
  p.src0 = x; p.src1 = skip; p.w = wp_split32; p.scale = scale; p.shift = shift; p.head_w = head_w; p.head_b = head_b; p.dst = out;
  p.B = B; p.Hq = H; p.Wq = W; p.strips = W / SW; p.jobs = B * p.strips; p.slope = 0.f;
  // last image first: the binaural head's consumer (the monaural U-Net's masked first stage) walks the images upwards and then finds the masks
  // written last still in the memory-side cache (268 MB of masks against its 256 MB: in the same direction every read misses).  Measured
  // on the masked first stage: 158.9 -> 145.7 us (profiles/r05_strip_rev_ab.txt); knob 39 bit 1 = the old direction.
  p.rev = ((g_strip_rev >> 1) & 1) ^ 1;
  p.jobs_padded = (B + 7) / 8 * 8 * p.strips;
  return launch_strip_last(p, Co, as_stream(stream));
}

}  // extern "C"
