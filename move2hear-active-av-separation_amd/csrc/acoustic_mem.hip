// AcousticMem forward at the rollout batch in ONE launch (memory_nets.py:40-69, DD-PPO variant: slice both inputs 16-way, concat,
// Conv2d(32, 32, 3, padding 1) + ReLU, Conv2d(32, 16, 3, padding 1), de-slice; ppo_trainer.py:310-319, :367-373), fp32 MFMA.
//
// As separate launches this was five kernels per call (slice + two tiled convs with their split-K reduces: ~31 us of a ~590 us
// rollout step for 0.4 GFLOP).  Here a workgroup owns R = 2 output rows of one image:
//   * the 6 input rows x 34 columns (zero halo) x 32 sliced channels are gathered straight from the two BHWC tensors into an LDS
//     patch [pixel][36 floats] (the not-done mask of the previous memory applied on the way: ppo_trainer.py:310-314);
//   * each wave keeps its B fragments of BOTH weight matrices' current layer in registers for the whole tile (lane (i, kq) holds 16
//     bytes of output channel i's row per 16-channel chunk: four consecutive v_mfma_f32_16x16x4_f32 worth), so the k-loop is
//     one ds_read_b128 per A fragment and four MFMAs per n-tile, no weight traffic;
//   * the hidden layer's 4 rows (the tile's two rows + one halo row each side, recomputed per tile: 2 x the first conv's work
//     instead of a round trip through HBM and a second launch) go to a second LDS patch after the ReLU -- rows outside the image
//     are ZERO there (they are the second conv's padding, not conv outputs) --
//   * and the second conv's 16 x 16 accumulator tile is stored de-sliced: a lane holds four consecutive time frames of one
//     frequency band, one 16-byte store each.
// Same sums as Conv2d up to fp32 association (tests/test_gpu_rl.py: against the oracle and the tiled path).
#include "igemm_common.h"

namespace m2h {

constexpr int MEM_R = 2;            // output rows per workgroup
constexpr int MEM_PC = 34;          // patch columns (32 + halo)
constexpr int MEM_PITCH = 36;       // floats per patch pixel (32 channels + 4: conflict-free b128 fragment reads)

__global__ __launch_bounds__(256) void acoustic_mem_small_kernel(const float* __restrict__ mono, const float* __restrict__ prev,
                                                                 const float* __restrict__ not_done, const float* __restrict__ w0,
                                                                 const float* __restrict__ w1, float* __restrict__ out, int B) {
  __shared__ __align__(16) float xin[(MEM_R + 4) * MEM_PC * MEM_PITCH];   // input rows r0-2 .. r0+R+1
  __shared__ __align__(16) float hid[(MEM_R + 2) * MEM_PC * MEM_PITCH];   // hidden rows r0-1 .. r0+R
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int b = blockIdx.x / (32 / MEM_R), r0 = (blockIdx.x % (32 / MEM_R)) * MEM_R;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- first conv's weights: this wave's B fragments for both 16-channel halves of the 32 hidden channels, all 18 chunks
  f32x4 wa[18], wb[18];
#pragma unroll
  for (int c = 0; c < 18; ++c) {
    wa[c] = *reinterpret_cast<const f32x4*>(w0 + (size_t)i * 288 + c * 16 + 4 * kq);
    wb[c] = *reinterpret_cast<const f32x4*>(w0 + (size_t)(16 + i) * 288 + c * 16 + 4 * kq);
  }

  // ---- input patch: channel c < 16 = band c of pred_mono, c >= 16 = band c - 16 of the previous memory x not-done flag.
  // A (row, channel) pair is one 128-byte row of its BHWC tensor: 8 lanes x 16 bytes, transposed into the pixel-major patch.
  const float nd = not_done != nullptr ? not_done[b] : 1.f;
  for (int it = tid; it < (MEM_R + 4) * 32 * 8; it += 256) {
    const int q = it & 7, c = (it >> 3) & 31, pr = it >> 8;
    const int ih = r0 - 2 + pr;
    f32x4 v = zero4;
    if ((unsigned)ih < 32u) {
      const float* src = (c < 16 ? mono : prev) + ((size_t)b * 512 + (size_t)(c & 15) * 32 + ih) * 32 + q * 4;
      v = *reinterpret_cast<const f32x4*>(src);
      if (c >= 16) v = v * nd;
    }
    float* d = xin + (pr * MEM_PC + 1 + q * 4) * MEM_PITCH + c;
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e * MEM_PITCH] = v[e];
  }
  for (int it = tid; it < (MEM_R + 4) * 2 * 8; it += 256) {      // halo columns 0 and 33
    const int q = it & 7, side = (it >> 3) & 1, pr = it >> 4;
    *reinterpret_cast<f32x4*>(xin + (pr * MEM_PC + side * 33) * MEM_PITCH + q * 4) = zero4;
  }
  for (int it = tid; it < (MEM_R + 2) * 2 * 8; it += 256) {
    const int q = it & 7, side = (it >> 3) & 1, pr = it >> 4;
    *reinterpret_cast<f32x4*>(hid + (pr * MEM_PC + side * 33) * MEM_PITCH + q * 4) = zero4;
  }
  __syncthreads();

  // ---- first conv + ReLU: hidden rows r0-1 .. r0+R = (R + 2) x 32 pixels = 8 m-tiles of 16; two per wave, both channel halves
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int mt = wave * 2 + t;                    // m-tile: hidden patch row mt / 2, columns (mt & 1) * 16 ...
    const int hr = mt >> 1, x0 = (mt & 1) * 16;
    f32x4 acc0 = zero4, acc1 = zero4;
    const float* abase = xin + (hr * MEM_PC + x0 + i) * MEM_PITCH + 4 * kq;     // tap (0, 0) of this lane's pixel
#pragma unroll
    for (int c = 0; c < 18; ++c) {
      const int tap = c >> 1, th = tap / 3, tw = tap - th * 3;
      const f32x4 a = *reinterpret_cast<const f32x4*>(abase + (th * MEM_PC + tw) * MEM_PITCH + (c & 1) * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], wa[c][e], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], wb[c][e], acc1, 0, 0, 0);
      }
    }
    // D[pixel kq*4 + e][channel i]; rows of the hidden patch outside the image are the second conv's zero padding
    const bool inside = (unsigned)(r0 - 1 + hr) < 32u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float* d = hid + (hr * MEM_PC + 1 + x0 + kq * 4 + e) * MEM_PITCH;
      d[i] = inside ? fmaxf(acc0[e], 0.f) : 0.f;
      d[16 + i] = inside ? fmaxf(acc1[e], 0.f) : 0.f;
    }
  }
  // ---- second conv's weights replace the first's in the same registers
#pragma unroll
  for (int c = 0; c < 18; ++c) wa[c] = *reinterpret_cast<const f32x4*>(w1 + (size_t)i * 288 + c * 16 + 4 * kq);
  __syncthreads();

  // ---- second conv: R x 32 pixels = 4 m-tiles, one per wave; de-sliced store (memory_nets.py:62-67)
  {
    const int orow = wave >> 1, x0 = (wave & 1) * 16;
    f32x4 acc = zero4;
    const float* abase = hid + (orow * MEM_PC + x0 + i) * MEM_PITCH + 4 * kq;
#pragma unroll
    for (int c = 0; c < 18; ++c) {
      const int tap = c >> 1, th = tap / 3, tw = tap - th * 3;
      const f32x4 a = *reinterpret_cast<const f32x4*>(abase + (th * MEM_PC + tw) * MEM_PITCH + (c & 1) * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], wa[c][e], acc, 0, 0, 0);
    }
    // lane (band i, kq): time frames x0 + 4 kq .. + 3 of frequency row band * 32 + (r0 + orow)
    *reinterpret_cast<f32x4*>(out + ((size_t)b * 512 + (size_t)i * 32 + r0 + orow) * 32 + x0 + kq * 4) = acc;
  }
}

}  // namespace m2h

extern "C" int m2h_acoustic_mem_small_fwd(const float* pred_mono, const float* prev_mem, const float* not_done, const float* w0p,
                                          const float* w1p, float* out, int B, int F, int T, m2h_stream stream) {
  M2H_REQUIRE(pred_mono && prev_mem && w0p && w1p && out, "acoustic_mem_small: null pointer");
  M2H_REQUIRE(B > 0 && F == 512 && T == 32, "acoustic_mem_small: built for [B, 512, 32, 1] spectrograms (got %d x %d)", F, T);
  M2H_LAUNCH(m2h::acoustic_mem_small_kernel, dim3(B * (32 / m2h::MEM_R)), dim3(256), 0, m2h::as_stream(stream), pred_mono, prev_mem,
                     not_done, w0p, w1p, out, B);
  return m2h::launch_status("acoustic_mem_small");
}
