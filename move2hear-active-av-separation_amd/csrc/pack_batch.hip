// Batched weight (re)packing (gfx950): every packed operand a training step needs -- forward packs of Conv2d / ConvTranspose2d
// weights, input-gradient packs of both -- produced by ONE launch from a table of items.  After an optimizer step every weight
// has changed, so a passive pre-training step re-packed 42 tensors with 42 launches of 10-16 us (0.56 ms of a 4.4 ms step) and
// a DD-PPO policy epoch 30 (0.15 ms of 2.4 ms); the work itself is ~0.5 GB of HBM traffic.  grid = (blocks, items); the
// index maps are those of the single-tensor kernels (layout.hip, conv_bwd.hip), bit-identical results.
#include "m2h_internal.h"

namespace m2h {

// Every pack is, for a fixed "middle" index m, a transpose of a [outer][T] matrix whose T = KH*KW taps are contiguous in the
// torch layout into rows of `outer` contiguous elements in the packed layout (outer = input channel for the forward packs,
// output channel for the input-gradient packs).  A block takes (m, 64 outer indices): coalesced reads of 64 runs of T floats
// into LDS, coalesced writes of T runs of 64 floats.  (The per-element index maps of the single-tensor kernels read with a
// stride of T floats: 16x read amplification, 10-67 us per tensor.)
struct PackBatchArgs {
  m2h_pack_item item[M2H_PACK_BATCH_MAX];
  unsigned first_block[M2H_PACK_BATCH_MAX + 1];   // prefix sums of the items' block counts
  int n_items;
};

constexpr int PACK_TILE = 64;       // outer indices per block
constexpr int PACK_TMAX = 144;      // taps per weight (12 x 12: VisualCNN's full-spatial Linear)

__global__ __launch_bounds__(256) void pack_batch_kernel(const PackBatchArgs a) {
  __shared__ float tile[PACK_TILE][PACK_TMAX + 1];
  int idx = 0;
  while (idx + 1 < a.n_items && blockIdx.x >= a.first_block[idx + 1]) ++idx;   // block-uniform
  const m2h_pack_item it = a.item[idx];
  const unsigned b = blockIdx.x - a.first_block[idx];
  const float* __restrict__ w = it.src;
  float* __restrict__ wp = it.dst;
  int n_outer_src, n_outer_dst, T, KW;
  size_t src_mid, src_outer;
  if (it.kind == M2H_PACK_CONVT) {             // m = co, outer = ci: w[ci][co][4][4]
    T = 16; KW = 4;
    n_outer_src = n_outer_dst = it.p[0];
    src_mid = 16; src_outer = (size_t)it.p[1] * 16;
  } else if (it.kind == M2H_PACK_CONV) {       // m = co, outer = ci: w[co][ci][KH][KW]
    T = it.p[2] * it.p[3]; KW = it.p[3];
    n_outer_src = it.p[4]; n_outer_dst = it.p[5];
    src_mid = (size_t)it.p[1] * T; src_outer = T;
  } else {                                     // DGRAD / FC_DGRAD: m = ci, outer = co
    T = it.p[2] * it.p[3]; KW = it.p[3];
    n_outer_src = n_outer_dst = it.p[0];
    src_mid = T; src_outer = (size_t)it.p[1] * T;
  }
  const int tiles = (n_outer_dst + PACK_TILE - 1) / PACK_TILE;
  if (T == 1) {
    // 1 x 1 weights (nn.Linear: the GRU's 1536 x 1536 and 1536 x 512 matrices, the heads): with one tap a (m, 64 outer) block would move 64
    // floats -- 37 000 blocks for one GRU matrix, 157 us per policy epoch for 23 MB of weights.  Here a block takes 64 middle x 64 outer
    // indices: a padded copy (forward pack: both layouts run along the input channel) or a 64 x 64 transpose through LDS (gradient packs).
    const int n_mid = it.kind == M2H_PACK_CONV ? it.p[0] : (it.kind == M2H_PACK_DGRAD ? it.p[1] : it.p[5]);
    const int m0 = (b / tiles) * PACK_TILE, o0 = (b - (b / tiles) * tiles) * PACK_TILE;
    const bool along_outer = src_outer == 1;   // which index runs contiguously in the source
    for (int i = threadIdx.x; i < PACK_TILE * PACK_TILE; i += 256) {
      const int o = along_outer ? i & 63 : i >> 6, ml = along_outer ? i >> 6 : i & 63;
      const int og = o0 + o, m = m0 + ml;
      const bool ok = m < n_mid && og < n_outer_src && (it.kind != M2H_PACK_FC_DGRAD || m < it.p[4]);
      tile[o][ml] = ok ? w[(size_t)m * src_mid + (size_t)og * src_outer] : 0.f;
    }
    __syncthreads();
    const int ld = it.kind == M2H_PACK_CONV ? it.p[5] : it.p[0];   // packed row length: base(m, t = 0) = m * ld for all three kinds
    for (int i = threadIdx.x; i < PACK_TILE * PACK_TILE; i += 256) {
      const int o = i & 63, ml = i >> 6;
      const int og = o0 + o, m = m0 + ml;
      if (m < n_mid && og < n_outer_dst) wp[(size_t)m * ld + og] = tile[o][ml];
    }
    return;
  }
  const int m = b / tiles, o0 = (b - m * tiles) * PACK_TILE;
  const bool mid_valid = it.kind != M2H_PACK_FC_DGRAD || m < it.p[4];     // FC_DGRAD: padded input channels are zero rows
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  // every 4x4 kernel (all of both U-Nets): one 16-byte load and one 16-byte store per thread -- where the source weight is 16-byte
  // aligned (parameters sit back to back in FlatAdam's flat buffer: a 4x4 weight behind a 1- or 3-element bias is only 4-byte aligned;
  // block-uniform, so no divergence)
  const bool vec = T == 16 && (reinterpret_cast<size_t>(w) & 15) == 0;
  if (vec) {
    const int o = threadIdx.x >> 2, c = threadIdx.x & 3;
    const int og = o0 + o;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (mid_valid && og < n_outer_src) v = *reinterpret_cast<const f32x4*>(w + (size_t)m * src_mid + (size_t)og * src_outer + 4 * c);
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[o][4 * c + j] = v[j];
  } else {
    for (int i = threadIdx.x; i < PACK_TILE * T; i += 256) {
      const int o = i / T, t = i - o * T;
      const int og = o0 + o;
      tile[o][t] = (mid_valid && og < n_outer_src) ? w[(size_t)m * src_mid + (size_t)og * src_outer + t] : 0.f;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (vec ? 256 : PACK_TILE * T); i += 256) {
    const int t = vec ? i >> 4 : i / PACK_TILE, o = vec ? (i & 15) * 4 : i - t * PACK_TILE;
    const int og = o0 + o;
    if (og >= n_outer_dst) continue;
    const int kh = t / KW, kw = t - kh * KW;
    size_t base;
    if (it.kind == M2H_PACK_CONV) {
      base = ((size_t)m * T + t) * it.p[5];
    } else if (it.kind == M2H_PACK_CONVT) {    // kh -> (ph, th): 0 -> (1,1), 1 -> (0,0), 2 -> (1,0), 3 -> (0,1)
      const int ph = (kh & 1) ^ 1, th = kh == 0 || kh == 3, pw = (kw & 1) ^ 1, tw = kw == 0 || kw == 3;
      base = ((((size_t)(ph * 2 + pw) * it.p[1] + m) * 2 + th) * 2 + tw) * it.p[0];
    } else if (it.kind == M2H_PACK_DGRAD) {    // kh = (ph + pad) % s + s * th
      const int s = it.p[4], pad = it.p[5], th_n = it.p[2] / s, tw_n = it.p[3] / s;
      const int ph = ((kh % s - pad) % s + s) % s, pw = ((kw % s - pad) % s + s) % s;
      base = ((((size_t)(ph * s + pw) * it.p[1] + m) * th_n + kh / s) * tw_n + kw / s) * it.p[0];
    } else {
      base = ((size_t)t * it.p[5] + m) * it.p[0];
    }
    if (vec) {   // four consecutive outer indices
      if (og + 3 < n_outer_dst && ((base + og) & 3) == 0) {
        *reinterpret_cast<f32x4*>(wp + base + og) = f32x4{tile[o][t], tile[o + 1][t], tile[o + 2][t], tile[o + 3][t]};
      } else {
        for (int j = 0; j < 4 && og + j < n_outer_dst; ++j) wp[base + og + j] = tile[o + j][t];
      }
    } else {
      wp[base + og] = tile[o][t];
    }
  }
}

}  // namespace m2h

using namespace m2h;

extern "C" int m2h_pack_batch(const m2h_pack_item* items, int n_items, m2h_stream stream) {
  M2H_REQUIRE(items && n_items > 0 && n_items <= M2H_PACK_BATCH_MAX, "pack_batch: 1..%d items", M2H_PACK_BATCH_MAX);
  PackBatchArgs a;
  a.n_items = n_items;
  unsigned blocks = 0;
  for (int i = 0; i < n_items; ++i) {
    const m2h_pack_item& it = items[i];
    M2H_REQUIRE(it.src && it.dst && it.kind >= M2H_PACK_CONV && it.kind <= M2H_PACK_FC_DGRAD, "pack_batch: item %d: null pointer or unknown kind", i);
    int n_mid, n_outer, T;
    if (it.kind == M2H_PACK_CONVT) {
      M2H_REQUIRE(it.p[0] > 0 && it.p[1] > 0, "pack_batch: item %d: non-positive size", i);
      n_mid = it.p[1]; n_outer = it.p[0]; T = 16;
    } else {
      for (int k = 0; k < 4; ++k) M2H_REQUIRE(it.p[k] > 0, "pack_batch: item %d: non-positive size", i);
      T = it.p[2] * it.p[3];
      if (it.kind == M2H_PACK_CONV) {
        M2H_REQUIRE(it.p[4] > 0 && it.p[4] <= it.p[1] && it.p[5] >= it.p[4], "pack_batch: item %d: ci_used / ci_out", i);
        n_mid = it.p[0]; n_outer = it.p[5];
      } else if (it.kind == M2H_PACK_DGRAD) {
        M2H_REQUIRE(it.p[4] > 0 && it.p[2] % it.p[4] == 0 && it.p[3] % it.p[4] == 0 && it.p[5] >= 0, "pack_batch: item %d: kernel %% stride", i);
        n_mid = it.p[1]; n_outer = it.p[0];
      } else {
        M2H_REQUIRE(it.p[4] > 0 && it.p[4] <= it.p[1] && it.p[5] >= it.p[4], "pack_batch: item %d: ci_used / ci_out", i);
        n_mid = it.p[5]; n_outer = it.p[0];
      }
    }
    M2H_REQUIRE(T <= PACK_TMAX, "pack_batch: item %d: %d taps per weight (max %d)", i, T, PACK_TMAX);
    a.item[i] = it;
    a.first_block[i] = blocks;
    blocks += (unsigned)(T == 1 ? (n_mid + PACK_TILE - 1) / PACK_TILE : n_mid) * (unsigned)((n_outer + PACK_TILE - 1) / PACK_TILE);
  }
  a.first_block[n_items] = blocks;
  M2H_LAUNCH(pack_batch_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), a);
  return launch_status("pack_batch");
}
