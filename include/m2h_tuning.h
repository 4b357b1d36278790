/* libm2h diagnostic surface -- NOT part of the product contract of include/m2h.h.
 *
 * Tuning / test knobs of the library's dispatch code: they only choose between kernels that compute the same values (0 = automatic
 * everywhere).  tests/ use them to pit one engine against another, tools/ for A/B timing, __graft_entry__.smoke() to force the
 * benchmark batch's engine at a small batch.  The state is THREAD-LOCAL (like m2h_set_math_mode: the library holds no process-global
 * mutable state; two host threads may A/B engines side by side): a launch reads the knobs of the host thread that makes it.
 * m2h_tuning_snapshot / m2h_tuning_restore copy the calling thread's whole state (M2H_TUNING_KNOBS ints) out / in: m2h.functional
 * uses them to carry a forward pass's knobs into the autograd thread that runs its backward.
 *
 * Knobs: 0 force split-K factor (-1 never), 1 / 2 LDS stages of the narrow / wide tiles,
 * 3 skinny-M tiles (-1 off), 4 16-wide MFMA tile (-1 off), 7 extra dynamic LDS, 8 phase-major transposed-conv order (-1 off),
 * 9 scalar-decode loader (-1 off), 11 weight-gradient block target, 14 = m2h_set_math_mode (kept for older callers; thread-local
 * like it), 15 / 16 tap-sharing transposed-conv kernel (-1 off, 2 = only for N <= 32 / tile: 128, 256, 512), 18 tap window (-1
 * off), 21 / 22 image-row 3x3 weight-gradient / conv kernels (-1 off), 23 skinny rows kernel for M <= 16 (-1 off), 24 skinny
 * gather kernel (-1 off, > 0 = pixel limit), 26 the 256 x 128 eight-wave tile of the bf16x3 arithmetic (-1 off, > 0 = minimum
 * tile count), 27 the LDS-DMA engine for split32 operands (csrc/conv_dma.hip; -1 off, 2 = below the tile-count threshold too),
 * 28 = 32: 32x32x16 instead of 16x16x32 MFMA fragments there, 30 the four-phase transposed-conv kernel (csrc/convt_quad.hip; -1
 * off, 1 = wherever its shape conditions hold), 34 = -1: no split-K launches of the LDS-DMA / shared-patch engines (two K-halves, K-parts of the deepest stages), 35 = -1: the whole-network
 * runner does not take the strip-walker kernels (csrc/conv_strip.hip), 36 the shared-patch LDS-DMA engine (csrc/conv_patch.hip; -1
 * off, 2 = below the tile-count threshold too, 3 = as 2 with the whole-image patch wherever it fits, 8 = one workgroup per tile instead of one per CU walking its tiles: A/B), 10 = n >= 8: the shared-patch engine's persistent launches take n workgroups instead of one per CU (tests, 25 = -1: weight gradients of layers with at most 1024 rows keep the 128-wide blocks (0: 64-wide, twice as many), 33 = -1: the skinny gather kernel does not take layers of 1024-4096 pixels with tiny weights (they go to the tiled engine's split-K launches), 39 walking direction of the strip kernels' images (bit 0: the masked first stage downwards, bit 1: the last stage upwards instead of downwards, bit 2: the unmasked first stage downwards; same values in any direction), 38 the skinny gather kernel's 16-row blocks (-1 never, 1 always; 0 = where 32-row blocks would leave most CUs empty).  Numbers of experiments that were measured and removed
 * (5, 6, 13, 17, 19, 20, 29, 31, 32) are accepted and ignored.  12 = -1: narrow weight-gradient blocks always take three k sub-tiles when K allows (0: two where that leaves fewer padding columns).  37 = -1: train-mode BatchNorm always takes its three-launch path (0: layers of at most 256 rows take one launch per direction, > 0: layers of at most that many rows, up to 4 096; csrc/bn.hip). */
#ifndef M2H_TUNING_H
#define M2H_TUNING_H
#ifdef __cplusplus
extern "C" {
#endif
#define M2H_TUNING_KNOBS 40
int m2h_tuning_set(int knob, int value);
int m2h_tuning_snapshot(int* out, int n /* == M2H_TUNING_KNOBS */);
int m2h_tuning_restore(const int* in, int n /* == M2H_TUNING_KNOBS */);
#ifdef __cplusplus
}
#endif
#endif
