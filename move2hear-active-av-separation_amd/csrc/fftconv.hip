// RIR convolution of the audio feeder (gfx950): scipy.signal.fftconvolve(mono, rir[:, ear]) for both ears of one (clip, source)
// pair per workgroup (pretrain/datasets/dataset.py:180, habitat_audio/simulator_train.py:419), as hand-written FFTs in LDS.
//
// The linear convolution needs N >= L + Lr - 1 points (32 768 for one-second clips and RIRs).  A real N-point transform is one
// complex M = N/2-point transform of z[m] = x[2m] + i x[2m+1] plus an O(N) unpack; M = 16 384 complex fp32 values are 128 KB,
// so the whole transform lives in the CU's 160 KB of LDS: 1024 threads, radix-2, 14 stages, one barrier per stage.
//   forward : decimation in frequency, natural order in -> bit-reversed order out (no reordering pass);
//   spectrum: the pair (k, M-k) is all that the unpack of X, H, the product Y = X H and the re-pack for the inverse need, so
//             each thread turns Z_h[k], Z_h[M-k] (LDS) and Z_x[k], Z_x[M-k] (global scratch, written once per source) into
//             Z'[k], Z'[M-k] in place -- still in bit-reversed slots;
//   inverse : the same butterfly network run backwards with conjugate twiddles: bit-reversed in -> natural order out.
// Twiddles exp(-2 pi i k / N) come from a table built on the host in float64 (setup data, like the DFT matrices of the STFT).
// Arithmetic fp32 throughout, like the library transform it replaces (rocFFT through torch.fft in round 1).
#include "m2h_internal.h"

namespace m2h {

namespace {
struct cf {
  float x, y;
};
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cf cmulc(cf a, cf b) { return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }   // a * conj(b)
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf conj(cf a) { return {a.x, -a.y}; }
}  // namespace

// z: M complex values in LDS.  tw[k] = exp(-2 pi i k / (2M)), k < M.
template <int LOGM>
__device__ __forceinline__ void fft_dif(cf* z, const cf* __restrict__ tw) {
  constexpr int M = 1 << LOGM;
  for (int s = 0; s < LOGM; ++s) {
    const int half = M >> (s + 1);
    for (int j = threadIdx.x; j < M / 2; j += blockDim.x) {
      const int g = j / half, pos = j - g * half;
      const int i0 = g * 2 * half + pos, i1 = i0 + half;
      const cf a = z[i0], b = z[i1];
      z[i0] = cadd(a, b);
      z[i1] = cmul(csub(a, b), tw[(pos << s) * 2]);
    }
    __syncthreads();
  }
}

template <int LOGM>
__device__ __forceinline__ void ifft_dit(cf* z, const cf* __restrict__ tw) {
  constexpr int M = 1 << LOGM;
  for (int s = LOGM - 1; s >= 0; --s) {
    const int half = M >> (s + 1);
    for (int j = threadIdx.x; j < M / 2; j += blockDim.x) {
      const int g = j / half, pos = j - g * half;
      const int i0 = g * 2 * half + pos, i1 = i0 + half;
      const cf a = z[i0], b = cmulc(z[i1], tw[(pos << s) * 2]);
      z[i0] = cadd(a, b);
      z[i1] = csub(a, b);
    }
    __syncthreads();
  }
}

template <int LOGM>
__global__ __launch_bounds__(1024) void fftconv_kernel(const float* __restrict__ mono, const float* __restrict__ rirs, const cf* __restrict__ tw,
                                                       cf* __restrict__ xspec, float* __restrict__ full, int L, int Lr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf* z = reinterpret_cast<cf*>(smem);
  constexpr int M = 1 << LOGM, N = 2 * M;
  const int cs = blockIdx.x, tid = threadIdx.x;
  const int ear0 = blockIdx.y;   // one workgroup per (clip, source, ear): 2 x CS workgroups fill the chip at the feeder's batch (64 x 2
                                 // sources); each redoes the clip's own transform (3 transforms per workgroup instead of 5 in one)
  auto br = [](int k) { return (int)(__brev((unsigned)k) >> (32 - LOGM)); };

  // ---- spectrum of the mono clip (packed, bit-reversed), kept in global scratch for the two ears ----
  const float* x = mono + (size_t)cs * L;
  for (int m = tid; m < M; m += 1024) z[m] = {2 * m < L ? x[2 * m] : 0.f, 2 * m + 1 < L ? x[2 * m + 1] : 0.f};
  __syncthreads();
  fft_dif<LOGM>(z, tw);
  cf* zx = xspec + ((size_t)cs * 2 + ear0) * M;
  for (int m = tid; m < M; m += 1024) zx[m] = z[m];
  __threadfence();   // the block reads zx back below (other threads' elements): stores drained to L2 before the barrier
  __syncthreads();

  for (int ear = ear0; ear <= ear0; ++ear) {
    const float* h = rirs + (size_t)cs * Lr * 2 + ear;
    for (int m = tid; m < M; m += 1024) z[m] = {2 * m < Lr ? h[(size_t)(2 * m) * 2] : 0.f, 2 * m + 1 < Lr ? h[(size_t)(2 * m + 1) * 2] : 0.f};
    __syncthreads();
    fft_dif<LOGM>(z, tw);
    // ---- per pair (k, M-k): unpack X and H, multiply, re-pack for the inverse; in place, bit-reversed slots ----
    for (int k = tid; k <= M / 2; k += 1024) {
      if (k == 0) {
        const cf zh = z[0], zxx = zx[0];
        const float y0 = (zxx.x + zxx.y) * (zh.x + zh.y), ym = (zxx.x - zxx.y) * (zh.x - zh.y);   // X[0] H[0], X[M] H[M] (all real)
        z[0] = {0.5f * (y0 + ym), 0.5f * (y0 - ym)};
        continue;
      }
      const int ik = br(k), im = br(M - k);
      const cf t = tw[k];
      const cf zhk = z[ik], zhm = z[im], zxk = zx[ik], zxm = zx[im];
      // E = (Zk + conj Zm)/2, T O = T * (-i/2)(Zk - conj Zm);  S[k] = E + T O, S[M-k] = conj(E - T O)
      auto unpack = [&](cf zk, cf zm, cf& e, cf& to) {
        e = {0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y)};
        const cf d = {0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y)};   // (Zk - conj Zm)/2
        to = cmul(t, cf{d.y, -d.x});                                   // * (-i)
      };
      cf ex, ox, eh, oh;
      unpack(zxk, zxm, ex, ox);
      unpack(zhk, zhm, eh, oh);
      const cf yk = cmul(cadd(ex, ox), cadd(eh, oh));                  // Y[k]
      const cf ymc = cmul(csub(ex, ox), csub(eh, oh));                 // conj(Y[M-k])
      const cf e2 = {0.5f * (yk.x + ymc.x), 0.5f * (yk.y + ymc.y)};    // E' = (Y[k] + conj Y[M-k]) / 2
      const cf w2 = {0.5f * (yk.x - ymc.x), 0.5f * (yk.y - ymc.y)};
      const cf o2 = cmulc(w2, t);                                      // O' = conj(T) (Y[k] - conj Y[M-k]) / 2
      z[ik] = {e2.x - o2.y, e2.y + o2.x};                              // Z'[k] = E' + i O'
      if (im != ik) z[im] = {e2.x + o2.y, -e2.y + o2.x};               // Z'[M-k] = conj(E') + i conj(O')
    }
    __syncthreads();
    ifft_dit<LOGM>(z, tw);
    float* out = full + ((size_t)cs * 2 + ear) * N;
    const float sc = 1.f / (float)M;
    for (int m = tid; m < M; m += 1024) {
      const cf v = z[m];
      *reinterpret_cast<float2*>(out + 2 * m) = make_float2(v.x * sc, v.y * sc);
    }
    __syncthreads();
  }
}

template <int LOGM>
static int launch_fftconv(const float* mono, const float* rirs, const float* tw, float* xspec, float* full, int CS, int L, int Lr, hipStream_t st) {
  const size_t lds = sizeof(cf) << LOGM;
  auto kern = fftconv_kernel<LOGM>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail((int)e, "fftconv_same: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
  }
  M2H_LAUNCH(kern, dim3(CS, 2), dim3(1024), lds, st, mono, rirs, reinterpret_cast<const cf*>(tw), reinterpret_cast<cf*>(xspec), full, L, Lr);
  return launch_status("fftconv_same");
}

}  // namespace m2h

using namespace m2h;

extern "C" int m2h_fftconv_full(const float* mono, const float* rirs, const float* twiddles, float* xspec, float* full, int CS, int L, int Lr,
                                int log2n, m2h_stream stream) {
  M2H_REQUIRE(mono && rirs && twiddles && xspec && full && CS > 0 && L > 0 && Lr > 0, "fftconv_full: bad arguments");
  M2H_REQUIRE(log2n >= 11 && log2n <= 15, "fftconv_full: transform length 2^%d outside 2^11 .. 2^15 (the packed transform must fit 128 KB of LDS)", log2n);
  M2H_REQUIRE((long)L + Lr - 1 <= (1L << log2n), "fftconv_full: 2^%d points cannot hold a %d + %d - 1 point linear convolution", log2n, L, Lr);
  hipStream_t st = as_stream(stream);
  switch (log2n) {
    case 11: return launch_fftconv<10>(mono, rirs, twiddles, xspec, full, CS, L, Lr, st);
    case 12: return launch_fftconv<11>(mono, rirs, twiddles, xspec, full, CS, L, Lr, st);
    case 13: return launch_fftconv<12>(mono, rirs, twiddles, xspec, full, CS, L, Lr, st);
    case 14: return launch_fftconv<13>(mono, rirs, twiddles, xspec, full, CS, L, Lr, st);
    default: return launch_fftconv<14>(mono, rirs, twiddles, xspec, full, CS, L, Lr, st);
  }
}
