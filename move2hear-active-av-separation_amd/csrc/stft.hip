// Feeder STFT and eval iSTFT on the GPU (gfx950, fp32) -- rows N1 / N2 of SURVEY 8f (A20, A21).
//
// Reference semantics (third-party, not under the reference tree): librosa==0.8.0 `stft(y, n_fft=1023, hop_length=512)`
// as called at audio_separation/pretrain/datasets/dataset.py:190-226 and habitat_audio/simulator_train.py:425-481
// (win_length = n_fft, periodic Hann, center=True with reflect padding n_fft//2, 1 + len//hop frames, 512 bins), followed by
// np.abs / np.angle and log1p (dataset.py:228; the simulator rounds the magnitude to fp16 first, simulator_train.py:437-441);
// and `istft(mag*exp(j*phase), hop_length=512, length=16000)` at common/eval_metrics.py:232-251 (n_fft inferred 2*(512-1) =
// 1022, periodic Hann(1022), overlap-add, division by the window sum-of-squares where > tiny, trim n_fft//2, fix length).
//
// Structure: 1023 (and 1022) are not powers of two and the transforms are tiny, so the DFT is a dense [frames x 1024] x
// [1024 x 1024] fp32 GEMM on the MFMA engine (m2h_conv_igemm_f32 as a Linear; the cos/-sin matrix is built once on the host
// in float64).  The kernels here are the HBM-bound glue: framing with reflect padding + window, magnitude/phase/log1p with
// the BHWC store, and the iSTFT pre/post (complex assembly, windowed overlap-add as a gather -- no atomics).
#include "m2h_internal.h"

namespace m2h {

// frames[s][t][n] = n < n_fft ? window[n] * y_reflect[s][t*hop + n - n_fft/2] : 0     (row length ldf >= n_fft, zero padded)
__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ y, const float* __restrict__ window, float* __restrict__ frames,
                                                          int S, int L, int T, int n_fft, int hop, int ldf) {
  const size_t total = (size_t)S * T * ldf;
  const int pad = n_fft / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % ldf);
    const size_t r = i / ldf;
    const int t = (int)(r % T);
    const int s = (int)(r / T);
    float v = 0.f;
    if (n < n_fft) {
      int idx = t * hop + n - pad;
      if (idx < 0) idx = -idx;                       // np.pad(mode="reflect"): edge sample not repeated
      if (idx >= L) idx = 2 * (L - 1) - idx;
      v = window[n] * y[(size_t)s * L + idx];
    }
    frames[i] = v;
  }
}

// spec[s][t][0..nb) = Re, [nb..2nb) = Im  ->  mag_out[b][k][t][c] = f(|X|), phase_out likewise (signal s = b*C + c)
// f: mode 0 |X|; 1 log1p(|X|); 2 log1p(fp16(|X|)) (simulator path).  Outputs are BHWC [B][nb][T][C]; either may be NULL.
__global__ __launch_bounds__(256) void stft_post_kernel(const float* __restrict__ spec, float* __restrict__ mag_out, float* __restrict__ phase_out,
                                                        int B, int C, int T, int nb, int lds, int mode) {
  const size_t total = (size_t)B * nb * T * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t r = i / C;
    const int t = (int)(r % T);
    r /= T;
    const int k = (int)(r % nb);
    const int b = (int)(r / nb);
    const float* row = spec + ((size_t)(b * C + c) * T + t) * lds;
    const float re = row[k], im = row[nb + k];
    if (mag_out != nullptr) {
      float m = sqrtf(re * re + im * im);
      if (mode == 2) m = (float)(_Float16)m;
      if (mode >= 1) m = log1pf(m);
      mag_out[i] = m;
    }
    if (phase_out != nullptr) phase_out[i] = atan2f(im, re);
  }
}

// iSTFT pre: rows[s][t][k] = mag*cos(phase), rows[s][t][nb+k] = mag*sin(phase) from BHWC mag/phase (channel c of C; s = b)
__global__ __launch_bounds__(256) void istft_pre_kernel(const float* __restrict__ mag, const float* __restrict__ phase, float* __restrict__ rows,
                                                        int B, int C, int c, int T, int nb, int ldr) {
  const size_t total = (size_t)B * T * nb;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % nb);
    const size_t r = i / nb;
    const int t = (int)(r % T);
    const int b = (int)(r / T);
    const size_t src = (((size_t)b * nb + k) * T + t) * C + c;
    const float m = mag[src], p = phase[src];
    float* row = rows + ((size_t)b * T + t) * ldr;
    row[k] = m * cosf(p);
    row[nb + k] = m * sinf(p);
  }
}

// overlap-add as a gather: y[s][j] = (sum_t frames[s][t][jj - t*hop] * window[jj - t*hop]) / wss(jj),  jj = j + n_fft/2,
// wss(jj) = sum_t window^2[jj - t*hop] (divide only where > tiny, librosa.util.tiny(float32) = 1.1754944e-38)
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ window, float* __restrict__ y,
                                                        int S, int T, int n_fft, int hop, int ldf, int length) {
  const size_t total = (size_t)S * length;
  const int full = n_fft + hop * (T - 1);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % length);
    const int s = (int)(i / length);
    const int jj = j + n_fft / 2;
    float acc = 0.f, wss = 0.f;
    if (jj < full) {
      int t1 = jj / hop;
      if (t1 > T - 1) t1 = T - 1;
      for (int t = t1; t >= 0; --t) {
        const int n = jj - t * hop;
        if (n >= n_fft) break;
        const float w = window[n];
        acc += frames[((size_t)s * T + t) * ldf + n] * w;
        wss += w * w;
      }
    }
    y[i] = wss > 1.1754944e-38f ? acc / wss : acc;
  }
}


// Waveform quality metrics of common/eval_metrics.py:12-166 (scale_bss_eval for ONE reference source, the only case the
// reference evaluates: evaluate_helper passes references[..., 0, :] with a single source): one 1024-thread block per clip.
//   preprocess (:170-196): every signal has its mean removed; the mixture is the mean of its two mean-removed channels.
//   helper(s, x) (:12-58): alpha = <s,x>/<s,s>;  snr = 10log10(<s,s>/|x-s|^2);  si_sdr = 10log10(|alpha s|^2/|x-alpha s|^2);
//   srr = -10log10((1-1/alpha)^2);  sd_sdr = snr + 10log10(alpha^2);  b = <s, x-alpha s>/<s,s> + EPS;
//   si_sir = 10log10(|alpha s|^2/|s b|^2);  si_sar = 10log10(|alpha s|^2/|x - alpha s - s b + EPS|^2).
// out[clip][11] = si_sdr, si_sir, si_sar, sd_sdr, snr, srr of the estimate, then si_sdri, sd_sdri, snri, si_siri, si_sari
// (estimate minus the same metric of the mixture, :112-122).  Sums are fp32 per thread and double across the block.
// (si_sir / si_sar of a single source divide by rounding noise -- <s, x - alpha s> is zero in exact arithmetic -- in the
// reference as here; only their order of magnitude is meaningful.)
__device__ __forceinline__ double block_sum_d(double v, double* sh /* 16 */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += sh[i];
  return t;
}

__global__ __launch_bounds__(1024) void bss_metrics_kernel(const float* __restrict__ ref, const float* __restrict__ est,
                                                           const float* __restrict__ mixl, const float* __restrict__ mixr,
                                                           float* __restrict__ out, int L) {
  __shared__ double sh[16];
  const int c = blockIdx.x;
  const float* s = ref + (size_t)c * L;
  const float* e = est + (size_t)c * L;
  const float* ml = mixl + (size_t)c * L;
  const float* mr = mixr != nullptr ? mixr + (size_t)c * L : nullptr;
  const double EPS = 1e-13;
  // pass 1: means
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int i = threadIdx.x; i < L; i += 1024) {
    a0 += s[i];
    a1 += e[i];
    a2 += ml[i];
    if (mr != nullptr) a3 += mr[i];
  }
  const float ms = (float)(block_sum_d(a0, sh) / L), me = (float)(block_sum_d(a1, sh) / L);
  const float mml = (float)(block_sum_d(a2, sh) / L), mmr = (float)(block_sum_d(a3, sh) / L);
  auto S = [&](int i) { return s[i] - ms; };
  auto X = [&](int i, int which) {
    if (which == 0) return e[i] - me;
    const float l = ml[i] - mml;
    return mr != nullptr ? 0.5f * (l + (mr[i] - mmr)) : l;
  };
  // pass 2: <s,s>, <s,x>
  float ss_ = 0.f, sx0 = 0.f, sx1 = 0.f;
  for (int i = threadIdx.x; i < L; i += 1024) {
    const float sv = S(i);
    ss_ += sv * sv;
    sx0 += sv * X(i, 0);
    sx1 += sv * X(i, 1);
  }
  const double ss = block_sum_d(ss_, sh);
  const double sx[2] = {block_sum_d(sx0, sh), block_sum_d(sx1, sh)};
  double m[2][6];
  for (int w = 0; w < 2; ++w) {
    const double alpha = sx[w] / ss;
    const float af = (float)alpha;
    // pass 3: |x - s|^2, |x - alpha s|^2, <s, x - alpha s>
    float n1 = 0.f, n2 = 0.f, sr = 0.f;
    for (int i = threadIdx.x; i < L; i += 1024) {
      const float sv = S(i), xv = X(i, w);
      const float d1 = xv - sv, d2 = xv - sv * af;
      n1 += d1 * d1;
      n2 += d2 * d2;
      sr += sv * d2;
    }
    const double noise1 = block_sum_d(n1, sh), noise2 = block_sum_d(n2, sh);
    const double b = block_sum_d(sr, sh) / ss + EPS;
    const float bf = (float)b;
    // pass 4: |x - alpha s - s b + EPS|^2
    float ar = 0.f;
    for (int i = threadIdx.x; i < L; i += 1024) {
      const float sv = S(i);
      const float d = (X(i, w) - sv * af) - sv * bf + (float)EPS;
      ar += d * d;
    }
    const double artif = block_sum_d(ar, sh);
    const double signal2 = alpha * alpha * ss;
    const double snr = 10.0 * log10(ss / noise1);
    m[w][0] = 10.0 * log10(signal2 / noise2);                     // si_sdr
    m[w][1] = 10.0 * log10(signal2 / (b * b * ss));               // si_sir
    m[w][2] = 10.0 * log10(signal2 / artif);                      // si_sar
    m[w][3] = snr + 10.0 * log10(alpha * alpha);                  // sd_sdr
    m[w][4] = snr;                                                // snr
    m[w][5] = -10.0 * log10((1.0 - 1.0 / alpha) * (1.0 - 1.0 / alpha));  // srr
  }
  if (threadIdx.x == 0) {
    float* o = out + (size_t)c * 11;
    for (int j = 0; j < 6; ++j) o[j] = (float)m[0][j];
    o[6] = (float)(m[0][0] - m[1][0]);   // si_sdri
    o[7] = (float)(m[0][3] - m[1][3]);   // sd_sdri
    o[8] = (float)(m[0][4] - m[1][4]);   // snri
    o[9] = (float)(m[0][1] - m[1][1]);   // si_siri
    o[10] = (float)(m[0][2] - m[1][2]);  // si_sari
  }
}

// RIR-convolution feeder glue (dataset.py:178-186,214-216; simulator_train.py:416-424): the "same"-mode slice of the full
// convolution, np.round (half to even) -> int16 (wrapping, as numpy's astype) -> float32 * (1/32768); the per-source result
// is written (GT binaural waveform of source 0 feeds the GT spectrogram) and accumulated into the mixture, which the caller
// divides by the number of sources through `mix_scale` on the last source.
//   full [S][ldfull] = full linear convolution (S = clips * 2 channels), conv_out [S][L] (may be NULL), mix [S][L]
__global__ __launch_bounds__(256) void feeder_round_mix_kernel(const float* __restrict__ full, int ldfull, int start, float* __restrict__ conv_out,
                                                               float* __restrict__ mix, int S, int L, int first, float mix_scale) {
  const size_t total = (size_t)S * L;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % L);
    const size_t s = i / L;
    const float v = full[s * ldfull + start + n];
    const int r = (int)rintf(v);                 // np.round: half to even
    const float q = (float)(short)r * (1.f / 32768.f);
    if (conv_out != nullptr) conv_out[i] = q;
    const float acc = first ? q : mix[i] + q;
    mix[i] = acc * mix_scale;
  }
}

// gt mono magnitude normalisation (dataset.py:205-206): mag *= norm / sqrt(mean(mag^2)) when that RMS is non-zero.
// One 1024-thread block per clip over n = F*T elements.
__global__ __launch_bounds__(1024) void rms_normalize_kernel(float* __restrict__ mag, int n, float norm) {
  __shared__ double sh[16];
  float* p = mag + (size_t)blockIdx.x * n;
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) a += p[i] * p[i];
  const double ms = block_sum_d(a, sh) / n;
  const float rms = sqrtf((float)ms);
  if (rms != 0.f) {
    const float k = norm / rms;
    for (int i = threadIdx.x; i < n; i += 1024) p[i] *= k;
  }
}

static inline unsigned sgrid(size_t total) {
  size_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace m2h

using namespace m2h;

extern "C" {

int m2h_stft_frames(const float* y, const float* window, float* frames, int S, int L, int T, int n_fft, int hop, int ldf, m2h_stream stream) {
  M2H_REQUIRE(y && window && frames && S > 0 && L > n_fft / 2 && T > 0 && n_fft > 1 && hop > 0 && ldf >= n_fft, "stft_frames: bad arguments");
  M2H_REQUIRE((T - 1) * hop + n_fft - n_fft / 2 <= L + n_fft / 2, "stft_frames: frames exceed the padded signal");
  M2H_LAUNCH(stft_frames_kernel, dim3(sgrid((size_t)S * T * ldf)), dim3(256), 0, as_stream(stream), y, window, frames, S, L, T, n_fft, hop, ldf);
  return launch_status("stft_frames");
}

int m2h_stft_post(const float* spec, float* mag_out, float* phase_out, int B, int C, int T, int nb, int lds, int mode, m2h_stream stream) {
  M2H_REQUIRE(spec && (mag_out || phase_out) && B > 0 && C > 0 && T > 0 && nb > 0 && lds >= 2 * nb && mode >= 0 && mode <= 2, "stft_post: bad arguments");
  M2H_LAUNCH(stft_post_kernel, dim3(sgrid((size_t)B * nb * T * C)), dim3(256), 0, as_stream(stream), spec, mag_out, phase_out, B, C, T, nb, lds, mode);
  return launch_status("stft_post");
}

int m2h_istft_pre(const float* mag, const float* phase, float* rows, int B, int C, int c, int T, int nb, int ldr, m2h_stream stream) {
  M2H_REQUIRE(mag && phase && rows && B > 0 && C > 0 && c >= 0 && c < C && T > 0 && nb > 0 && ldr >= 2 * nb, "istft_pre: bad arguments");
  M2H_LAUNCH(istft_pre_kernel, dim3(sgrid((size_t)B * T * nb)), dim3(256), 0, as_stream(stream), mag, phase, rows, B, C, c, T, nb, ldr);
  return launch_status("istft_pre");
}

int m2h_istft_ola(const float* frames, const float* window, float* y, int S, int T, int n_fft, int hop, int ldf, int length, m2h_stream stream) {
  M2H_REQUIRE(frames && window && y && S > 0 && T > 0 && n_fft > 1 && hop > 0 && ldf >= n_fft && length > 0, "istft_ola: bad arguments");
  M2H_LAUNCH(istft_ola_kernel, dim3(sgrid((size_t)S * length)), dim3(256), 0, as_stream(stream), frames, window, y, S, T, n_fft, hop, ldf, length);
  return launch_status("istft_ola");
}

int m2h_feeder_round_mix(const float* full, int ldfull, int start, float* conv_out, float* mix, int S, int L, int first, float mix_scale,
                          m2h_stream stream) {
  M2H_REQUIRE(full && mix && S > 0 && L > 0 && start >= 0 && start + L <= ldfull, "feeder_round_mix: bad arguments");
  M2H_LAUNCH(feeder_round_mix_kernel, dim3(sgrid((size_t)S * L)), dim3(256), 0, as_stream(stream), full, ldfull, start, conv_out, mix, S, L,
                     first, mix_scale);
  return launch_status("feeder_round_mix");
}

int m2h_rms_normalize(float* mag, int S, int n, float norm, m2h_stream stream) {
  M2H_REQUIRE(mag && S > 0 && n > 0, "rms_normalize: bad arguments");
  M2H_LAUNCH(rms_normalize_kernel, dim3(S), dim3(1024), 0, as_stream(stream), mag, n, norm);
  return launch_status("rms_normalize");
}

int m2h_bss_metrics(const float* ref, const float* est, const float* mix_l, const float* mix_r, float* out, int S, int L, m2h_stream stream) {
  M2H_REQUIRE(ref && est && mix_l && out && S > 0 && L > 1, "bss_metrics: bad arguments");
  M2H_LAUNCH(bss_metrics_kernel, dim3(S), dim3(1024), 0, as_stream(stream), ref, est, mix_l, mix_r, out, L);
  return launch_status("bss_metrics");
}

}  // extern "C"
