/*
 * m2h.h -- C-ABI of libm2h.so: the MI355X (gfx950) kernels behind Move2Hear's data-parallel hot path.
 *
 * The reference (SAGNIKMJR/move2hear-active-AV-separation) has no FFI layer: its hot path is stock
 * torch.nn ops.  Each entry point below therefore names the reference torch-op sequence it replaces
 * (file:line relative to the reference root).  A maintainer binds these with ctypes (see
 * INTEGRATION.md); the shipped binding is move2hear-active-av-separation_amd/m2h/_lib.py.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (the library
 *     never allocates, frees or retains device memory) unless marked "host".
 *   - every call only enqueues work on `stream` (a hipStream_t passed as void*); it never synchronises.
 *   - return value: 0 = OK; negative = argument/shape error, nothing was launched; positive = hipError_t.
 *     The message is available from m2h_last_error() (thread-local).  No exceptions cross the ABI.
 *   - activations between kernels are NHWC ("channels-last") fp32: [B][H][W][C], C contiguous.
 *   - spectrogram tensors at the module boundary keep the reference layout BHWC = [B][F][T][C].
 */
#ifndef M2H_H
#define M2H_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* m2h_stream; /* hipStream_t */

#define M2H_VERSION 100

/* output layouts of m2h_conv_igemm_f32 */
#define M2H_OUT_NHWC 0     /* dst[((b*Ho+oh)*Wo+ow)*ldc + n]                                   */
#define M2H_OUT_DESLICE 1  /* n = c*16+s  ->  dst[((b*16*Ho + s*Ho + oh)*Wo + ow)*(N/16) + c]   */

int m2h_version(void);
const char* m2h_last_error(void);

/*
 * K1/K2  --  separator input glue.
 *   masks == NULL : x = mix                                    (separator_cnn.py:82)
 *   masks != NULL : x = log1p(max(0, masks * (exp(mix) - 1)))  (separator_cnn.py:73-79)
 * then BHWC -> 16-way frequency slice, written NHWC:            (separator_cnn.py:85-90)
 *   out[b][h][t][c*16 + s] = x[b][s*(F/16) + h][t][c]
 * mix, masks: [B][F][T][C] fp32;  out: [B][F/16][T][16*C] fp32.  F % 16 == 0, (16*C) % 4 == 0.
 * The (target_class + 1) plane of separator_cnn.py:93-99 is not materialised: it enters the first
 * conv as a border-aware per-channel bias (m2h_unet_class_table + cls_* of m2h_conv_args).
 */
int m2h_sep_slice_input(const float* mix, const float* masks, float* out, int B, int F, int T, int C,
                        m2h_stream stream);

/*
 * Weight packing (run once per weight version, into caller-owned buffers).
 *
 * m2h_pack_conv_weight: torch Conv2d weight [Co][Ci][KH][KW] -> [Co][KH][KW][ci_used] (K contiguous,
 *   K = KH*KW*ci_used, channel fastest); input channels >= ci_used are dropped (the class plane).
 * m2h_pack_convT_weight: torch ConvTranspose2d(k=4,s=2,p=1) weight [Ci][Co][4][4] -> 4 sub-pixel phase
 *   matrices [ph*2+pw][Co][th][tw][Ci]: output pixel (2q+ph, 2r+pw) reads input (q + th*(2ph-1),
 *   r + tw*(2pw-1)) through kernel tap kh = (ph ? 2 : 1) + th*(ph ? -2 : 2), kw likewise.
 * m2h_unet_class_table: table[(ch*3+cw)][co] = sum over the taps of channel `plane` that fall inside the
 *   image for an output pixel of border class (ch, cw) (0 = first row/col, 1 = interior, 2 = last)
 *   of a 4x4 stride-2 pad-1 conv; w is the torch weight [Co][Ci][4][4].
 * m2h_fold_bn: eval-mode BatchNorm2d as y = x*scale + shift  (scale = gamma/sqrt(var+eps),
 *   shift = beta - mean*scale).
 */
int m2h_pack_conv_weight(const float* w, float* wp, int Co, int Ci, int KH, int KW, int ci_used, m2h_stream stream);
int m2h_pack_convT_weight(const float* w, float* wp, int Ci, int Co, m2h_stream stream);
int m2h_unet_class_table(const float* w, float* table, int Co, int Ci, int plane, m2h_stream stream);
int m2h_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                float* scale, float* shift, int C, m2h_stream stream);

/*
 * Implicit-GEMM convolution, fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain).
 *   D[m][n] = act( (sum_k A[m][k] * Wp[n][k]) * scale[n] + shift[n] + cls_val[b]*cls_table[cls(m)][n] )
 * m enumerates (b, q, r) over B x Hq x Wq; k enumerates (th, tw, ci) over taps x (C0 + C1) input
 * channels, the channels being the concatenation of two NHWC sources (the U-Net skip concat of
 * separator_cnn.py:160-161 is never materialised); A[m][k] = src[b][q*stride + offh + th*mulh]
 * [r*stride + offw + tw*mulw][ci], zero outside the image.  The output pixel is (q*os + ph, r*os + pw).
 * With conv_transpose != 0 the launch covers the 4 sub-pixel phases of a 4x4/s2/p1 transposed conv
 * (grid z = phase; weights from m2h_pack_convT_weight; mul/off/ph/pw derived per phase).
 */
typedef struct m2h_conv_args {
  const float* src0; /* NHWC [B][Hi][Wi][C0] */
  const float* src1; /* NHWC [B][Hi][Wi][C1] or NULL (C1 = 0) */
  int C0, C1;
  int B, Hi, Wi;
  int Hq, Wq;  /* GEMM pixel grid: M = B*Hq*Wq */
  int stride;  /* input step per q/r */
  int nth, ntw; /* taps along h, w */
  int mulh, offh, mulw, offw;
  int conv_transpose; /* 0 | 1 */
  const float* wp;    /* packed weights [phase][N][K], K = nth*ntw*(C0+C1) */
  int N;
  const float* scale; /* [N] or NULL (=1) */
  const float* shift; /* [N] or NULL (=0) */
  float slope;        /* y = v > 0 ? v : v*slope  (0 = ReLU, 0.2 = LeakyReLU, 1 = identity) */
  const float* cls_table; /* [9][N] or NULL */
  const float* cls_val;   /* [B] (target_class + 1 as float) or NULL */
  float* dst;
  int Ho, Wo; /* output spatial size */
  int os;     /* output step per q/r (1 conv, 2 transposed conv) */
  int ph, pw; /* output phase offset when conv_transpose == 0 */
  int ldc;    /* channel count of dst rows (M2H_OUT_NHWC) */
  int out_mode;
  void* workspace;        /* optional split-K scratch (device); NULL = never split */
  size_t workspace_bytes; /* size of workspace; m2h_conv_igemm_workspace_bytes() says how much the launch can use */
  const float* head_w;    /* optional fused 1x1 head (N in {16,32}, M2H_OUT_DESLICE, workspace NULL): [N][N], applied after the */
  const float* head_b;    /*   activation; out = head_w . act(...) + head_b, stored de-sliced.  NULL = no head.                */
  int operand_format;     /* bit set of M2H_FMT_*: the arithmetic of THIS call (M2H_FMT_MATH_*; neither bit = the calling       */
                          /*   thread's mode, m2h_set_math_mode) and, in bf16x3 math, which operands already are in the split32     */
                          /*   layout (below) and whether dst is to be written in it.  0 = plain fp32 tensors, thread's arithmetic. */
} m2h_conv_args;

/* split32 layout (internal operand format of the bf16x3 math mode): every aligned group of 32 consecutive fp32 values of the
 * innermost (channel / k) dimension is replaced, in place of its 128 bytes, by 32 bf16 high parts followed by 32 bf16 low
 * parts (x = hi + lo, hi = bf16(x), lo = bf16(x - hi), round to nearest even).  Same footprint, strides and alignment as the
 * fp32 tensor; the channel count must be a multiple of 32.  m2h_split32 converts; the conv engine reads it with
 * M2H_FMT_SRC_SPLIT / M2H_FMT_W_SPLIT and writes it with M2H_FMT_DST_SPLIT (NHWC output, N % 32 == 0). */
#define M2H_FMT_SRC_SPLIT 1
#define M2H_FMT_W_SPLIT 2
#define M2H_FMT_DST_SPLIT 4
#define M2H_FMT_MATH_BF16X3 8 /* this call computes in bf16x3 split products, whatever the calling thread's mode */
#define M2H_FMT_MATH_FP32 16  /* this call computes in fp32 MFMA, whatever the calling thread's mode */
int m2h_split32(const float* src, float* dst, size_t count /* floats, multiple of 32 */, m2h_stream stream);

int m2h_conv_igemm_f32(const m2h_conv_args* args /* host */, m2h_stream stream);

/* Arithmetic of the igemm forward engine for launches made BY THE CALLING THREAD through entry points that carry no
 * m2h_conv_args / m2h_unet_weights of their own (m2h_unet_down_fwd, m2h_unet_up_fwd, ...): M2H_MATH_FP32 (default: fp32 MFMA,
 * exact fp32 products) or M2H_MATH_BF16X3 (fp32 operands split into bf16 hi + lo, products hi*hi + hi*lo + lo*hi on the bf16
 * matrix pipe, fp32 accumulate; shapes the scalar loader takes).  The value is thread-local: the library holds no
 * process-global arithmetic state, two host threads may run different modes side by side.  Calls that do carry a struct
 * may pin their arithmetic there (M2H_FMT_MATH_*, m2h_unet_weights.math_mode) and then ignore this. */
#define M2H_MATH_FP32 0
#define M2H_MATH_BF16X3 1
#define M2H_MATH_BF16 2 /* BASELINE config 2's literal dtype, a REPORTED mode (bench.py other_math_modes), never the default: every dispatch
                          decision as M2H_MATH_BF16X3, but the split32 engines of the benchmark batch (csrc/conv_patch.hip, conv_dma.hip,
                          conv_strip.hip) multiply the bf16 hi halves only -- one MFMA product per product instead of three; operands,
                          accumulation and outputs unchanged (fp32 sums, split32 tensors).  Other engines compute bf16x3. */
int m2h_set_math_mode(int mode);
/* AcousticMem's last conv and update_sep's loss in ONE launch (rl/models/memory_nets.py:16,62-67; rl/ppo/ppo.py:206-216):
 * loss = F.l1_loss(deslice(conv3x3(h, w)), gt) with h NHWC [B][H][T][C], wp the packed 3x3 weight [16][9 C] (m2h_pack_conv_weight_ex) and gt_plane
 * the target as a contiguous plane [B][16 H][T] (gt_mono_comps[..., 0]); the conv's output is never stored: the image-row kernel's epilogue adds
 * |y - g| to its block's partial sum and writes d loss / d y = sign(y - g) / n to dy NHWC [B][H][T][16] -- the layout the conv's weight- and
 * input-gradient launches read.  partials: >= 1024 floats of scratch; loss: 1 float (sum of the partials in block order: bit-reproducible).
 * Arithmetic: the calling thread's (m2h_set_math_mode).  Shapes: H = T = C = 32, B >= 64 (m2h_conv3x3_l1_nhwc16_supported; other shapes:
 * m2h_conv_igemm_f32 + m2h_l1_loss_nhwc16, which this replaces where it applies: 110 MB written and read back per update_sep epoch, one launch). */
int m2h_conv3x3_l1_nhwc16_supported(int B, int H, int T, int C);
int m2h_conv3x3_l1_nhwc16(const float* h, const float* wp, const float* gt_plane, float* dy, float* loss, float* partials, int B, int H, int T, int C,
                          m2h_stream stream);

/* Diagnostic: kernels this process has enqueued (or captured into a HIP graph) through libm2h so far.  bench.py's per-phase launch
 * counts (a replayed graph counts its captured kernels once per replay, on the host side: m2h/graphs.py). */
long long m2h_launch_count(void);
int m2h_get_math_mode(void);

/* Bytes of split-K scratch the launch described by args would use (0 = the grid already fills the chip).
 * Small-M layers (deep U-Net stages, rollout batches) are split along K over up to 32 blocks; partial sums go
 * to the workspace as [phase][split][M][N] fp32 and a second kernel reduces them in a fixed order
 * (bit-reproducible) and applies the fused epilogue. */
size_t m2h_conv_igemm_workspace_bytes(const m2h_conv_args* args /* host */);


/*
 * Named fused ops of the separator U-Nets; thin argument adapters over m2h_conv_igemm_f32.
 *
 * K3  m2h_unet_down_fwd: Conv2d(4x4, s2, p1, no bias) + BatchNorm2d(eval) + LeakyReLU(0.2)
 *     (separator_cnn.py:5-12,101-105).  x NHWC [B][H][W][Ci] -> y NHWC [B][H/2][W/2][Co].
 *     cls_table/cls_val non-NULL only for binSep stage 0 (the target-class plane).
 *     workspace: optional split-K scratch (see m2h_conv_igemm_workspace_bytes), NULL allowed.
 * K4  m2h_unet_up_fwd: cat(x, skip) + ConvTranspose2d(4x4, s2, p1, no bias) + BatchNorm2d(eval) + ReLU
 *     (separator_cnn.py:15-24,156-161).  x [B][H][W][C0], skip [B][H][W][C1] or NULL -> y [B][2H][2W][Co].
 * K5  m2h_unet_head_fwd: Conv2d(1x1, bias) + de-slice + permute to BHWC (separator_cnn.py:134,163-168).
 *     x NHWC [B][H][W][Ci] -> out BHWC [B][16*H][W][Co/16].
 */
int m2h_unet_down_fwd(const float* x, const float* wp, const float* scale, const float* shift,
                      const float* cls_table, const float* cls_val, float* y,
                      int B, int H, int W, int Ci, int Co, void* workspace, size_t workspace_bytes, m2h_stream stream);
int m2h_unet_up_fwd(const float* x, const float* skip, const float* wp, const float* scale, const float* shift,
                    float* y, int B, int H, int W, int C0, int C1, int Co, void* workspace, size_t workspace_bytes,
                    m2h_stream stream);
/* K4+K5 fused: the last decoder stage (Co in {16,32}) followed by the 1x1 head, de-sliced straight into BHWC
 * out [B][16*2H][2W][Co/16]: the stage's activation never leaves the chip (separator_cnn.py:133-134,163-168). */
int m2h_unet_up_head_fwd(const float* x, const float* skip, const float* wp, const float* scale, const float* shift,
                         const float* head_w, const float* head_b, float* out, int B, int H, int W, int C0, int C1, int Co,
                         m2h_stream stream);
/* split-K scratch the two ops above can use for these shapes (0 = none needed) */
size_t m2h_unet_down_workspace_bytes(int B, int H, int W, int Ci, int Co);
size_t m2h_unet_up_workspace_bytes(int B, int H, int W, int C0, int C1, int Co);
int m2h_unet_head_fwd(const float* x, const float* wp, const float* bias, float* out,
                      int B, int H, int W, int Ci, int Co, m2h_stream stream);

/* ------------------------------------------------------------------------------------------------------------------
 * RL path (forward): layout glue, GRU cell, heads, scans and reductions.  All fp32, all HBM-bound or tiny.
 * ------------------------------------------------------------------------------------------------------------------ */

/* m2h_pack_conv_weight_ex: as m2h_pack_conv_weight but the packed channel count ci_out >= ci_used is zero-padded
 * (VisualCNN's 3-channel first conv is packed to 4 channels, visual_cnn.py:65-72).  A Linear weight [Co][C*H*W] applied
 * to an NCHW-flattened map (visual_cnn.py:140-141, audio_cnn.py:131-132) is packed as the conv weight [Co][C][H][W]. */
int m2h_pack_conv_weight_ex(const float* w, float* wp, int Co, int Ci, int KH, int KW, int ci_used, int ci_out, m2h_stream stream);

/* Input glue of AcousticMem (memory_nets.py:40-59: cat(slice(pred_mono), slice(prev_mem * mask))) and AudioCNN
 * (audio_cnn.py:117-133).  Virtual BHWC tensor with channels [a (Ca) | b (Cb)], b optionally scaled per batch row by
 * bscale[B] (the not-done mask of ppo_trainer.py:310-314), then
 *   op 0: x          op 1: log1p(max(0, mul * (exp(a) - 1)))  (mul: same shape as a)      op 2: log1p(max(0, x))
 * and the 16-way frequency slice into NHWC  out[b][h][t][c*16+s].  a,b: [B][F][T][Ca|Cb];  out: [B][F/16][T][16*(Ca+Cb)]. */
int m2h_slice_concat_input(const float* a, int Ca, const float* b, int Cb, const float* mul, const float* bscale, int op,
                           float* out, int B, int F, int T, m2h_stream stream);

/* AcousticMem forward (memory_nets.py:40-69, the DD-PPO variant without BatchNorm) for small batches in ONE launch: both inputs
 * sliced 16-way and concatenated, the previous memory scaled by not_done[b] (ppo_trainer.py:310-314; NULL = 1), Conv2d(32, 32, 3, 1, 1)
 * + ReLU, Conv2d(32, 16, 3, 1, 1), de-slice.  pred_mono, prev_mem, out: BHWC [B][512][32][1]; w0p [32][9*32], w1p [16][9*32]: the
 * two conv weights packed by m2h_pack_conv_weight.  fp32 MFMA; the rollout step's call (14 environments). */
int m2h_acoustic_mem_small_fwd(const float* pred_mono, const float* prev_mem, const float* not_done, const float* w0p, const float* w1p,
                               float* out, int B, int F, int T, m2h_stream stream);

/* VisualCNN input (visual_cnn.py:135-150): rgb [B][H][W][3] in 0..255 -> out [B][H][W][4] = (rgb/255, depth or 0). */
int m2h_visual_input(const float* rgb, const float* depth, float* out, int B, int H, int W, m2h_stream stream);

/* GRU cell pointwise part (torch.nn.GRU as used by rnn_state_encoder.py:74-84), gate order r,z,n:
 *   gi = x W_ih^T + b_ih  [M][3H] (from m2h_conv_igemm_f32),  gh_raw = h W_hh^T (no bias, UNMASKED h) [M][3H],
 *   mask[M] in {0,1} or NULL: the hidden-state reset h*mask is applied here ((h*m) W^T == m * (h W^T)).
 *   hout = (1-z)*n + z*(m*h). */
int m2h_gru_gates(const float* gi, const float* gh_raw, const float* bhh, const float* hprev, const float* mask, float* hout,
                  int M, int H, m2h_stream stream);

/* One whole GRU time step for M <= 16 rows (rnn_state_encoder.py:74-84 single_forward, and each step of seq_forward :86-137 at
 * the rollout width): gh_raw = hprev W_hh^T ([M][3H], written because the backward pass reads it) and the gate math of
 * m2h_gru_gates in one launch.  whh: [3H][H] (torch weight_hh_l0).  H % 16 == 0. */
int m2h_gru_step(const float* gi, const float* whh, const float* bhh, const float* hprev, const float* mask, float* gh_raw, float* hout,
                 int M, int H, m2h_stream stream);

/* The whole cell of a NO-GRAD single step for M <= 16 rows in one launch (the rollout step, ppo_trainer.py:322-335): the input
 * projection gi = x W_ih^T + b_ih, the recurrent product and the gates of m2h_gru_gates; only hout is written.  x [M][I],
 * wih [3H][I], whh [3H][H] (torch weight_ih_l0 / weight_hh_l0), hprev [M][H], mask [M] or NULL.  I % 16 == 0, H % 16 == 0. */
int m2h_gru_cell(const float* x, const float* wih, const float* bih, const float* whh, const float* bhh, const float* hprev,
                 const float* mask, float* hout, int M, int I, int H, m2h_stream stream);

/* CategoricalNet + CriticHead (common/utils.py:16-50, rl/ppo/policy.py:15-23): logits = feats Wa^T + ba (A <= 8),
 * value = feats Wc^T + bc, logp_all = log_softmax, probs = softmax, entropy = -sum p*logp per row; when actions != NULL
 * also logp_act[row] = logp_all[row][actions[row]] (CustomFixedCategorical.log_probs).  actions: int64. */
int m2h_policy_heads(const float* feats, const float* Wa, const float* ba, const float* Wc, const float* bc,
                     const long long* actions, float* value, float* logp_all, float* probs, float* entropy, float* logp_act,
                     int M, int H, int A, m2h_stream stream);
int m2h_gather_logp(const float* logp_all, const long long* actions, float* out, int M, int A, m2h_stream stream);
/* LSTM cell, pointwise part -- RNNStateEncoder's rnn_type = "LSTM" variant (rl/models/rnn_state_encoder.py:10-34,49-69; nn.LSTM gate order i, f, g, o):
 * pre = gi + gh [M][4H] (the two products with their biases); c' = sigmoid(f) (c_prev mask) + sigmoid(i) tanh(g); h' = sigmoid(o) tanh(c').
 * mask [M]: the reset mask of the carried cell state (:63-69).  gates_out: NULL or [M][4H], the activated gates the backward reads.
 * m2h_lstm_cell_bwd: dh / dc (either may be NULL = zero) -> dpre [M][4H] (gradient of gi and of gh) and dc_prev [M][H]. */
int m2h_lstm_cell(const float* gi, const float* gh, const float* c_prev, const float* mask, float* h_out, float* c_out, float* gates_out, int M, int H,
                  m2h_stream stream);
int m2h_lstm_cell_bwd(const float* dh, const float* dc, const float* gates, const float* c_prev, const float* mask, const float* c, float* dpre,
                      float* dc_prev, int M, int H, m2h_stream stream);
/* Policy.act's tail in ONE launch (rl/ppo/policy.py:217-225): m2h_policy_heads, then the action -- noise != NULL: the single draw of
 * torch.multinomial(probs, 1, True) == argmax(probs / noise) with caller-supplied Exp(1) noise [M][A] (as m2h_sample_actions);
 * noise == NULL: CustomFixedCategorical.mode() == argmax(probs) -- and logp_act[row] = logp_all[row][action]. */
int m2h_policy_heads_act(const float* feats, const float* Wa, const float* ba, const float* Wc, const float* bc, const float* noise,
                         float* value, float* logp_all, float* probs, float* entropy, long long* actions, float* logp_act, int M, int H,
                         int A, m2h_stream stream);
/* Same with the draw's Exp(1) noise made INSIDE the kernel ("fused" sampling: no generator launch in the rollout step): element (row, a)
 * takes -log(u), u from Philox4x32-10 keyed by rng_state[0] (seed) at counter rng_state[1] + row A + a; rng_state: two uint64 on the
 * device, the counter is advanced by the caller (m2h_step_index_advance_rng inside a replayed step).  u = (23 random bits + 0.5) / 2^23
 * (exact in fp32, never 0 or 1).  noise_out: NULL, or [M][A] floats that receive the noise drawn -- the record a parity test hands the CPU
 * oracle in place of its generator's draw (common/utils.py:16-24: "same probs + same noise => same actions"). */
int m2h_policy_heads_act_rng(const float* feats, const float* Wa, const float* ba, const float* Wc, const float* bc,
                             const unsigned long long* rng_state, float* value, float* logp_all, float* probs, float* entropy,
                             long long* actions, float* logp_act, float* noise_out, int M, int H, int A, m2h_stream stream);

/* CustomFixedCategorical.sample (common/utils.py:16-24) with the noise supplied by the caller: the single-draw path of
 * torch.multinomial(probs, 1, True) is argmax(probs / q), q ~ Exp(1) drawn from the tensor's generator -- on the reference's
 * CPU path the default mt19937 generator.  The host draws q there (same call, same stream position), ships it to the device,
 * and this kernel does the rest: actions[row] = first index of max_j probs[row][j] / noise[row][j] (IEEE fp32 division, ties to
 * the lowest index as ATen's CPU argmax).  probs, noise [M][A] fp32, actions [M] int64.  A <= 64. */
int m2h_sample_actions(const float* probs, const float* noise, long long* actions, int M, int A, m2h_stream stream);

/* RolloutStoragePol.compute_returns (common/rollout_storage.py:155-180).  rewards [T][N], value_preds [T+1][N] (row T is
 * overwritten by next_value when use_gae), masks [T+1][N], next_value [N], returns [T+1][N]. */
int m2h_gae_returns(const float* rewards, float* value_preds, const float* masks, const float* next_value, float* returns, int T,
                    int N, int use_gae, float gamma, float tau, m2h_stream stream);

/* PPO.get_advantages (ppo.py:75-80): adv = returns - value_preds over n = T*N elements; mode 0 raw, mode 1 local
 * normalisation (adv-mean)/(std_unbiased+eps); mode 2 raw + stats[0] = local mean (first step of the distributed variant,
 * ppo.py:275-284 + ddppo_utils.py:168-190: all-reduce the mean, m2h_adv_sqdiff, all-reduce, m2h_adv_apply). */
int m2h_advantages(const float* returns, const float* value_preds, float* adv, float* stats, int n, int mode, float eps,
                   m2h_stream stream);
int m2h_adv_sqdiff(const float* adv, const float* gmean, float* out, int n, m2h_stream stream);
int m2h_adv_apply(float* adv, const float* gmean, const float* gvar, int n, float eps, m2h_stream stream);

/* PPO losses (ppo.py:125-157) forward and analytic gradients: out[4] = (value_loss, action_loss, mean entropy, total_loss =
 * value_loss*value_loss_coef + action_loss - entropy*entropy_coef); entropy[n] per-row entropies or NULL;
 * g_values = d(total)/dvalues, g_logp = d(total)/d(action_log_probs) (either may be NULL); d(total)/d(entropy_row) is the
 * constant -entropy_coef/n.  clip_dev: NULL, or a device scalar that replaces `clip` and is read when the kernel runs (a
 * captured HIP graph of the update then follows the linear clip decay of ppo_trainer.py:736-739). */
int m2h_ppo_loss(const float* values, const float* logp, const float* old_values, const float* returns, const float* adv,
                 const float* old_logp, const float* entropy, float clip, const float* clip_dev, int use_clipped_value_loss,
                 float value_loss_coef, float entropy_coef, float* out, float* g_values, float* g_logp, int n, m2h_stream stream);

/* reward_util / override_rewards (common/env_utils.py:690-713).  m2h_sq_stats: per env e, stats[e] = (sum (pred-gt)^2,
 * sum gt^2) over L elements, gt read with stride/offset from an interleaved components tensor (gt_mono_comps[...,0]).
 * m2h_rewards_from_stats: done -> 0; else -(mse/mean gt^2) of the next step, minus the same at the current step
 * (quality_improvement) or times mult. */
int m2h_sq_stats(const float* pred, const float* gt_comps, int gt_stride, int gt_off, float* stats, int N, int L, m2h_stream stream);
int m2h_rewards_from_stats(const float* next_stats, const float* cur_stats, const float* not_done, float* rewards, int N, int L,
                           int quality_improvement, float mult, m2h_stream stream);

/* Minibatch gather of the recurrent generators (common/rollout_storage.py:182-298, 392-457): for a storage tensor
 * src [T][N][row] and an env permutation perm[Nsel] (int64, device), dst[t][j][:] = src[t][perm[j]][:]; row_bytes % 4 == 0.
 * The result viewed as [T*Nsel][row] is the reference's stacked + flattened minibatch. */
int m2h_gather_envs(const void* src, const long long* perm, void* dst, int T, int N, int Nsel, size_t row_bytes, m2h_stream stream);

/* Per-episode statistics kept by the rollout step (ppo_trainer.py:407-478): after every env step, for each of N envs the
 * running sums of the current episode take the step's reward / action probabilities / STFT-L2 losses, finished episodes
 * (not_done == 0) are folded into the totals (rewards, steps, counts, per-step means, last-step values) and their running
 * sums reset.  All tensors are [N] floats (dist_probs: [N][A]) owned by the caller. */
typedef struct m2h_episode_stats {
  float* episode_rewards;
  float* episode_counts;
  float* episode_steps;
  float* episode_dist_probs;
  float* episode_bin_losses_allSteps;
  float* episode_mono_losses_lastStep;
  float* episode_mono_losses_allSteps;
  float* episode_monoFromMem_losses_lastStep;
  float* episode_monoFromMem_losses_allSteps;
  float* current_episode_reward;
  float* current_episode_step;
  float* current_episode_dist_probs;
  float* current_episode_bin_losses;
  float* current_episode_mono_losses;
  float* current_episode_monoFromMem_losses;
  float* episode_ndgs; /* the env's two distance infos of the step that ends an episode (ppo_trainer.py:332-337, :434-435) */
  float* episode_dgs;
} m2h_episode_stats;
/* ndgs / dgs: [N] normalised / absolute geodesic distance to the target reported by the env for this step; NULL = zeros. */
int m2h_episode_stats_update(const m2h_episode_stats* st, const float* rewards, const float* dist_probs, const float* bin_losses,
                             const float* mono_losses, const float* monoFromMem_losses, const float* not_done, const float* ndgs,
                             const float* dgs, int N, int A, m2h_stream stream);

/* Full linear convolution of the audio feeder: for each of CS (clip, source) pairs, mono [CS][L] convolved with the two ears of
 * rirs [CS][Lr][2] -> full [CS][2][N], N = 2^log2n >= L + Lr - 1 (scipy.signal.fftconvolve in pretrain/datasets/dataset.py:180,
 * habitat_audio/simulator_train.py:419; the "same" window and the int16 round trip follow in m2h_feeder_round_mix).  Hand-written
 * radix-2 FFTs in LDS, one workgroup per pair (N <= 2^15: the packed N/2-point complex transform is 128 KB).
 * twiddles: [N/2] complex fp32 exp(-2 pi i k / N) (interleaved re, im; caller-built); xspec: [CS][2][N/2] complex fp32 scratch (2 N floats per pair). */
int m2h_fftconv_full(const float* mono, const float* rirs, const float* twiddles, float* xspec, float* full, int CS, int L, int Lr,
                     int log2n, m2h_stream stream);

/* Batched weight packing: up to M2H_PACK_BATCH_MAX tensors (re)packed by ONE launch; same index maps and results as the
 * single-tensor entry points.  kind / p[]:
 *   M2H_PACK_CONV      m2h_pack_conv_weight_ex:  p = {Co, Ci, KH, KW, ci_used, ci_out}     [Co][Ci][KH][KW] -> [Co][KH][KW][ci_out]
 *   M2H_PACK_CONVT     m2h_pack_convT_weight:    p = {Ci, Co, -, -, -, -}                  [Ci][Co][4][4]   -> [4 phases][Co][2][2][Ci]
 *   M2H_PACK_DGRAD     m2h_pack_dgrad_weight:    p = {Co, Ci, KH, KW, stride, pad}         -> [stride^2 phases][Ci][KH/s][KW/s][Co]
 *   M2H_PACK_FC_DGRAD  input gradient of a full-spatial conv (a Linear over the flattened map, visual_cnn.py:140-141):
 *                                                p = {Co, Ci, KH, KW, ci_used, ci_out}     -> [KH][KW][ci_out][Co] */
#define M2H_PACK_BATCH_MAX 48
#define M2H_PACK_CONV 0
#define M2H_PACK_CONVT 1
#define M2H_PACK_DGRAD 2
#define M2H_PACK_FC_DGRAD 3
typedef struct m2h_pack_item {
  const float* src;
  float* dst;
  int kind;
  int p[6];
} m2h_pack_item;
int m2h_pack_batch(const m2h_pack_item* items /* host */, int n_items, m2h_stream stream);

/* The whole per-env bookkeeping of one rollout step (ppo_trainer.py:375-455) in one launch: reward (override_rewards /
 * reward_util, env_utils.py:690-713; with extra_reward the step carries extra_mult x util(next), the effective value of
 * ppo_trainer.py:395-405), the three STFT-L2 distances of eval_metrics.py:306-366 (binaural masks on exp(mix)-1, mono, mono from
 * memory; GT phase on both sides) and the per-episode statistics update of m2h_episode_stats_update.  All spectrogram tensors are
 * [N][L][C] with L = F*T bins: next_mem / mem / mono C = 1, masks / mix C = 2, gt_mono_comps C = 4 (magnitude at 0),
 * gt_bin_comps C = 8 (magnitudes at 0 and 2).  rewards [N], losses [3][N] (bin, mono, mono-from-memory) are written; stats is
 * updated in place.  partial: m2h_step_stats_workspace_bytes(N) bytes of scratch; tickets: N uint32 counters, zero before the
 * first call (the kernel leaves them zero).  override_rewards = 0: rewards = env_rewards (farTarget.yaml: no override). */
#define M2H_STEP_STATS_CHUNKS 16
typedef struct m2h_step_stats_args {
  const float* next_mem;            /* pred_monoFromMem of the NEXT observation (override only) */
  const float* next_gt_mono_comps;  /* (override only) */
  const float* mem;                 /* pred_monoFromMem of the current observation */
  const float* gt_mono_comps;
  const float* masks;               /* pred_binSepMasks */
  const float* mix;                 /* mixed_bin_audio_mag */
  const float* gt_bin_comps;
  const float* mono;                /* pred_mono */
  const float* not_done;            /* [N] */
  const float* env_rewards;         /* [N], used when override_rewards == 0 */
  const float* probs;               /* [N][A] action probabilities of the step */
  const float* ndgs;                /* [N] or NULL */
  const float* dgs;                 /* [N] or NULL */
  float* rewards;                   /* [N] out */
  float* losses;                    /* [3][N] out */
  m2h_episode_stats stats;
  float* partial;
  unsigned* tickets;
  int N, L, A;
  int override_rewards, extra_reward;
  float extra_mult;
} m2h_step_stats_args;
size_t m2h_step_stats_workspace_bytes(int N);
int m2h_rollout_step_stats(const m2h_step_stats_args* args /* host */, m2h_stream stream);

/* Batched row copies with DEVICE-resident row indices: RolloutStoragePol.insert / RolloutStorageSep.insert and the rollout
 * step's reads of row `step` (common/rollout_storage.py:68-96, 372-390; ppo_trainer.py:262-300) when the step is replayed from
 * a HIP graph and the host step counter cannot be baked into addresses.  Item i copies `bytes` bytes (multiple of 4) from
 * src + idx[src_slot]*bytes to dst + idx[dst_slot]*bytes; a negative slot means no offset.  idx: int64 device array. */
#define M2H_ROWS_COPY_MAX 32
typedef struct m2h_row_copy {
  const void* src;
  void* dst;
  size_t bytes;
  int src_slot;
  int dst_slot;
} m2h_row_copy;
int m2h_rows_copy(const m2h_row_copy* items, int n_items, const long long* idx, m2h_stream stream);

/* idx = (pol_step, pol_step + 1, sep_step + 1), the device-resident step counters m2h_rows_copy addresses rows with, advanced as
 * RolloutStoragePol/Sep.insert advance theirs: step = (step + 1) % num_steps (common/rollout_storage.py:96,390). */
int m2h_step_index_advance(long long* idx, int T_pol, int T_sep, m2h_stream stream);
/* Same, and the fused sampler's counter (rng_state[1], m2h_policy_heads_act_rng) moves past the step's rng_inc draws. */
int m2h_step_index_advance_rng(long long* idx, int T_pol, int T_sep, unsigned long long* rng_state, unsigned long long rng_inc, m2h_stream stream);

/* Synthetic on-device vector env (m2h/envs/synthetic_env.py; stands where the simulator's pose update and sensor suite stand,
 * habitat_audio/simulator_train.py:216-227,386-486).  m2h_synth_env_step: per env, action 0 moves to node + 1, 1 / 2 turn by
 * +1 / +3 quarter turns (all modulo).  m2h_synth_env_observe: per item, row e of dst = row r(e) of src, r = node*4 + angle
 * (items[i].src_slot == 0: cached frames) or audio_idx (src_slot == 1: spectrogram pool); items[i].bytes = one row. */
int m2h_synth_env_step(const long long* actions, long long* node, long long* angle, int n_nodes, int N, m2h_stream stream);
int m2h_synth_env_observe(const m2h_row_copy* items, int n_items, const long long* node, const long long* angle, const long long* audio_idx,
                          int N, m2h_stream stream);

/* STFT_L2_distance (common/eval_metrics.py:306-366) for nch channels: out[e] = sum_ch mean_{re/im,F,T} of the squared
 * distance between (gt_mag, pred_mag) x (cos, sin)(gt_phase); pred_mag = pred (use_mix 0) or (exp(mix)-1)*pred (use_mix 1).
 * pred/mix: [N][L][Cp]; gt_comps: [N][L][Cg] = per channel (mag, phase). */
int m2h_stft_l2(const float* mix, const float* pred, int Cp, const float* gt_comps, int Cg, int nch, int use_mix, float* out, int N,
                int L, m2h_stream stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Backward of the implicit-GEMM convolution (training rows: ppo.py:159-175 update_pol, :224-241 update_sep).
 * ------------------------------------------------------------------------------------------------------------------ */

/* Weight gradient in the packed layout: dw[n][k] = sum_m dy[m][n] * A[m][k], with A gathered from args->src0/src1 exactly as
 * the forward launch described by args does (same geometry fields; wp/scale/shift/dst are ignored; conv_transpose must be
 * 0).  dy: NHWC output gradient [B][Ho][Wo][ldy] (activation already back-propagated); row m is read at output pixel
 * (q*os+ph, r*os+pw), i.e. at row m itself for an ordinary conv.  args->workspace must
 * hold m2h_conv_wgrad_workspace_bytes(args) bytes (pixel range is split over blocks, partial tiles are summed in a fixed
 * order: bit-reproducible). */
size_t m2h_conv_wgrad_workspace_bytes(const m2h_conv_args* args /* host */);
int m2h_conv_wgrad_f32(const m2h_conv_args* args /* host */, const float* dy, int ldy, float* dw, m2h_stream stream);
/* The same with the backward of the layer's fused activation folded into the load of dy: dy is taken as dy * (y > 0 ? 1 : slope),
 * y = the layer's forward output (same layout as dy) -- m2h_act_bwd followed by m2h_conv_wgrad_f32 without the 3-tensor pass in
 * between.  Built into the image-row kernel only (3x3 / stride 1 / pad 1 over 32-channel, 32-pixel-wide images: AcousticMem's first
 * conv, memory_nets.py:11-16, at update_sep's 1.7 M pixels); other shapes are refused (rc < 0). */
int m2h_conv_wgrad_gated_f32(const m2h_conv_args* args /* host */, const float* dy, int ldy, const float* y, float slope, float* dw,
                             m2h_stream stream);
/* The same gradient delivered in nn.Conv2d's own layout dw[N][Ci][KH][KW] (Ci <= C0 + C1: packed channels from Ci on are input
 * padding and are dropped) -- the split sum and the re-layout are one launch, where m2h_conv_wgrad_f32 + a permute copy were two
 * (one launch less per conv layer of every backward pass; same bits).  y: optional gate as in m2h_conv_wgrad_gated_f32 (NULL: none). */
int m2h_conv_wgrad_torch_f32(const m2h_conv_args* args /* host */, const float* dy, int ldy, const float* y, float slope, float* dw, int Ci,
                             m2h_stream stream);
/* m2h_conv_wgrad_torch_f32 of AcousticMem's FIRST conv with the input gradient of its SECOND conv fused in (update_sep's backward,
 * rl/ppo/ppo.py:226 through rl/models/memory_nets.py:11-16): the gradient the weight gradient contracts with, d loss / d h (h = ReLU(conv0(x)) = y),
 * is conv1's input gradient -- a 3x3 convolution of dy2 = d loss / d conv1-output, NHWC [B][H][W][16], with conv1's packed weight
 * w2_packed [16][9 * 32] -- and is made row by row inside the image-row kernel instead of being written by one launch (220 MB at 1680 samples)
 * and read back by the next.  y / slope: the ReLU gate as in m2h_conv_wgrad_gated_f32 (required).  bf16x3 arithmetic of the calling thread, args =
 * the first conv's geometry (3x3 / 1 / 1, 32 -> 32 channels, 32-pixel rows): m2h_conv_wgrad_dgrad_fused_supported(args) tells. */
int m2h_conv_wgrad_dgrad_fused_supported(const m2h_conv_args* args /* host */);
int m2h_conv_wgrad_dgrad_fused_f32(const m2h_conv_args* args /* host */, const float* dy2, const float* w2_packed, const float* y, float slope, float* dw,
                                   int Ci, m2h_stream stream);

/* Input gradient = forward engine on re-laid-out weights: for a Conv2d(k, stride s, pad p) weight w [Co][Ci][KH][KW]
 * (KH, KW multiples of s) writes s*s phase matrices wp[ph*s+pw][ci][th][tw][co] = w[co][ci][(ph+p)%s + s*th][(pw+p)%s + s*tw].
 * Phase (ph,pw) of dx is then m2h_conv_igemm_f32 with src0 = dy (C0 = Co), N = Ci, taps (KH/s, KW/s), mulh = mulw = -1,
 * offh = (ph + p - (ph+p)%s)/s (likewise offw), stride 1, os = s, output phase (ph,pw), Hq = ceil((H_in - ph)/s). */
int m2h_pack_dgrad_weight(const float* w, float* wp, int Co, int Ci, int KH, int KW, int stride, int pad, m2h_stream stream);

/* out = dy * (y > 0 ? 1 : slope): backward of the fused ReLU (slope 0) / LeakyReLU epilogue, y = the forward output. */
int m2h_act_bwd(const float* dy, const float* y, float slope, float* out, size_t n, m2h_stream stream);

/* db[n] = sum_m dy[m][n]  (Conv2d / Linear bias gradient); two ordered reduction stages (deterministic).
 * workspace: m2h_bias_grad_workspace_bytes(M, N) bytes of device scratch. */
size_t m2h_bias_grad_workspace_bytes(int M, int N);
int m2h_bias_grad(const float* dy, float* db, int M, int N, float* workspace, m2h_stream stream);
/* m2h_act_bwd followed by m2h_bias_grad in one pass over dy: out[m][n] = y[m][n] > 0 ? dy[m][n] : dy[m][n] * slope, db[n] = sum_m out[m][n]
 * (the same partition and summation order as m2h_bias_grad: the same bits).  The backward of a conv / Linear layer with bias and a fused
 * ReLU / LeakyReLU (ppo.py:159-161).  workspace: m2h_bias_grad_workspace_bytes(M, N). */
int m2h_act_bwd_bias(const float* dy, const float* y, float slope, float* out, float* db, int M, int N, float* workspace, m2h_stream stream);

/* GRU backward, one time step (torch.nn.GRU semantics, rnn_state_encoder.py:86-137 under autograd): inputs of
 * m2h_gru_gates plus dh = dL/dh_out; outputs dgi = dL/d(gi), dpre = dL/d(mask*gh_raw + b_hh) (so dL/dgh_raw = mask*dpre,
 * db_hh = column sums of dpre), dhp = dh*z (direct path to the masked h_prev) and hpm = mask*h_prev.
 * m2h_gru_bwd_combine: out = a + mask_row*(b + c)  (a may be NULL): total gradient reaching h_{t-1}. */
int m2h_gru_gates_bwd(const float* gi, const float* gh_raw, const float* bhh, const float* hprev, const float* mask, const float* dh,
                      float* dgi, float* dpre, float* dhp, float* hpm, int M, int H, m2h_stream stream);
int m2h_gru_bwd_combine(const float* a, const float* b, const float* c, const float* mask, float* out, int M, int H, m2h_stream stream);
/* The recurrent half of one GRU backward step for M <= 16 rows in one launch: out = a + mask_row*(dpre W_hh + dhp), i.e. the
 * [M,3H]x[3H,H] product and m2h_gru_bwd_combine (torch autograd of rnn_state_encoder.py:86-137 at the rollout width).
 * whh_t: W_hh^T as [H][3H] (m2h_pack_dgrad_weight of weight_hh_l0); a may be NULL.  H % 16 == 0. */
int m2h_gru_bwd_rec(const float* dpre, const float* whh_t, const float* a, const float* dhp, const float* mask, float* out, int M, int H,
                    m2h_stream stream);
/* m2h_gru_bwd_rec of time step t followed, in the same launch, by m2h_gru_gates_bwd of step t-1 with dh = the out just produced
 * (the gate backward is elementwise in (row, hidden unit): each thread continues with the element it wrote): one launch per BPTT
 * step.  *_prev: step t-1's gi, gh_raw, its previous hidden state and mask (inputs), dgi, dpre, hpm (outputs); dhp is read
 * (step t's) and rewritten (step t-1's) in place.  dpre_prev must not alias dpre. */
int m2h_gru_bwd_step(const float* dpre, const float* whh_t, const float* a, float* dhp, const float* mask, float* out, const float* gi_prev,
                     const float* gh_prev, const float* bhh, const float* hprev_prev, const float* mask_prev, float* dgi_prev, float* dpre_prev,
                     float* hpm_prev, int M, int H, m2h_stream stream);

/* Backward of m2h_policy_heads: g_value[M], g_logp[M], g_ent[M] = dL/d(value | logp_act | entropy row) (each may be NULL); dz [M][ZS] receives
 * (dL/dlogits[0..A), dL/dvalue, 0...) with ZS = A+1 rounded up to a multiple of 4; dfeats [M][H]. */
int m2h_policy_heads_bwd(const float* logp_all, const float* probs, const long long* actions, const float* g_value, const float* g_logp,
                         const float* g_ent, const float* Wa, const float* Wc, float* dz, float* dfeats, int M, int H, int A, int ZS,
                         m2h_stream stream);

/* Parameter gradients of the two heads (common/utils.py:42-50, rl/ppo/policy.py:15-23 through autograd) from m2h_policy_heads_bwd's dz [M][ZS] and the
 * features [M][H]: dw [ZS][H] = dz^T feats (rows 0..A-1: the action head's weight, row A: the critic's), db [ZS] = column sums of dz.  H % 64 == 0, ZS 4 | 8. */
int m2h_policy_heads_wgrad(const float* feats, const float* dz, float* dw, float* db, int M, int H, int ZS, m2h_stream stream);

/* F.l1_loss(pred, gt) with gt read strided from an interleaved tensor (ppo.py:212-221; passive_trainer.py:271-275):
 * loss[0] = mean |pred - gt|, grad[i] = sign(pred-gt)/n (NULL to skip).  scratch: >= 1024 floats. */
int m2h_l1_loss(const float* pred, const float* gt, int gt_stride, int gt_off, float* loss, float* grad, float* scratch, size_t n,
                m2h_stream stream);

/* m2h_l1_loss for a prediction still in its convolution's NHWC layout y[B][32][T][16] (AcousticMem's last conv before the de-slice of
 * memory_nets.py:62-67: pred[b][band * 32 + row][t] = y[b][row][t][band]) against gt [B][512][T][gt_stride]; dy (NULL to skip) =
 * d loss / d y in the same NHWC layout, i.e. what m2h_conv_wgrad_* / the input-gradient launch of that conv read.  One pass instead
 * of de-sliced store + m2h_l1_loss + re-slice of the gradient.  scratch: >= min(32 * B, 2048) floats. */
int m2h_l1_loss_nhwc16(const float* y, const float* gt, int gt_stride, int gt_off, float* loss, float* dy, float* scratch, int B, int T,
                       m2h_stream stream);

/* Binaural separation L1 (ppo.py:219-221; passive_trainer.py:270-272): loss[0] = mean |(exp(mix)-1)*masks - gt[..., cstep*c]|, c < 2,
 * over [npix][2]; gt: [npix][Cg] (cstep 2 reads the magnitudes of interleaved (mag, phase) components, cstep 1 a plain
 * 2-channel tensor); grad_masks (NULL to skip) = d loss / d masks.  mix, masks: [npix][2]. */
int m2h_bin_l1_loss(const float* mix, const float* masks, const float* gt_bin_comps, int Cg, int cstep, float* loss, float* grad_masks,
                    float* scratch, size_t npix, m2h_stream stream);

/* nn.utils.clip_grad_norm_ (ppo.py:254-268) over one flat gradient buffer: coef[0] = min(1, max_norm/(||g||_2 + 1e-6))
 * (1 when max_norm <= 0), coef[1] = ||g||_2; stays on the device (no host sync).  scratch: >= 1024 floats. */
int m2h_grad_clip_coef(const float* g, size_t n, float max_norm, float* coef, float* scratch, m2h_stream stream);

/* torch.optim.Adam step (ppo.py:48-55: eps=1e-5, no weight decay, no amsgrad) over flat buffers; g is first scaled in place by
 * coef[0]*gscale (clip coefficient from m2h_grad_clip_coef, gscale = 1/world_size after a sum all-reduce). step >= 1. */
int m2h_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step,
                  const float* coef, float gscale, m2h_stream stream);

/* The same step with its step-dependent scalars read from device memory -- hyper[0] = lr, hyper[1] = 1 - beta1^t, hyper[2] =
 * sqrt(1 - beta2^t) -- so that the launch can sit inside a captured HIP graph (the host refreshes `hyper` before each replay:
 * m2h_adam_hyper writes the three values for step t, computed with m2h_adam_step's own host arithmetic, by a one-thread launch).  Same
 * arithmetic as m2h_adam_step; p / g / m / v may be a sub-range of the flat buffers (one network's parameters). */
int m2h_adam_hyper(float lr, float beta1, float beta2, int step, float* hyper /* 3 floats, DEVICE memory */, m2h_stream stream);
int m2h_adam_step_dev(float* p, float* g, float* m, float* v, size_t n, const float* hyper, float beta1, float beta2, float eps,
                      const float* coef, float gscale, m2h_stream stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Passive pre-training (pretrain/passive/passive_trainer.py:218-286): train-mode BatchNorm and transposed-conv gradients.
 * ------------------------------------------------------------------------------------------------------------------ */

/* binSep stage-0 input for training: the 16-way slice of mix plus the (target_class+1) plane as a real channel
 * (separator_cnn.py:85-99): out [B][F/16][T][ldo], channels [0,16C) = slice, 16C = cls_val[b], rest 0; ldo % 4 == 0. */
int m2h_sep_slice_input_plane(const float* mix, const float* cls_val, float* out, int B, int F, int T, int C, int ldo, m2h_stream stream);

/* nn.BatchNorm2d in train mode + LeakyReLU/ReLU (separator_cnn.py:9-12,21-24) on z [M][C] (NHWC conv output), C % 4 == 0:
 * batch mean / biased variance (Welford-combined, ordered), y = act((z-mean)*invstd*gamma+beta), running statistics updated
 * in place with `momentum` and the unbiased variance (NULL to skip).  mean/invstd [C] are saved for the backward.
 * Backward: given dy (w.r.t. y), returns dgamma, dbeta and dz.  workspace: m2h_bn_workspace_bytes(M, C). */
size_t m2h_bn_workspace_bytes(int M, int C);
int m2h_bn_train_fwd(const float* z, const float* gamma, const float* beta, float eps, float momentum, float slope, float* running_mean,
                     float* running_var, float* mean, float* invstd, float* y, int M, int C, float* workspace, m2h_stream stream);
int m2h_bn_train_bwd(const float* dy, const float* y, const float* z, const float* mean, const float* invstd, const float* gamma, float slope,
                     float* dgamma, float* dbeta, float* dz, int M, int C, float* workspace, m2h_stream stream);

/* Weight gradient of ConvTranspose2d(4,2,1) (separator_cnn.py:15-24; passive_trainer.py:218-286 backward) in ONE launch + one
 * reduce + one scatter: args = the geometry of a sub-pixel phase (taps 2x2, stride 1, os 2, off 0, Hq = Hi, Ho = 2*Hi; ph / pw / mul are
 * ignored: grid y walks the four phases, each reading dy at its output pixels (2q+ph, 2r+pw) and stepping its taps by 2*ph-1 /
 * 2*pw-1); the ordered reduce over the pixel splits runs once for all phases and a transposing scatter writes dw in the torch
 * layout [Ci][Co][4][4].  workspace: m2h_convT_wgrad_workspace_bytes(args) (four phases of slabs + the packed gradients).  Same
 * sums, in the same order, as the per-phase calls + m2h_unpack_convT_wgrad below (3 launches per layer instead of 9). */
size_t m2h_convT_wgrad_workspace_bytes(const m2h_conv_args* args /* host */);
int m2h_convT_wgrad_f32(const m2h_conv_args* args /* host */, const float* dy, int ldy, float* dw, m2h_stream stream);

/* The same gradient phase by phase: run m2h_conv_wgrad_f32 once per sub-pixel phase (args: taps 2x2, mul = 2*ph-1,
 * off 0, stride 1, os 2, phase (ph,pw), Ho = 2*Hi: the dy rows are then read at output pixel (2q+ph, 2r+pw)) into
 * dwp[phase][Co][4*Ci], then scatter to the torch layout dw [Ci][Co][4][4] with this call. */
int m2h_unpack_convT_wgrad(const float* dwp, float* dw, int Ci, int Co, m2h_stream stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Feeder STFT (A20 / N1) and eval iSTFT (A21 / N2).  librosa 0.8.0 semantics (third-party dependency of the reference:
 * dataset.py:190-228, simulator_train.py:425-486, eval_metrics.py:232-251).  The DFT itself is a [frames x 1024] x
 * [1024 x 1024] fp32 GEMM through m2h_conv_igemm_f32 (cos/-sin matrix built on the host); these are the glue kernels.
 * ------------------------------------------------------------------------------------------------------------------ */

/* frames[s][t][n] = window[n] * y_reflect[s][t*hop + n - n_fft/2] for n < n_fft, 0 for n_fft <= n < ldf.
 * y: [S][L] waveforms; np.pad(mode="reflect") indexing; T = 1 + (L + 2*(n_fft/2) - n_fft)/hop frames. */
int m2h_stft_frames(const float* y, const float* window, float* frames, int S, int L, int T, int n_fft, int hop, int ldf, m2h_stream stream);

/* spec [S = B*C][T][lds] holds Re in [0,nb) and Im in [nb,2nb) (GEMM output).  Writes BHWC [B][nb][T][C]:
 * mag_out = |X| (mode 0), log1p|X| (mode 1, dataset.py:228) or log1p(fp16(|X|)) (mode 2, simulator_train.py:437-441,483-486);
 * phase_out = atan2(Im, Re) (np.angle).  Either output may be NULL. */
int m2h_stft_post(const float* spec, float* mag_out, float* phase_out, int B, int C, int T, int nb, int lds, int mode, m2h_stream stream);

/* iSTFT: rows[b][t][k] = mag*cos(phase), rows[b][t][nb+k] = mag*sin(phase) for channel c of BHWC mag/phase [B][nb][T][C]
 * (eval_metrics.py:243,248: mag * exp(1j*phase)); the inverse real DFT is again a GEMM; then windowed overlap-add with
 * window-sum-of-squares normalisation, centre trim n_fft/2 and fixed output length: y [S][length]. */
int m2h_istft_pre(const float* mag, const float* phase, float* rows, int B, int C, int c, int T, int nb, int ldr, m2h_stream stream);
int m2h_istft_ola(const float* frames, const float* window, float* y, int S, int T, int n_fft, int hop, int ldf, int length, m2h_stream stream);

/* RIR-convolution feeder glue (pretrain/datasets/dataset.py:178-186,214-216; habitat_audio/simulator_train.py:416-424).
 * m2h_feeder_round_mix: takes the "same"-mode window [start, start+L) of S full linear convolutions (rows of ldfull floats),
 * applies np.round -> int16 -> float32 / 32768, optionally stores it (conv_out [S][L], NULL to skip) and accumulates it into
 * mix [S][L] (first != 0: overwrite), scaling the accumulated mixture by mix_scale (1 / num_sources on the last source).
 * m2h_rms_normalize: mag[s] *= norm / sqrt(mean(mag[s]^2)) unless that RMS is 0 (GT_MONO_MAG_NORM, dataset.py:205-206). */
int m2h_feeder_round_mix(const float* full, int ldfull, int start, float* conv_out, float* mix, int S, int L, int first, float mix_scale,
                         m2h_stream stream);
int m2h_rms_normalize(float* mag, int S, int n, float norm, m2h_stream stream);

/* Waveform quality metrics (common/eval_metrics.py:12-166, scale_bss_eval / evaluate_helper for the single reference source
 * the evaluation uses; preprocess :170-196 removes every signal's mean and averages the mixture's channels).
 * ref, est, mix_l, mix_r: [S][L] waveforms (mix_r may be NULL for a mono mixture).
 * out [S][11] = si_sdr, si_sir, si_sar, sd_sdr, snr, srr, si_sdri, sd_sdri, snri, si_siri, si_sari (dB; the "i" metrics are the
 * improvement over using the mixture as the estimate). */
int m2h_bss_metrics(const float* ref, const float* est, const float* mix_l, const float* mix_r, float* out, int S, int L, m2h_stream stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Whole-network runner: one C call enqueues every kernel of one separator U-Net forward (K1/K2 slice, 5 down stages, 5 up
 * stages, head) = PassiveSepEncCNN.forward + PassiveSepDecCNN.forward (separator_cnn.py:70-108,153-170) in eval mode.  Same
 * kernels and results as the per-op entry points; exists because at rollout batch sizes (14 envs) the per-launch host cost
 * of a Python-driven chain exceeds the GPU time.  All pointers are device pointers to buffers produced by the m2h_pack_* /
 * m2h_fold_bn calls; intermediates live in the caller-owned workspace.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct m2h_unet_weights {
  const float* down_w[5];     /* packed conv weights [Co][16*Ci] (stage 0: Ci = 32, class plane dropped) */
  const float* down_scale[5]; /* folded BN */
  const float* down_shift[5];
  const float* cls_table;     /* [9][64] for binSep stage 0, NULL for bin2mono */
  const float* up_w[5];       /* packed transposed-conv weights [4][Co][4*Cin] */
  const float* up_scale[5];
  const float* up_shift[5];
  const float* head_w;        /* [n_out][n_out] */
  const float* head_b;        /* [n_out] */
  int n_out;                  /* 32 (binSep) or 16 (bin2mono) */
  int weights_split32;        /* bf16x3 math only: down_w / up_w are in the split32 layout (m2h_split32 of the packed weights); the runner
                                 then keeps every intermediate activation in split32 too, so no kernel converts operands in its k-loop */
  int math_mode;              /* arithmetic of this call: 0 = the calling thread's (m2h_set_math_mode), 1 = fp32 MFMA, 2 = bf16x3 */
  const void* down0_strip;    /* first-stage weights in the strip kernel's register image (m2h_pack_strip_conv1) or NULL.  With it, split32
                                 weights and T % 64 == 0 the runner replaces the slice kernel + first encoder stage by m2h_strip_conv1_fwd */
  int cls_kind;               /* what m2h_unet_fwd's cls_val argument points at: 0 = [B] floats holding target_class + 1 (the class plane's value,
                                 separator_cnn.py:93-96); 1 = the raw target_class as [B] float32, 2 = as [B] int64: the ".float() + 1" of :96 then
                                 happens inside the network's first kernel (no launch of its own) */
} m2h_unet_weights;

size_t m2h_unet_fwd_workspace_bytes(int B, int F, int T);
/* mix [B][F][T][2]; masks [B][F][T][2] or NULL (binSep); cls_val [B] = target_class + 1 (binSep) or NULL; out BHWC
 * [B][F][T][n_out/16].  F = 512 (16 slices of 32 rows), T % 32 == 0. */
int m2h_unet_fwd(const m2h_unet_weights* w /* host */, const float* mix, const float* masks, const float* cls_val, float* out,
                 int B, int F, int T, void* workspace, size_t workspace_bytes, m2h_stream stream);
/* Same, recording caller-created events (hipEvent_t handles, e.g. torch.cuda.Event.cuda_event) on `stream` before the first
 * kernel and after each of the 11 kernels: events[0..11] (n_events must be 12).  Kernel i ran between events[i] and
 * events[i+1]: 0 slice, 1-5 encoder stages, 6-9 decoder stages, 10 last stage + head.  When the strip kernel takes the slice and
 * the first stage together (down0_strip), interval 0 is empty and interval 1 is that one kernel; interval 10 is then
 * m2h_strip_last_fwd. */
#define M2H_UNET_FWD_EVENTS 12
int m2h_unet_fwd_events(const m2h_unet_weights* w /* host */, const float* mix, const float* masks, const float* cls_val, float* out,
                        int B, int F, int T, void* workspace, size_t workspace_bytes, void* const* events, int n_events,
                        m2h_stream stream);
/* Which kernel ran: the label of the calling thread's most recent launch through this library (thread-local, like the error
 * string), and the labels of the 11 stages of its most recent m2h_unet_fwd / m2h_unet_fwd_events call (stage numbering as the
 * event intervals above).  For benchmark tables and profiles -- the dispatch rules live in the library, not in its callers. */
const char* m2h_last_kernel(void);
const char* m2h_unet_fwd_stage_kernel(int stage);

/* Strip-walker kernel of the first encoder stage (csrc/conv_strip.hip; bf16x3 arithmetic): the 16-way frequency slice
 * (separator_cnn.py:85-90), for bin2mono the pre-op log1p(clamp0(mask (exp(mix) - 1))) (:77-79), the (target_class + 1) plane
 * (:93-99, as cls_val[b] * cls_table[border class][n]) and Conv2d(4x4, s2, p1) + BN(eval) + LeakyReLU (:5-12, :101-105) in ONE
 * launch: mix / masks [B][512][T][2] fp32 -> dst split32 NHWC [B][16][T/2][64].  A workgroup walks a strip of 32 output columns
 * down the image with a rolling window of input rows in LDS (every input byte fetched once) and the whole weight matrix in
 * registers.  wreg: m2h_pack_strip_conv1 of the torch weight [64][Ci >= 32][4][4] (m2h_strip_conv1_weight_bytes() bytes).
 * F = 512, T % 64 == 0. */
size_t m2h_strip_conv1_weight_bytes(void);
int m2h_pack_strip_conv1(const float* w, int Ci, void* out, m2h_stream stream);
int m2h_strip_conv1_fwd(const float* mix, const float* masks, const void* wreg, const float* scale, const float* shift,
                        const float* cls_table, const float* cls_val, float* dst, int B, int F, int T, float slope, m2h_stream stream);

/* Strip-walker kernel of the last decoder stage + head (csrc/conv_strip.hip; bf16x3 arithmetic): cat(x, skip) ->
 * ConvTranspose2d(128 -> Co, 4x4, s2, p1) + BN(eval) + ReLU (separator_cnn.py:15-24,128-133,156-161), the biased 1x1 conv (:134)
 * and the de-slice to BHWC (:163-168) in ONE launch: x, skip split32 NHWC [B][H][W][64]; wp_split32 = m2h_split32 of
 * m2h_pack_convT_weight ([4][Co][4 * 128]); out [B][32 H][2 W][Co / 16] fp32.  Same structure as m2h_strip_conv1_fwd: strips of
 * 32 columns, a rolling window of three input rows in LDS, all weights in registers (wave = sub-pixel phase x 16 channels), the
 * head on the matrix pipe from an LDS image of the activated tile.  W % 32 == 0, Co 32 or 16. */
int m2h_strip_last_fwd(const float* x, const float* skip, const float* wp_split32, const float* scale, const float* shift, const float* head_w,
                       const float* head_b, float* out, int B, int H, int W, int Co, m2h_stream stream);

/* m2h_sep_slice_input with the output written in the split32 layout when split_out != 0 (C == 2 only). */
int m2h_sep_slice_input_fmt(const float* mix, const float* masks, float* out, int B, int F, int T, int C, int split_out, m2h_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* M2H_H */
