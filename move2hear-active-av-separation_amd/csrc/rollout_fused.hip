// Fused bookkeeping of one rollout step (gfx950): everything ppo_trainer.py:375-455 computes per environment after the two
// separator passes -- the reward (override_rewards / reward_util, env_utils.py:690-713, incl. the effective extra reward of
// :395-405), the three STFT-L2 distances (eval_metrics.py:306-366) and the per-episode statistics (:426-455) -- in ONE launch.
// As separate kernels (sq_stats x2, rewards_from_stats, stft_l2 x3, episode_stats_update) these were 61 us of a 670 us step:
// each is a per-env reduction over 16 384 spectrogram bins that occupied 14 blocks (one per env) of a 256-CU chip.
//
// Here env e is reduced by CH blocks (grid = CH x N): a block sums its slice of the bins for all eight quantities in one pass
// (every input element is read exactly once), writes eight partial sums, and takes a ticket on the env's counter; the block that
// draws the last ticket adds the CH partials in slice order (bit-reproducible: the order never depends on arrival) and does the
// per-env scalar work.  Hand-off = the guide's counter form (cdna_hip_programming.md, Guideline 16): plain partial stores ->
// vmcnt(0) -> block barrier -> lane 0: agent-scope release fence, vmcnt(0), relaxed agent-scope fetch_add; the last arriver:
// agent-scope acquire fence, then plain loads.  The counter is left at zero for the next launch.
#include "m2h_internal.h"

namespace m2h {

namespace {
constexpr int NQ = 8;   // partial sums per block: next (p-g)^2, next g^2, cur (p-g)^2, cur g^2, bin ch0, bin ch1, mono, mem

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
}  // namespace

template <int CH>
__global__ __launch_bounds__(256) void rollout_step_stats_kernel(const m2h_step_stats_args a) {
  __shared__ float sh[4][NQ];
  __shared__ int is_last;
  const int e = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = a.L, per = (L + CH - 1) / CH;
  const int i0 = chunk * per, i1 = min(L, i0 + per);
  float s[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) s[q] = 0.f;
  const size_t base = (size_t)e * L;
  for (int i = i0 + tid; i < i1; i += 256) {
    const size_t pix = base + i;
    const float mem = a.mem[pix], gt = a.gt_mono_comps[pix * 4];
    const float2 pm = *reinterpret_cast<const float2*>(a.masks + pix * 2);
    const float2 mx = *reinterpret_cast<const float2*>(a.mix + pix * 2);
    const float gl = a.gt_bin_comps[pix * 8], gr = a.gt_bin_comps[pix * 8 + 2];
    const float mono = a.mono[pix];
    if (a.override_rewards) {
      const float nm = a.next_mem[pix], ng = a.next_gt_mono_comps[pix * 4];
      const float dn = nm - ng;
      s[0] += dn * dn;
      s[1] += ng * ng;
      const float dc = mem - gt;
      s[2] += dc * dc;
      s[3] += gt * gt;
    }
    const float dl = gl - (expf(mx.x) - 1.f) * pm.x, dr = gr - (expf(mx.y) - 1.f) * pm.y;
    s[4] += dl * dl;
    s[5] += dr * dr;
    const float dm = gt - mono, df = gt - mem;
    s[6] += dm * dm;
    s[7] += df * df;
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const float t = wave_sum64(s[q]);
    if (lane == 0) sh[wave][q] = t;
  }
  __syncthreads();
  if (tid < NQ) a.partial[((size_t)e * CH + chunk) * NQ + tid] = (sh[0][tid] + sh[1][tid]) + (sh[2][tid] + sh[3][tid]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(&a.tickets[e], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = old == (unsigned)(CH - 1);
    if (is_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!is_last) return;
  // ---- the env's last block: slice-ordered sums, then the scalar work of ppo_trainer.py:385-455 for env e ----
  if (tid < NQ) {
    float t = 0.f;
    for (int c = 0; c < CH; ++c) t += a.partial[((size_t)e * CH + c) * NQ + tid];
    sh[0][tid] = t;
  }
  __syncthreads();
  if (tid != 0) return;
  a.tickets[e] = 0u;   // ready for the next launch (ordered by the kernel boundary)
  const float Lf = (float)L, L2 = (float)(2 * L);
  const float m = a.not_done[e], nd = 1.f - m;
  float r;
  if (a.override_rewards) {
    r = 0.f;
    if (m != 0.f) {
      r = -(sh[0][0] / Lf) / (sh[0][1] / Lf);
      if (a.extra_reward) r *= a.extra_mult;                       // 2 x extra_reward_multiplier: see ppo_trainer.py (m2h) on :395-405
      else r -= -(sh[0][2] / Lf) / (sh[0][3] / Lf);
    }
  } else {
    r = a.env_rewards[e];
  }
  const float bin_loss = sh[0][4] / L2 + sh[0][5] / L2;
  const float mono_loss = sh[0][6] / L2, mem_loss = sh[0][7] / L2;
  a.rewards[e] = r;
  a.losses[e] = bin_loss;
  a.losses[a.N + e] = mono_loss;
  a.losses[2 * a.N + e] = mem_loss;
  // per-episode statistics: the arithmetic of episode_stats_kernel (rl_ops.hip), same operation order, no fused multiply-adds
  const m2h_episode_stats& st = a.stats;
  const int A = a.A;
  if (a.ndgs) st.episode_ndgs[e] = __fadd_rn(st.episode_ndgs[e], __fmul_rn(nd, a.ndgs[e]));
  if (a.dgs) st.episode_dgs[e] = __fadd_rn(st.episode_dgs[e], __fmul_rn(nd, a.dgs[e]));
  const float cr = __fadd_rn(st.current_episode_reward[e], r);
  const float cs = __fadd_rn(st.current_episode_step[e], 1.f);
  const float cb = __fadd_rn(st.current_episode_bin_losses[e], bin_loss);
  const float cm = __fadd_rn(st.current_episode_mono_losses[e], mono_loss);
  const float cf = __fadd_rn(st.current_episode_monoFromMem_losses[e], mem_loss);
  st.episode_rewards[e] = __fadd_rn(st.episode_rewards[e], __fmul_rn(nd, cr));
  st.episode_steps[e] = __fadd_rn(st.episode_steps[e], __fmul_rn(nd, cs));
  st.episode_counts[e] = __fadd_rn(st.episode_counts[e], nd);
  for (int k = 0; k < A; ++k) {
    const float cp = __fadd_rn(st.current_episode_dist_probs[e * A + k], a.probs[e * A + k]);
    st.episode_dist_probs[e * A + k] = __fadd_rn(st.episode_dist_probs[e * A + k], __fmul_rn(nd, __fdiv_rn(cp, cs)));
    st.current_episode_dist_probs[e * A + k] = __fmul_rn(cp, m);
  }
  st.episode_bin_losses_allSteps[e] = __fadd_rn(st.episode_bin_losses_allSteps[e], __fmul_rn(nd, __fdiv_rn(cb, cs)));
  st.episode_mono_losses_lastStep[e] = __fadd_rn(st.episode_mono_losses_lastStep[e], __fmul_rn(nd, mono_loss));
  st.episode_mono_losses_allSteps[e] = __fadd_rn(st.episode_mono_losses_allSteps[e], __fmul_rn(nd, __fdiv_rn(cm, cs)));
  st.episode_monoFromMem_losses_lastStep[e] = __fadd_rn(st.episode_monoFromMem_losses_lastStep[e], __fmul_rn(nd, mem_loss));
  st.episode_monoFromMem_losses_allSteps[e] = __fadd_rn(st.episode_monoFromMem_losses_allSteps[e], __fmul_rn(nd, __fdiv_rn(cf, cs)));
  st.current_episode_reward[e] = __fmul_rn(cr, m);
  st.current_episode_step[e] = __fmul_rn(cs, m);
  st.current_episode_bin_losses[e] = __fmul_rn(cb, m);
  st.current_episode_mono_losses[e] = __fmul_rn(cm, m);
  st.current_episode_monoFromMem_losses[e] = __fmul_rn(cf, m);
}

}  // namespace m2h

using namespace m2h;

extern "C" {

size_t m2h_step_stats_workspace_bytes(int N) { return N > 0 ? (size_t)N * M2H_STEP_STATS_CHUNKS * NQ * sizeof(float) : 0; }

int m2h_rollout_step_stats(const m2h_step_stats_args* args, m2h_stream stream) {
  M2H_REQUIRE(args != nullptr, "rollout_step_stats: null args");
  const m2h_step_stats_args& a = *args;
  M2H_REQUIRE(a.N > 0 && a.L > 0 && a.A > 0, "rollout_step_stats: non-positive size");
  M2H_REQUIRE(a.mem && a.gt_mono_comps && a.masks && a.mix && a.gt_bin_comps && a.mono && a.not_done && a.probs && a.rewards && a.losses &&
                  a.partial && a.tickets,
              "rollout_step_stats: null tensor");
  M2H_REQUIRE(!a.override_rewards || (a.next_mem && a.next_gt_mono_comps), "rollout_step_stats: reward override needs the next-step tensors");
  M2H_REQUIRE(a.override_rewards || a.env_rewards, "rollout_step_stats: env rewards missing");
  const float* const* fields = reinterpret_cast<const float* const*>(&a.stats);
  for (size_t i = 0; i < sizeof(m2h_episode_stats) / sizeof(float*); ++i) M2H_REQUIRE(fields[i], "rollout_step_stats: null statistics tensor");
  M2H_LAUNCH((rollout_step_stats_kernel<M2H_STEP_STATS_CHUNKS>), dim3(M2H_STEP_STATS_CHUNKS, a.N), dim3(256), 0, as_stream(stream), a);
  return launch_status("rollout_step_stats");
}

}  // extern "C"
