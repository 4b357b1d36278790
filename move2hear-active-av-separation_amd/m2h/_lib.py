"""ctypes binding of libm2h.so (the C-ABI declared in include/m2h.h) and its in-tree build.

The library is built IN-TREE (``m2h/libm2h.so``) with ``hipcc --offload-arch=gfx950`` so that it travels
with the repository snapshot to the GPU box.  There is no CPU fallback: if the library is missing or does
not load, every op raises ``RuntimeError``.
"""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_ROOT = os.path.dirname(_HERE)
_REPO_ROOT = os.path.dirname(_PKG_ROOT)
CSRC = os.path.join(_PKG_ROOT, "csrc")
INCLUDE = os.path.join(_REPO_ROOT, "include")
# M2H_LIB: kernel-tuning override -- load an experimental build of the same C-ABI (tools/build_variant.sh) instead of the in-tree one
LIB_PATH = os.environ.get("M2H_LIB") or os.path.join(_HERE, "libm2h.so")
SOURCES = ["conv_igemm.hip", "conv_dma.hip", "conv_patch.hip", "convt_quad.hip", "conv_strip.hip", "acoustic_mem.hip", "conv_bwd.hip", "bn.hip", "stft.hip", "layout.hip", "rl_ops.hip", "rollout_fused.hip", "pack_batch.hip", "fftconv.hip", "api.hip"]

_lock = threading.Lock()
_lib = None


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "m2h_internal.h"), os.path.join(CSRC, "igemm_common.h"), os.path.join(CSRC, "lds_dma.h"), os.path.join(INCLUDE, "m2h.h"), os.path.join(INCLUDE, "m2h_tuning.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compiles csrc/*.hip for gfx950 into m2h/libm2h.so (cross-compiles without a GPU).  One object per source under
    csrc/build/, compiled in parallel and re-used while the source and the headers are older (force=True recompiles all)."""
    if not force and not _stale():
        if os.path.exists(CLOCK_DIAG_LIB) and LIB_PATH == os.path.join(_HERE, "libm2h.so"):
            build_clock_diag(_from_build=True)   # (a no-op while the diagnostic copy is newer than every source)
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC]
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    newest_header = max(os.path.getmtime(h) for h in headers)

    def compile_one(src):
        path, obj = os.path.join(CSRC, src), os.path.join(objdir, src + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), newest_header):
            return obj
        cmd = [hipcc] + flags + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stdout))
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), 7)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    if os.path.exists(CLOCK_DIAG_LIB) and LIB_PATH == os.path.join(_HERE, "libm2h.so"):
        build_clock_diag(_from_build=True)   # the diagnostic copy exports the same C-ABI: it never lags behind the library it shadows
    return LIB_PATH


CLOCK_DIAG_LIB = os.path.join(_HERE, "libm2h_clockdiag.so")
CLOCK_DIAG_SOURCES = ("conv_igemm.hip", "conv_dma.hip", "conv_patch.hip")   # the units that carry M2H_CLOCK_DIAG stamps


def build_clock_diag(force=False, _from_build=False):
    """DIAGNOSTIC copy of the library, never loaded by the product path: the three conv-engine units compiled with -DM2H_CLOCK_DIAG (two
    s_memtime / s_memrealtime stamps around each block's k-loop, written to a buffer of their own), every other unit's object shared
    with ``build()``.  tools/clock_probe.py loads it in a child process of bench.py -- one stamped run of the headline's dominant kernel
    OUTSIDE the timed region -- to report the shader clock the chip holds inside that kernel (``roofline.clock_ghz``)."""
    if not _from_build:
        build()
    objdir = os.path.join(CSRC, "build")
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    if not force and os.path.exists(CLOCK_DIAG_LIB) and os.path.getmtime(CLOCK_DIAG_LIB) > max(os.path.getmtime(d) for d in deps):
        return CLOCK_DIAG_LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DM2H_CLOCK_DIAG", "-I" + INCLUDE, "-I" + CSRC]

    def compile_one(src):
        obj = os.path.join(objdir, src + ".clockdiag.o")
        r = subprocess.run([hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s (clock-diag build):\n%s" % (src, r.stdout))
        return obj

    with ThreadPoolExecutor(max_workers=3) as ex:
        diag_objs = dict(zip(CLOCK_DIAG_SOURCES, ex.map(compile_one, CLOCK_DIAG_SOURCES)))
    objs = [diag_objs.get(s, os.path.join(objdir, s + ".o")) for s in SOURCES]
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", CLOCK_DIAG_LIB + ".tmp"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed (clock-diag build):\n" + r.stdout)
    os.replace(CLOCK_DIAG_LIB + ".tmp", CLOCK_DIAG_LIB)
    return CLOCK_DIAG_LIB


class ConvArgs(ctypes.Structure):
    """Mirror of ``struct m2h_conv_args`` (include/m2h.h)."""
    _fields_ = [
        ("src0", ctypes.c_void_p), ("src1", ctypes.c_void_p),
        ("C0", ctypes.c_int), ("C1", ctypes.c_int),
        ("B", ctypes.c_int), ("Hi", ctypes.c_int), ("Wi", ctypes.c_int),
        ("Hq", ctypes.c_int), ("Wq", ctypes.c_int),
        ("stride", ctypes.c_int),
        ("nth", ctypes.c_int), ("ntw", ctypes.c_int),
        ("mulh", ctypes.c_int), ("offh", ctypes.c_int), ("mulw", ctypes.c_int), ("offw", ctypes.c_int),
        ("conv_transpose", ctypes.c_int),
        ("wp", ctypes.c_void_p),
        ("N", ctypes.c_int),
        ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
        ("slope", ctypes.c_float),
        ("cls_table", ctypes.c_void_p), ("cls_val", ctypes.c_void_p),
        ("dst", ctypes.c_void_p),
        ("Ho", ctypes.c_int), ("Wo", ctypes.c_int),
        ("os", ctypes.c_int), ("ph", ctypes.c_int), ("pw", ctypes.c_int),
        ("ldc", ctypes.c_int), ("out_mode", ctypes.c_int),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
        ("head_w", ctypes.c_void_p), ("head_b", ctypes.c_void_p),
        ("operand_format", ctypes.c_int),
    ]


class UnetWeights(ctypes.Structure):
    """Mirror of ``struct m2h_unet_weights`` (include/m2h.h)."""
    _fields_ = [
        ("down_w", ctypes.c_void_p * 5), ("down_scale", ctypes.c_void_p * 5), ("down_shift", ctypes.c_void_p * 5),
        ("cls_table", ctypes.c_void_p),
        ("up_w", ctypes.c_void_p * 5), ("up_scale", ctypes.c_void_p * 5), ("up_shift", ctypes.c_void_p * 5),
        ("head_w", ctypes.c_void_p), ("head_b", ctypes.c_void_p),
        ("n_out", ctypes.c_int),
        ("weights_split32", ctypes.c_int),
        ("math_mode", ctypes.c_int),
        ("down0_strip", ctypes.c_void_p), ("cls_kind", ctypes.c_int),
    ]


EPISODE_STATS_FIELDS = (
    "episode_rewards", "episode_counts", "episode_steps", "episode_dist_probs", "episode_bin_losses_allSteps",
    "episode_mono_losses_lastStep", "episode_mono_losses_allSteps", "episode_monoFromMem_losses_lastStep",
    "episode_monoFromMem_losses_allSteps", "current_episode_reward", "current_episode_step", "current_episode_dist_probs",
    "current_episode_bin_losses", "current_episode_mono_losses", "current_episode_monoFromMem_losses", "episode_ndgs", "episode_dgs")


class EpisodeStats(ctypes.Structure):
    """Mirror of ``struct m2h_episode_stats`` (include/m2h.h)."""
    _fields_ = [(name, ctypes.c_void_p) for name in EPISODE_STATS_FIELDS]


class StepStatsArgs(ctypes.Structure):
    """Mirror of ``struct m2h_step_stats_args`` (include/m2h.h)."""
    _fields_ = ([(n, ctypes.c_void_p) for n in ("next_mem", "next_gt_mono_comps", "mem", "gt_mono_comps", "masks", "mix", "gt_bin_comps", "mono",
                                                "not_done", "env_rewards", "probs", "ndgs", "dgs", "rewards", "losses")] +
                [("stats", EpisodeStats), ("partial", ctypes.c_void_p), ("tickets", ctypes.c_void_p), ("N", ctypes.c_int), ("L", ctypes.c_int),
                 ("A", ctypes.c_int), ("override_rewards", ctypes.c_int), ("extra_reward", ctypes.c_int), ("extra_mult", ctypes.c_float)])


class PackItem(ctypes.Structure):
    """Mirror of ``struct m2h_pack_item`` (include/m2h.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("kind", ctypes.c_int), ("p", ctypes.c_int * 6)]


PACK_BATCH_MAX = 48
PACK_CONV, PACK_CONVT, PACK_DGRAD, PACK_FC_DGRAD = 0, 1, 2, 3
STEP_STATS_CHUNKS = 16
TUNING_KNOBS = 40   # include/m2h_tuning.h
ROWS_COPY_MAX = 32


class RowCopy(ctypes.Structure):
    """Mirror of ``struct m2h_row_copy`` (include/m2h.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("bytes", ctypes.c_size_t), ("src_slot", ctypes.c_int),
                ("dst_slot", ctypes.c_int)]


_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_Z = ctypes.c_size_t

# name -> argtypes; every function returns int except m2h_last_error.  Must list every symbol of include/m2h.h
SIGNATURES = {
    "m2h_version": [],
    "m2h_sep_slice_input": [_P, _P, _P, _I, _I, _I, _I, _P],
    "m2h_pack_conv_weight": [_P, _P, _I, _I, _I, _I, _I, _P],
    "m2h_pack_convT_weight": [_P, _P, _I, _I, _P],
    "m2h_unet_class_table": [_P, _P, _I, _I, _I, _P],
    "m2h_fold_bn": [_P, _P, _P, _P, _F, _P, _P, _I, _P],
    "m2h_conv_igemm_f32": [ctypes.POINTER(ConvArgs), _P],
    "m2h_conv_igemm_workspace_bytes": [ctypes.POINTER(ConvArgs)],
    "m2h_tuning_set": [_I, _I],
    "m2h_tuning_snapshot": [_P, _I],
    "m2h_tuning_restore": [_P, _I],
    "m2h_set_math_mode": [_I],
    "m2h_get_math_mode": [],
    "m2h_launch_count": [],
    "m2h_unet_down_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _Z, _P],
    "m2h_unet_up_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P],
    "m2h_unet_down_workspace_bytes": [_I, _I, _I, _I, _I],
    "m2h_unet_up_workspace_bytes": [_I, _I, _I, _I, _I, _I],
    "m2h_unet_up_head_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "m2h_unet_head_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "m2h_pack_conv_weight_ex": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "m2h_slice_concat_input": [_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _P],
    "m2h_visual_input": [_P, _P, _P, _I, _I, _I, _P],
    "m2h_gru_gates": [_P, _P, _P, _P, _P, _P, _I, _I, _P],
    "m2h_policy_heads": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "m2h_gather_logp": [_P, _P, _P, _I, _I, _P],
    "m2h_policy_heads_act": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "m2h_policy_heads_act_rng": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "m2h_sample_actions": [_P, _P, _P, _I, _I, _P],
    "m2h_lstm_cell": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "m2h_lstm_cell_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "m2h_conv3x3_l1_nhwc16_supported": [_I, _I, _I, _I],
    "m2h_conv3x3_l1_nhwc16": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "m2h_gae_returns": [_P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _P],
    "m2h_advantages": [_P, _P, _P, _P, _I, _I, _F, _P],
    "m2h_adv_sqdiff": [_P, _P, _P, _I, _P],
    "m2h_adv_apply": [_P, _P, _P, _I, _F, _P],
    "m2h_ppo_loss": [_P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _F, _F, _P, _P, _P, _I, _P],
    "m2h_gru_gates_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "m2h_gru_bwd_combine": [_P, _P, _P, _P, _P, _I, _I, _P],
    "m2h_gru_bwd_rec": [_P, _P, _P, _P, _P, _P, _I, _I, _P],
    "m2h_gru_bwd_step": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "m2h_policy_heads_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "m2h_policy_heads_wgrad": [_P, _P, _P, _P, _I, _I, _I, _P],
    "m2h_l1_loss": [_P, _P, _I, _I, _P, _P, _P, _Z, _P],
    "m2h_l1_loss_nhwc16": [_P, _P, _I, _I, _P, _P, _P, _I, _I, _P],
    "m2h_bin_l1_loss": [_P, _P, _P, _I, _I, _P, _P, _P, _Z, _P],
    "m2h_grad_clip_coef": [_P, _Z, _F, _P, _P, _P],
    "m2h_adam_step": [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _I, _P, _F, _P],
    "m2h_adam_hyper": [_F, _F, _F, _I, _P, _P],
    "m2h_adam_step_dev": [_P, _P, _P, _P, _Z, _P, _F, _F, _F, _P, _F, _P],
    "m2h_sq_stats": [_P, _P, _I, _I, _P, _I, _I, _P],
    "m2h_rewards_from_stats": [_P, _P, _P, _P, _I, _I, _I, _F, _P],
    "m2h_conv_wgrad_workspace_bytes": [ctypes.POINTER(ConvArgs)],
    "m2h_conv_wgrad_f32": [ctypes.POINTER(ConvArgs), _P, _I, _P, _P],
    "m2h_conv_wgrad_gated_f32": [ctypes.POINTER(ConvArgs), _P, _I, _P, _F, _P, _P],
    "m2h_conv_wgrad_torch_f32": [ctypes.POINTER(ConvArgs), _P, _I, _P, _F, _P, _I, _P],
    "m2h_conv_wgrad_dgrad_fused_supported": [ctypes.POINTER(ConvArgs)],
    "m2h_conv_wgrad_dgrad_fused_f32": [ctypes.POINTER(ConvArgs), _P, _P, _P, _F, _P, _I, _P],
    "m2h_pack_dgrad_weight": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "m2h_act_bwd": [_P, _P, _F, _P, _Z, _P],
    "m2h_bias_grad_workspace_bytes": [_I, _I],
    "m2h_bias_grad": [_P, _P, _I, _I, _P, _P],
    "m2h_act_bwd_bias": [_P, _P, _F, _P, _P, _I, _I, _P, _P],
    "m2h_sep_slice_input_plane": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "m2h_bn_workspace_bytes": [_I, _I],
    "m2h_bn_train_fwd": [_P, _P, _P, _F, _F, _F, _P, _P, _P, _P, _P, _I, _I, _P, _P],
    "m2h_bn_train_bwd": [_P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _P, _P],
    "m2h_unpack_convT_wgrad": [_P, _P, _I, _I, _P],
    "m2h_convT_wgrad_f32": [ctypes.POINTER(ConvArgs), _P, _I, _P, _P],
    "m2h_convT_wgrad_workspace_bytes": [ctypes.POINTER(ConvArgs)],
    "m2h_stft_frames": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "m2h_stft_post": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "m2h_istft_pre": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "m2h_istft_ola": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "m2h_bss_metrics": [_P, _P, _P, _P, _P, _I, _I, _P],
    "m2h_split32": [_P, _P, ctypes.c_size_t, _P],
    "m2h_feeder_round_mix": [_P, _I, _I, _P, _P, _I, _I, _I, _F, _P],
    "m2h_rms_normalize": [_P, _I, _I, _F, _P],
    "m2h_unet_fwd_workspace_bytes": [_I, _I, _I],
    "m2h_unet_fwd": [ctypes.POINTER(UnetWeights), _P, _P, _P, _P, _I, _I, _I, _P, _Z, _P],
    "m2h_unet_fwd_events": [ctypes.POINTER(UnetWeights), _P, _P, _P, _P, _I, _I, _I, _P, _Z, _P, _I, _P],
    "m2h_sep_slice_input_fmt": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "m2h_last_kernel": [],
    "m2h_unet_fwd_stage_kernel": [_I],
    "m2h_strip_conv1_weight_bytes": [],
    "m2h_pack_strip_conv1": [_P, _I, _P, _P],
    "m2h_strip_conv1_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P],
    "m2h_strip_last_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "m2h_gather_envs": [_P, _P, _P, _I, _I, _I, _Z, _P],
    "m2h_gru_step": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "m2h_gru_cell": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "m2h_episode_stats_update": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "m2h_step_stats_workspace_bytes": [_I],
    "m2h_rollout_step_stats": [ctypes.POINTER(StepStatsArgs), _P],
    "m2h_pack_batch": [_P, _I, _P],
    "m2h_fftconv_full": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "m2h_rows_copy": [_P, _I, _P, _P],
    "m2h_step_index_advance": [_P, _I, _I, _P],
    "m2h_step_index_advance_rng": [_P, _I, _I, _P, ctypes.c_ulonglong, _P],
    "m2h_synth_env_step": [_P, _P, _P, _I, _I, _P],
    "m2h_synth_env_observe": [_P, _I, _P, _P, _P, _I, _P],
    "m2h_stft_l2": [_P, _P, _I, _P, _I, _I, _I, _P, _I, _I, _P],
    "m2h_acoustic_mem_small_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
}


def load():
    """Returns the loaded library; raises RuntimeError when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libm2h.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "-- the m2h ops have no CPU/PyTorch fallback." % LIB_PATH)
        # torch first: libm2h.so's libamdhip64 dependency must bind to the HIP runtime torch has loaded (one runtime per
        # process); loaded the other way round the two copies disagree about the device ("no ROCm-capable device").
        import torch  # noqa: F401
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise RuntimeError("libm2h.so failed to load: %s" % e)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.argtypes = argtypes
            fn.restype = ctypes.c_longlong if name == "m2h_launch_count" else ctypes.c_size_t if name.endswith("_bytes") else (ctypes.c_char_p if name in ("m2h_last_kernel", "m2h_unet_fwd_stage_kernel") else ctypes.c_int)
        lib.m2h_last_error.argtypes = []
        lib.m2h_last_error.restype = ctypes.c_char_p
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().m2h_last_error().decode("utf-8", "replace")
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, msg))
