"""Phase timeline inside the small-batch conv kernels (diagnostic build with M2H_SMALL_DBG=16: wall-clock stamps of thread 0 of every
block): the binSep U-Net at 14 envs, stage by stage.
    bash tools/build_variant.sh sstamp conv_small.hip -DM2H_SMALL_DBG=16
    M2H_LIB=build/libm2h_sstamp.so python tools/small_stamps.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from m2h import _lib, ops, synthetic  # noqa: E402
from m2h.common.spaces import move2hear_observation_space  # noqa: E402
from m2h.pretrain.passive.policy import Move2HearPassiveWoMemoryPolicy  # noqa: E402

B = 14
dev = torch.device("cuda", 0)
pol = Move2HearPassiveWoMemoryPolicy(move2hear_observation_space())
pol.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.passive_shapes(), 1).items()})
pol = pol.to(dev).eval()
mixed, tc = synthetic.make_passive_inputs(B, 32, 3)
obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.m2h_debug_small_stamps.argtypes = [ctypes.c_void_p]
buf = np.zeros((4096, 8), dtype=np.uint64)
# run the network stage by stage: stage i alone is what the stamps of the LAST launch hold, so launch the network with the
# later stages' stamps overwritten ... simpler: the stamps array is indexed by block only, so run the whole U-Net once per stage of
# interest with every OTHER stage on a 1-block-is-enough check: here we just read after each full pass and report the LAST stage
# whose block count covers the index -- instead, use ops.conv_small through the runner's per-stage override being identical.
ops.debug_set(37, 1)
names = ["down0", "down1", "down2", "down3", "down4", "up0", "up1", "up2", "up3", "up4+head"]
with torch.no_grad():
    for _ in range(3):
        pol.get_binSepMasks(obs)
torch.cuda.synchronize()
# one U-Net pass leaves, per block index, the stamps of the last stage that had such a block; to separate stages, launch the
# pass ten times, each time stopping after stage i (M2H_SMALL_STOP env read by this tool -> unet_small_tiling(stage, ...) cannot
# stop a pass, so the tool re-runs the pass with later stages given a tiling the engine refuses and catches the error).
for i in range(10):
    for j in range(10):
        ops.unet_small_tiling(j, None)
    if i < 9:
        ops.unet_small_tiling(i + 1, (1, 1, 48, 16, 1))   # refused ("power of two"): the pass ends after stage i
    try:
        with torch.no_grad():
            pol.get_binSepMasks(obs)
    except RuntimeError:
        pass
    torch.cuda.synchronize()
    rc = lib.m2h_debug_small_stamps(buf.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    st = buf.astype(np.int64)
    live = st[:, 0] > 0
    # blocks of THIS stage: those whose start stamp lies after the previous stage's
    t0 = st[live, 0]
    newest = t0 >= t0.max() - 3000   # within 30 us of the latest start
    s = st[live][newest]
    base = s[:, 0].min()
    d = (s - base) / 100.0   # us
    ph = np.diff(d, axis=1)
    print("%-9s blocks %4d  start spread %5.2f us | W issue %5.2f  stage %5.2f  rows %5.2f  mfma %5.2f  barrier %5.2f  reduce+store %5.2f  head %5.2f | block total med %5.2f max-end %5.2f"
          % (names[i], len(s), d[:, 0].max(), *np.median(ph, axis=0).tolist(), np.median(d[:, 7] - d[:, 0]), d[:, 7].max()))
    buf[:] = 0
