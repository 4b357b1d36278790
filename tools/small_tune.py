"""Coordinate descent over the small-batch runner's per-stage tilings (m2h_unet_small_tiling) at the rollout batch: the objective
is the replay time of a HIP graph holding the separator pair (binSep + bin2mono: 20 launches).  Prints the table to paste into
csrc/api.hip (kSmallTiling).     python tools/small_tune.py [B] [passes]"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from m2h import graphs, ops, synthetic  # noqa: E402
from m2h.common.spaces import move2hear_observation_space  # noqa: E402
from m2h.pretrain.passive.policy import Move2HearPassiveWoMemoryPolicy  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 14
PASSES = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
pol = Move2HearPassiveWoMemoryPolicy(move2hear_observation_space())
pol.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.passive_shapes(), 1).items()})
pol = pol.to(dev).eval()
mixed, tc = synthetic.make_passive_inputs(B, 32, 3)
obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}

# stage geometry: (rows the tiles walk, pixels per tile row, Ctot, C0, N, transposed)
GEO = [(16, 16, 32, 32, 64, 0), (8, 8, 64, 64, 128, 0), (4, 4, 128, 128, 256, 0), (2, 2, 256, 256, 512, 0), (1, 1, 512, 512, 512, 0),
       (1, 1, 512, 512, 512, 1), (2, 2, 1024, 512, 256, 1), (4, 4, 512, 256, 128, 1), (8, 8, 256, 128, 64, 1), (16, 16, 128, 64, 0, 1)]
BASE = [(1, 2, 32, 32, 8), (1, 4, 64, 16, 16), (2, 4, 64, 16, 16), (8, 2, 64, 16, 16), (32, 1, 128, 16, 16), (16, 1, 128, 16, 4), (4, 2, 128, 32, 2),
        (1, 4, 128, 32, 2), (1, 8, 64, 16, 4), (1, 1, 128, 0, 2)]


def candidates(i):
    rows, wt, ctot, c0, n, up = GEO[i]
    ph = 4 if up else 1
    tiles = set()
    for target in (64, 32, 16):
        px = rows * wt
        if px >= target:
            qr = max(1, target // wt)
            tiles.add((1, min(qr, rows)))
        else:
            tiles.add((min(64, max(1, target // px)), rows))
    cgs = [ctot] if i in (0, 9) else sorted(set([c for c in (32, 64, 128, 256) if c <= c0] + [ctot]))
    colss = [0] if i == 9 else [c for c in (16, 32, 64) if c <= n]
    out = []
    for (ib, qr), cg, cols, kw in itertools.product(sorted(tiles), cgs, colss, (1, 2, 4, 8, 16)):
        nwn = (cols // 16) if cols else 2
        if ph * nwn * kw > 16:
            continue
        ntap = (4 if up else 16) if rows > 1 else (1 if up else 4)
        if kw > ntap * cg // 16:
            continue
        out.append((ib, qr, cg, cols, kw))
    return out


def measure(reps=40):
    """replay time (us) of the pair's graph with the tilings set now; None when a stage refuses its tiling"""
    try:
        with torch.no_grad():
            m = pol.get_binSepMasks(obs)
            pol.convert_bin2mono(m, mixed_audio=obs["mixed_bin_audio_mag"])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), graphs.capture(g):
            m = pol.get_binSepMasks(obs)
            pol.convert_bin2mono(m, mixed_audio=obs["mixed_bin_audio_mag"])
    except RuntimeError:
        torch.cuda.synchronize()
        return None
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _r in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


cur = list(BASE)
for i, t in enumerate(cur):
    ops.unet_small_tiling(i, t)
ops.debug_set(37, 0)
print("tiled engines (small-batch engine off): %.1f us per pair" % measure())
ops.debug_set(37, 1)
best = measure()
print("start: %.1f us per pair" % best)
for p in range(PASSES):
    for i in range(10):
        for cand in candidates(i):
            if cand == cur[i]:
                continue
            ops.unet_small_tiling(i, cand)
            t = measure()
            if t is not None and t < best - 0.15:
                best, cur[i] = t, cand
                print("  pass %d stage %d -> %s: %.1f us" % (p, i, cand, best), flush=True)
        ops.unet_small_tiling(i, cur[i])
    print("after pass %d: %.1f us per pair" % (p, best), flush=True)
print("static const int kSmallTiling[10][5] = {")
for t in cur:
    print("    {%d, %d, %d, %d, %d}," % t)
print("};")
# per-stage times with events, final table
from m2h.rl.models.separator_cnn import unet_forward  # noqa: E402


def stages(tag):
    for name, enc, dec, masks in (("binSep", pol.binSep_enc.passive_sep_encoder, pol.binSep_dec.passive_sep_decoder, None),):
        acc = np.zeros(11)
        for _ in range(20):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(12)]
            for e in ev:
                e.record()
            with torch.no_grad():
                unet_forward(enc, dec, obs["mixed_bin_audio_mag"], masks, obs["target_class"], events=ev)
            torch.cuda.synchronize()
            acc += np.array([1e3 * ev[k].elapsed_time(ev[k + 1]) for k in range(11)])
        print(tag, name, "per-stage us (eager, event-bracketed):", np.round(acc / 20, 1).tolist(), "sum %.1f" % (acc.sum() / 20))


stages("small")
ops.debug_set(37, 0)
stages("tiled")
