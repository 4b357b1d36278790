#!/usr/bin/env python3
"""Copies the summaries of tools/profile_round3.sh (gpurun_out/r03) into profiles/ (tracked) and derives the two JSON files bench.py
reads back: r03_pmc_hbm_traffic.json (HBM bytes per launch of every headline kernel: 2 x FETCH_SIZE + WRITE_SIZE, the guide's
gfx950 correction for wide coalesced reads) and r03_ddppo_summary.json (launches and kernel time per DD-PPO cycle).
usage: python tools/collect_profiles.py [gpurun_out/r05] [r05]"""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r04")
TAG = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.path.normpath(SRC))
DST = os.path.join(ROOT, "profiles")


def copy(src, dst):
    if os.path.exists(os.path.join(SRC, src)):
        shutil.copyfile(os.path.join(SRC, src), os.path.join(DST, dst))


copy("bench/bench_kernel_stats.csv", TAG + "_bench_kernel_stats.csv")
copy("ddppo/dd_kernel_stats.csv", TAG + "_ddppo_kernel_stats.csv")
copy("ptrain/pt_kernel_stats.csv", TAG + "_passive_train_kernel_stats.csv")
copy("feeder/fd_kernel_stats.csv", TAG + "_feeder_kernel_stats.csv")
copy("bench_line_under_rocprof.json", TAG + "_bench_line_under_rocprof.json")
copy("pmc_sq_tcc.txt", TAG + "_pmc_sq_tcc.txt")


def pmc(path):
    out, name = {}, None
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+calls=(\d+)", line)
        if m:
            name = m.group(1).replace("void ", "")
            continue
        m = re.match(r"^\s+(\w+)\s+total=\S+\s+per_call=(\S+)", line)
        if m and name:
            out[name] = float(m.group(2))
    return out


fetch, write = pmc(os.path.join(SRC, "pmc_FETCH_SIZE.txt")), pmc(os.path.join(SRC, "pmc_WRITE_SIZE.txt"))
with open(os.path.join(DST, TAG + "_pmc_hbm_traffic.txt"), "w") as f:
    for part in ("pmc_FETCH_SIZE.txt", "pmc_WRITE_SIZE.txt"):
        if os.path.exists(os.path.join(SRC, part)):
            f.write("==== %s (rocprofv3 --pmc, KiB; bench.py --steps 2 --warmup 1 --no-other-mode --no-graph) ====\n" % part)
            f.write(open(os.path.join(SRC, part)).read())
kern = {}
for name in fetch:
    if name.startswith("m2h::") and name in write:
        kern[name] = {"fetch_kib_per_launch_raw": fetch[name], "fetch_correction": 2.0, "write_kib_per_launch": write[name],
                      "traffic_bytes_per_launch": int((2.0 * fetch[name] + write[name]) * 1024)}
# the bench line's dominant instantiation "igemm_patch<256,128>" (down1, down2, up1, up2 of both U-Nets) is three template
# instantiations (halo / conv, halo / transposed, whole-image / transposed): launch-weighted mean of their traffic
def calls(path):
    out = {}
    if os.path.exists(path):
        for line in open(path):
            m = re.match(r"^(\S.*?)\s+calls=(\d+)", line)
            if m:
                out[m.group(1).replace("void ", "")] = int(m.group(2))
    return out


ncalls = calls(os.path.join(SRC, "pmc_FETCH_SIZE.txt"))
group = [k for k in kern if re.match(r"m2h::igemm_patch_kernel<4, 2, (0, 0|0, 1|1, 1), 0>", k)]
dom = None
if group:
    tot = sum(ncalls.get(k, 0) for k in group)
    if tot:
        dom = "m2h::igemm_patch_kernel<4, 2, *> (launch-weighted over %s)" % ", ".join(k.split("igemm_patch_kernel")[1] for k in group)
        kern[dom] = {"fetch_correction": 2.0,
                     "traffic_bytes_per_launch": int(sum(kern[k]["traffic_bytes_per_launch"] * ncalls.get(k, 0) for k in group) / tot)}
if dom is None:
    dom = next((k for k in kern if k.startswith("m2h::igemm_dma_kernel")), None)
src = "profiles/%s_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 2 --warmup 1 --no-other-mode --no-graph; tools/profile_round%s.sh)" % (TAG, TAG[-1])
traffic = {"bf16x3": dict(kern.get(dom, {}), source=src, kernel=dom), "per_kernel": kern}
with open(os.path.join(DST, TAG + "_pmc_hbm_traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)

stats = os.path.join(SRC, "ddppo", "dd_kernel_stats.csv")
if os.path.exists(stats):
    rows = list(csv.DictReader(open(stats)))
    launches = sum(int(r["Calls"]) for r in rows)
    ms = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
    cycles = 4.0   # two warm-up + two timed cycles (bench.py --ddppo-cycles 2); set-up launches are a few hundred of the total
    top = [(r["Name"].split("(")[0].replace("void ", ""), int(r["Calls"]), round(float(r["Percentage"]), 2)) for r in rows[:10]]
    summ = {"near_target": {"launches_per_cycle": int(launches / cycles), "kernel_ms_per_cycle": round(ms / cycles, 2), "top10": top,
                            "source": "profiles/%s_ddppo_kernel_stats.csv (rocprofv3 --kernel-trace --stats, bench.py --ddppo-cycles 2 --no-far-target: four cycles incl. two of warm-up)" % TAG}}
    line = os.path.join(SRC, "ddppo_line.json")   # the cycle time of THAT run: the kernel-time share is computed against it, not against a later run's
    if os.path.exists(line):
        try:
            summ["near_target"]["s_per_cycle"] = json.load(open(line))["ddppo"]["s_per_cycle"]
        except Exception:  # noqa: BLE001
            pass
    with open(os.path.join(DST, TAG + "_ddppo_summary.json"), "w") as f:
        json.dump(summ, f, indent=1)
print(json.dumps({k: v["traffic_bytes_per_launch"] for k, v in kern.items()}, indent=1))
