cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_gap
rocprofv3 --kernel-trace -d gpurun_out/prof_gap -o g --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --ddppo-cycles 0 --train-steps 0 > gpurun_out/prof_gap_log.txt 2>&1
python3 - <<'P'
import csv, glob
f = glob.glob("gpurun_out/prof_gap/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the headline phase: find runs of 22+ kernels starting with sep_slice_input and look at the first 8 complete steps
names = [r["Kernel_Name"].split("(")[0].replace("void m2h::", "").replace("m2h::", "")[:44] for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("sep_slice_input")]
# steps = pairs of U-Nets: take windows between every second slice kernel
seen = 0
for a, b in zip(idx[0::2], idx[2::2]):
    if b - a > 30:
        continue
    seen += 1
    if seen < 5 or seen > 7:
        continue
    tot = st[b] - st[a]
    busy = sum(en[i] - st[i] for i in range(a, b))
    print("step window: %d kernels, wall %.1f us, kernel time %.1f us, gaps %.1f us" % (b - a, tot / 1e3, busy / 1e3, (tot - busy) / 1e3))
    if seen == 6:
        for i in range(a, b):
            print("   %-46s dur %7.1f us   gap after %6.1f us" % (names[i], (en[i] - st[i]) / 1e3, (st[i + 1] - en[i]) / 1e3))
P
rm -rf gpurun_out/prof_gap
