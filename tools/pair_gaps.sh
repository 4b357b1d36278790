# kernel trace of the replayed headline step: per kernel duration and the gap in front of it (one replay, launch order)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_gaps
rocprofv3 --kernel-trace -d gpurun_out/prof_gaps -o g --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 0 --train-steps 0 --feeder-steps 0 > gpurun_out/pair_gaps_log.txt 2>&1
python3 - <<'P'
import csv, glob
f = glob.glob("gpurun_out/prof_gaps/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void m2h::", "").replace("m2h::", "")[:48] for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("conv1_strip_kernel<false")]
a, b = idx[-2], idx[-1]          # one whole replay: from the first kernel of the last-but-one step to the first kernel of the last
print("one replay: %d kernels, wall %.1f us, kernel time %.1f us, gaps %.1f us" % (b - a, (st[b] - st[a]) / 1e3, sum(en[i] - st[i] for i in range(a, b)) / 1e3, sum(st[i + 1] - en[i] for i in range(a, b)) / 1e3))
for i in range(a, b):
    print("  %-50s %7.1f us   gap behind it %5.1f us" % (names[i], (en[i] - st[i]) / 1e3, (st[i + 1] - en[i]) / 1e3))
P
rm -rf gpurun_out/prof_gaps
