# kernel timeline of one passive training step (two graph branches): start offset, duration, HW queue of every kernel between two bin_l1 launches
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_nodes
rocprofv3 --kernel-trace -d gpurun_out/prof_nodes -o g --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 0 --feeder-steps 0 --train-steps 12 > gpurun_out/ptrain_timeline_log.txt 2>&1
python3 - > gpurun_out/ptrain_timeline.txt <<'P'
import csv, glob
f = glob.glob("gpurun_out/prof_nodes/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void m2h::", "").replace("m2h::", "").replace("void at::native::", "at::")[:110] for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
q = [r.get("Queue_Id", "?") for r in rows]
gsz = [r.get("Grid_Size_X", "?") for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("bin_l1")]
a, b = idx[-3], idx[-2]
t0 = st[a]
print("one passive training step: %d kernels, wall %.1f us, summed kernel time %.1f us" % (b - a, (st[b] - t0) / 1e3, sum(en[i] - st[i] for i in range(a, b)) / 1e3))
busy, last = 0, t0
per_q = {}
for i in range(a, b):
    s, e = max(st[i], last), en[i]
    if e > s:
        busy += e - s
        last = e
    per_q.setdefault(q[i], [0, 0])
    per_q[q[i]][0] += 1
    per_q[q[i]][1] += en[i] - st[i]
print("time with at least one kernel running: %.1f us; per queue (kernels, us): %s" % (busy / 1e3, {k: (v[0], round(v[1] / 1e3, 1)) for k, v in per_q.items()}))
for i in range(a, b):
    print("  +%8.1f  %7.1f us  q%-3s grid %-9s %s" % ((st[i] - t0) / 1e3, (en[i] - st[i]) / 1e3, q[i], gsz[i], names[i]))
P
rm -rf gpurun_out/prof_nodes
head -3 gpurun_out/ptrain_timeline.txt
