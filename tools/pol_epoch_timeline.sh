# kernel timeline of one update_pol epoch (parallel graph branches): start offset, duration, HW queue of every kernel between two ppo_loss launches
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_nodes
rocprofv3 --kernel-trace -d gpurun_out/prof_nodes -o g --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --ddppo-cycles 1 --no-far-target --train-steps 0 --feeder-steps 0 --knobs "${KNOBS:-}" > gpurun_out/pol_timeline_log.txt 2>&1
python3 - > gpurun_out/pol_epoch_timeline.txt <<'P'
import csv, glob
f = glob.glob("gpurun_out/prof_nodes/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void m2h::", "").replace("m2h::", "").replace("void at::native::", "at::")[:56] for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
q = [r.get("Queue_Id", "?") for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("ppo_loss")]
a, b = idx[-3], idx[-2]
t0 = en[a]
print("one update_pol epoch: %d kernels, wall %.1f us, summed kernel time %.1f us" % (b - a, (en[b] - t0) / 1e3, sum(en[i] - st[i] for i in range(a + 1, b + 1)) / 1e3))
busy = 0
last = t0
for i in range(a + 1, b + 1):
    s, e = max(st[i], last), en[i]
    if e > s:
        busy += e - s
        last = e
print("time with at least one kernel running: %.1f us" % (busy / 1e3))
for i in range(a + 1, b + 1):
    print("  +%8.1f  %7.1f us  q%-3s %s" % ((st[i] - t0) / 1e3, (en[i] - st[i]) / 1e3, q[i], names[i]))
P
rm -rf gpurun_out/prof_nodes
head -3 gpurun_out/pol_epoch_timeline.txt
