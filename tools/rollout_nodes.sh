cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_nodes
rocprofv3 --kernel-trace -d gpurun_out/prof_nodes -o g --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --ddppo-cycles 1 --no-far-target --train-steps 0 --feeder-steps 0 > gpurun_out/prof_nodes_log.txt 2>&1
python3 - <<'P'
import csv, glob
f = glob.glob("gpurun_out/prof_nodes/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void m2h::", "").replace("m2h::", "").replace("void at::native::", "at::")[:70] for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
# one replayed rollout step = the kernels between two consecutive step_index_advance kernels
idx = [i for i, n in enumerate(names) if n.startswith("step_index_advance")]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
print("kernels in one replayed rollout step: %d, wall %.1f us, kernel time %.1f us" % (b - a, (en[b] - en[a]) / 1e3, sum(en[i] - st[i] for i in range(a + 1, b + 1)) / 1e3))
for i in range(a + 1, b + 1):
    print("  %-72s %7.1f us  gap %5.1f" % (names[i], (en[i] - st[i]) / 1e3, (st[i] - en[i - 1]) / 1e3))
P
rm -rf gpurun_out/prof_nodes
