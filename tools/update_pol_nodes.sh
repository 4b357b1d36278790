cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_nodes
rocprofv3 --kernel-trace -d gpurun_out/prof_nodes -o g --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --ddppo-cycles 1 --no-far-target --train-steps 0 --feeder-steps 0 > gpurun_out/prof_nodes_log.txt 2>&1
python3 - <<'P'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_nodes/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void m2h::", "").replace("m2h::", "").replace("void at::native::", "at::")[:60] for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
# one replayed update_pol epoch = the kernels between two consecutive ppo_loss kernels (forward tail + backward + step + next forward)
idx = [i for i, n in enumerate(names) if n.startswith("ppo_loss")]
a, b = idx[-3], idx[-2]
print("kernels between two ppo_loss launches (one epoch): %d, wall %.1f us, kernel time %.1f us" % (b - a, (en[b] - en[a]) / 1e3, sum(en[i] - st[i] for i in range(a + 1, b + 1)) / 1e3))
agg = collections.OrderedDict()
for i in range(a + 1, b + 1):
    c = agg.setdefault(names[i], [0, 0.0])
    c[0] += 1
    c[1] += (en[i] - st[i]) / 1e3
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-62s x%4d  %8.1f us" % (n, c, t))
P
rm -rf gpurun_out/prof_nodes
