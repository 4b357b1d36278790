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
# one update_sep epoch = the kernels between two consecutive l1_loss launches near the end of the run (the last update_sep)
idx = [i for i, n in enumerate(names) if (n.startswith("l1_nhwc16") or n.startswith("l1_loss_kernel"))]
a, b = idx[-3], idx[-2]
print("kernels between two l1_loss launches (one update_sep epoch): %d, wall %.1f us, kernel time %.1f us" % (b - a, (en[b] - en[a]) / 1e3, sum(en[i] - st[i] for i in range(a + 1, b + 1)) / 1e3))
for i in range(a + 1, b + 1):
    print("  %-62s %8.1f us" % (names[i], (en[i] - st[i]) / 1e3))
P
rm -rf gpurun_out/prof_nodes
