# every kernel of ONE update_sep call (from the first kernel after the previous call's last Adam to this call's last Adam), epochs included:
# what surrounds the four epochs?  usage: gpurun -- 'bash tools/r05_sep_phase_nodes.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_nodes
rocprofv3 --kernel-trace -d gpurun_out/prof_nodes -o g --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --ddppo-cycles 1 --no-far-target --train-steps 0 --feeder-steps 0 > gpurun_out/prof_nodes_log.txt 2>&1
python3 - > gpurun_out/r05_sep_phase_nodes.txt <<'P'
import csv, glob
f = glob.glob("gpurun_out/prof_nodes/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void m2h::", "").replace("m2h::", "").replace("void at::native::", "at::")[:70] for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
l1 = [i for i, n in enumerate(names) if n.startswith("l1_nhwc16")]
# the last cycle's update_sep calls: groups of 4 consecutive l1 launches (epochs); take the 5th call of the last 6
calls = [l1[i:i + 4] for i in range(0, len(l1), 4)]
c_prev, c = calls[-3], calls[-2]
a, b = c_prev[-1], c[-1]
print("one update_sep call: kernels %d, wall %.1f us (from the previous call's last loss kernel to this call's)" % (b - a, (en[b] - en[a]) / 1e3))
ep = 0
for i in range(a + 1, b + 1):
    print("  +%8.1f  %7.1f us  %s" % ((st[i] - en[a]) / 1e3, (en[i] - st[i]) / 1e3, names[i]))
P
rm -rf gpurun_out/prof_nodes
head -2 gpurun_out/r05_sep_phase_nodes.txt
