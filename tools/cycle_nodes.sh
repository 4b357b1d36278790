# one kernel trace of a DD-PPO cycle; prints, in launch order, the kernels of one replayed rollout step, one update_pol epoch
# and one update_sep epoch (name, duration, gap to the previous kernel).  usage: gpurun -- 'bash tools/cycle_nodes.sh [tag]'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-nodes}
rm -rf gpurun_out/prof_nodes
rocprofv3 --kernel-trace -d gpurun_out/prof_nodes -o g --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --ddppo-cycles 1 --no-far-target --train-steps 0 --feeder-steps 0 > gpurun_out/${TAG}_log.txt 2>&1
python3 - > gpurun_out/${TAG}.txt <<'P'
import csv, glob
f = glob.glob("gpurun_out/prof_nodes/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void m2h::", "").replace("m2h::", "").replace("void at::native::", "at::")[:70] for r in rows]
grid = [(r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")) for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
def show(title, a, b):
    print("%s: %d kernels, wall %.1f us, kernel time %.1f us" % (title, b - a, (en[b] - en[a]) / 1e3, sum(en[i] - st[i] for i in range(a + 1, b + 1)) / 1e3))
    for i in range(a + 1, b + 1):
        print("  %-72s %7.1f us  gap %5.1f  grid %s/%s" % (names[i], (en[i] - st[i]) / 1e3, (st[i] - en[i - 1]) / 1e3, grid[i][0], grid[i][1]))
idx = [i for i, n in enumerate(names) if n.startswith("step_index_advance")]
show("one replayed rollout step", idx[len(idx) // 2], idx[len(idx) // 2 + 1])
# gaps inside replayed steps, over all of them: how much of a step is the device waiting between two kernels of the chain?
import collections
walls, gaps, where = [], [], collections.Counter()
for a, b in zip(idx[:-1], idx[1:]):
    if b - a > 80 or b - a < 30:
        continue   # an update phase lies in between / not a whole step
    g = [(st[i] - en[i - 1]) / 1e3 for i in range(a + 2, b + 1)]   # (the first gap of a step is the graph launch itself)
    walls.append((en[b] - en[a]) / 1e3)
    gaps.append(sum(x for x in g if x > 1.0))
    for k, x in enumerate(g):
        if x > 5.0:
            where[names[a + 2 + k][:40]] += 1
if walls:
    import statistics as S
    print("all %d replayed steps: wall median %.1f mean %.1f p90 %.1f us; in-chain gaps > 1 us per step: median %.1f mean %.1f p90 %.1f us" % (
        len(walls), S.median(walls), S.mean(walls), sorted(walls)[int(0.9 * len(walls))], S.median(gaps), S.mean(gaps), sorted(gaps)[int(0.9 * len(gaps))]))
    print("kernels most often preceded by a gap > 5 us:", where.most_common(8))
idx = [i for i, n in enumerate(names) if n.startswith("ppo_loss")]
show("one update_pol epoch (between two ppo_loss launches)", idx[-3], idx[-2])
idx = [i for i, n in enumerate(names) if (n.startswith("l1_nhwc16") or n.startswith("l1_loss_kernel"))]
show("one update_sep epoch (between two l1_loss launches)", idx[-3], idx[-2])
P
rm -rf gpurun_out/prof_nodes
tail -3 gpurun_out/${TAG}.txt
