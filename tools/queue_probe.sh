cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_q
rocprofv3 --kernel-trace -d gpurun_out/prof_q -o g --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --ddppo-cycles 1 --no-far-target --train-steps 0 --feeder-steps 0 > gpurun_out/queue_probe_log.txt 2>&1
python3 - > gpurun_out/queue_probe.txt <<'P'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_q/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void m2h::", "").replace("m2h::", "")[:50] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("step_index_advance")]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
for i in range(a + 1, b + 1):
    r = rows[i]
    print("%-52s q=%s stream=%s dur=%.1f lds=%s scratch=%s vgpr=%s sgpr=%s" % (names[i], r.get("Queue_Id"), r.get("Stream_Id"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
          r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("VGPR_Count"), r.get("SGPR_Count")))
print(collections.Counter(r.get("Queue_Id") for r in rows))
P
rm -rf gpurun_out/prof_q
head -3 gpurun_out/queue_probe.txt | cut -c1-600; tail -2 gpurun_out/queue_probe.txt
