cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_ptrain
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ptrain -o pt --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --ddppo-cycles 0 --feeder-steps 0 --train-steps 20 > gpurun_out/prof_ptrain_log.txt 2>&1
rm -f gpurun_out/prof_ptrain/*kernel_trace.csv
python3 tools/kstats.py gpurun_out/prof_ptrain/pt_kernel_stats.csv 30
