cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_ddppo4
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ddppo4 -o dd --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --ddppo-cycles 1 --no-far-target --train-steps 0 --feeder-steps 0 > gpurun_out/prof_ddppo4_log.txt 2>&1
rm -f gpurun_out/prof_ddppo4/*kernel_trace.csv
python3 tools/kstats.py gpurun_out/prof_ddppo4/dd_kernel_stats.csv 32
