# Kernel stats of the passive training step alone (one gpurun call): rocprofv3 --kernel-trace --stats over 20 timed steps.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/ptrain; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/pt -o pt --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 0 --feeder-steps 0 --train-steps 20 > $O/line.json 2> $O/err.log
rm -f $O/pt/*kernel_trace.csv
python3 tools/kstats.py $O/pt/pt_kernel_stats.csv 40
