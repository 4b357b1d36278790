# Kernel stats of the DD-PPO loop alone (one gpurun call): rocprofv3 --kernel-trace --stats over three near-target cycles.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/ddp; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/dd -o dd --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 2 --no-far-target --train-steps 0 --feeder-steps 0 > $O/line.json 2> $O/err.log
rm -f $O/dd/*kernel_trace.csv
python3 tools/kstats.py $O/dd/dd_kernel_stats.csv ${1:-40}
