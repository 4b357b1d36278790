# Round-3 probe: what one replayed rollout step and the passive training step (fp32 / bf16x3 GEMMs) are made of.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3p; rm -rf $O; mkdir -p $O
bash tools/rollout_nodes.sh > $O/rollout_nodes.txt 2>&1
for M in fp32 bf16x3; do
  rocprofv3 --kernel-trace --stats -d $O/pt_$M -o pt --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 0 --feeder-steps 0 --train-steps 20 --train-math $M > $O/ptrain_$M.json 2> $O/ptrain_$M.err
  python3 tools/kstats.py $O/pt_$M/pt_kernel_stats.csv 30 > $O/ptrain_${M}_kstats.txt
  rm -rf $O/pt_$M
done
head -70 $O/rollout_nodes.txt
