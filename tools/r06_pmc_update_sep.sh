# update_sep's image-row kernels: HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (KiB), one DD-PPO cycle, separate --pmc passes (the last block of tools/profile_round6.sh alone)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
: > $O/pmc_update_sep.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmcx; rocprofv3 --pmc $C -d $O/pmcx -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 1 --no-far-target --train-steps 0 --feeder-steps 0 --no-clock-probe > /dev/null 2>&1
  echo "== $C ==" >> $O/pmc_update_sep.txt
  python3 tools/pmc_summary.py $(ls $O/pmcx/*counter_collection.csv | head -1) 2>&1 | grep -A2 "conv3x3_row_bf16x3_kernel\|wgrad3x3_row\|l1_nhwc16_kernel\|conv_wgrad_reduce_torch" >> $O/pmc_update_sep.txt
done
rm -rf $O/pmcx
cat $O/pmc_update_sep.txt
