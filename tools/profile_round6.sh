# Round-6 evidence (one gpurun call): kernel stats of the headline command in the headline arithmetic ONLY (--no-other-mode), of a
# DD-PPO cycle, of the passive training step and of the feeder; HBM traffic counters (separate --pmc passes, no trace domains) and
# SQ / TCC counters for every kernel of the headline step.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; rm -rf $O; mkdir -p $O
HEAD="--no-other-mode --ddppo-cycles 0 --train-steps 0 --feeder-steps 0 --no-clock-probe"
rocprofv3 --kernel-trace --stats -d $O/bench -o bench --output-format csv -- python3 bench.py $HEAD --cpu-seconds 5 > $O/bench_line_under_rocprof.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/ddppo -o dd --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 2 --no-far-target --train-steps 0 --feeder-steps 0 > $O/ddppo_line.json 2> $O/ddppo.err
rocprofv3 --kernel-trace --stats -d $O/ptrain -o pt --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 0 --feeder-steps 0 --train-steps 20 > $O/ptrain_line.json 2> $O/ptrain.err
rocprofv3 --kernel-trace --stats -d $O/feeder -o fd --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 0 --train-steps 0 --feeder-steps 20 > $O/feeder_line.json 2> $O/feeder.err
rm -f $O/*/*kernel_trace.csv
PM="--steps 2 --warmup 1 --no-cpu-baseline $HEAD --no-kernel-timing --no-graph"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $O/pmc_$C -o c --output-format csv -- python3 bench.py $PM > $O/pmc_$C.log 2>&1
  python3 tools/pmc_summary.py $(ls $O/pmc_$C/*counter_collection.csv | head -1) > $O/pmc_$C.txt 2>&1
  rm -rf $O/pmc_$C
done
: > $O/pmc_sq_tcc.txt
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf $O/pmcx; rocprofv3 --pmc $grp -d $O/pmcx -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $HEAD --no-kernel-timing --no-graph > /dev/null 2>&1
  python3 tools/pmc_summary.py $(ls $O/pmcx/*counter_collection.csv | head -1) 2>&1 | grep -A10 "strip_kernel\|igemm_patch_kernel\|igemm_dma_kernel\|convT_quad_kernel\|igemm_f32_kernel<128, 128" >> $O/pmc_sq_tcc.txt
done
rm -rf $O/pmcx
# update_sep's image-row kernels: HBM bytes per launch from FETCH_SIZE / WRITE_SIZE, one DD-PPO cycle, separate passes
: > $O/pmc_update_sep.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmcx; rocprofv3 --pmc $C -d $O/pmcx -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 1 --no-far-target --train-steps 0 --feeder-steps 0 --no-clock-probe > /dev/null 2>&1
  echo "== $C ==" >> $O/pmc_update_sep.txt
  python3 tools/pmc_summary.py $(ls $O/pmcx/*counter_collection.csv | head -1) 2>&1 | grep -A6 "conv3x3_row_bf16x3_kernel\|wgrad3x3_row\|l1_nhwc16_kernel\|conv_wgrad_reduce_torch" >> $O/pmc_update_sep.txt
done
rm -rf $O/pmcx
python3 tools/kstats.py $O/bench/bench_kernel_stats.csv 14
python3 tools/kstats.py $O/ddppo/dd_kernel_stats.csv 14
tail -c 400 $O/bench_line_under_rocprof.json
