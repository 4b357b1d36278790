# Round-2 evidence: kernel stats of the default bench command, of a DD-PPO cycle and of the passive training step; HBM traffic
# counters (separate --pmc passes, no trace domains); SQ / TCC counters of the dominant headline instantiation.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/bench -o bench --output-format csv -- python3 bench.py --ddppo-cycles 0 --train-steps 0 --feeder-steps 0 --cpu-seconds 5 > $O/bench_line_under_rocprof.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/ddppo -o dd --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --ddppo-cycles 2 --no-far-target --train-steps 0 --feeder-steps 0 > $O/ddppo_log.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/ptrain -o pt --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --ddppo-cycles 0 --feeder-steps 0 --train-steps 20 > $O/ptrain_log.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/feeder -o fd --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --ddppo-cycles 0 --train-steps 0 --feeder-steps 20 > $O/feeder_log.txt 2>&1
rm -f $O/*/*kernel_trace.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $O/pmc_$C -o c --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --ddppo-cycles 0 --train-steps 0 --feeder-steps 0 --no-kernel-timing --no-graph > $O/pmc_$C.log 2>&1
  python3 tools/pmc_summary.py $(ls $O/pmc_$C/*counter_collection.csv | head -1) > $O/pmc_$C.txt 2>&1
  rm -rf $O/pmc_$C
done
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf $O/pmcx; rocprofv3 --pmc $grp -d $O/pmcx -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --ddppo-cycles 0 --train-steps 0 --feeder-steps 0 --no-kernel-timing --no-graph > /dev/null 2>&1
  python3 tools/pmc_summary.py $(ls $O/pmcx/*counter_collection.csv | head -1) 2>&1 | grep -A10 "igemm_dma_kernel<256, 128\|igemm_f32_kernel<128, 128, 2, 2, 2, 32, 1, 2>\|convT_tap_kernel<32, 32, 1, 256>" | head -40 >> $O/pmc_sq_tcc_bf16x3.txt
done
rm -rf $O/pmcx
python3 tools/kstats.py $O/bench/bench_kernel_stats.csv 14
python3 tools/kstats.py $O/ddppo/dd_kernel_stats.csv 14
tail -c 600 $O/bench_line_under_rocprof.json
