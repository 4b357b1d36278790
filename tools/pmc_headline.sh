cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf gpurun_out/pmcx; rocprofv3 --pmc $grp -d gpurun_out/pmcx -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --ddppo-cycles 0 --train-steps 0 --no-kernel-timing > /dev/null 2>&1
  python3 tools/pmc_summary.py $(ls gpurun_out/pmcx/*counter_collection.csv | head -1) 2>&1 | grep -A9 "convT_tap_kernel<32, 32, 1>\|igemm_f32_kernel<128, 128, 2, 2, 2, 32, 1, 2>" | head -24
done
rm -rf gpurun_out/pmcx
