cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python3 tools/wgrad_bench.py
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf gpurun_out/pmcw; rocprofv3 --pmc $grp -d gpurun_out/pmcw -o w --output-format csv -- python3 tools/wgrad_bench.py --only amem.conv0 --reps 3 > /dev/null 2>&1
  python3 tools/pmc_summary.py $(ls gpurun_out/pmcw/*counter_collection.csv | head -1) 2>&1 | grep -A10 "wgrad_kernel" | head -11
done
rm -rf gpurun_out/pmcw
