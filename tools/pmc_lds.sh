cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3p
rm -rf $O/pmcx; rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $O/pmcx -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-mode --ddppo-cycles 0 --train-steps 0 --feeder-steps 0 --no-kernel-timing --no-graph > /dev/null 2>&1
python3 tools/pmc_summary.py $(ls $O/pmcx/*counter_collection.csv | head -1) 2>&1 | grep -A8 "igemm_patch_kernel" > $O/pmc_lds.txt
rm -rf $O/pmcx
cat $O/pmc_lds.txt
