# SQ counters of the weight-gradient kernels inside the passive training step (kernel by kernel: --no-graph is not a flag of that leg; the
# trainer's first batch runs eagerly, the graph replays carry the rest): how busy is the matrix pipe?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06w; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 0 --feeder-steps 0 --train-steps 4 --no-clock-probe > $O/line.json 2> $O/err.txt
python3 tools/pmc_summary.py $(ls $O/pmc/*counter_collection.csv | head -1) > $O/pmc_all.txt 2>&1
grep -A9 "wgrad_kernel\|igemm_f32_kernel\|skinny_gather" $O/pmc_all.txt > $O/pmc_wgrad.txt
rm -rf $O/pmc
head -80 $O/pmc_wgrad.txt
