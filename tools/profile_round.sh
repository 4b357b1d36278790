cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof_r1c gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/prof_ddppo3
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1c -o bench --output-format csv -- python3 bench.py --ddppo-cycles 0 --train-steps 0 --feeder-steps 0 --cpu-seconds 5 > gpurun_out/prof_r1c_log.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --ddppo-cycles 0 --train-steps 0 --feeder-steps 0 --no-kernel-timing --no-graph > gpurun_out/pmc_fetch_log.txt 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --ddppo-cycles 0 --train-steps 0 --feeder-steps 0 --no-kernel-timing --no-graph > gpurun_out/pmc_write_log.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ddppo3 -o dd --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --ddppo-cycles 1 --no-far-target --train-steps 0 --feeder-steps 0 > gpurun_out/prof_ddppo3_log.txt 2>&1
python3 tools/pmc_summary.py $(ls gpurun_out/pmc_fetch/*counter_collection.csv | head -1) > gpurun_out/pmc_fetch_summary.txt 2>&1
python3 tools/pmc_summary.py $(ls gpurun_out/pmc_write/*counter_collection.csv | head -1) > gpurun_out/pmc_write_summary.txt 2>&1
rm -f gpurun_out/pmc_fetch/*counter_collection.csv gpurun_out/pmc_write/*counter_collection.csv gpurun_out/*/*kernel_trace.csv
python3 tools/kstats.py gpurun_out/prof_r1c/bench_kernel_stats.csv 12
python3 tools/kstats.py gpurun_out/prof_ddppo3/dd_kernel_stats.csv 12
ls -la gpurun_out/prof_r1c gpurun_out/pmc_fetch gpurun_out/prof_ddppo3
tail -2 gpurun_out/prof_r1c_log.txt | cut -c1-300
