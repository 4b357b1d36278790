# Round-4 measurements that are not kernel statistics (one gpurun call): the per-node cost of replayed HIP graphs, the marginal node cost
# inside the real rollout-step graph, the small-batch engine against the tiled engines (tuned table, per-stage times), its in-kernel
# phase stamps, the DD-PPO cycle with that engine on / off, and the launch-order trace of one cycle.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04x; rm -rf $O; mkdir -p $O
python tools/graph_floor.py 2>&1 | grep -v amdgpu > $O/graph_floor.txt
python tools/node_cost.py 2>&1 | grep -v amdgpu > $O/node_cost.txt
python tools/small_tune.py 14 0 2>&1 | grep -v amdgpu > $O/small_engine_vs_tiled.txt
if [ -f build/libm2h_sstamp.so ]; then M2H_LIB=$GRAFT_REPO_ROOT/build/libm2h_sstamp.so python tools/small_stamps.py 2>&1 | grep -v amdgpu > $O/small_engine_stamps.txt; fi
bash tools/dd_small_ab.sh > $O/ddppo_small_engine_ab.txt 2>&1
bash tools/cycle_nodes.sh r04x/cycle_nodes > /dev/null 2>&1
tail -5 $O/node_cost.txt; tail -3 $O/small_engine_vs_tiled.txt
