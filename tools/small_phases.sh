# phase costs of the small-batch conv engine by elimination: the separator pair at 14 envs under diagnostic builds of
# csrc/conv_small.hip (M2H_SMALL_DBG bits: 1 no staging, 2 no MFMA loop, 4 no output, 8 no weight loads).
# build here:  for v in 0 1 2 4 8 15; do bash tools/build_variant.sh sdbg$v conv_small.hip -DM2H_SMALL_DBG=$v; done
# run:         gpurun -- 'bash tools/small_phases.sh'
cd $GRAFT_REPO_ROOT
for v in 0 1 2 4 8 3 15; do
  echo "== M2H_SMALL_DBG=$v"
  M2H_LIB=$GRAFT_REPO_ROOT/build/libm2h_sdbg$v.so python tools/small_tune.py 14 0 2>&1 | grep -E "start:|per-stage"
done
