# Round 5: the persistent shared-patch engine -- parity, then interleaved A/B of knob-36 variants (and optionally the round-4 kernel, build/libm2h_old.so).
# usage (GPU box): bash tools/r05_patch_ab.sh "36=9;auto" [old]
cd $GRAFT_REPO_ROOT
V=${1:-auto}
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_patch.py tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/r05_patch_tests.txt 2>&1 || { tail -30 gpurun_out/r05_patch_tests.txt; exit 1; }
tail -3 gpurun_out/r05_patch_tests.txt
timeout -k 10 300 python tools/patch_fuzz.py > gpurun_out/r05_patch_fuzz.txt 2>&1 || { tail -30 gpurun_out/r05_patch_fuzz.txt; exit 1; }
tail -2 gpurun_out/r05_patch_fuzz.txt
for r in 1 2; do
  if [ "$2" == "old" ]; then echo "== old"; M2H_LIB=$GRAFT_REPO_ROOT/build/libm2h_old.so timeout -k 10 200 python tools/pair_ab.py --variants auto --rounds 6 --steps 10 2>&1 | grep -v amdgpu | tail -3; fi
  echo "== new"; timeout -k 10 200 python tools/pair_ab.py --variants "$V" --rounds 6 --steps 10 2>&1 | grep -v amdgpu | tail -4
done
timeout -k 10 200 python tools/pair_ab.py --variants "$V" --rounds 2 --steps 5 --layers > gpurun_out/r05_patch_layers.txt 2>&1; tail -16 gpurun_out/r05_patch_layers.txt
