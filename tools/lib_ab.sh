# A/B of library builds (tools/build_variant.sh) on ONE box: the headline pair under each build, interleaved, twice.
# usage (on the GPU box): bash tools/lib_ab.sh base ilv mask ...   (names of build/libm2h_NAME.so)
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "$@"; do echo "== $v"; M2H_LIB=$GRAFT_REPO_ROOT/build/libm2h_$v.so python tools/pair_ab.py --variants auto --rounds 6 --steps 10 2>&1 | grep -v amdgpu; done; done
