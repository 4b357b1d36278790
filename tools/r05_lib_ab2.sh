cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in "$@"; do
  if [ "$v" == "tree" ]; then L=""; else L=$GRAFT_REPO_ROOT/build/libm2h_$v.so; fi
  echo "== $v $(M2H_LIB=$L timeout -k 10 200 python tools/pair_ab.py --variants auto --rounds 6 --steps 10 2>&1 | grep -v amdgpu | tail -1)"
done; done
