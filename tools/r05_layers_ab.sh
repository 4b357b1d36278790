# per-layer A/B of two library builds on one box: bash tools/r05_layers_ab.sh [variants for the new build]
cd $GRAFT_REPO_ROOT
V=${1:-auto}
for r in 1 2; do
echo "== old"; M2H_LIB=$GRAFT_REPO_ROOT/build/libm2h_old.so timeout -k 10 200 python tools/pair_ab.py --variants auto --rounds 4 --steps 10 --layers 2>&1 | grep -v amdgpu | tail -14
echo "== new"; timeout -k 10 200 python tools/pair_ab.py --variants "$V" --rounds 4 --steps 10 --layers 2>&1 | grep -v amdgpu | tail -16
done
