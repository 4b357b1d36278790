# DD-PPO cycle under library builds, interleaved on one box: bash tools/r05_dd_lib_ab.sh name1 name2 ... (build/libm2h_NAME.so; "tree" = in-tree)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in "$@"; do
  if [ "$v" == "tree" ]; then L=""; else L=$GRAFT_REPO_ROOT/build/libm2h_$v.so; fi
  M2H_LIB=$L python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 2 --no-far-target --train-steps 10 --feeder-steps 0 2>/dev/null | python -c "
import json,sys; x=json.loads(sys.stdin.read()); d=x['ddppo']; print('%-6s' % '$v', d['value'], d['phases']['rollout_ms'], d['phases']['update_pol_ms'], d['phases']['update_sep_ms'], 'ptrain', x['passive_train']['value'])"
done; done
