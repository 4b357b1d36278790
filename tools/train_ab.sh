# A/B of library builds (tools/build_variant.sh) on ONE box: the passive training leg and (optionally, DD=1) a DD-PPO cycle under each build, twice.
# usage (on the GPU box): [DD=1] bash tools/train_ab.sh u4 u8 ...   (names of build/libm2h_NAME.so)
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "$@"; do
  M2H_LIB=$GRAFT_REPO_ROOT/build/libm2h_$v.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles ${DD:-0} --no-far-target --feeder-steps 0 --train-steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'train', d['passive_train']['value'], d['passive_train']['ms_per_step'], 'ddppo', (d.get('ddppo') or {}).get('value'), ((d.get('ddppo') or {}).get('phases') or {}).get('rollout_ms'))"
done; done
