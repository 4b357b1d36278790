# DD-PPO cycle with the update batches' three encoders on parallel graph branches under runtime switches: bash tools/dd_fork_env.sh "ENV=.. ENV=.." ...
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  env $cfg python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 2 --no-far-target --train-steps 0 --feeder-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['ddppo']; print('$cfg', d['value'], d['phases']['rollout_ms'], d['phases']['update_pol_ms'], d['phases']['update_sep_ms'])"
done
