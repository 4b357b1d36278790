# DD-PPO cycle with the rollout step's visual encoder forked onto a side stream inside the step's HIP graph (M2H_PARALLEL_BRANCHES=1) vs the plain chain
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  M2H_PARALLEL_BRANCHES=$v python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 2 --no-far-target --train-steps 0 --feeder-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['ddppo']; print('fork=$v', d['value'], d['phases']['rollout_ms'], d['phases']['update_pol_ms'], d['phases']['update_sep_ms'])"
done
