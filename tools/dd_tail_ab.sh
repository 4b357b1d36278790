# DD-PPO cycle with the six update_sep beside the last update_pol (--tail-overlap) vs after it (default)
cd $GRAFT_REPO_ROOT
for f in "--tail-overlap" "" "--tail-overlap" ""; do
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 3 --no-far-target --train-steps 0 --feeder-steps 0 $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['ddppo']; print('tail overlap %s' % ('on ' if '$f' else 'off'), d['value'], 1e3*d['s_per_cycle'], d['phases']['rollout_ms'], d['phases']['update_pol_ms'], d['phases']['update_sep_ms'], d['last_sep_losses'])"
done
