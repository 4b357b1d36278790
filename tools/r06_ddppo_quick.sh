# the DD-PPO leg alone, twice (env-steps/s and phase times per cycle)
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --no-far-target --train-steps 0 --feeder-steps 0 --no-clock-probe --ddppo-cycles 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['ddppo']
print(sys.argv[1:], d['value'], {k:d['phases'][k] for k in ('rollout_ms','update_pol_ms','update_sep_ms')}, d.get('launches_per_cycle') or d.get('phase_launches'))" "$@"; }
run A=1 && run A=2
