# weight gradients as side branches of the captured training step (M2H_WGRAD_SIDE, m2h/functional.py wgrad_side_branches): A/B on one box
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --no-far-target --train-steps ${TRAIN:-30} --feeder-steps 0 --no-clock-probe --ddppo-cycles ${CYCLES:-0} 2>gpurun_out/r06_wgrad_side_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d['passive_train']; dd=d.get('ddppo')
print(sys.argv[1:], 'ptrain', p and (p['value'], p['ms_per_step'], p['last_losses']), 'ddppo', dd and (dd['value'], {k:dd['phases'][k] for k in ('rollout_ms','update_pol_ms','update_sep_ms')}))" "$@"; }
run M2H_WGRAD_SIDE=1 && run M2H_WGRAD_SIDE=0 && run M2H_WGRAD_SIDE=1 && run M2H_WGRAD_SIDE=0
