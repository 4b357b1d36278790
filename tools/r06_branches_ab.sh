# update_pol's three encoder branches inside the epoch's HIP graph (M2H_PARALLEL_BRANCHES, m2h/graphs.py) on the round-6 tree: A/B on one box
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --no-far-target --train-steps 0 --feeder-steps 0 --no-clock-probe --ddppo-cycles 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['ddppo']
print(sys.argv[1:], d['value'], {k:d['phases'][k] for k in ('rollout_ms','update_pol_ms','update_sep_ms')})" "$@"; }
run M2H_PARALLEL_BRANCHES=1
run M2H_PARALLEL_BRANCHES=0
run M2H_PARALLEL_BRANCHES=1
run M2H_PARALLEL_BRANCHES=0
