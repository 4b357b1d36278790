cd $GRAFT_REPO_ROOT
for r in 1 2; do for br in 1 0; do
M2H_PARALLEL_BRANCHES=$br python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --feeder-steps 0 --train-steps 0 --no-far-target --ddppo-cycles 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['ddppo']['phases']
print('branches $br', d['ddppo']['value'], p['rollout_ms'], p['update_pol_ms'], p['update_sep_ms'])"
done; done
