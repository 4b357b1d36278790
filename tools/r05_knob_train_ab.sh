cd $GRAFT_REPO_ROOT
for r in 1 2; do for k in "" "24=4096" "24=16384"; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --feeder-steps 0 --train-steps 30 --ddppo-cycles 0 --knobs "$k" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('knobs [$k] train', d['passive_train']['ms_per_step'], d['passive_train']['value'])"
done; done
