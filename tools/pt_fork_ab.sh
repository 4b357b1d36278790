# passive training step with the two U-Nets' chains as parallel graph branches (M2H_PARALLEL_BRANCHES=1) vs one chain
cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  M2H_PARALLEL_BRANCHES=$v python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 0 --feeder-steps 0 --train-steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fork=$v', d['passive_train']['value'], d['passive_train']['ms_per_step'])"
done
