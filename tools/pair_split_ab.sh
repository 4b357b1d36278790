# headline pair graph: the batch's two halves as two graph branches (M2H_PAIR_SPLIT=1, experiment) vs one chain
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  M2H_PAIR_SPLIT=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 0 --train-steps 0 --feeder-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pair_split=$v', d['value'], d['ms_per_step'])"
done
