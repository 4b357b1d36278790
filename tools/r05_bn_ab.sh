# passive training step with train-mode BatchNorm in one launch per direction where the rows fit registers (default) against three (knob 25 = -1),
# interleaved on one box; as two graph branches (the bench's form) and as one chain (M2H_PARALLEL_BRANCHES=0)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for br in 1 0; do for k in "" "25=-1"; do
  M2H_PARALLEL_BRANCHES=$br python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 0 --feeder-steps 0 --train-steps 30 --knobs "$k" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('branches $br knobs [$k]', 'train', d['passive_train']['value'], d['passive_train']['ms_per_step'], d['passive_train']['last_losses'])"
done; done; done
