# DD-PPO cycle and passive training step under different numbers of HIP hardware queues (GPU_MAX_HW_QUEUES; the runtime's default is 4)
cd $GRAFT_REPO_ROOT
for q in 4 2 3; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 2 --no-far-target --train-steps 30 --feeder-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); x=d['ddppo']; print('hw_queues=$q ddppo', x['value'], x['phases']['rollout_ms'], x['phases']['update_pol_ms'], x['phases']['update_sep_ms'], ' passive_train', d['passive_train']['value'], d['passive_train']['ms_per_step'])"
done
