# DD-PPO cycle and passive training step under a knob setting against the default, interleaved on one box: bash tools/r05_knob_dd_ab.sh "38=192"
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for k in "" "$1"; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --feeder-steps 0 --train-steps 20 --no-far-target --ddppo-cycles 3 --knobs "$k" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['ddppo']['phases']
print('knobs [$k]', d['ddppo']['value'], p['rollout_ms'], p['update_pol_ms'], p['update_sep_ms'], 'train', d['passive_train']['ms_per_step'])"
done; done
