# DD-PPO cycle under a tuning knob: bash tools/dd_knob_ab.sh KNOB "values"
cd $GRAFT_REPO_ROOT
K=${1:-24}; VALS=${2:-"0 2048 4096"}
for v in $VALS; do
  python tools/knob_bench.py $K $v --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 2 --no-far-target --train-steps 0 --feeder-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['ddppo']; print('knob$K=$v', d['value'], d['phases']['rollout_ms'], d['phases']['update_pol_ms'], d['phases']['update_sep_ms'])"
done
