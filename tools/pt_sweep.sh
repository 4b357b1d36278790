# passive training step under a tuning knob: bash tools/pt_sweep.sh KNOB "values"
cd $GRAFT_REPO_ROOT
K=${1:-24}; VALS=${2:-"0 -1 64 256 512"}
for v in $VALS; do
  python tools/knob_bench.py $K $v --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 0 --feeder-steps 0 --train-steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('knob$K=$v', d['passive_train']['value'], d['passive_train']['ms_per_step'])"
done
