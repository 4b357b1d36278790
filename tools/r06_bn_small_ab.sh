# train-mode BatchNorm as one launch per direction (csrc/bn.hip, knob 37 = -1: the three-launch path): passive training step, A/B on one box
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --no-far-target --train-steps 30 --feeder-steps 0 --no-clock-probe --ddppo-cycles 0 "$@" 2>/dev/null | python -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])['passive_train']
print(sys.argv[1:], p['value'], p['ms_per_step'], p['last_losses'])" "$@"; }
run --knobs 37=-1 && run --knobs 37=256 && run --knobs 37=1024 && run --knobs 37=4096 && run --knobs 37=-1 && run --knobs 37=1024
