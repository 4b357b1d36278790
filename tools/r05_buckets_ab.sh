# what the N > 1 gradient schedule costs at one rank (no collective): bash tools/r05_buckets_ab.sh
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for f in "" "--force-schedule overlap" "--force-schedule buckets" "--force-buckets"; do
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-mode --ddppo-cycles 2 --no-far-target --train-steps 0 --feeder-steps 0 $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['ddppo']; print('%-28s' % ('$f' or 'flat'), d['value'], d['phases']['rollout_ms'], d['phases']['update_pol_ms'], d['phases']['update_sep_ms'])"
done; done
