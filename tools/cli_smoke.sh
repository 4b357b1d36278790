# main.py end to end on the GPU box: ppo train -> checkpoint -> eval; two-policy switch eval; passive train (tiny schedules).
cd $GRAFT_REPO_ROOT
D=/tmp/m2h_cli; rm -rf $D; mkdir -p $D/configs/tasks/nearTarget
printf 'ENVIRONMENT:\n  MAX_EPISODE_STEPS: 5\n' > $D/configs/tasks/nearTarget/train_nearTarget.yaml
cat > $D/near.yaml <<'Y'
BASE_TASK_CONFIG_PATH: "configs/tasks/nearTarget/train_nearTarget.yaml"
NUM_PROCESSES: 4
EXTRA_DEPTH: True
TRAINER_NAME: "ppo"
RL:
  PPO:
    num_steps: 5
    num_updates_per_cycle: 2
    ppo_epoch: 2
Y
cat > $D/far_eval.yaml <<'Y'
BASE_TASK_CONFIG_PATH: "configs/tasks/nearTarget/train_nearTarget.yaml"
NUM_PROCESSES: 4
EXTRA_DEPTH: True
TRAINER_NAME: "ppo"
RL:
  PPO:
    switch_policy: True
    time_thres_for_pol_switch: 2
    deterministic_eval: True
Y
cat > $D/passive.yaml <<'Y'
TRAINER_NAME: "passive"
NUM_EPOCHS: 2
Pretrain:
  Passive:
    BATCH_SIZE: 4
    BATCHES_PER_EPOCH: 3
    VAL_BATCHES: 1
Y
set -e
python main.py --exp-config $D/near.yaml --run-type train --model-dir $D/run_near --cycles 2
python main.py --exp-config $D/near.yaml --run-type eval --model-dir $D/run_near --eval-episodes 4 | tail -n 6
python - <<P
import sys; sys.path.insert(0, "move2hear-active-av-separation_amd")
from m2h.rl.ppo.ppo_trainer import PPOTrainer
PPOTrainer.save_switch_checkpoint("$D/ckpt_polSwitch.pth", "$D/run_near/data/ckpt.0.pth", "$D/run_near/data/ckpt.0.pth")
P
python main.py --exp-config $D/far_eval.yaml --run-type eval --model-dir $D/run_far --eval-ckpt $D/ckpt_polSwitch.pth --eval-episodes 4 | tail -n 4
python main.py --exp-config $D/passive.yaml --run-type train --model-dir $D/run_passive --cycles 2
echo CLI_SMOKE_OK
