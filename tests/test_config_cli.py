"""CPU: configuration merge order (defaults -> experiment YAML -> task YAML -> KEY VALUE opts), trainer registry names, and
the checkpoint key format shared with the reference."""
import os

import torch

from m2h import synthetic
from m2h.config.default import get_config, get_trainer


def test_yaml_merge_and_opts(tmp_path):
    exp = tmp_path / "near.yaml"
    task = tmp_path / "configs" / "tasks" / "nearTarget"
    task.mkdir(parents=True)
    (task / "train_nearTarget.yaml").write_text("ENVIRONMENT:\n  MAX_EPISODE_STEPS: 20\n")
    exp.write_text('BASE_TASK_CONFIG_PATH: "configs/tasks/nearTarget/train_nearTarget.yaml"\nNUM_PROCESSES: 14\nEXTRA_DEPTH: True\n'
                   "NUM_UPDATES: 16786\nRL:\n  PPO:\n    num_updates_per_cycle: 6\n    entropy_coef: 0.20\n    lr_pol: 1.0e-4\n    ppo_epoch: 4\n")
    cfg = get_config(str(exp), ["RL.PPO.ppo_epoch", "2", "SEED", "7"], str(tmp_path / "run"), "train", search_dirs=(str(tmp_path),))
    assert cfg.TRAINER_NAME == "ppo" and cfg.NUM_PROCESSES == 14 and cfg.EXTRA_DEPTH is True and cfg.NUM_UPDATES == 16786
    assert cfg.ppo_epoch == 2 and cfg.entropy_coef == 0.20 and cfg.lr_pol == 1.0e-4 and cfg.SEED == 7
    assert cfg.MAX_EPISODE_STEPS == 20 and cfg.CHECKPOINT_FOLDER == os.path.join(str(tmp_path / "run"), "data")
    passive = tmp_path / "passive.yaml"
    passive.write_text('TRAINER_NAME: "passive"\nNUM_EPOCHS: 1000\nPretrain:\n  Passive:\n    lr: 5.0e-4\n    eps: 1.0e-5\n    max_grad_norm: 0.8\n')
    pc = get_config(str(passive), None, None)
    assert pc.TRAINER_NAME == "passive" and pc.lr == 5.0e-4 and pc.NUM_EPOCHS == 1000 and pc.max_grad_norm == 0.8


def test_trainer_registry_names():
    assert get_trainer("passive").__name__ == "PassiveTrainer" and get_trainer("ppo").__name__ == "PPOTrainer"
    assert get_trainer("nope") is None


def test_checkpoint_keys_are_the_reference_ones():
    """agent.state_dict() keys are rooted at ``actor_critic.`` (ppo.py:59-70 reads them that way)."""
    from m2h.common.spaces import Discrete, move2hear_observation_space
    from m2h.pretrain.passive.passive import Passive
    from m2h.pretrain.passive.policy import Move2HearPassiveWoMemoryPolicy
    from m2h.rl.ppo.policy import Move2HearPolicy
    pas = Passive(Move2HearPassiveWoMemoryPolicy(move2hear_observation_space()))
    keys = list(pas.state_dict().keys())
    assert keys == ["actor_critic." + k for k in synthetic.passive_shapes()]
    rl = Move2HearPolicy(move2hear_observation_space(), Discrete(3), "spectrogram", 512, False, True, use_ddppo=True)
    assert list(rl.state_dict().keys()) == list(synthetic.policy_shapes().keys())
    # a passive checkpoint's separator entries are exactly what load_pretrained_passive_separators looks up
    for k in pas.state_dict():
        assert k[len("actor_critic."):] in rl.state_dict()
