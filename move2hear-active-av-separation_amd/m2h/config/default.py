"""Experiment configuration: the keys of audio_separation/config/default.py:15-111 that the hot path reads, with the merge
order of get_config (:228-288): built-in defaults -> experiment YAML -> its BASE_TASK_CONFIG_PATH task YAML (only
ENVIRONMENT.MAX_EPISODE_STEPS is used here) -> trailing ``KEY VALUE`` command-line opts; MODEL_DIR-derived folders.
The reference builds this on yacs through Habitat (absent here); this is a plain-dict equivalent that accepts the reference's
own YAML files unchanged.  Returns the flat namespaces the m2h trainers take.
"""
import ast
import os

import yaml

from ..pretrain.passive.passive_trainer import passive_config
from ..rl.ppo.ppo_trainer import near_target_config

TOP_KEYS = ("SEED", "NUM_PROCESSES", "NUM_UPDATES", "CHECKPOINT_INTERVAL", "LOG_INTERVAL", "EXTRA_RGB", "EXTRA_DEPTH", "TRAINER_NAME",
            "NUM_EPOCHS", "BASE_TASK_CONFIG_PATH", "SENSORS")


def _set_path(d, dotted, value):
    parts = dotted.split(".")
    for p in parts[:-1]:
        d = d.setdefault(p, {})
    d[parts[-1]] = value


def _parse(v):
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def load_raw(exp_config, opts=None):
    raw = {}
    if exp_config:
        with open(exp_config) as f:
            raw = yaml.safe_load(f) or {}
    if opts:
        if len(opts) % 2:
            raise ValueError("opts must be KEY VALUE pairs")
        for k, v in zip(opts[0::2], opts[1::2]):
            _set_path(raw, k, _parse(v))
    return raw


def get_config(exp_config=None, opts=None, model_dir=None, run_type="train", search_dirs=(".",)):
    raw = load_raw(exp_config, opts)
    trainer = raw.get("TRAINER_NAME", "ppo")  # default.py:20
    if trainer == "passive":
        cfg = passive_config()
        for k, v in (raw.get("Pretrain", {}).get("Passive", {}) or {}).items():
            setattr(cfg, k, v)
        if "NUM_EPOCHS" in raw:
            cfg.NUM_EPOCHS = raw["NUM_EPOCHS"]
    else:
        cfg = near_target_config()
        for k, v in (raw.get("RL", {}).get("PPO", {}) or {}).items():
            setattr(cfg, k, v)
        for k in ("NUM_PROCESSES", "NUM_UPDATES", "CHECKPOINT_INTERVAL", "LOG_INTERVAL", "EXTRA_RGB", "EXTRA_DEPTH"):
            if k in raw:
                setattr(cfg, k, raw[k])
        task = raw.get("BASE_TASK_CONFIG_PATH")
        steps = (raw.get("TASK_CONFIG", {}).get("ENVIRONMENT", {}) or {}).get("MAX_EPISODE_STEPS")
        if steps is None and task:
            for d in search_dirs:
                p = os.path.join(d, task)
                if os.path.exists(p):
                    with open(p) as f:
                        steps = ((yaml.safe_load(f) or {}).get("ENVIRONMENT", {}) or {}).get("MAX_EPISODE_STEPS")
                    break
        if steps is not None:
            cfg.MAX_EPISODE_STEPS = int(steps)
    if "SEED" in raw:
        cfg.SEED = raw["SEED"]
    cfg.TRAINER_NAME = trainer
    cfg.RUN_TYPE = run_type
    if model_dir is not None:  # default.py:252-258
        cfg.MODEL_DIR = model_dir
        cfg.CHECKPOINT_FOLDER = os.path.join(model_dir, "data")
        cfg.LOG_FILE = os.path.join(model_dir, "train.log")
    return cfg


def get_trainer(name):
    """baseline_registry.get_trainer (common/baseline_registry.py:37-38): trainers are registered as "passive" and "ppo"."""
    from ..pretrain.passive.passive_trainer import PassiveTrainer
    from ..rl.ppo.ppo_trainer import PPOTrainer
    return {"passive": PassiveTrainer, "ppo": PPOTrainer}.get(name)
