#!/usr/bin/env python3
"""Command line of the reference (main.py:20-78) in front of the m2h trainers:

    python main.py --exp-config <yaml> --run-type train --model-dir <dir> [KEY VALUE ...]

Accepts the reference's experiment YAMLs (config/pretrain_passive.yaml, config/train/nearTarget.yaml, farTarget.yaml).  The
environment is the synthetic on-device feeder; ``--run-type eval`` (the Habitat episodic eval loop) is out of scope.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "move2hear-active-av-separation_amd"))


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--run-type", choices=["train", "eval"], default="train")
    parser.add_argument("--exp-config", type=str, default=None, help="path to the experiment YAML")
    parser.add_argument("--model-dir", default=None)
    parser.add_argument("--cycles", type=int, default=1, help="ppo: training cycles to run; passive: epochs")
    parser.add_argument("opts", default=None, nargs=argparse.REMAINDER, help="KEY VALUE overrides")
    args = parser.parse_args()
    if args.run_type == "eval":
        raise SystemExit("eval needs the Habitat simulator (out of scope); see DESIGN.md section 7")
    import torch
    from m2h.config.default import get_config, get_trainer
    config = get_config(args.exp_config, args.opts, args.model_dir, args.run_type, search_dirs=(".", os.path.dirname(args.exp_config or ".")))
    trainer_init = get_trainer(config.TRAINER_NAME)
    assert trainer_init is not None, f"{config.TRAINER_NAME} is not supported"
    trainer = trainer_init(config, torch.device("cuda", 0))
    trainer.setup()
    if config.TRAINER_NAME == "passive":
        for i, rec in enumerate(trainer.train(num_epochs=args.cycles)):
            print("epoch %d  train bin/mono %.4f %.4f   val %.4f %.4f" % (i, *rec["train"], *rec["val"]))
    else:
        for i, rec in enumerate(trainer.train(args.cycles)):
            print("cycle %d  %d env-steps in %.2f s  pol losses %s  sep losses %s" % (i, rec["env_steps"], rec["seconds"], rec["pol_losses"], rec["sep_losses"]))
        if config.CHECKPOINT_FOLDER:
            trainer.save_checkpoint("ckpt.0.pth")


if __name__ == "__main__":
    main()
