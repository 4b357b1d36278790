#!/usr/bin/env python3
"""Command line of the reference (main.py:20-78) in front of the m2h trainers:

    python main.py --exp-config <yaml> --run-type train --model-dir <dir> [KEY VALUE ...]

    python main.py --exp-config <yaml> --run-type eval  --model-dir <dir> [--eval-ckpt <file>] [--eval-episodes N]

Accepts the reference's experiment YAMLs (config/pretrain_passive.yaml, config/train/nearTarget.yaml, farTarget.yaml).  The
environment is the synthetic on-device feeder; ``--run-type eval`` runs the evaluation loop of ``_eval_checkpoint``
(ppo_trainer.py:1015-1551: eval-mode policy, per-step STFT-L2, waveform SI-SDR of the last step, mean/std aggregation) on
that feeder -- the Habitat episode datasets and simulator it iterates in the reference are out of scope.
"""
import json
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "move2hear-active-av-separation_amd"))


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--run-type", choices=["train", "eval"], default="train")
    parser.add_argument("--exp-config", type=str, default=None, help="path to the experiment YAML")
    parser.add_argument("--model-dir", default=None)
    parser.add_argument("--cycles", type=int, default=None,
                        help="ppo: cap on the training cycles (default: NUM_UPDATES / num_updates_per_cycle, as the reference); passive: epochs (default: the YAML's NUM_EPOCHS)")
    parser.add_argument("--eval-ckpt", default=None, help="eval: checkpoint file (default: <model-dir>/data/ckpt.0.pth if present)")
    parser.add_argument("--eval-episodes", type=int, default=None, help="eval: episodes to aggregate (default: NUM_PROCESSES)")
    parser.add_argument("opts", default=None, nargs=argparse.REMAINDER, help="KEY VALUE overrides")
    args = parser.parse_args()
    import torch
    from m2h.config.default import get_config, get_trainer
    config = get_config(args.exp_config, args.opts, args.model_dir, args.run_type, search_dirs=(".", os.path.dirname(args.exp_config or ".")))
    trainer_init = get_trainer(config.TRAINER_NAME)
    assert trainer_init is not None, f"{config.TRAINER_NAME} is not supported"
    if config.TRAINER_NAME == "ppo":
        # one process per GPU: rendezvous from torch.distributed.run's variables (what init_distrib_slurm reads, ddppo_utils.py:117-165)
        # BEFORE any other GPU work; world rank / size go to the trainer (per-rank seeds, parameter broadcast, gradient all-reduce)
        from m2h.rl.ppo import ddppo_utils
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        device = torch.device("cuda", local_rank)
        world_rank, world_size = 0, 1
        if getattr(config, "use_ddppo", False):
            os.environ.setdefault("MASTER_ADDR", str(getattr(config, "master_addr", "127.0.0.1")))
            os.environ.setdefault("MASTER_PORT", str(getattr(config, "master_port", 8738)))
            torch.cuda.set_device(device)
            backend = {"nccl": "nccl", "gloo": "gloo"}[str(getattr(config, "ddppo_distrib_backend", "NCCL")).lower()]
            _lr, world_rank, world_size = ddppo_utils.init_distrib(backend, device=device)
        trainer = trainer_init(config, device, world_rank=world_rank, world_size=world_size)
    else:
        trainer = trainer_init(config, torch.device("cuda", 0))
    trainer.setup()
    if args.run_type == "eval":
        if config.TRAINER_NAME != "ppo":
            raise SystemExit("--run-type eval is the RL evaluation loop (TRAINER_NAME ppo)")
        ckpt = args.eval_ckpt
        if ckpt is None and config.CHECKPOINT_FOLDER and os.path.exists(os.path.join(config.CHECKPOINT_FOLDER, "ckpt.0.pth")):
            ckpt = os.path.join(config.CHECKPOINT_FOLDER, "ckpt.0.pth")
        if getattr(config, "switch_policy", False):  # config/test/farTarget.yaml: two-policy checkpoint (state_dict_nav / state_dict_qualImprov)
            stats = trainer.eval(num_episodes=args.eval_episodes, switch_checkpoint_path=ckpt)
        else:
            stats = trainer.eval(num_episodes=args.eval_episodes, checkpoint_path=ckpt)
        print(json.dumps(stats, indent=1))
        return
    if config.TRAINER_NAME == "passive":
        # --cycles absent: the YAML's NUM_EPOCHS, as the reference's passive loop (passive_trainer.py:252-257)
        for i, rec in enumerate(trainer.train(num_epochs=args.cycles)):
            print("epoch %d  train bin/mono %.4f %.4f   val %.4f %.4f" % (i, *rec["train"], *rec["val"]))
    else:
        # the reference loop (ppo_trainer.py:730-1011): NUM_UPDATES / num_updates_per_cycle cycles, window statistics per policy
        # update, ckpt.<k>.pth every CHECKPOINT_INTERVAL separator updates, written by world rank 0 only
        for i, rec in enumerate(trainer.train(args.cycles)):
            if trainer.world_rank == 0:
                print("cycle %d  %d env-steps in %.2f s  pol losses %s  sep losses %s" % (i, rec["env_steps"], rec["seconds"], rec["pol_losses"], rec["sep_losses"]))
        # every rank: a digest of the replica it ends with (DD-PPO keeps the replicas bit-identical; a desynchronised run shows here)
        import hashlib
        h = hashlib.sha1()
        for k, v in sorted(trainer.actor_critic.state_dict().items()):
            h.update(k.encode())
            h.update(v.detach().cpu().contiguous().numpy().tobytes())
        print("rank %d of %d: final weights sha1 %s" % (trainer.world_rank, trainer.world_size, h.hexdigest()), flush=True)


if __name__ == "__main__":
    main()
