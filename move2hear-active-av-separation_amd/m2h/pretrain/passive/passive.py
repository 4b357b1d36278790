"""Agent wrapper of passive pre-training: the state_dict root ``actor_critic.`` of the checkpoints that
audio_separation/pretrain/passive/passive.py saves; the training step itself is PassiveTrainer.train_batch."""
import torch.nn as nn


class Passive(nn.Module):
    def __init__(self, actor_critic):
        super().__init__()
        self.actor_critic = actor_critic
