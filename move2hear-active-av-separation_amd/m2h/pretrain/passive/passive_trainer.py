"""Passive pre-training on MI355X: the training loop of audio_separation/pretrain/passive/passive_trainer.py
(PassiveTrainer, :50-286) around the m2h modules and a synthetic on-device data feeder.

Kept: model construction (:57-82), Adam(lr 5e-4, eps 1e-5) over the trainable parameters (:194-195), the per-batch flow
get_binSepMasks -> convert_bin2mono(masks.detach()) -> optimize_supervised_loss (:218-249, 269-286) including the fact that
``clip_grad_norm_`` runs BEFORE ``backward`` and therefore clips nothing, train-mode BatchNorm for the train
split and eval mode for validation (:211-214), best-validation checkpointing in the reference format (:84-99, 259-266).
Replaced: PassiveDataset/DataLoader (RIR convolution + librosa STFT on 60 CPU workers) by a seeded on-device batch feeder
(the GPU STFT feeder is row N1 of SURVEY 8f); TensorBoard logging dropped.
Losses stay on the device; one host read per epoch instead of two ``.item()`` per batch (:248-249).
"""
import os
from types import SimpleNamespace

import torch

from ... import functional as MF
from ...common.spaces import move2hear_observation_space
from ...optim import FlatAdam
from .passive import Passive
from .policy import Move2HearPassiveWoMemoryPolicy


def passive_config(**over):
    """Pretrain.Passive.* of config/default.py:106-111 + config/pretrain_passive.yaml (BATCH_SIZE 64)."""
    c = dict(SEED=0, lr=5.0e-4, eps=1.0e-5, max_grad_norm=0.8, NUM_EPOCHS=1000, BATCH_SIZE=64, BATCHES_PER_EPOCH=8, VAL_BATCHES=2,
             CHECKPOINT_FOLDER=None, TM=32,
             use_hip_graphs=True,  # build-side key: replay forward + losses + backward of a training batch from a HIP graph
             wgrad_side_branches=True)   # build-side key: inside that graph the weight gradients are side branches of the backward chains (functional.wgrad_side_branches)
    c.update(over)
    return SimpleNamespace(**c)


class SyntheticPassiveFeeder:
    """Seeded batches with the tensors the reference DataLoader yields (:219-222): mixed_audio [B,512,Tm,2], gt_bin_mag
    [B,512,Tm,2], gt_mono_mag [B,512,Tm,1], target_class [B,1]; log1p-magnitude statistics of dataset.py:190-228."""

    def __init__(self, device, batch_size, tm, seed):
        self.device, self.bs, self.tm = device, batch_size, tm
        self.g = torch.Generator(device=device).manual_seed(int(seed))

    def batch(self):
        d, g, B, T = self.device, self.g, self.bs, self.tm
        gain = torch.exp(torch.rand(B, 512, 1, 1, device=d, generator=g) * 3.0 - 2.0)
        src = [torch.sqrt(torch.randn(B, 512, T, 2, device=d, generator=g) ** 2 + torch.randn(B, 512, T, 2, device=d, generator=g) ** 2) * gain
               for _ in range(2)]
        mixed = torch.log1p(0.5 * (src[0] + src[1])).contiguous()
        gt_bin = (0.5 * src[0]).contiguous()                       # magnitude of the target source, binaural
        gt_mono = (0.5 * src[0].mean(dim=3, keepdim=True)).contiguous()
        tc = torch.randint(0, 11, (B, 1), device=d, generator=g)
        return mixed, gt_bin, gt_mono, tc


class PassiveTrainer:
    def __init__(self, config=None, device=None):
        self.config = config if config is not None else passive_config()
        self.device = device if device is not None else torch.device("cuda", 0)
        self.actor_critic = None
        self.agent = None
        self.optimizer = None
        self._train_graph = None
        self._train_batches = 0

    def _setup_passive_agent(self):
        self.actor_critic = Move2HearPassiveWoMemoryPolicy(observation_space=move2hear_observation_space(self.config.TM))
        self.agent = Passive(actor_critic=self.actor_critic)
        self.actor_critic.to(self.device)
        self.actor_critic.train()

    def optimize_supervised_loss(self, mixed_audio, pred_binSepMasks, gt_bin_mag, pred_mono, gt_mono_mag, split="train"):
        """:269-286.  bin_loss = L1(mask*(exp(mix)-1), gt_bin); mono_loss = L1(mono, gt_mono); train: zero_grad, (no-op clip),
        backward, Adam step.  Returns device scalars."""
        if split == "train":
            bin_loss = MF.bin_l1_loss(pred_binSepMasks, mixed_audio, gt_bin_mag, cstep=1)
            mono_loss = MF.l1_loss(pred_mono, gt_mono_mag, 0)
            self.optimizer.zero_grad()
            loss = bin_loss + mono_loss
            # nn.utils.clip_grad_norm_ is called here in the reference, on freshly zeroed gradients: no effect (D11)
            loss.backward()
            self.optimizer.step(max_grad_norm=None)
        else:
            from ... import ops
            bin_loss = ops.bin_l1_loss(mixed_audio, pred_binSepMasks, gt_bin_mag, cstep=1)
            mono_loss = MF.l1_loss(pred_mono, gt_mono_mag, 0)
        return bin_loss.detach(), mono_loss.detach()

    def _train_batch_graph(self, mixed_audio, gt_bin_mag, gt_mono_mag, target_class):
        """A training batch with forward (train-mode BN, running statistics included), both losses and the whole backward
        replayed from one HIP graph: at the reference batch (64 x 512x32) the step is ~700 small launches behind Python and
        autograd dispatch.  The batch is copied into four static tensors; weight (re)packing is part of the graph (the packed
        copies are made from the current weights at every replay) and so is the optimizer step (its step count and learning
        rate reach the kernel through a device buffer: FlatAdam.captured_step).  Returns the two loss scalars of THIS replay (static tensors,
        overwritten by the next one)."""
        from ... import graphs, ops
        gs = self._train_graph
        self.optimizer.build()
        sig = (tuple(mixed_audio.shape), tuple(gt_bin_mag.shape), tuple(gt_mono_mag.shape), tuple(target_class.shape), target_class.dtype,
               tuple(p.data_ptr() for p in self.actor_critic.parameters()))
        if gs is None or gs.sig != sig:
            gs = self._train_graph = SimpleNamespace(sig=sig, graph=None, inputs=tuple(torch.empty_like(t) for t in (
                mixed_audio, gt_bin_mag, gt_mono_mag, target_class)), losses=None, forked=False, memos_a=[], memos_b=[])
        srcs = (mixed_audio, gt_bin_mag, gt_mono_mag, target_class)
        if all(s.is_cuda and s.is_contiguous() and s.dtype == d.dtype for s, d in zip(srcs, gs.inputs)):
            if getattr(gs, "no_idx", None) is None:
                gs.no_idx = torch.zeros(1, dtype=torch.int64, device=self.device)
            ops.rows_copy([(s, d, -1, -1) for s, d in zip(srcs, gs.inputs)], gs.no_idx)     # the four batch tensors in ONE launch
        else:
            for dst, src in zip(gs.inputs, srcs):
                dst.copy_(src)
        ac = self.actor_critic
        if gs.graph is None:
            g = torch.cuda.CUDAGraph()
            gs.forked = graphs.parallel_branches
            # which packed-weight memos the step reads (the warm-up step has told every memo which packs it needs)
            gs.memos_a = MF.memos_of(ac.binSep_enc, ac.binSep_dec)
            gs.memos_b = MF.memos_of(ac.bin2mono_enc, ac.bin2mono_dec)
            if gs.forked:
                MF.refresh_pack_memos()    # the first replay finds the first network's packs made (later ones: packed at the end of the step before)
            with graphs.capture(g), MF.wgrad_side_branches(getattr(self.config, "wgrad_side_branches", True)):
                mix, gtb, gtm, tc = gs.inputs
                self.optimizer.zero_grad()
                if gs.forked:
                    # Weight packs (the packed copies are made from the current weights inside the graph): the second network's at the top of
                    # ITS branch, under the first network's forward; the first network's at the END of its branch, behind its Adam step, for
                    # the next replay -- that branch is idle there while the other one still runs its backward.
                    side = graphs.side_stream(self.device)
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        MF.refresh_pack_memos(hooks=False, only=gs.memos_b, force=True)
                else:
                    MF.refresh_pack_memos(hooks=False, only=gs.memos_a + gs.memos_b, force=True)
                if gs.forked:
                    # The mono separator reads the binaural one's masks DETACHED (:218-249): the two networks' backward passes are
                    # independent, and so is the second network's forward from the first one's backward.  Two branches of the graph:
                    #   this stream:  forward A -> bin loss -> backward A
                    #   side stream:  (after forward A) forward B -> mono loss -> backward B
                    # (two backward() calls instead of one on the sum: the same gradients, each network's from its own loss).
                    main = torch.cuda.current_stream()
                    side = graphs.side_stream(self.device)
                    with MF.batched_bn_counters():   # the 20 num_batches_tracked increments as one launch
                        masks = self.actor_critic.get_binSepMasks({"mixed_bin_audio_mag": mix, "target_class": tc})
                        side.wait_stream(main)
                        with torch.cuda.stream(side):
                            mono = self.actor_critic.convert_bin2mono(masks.detach(), mixed_audio=mix)
                    with torch.cuda.stream(side):
                        mono_loss = MF.l1_loss(mono, gtm, 0)
                        mono_loss.backward(MF.unit_grad(self.device))
                        MF.flush_deferred_wgrads(self.device)      # (the encoder's weight gradients: deferred behind the last flush point)
                    bin_loss = MF.bin_l1_loss(masks, mix, gtb, cstep=1)
                    bin_loss.backward(MF.unit_grad(self.device))
                    MF.flush_deferred_wgrads(self.device)
                    # the first network's Adam step right behind its own backward (FlatAdam.captured_step: lr and the
                    # bias corrections come from a device buffer the host refreshes before each replay)
                    self.optimizer.captured_step(list(ac.binSep_enc.parameters()) + list(ac.binSep_dec.parameters()))
                    MF.refresh_pack_memos(hooks=False, only=gs.memos_a, force=True)
                    main.wait_stream(side)
                    # The second network's Adam step HERE, on the stream the capture began on: its weight gradients were forked from the side
                    # stream to a side stream of their own (MF.wgrad_side_branches), and only the origin stream may join those (MF.join_wgrad_branches).
                    # Nothing is lost: that branch is the step's longest chain, its Adam step ends the step either way.
                    MF.join_wgrad_branches(self.device)
                    self.optimizer.captured_step(list(ac.bin2mono_enc.parameters()) + list(ac.bin2mono_dec.parameters()))
                else:
                    with MF.batched_bn_counters():
                        masks = self.actor_critic.get_binSepMasks({"mixed_bin_audio_mag": mix, "target_class": tc})
                        mono = self.actor_critic.convert_bin2mono(masks.detach(), mixed_audio=mix)
                    bin_loss = MF.bin_l1_loss(masks, mix, gtb, cstep=1)
                    mono_loss = MF.l1_loss(mono, gtm, 0)
                    (bin_loss + mono_loss).backward()
                    MF.flush_deferred_wgrads(self.device)
                    self.optimizer.captured_step()
                gs.losses = (bin_loss.detach(), mono_loss.detach())
            gs.graph = g
        if gs.forked:
            MF.refresh_pack_memos(hooks=False, only=gs.memos_a)   # nothing, unless somebody else changed the weights since the last replay (load_state_dict)
        self.optimizer.begin_replayed_step()
        if gs.forked:
            torch.cuda.current_stream().synchronize()   # a graph with parallel branches goes onto a drained stream (m2h/graphs.py)
        graphs.replay(gs.graph)
        self.optimizer.end_replayed_step()
        if gs.forked:
            for m in gs.memos_a:
                m.mark_fresh()            # (the replay packed them from the weights it left)
        return gs.losses

    def train_batch(self, mixed_audio, gt_bin_mag, gt_mono_mag, target_class, split="train"):
        if split == "train":
            self._train_batches += 1
            from ... import ops
            if (getattr(self.config, "use_hip_graphs", False) and self._train_batches > 1 and self.actor_critic.training
                    and not ops.timing_enabled()):
                return self._train_batch_graph(mixed_audio, gt_bin_mag, gt_mono_mag, target_class)  # (first batch: warm-up, kernel by kernel)
        obs_batch = {"mixed_bin_audio_mag": mixed_audio, "target_class": target_class}
        if split == "train":
            with MF.batched_bn_counters():
                pred_binSepMasks = self.actor_critic.get_binSepMasks(obs_batch)
                pred_mono = self.actor_critic.convert_bin2mono(pred_binSepMasks.detach(), mixed_audio=mixed_audio)
        else:
            with torch.no_grad():
                pred_binSepMasks = self.actor_critic.get_binSepMasks(obs_batch)
                pred_mono = self.actor_critic.convert_bin2mono(pred_binSepMasks.detach(), mixed_audio=mixed_audio)
        return self.optimize_supervised_loss(mixed_audio, pred_binSepMasks, gt_bin_mag, pred_mono, gt_mono_mag, split)

    def setup(self):
        cfg = self.config
        torch.manual_seed(cfg.SEED)
        self._setup_passive_agent()
        self.optimizer = FlatAdam([p for p in self.actor_critic.parameters() if p.requires_grad], lr=cfg.lr, eps=cfg.eps)
        self.feeders = {"train": SyntheticPassiveFeeder(self.device, cfg.BATCH_SIZE, cfg.TM, cfg.SEED + 1),
                        "val": SyntheticPassiveFeeder(self.device, cfg.BATCH_SIZE, cfg.TM, cfg.SEED + 2)}

    def save_checkpoint(self, file_name):
        ckpt = {"state_dict": self.agent.state_dict(), "config": vars(self.config)}
        os.makedirs(self.config.CHECKPOINT_FOLDER, exist_ok=True)
        torch.save(ckpt, os.path.join(self.config.CHECKPOINT_FOLDER, file_name))

    def load_checkpoint(self, checkpoint_path, *args, **kwargs):
        """torch.load of a {"state_dict", "config"} checkpoint (reference passive_trainer.py save/load pair); returns the dict."""
        kwargs.setdefault("map_location", "cpu")
        kwargs.setdefault("weights_only", False)
        return torch.load(checkpoint_path, *args, **kwargs)

    def load_state_dict(self, state_dict, strict=True):
        out = self.agent.load_state_dict(state_dict, strict=strict)
        from ... import functional as MF
        MF.bump_param_epoch()
        return out

    def train(self, num_epochs=None):
        cfg = self.config
        if self.actor_critic is None:
            self.setup()
        best = float("inf")
        log = []
        for epoch in range(num_epochs if num_epochs is not None else cfg.NUM_EPOCHS):
            rec = {}
            for split, nb in (("train", cfg.BATCHES_PER_EPOCH), ("val", cfg.VAL_BATCHES)):
                self.actor_critic.train() if split == "train" else self.actor_critic.eval()
                acc = torch.zeros(2, device=self.device)
                for _ in range(nb):
                    b, m = self.train_batch(*self.feeders[split].batch(), split=split)
                    acc += torch.stack((b, m))
                rec[split] = (acc / nb).tolist()
            if rec["val"][1] < best and cfg.CHECKPOINT_FOLDER:
                best = rec["val"][1]
                self.save_checkpoint("best_ckpt_val.pth")
            log.append(rec)
        return log
