"""GRU state encoder on MI355X: drop-in for audio_separation/rl/models/rnn_state_encoder.py (RNNStateEncoder, :5-143).

``nn.GRU`` is the parameter container (keys ``rnn.weight_ih_l0`` ...; orthogonal init, :36-41).  The two GEMMs of a step run on
the MFMA engine (torch's [3H][K] weight layout is already the packed [N][K] form); the gate math and the hidden-state reset
``h * mask`` are one fused pointwise kernel.  seq_forward (:86-137) splits the sequence at reset steps and runs cuDNN per
stretch; masking h with masks[t] before every step is the same function, needs no device->host sync (the reference's
``.nonzero().cpu()`` at :105) and is what is done here; the input GEMM is batched over all T*N rows.
"""
import torch
import torch.nn as nn

from ... import functional as MF
from ... import ops


class RNNStateEncoder(nn.Module):
    def __init__(self, input_size: int, hidden_size: int, num_layers: int = 1, rnn_type: str = "GRU"):
        super().__init__()
        if rnn_type != "GRU" or num_layers != 1:
            raise NotImplementedError("m2h RNNStateEncoder: single-layer GRU only (what the reference configs use)")
        self._num_recurrent_layers = num_layers
        self._rnn_type = rnn_type
        self.rnn = nn.GRU(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)
        self.layer_init()

    def layer_init(self):
        for name, param in self.rnn.named_parameters():
            if "weight" in name:
                nn.init.orthogonal_(param)
            elif "bias" in name:
                nn.init.constant_(param, 0)

    @property
    def num_recurrent_layers(self):
        return self._num_recurrent_layers

    def forward(self, x, hidden_states, masks):
        n = hidden_states.size(1)
        t = x.size(0) // n  # 1: single_forward (:74-84); > 1: seq_forward (:86-137)
        r = self.rnn
        if (t == 1 and not torch.is_grad_enabled() and n <= ops.GRU_STEP_MAX_ROWS and r.hidden_size % 16 == 0 and x.size(1) % 16 == 0
                and ops.math_mode() == ops.MATH_FP32 and not ops.timing_enabled()):
            # the rollout step (no autograd, 14 rows): input projection, recurrent product and gates in ONE launch (m2h_gru_cell)
            h = ops.gru_cell(x.contiguous(), r.weight_ih_l0.detach(), r.bias_ih_l0.detach(), r.weight_hh_l0.detach(), r.bias_hh_l0.detach(),
                             hidden_states[0].contiguous(), masks.reshape(n).contiguous())
            return h, h.unsqueeze(0)
        if getattr(self, "_memo", None) is None:
            self._memo = [MF._PackMemo(), MF._PackMemo()]      # the transposed weights of the backward's input-gradient products
        out, h = MF.GRUSequence.apply(x, hidden_states[0], masks, r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0, t, self._memo)
        return out, h.unsqueeze(0)
