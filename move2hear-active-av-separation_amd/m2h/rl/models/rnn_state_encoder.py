"""GRU state encoder on MI355X: drop-in for audio_separation/rl/models/rnn_state_encoder.py (RNNStateEncoder, :5-143), any number of layers.

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
        if rnn_type != "GRU":
            # (the reference's class also takes "LSTM" (:10-34); policy.py:63 never passes it and no config key reaches it)
            raise NotImplementedError("m2h RNNStateEncoder: GRU only (what policy.py:63 constructs); LSTM cells are not built")
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
        L = self._num_recurrent_layers
        if (L == 1 and t == 1 and not torch.is_grad_enabled() and n <= ops.GRU_STEP_MAX_ROWS and r.hidden_size % 16 == 0 and x.size(1) % 16 == 0
                and ops.math_mode() == ops.MATH_FP32 and not ops.timing_enabled()):
            # the rollout step (no autograd, 14 rows): input projection, recurrent product and gates in ONE launch (m2h_gru_cell)
            h = ops.gru_cell(x.contiguous(), r.weight_ih_l0.detach(), r.bias_ih_l0.detach(), r.weight_hh_l0.detach(), r.bias_hh_l0.detach(),
                             hidden_states[0].contiguous(), masks.reshape(n).contiguous())
            return h, h.unsqueeze(0)
        if getattr(self, "_memo", None) is None:
            self._memo = [[MF._PackMemo(), MF._PackMemo()] for _ in range(L)]   # the transposed weights of the backward's input-gradient products
        # nn.GRU with num_layers > 1 (:28-32): layer l runs over layer l-1's output sequence from its own hidden state hidden_states[l];
        # the reset mask multiplies every layer's hidden state (_mask_hidden, :63-69), no dropout between layers (the default)
        out, hs = x, []
        for l in range(L):
            w = [getattr(r, "%s_l%d" % (name, l)) for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            out, h = MF.GRUSequence.apply(out, hidden_states[l], masks, w[0], w[1], w[2], w[3], t, self._memo[l])
            hs.append(h)
        return out, (hs[0].unsqueeze(0) if L == 1 else torch.stack(hs, 0))
