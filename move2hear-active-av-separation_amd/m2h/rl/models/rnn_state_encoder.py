"""GRU state encoder on MI355X: drop-in for audio_separation/rl/models/rnn_state_encoder.py (RNNStateEncoder, :5-143).

``nn.GRU`` is the parameter container (keys ``rnn.weight_ih_l0`` ...; orthogonal init, :36-41).  The two GEMMs of a step run on
the MFMA engine (torch's [3H][K] weight layout is already the packed [N][K] form); the gate math and the hidden-state reset
``h * mask`` are one fused pointwise kernel.  seq_forward (:86-137) splits the sequence at reset steps and runs cuDNN per
stretch; masking h with masks[t] before every step is the same function, needs no device->host sync (the reference's
``.nonzero().cpu()`` at :105) and is what is done here; the input GEMM is batched over all T*N rows.
"""
import torch
import torch.nn as nn

from ... import ops
from ._common import check_inference


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

    def _step(self, gi, h, mask):
        gh = ops.linear(h, self.rnn.weight_hh_l0.detach(), None, name="gru.hh")
        return ops.gru_gates(gi, gh, self.rnn.bias_hh_l0.detach(), h, mask)

    def single_forward(self, x, hidden_states, masks):
        h = hidden_states[0].contiguous()
        gi = ops.linear(x.contiguous(), self.rnn.weight_ih_l0.detach(), self.rnn.bias_ih_l0.detach(), name="gru.ih")
        h = self._step(gi, h, masks.reshape(-1).contiguous())
        return h, h.unsqueeze(0)

    def seq_forward(self, x, hidden_states, masks):
        n = hidden_states.size(1)
        t = int(x.size(0) / n)
        gi = ops.linear(x.contiguous(), self.rnn.weight_ih_l0.detach(), self.rnn.bias_ih_l0.detach(), name="gru.ih")
        gi = gi.view(t, n, -1)
        masks = masks.reshape(t, n).contiguous()
        h = hidden_states[0].contiguous()
        outs = torch.empty((t, n, h.shape[1]), device=x.device, dtype=torch.float32)
        for i in range(t):
            h = self._step(gi[i], h, masks[i])
            outs[i].copy_(h)
        return outs.view(t * n, -1), h.unsqueeze(0)

    def forward(self, x, hidden_states, masks):
        check_inference(self, x, hidden_states)
        if x.size(0) == hidden_states.size(1):
            return self.single_forward(x, hidden_states, masks)
        return self.seq_forward(x, hidden_states, masks)
