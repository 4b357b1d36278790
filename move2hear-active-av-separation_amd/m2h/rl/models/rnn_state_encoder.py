"""Recurrent state encoder on MI355X: drop-in for audio_separation/rl/models/rnn_state_encoder.py (RNNStateEncoder, :5-143): GRU (what
policy.py:63 constructs: fused kernels) with any number of layers, and the class's LSTM variant (library GEMMs + a fused cell kernel).

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
        if rnn_type not in ("GRU", "LSTM"):
            raise ValueError("m2h RNNStateEncoder: rnn_type must be GRU or LSTM (rnn_state_encoder.py:22)")
        self._num_recurrent_layers = num_layers
        self._rnn_type = rnn_type
        self.rnn = getattr(nn, rnn_type)(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)
        self.layer_init()

    def layer_init(self):
        for name, param in self.rnn.named_parameters():
            if "weight" in name:
                nn.init.orthogonal_(param)
            elif "bias" in name:
                nn.init.constant_(param, 0)

    @property
    def num_recurrent_layers(self):
        return self._num_recurrent_layers * (2 if "LSTM" in self._rnn_type else 1)   # (:43-47: h and c packed along dim 0)

    def _lstm_forward(self, x, hidden_states, masks, n, t):
        """The "LSTM" variant (:10-34, 49-61; policy.py:63 never selects it and no config key reaches it).  hidden_states packs
        (h, c) along dim 0 ([2 L, N, H], :49-61); both are multiplied by the step's reset mask (:63-69).  The four GEMMs of a step
        run on the library's engine (the input projection batched over all T N rows); the gate math of a step -- sum of the two products,
        four activations, cell update, the cell state's reset mask -- is one fused launch forward and one backward (functional.LSTMCell,
        m2h_lstm_cell / m2h_lstm_cell_bwd; round 6: it was ~12 torch pointwise kernels per step under torch autograd)."""
        r, L, H = self.rnn, self._num_recurrent_layers, self.rnn.hidden_size
        hs, cs = hidden_states[:L], hidden_states[L:]
        m = masks.reshape(t, n, 1)
        out, h_out, c_out = x, [], []
        for l in range(L):
            w_ih, w_hh, b_ih, b_hh = (getattr(r, "%s_l%d" % (name, l)) for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
            gi = MF.linear(out.contiguous(), w_ih, b_ih, name="lstm.ih").reshape(t, n, 4 * H)
            h, c, steps = hs[l], cs[l], []
            for k in range(t):
                gh = MF.linear((h * m[k]).contiguous(), w_hh, b_hh, name="lstm.hh")
                h, c = MF.LSTMCell.apply(gi[k], gh, c, m[k])         # (the cell state's mask is applied inside; nn.LSTM's gate order i, f, g, o)
                steps.append(h)
            out = torch.cat(steps, 0)
            h_out.append(h)
            c_out.append(c)
        return out, torch.stack(h_out + c_out, 0)

    def forward(self, x, hidden_states, masks):
        n = hidden_states.size(1)
        t = x.size(0) // n  # 1: single_forward (:74-84); > 1: seq_forward (:86-137)
        r = self.rnn
        L = self._num_recurrent_layers
        if self._rnn_type == "LSTM":
            return self._lstm_forward(x, hidden_states, masks, n, t)
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
