"""Audio encoder on MI355X: drop-in for audio_separation/rl/models/audio_cnn.py (AudioCNN, :8-140).

Same constructor / forward signature / state_dict keys (``cnn.{0,2,4}.{weight,bias}``, ``cnn.7.{weight,bias}``).
The pre-op (:121-127), BHWC->slice glue (:129-133) run as one HBM-bound kernel; conv8x8s4, conv4x4s2, conv2x2s1 (+bias+ReLU)
and the Linear(+ReLU) run on the MFMA implicit-GEMM engine; the NCHW flatten order of :131-132 is absorbed in the packed
Linear weight.
"""
import numpy as np
import torch
import torch.nn as nn

from ... import functional as MF
from ... import ops


class Flatten(nn.Module):
    """Placeholder keeping the reference's nn.Sequential indices (common/utils.py:11-13); never executed."""

    def forward(self, x):
        return x.reshape(x.size(0), -1)


def conv_output_dim(dimension, padding, dilation, kernel_size, stride):
    return tuple(int(np.floor(((dimension[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) / stride[i]) + 1))
                 for i in range(len(dimension)))


class AudioCNN(nn.Module):
    def __init__(self, observation_space, output_size, encode_monoNmonoFromMem=False):
        super().__init__()
        self.encode_monoNmonoFromMem = encode_monoNmonoFromMem
        self._slice_factor = 16
        self._n_input_audio = 2 * self._slice_factor
        self._cnn_layers_kernel_size = [(8, 8), (4, 4), (2, 2)]
        self._cnn_layers_stride = [(4, 4), (2, 2), (1, 1)]
        shp = observation_space.spaces["mixed_bin_audio_mag"].shape
        cnn_dims = (shp[0] // 16, shp[1])
        for k, s in zip(self._cnn_layers_kernel_size, self._cnn_layers_stride):
            cnn_dims = conv_output_dim(cnn_dims, (0, 0), (1, 1), k, s)
        self._out_dims = cnn_dims
        self.cnn = nn.Sequential(
            nn.Conv2d(self._n_input_audio, 32, kernel_size=self._cnn_layers_kernel_size[0], stride=self._cnn_layers_stride[0]),
            nn.ReLU(True),
            nn.Conv2d(32, 64, kernel_size=self._cnn_layers_kernel_size[1], stride=self._cnn_layers_stride[1]),
            nn.ReLU(True),
            nn.Conv2d(64, 32, kernel_size=self._cnn_layers_kernel_size[2], stride=self._cnn_layers_stride[2]),
            nn.ReLU(True),
            Flatten(),
            nn.Linear(32 * cnn_dims[0] * cnn_dims[1], output_size),
            nn.ReLU(True),
        )
        self.layer_init()
        self._memo = [MF._PackMemo() for _ in range(4)]

    def layer_init(self):
        for layer in self.cnn:
            if isinstance(layer, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(layer.weight, nn.init.calculate_gain("relu"))
                if layer.bias is not None:
                    nn.init.constant_(layer.bias, val=0)

    def encode(self, x_nhwc):
        """conv stack + FC on an already sliced NHWC input [B,32,T,32] (differentiable w.r.t. the parameters)."""
        c0, c1, c2, fc = self.cnn[0], self.cnn[2], self.cnn[4], self.cnn[7]
        x = MF.conv2d(x_nhwc, c0.weight, c0.bias, 4, 0, slope=0.0, memo=self._memo[0], name="audio_cnn.conv0")
        x = MF.conv2d(x, c1.weight, c1.bias, 2, 0, slope=0.0, memo=self._memo[1], name="audio_cnn.conv1")
        x = MF.conv2d(x, c2.weight, c2.bias, 1, 0, slope=0.0, memo=self._memo[2], name="audio_cnn.conv2")
        h, w = self._out_dims
        if x.shape[1] != h or x.shape[2] != w:
            raise RuntimeError("m2h AudioCNN: conv output %s does not match the Linear built for %s" % (tuple(x.shape[1:3]), (h, w)))
        # Linear over the NCHW-flattened map == conv with an (h x w) kernel over the NHWC map (weight viewed [N,32,h,w])
        y = MF.conv2d(x, fc.weight.view(fc.weight.shape[0], 32, h, w), fc.bias, 1, 0, slope=0.0, memo=self._memo[3], name="audio_cnn.fc")
        return y.reshape(y.shape[0], -1)

    def forward_pair(self, pred_mono, pred_monoFromMem):
        """monoNmonoFromMem path without materialising torch.cat((mono, mem), dim=3) (rl/ppo/policy.py:103)."""
        return self.encode(ops.slice_concat_input(pred_mono.contiguous(), pred_monoFromMem.contiguous(), op=2))

    def forward(self, observations, pred_binSepMasks=None, pred_monoNmonoFromMem=None):
        if self.encode_monoNmonoFromMem:
            assert pred_monoNmonoFromMem is not None
            x = ops.slice_concat_input(pred_monoNmonoFromMem.contiguous(), op=2)  # log1p(clamp0(x))  reference :121-122
        else:
            assert pred_binSepMasks is not None
            mix = observations["mixed_bin_audio_mag"]
            x = ops.slice_concat_input(mix.contiguous(), mul=pred_binSepMasks.contiguous(), op=1)  # reference :125-128
        return self.encode(x)


class FusedAudioPair:
    """The two audio encoders of the policy (rl/ppo/policy.py:65-66, :87-89: identical stacks over two different inputs) as ONE
    launch chain for the no-grad rollout step: every layer runs once with the two encoders' channels side by side and a
    block-diagonal weight (zeros off the diagonal), the first conv reading the two sliced inputs as the two sources of a
    concatenating conv.  At 14 envs each of the 8 launches (+ 4 split-K reduces) is latency-bound, so half as many launches is
    what counts; every output channel is its own encoder's sum plus exact zeros.  Not an nn.Module: it owns no parameters, only
    derived tensors that sync() rebuilds in place when the encoders' weights changed (functional.refresh_pack_memos calls it
    before a HIP-graph replay, forward() when run kernel by kernel)."""

    def __init__(self, enc_a, enc_b):
        self.a, self.b = enc_a, enc_b
        self.w = None       # block-diagonal weights (torch layout) of conv0, conv1, conv2, fc
        self.bias = None
        self.key = None
        self._memo = [MF._PackMemo() for _ in range(4)]
        MF._refresh_hooks.add(self)

    def __deepcopy__(self, memo):
        import copy
        return FusedAudioPair(copy.deepcopy(self.a, memo), copy.deepcopy(self.b, memo))  # registered, rebuilt on first use

    def _layers(self, enc):
        h, w = enc._out_dims
        fc = enc.cnn[7]
        return [(enc.cnn[0].weight, enc.cnn[0].bias), (enc.cnn[2].weight, enc.cnn[2].bias), (enc.cnn[4].weight, enc.cnn[4].bias),
                (fc.weight.view(fc.weight.shape[0], 32, h, w), fc.bias)]

    def _key(self, la, lb):
        return tuple((t.data_ptr(), t._version) for pair in la + lb for t in pair) + (MF.param_epoch(),)

    def sync(self):
        """Rebuild the block-diagonal tensors if the encoders' weights changed.  Never raises: inside a HIP-graph capture, or with
        the encoders on the host, it leaves them as they are -- encode() is what refuses to run on stale ones."""
        la, lb = self._layers(self.a), self._layers(self.b)
        if not la[0][0].is_cuda or torch.cuda.is_current_stream_capturing():
            return
        key = self._key(la, lb)
        if key == self.key:
            return
        with torch.no_grad():
            if self.w is None or self.w[0].device != la[0][0].device:
                self.w, self.bias = [], []
                for (wa, ba), (wb, bb) in zip(la, lb):
                    Co, Ci = wa.shape[0], wa.shape[1]
                    self.w.append(torch.zeros((2 * Co, 2 * Ci) + tuple(wa.shape[2:]), device=wa.device, dtype=torch.float32))
                    self.bias.append(torch.zeros(2 * Co, device=wa.device, dtype=torch.float32))
            for W, Bv, (wa, ba), (wb, bb) in zip(self.w, self.bias, la, lb):
                Co, Ci = wa.shape[0], wa.shape[1]
                W[:Co, :Ci].copy_(wa)
                W[Co:, Ci:].copy_(wb)      # the off-diagonal blocks stay exact zeros
                Bv[:Co].copy_(ba)
                Bv[Co:].copy_(bb)
        self.key = key

    def usable(self, xa, xb):
        return (not torch.is_grad_enabled() and self.a._out_dims == self.b._out_dims and xa.shape == xb.shape
                and all(p.is_cuda for p in self.a.parameters()))

    def encode(self, xa, xb, out=None):
        """xa, xb: the two encoders' sliced NHWC inputs [B,32,T,32] -> (features of a [B,512], features of b [B,512]).
        out: optional [B, 1024] destination (a column block of the policy's concatenated feature matrix)."""
        if not torch.cuda.is_current_stream_capturing():
            self.sync()
        elif self.w is None or self.key != self._key(self._layers(self.a), self._layers(self.b)):
            raise RuntimeError("m2h FusedAudioPair: used inside a HIP-graph capture with stale block-diagonal weights "
                               "(call functional.refresh_pack_memos() before the capture)")
        W, Bv, M = self.w, self.bias, self._memo
        x = ops.conv2d_nhwc(xa, M[0].get(W[0], 64), 64, 8, 8, stride=4, pad=0, bias=Bv[0], slope=0.0, x2=xb, name="audio_pair.conv0")
        x = ops.conv2d_nhwc(x, M[1].get(W[1], 64), 128, 4, 4, stride=2, pad=0, bias=Bv[1], slope=0.0, name="audio_pair.conv1")
        x = ops.conv2d_nhwc(x, M[2].get(W[2], 128), 64, 2, 2, stride=1, pad=0, bias=Bv[2], slope=0.0, name="audio_pair.conv2")
        h, w = self.a._out_dims
        if x.shape[1] != h or x.shape[2] != w:
            raise RuntimeError("m2h FusedAudioPair: conv output %s does not match the Linear built for %s" % (tuple(x.shape[1:3]), (h, w)))
        n = W[3].shape[0]
        y = ops.conv2d_nhwc(x, M[3].get(W[3], 64), n, h, w, stride=1, pad=0, bias=Bv[3], slope=0.0, name="audio_pair.fc", out=out)
        y = y.reshape(x.shape[0], n) if out is None else out
        return y[:, :n // 2], y[:, n // 2:]
