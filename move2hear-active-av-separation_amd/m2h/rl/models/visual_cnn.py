"""Visual encoder on MI355X: drop-in for audio_separation/rl/models/visual_cnn.py (VisualCNN, :8-152).

Same constructor / forward / state_dict keys (``cnn.{0,2,4}.{weight,bias}``, ``cnn.6.{weight,bias}``).  rgb/255 (+depth) is
written once as a 4-channel NHWC tensor (the observation already is BHWC == NHWC); the 3- or 4-channel first conv is packed
to 4 input channels; conv8x8s4+ReLU, conv4x4s2+ReLU, conv3x3s1 and the Linear+ReLU run on the MFMA implicit-GEMM engine.
"""
import torch
import torch.nn as nn

from ... import functional as MF
from ... import ops
from .audio_cnn import Flatten, conv_output_dim


class VisualCNN(nn.Module):
    def __init__(self, observation_space, output_size, extra_rgb, extra_depth):
        super().__init__()
        if "rgb" in observation_space.spaces and (not extra_rgb):
            self._n_input_rgb = observation_space.spaces["rgb"].shape[2]
        else:
            self._n_input_rgb = 0
        if "depth" in observation_space.spaces and (not extra_depth):
            self._n_input_depth = observation_space.spaces["depth"].shape[2]
        else:
            self._n_input_depth = 0
        self._cnn_layers_kernel_size = [(8, 8), (4, 4), (3, 3)]
        self._cnn_layers_stride = [(4, 4), (2, 2), (1, 1)]
        if self._n_input_rgb > 0:
            cnn_dims = (128, 128)  # hard-coded in the reference (:43-45)
        elif self._n_input_depth > 0:
            cnn_dims = tuple(observation_space.spaces["depth"].shape[:2])
        if self.is_blind:
            self.cnn = nn.Sequential()
        else:
            for k, s in zip(self._cnn_layers_kernel_size, self._cnn_layers_stride):
                cnn_dims = conv_output_dim(cnn_dims, (0, 0), (1, 1), k, s)
            self._out_dims = cnn_dims
            self.cnn = nn.Sequential(
                nn.Conv2d(self._n_input_rgb + self._n_input_depth, 32, kernel_size=self._cnn_layers_kernel_size[0],
                          stride=self._cnn_layers_stride[0]),
                nn.ReLU(True),
                nn.Conv2d(32, 64, kernel_size=self._cnn_layers_kernel_size[1], stride=self._cnn_layers_stride[1]),
                nn.ReLU(True),
                nn.Conv2d(64, 32, kernel_size=self._cnn_layers_kernel_size[2], stride=self._cnn_layers_stride[2]),
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

    @property
    def is_blind(self):
        return self._n_input_rgb + self._n_input_depth == 0

    def prepare(self, observations, out=None):
        """The conv stack's input: rgb (+ depth) / 255 as one NHWC [B, H, W, 4] tensor (visual_cnn.py:135-140).  A function of the observations
        alone: update_pol makes it once for its four epochs (``forward(..., x=)``)."""
        rgb = observations["rgb"]
        depth = observations["depth"] if self._n_input_depth > 0 else None
        return ops.visual_input(rgb.contiguous(), depth.contiguous() if depth is not None else None, out=out)

    def forward(self, observations, out=None, x=None):
        """out: optional [B, output_size] destination (a column block of the policy's concatenated feature matrix) for the no-grad
        rollout path: the last layer then writes there instead of into a tensor of its own.  x: ``prepare(observations)`` when the caller holds it."""
        if self.is_blind:
            # (the reference fails here too: its forward concatenates an empty list, visual_cnn.py:147-150, and policy.py:98 always calls it)
            raise NotImplementedError("m2h VisualCNN: blind configuration has no encoder")
        if self._n_input_rgb != 3 or self._n_input_depth not in (0, 1):
            raise NotImplementedError("m2h VisualCNN: built for rgb (3 ch) with optional depth (1 ch)")
        if x is None:
            x = self.prepare(observations)
        c0, c1, c2, fc = self.cnn[0], self.cnn[2], self.cnn[4], self.cnn[6]
        x = MF.conv2d(x, c0.weight, c0.bias, 4, 0, slope=0.0, memo=self._memo[0], name="visual_cnn.conv0")  # Ci 3 packed to 4
        x = MF.conv2d(x, c1.weight, c1.bias, 2, 0, slope=0.0, memo=self._memo[1], name="visual_cnn.conv1")
        x = MF.conv2d(x, c2.weight, c2.bias, 1, 0, slope=1.0, memo=self._memo[2], name="visual_cnn.conv2")  # no ReLU (:81-88)
        h, w = self._out_dims
        fcw = fc.weight.view(fc.weight.shape[0], 32, h, w)
        if out is not None and not torch.is_grad_enabled():
            ops.conv2d_nhwc(x, self._memo[3].get(fcw, 32), fcw.shape[0], h, w, bias=fc.bias.detach(), slope=0.0, out=out, name="visual_cnn.fc")
            return out
        y = MF.conv2d(x, fcw, fc.bias, 1, 0, slope=0.0, memo=self._memo[3], name="visual_cnn.fc")
        return y.reshape(y.shape[0], -1)
