"""Decoder head of RAGraph (mirrors RAGraph_*/ragraph_utils/TaskDecoder.py: a two-layer MLP with LeakyReLU between,
state_dict keys fc1.* / fc2.*).  The nn.Linear objects only hold the parameters; both GEMMs run on the HIP linear
kernel, the first with LeakyReLU fused into its epilogue, and are differentiable through ragraph_amd.autograd."""
import torch.nn as nn

from .. import autograd as A
from .. import kernels as K


class TaskDecoder(nn.Module):
    NEGATIVE_SLOPE = 0.01  # nn.LeakyReLU() default, TaskDecoder.py:7

    def __init__(self, input_dim, hiddden_dim, output_dim):
        super().__init__()
        widths = {"fc1": (input_dim, hiddden_dim), "fc2": (hiddden_dim, output_dim)}
        for name, (fan_in, fan_out) in widths.items():
            self.add_module(name, nn.Linear(fan_in, fan_out))

    def layers(self):
        return self.fc1, self.fc2

    def reset_parameters(self):
        for layer in self.layers():
            layer.reset_parameters()

    def forward(self, x):
        first, second = self.layers()
        hidden = A.linear(x, first.weight, first.bias, act=K.ACT_LEAKY, alpha=self.NEGATIVE_SLOPE)
        return A.linear(hidden, second.weight, second.bias)

    def extra_repr(self):
        return f"{self.fc1.in_features} -> {self.fc1.out_features} -> {self.fc2.out_features}, HIP linear kernels"
