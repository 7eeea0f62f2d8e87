"""Mirror of RAGraph_*/ragraph_utils/TaskDecoder.py: fc2(LeakyReLU(fc1(x))), same parameter names (fc1, fc2)."""
import torch.nn as nn

from .. import autograd as A
from .. import kernels as K


class TaskDecoder(nn.Module):
    def __init__(self, input_dim, hiddden_dim, output_dim):
        super().__init__()
        self.fc1 = nn.Linear(input_dim, hiddden_dim)   # parameter containers only; the math runs in libragraph_hip
        self.act = nn.LeakyReLU()
        self.fc2 = nn.Linear(hiddden_dim, output_dim)

    def reset_parameters(self):
        self.fc1.reset_parameters()
        self.fc2.reset_parameters()

    def forward(self, x):
        # TaskDecoder.py:14-17; LeakyReLU is fused into the first GEMM's epilogue
        h = A.linear(x, self.fc1.weight, self.fc1.bias, act=K.ACT_LEAKY, alpha=self.act.negative_slope)
        return A.linear(h, self.fc2.weight, self.fc2.bias)
