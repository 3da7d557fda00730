"""Drop-in for the reference's model/generators.py:4-19."""
import torch.nn as nn

from .. import ops


class Generator(nn.Module):

    def __init__(self, d_model, voc_size):
        super(Generator, self).__init__()
        self.linear = nn.Linear(d_model, voc_size)
        print('Using vanilla Generator')

    def forward(self, x):
        """decoder states (B, T_c, d_caps) -> log-softmax over the vocabulary (B, T_c, V)"""
        return ops.generator(x, self.linear.weight, self.linear.bias)
