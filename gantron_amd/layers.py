"""Parameter containers with the reference's names and initialisers (reference layers.py:9-73).

Only construction matters here: sub-module names fix the state_dict keys
(`...linear_layer.weight`, `...conv.weight`, `discriminator.N.module.0.weight`) and the
xavier-uniform gains fix the initial values.  The modules are never called: the arithmetic of the
hot path lives in the HIP kernels (engine.py)."""
import torch
from torch import nn


class LinearNorm(nn.Module):
    def __init__(self, in_dim, out_dim, bias=True, w_init_gain='linear'):
        super().__init__()
        self.linear_layer = nn.Linear(in_dim, out_dim, bias=bias)
        nn.init.xavier_uniform_(self.linear_layer.weight, gain=nn.init.calculate_gain(w_init_gain))


class ConvNorm(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, dilation=1, bias=True,
                 w_init_gain='linear'):
        super().__init__()
        if padding is None:
            assert kernel_size % 2 == 1
            padding = int(dilation * (kernel_size - 1) / 2)
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              dilation=dilation, bias=bias)
        nn.init.xavier_uniform_(self.conv.weight, gain=nn.init.calculate_gain(w_init_gain))


class DiscConv1d(nn.Module):
    """Conv1d -> Dropout(0.5) -> Tanh (reference layers.py:43-60)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=None, dilation=1, bias=True,
                 dropout=0.5):
        super().__init__()
        if padding is None:
            assert kernel_size % 2 == 1
            padding = int(dilation * (kernel_size - 1) / 2)
        conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                         dilation=dilation, bias=bias)
        nn.init.xavier_uniform_(conv.weight, gain=nn.init.calculate_gain('tanh'))
        self.module = nn.Sequential(conv, nn.Dropout(dropout), nn.Tanh())


class DiscDense(nn.Module):
    """Linear -> Dropout(0.5) -> Tanh (reference layers.py:63-74); parameters only, the HIP engine does the arithmetic."""

    def __init__(self, in_dim, out_dim, bias=True, dropout=0.5):
        super().__init__()
        self.module = nn.Sequential(nn.Linear(in_dim, out_dim, bias=bias), nn.Dropout(dropout), nn.Tanh())
