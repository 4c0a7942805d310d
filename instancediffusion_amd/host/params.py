"""Parameter containers for the host-side model mirror.

The HIP engine owns all arithmetic; these ``nn.Module``s exist only so that the model exposes the reference's
state-dict key names, ``.modules()`` traversal, ``.to()/.eval()/.load_state_dict()``.
They have NO forward().
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn as nn


class Dense(nn.Module):
    """weight [out, in] (+ bias [out]) -- the layout of nn.Linear / the reference ``linear()``."""

    def __init__(self, n_in: int, n_out: int, bias: bool = True, zero: bool = False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_out, n_in))
        self.bias = nn.Parameter(torch.empty(n_out)) if bias else None
        self._zero = zero
        self.reset_parameters()

    def reset_parameters(self):
        _init_(self.weight, self.bias, self._zero)


class Conv(nn.Module):
    """weight [Cout, Cin/groups, k, k] + bias [Cout] -- the layout of nn.Conv2d."""

    def __init__(self, c_in: int, c_out: int, k: int, zero: bool = False, groups: int = 1):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c_out, c_in // groups, k, k))
        self.bias = nn.Parameter(torch.empty(c_out))
        self._zero = zero
        self.reset_parameters()

    def reset_parameters(self):
        _init_(self.weight, self.bias, self._zero)


class Affine(nn.Module):
    """weight/bias [C] of a GroupNorm / LayerNorm."""

    def __init__(self, c: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


def _init_(w: torch.Tensor, b, zero: bool):
    if w.device.type == "meta":
        return
    with torch.no_grad():
        if zero:
            w.zero_()
            if b is not None:
                b.zero_()
            return
        fan_in = 1
        for s in w.shape[1:]:
            fan_in *= s
        bound = 1.0 / math.sqrt(max(fan_in, 1))
        w.uniform_(-bound, bound)
        if b is not None:
            b.uniform_(-bound, bound)


class Slots(nn.Module):
    """Children registered under integer names at chosen indices (``in_layers.0``, ``in_layers.2`` ...), the way
    an ``nn.Sequential`` with parameter-free layers in between would name them.  Indexable like a Sequential."""

    def __init__(self, children: Dict[int, nn.Module]):
        super().__init__()
        for i, m in children.items():
            self.add_module(str(i), m)

    def __getitem__(self, i: int) -> nn.Module:
        return self._modules[str(i)]

    def __setitem__(self, i: int, m: nn.Module):
        self._modules[str(i)] = m

    def __len__(self):
        return len(self._modules)

    def items(self):
        return [(int(k), v) for k, v in self._modules.items()]


def mlp3(n_in: int, n_mid: int, n_out: int) -> Slots:
    """Linear-SiLU-Linear-SiLU-Linear at indices 0/2/4 (text_grounding_net.py:73-81)."""
    return Slots({0: Dense(n_in, n_mid), 2: Dense(n_mid, n_mid), 4: Dense(n_mid, n_out)})
