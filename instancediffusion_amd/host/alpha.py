"""Host mirror of the two sampling-path functions of ``utils/model.py``: the per-step alpha gate."""
from __future__ import annotations

import numpy as np

from .attention import GatedSelfAttentionDense


def set_alpha_scale(model, alpha_scale):
    """utils/model.py:78-81 -- exact-type match on the fuser class, writes ``.scale``."""
    for module in model.modules():
        if type(module) == GatedSelfAttentionDense:
            module.scale = alpha_scale


def alpha_generator(length, type=None):
    """utils/model.py:83-117 -- [alpha=1 stage | linear decay stage | alpha=0 stage], fractions in ``type``."""
    if type is None:
        type = [1, 0, 0]
    assert len(type) == 3
    assert type[0] + type[1] + type[2] == 1
    stage0 = int(type[0] * length)
    stage1 = int(type[1] * length)
    stage2 = length - stage0 - stage1
    decay = list(np.arange(start=0, stop=1, step=1 / stage1)[::-1]) if stage1 != 0 else []
    alphas = [1] * stage0 + decay + [0] * stage2
    assert len(alphas) == length
    return alphas
