"""Differentiable convenience wrapper (the counterpart of test.py:122-158's autograd.Function + Module)."""
from __future__ import annotations

from typing import Sequence

import torch

_MODES = {"bilinear": "linear_forward", "linear": "linear_forward", "bicubic": "cubic_forward",
          "cubic": "cubic_forward", "nearest": "nearest_forward", "box": "nearest_forward"}


def interpolate_aa(input: torch.Tensor, size: Sequence[int], mode: str = "bilinear", align_corners: bool = False) -> torch.Tensor:
    """Antialiased resize of a 4-D GPU tensor to ``size`` = (H, W); differentiable for float dtypes.
    ``mode``: bilinear | bicubic | nearest (= box filter, as in the reference)."""
    if mode not in _MODES:
        raise ValueError(mode)  # test.py:78-79
    op = getattr(torch.ops.extension_interpolate, _MODES[mode])
    return op(input, [int(size[0]), int(size[1])], bool(align_corners))
