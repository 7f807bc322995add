"""Differentiable convenience wrapper (the counterpart of test.py:122-158's autograd.Function + Module)."""
from __future__ import annotations

from typing import Sequence

import torch

_MODES = {"bilinear": "linear_forward", "linear": "linear_forward", "bicubic": "cubic_forward",
          "cubic": "cubic_forward", "nearest": "nearest_forward", "box": "nearest_forward"}


def interpolate_aa(input: torch.Tensor, size: Sequence[int], mode: str = "bilinear", align_corners: bool = False) -> torch.Tensor:
    """Antialiased resize of a 4-D GPU tensor to ``size`` = (H, W); differentiable for float dtypes.
    ``mode``: bilinear | bicubic | nearest (= box filter, as in the reference).  3-D (NCL) and 5-D (NCDHW) inputs take
    the N-d front-ends (forward only): ``mode`` linear/bilinear/trilinear | bicubic | nearest."""
    if input.dim() in (3, 5):
        from . import extension_interpolate as ext

        fn = {"linear": ext.linear_forward_nd, "bilinear": ext.linear_forward_nd, "trilinear": ext.linear_forward_nd,
              "bicubic": ext.cubic_forward_nd, "cubic": ext.cubic_forward_nd, "nearest": ext.nearest_forward_nd,
              "box": ext.nearest_forward_nd}.get(mode)
        if fn is None:
            raise ValueError(mode)
        return fn(input, [int(v) for v in size], bool(align_corners))
    if mode not in _MODES:
        raise ValueError(mode)  # test.py:78-79
    op = getattr(torch.ops.extension_interpolate, _MODES[mode])
    return op(input, [int(size[0]), int(size[1])], bool(align_corners))
