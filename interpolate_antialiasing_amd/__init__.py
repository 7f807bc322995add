"""interpolate_antialiasing_amd — MI355X-native antialiased separable resample (PIL-style), drop-in for the
hot path of vfdev-5/interpolate-antialiasing.

    from interpolate_antialiasing_amd import extension_interpolate as aa_interp
    y = aa_interp.linear_forward(x_gpu, [196, 320], False)

Python here is plumbing (device memory, streams, torch.distributed); the product is the HIP library
``csrc/libaa_interp.so`` behind the C-ABI in ``include/aa_interp.h``.
"""
from . import _lib, tables, sharding  # noqa: F401
from . import extension_interpolate  # noqa: F401
from .functional import interpolate_aa  # noqa: F401

__version__ = "0.1.0"
