"""CPU ORACLE bindings — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker.  The product (``interpolate_antialiasing_amd``) never imports it.

``liboracle.so`` is our plain-C restatement of the reference algorithm (``oracle/aa_oracle.c``; every
function there cites the reference file:line it follows).  ``oracle/_ref/*.so`` are the reference's own
C++ sources compiled in the build container (``make -C oracle ref``); :func:`load_ref` imports them when
they are present (they travel to the GPU box as prebuilt files; the reference sources do not).
"""
from __future__ import annotations

import ctypes
import importlib.util
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

FILTER_LINEAR, FILTER_CUBIC, FILTER_BOX = 0, 1, 2
FILTERS = {"linear": 0, "bilinear": 0, "cubic": 1, "bicubic": 1, "box": 2, "nearest": 2}

_lib = None


def build(force: bool = False) -> str:
    """Compile liboracle.so (gcc, a second or two)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
        os.path.join(_HERE, "aa_oracle.c")
    ):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, i32, dbl, vp = ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p
        L.aao_ksize.argtypes = [i32, i64, i64, i32, dbl, i32]
        L.aao_ksize.restype = i32
        for name in ("aao_weights_f32", "aao_weights_f64"):
            f = getattr(L, name)
            f.argtypes = [i32, i64, i64, i32, dbl, vp, vp, vp]
            f.restype = i32
        for name in ("aao_forward_f32", "aao_forward_f64"):
            f = getattr(L, name)
            f.argtypes = [i32, vp, vp, i64, i64, i64, i64, i64, i64, vp, vp, i32, i32]
            f.restype = i32
        for name in ("aao_backward_f32", "aao_backward_f64"):
            f = getattr(L, name)
            f.argtypes = [i32, vp, vp, i64, i64, i64, i64, i64, i64, i32]
            f.restype = i32
        L.aao_legacy_nonaa_linear_backward_f32.argtypes = [vp, vp, i64, i64, i64, i64, i64, i64, i32]
        L.aao_legacy_nonaa_linear_backward_f32.restype = i32
        L.aao_pil_ksize.argtypes = [i32, i64, i64]
        L.aao_pil_ksize.restype = i32
        L.aao_pil_coeffs.argtypes = [i32, i64, i64, vp, vp, vp, vp]
        L.aao_pil_coeffs.restype = i32
        for name in ("aao_pil_resize_u8", "aao_harness_u8"):
            f = getattr(L, name)
            f.argtypes = [i32, vp, vp, i64, i64, i64, i64, i64, i64, vp, vp, i32]
            f.restype = i32
        L.aao_max_threads.restype = i32
        _lib = L
    return _lib


def _filter_id(f) -> int:
    return FILTERS[f] if isinstance(f, str) else int(f)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _strides_elems(a: np.ndarray) -> np.ndarray:
    return np.asarray([s // a.itemsize for s in a.strides], dtype=np.int64)


def max_threads() -> int:
    return int(lib().aao_max_threads())


def ksize(filt, in_size: int, out_size: int, align_corners: bool = False, dtype=np.float32) -> int:
    return int(lib().aao_ksize(_filter_id(filt), in_size, out_size, int(align_corners), 0.0, int(np.dtype(dtype) == np.float64)))


def weights(filt, in_size: int, out_size: int, align_corners: bool = False, dtype=np.float32, scale: float = 0.0):
    """-> (ksize, xmin[int64 out], xsize[int64 out], w[out, ksize])  (s2.2/aa_interpolation_impl.h:195-281)."""
    dtype = np.dtype(dtype)
    fid = _filter_id(filt)
    k = int(lib().aao_ksize(fid, in_size, out_size, int(align_corners), float(scale), int(dtype == np.float64)))
    xmin = np.zeros(out_size, np.int64)
    xsize = np.zeros(out_size, np.int64)
    w = np.zeros((out_size, k), dtype)
    fn = lib().aao_weights_f64 if dtype == np.float64 else lib().aao_weights_f32
    k2 = fn(fid, in_size, out_size, int(align_corners), float(scale), _ptr(xmin), _ptr(xsize), _ptr(w))
    assert k2 == k, (k2, k)
    return k, xmin, xsize, w


def forward(filt, x: np.ndarray, out_hw: Sequence[int], align_corners: bool = False, channels_last_out: Optional[bool] = None,
            nthreads: int = 1) -> np.ndarray:
    """fp32/fp64 forward on an NCHW-shaped array with arbitrary strides (a channels_last tensor is passed as the
    NCHW *view* of NHWC storage).  Output memory format follows the input (s2.2:752) unless overridden."""
    assert x.ndim == 4 and x.dtype in (np.float32, np.float64)
    N, C, H, W = x.shape
    oH, oW = int(out_hw[0]), int(out_hw[1])
    if channels_last_out is None:
        channels_last_out = _is_channels_last(x)
    out = _empty_like_format(N, C, oH, oW, x.dtype, channels_last_out)
    fn = lib().aao_forward_f64 if x.dtype == np.float64 else lib().aao_forward_f32
    is_, os_ = _strides_elems(x), _strides_elems(out)
    rc = fn(_filter_id(filt), _ptr(x), _ptr(out), N, C, H, W, oH, oW, _ptr(is_), _ptr(os_), int(align_corners), nthreads)
    if rc != 0:
        raise RuntimeError(f"oracle forward failed rc={rc}")
    return out


def backward(filt, grad_out: np.ndarray, in_hw: Sequence[int], align_corners: bool = False) -> np.ndarray:
    """True adjoint of :func:`forward` (contiguous NCHW)."""
    go = np.ascontiguousarray(grad_out)
    N, C, oH, oW = go.shape
    H, W = int(in_hw[0]), int(in_hw[1])
    gi = np.zeros((N, C, H, W), go.dtype)
    fn = lib().aao_backward_f64 if go.dtype == np.float64 else lib().aao_backward_f32
    rc = fn(_filter_id(filt), _ptr(go), _ptr(gi), N, C, H, W, oH, oW, int(align_corners))
    if rc != 0:
        raise RuntimeError(f"oracle backward failed rc={rc}")
    return gi


def legacy_nonaa_linear_backward(grad_out: np.ndarray, in_hw: Sequence[int], align_corners: bool = False) -> np.ndarray:
    go = np.ascontiguousarray(grad_out, dtype=np.float32)
    N, C, oH, oW = go.shape
    H, W = int(in_hw[0]), int(in_hw[1])
    gi = np.zeros((N, C, H, W), np.float32)
    rc = lib().aao_legacy_nonaa_linear_backward_f32(_ptr(go), _ptr(gi), N, C, H, W, oH, oW, int(align_corners))
    if rc != 0:
        raise RuntimeError(f"oracle legacy backward failed rc={rc}")
    return gi


def pil_coeffs(filt, in_size: int, out_size: int):
    """-> (ksize, xmin[int32], xsize[int32], kk[out,ksize] int32 22-bit fixed point, prekk[out,ksize] float64)."""
    fid = _filter_id(filt)
    k = int(lib().aao_pil_ksize(fid, in_size, out_size))
    xmin = np.zeros(out_size, np.int32)
    xsize = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, k), np.int32)
    pre = np.zeros((out_size, k), np.float64)
    k2 = lib().aao_pil_coeffs(fid, in_size, out_size, _ptr(xmin), _ptr(xsize), _ptr(kk), _ptr(pre))
    assert k2 == k
    return k, xmin, xsize, kk, pre


def _u8_call(fn, filt, x: np.ndarray, out_hw, channels_last_out, nthreads):
    assert x.ndim == 4 and x.dtype == np.uint8
    N, C, H, W = x.shape
    oH, oW = int(out_hw[0]), int(out_hw[1])
    if channels_last_out is None:
        channels_last_out = _is_channels_last(x)
    out = _empty_like_format(N, C, oH, oW, np.uint8, channels_last_out)
    is_, os_ = _strides_elems(x), _strides_elems(out)
    rc = fn(_filter_id(filt), _ptr(x), _ptr(out), N, C, H, W, oH, oW, _ptr(is_), _ptr(os_), nthreads)
    if rc != 0:
        raise RuntimeError(f"oracle u8 call failed rc={rc}")
    return out


def pil_resize_u8(filt, x: np.ndarray, out_hw, channels_last_out: Optional[bool] = None, nthreads: int = 1) -> np.ndarray:
    """uint8 with Pillow semantics on an NCHW-shaped (any strides) array."""
    return _u8_call(lib().aao_pil_resize_u8, filt, x, out_hw, channels_last_out, nthreads)


def harness_u8(filt, x: np.ndarray, out_hw, channels_last_out: Optional[bool] = None, nthreads: int = 1) -> np.ndarray:
    """uint8 through the reference harness: float() -> fp32 op -> (bicubic clamp) -> truncating byte()."""
    return _u8_call(lib().aao_harness_u8, filt, x, out_hw, channels_last_out, nthreads)


def _is_channels_last(x: np.ndarray) -> bool:
    N, C, H, W = x.shape
    if C == 1:
        return False
    return x.strides == (H * W * C * x.itemsize, x.itemsize, W * C * x.itemsize, C * x.itemsize)


def _empty_like_format(N, C, H, W, dtype, channels_last: bool) -> np.ndarray:
    if channels_last:
        return np.zeros((N, H, W, C), dtype).transpose(0, 3, 1, 2)
    return np.zeros((N, C, H, W), dtype)


def as_channels_last(x: np.ndarray) -> np.ndarray:
    """NCHW-shaped view over freshly allocated NHWC storage holding the same values."""
    return np.ascontiguousarray(x.transpose(0, 2, 3, 1)).transpose(0, 3, 1, 2)


def load_ref(name: str = "ref_s22"):
    """Import one of the reference's own compiled modules from oracle/_ref (None when absent).
    ref_s22: linear_forward/cubic_forward/nearest_forward/linear_backward; ref_s3 / ref_s3sep: forward."""
    path = os.path.join(_HERE, "_ref", name + ".so")
    if not os.path.exists(path):
        return None
    import torch  # noqa: F401  (the modules link against libtorch)

    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
