"""ctypes binding of libaa_interp.so (the C-ABI declared in include/aa_interp.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C interpolate_antialiasing_amd/csrc``.
There is NO fallback: if the shared object is missing, loading fails loudly, and every op in this package
needs it (the CPU oracle under oracle/ is test infrastructure and is never imported from here).
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AA_INTERP_LIB: developer A/B runs of an alternative build of the SAME library (tools/ab_build.sh); never a fallback
LIB_PATH = os.environ.get("AA_INTERP_LIB") or os.path.join(_HERE, "csrc", "libaa_interp.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "aa_interp.h")

# enums (include/aa_interp.h)
AA_OK = 0
FILTER_LINEAR, FILTER_CUBIC, FILTER_BOX = 0, 1, 2
U8, F32, F64, F16, BF16 = 0, 1, 2, 3, 4
NCHW, NHWC = 0, 1
TABLE_PIL, TABLE_F32, TABLE_F64 = 0, 1, 2
ERR_BAD_DTYPE = -2
ERR_STRIDES = -10
FLAG_FAST = 1

# every symbol include/aa_interp.h declares (tests check the .so exports exactly these)
EXPORTS = (
    "aa_abi_version", "aa_strerror", "aa_device_count", "aa_table_ksize", "aa_table_bytes", "aa_table_build_bytes", "aa_table_build",
    "aa_table_transposed_ksize", "aa_table_transpose", "aa_table_query", "aa_table_query2", "aa_table_build2", "aa_workspace_bytes", "aa_resample_fwd",
    "aa_resample_bwd", "aa_resample_bwd_atomic", "aa_workspace_bytes_bwd", "aa_resample_axis_fwd", "aa_set_fused",
    "aa_last_variant", "aa_probe_copy", "aa_workspace_bytes_u8_to_f32", "aa_resample_fwd_u8_to_f32", "aa_set_store_form", "aa_set_plane_groups", "aa_resample_fwd_ex", "aa_resample_fwd_strided",
)


class TableHeader(ctypes.Structure):
    _fields_ = [
        ("magic", ctypes.c_int32), ("filter", ctypes.c_int32), ("kind", ctypes.c_int32), ("in_size", ctypes.c_int32),
        ("out_size", ctypes.c_int32), ("ksize", ctypes.c_int32), ("align_corners", ctypes.c_int32),
        ("max_taps", ctypes.c_int32), ("transposed", ctypes.c_int32), ("scatter_off", ctypes.c_int32),
        ("scatter_ksize", ctypes.c_int32), ("scatter_max", ctypes.c_int32), ("span64p1", ctypes.c_int32), ("span4p1", ctypes.c_int32), ("gather_off", ctypes.c_int32), ("reserved", ctypes.c_int32 * 1),
    ]


class Axis(ctypes.Structure):
    _fields_ = [
        ("table_dev", ctypes.c_void_p), ("in_size", ctypes.c_int32), ("out_size", ctypes.c_int32),
        ("ksize", ctypes.c_int32), ("max_taps", ctypes.c_int32), ("kind", ctypes.c_int32), ("filter", ctypes.c_int32),
        ("scatter_off", ctypes.c_int32), ("scatter_ksize", ctypes.c_int32), ("scatter_max", ctypes.c_int32),
        ("span64p1", ctypes.c_int32), ("span4p1", ctypes.c_int32), ("gather_off", ctypes.c_int32), ("reserved", ctypes.c_int32 * 2),
    ]


class Convert(ctypes.Structure):
    _fields_ = [("out_layout", ctypes.c_int32), ("normalize", ctypes.c_int32), ("mean", ctypes.c_float * 4),
                ("std", ctypes.c_float * 4), ("flags", ctypes.c_uint32)]


class AAInterpError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """Load libaa_interp.so, declaring every prototype.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AAInterpError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C interpolate_antialiasing_amd/csrc`). "
            "There is no CPU fallback."
        )
    L = ctypes.CDLL(LIB_PATH)
    i32, i64, dbl, vp, sz = ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t
    ax = ctypes.POINTER(Axis)
    L.aa_abi_version.restype = i32
    L.aa_strerror.argtypes = [i32]
    L.aa_strerror.restype = ctypes.c_char_p
    L.aa_device_count.restype = i32
    L.aa_table_ksize.argtypes = [i32, i32, i64, i64, i32, dbl]
    L.aa_table_ksize.restype = i32
    L.aa_table_bytes.argtypes = [i32, i64, i32]
    L.aa_table_bytes.restype = sz
    L.aa_table_build_bytes.argtypes = [i32, i32, i64, i64, i32, dbl]
    L.aa_table_build_bytes.restype = sz
    L.aa_table_build.argtypes = [i32, i32, i64, i64, i32, dbl, vp, sz, vp]
    L.aa_table_build.restype = i32
    L.aa_table_transposed_ksize.argtypes = [i32, i32, i64, i64, i32, dbl]
    L.aa_table_transposed_ksize.restype = i32
    L.aa_table_transpose.argtypes = [vp, vp, sz, i32, vp]
    L.aa_table_transpose.restype = i32
    L.aa_table_query.argtypes = [vp, ctypes.POINTER(TableHeader), vp]
    L.aa_table_query.restype = i32
    L.aa_table_query2.argtypes = [vp, vp, ctypes.POINTER(TableHeader), ctypes.POINTER(TableHeader), vp]
    L.aa_table_query2.restype = i32
    L.aa_table_build2.argtypes = [i32, i32, i32, i64, i64, ctypes.c_double, vp, sz, i64, i64, ctypes.c_double, vp, sz, vp]
    L.aa_table_build2.restype = i32
    L.aa_workspace_bytes.argtypes = [i32, i32, i64, i64, i64, i64, i64, i64, ax, ax]
    L.aa_workspace_bytes.restype = sz
    L.aa_resample_fwd.argtypes = [vp, vp, vp, sz, i32, i32, i64, i64, i64, i64, ax, ax, vp]
    L.aa_resample_fwd.restype = i32
    L.aa_resample_fwd_ex.argtypes = [vp, vp, vp, sz, i32, i32, i64, i64, i64, i64, ax, ax, ctypes.c_uint, vp]
    L.aa_resample_fwd_ex.restype = i32
    L.aa_resample_fwd_strided.argtypes = [vp, vp, i32, i32, i64, i64, i64, i64, ctypes.POINTER(ctypes.c_int64), ax, ax, ctypes.c_uint, vp]
    L.aa_resample_fwd_strided.restype = i32
    L.aa_resample_bwd.argtypes = [vp, vp, vp, sz, i32, i32, i64, i64, i64, i64, ax, ax, vp]
    L.aa_resample_bwd.restype = i32
    L.aa_resample_bwd_atomic.argtypes = [vp, vp, vp, sz, i32, i32, i64, i64, i64, i64, ax, ax, vp]
    L.aa_resample_bwd_atomic.restype = i32
    L.aa_workspace_bytes_bwd.argtypes = [i32, i32, i64, i64, i64, i64, i64, i64]
    L.aa_workspace_bytes_bwd.restype = sz
    L.aa_resample_axis_fwd.argtypes = [vp, vp, i32, i64, i64, i64, ax, vp]
    L.aa_resample_axis_fwd.restype = i32
    L.aa_last_variant.restype = ctypes.c_char_p
    cvp = ctypes.POINTER(Convert)
    L.aa_workspace_bytes_u8_to_f32.argtypes = [i32, i64, i64, i64, i64, ax, ax, cvp]
    L.aa_workspace_bytes_u8_to_f32.restype = sz
    L.aa_resample_fwd_u8_to_f32.argtypes = [vp, vp, vp, sz, i32, i64, i64, i64, i64, ax, ax, cvp, vp]
    L.aa_resample_fwd_u8_to_f32.restype = i32
    L.aa_probe_copy.argtypes = [vp, vp, sz, i32, vp]
    L.aa_probe_copy.restype = i32
    L.aa_set_fused.argtypes = [i32]
    L.aa_set_fused.restype = i32
    L.aa_set_store_form.argtypes = [i32]
    L.aa_set_store_form.restype = i32
    L.aa_set_plane_groups.argtypes = [i32]
    L.aa_set_plane_groups.restype = i32
    if L.aa_abi_version() != 3:
        raise AAInterpError("libaa_interp.so ABI version mismatch")
    _lib = L
    return L


def strerror(rc: int) -> str:
    return load().aa_strerror(rc).decode()


def check(rc: int, what: str = "") -> None:
    if rc < 0:
        msg = f"{what}: {strerror(rc)} (aa_status {rc})" if what else f"{strerror(rc)} (aa_status {rc})"
        if rc == ERR_BAD_DTYPE:
            raise NotImplementedError(msg)
        raise AAInterpError(msg)


fused_epoch = 0  # bumped by set_fused: host-side plans that cached a workspace size are keyed on it


def set_fused(enabled: int) -> int:
    """Enable/disable the fused kernels (process-wide); returns the previous setting."""
    global fused_epoch
    fused_epoch += 1
    return int(load().aa_set_fused(int(enabled)))


def set_store_form(form: int) -> int:
    """Test hook: -1 automatic, 0 never / 1 always the streaming store forms of the up-scaling / backward kernel; returns the previous setting."""
    return int(load().aa_set_store_form(int(form)))


def set_plane_groups(enabled: int) -> int:
    """Test / A-B hook: 1 (default) planar three-channel uint8 images run all three planes in one wave, 0 one wave per plane; returns the
    previous setting."""
    return int(load().aa_set_plane_groups(int(enabled)))


def last_variant() -> str:
    return load().aa_last_variant().decode()


# sources a kernel variant is compiled from (csrc/): profiles/*.json summaries are stamped with their fingerprint, and bench.py
# flags a committed counter summary as stale when the sources of the kernel it describes have changed since
KERNEL_SOURCES = {
    "fused_u8_nhwc_pil_v3": ("aa_fused_u8_v3_impl.h", "aa_fused_u8_v3.hip", "aa_fused_u8_v3_c3.hip", "aa_common.h"),
    "fused_u8_nhwc_pil": ("aa_fused_u8.hip", "aa_common.h"),
    "fused_u8_planar_pil_v3": ("aa_fused_u8_v3_impl.h", "aa_fused_u8_v3.hip", "aa_fused_u8_v3_c3g.hip", "aa_fused_u8_v3_c1.hip", "aa_common.h"),
}


def source_fingerprint(variant: str):
    """sha256[:12] over the source files of `variant` (None for a variant without an entry in KERNEL_SOURCES)."""
    import hashlib

    names = KERNEL_SOURCES.get(variant)
    if not names:
        return None
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(_HERE, "csrc", n), "rb") as f:
            h.update(n.encode() + b"\0" + f.read())
    return h.hexdigest()[:12]
