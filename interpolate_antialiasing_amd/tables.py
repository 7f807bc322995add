"""Weight tables: device-side precompute, host-side cache, pack/unpack for the one RCCL broadcast.

A table replaces what HelperInterpBase::_compute_indices_weights_aa returns (reference
step_two_dot_two/aa_interpolation_impl.h:195-281: xmin, size, stride, weights, weight-index tensors, recomputed
on every call and every pass).  Here it is ONE packed device buffer per (filter, kind, in, out, align_corners,
scale, device), built once by a HIP kernel and cached.
"""
from __future__ import annotations

import ctypes
import threading
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib

FILTER_IDS = {"linear": _lib.FILTER_LINEAR, "bilinear": _lib.FILTER_LINEAR, "cubic": _lib.FILTER_CUBIC,
              "bicubic": _lib.FILTER_CUBIC, "box": _lib.FILTER_BOX, "nearest": _lib.FILTER_BOX}
KIND_IDS = {"pil": _lib.TABLE_PIL, "f32": _lib.TABLE_F32, "f64": _lib.TABLE_F64}
HEADER_BYTES = ctypes.sizeof(_lib.TableHeader)  # 64

# fixed-length int64 descriptor broadcast ahead of the payload so receivers can allocate
META_LEN = 16


def _stream_ptr(device: torch.device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


@dataclass
class WeightTable:
    """One packed table.  ``buf`` is a uint8 tensor: on the GPU when used by kernels, possibly on the CPU while
    it is only being transported (tests with the gloo backend)."""
    buf: torch.Tensor
    filter: int
    kind: int
    in_size: int
    out_size: int
    ksize: int
    max_taps: int
    align_corners: bool = False
    transposed: bool = False
    scatter_off: int = 0
    scatter_ksize: int = 0
    scatter_max: int = 0
    span64p1: int = 0  # 1 + the widest spread of 64 consecutive outputs' window starts (measured on device; 0 = unknown)
    span4p1: int = 0   # the same over 4 consecutive outputs
    gather_off: int = 0  # byte offset of the per-output gather records (F32 and Pillow tables)

    def axis(self) -> _lib.Axis:
        if not self.buf.is_cuda:
            raise _lib.AAInterpError("weight table is not on a GPU")
        return _lib.Axis(ctypes.c_void_p(self.buf.data_ptr()), self.in_size, self.out_size, self.ksize, self.max_taps,
                         self.kind, self.filter, self.scatter_off, self.scatter_ksize, self.scatter_max, self.span64p1, self.span4p1, self.gather_off)

    # ---- transport -----------------------------------------------------------------------------------
    def meta(self) -> torch.Tensor:
        return torch.tensor([0x42544141, self.filter, self.kind, self.in_size, self.out_size, self.ksize,
                             self.max_taps, int(self.align_corners), int(self.transposed), self.buf.numel(),
                             self.scatter_off, self.scatter_ksize, self.scatter_max, self.span64p1, self.span4p1, self.gather_off],
                            dtype=torch.int64)

    @staticmethod
    def from_meta(meta: torch.Tensor, buf: torch.Tensor) -> "WeightTable":
        m = [int(v) for v in meta.tolist()]
        if m[0] != 0x42544141 or len(m) != META_LEN:
            raise _lib.AAInterpError("bad weight-table descriptor")
        if buf.numel() != m[9]:
            raise _lib.AAInterpError("weight-table payload size mismatch")
        return WeightTable(buf=buf, filter=m[1], kind=m[2], in_size=m[3], out_size=m[4], ksize=m[5], max_taps=m[6],
                           align_corners=bool(m[7]), transposed=bool(m[8]), scatter_off=m[10], scatter_ksize=m[11],
                           scatter_max=m[12], span64p1=m[13], span4p1=m[14], gather_off=m[15])

    # ---- inspection (tests) ----------------------------------------------------------------------------
    def unpack_scatter(self):
        """-> (first int32[in], count int32[in], w [in, 6], completes int32[in]) (CPU numpy).  Record layout by table kind
        (include/aa_interp.h): PIL / F32 tables 32 bytes {int32 first, int32 count | completes << 16, int32 / float w[6]};
        F64 tables 64 bytes {int32 first, int32 count | completes << 16, double w[6], 8 bytes padding}."""
        import numpy as np

        if not self.scatter_off:
            return None
        raw = self.buf.detach().cpu().numpy()
        n, off = self.in_size, self.scatter_off
        if self.kind == _lib.TABLE_F64:
            blk = raw[off:off + 64 * n].reshape(n, 64)
            head = blk[:, :8].copy().view(np.int32).reshape(n, 2)
            w = blk[:, 8:56].copy().view(np.float64).reshape(n, 6)
            return head[:, 0], head[:, 1] & 0xFFFF, w, head[:, 1] >> 16
        rec = raw[off:off + 32 * n].view(np.int32).reshape(n, 8).copy()
        w = rec[:, 2:] if self.kind == _lib.TABLE_PIL else rec[:, 2:].copy().view(np.float32)
        return rec[:, 0], rec[:, 1] & 0xFFFF, w, rec[:, 1] >> 16

    def unpack(self):
        """-> (xmin int32[out], xsize int32[out], w [out,ksize]) as CPU numpy arrays."""
        import numpy as np

        raw = self.buf.detach().cpu().numpy()
        out, k = self.out_size, self.ksize
        xmin = raw[HEADER_BYTES:HEADER_BYTES + 4 * out].view(np.int32).copy()
        xsize = raw[HEADER_BYTES + 4 * out:HEADER_BYTES + 8 * out].view(np.int32).copy()
        woff = (HEADER_BYTES + 8 * out + 15) & ~15
        wdt = {_lib.TABLE_PIL: np.int32, _lib.TABLE_F32: np.float32, _lib.TABLE_F64: np.float64}[self.kind]
        w = raw[woff:woff + out * k * np.dtype(wdt).itemsize].view(wdt).reshape(out, k).copy()
        return xmin, xsize, w


_cache: Dict[Tuple, WeightTable] = {}
_cache_lock = threading.Lock()


def clear_cache() -> None:
    with _cache_lock:
        _cache.clear()


def cache_key(filter_id: int, kind: int, in_size: int, out_size: int, align_corners: bool, scale: float,
              device: torch.device, transposed: bool = False) -> Tuple:
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else -1)
    return (int(filter_id), int(kind), int(in_size), int(out_size), bool(align_corners), float(scale or 0.0),
            dev.type, idx, bool(transposed))


def build_table(filter_id: int, kind: int, in_size: int, out_size: int, align_corners: bool, scale: float,
                device: torch.device) -> WeightTable:
    """Uncached device-side build (one launch + one 64-byte header read-back)."""
    L = _lib.load()
    k = L.aa_table_ksize(filter_id, kind, in_size, out_size, int(align_corners), float(scale or 0.0))
    _lib.check(k, "aa_table_ksize")
    nbytes = L.aa_table_build_bytes(filter_id, kind, in_size, out_size, int(align_corners), float(scale or 0.0))
    with torch.cuda.device(device):
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        s = _stream_ptr(device)
        _lib.check(L.aa_table_build(filter_id, kind, in_size, out_size, int(align_corners), float(scale or 0.0),
                                    ctypes.c_void_p(buf.data_ptr()), nbytes, s), "aa_table_build")
        hdr = _lib.TableHeader()
        _lib.check(L.aa_table_query(ctypes.c_void_p(buf.data_ptr()), ctypes.byref(hdr), s), "aa_table_query")
    return WeightTable(buf, filter_id, kind, in_size, out_size, k, int(hdr.max_taps), bool(align_corners), False,
                       int(hdr.scatter_off), int(hdr.scatter_ksize), int(hdr.scatter_max), int(hdr.span64p1), int(hdr.span4p1), int(hdr.gather_off))


def _from_header(buf, filter_id, kind, in_size, out_size, k, align_corners, hdr) -> WeightTable:
    return WeightTable(buf, filter_id, kind, in_size, out_size, k, int(hdr.max_taps), bool(align_corners), False,
                       int(hdr.scatter_off), int(hdr.scatter_ksize), int(hdr.scatter_max), int(hdr.span64p1), int(hdr.span4p1), int(hdr.gather_off))


def get_table_pair(filter_id: int, kind: int, in_h: int, out_h: int, in_w: int, out_w: int, align_corners: bool, scale_h: float, scale_w: float,
                   device: torch.device):
    """The two tables of a 2-D call.  When BOTH are new (a shape never seen: every call of a random-crop pipeline) they are built
    back to back and their headers read with one synchronisation instead of two."""
    device = torch.device(device)
    kh = cache_key(filter_id, kind, in_h, out_h, align_corners, scale_h, device)
    kw = cache_key(filter_id, kind, in_w, out_w, align_corners, scale_w, device)
    with _cache_lock:
        th, tw = _cache.get(kh), _cache.get(kw)
    if th is None and tw is None and kh != kw:
        L = _lib.load()
        with torch.cuda.device(device):
            k_h = L.aa_table_ksize(filter_id, kind, in_h, out_h, int(align_corners), float(scale_h or 0.0))
            _lib.check(k_h, "aa_table_ksize")
            k_w = L.aa_table_ksize(filter_id, kind, in_w, out_w, int(align_corners), float(scale_w or 0.0))
            _lib.check(k_w, "aa_table_ksize")
            nb_h = L.aa_table_build_bytes(filter_id, kind, in_h, out_h, int(align_corners), float(scale_h or 0.0))
            nb_w = L.aa_table_build_bytes(filter_id, kind, in_w, out_w, int(align_corners), float(scale_w or 0.0))
            bh = torch.empty(nb_h, dtype=torch.uint8, device=device)
            bw = torch.empty(nb_w, dtype=torch.uint8, device=device)
            _lib.check(L.aa_table_build2(filter_id, kind, int(align_corners), in_h, out_h, float(scale_h or 0.0), ctypes.c_void_p(bh.data_ptr()), nb_h,
                                         in_w, out_w, float(scale_w or 0.0), ctypes.c_void_p(bw.data_ptr()), nb_w, _stream_ptr(device)), "aa_table_build2")
            hh, hw = _lib.TableHeader(), _lib.TableHeader()
            _lib.check(L.aa_table_query2(ctypes.c_void_p(bh.data_ptr()), ctypes.c_void_p(bw.data_ptr()), ctypes.byref(hh), ctypes.byref(hw),
                                         _stream_ptr(device)), "aa_table_query2")
        th = _from_header(bh, filter_id, kind, in_h, out_h, k_h, align_corners, hh)
        tw = _from_header(bw, filter_id, kind, in_w, out_w, k_w, align_corners, hw)
        with _cache_lock:
            _cache[kh], _cache[kw] = th, tw
        return th, tw
    if th is None:
        th = get_table(filter_id, kind, in_h, out_h, align_corners, scale_h, device)
    if tw is None:
        tw = get_table(filter_id, kind, in_w, out_w, align_corners, scale_w, device)
    return th, tw


def get_table(filter_id: int, kind: int, in_size: int, out_size: int, align_corners: bool = False, scale: float = 0.0,
              device: Optional[torch.device] = None) -> WeightTable:
    device = torch.device(device if device is not None else "cuda")
    key = cache_key(filter_id, kind, in_size, out_size, align_corners, scale, device)
    with _cache_lock:
        t = _cache.get(key)
    if t is None:
        t = build_table(filter_id, kind, in_size, out_size, align_corners, scale, device)
        with _cache_lock:
            _cache[key] = t
    return t


def put_table(t: WeightTable, scale: float = 0.0) -> None:
    """Install a table received from another rank into this rank's cache."""
    key = cache_key(t.filter, t.kind, t.in_size, t.out_size, t.align_corners, scale, t.buf.device, t.transposed)
    with _cache_lock:
        _cache[key] = t


def get_transposed_table(fwd: WeightTable, scale: float = 0.0) -> WeightTable:
    """Adjoint (gather-form) table of ``fwd``: in/out swapped, built on device, cached."""
    device = fwd.buf.device
    key = cache_key(fwd.filter, fwd.kind, fwd.in_size, fwd.out_size, fwd.align_corners, scale, device, True)
    with _cache_lock:
        t = _cache.get(key)
    if t is not None:
        return t
    L = _lib.load()
    tk = L.aa_table_transposed_ksize(fwd.filter, fwd.kind, fwd.in_size, fwd.out_size, int(fwd.align_corners), float(scale or 0.0))
    _lib.check(tk, "aa_table_transposed_ksize")
    nbytes = L.aa_table_bytes(fwd.kind, fwd.in_size, tk)
    with torch.cuda.device(device):
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        s = _stream_ptr(device)
        _lib.check(L.aa_table_transpose(ctypes.c_void_p(fwd.buf.data_ptr()), ctypes.c_void_p(buf.data_ptr()), nbytes, tk, s),
                   "aa_table_transpose")
        hdr = _lib.TableHeader()
        _lib.check(L.aa_table_query(ctypes.c_void_p(buf.data_ptr()), ctypes.byref(hdr), s), "aa_table_query")
    t = WeightTable(buf, fwd.filter, fwd.kind, fwd.out_size, fwd.in_size, tk, int(hdr.max_taps), fwd.align_corners, True,
                    span64p1=int(hdr.span64p1), span4p1=int(hdr.span4p1), gather_off=int(hdr.gather_off))
    with _cache_lock:
        _cache[key] = t
    return t
