"""Drop-in operator surface of the reference's pybind11 module, running on MI355X.

Mirrors step_two_dot_two/extension_interpolate.cpp:46-51 (same names, argument meaning, error behaviour):

    linear_forward(input, output_size, align_corners=False)   -> Tensor      (:7-14)
    nearest_forward(input, output_size, align_corners=False)  -> Tensor      (:26-33; "it's not nearest but box")
    cubic_forward(input, output_size, align_corners=False)    -> Tensor      (:35-42)
    linear_backward(grad_output, output_size, input_size, align_corners=False) -> Tensor   (:16-24)
    forward(...)                                               legacy name of linear_forward used by every other
                                                               step (step_three/extension_interpolate.cpp:17-19)
plus cubic_backward / nearest_backward (the commented-out intent at test.py:111-116).

``output_size`` is (H, W); ``input_size`` is the full NCHW size (test.py:140-143).  antialias=True and
scale_factors={} are hard-wired exactly as in the reference wrappers.  The callee allocates and returns a fresh
tensor whose memory format follows the input (aa_interpolation_impl.h:739,752).

Differences, all additive:
  * tensors must live on a ROCm GPU — this package is the HIP path only and has no CPU implementation;
  * uint8 input is accepted for linear/cubic/box (the reference dispatches floating types only, :609-614): the
    default ``uint8_mode="pil"`` is bit-exact with PIL.Image.resize (integer arithmetic, uint8 intermediate);
    ``uint8_mode="harness"`` reproduces test.py:52-58,72,75 (float(), fp32 op, clamp for bicubic, truncating byte());
  * the backward is the TRUE adjoint of the antialiased forward (the reference header's is the non-AA one);
  * ``precision="fast"`` (f32 / f16 / bf16): the opt-in tolerance mode — FMA accumulation, results within 1e-4 relative of the
    reference's (BASELINE.json's float bar) instead of bit-identical; the default ``"exact"`` rounds product and sum separately in
    the reference's tap order.  In fast mode a NaN / Inf pixel also reaches outputs whose 16-byte-aligned window holds it;
  * the same callables are registered as ``torch.ops.extension_interpolate.*``.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import torch

from . import _lib, tables

__all__ = ["linear_forward", "nearest_forward", "cubic_forward", "linear_backward", "cubic_backward",
           "nearest_backward", "forward", "linear_forward_nd", "cubic_forward_nd", "nearest_forward_nd", "linear_backward_nd",
           "cubic_backward_nd", "set_uint8_mode",
           "get_uint8_mode", "last_variant"]

_uint8_mode = "pil"
_plans = {}  # per call shape: (axis descriptors, workspace bytes); see _forward

_DTYPE_IDS = {torch.uint8: _lib.U8, torch.float32: _lib.F32, torch.float64: _lib.F64, torch.float16: _lib.F16,
              torch.bfloat16: _lib.BF16}
_DTYPE_NAMES = {torch.float16: "Half", torch.bfloat16: "BFloat16", torch.int8: "Char", torch.int16: "Short",
                torch.int32: "Int", torch.int64: "Long", torch.bool: "Bool", torch.uint8: "Byte"}


def set_uint8_mode(mode: str) -> None:
    global _uint8_mode
    if mode not in ("pil", "harness"):
        raise ValueError("uint8_mode must be 'pil' or 'harness'")
    _uint8_mode = mode


def get_uint8_mode() -> str:
    return _uint8_mode


def last_variant() -> str:
    """Kernel variant the last forward on this thread dispatched to."""
    return _lib.last_variant()


# ---- argument checks with the reference's wording (ATen upsample_2d_common_check; s2.2:744-750) -------------
def _check_sizes(input_size: Sequence[int], output_size: Sequence[int]):
    if len(output_size) != 2:
        raise RuntimeError(f"It is expected output_size equals to 2, but got size {len(output_size)}")
    if len(input_size) != 4:
        raise RuntimeError(f"It is expected input_size equals to 4, but got size {len(input_size)}")
    n, c, h, w = (int(v) for v in input_size)
    oh, ow = int(output_size[0]), int(output_size[1])
    if not (h > 0 and w > 0 and oh > 0 and ow > 0):
        raise RuntimeError("Input and output sizes should be greater than 0, but got "
                           f"input (H: {h}, W: {w}) output (H: {oh}, W: {ow})")
    return n, c, h, w, oh, ow


def _memory_format(x: torch.Tensor):
    """input.suggest_memory_format() restricted to the two dense layouts the kernels take; anything else is
    made contiguous first (the reference walks arbitrary strides through TensorIterator)."""
    if x.is_contiguous():
        return x, _lib.NCHW
    if x.is_contiguous(memory_format=torch.channels_last):
        return x, _lib.NHWC
    return x.contiguous(), _lib.NCHW


def _pitched_view(x: torch.Tensor):
    """A non-dense 4-D view the kernels can read where it lies (aa_resample_fwd_strided): rows of consecutive elements, any row pitch,
    planes uniformly spaced — what a crop x[:, :, y0:y1, x0:x1] or (channels_last) a batch slice of a dense tensor is.
    -> (layout, strides) or None."""
    n, c, h, w = x.shape
    sn, sc, sh, sw = x.stride()
    if min(sn, sc, sh, sw) < 0 or x.numel() == 0:
        return None
    if sw == 1 and sh >= w and (c == 1 or n == 1 or sn == c * sc) and (c == 1 or sc >= 1):
        return _lib.NCHW, (sn, sc, sh, sw)
    if sc == 1 and sw == c and sh >= w * c and c > 1:
        return _lib.NHWC, (sn, sc, sh, sw)
    return None


def _table_kind(dtype: torch.dtype, uint8_mode: Optional[str]) -> int:
    if dtype in (torch.float32, torch.float16, torch.bfloat16):  # 16-bit floats compute in fp32
        return _lib.TABLE_F32
    if dtype == torch.float64:
        return _lib.TABLE_F64
    mode = uint8_mode or _uint8_mode
    if mode not in ("pil", "harness"):
        raise ValueError("uint8_mode must be 'pil' or 'harness'")
    return _lib.TABLE_PIL if mode == "pil" else _lib.TABLE_F32


def _require_gpu(x: torch.Tensor, what: str):
    if not x.is_cuda:
        raise _lib.AAInterpError(
            f"{what}: expected a tensor on a ROCm GPU, got device '{x.device}'. interpolate_antialiasing_amd is the "
            "MI355X HIP path only and has no CPU implementation.")


def _user_scales(scale_factors, n: int):
    """ATen's optional per-axis scale factors (`scale_h`, `scale_w` of upsample_*2d; the reference hard-wires none,
    s2.2:11-13).  A given factor s replaces in/out by 1/s in area_pixel_compute_scale unless align_corners."""
    if scale_factors is None:
        return [0.0] * n
    sf = [float(v) if v is not None else 0.0 for v in scale_factors]
    if len(sf) != n or any(v < 0 for v in sf):
        raise RuntimeError(f"scale_factors must hold {n} positive values, got {list(scale_factors)}")
    return sf


def _forward(filter_id: int, name: str, input: torch.Tensor, output_size: Sequence[int], align_corners: bool,
             uint8_mode: Optional[str] = None, scale_factors: Optional[Sequence[float]] = None, out_dtype=None,
             out_format: Optional[str] = None, mean=None, std=None, precision: Optional[str] = None) -> torch.Tensor:
    if not isinstance(input, torch.Tensor):
        raise TypeError(f"{name}(): argument 'input' must be Tensor")
    if precision not in (None, "exact", "fast"):
        raise ValueError("precision must be 'exact' (default: the reference's results bit for bit) or 'fast' (within 1e-4 relative)")
    flags = _lib.FLAG_FAST if precision == "fast" else 0
    if out_dtype is not None or out_format is not None or mean is not None or std is not None:
        if input.dtype != torch.uint8 or out_dtype not in (None, torch.float32):
            raise NotImplementedError("out_dtype / out_format / mean / std: the fused conversion takes uint8 input and gives float32")
        if uint8_mode == "pil":
            raise NotImplementedError("float32 output is the reference's fp32 arithmetic (uint8_mode='harness'), not Pillow's integers")
        return _forward_to_float(filter_id, name, input, output_size, align_corners, scale_factors, out_format, mean, std, flags)
    n, c, h, w, oh, ow = _check_sizes(input.shape, output_size)
    if input.numel() == 0 and (c == 0):  # empty batch allowed, nothing else (s2.2:747-750)
        raise RuntimeError(f"Non-empty 4D data tensor expected but got a tensor with sizes {list(input.shape)}")
    if input.dtype not in _DTYPE_IDS:
        raise NotImplementedError(f'"upsample_generic_Nd" not implemented for \'{_DTYPE_NAMES.get(input.dtype, str(input.dtype))}\'')
    _require_gpu(input, name)
    L = _lib.load()
    pitched = None
    if not (input.is_contiguous() or input.is_contiguous(memory_format=torch.channels_last)):
        pitched = _pitched_view(input)  # a crop / batch slice: read in place when a fused kernel takes it (no .contiguous() round trip)
    if pitched is not None:
        x, layout = input, pitched[0]
    else:
        x, layout = _memory_format(input)
    kind = _table_kind(x.dtype, uint8_mode)
    if kind == _lib.TABLE_PIL and align_corners:
        raise NotImplementedError("uint8_mode='pil' has no align_corners (Pillow has none); use uint8_mode='harness'")
    sh, sw = _user_scales(scale_factors, 2)
    if kind == _lib.TABLE_PIL and (sh or sw):
        raise NotImplementedError("uint8_mode='pil' has no scale factors (Pillow derives the scale from the sizes)")
    dev = x.device
    mf = torch.channels_last if layout == _lib.NHWC else torch.contiguous_format
    out = torch.empty((n, c, oh, ow), dtype=x.dtype, device=dev, memory_format=mf)
    if n == 0:
        return out
    dt = _DTYPE_IDS[x.dtype]
    # host-side plan: the two cached tables' axis descriptors and the workspace size for this exact call shape
    key = (filter_id, dt, layout, n, c, h, w, oh, ow, bool(align_corners), kind, sh, sw, dev.index, _lib.fused_epoch)
    plan = _plans.get(key)
    cur = torch.cuda.current_device()
    if plan is None:
        with torch.cuda.device(dev):
            th, tw = tables.get_table_pair(filter_id, kind, h, oh, w, ow, align_corners, sh, sw, dev)
            ah, aw = th.axis(), tw.axis()
            ws_bytes = L.aa_workspace_bytes(dt, layout, n, c, h, w, oh, ow, ctypes.byref(ah), ctypes.byref(aw))
        plan = (ah, aw, ws_bytes, ctypes.byref(ah), ctypes.byref(aw), th, tw)  # (th, tw keep the device buffers alive)
        if len(_plans) > 4096:
            _plans.clear()
        _plans[key] = plan
    ah, aw, ws_bytes, pah, paw = plan[:5]
    if pitched is not None:
        strides = (ctypes.c_int64 * 4)(*pitched[1])
        with torch.cuda.device(dev):
            rc = L.aa_resample_fwd_strided(x.data_ptr(), out.data_ptr(), dt, layout, n, c, h, w, strides, pah, paw, flags,
                                           torch.cuda.current_stream(dev).cuda_stream)
        if rc != _lib.ERR_STRIDES:
            _lib.check(rc, name)
            return out
        x, layout2 = _memory_format(input)  # no kernel for this view: the dense copy after all (the plan was made for this layout)
        if layout2 != layout:
            return _forward(filter_id, name, x, output_size, align_corners, uint8_mode, scale_factors, None, None, None, None, precision)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
    if dev.index == cur:
        rc = L.aa_resample_fwd_ex(x.data_ptr(), out.data_ptr(), ws.data_ptr() if ws is not None else None, ws_bytes, dt, layout,
                                  n, c, h, w, pah, paw, flags, torch.cuda.current_stream(dev).cuda_stream)
    else:
        with torch.cuda.device(dev):
            rc = L.aa_resample_fwd_ex(x.data_ptr(), out.data_ptr(), ws.data_ptr() if ws is not None else None, ws_bytes, dt,
                                      layout, n, c, h, w, pah, paw, flags, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, name)
    return out


def _forward_to_float(filter_id: int, name: str, input: torch.Tensor, output_size: Sequence[int], align_corners: bool,
                      scale_factors, out_format: Optional[str], mean, std, flags: int = 0) -> torch.Tensor:
    """Decode-adjacent forward: uint8 in, float32 out, one launch.  The reference's harness spends 0.33 of its
    2.27 ms per image on np.asarray(pil) -> transpose -> .float() before the op (test.py:337-339,55; README.md:416); here the
    uint8 bytes (HWC = channels_last, or CHW) are read directly, the op runs in the reference's fp32 arithmetic and the
    float32 result is written in the requested layout ("nchw" / "nhwc"; default: the input's), optionally normalised
    per channel as (v - mean[c]) / std[c].  Equals ``op(input.float())`` bit for bit."""
    n, c, h, w, oh, ow = _check_sizes(input.shape, output_size)
    if input.numel() == 0 and c == 0:
        raise RuntimeError(f"Non-empty 4D data tensor expected but got a tensor with sizes {list(input.shape)}")
    _require_gpu(input, name)
    L = _lib.load()
    x, layout = _memory_format(input)
    if out_format not in (None, "nchw", "nhwc"):
        raise ValueError("out_format must be 'nchw', 'nhwc' or None (same as the input)")
    out_layout = layout if out_format is None else (_lib.NHWC if out_format == "nhwc" else _lib.NCHW)
    cv = _lib.Convert()
    cv.out_layout = out_layout
    cv.normalize = 0
    cv.flags = flags
    if (mean is None) != (std is None):
        raise ValueError("mean and std must be given together")
    if mean is not None:
        mean, std = [float(v) for v in mean], [float(v) for v in std]
        if len(mean) != c or len(std) != c or c > 4:
            raise RuntimeError(f"mean/std must hold one value per channel (C = {c} <= 4)")
        cv.normalize = 1
        for i in range(c):
            cv.mean[i], cv.std[i] = mean[i], std[i]
    sh, sw = _user_scales(scale_factors, 2)
    dev = x.device
    mf = torch.channels_last if out_layout == _lib.NHWC else torch.contiguous_format
    out = torch.empty((n, c, oh, ow), dtype=torch.float32, device=dev, memory_format=mf)
    if n == 0:
        return out
    with torch.cuda.device(dev):
        th, tw = tables.get_table_pair(filter_id, _lib.TABLE_F32, h, oh, w, ow, align_corners, sh, sw, dev)
        ah, aw = th.axis(), tw.axis()
        ws_bytes = L.aa_workspace_bytes_u8_to_f32(layout, n, c, h, w, ctypes.byref(ah), ctypes.byref(aw), ctypes.byref(cv))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
        rc = L.aa_resample_fwd_u8_to_f32(x.data_ptr(), out.data_ptr(), ws.data_ptr() if ws is not None else None, ws_bytes, layout,
                                         n, c, h, w, ctypes.byref(ah), ctypes.byref(aw), ctypes.byref(cv),
                                         torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, name)
    return out


def _backward(filter_id: int, name: str, grad_output: torch.Tensor, output_size: Sequence[int],
              input_size: Sequence[int], align_corners: bool, atomic: bool = False) -> torch.Tensor:
    n, c, h, w, oh, ow = _check_sizes(input_size, output_size)
    # ti_upsample_bilinear2d_backward_cpu checks (s2.2/aa_interpolation_backward_impl.h:202-212)
    if grad_output.dim() != 4:
        raise RuntimeError(f"Expected grad_output to be a tensor of dimension 4 but got: dimension {grad_output.dim()}")
    full = (n, c, oh, ow)
    for i in range(4):
        if grad_output.size(i) != full[i]:
            raise RuntimeError("Expected grad_output to have the same shape as output; "
                               f"output.size({i}) = {full[i]} but got grad_output.size({i}) = {grad_output.size(i)}")
    if grad_output.dtype not in (torch.float32, torch.float64):
        raise NotImplementedError(f'"ti_upsample_bilinear2d_backward_cpu" not implemented for '
                                  f'\'{_DTYPE_NAMES.get(grad_output.dtype, str(grad_output.dtype))}\'')
    _require_gpu(grad_output, name)
    L = _lib.load()
    go, layout = _memory_format(grad_output)
    dev = go.device
    mf = torch.channels_last if layout == _lib.NHWC else torch.contiguous_format
    gi = torch.empty((n, c, h, w), dtype=go.dtype, device=dev, memory_format=mf)
    if n == 0:
        return gi
    kind = _lib.TABLE_F32 if go.dtype == torch.float32 else _lib.TABLE_F64
    dt = _DTYPE_IDS[go.dtype]
    with torch.cuda.device(dev):
        th = tables.get_table(filter_id, kind, h, oh, align_corners, 0.0, dev)
        tw = tables.get_table(filter_id, kind, w, ow, align_corners, 0.0, dev)
        s = tables._stream_ptr(dev)
        if atomic:
            ah, aw = th.axis(), tw.axis()
            ws_bytes = L.aa_workspace_bytes_bwd(dt, layout, n, c, h, w, oh, ow)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            rc = L.aa_resample_bwd_atomic(ctypes.c_void_p(go.data_ptr()), ctypes.c_void_p(gi.data_ptr()),
                                          ctypes.c_void_p(ws.data_ptr()), ws_bytes, dt, layout, n, c, h, w,
                                          ctypes.byref(ah), ctypes.byref(aw), s)
        else:
            trh = tables.get_transposed_table(th).axis()
            trw = tables.get_transposed_table(tw).axis()
            # the gather-form adjoint is a forward resample of grad_out with the transposed tables
            ws_bytes = L.aa_workspace_bytes(dt, layout, n, c, oh, ow, h, w, ctypes.byref(trh), ctypes.byref(trw))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
            rc = L.aa_resample_bwd(ctypes.c_void_p(go.data_ptr()), ctypes.c_void_p(gi.data_ptr()),
                                   ctypes.c_void_p(ws.data_ptr() if ws is not None else 0), ws_bytes, dt, layout,
                                   n, c, h, w, ctypes.byref(trh), ctypes.byref(trw), s)
    _lib.check(rc, name)
    return gi


def _axis_pass_2d(L, x, y, dt, kind, outer, n_in, n_out, inner, table, dev, stream) -> bool:
    """One separable pass of an N-d resample through the FUSED 2-D kernels: the dense array [outer][n][inner] is a stack of 2-D
    images in which only one axis changes — [1,1,outer,n] -> [1,1,outer,n_out] when inner == 1 (the pass runs along rows),
    [outer,1,n,inner] -> [outer,1,n_out,inner] otherwise (along columns) — and the other axis gets the IDENTITY table (box filter,
    same size: one tap of weight exactly 1.0, so x * 1.0 = x bit for bit and no neighbour is ever touched).  Same bytes moved as
    the generic single-axis kernel, but through the streaming kernels.  Returns False (caller runs the single-axis kernel
    aa_resample_axis_fwd, one launch, no workspace) whenever that would NOT be one fused launch: sizes the 2-D entry point cannot
    index, a problem no fused kernel takes (aa_workspace_bytes answers non-zero: the two-launch path would add an identity pass and
    a full-size intermediate, twice the traffic of the single-axis kernel), the fused kernels switched off, and rows so short that
    the identity table (one record per row of the stack) would outweigh them."""
    es = x.element_size()
    if inner == 1:
        if n_in * es < 4 * 84 or outer > (1 << 22):  # identity table: ~84 bytes per row of the stack, cached per distinct `outer`
            return False
        n2, h2, w2, oh2, ow2 = 1, outer, n_in, outer, n_out
        th = tables.get_table(_lib.FILTER_BOX, kind, outer, outer, False, 0.0, dev)
        tw = table
    else:
        n2, h2, w2, oh2, ow2 = outer, n_in, inner, n_out, inner
        th = table
        tw = tables.get_table(_lib.FILTER_BOX, kind, inner, inner, False, 0.0, dev)
    if max(h2, w2, oh2, ow2) >= (1 << 24):
        return False
    ah, aw = th.axis(), tw.axis()
    if L.aa_workspace_bytes(dt, _lib.NCHW, n2, 1, h2, w2, oh2, ow2, ctypes.byref(ah), ctypes.byref(aw)) != 0:
        return False  # no fused kernel for this pass (or they are disabled): the single-axis kernel is the cheaper form
    rc = L.aa_resample_fwd(x.data_ptr(), y.data_ptr(), None, 0, dt, _lib.NCHW, n2, 1, h2, w2, ctypes.byref(ah), ctypes.byref(aw), stream)
    if rc == -6:  # AA_ERR_WORKSPACE: a pointer-dependent decline (an unaligned view) after the shape said yes
        return False
    _lib.check(rc, "aa_resample_fwd (axis pass)")
    return True


def _forward_nd(filter_id: int, name: str, input: torch.Tensor, output_size: Sequence[int], align_corners: bool) -> torch.Tensor:
    """1-D (NCL) and 3-D (NCDHW) front-ends: the reference's separable driver is N-d generic
    (s2.2/aa_interpolation_impl.h:536-683, "NCHW, NCL or NCKHW" :545) although only the 2-D callables are bound.
    One aa_resample_axis_fwd per resampled axis, LAST axis first like the reference (:658), contiguous intermediates."""
    if not isinstance(input, torch.Tensor):
        raise TypeError(f"{name}(): argument 'input' must be Tensor")
    nd = input.dim() - 2
    if nd not in (1, 2, 3):
        raise RuntimeError(f"It is expected input_size equals to 3, 4 or 5, but got size {input.dim()}")
    if len(output_size) != nd:
        raise RuntimeError(f"It is expected output_size equals to {nd}, but got size {len(output_size)}")
    if nd == 2:
        return _forward(filter_id, name, input, output_size, align_corners)
    sizes = [int(v) for v in input.shape[2:]]
    osizes = [int(v) for v in output_size]
    if not all(v > 0 for v in sizes + osizes):
        raise RuntimeError(f"Input and output sizes should be greater than 0, but got input {sizes} output {osizes}")
    if input.dtype not in _DTYPE_IDS or input.dtype == torch.uint8:  # Pillow has no 1-D/3-D resize to be exact against
        raise NotImplementedError(f'"upsample_generic_Nd" not implemented for \'{_DTYPE_NAMES.get(input.dtype, str(input.dtype))}\'')
    _require_gpu(input, name)
    L = _lib.load()
    x = input.contiguous()
    dev = x.device
    kind = _table_kind(x.dtype, None)
    dt = _DTYPE_IDS[x.dtype]
    if x.numel() == 0:
        if x.shape[1] == 0:
            raise RuntimeError(f"Non-empty {nd + 2}D data tensor expected but got a tensor with sizes {list(input.shape)}")
        return torch.empty(list(x.shape[:2]) + osizes, dtype=x.dtype, device=dev)
    with torch.cuda.device(dev):
        s = tables._stream_ptr(dev)
        for k in range(nd - 1, -1, -1):
            shape = list(x.shape)
            n_in, n_out = shape[2 + k], osizes[k]
            outer = 1
            for v in shape[:2 + k]:
                outer *= v
            inner = 1
            for v in shape[3 + k:]:
                inner *= v
            t = tables.get_table(filter_id, kind, n_in, n_out, align_corners, 0.0, dev)
            shape[2 + k] = n_out
            y = torch.empty(shape, dtype=x.dtype, device=dev)
            if not _axis_pass_2d(L, x, y, dt, kind, outer, n_in, n_out, inner, t, dev, s):
                ax = t.axis()
                rc = L.aa_resample_axis_fwd(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), dt, outer, n_in, inner,
                                            ctypes.byref(ax), s)
                _lib.check(rc, name)
            x = y
    return x


def _backward_nd(filter_id: int, name: str, grad_output: torch.Tensor, output_size: Sequence[int], input_size: Sequence[int],
                 align_corners: bool) -> torch.Tensor:
    """True adjoint of `_forward_nd`: one pass per axis with the transposed table of that axis (the 1-D adjoints act on
    different axes and commute).  The reference's backward header carries 1-D/3-D loops as well
    (aa_interpolation_backward_impl.h:58-78,110-150), non-antialiased like its 2-D one."""
    nd = len(input_size) - 2
    if nd not in (1, 2, 3) or len(output_size) != nd:
        raise RuntimeError(f"It is expected input_size equals to 3, 4 or 5 and output_size to match, but got {list(input_size)} "
                           f"and {list(output_size)}")
    if nd == 2:
        return _backward(filter_id, name, grad_output, output_size, input_size, align_corners)
    full = [int(v) for v in input_size[:2]] + [int(v) for v in output_size]
    if list(grad_output.shape) != full:
        raise RuntimeError(f"Expected grad_output to have the same shape as output; output.shape = {full} but got "
                           f"grad_output.shape = {list(grad_output.shape)}")
    if grad_output.dtype not in (torch.float32, torch.float64):
        raise NotImplementedError(f'"ti_upsample_bilinear2d_backward_cpu" not implemented for '
                                  f'\'{_DTYPE_NAMES.get(grad_output.dtype, str(grad_output.dtype))}\'')
    _require_gpu(grad_output, name)
    L = _lib.load()
    g = grad_output.contiguous()
    dev = g.device
    kind = _lib.TABLE_F32 if g.dtype == torch.float32 else _lib.TABLE_F64
    dt = _DTYPE_IDS[g.dtype]
    if g.numel() == 0:
        return torch.zeros([int(v) for v in input_size], dtype=g.dtype, device=dev)
    with torch.cuda.device(dev):
        s = tables._stream_ptr(dev)
        for k in range(nd):
            shape = list(g.shape)
            n_out_fwd, n_in_fwd = shape[2 + k], int(input_size[2 + k])
            outer = 1
            for v in shape[:2 + k]:
                outer *= v
            inner = 1
            for v in shape[3 + k:]:
                inner *= v
            fwd = tables.get_table(filter_id, kind, n_in_fwd, n_out_fwd, align_corners, 0.0, dev)
            tr = tables.get_transposed_table(fwd)  # maps n_out_fwd -> n_in_fwd
            shape[2 + k] = n_in_fwd
            y = torch.empty(shape, dtype=g.dtype, device=dev)
            if not _axis_pass_2d(L, g, y, dt, kind, outer, n_out_fwd, n_in_fwd, inner, tr, dev, s):
                ax = tr.axis()
                rc = L.aa_resample_axis_fwd(ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(y.data_ptr()), dt, outer, n_out_fwd, inner,
                                            ctypes.byref(ax), s)
                _lib.check(rc, name)
            g = y
    return g


def linear_backward_nd(grad_output: torch.Tensor, output_size: Sequence[int], input_size: Sequence[int],
                       align_corners: bool = False) -> torch.Tensor:
    return _backward_nd(_lib.FILTER_LINEAR, "linear_backward_nd", grad_output, output_size, input_size, align_corners)


def cubic_backward_nd(grad_output: torch.Tensor, output_size: Sequence[int], input_size: Sequence[int],
                      align_corners: bool = False) -> torch.Tensor:
    return _backward_nd(_lib.FILTER_CUBIC, "cubic_backward_nd", grad_output, output_size, input_size, align_corners)


def linear_forward_nd(input: torch.Tensor, output_size: Sequence[int], align_corners: bool = False) -> torch.Tensor:
    """Antialiased linear / bilinear / trilinear resize of an NCL, NCHW or NCDHW tensor."""
    return _forward_nd(_lib.FILTER_LINEAR, "linear_forward_nd", input, output_size, align_corners)


def cubic_forward_nd(input: torch.Tensor, output_size: Sequence[int], align_corners: bool = False) -> torch.Tensor:
    return _forward_nd(_lib.FILTER_CUBIC, "cubic_forward_nd", input, output_size, align_corners)


def nearest_forward_nd(input: torch.Tensor, output_size: Sequence[int], align_corners: bool = False) -> torch.Tensor:
    return _forward_nd(_lib.FILTER_BOX, "nearest_forward_nd", input, output_size, align_corners)


# ---- the reference's callables ---------------------------------------------------------------------------
def linear_forward(input: torch.Tensor, output_size: Sequence[int], align_corners: bool = False, *,
                    uint8_mode: Optional[str] = None, scale_factors: Optional[Sequence[float]] = None, out_dtype=None,
                    out_format: Optional[str] = None, mean=None, std=None, precision: Optional[str] = None) -> torch.Tensor:
    """Anti-Aliased Linear Interpolation forward (s2.2/extension_interpolate.cpp:7-14,47)."""
    return _forward(_lib.FILTER_LINEAR, "linear_forward", input, output_size, align_corners, uint8_mode, scale_factors, out_dtype, out_format,
                    mean, std, precision)


def nearest_forward(input: torch.Tensor, output_size: Sequence[int], align_corners: bool = False, *,
                    uint8_mode: Optional[str] = None, scale_factors: Optional[Sequence[float]] = None, out_dtype=None,
                    out_format: Optional[str] = None, mean=None, std=None, precision: Optional[str] = None) -> torch.Tensor:
    """Anti-Aliased "Nearest" (really: box filter) forward (s2.2/extension_interpolate.cpp:26-33,48)."""
    return _forward(_lib.FILTER_BOX, "nearest_forward", input, output_size, align_corners, uint8_mode, scale_factors, out_dtype, out_format,
                    mean, std, precision)


def cubic_forward(input: torch.Tensor, output_size: Sequence[int], align_corners: bool = False, *,
                    uint8_mode: Optional[str] = None, scale_factors: Optional[Sequence[float]] = None, out_dtype=None,
                    out_format: Optional[str] = None, mean=None, std=None, precision: Optional[str] = None) -> torch.Tensor:
    """Anti-Aliased Cubic Interpolation forward (s2.2/extension_interpolate.cpp:35-42,49)."""
    return _forward(_lib.FILTER_CUBIC, "cubic_forward", input, output_size, align_corners, uint8_mode, scale_factors, out_dtype, out_format,
                    mean, std, precision)


def linear_backward(grad_output: torch.Tensor, output_size: Sequence[int], input_size: Sequence[int],
                    align_corners: bool = False, *, atomic: bool = False) -> torch.Tensor:
    """Backward of linear_forward (s2.2/extension_interpolate.cpp:16-24,50) — the true AA adjoint."""
    return _backward(_lib.FILTER_LINEAR, "linear_backward", grad_output, output_size, input_size, align_corners, atomic)


def cubic_backward(grad_output: torch.Tensor, output_size: Sequence[int], input_size: Sequence[int],
                   align_corners: bool = False, *, atomic: bool = False) -> torch.Tensor:
    """Backward of cubic_forward (intent at test.py:111-116)."""
    return _backward(_lib.FILTER_CUBIC, "cubic_backward", grad_output, output_size, input_size, align_corners, atomic)


def nearest_backward(grad_output: torch.Tensor, output_size: Sequence[int], input_size: Sequence[int],
                     align_corners: bool = False, *, atomic: bool = False) -> torch.Tensor:
    return _backward(_lib.FILTER_BOX, "nearest_backward", grad_output, output_size, input_size, align_corners, atomic)


# legacy export of every step but step_two_dot_two (step_three/extension_interpolate.cpp:17-19)
forward = linear_forward


# ---- torch.ops.extension_interpolate.* ---------------------------------------------------------------------
def _register_torch_ops() -> None:
    lib = torch.library.Library("extension_interpolate", "DEF")
    fwd_schema = "(Tensor input, int[] output_size, bool align_corners=False) -> Tensor"
    bwd_schema = "(Tensor grad_output, int[] output_size, int[] input_size, bool align_corners=False) -> Tensor"
    fwds = {"linear_forward": linear_forward, "nearest_forward": nearest_forward, "cubic_forward": cubic_forward,
            "forward": linear_forward}
    bwds = {"linear_backward": linear_backward, "cubic_backward": cubic_backward, "nearest_backward": nearest_backward}
    for name in fwds:
        lib.define(name + fwd_schema)
    for name in bwds:
        lib.define(name + bwd_schema)

    def _mf(x):
        return torch.channels_last if (x.dim() == 4 and not x.is_contiguous()
                                       and x.is_contiguous(memory_format=torch.channels_last)) else torch.contiguous_format

    for name, fn in fwds.items():
        lib.impl(name, (lambda f: lambda input, output_size, align_corners=False: f(input, output_size, align_corners))(fn), "CUDA")
        lib.impl(name, lambda input, output_size, align_corners=False: torch.empty(
            (input.shape[0], input.shape[1], output_size[0], output_size[1]), dtype=input.dtype, device=input.device,
            memory_format=_mf(input)), "Meta")
    for name, fn in bwds.items():
        lib.impl(name, (lambda f: lambda grad_output, output_size, input_size, align_corners=False:
                        f(grad_output, output_size, input_size, align_corners))(fn), "CUDA")
        lib.impl(name, lambda grad_output, output_size, input_size, align_corners=False: torch.empty(
            tuple(input_size), dtype=grad_output.dtype, device=grad_output.device, memory_format=_mf(grad_output)), "Meta")

    # autograd: d(forward)/d(input) is the matching backward op (true adjoint)
    def _make_autograd(fwd_name, bwd_name):
        def setup_context(ctx, inputs, output):
            input, output_size, align_corners = inputs
            ctx.in_shape = tuple(input.shape)
            ctx.out_size = tuple(output_size)
            ctx.align_corners = align_corners

        def backward(ctx, grad):
            op = getattr(torch.ops.extension_interpolate, bwd_name)
            return op(grad, list(ctx.out_size), list(ctx.in_shape), ctx.align_corners), None, None

        torch.library.register_autograd(f"extension_interpolate::{fwd_name}", backward, setup_context=setup_context, lib=lib)

    _make_autograd("linear_forward", "linear_backward")
    _make_autograd("forward", "linear_backward")
    _make_autograd("cubic_forward", "cubic_backward")
    _make_autograd("nearest_forward", "nearest_backward")
    globals()["_torch_library"] = lib  # keep alive


try:
    _register_torch_ops()
except Exception as _e:  # pragma: no cover - registration problems must not hide the direct callables
    import warnings

    warnings.warn(f"torch.ops.extension_interpolate registration failed: {_e}")
