"""Can the backward kernel's two halves overlap when they sit in DIFFERENT waves?  Runs the no-store build and the stores-only build
of the same launch concurrently on two streams (two copies of the library in one process) and compares with each alone."""
import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from interpolate_antialiasing_amd import _lib, tables, extension_interpolate as aa
root = os.path.join(os.getcwd(), "interpolate_antialiasing_amd", "csrc")
def load(name):
    L = ctypes.CDLL(os.path.join(root, name))
    vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
    ax = ctypes.POINTER(_lib.Axis)
    L.aa_resample_bwd.argtypes = [vp, vp, vp, sz, i32, i32, i64, i64, i64, i64, ax, ax, vp]
    L.aa_resample_bwd.restype = i32
    return L
L1, L2 = load("libaa_interp_up1.so"), load("libaa_interp_up2.so")
dev = torch.device("cuda")
N = 256
g = torch.randn(N, 3, 196, 320, device=dev)
o1 = torch.empty(N, 3, 438, 906, device=dev); o2 = torch.empty_like(o1)
th = tables.get_table(_lib.FILTER_LINEAR, _lib.TABLE_F32, 438, 196, False, 0.0, dev)
tw = tables.get_table(_lib.FILTER_LINEAR, _lib.TABLE_F32, 906, 320, False, 0.0, dev)
trh, trw = tables.get_transposed_table(th).axis(), tables.get_transposed_table(tw).axis()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def call(L, out, stream):
    rc = L.aa_resample_bwd(g.data_ptr(), out.data_ptr(), None, 0, _lib.F32, _lib.NCHW, N, 3, 438, 906, ctypes.byref(trh), ctypes.byref(trw), stream.cuda_stream)
    assert rc == 0, rc
def timed(fn, reps=40):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    s1.synchronize(); s2.synchronize()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
torch.cuda.synchronize()
def a(): call(L1, o1, s1)
def b(): call(L2, o2, s2)
def both(): call(L1, o1, s1); call(L2, o2, s2)
import time
def wall(fn, reps=200):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("no-store kernel alone   %.4f ms" % wall(a))
print("stores-only kernel alone %.4f ms" % wall(b))
print("both, two streams        %.4f ms (sum if serial, max if they overlap)" % wall(both))
