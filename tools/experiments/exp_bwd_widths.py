import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from interpolate_antialiasing_amd import _lib, extension_interpolate as aa
def timed(fn, reps=30):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
g = torch.randn(256, 3, 196, 320, device="cuda")
for W in [int(a) for a in sys.argv[1:]]:
    ms = timed(lambda: aa.linear_backward(g, [196, 320], [256, 3, 438, W]))
    nbytes = 256 * 3 * 4 * (438 * W + 196 * 320)
    print(f"bwd -> [438,{W}]: {ms:.4f} ms  {nbytes/ms/1e6:.0f} GB/s  pitch%64={W*4%64}", flush=True)
