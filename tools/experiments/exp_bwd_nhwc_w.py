import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from interpolate_antialiasing_amd import _lib, extension_interpolate as aa
def timed(fn, reps=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
B = 256
g = torch.randn(B, 3, 196, 320, device="cuda").contiguous(memory_format=torch.channels_last)
for W in (896, 904, 906, 1024):
    ms = timed(lambda: aa.linear_backward(g, [196, 320], [B, 3, 438, W]))
    nbytes = B * 3 * 4 * (438 * W + 196 * 320)
    print(f"bwd nhwc -> [438,{W}]: {ms:.4f} ms {nbytes/ms/1e6:.0f} GB/s {_lib.last_variant()}", flush=True)
