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
for B in (1, 8, 32, 64, 128, 256, 512):
    x = torch.rand(B, 3, 438, 906, device="cuda") * 255
    xl = x.contiguous(memory_format=torch.channels_last)
    a = timed(lambda: aa.linear_forward(x, [196, 320])); b = timed(lambda: aa.linear_forward(xl, [196, 320])); c = timed(lambda: aa.cubic_forward(x, [196, 320]))
    nb = B * 3 * 4 * (438 * 906 + 196 * 320)
    print(f"B={B:4d} nchw {a:.4f} ms {nb/a/1e6:6.0f} GB/s | nhwc {b:.4f} ms {nb/b/1e6:6.0f} | cubic nchw {c:.4f} ms {nb/c/1e6:6.0f}", flush=True)
