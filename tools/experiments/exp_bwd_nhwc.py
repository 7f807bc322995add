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
nbytes = B * 3 * 4 * (438 * 906 + 196 * 320)
for name, g in (("f32 nchw", torch.randn(B, 3, 196, 320, device="cuda")),
                ("f32 nhwc", torch.randn(B, 3, 196, 320, device="cuda").contiguous(memory_format=torch.channels_last)),
                ("f64 nchw", torch.randn(B // 2, 3, 196, 320, device="cuda", dtype=torch.float64))):
    for fn in (aa.linear_backward, aa.cubic_backward):
        ms = timed(lambda: fn(g, [196, 320], [g.shape[0], 3, 438, 906]))
        print(f"{fn.__name__} {name}: {ms:.4f} ms {nbytes/ms/1e6:.0f} GB/s {_lib.last_variant()}", flush=True)
