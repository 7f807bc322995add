"""Throughput of the headline shape over batch sizes (uint8 channels_last 438x906 -> 196x320, Pillow-exact)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interpolate_antialiasing_amd import _lib, extension_interpolate as aa  # noqa: E402

torch.manual_seed(0)
for n in (1, 2, 4, 8, 16, 32, 64, 128, 256, 384, 512, 768, 1024, 1500, 2048, 4096):
    x = torch.randint(0, 256, (n, 438, 906, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
    for _ in range(5):
        y = aa.linear_forward(x, [196, 320])
    reps = 200 if n <= 64 else 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = aa.linear_forward(x, [196, 320])
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"N={n:5d}  {ms * 1e3:9.1f} us/call  {ms * 1e3 / n:8.2f} us/image  {n * 1378644 / ms / 1e6:8.1f} GB/s  [{_lib.last_variant()}]", flush=True)
    del x, y
