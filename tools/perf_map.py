#!/usr/bin/env python3
"""Throughput map over test.py's five output sizes (test.py:15-21) x dtype / layout / arithmetic: which kernel ran and at what
algorithmic GB/s.  Usage: python tools/perf_map.py [batch]   (GPU only; writes one line per case)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from interpolate_antialiasing_amd import _lib, extension_interpolate as aa  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H, W = 438, 906
SIZES = [(320, 196), (460, 220), (120, 96), (1200, 196), (120, 1200), (1200, 1200)]  # (W, H) as test.py writes them


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


torch.manual_seed(0)
u8 = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
cases = [
    ("u8 nhwc pil", u8, dict(uint8_mode="pil"), 1),
    ("u8 nhwc harness", u8, dict(uint8_mode="harness"), 1),
    ("u8 nchw pil", u8.contiguous(), dict(uint8_mode="pil"), 1),
    ("f32 nchw", u8.float().contiguous(), {}, 4),
    ("f32 nhwc", u8.float().contiguous(memory_format=torch.channels_last), {}, 4),
    ("f16 nchw", u8.half().contiguous(), {}, 2),
]
for fname, op in (("linear", aa.linear_forward), ("cubic", aa.cubic_forward)):
    for (ow, oh) in SIZES:
        for cname, x, kw, es in cases:
            ms = timed(lambda: op(x, [oh, ow], **kw))
            nbytes = B * 3 * es * (H * W + oh * ow)
            print(f"{fname:6s} ({ow:4d},{oh:4d}) {cname:16s} {ms:8.4f} ms {nbytes / ms / 1e6:7.0f} GB/s  {_lib.last_variant()}", flush=True)
