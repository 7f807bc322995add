#!/usr/bin/env python3
"""One named workload, a few launches, for rocprofv3 (kernel-trace / PMC passes): tools/workload.py NAME [launches]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from interpolate_antialiasing_amd import _lib, extension_interpolate as aa  # noqa: E402

name = sys.argv[1]
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
torch.manual_seed(0)
dev = "cuda"
if name == "headline":
    x = torch.randint(0, 256, (1024, 438, 906, 3), dtype=torch.uint8, device=dev).permute(0, 3, 1, 2)
    fn = lambda: aa.linear_forward(x, [196, 320])
elif name == "c2":
    x = torch.rand(64, 3, 1024, 1024, device=dev) * 255
    fn = lambda: aa.cubic_forward(x, [224, 224])
elif name == "c2fast":
    x = torch.rand(64, 3, 1024, 1024, device=dev) * 255
    fn = lambda: aa.cubic_forward(x, [224, 224], precision="fast")
elif name == "c2f16fast":
    x = (torch.rand(64, 3, 1024, 1024, device=dev) * 255).half()
    fn = lambda: aa.linear_forward(x, [224, 224], precision="fast")
elif name == "c0f32":
    x = torch.rand(256, 3, 438, 906, device=dev) * 255
    fn = lambda: aa.linear_forward(x, [196, 320])
elif name == "c0nhwc":
    x = (torch.rand(256, 3, 438, 906, device=dev) * 255).contiguous(memory_format=torch.channels_last)
    fn = lambda: aa.linear_forward(x, [196, 320])
elif name == "c2f16":
    x = (torch.rand(64, 3, 1024, 1024, device=dev) * 255).half()
    fn = lambda: aa.linear_forward(x, [224, 224])
elif name == "shard3":
    x = torch.randint(0, 256, (1024, 906, 438, 3), dtype=torch.uint8, device=dev).permute(0, 3, 1, 2)
    fn = lambda: aa.linear_forward(x, [320, 196])
elif name == "harness":
    x = torch.randint(0, 256, (1024, 906, 438, 3), dtype=torch.uint8, device=dev).permute(0, 3, 1, 2)
    fn = lambda: aa.linear_forward(x, [320, 196], uint8_mode="harness")
elif name == "convert":
    x = torch.randint(0, 256, (1024, 906, 438, 3), dtype=torch.uint8, device=dev).permute(0, 3, 1, 2)
    fn = lambda: aa.linear_forward(x, [320, 196], out_dtype=torch.float32, out_format="nchw")
elif name == "convertfast":
    x = torch.randint(0, 256, (1024, 906, 438, 3), dtype=torch.uint8, device=dev).permute(0, 3, 1, 2)
    fn = lambda: aa.linear_forward(x, [320, 196], out_dtype=torch.float32, out_format="nchw", precision="fast")
elif name == "harnessfast":
    x = torch.randint(0, 256, (1024, 906, 438, 3), dtype=torch.uint8, device=dev).permute(0, 3, 1, 2)
    fn = lambda: aa.linear_forward(x, [320, 196], uint8_mode="harness", precision="fast")
elif name == "planar":
    x = torch.randint(0, 256, (1024, 3, 906, 438), dtype=torch.uint8, device=dev)
    fn = lambda: aa.linear_forward(x, [320, 196])
elif name == "bwd":
    x = torch.randn(256, 3, 196, 320, device=dev)
    fn = lambda: aa.linear_backward(x, [196, 320], [256, 3, 438, 906])
elif name == "bwd_nhwc":
    x = torch.randn(256, 3, 196, 320, device=dev).contiguous(memory_format=torch.channels_last)
    fn = lambda: aa.linear_backward(x, [196, 320], [256, 3, 438, 906])
elif name == "bwd_f64":
    x = torch.randn(128, 3, 196, 320, device=dev, dtype=torch.float64)
    fn = lambda: aa.linear_backward(x, [196, 320], [128, 3, 438, 906])
elif name == "up":
    x = torch.rand(64, 3, 438, 906, device=dev) * 255
    fn = lambda: aa.linear_forward(x, [1200, 1200])
elif name.startswith("bwdw:"):  # bwdw:<W> : the backward of config A with W input columns (the 906-column residue study)
    wcols = int(name.split(":")[1])
    x = torch.randn(256, 3, 196, 320, device=dev)
    fn = lambda: aa.linear_backward(x, [196, 320], [256, 3, 438, wcols])
elif name.startswith("planar:"):  # planar:<H>:<W>:<oH>:<oW>:<linear|cubic>:<B>:<plane groups 0|1>   (uint8 NCHW, Pillow arithmetic)
    _, h, w, oh, ow, filt, b, grp = name.split(":")
    _lib.set_plane_groups(int(grp))
    x = torch.randint(0, 256, (int(b), 3, int(h), int(w)), dtype=torch.uint8, device=dev)
    op = aa.linear_forward if filt == "linear" else aa.cubic_forward
    fn = lambda: op(x, [int(oh), int(ow)])
elif name.startswith("custom:"):  # custom:<u8|u8h|f32|f16>:<nchw|nhwc>:<linear|cubic>:<oW>:<oH>:<B>   (input 438x906x3)
    _, dt, lay, filt, ow, oh, b = name.split(":")
    x = torch.randint(0, 256, (int(b), 438, 906, 3), dtype=torch.uint8, device=dev).permute(0, 3, 1, 2)
    if lay == "nchw":
        x = x.contiguous()
    kw = {}
    if dt == "u8h":
        kw["uint8_mode"] = "harness"
    elif dt == "f32":
        x = x.float()
    elif dt == "f16":
        x = x.half()
    op = aa.linear_forward if filt == "linear" else aa.cubic_forward
    fn = lambda: op(x, [int(oh), int(ow)], **kw)
else:
    raise SystemExit("unknown workload " + name)
for _ in range(launches):
    y = fn()
torch.cuda.synchronize()
if os.environ.get("AA_TIME"):  # event-timed average over the same launches (A/B experiments)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        y = fn()
    e1.record()
    e1.synchronize()
    print(f"{name} {e0.elapsed_time(e1) / launches:.4f} ms", _lib.last_variant(), flush=True)
else:
    print(name, _lib.last_variant(), tuple(y.shape))
