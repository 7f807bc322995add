#!/usr/bin/env python3
"""Where a B = 1 call spends its microseconds (verdict round 2, weak #10): the headline op on one image,
[1,3,438,906] uint8 channels_last -> [196,320], timed piece by piece on the host (perf_counter over 20000 repetitions each, GPU
idle in between so that nothing queues), next to the whole call timed both ways (host wall clock per call with a sync at the end of
the loop = sustained call rate; HIP events = device-side spacing).  Prints one JSON object."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from interpolate_antialiasing_amd import _lib, tables  # noqa: E402
from interpolate_antialiasing_amd import extension_interpolate as aa  # noqa: E402

dev = torch.device("cuda", 0)
x = torch.randint(0, 256, (1, 438, 906, 3), dtype=torch.uint8, device=dev).permute(0, 3, 1, 2)
L = _lib.load()
REPS = 20000


def per_call(fn, reps=REPS, sync=True):
    for _ in range(200):
        fn()
    if sync:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    if sync:
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


res = {}
res["whole call (linear_forward), host wall clock per call"] = per_call(lambda: aa.linear_forward(x, [196, 320]))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(REPS):
    aa.linear_forward(x, [196, 320])
e1.record()
e1.synchronize()
res["whole call, HIP events"] = e0.elapsed_time(e1) / REPS * 1e3
# the pieces
res["_check_sizes + dtype lookup + _memory_format"] = per_call(lambda: (aa._check_sizes(x.shape, [196, 320]), aa._DTYPE_IDS[x.dtype], aa._memory_format(x)), sync=False)
res["torch.empty (channels_last output)"] = per_call(lambda: torch.empty((1, 3, 196, 320), dtype=torch.uint8, device=dev, memory_format=torch.channels_last))
key = (0, 0, 1, 1, 3, 438, 906, 196, 320, False, 0, 0.0, 0.0, 0, 0)
d = {key: 1}
res["plan key tuple + dict lookup"] = per_call(lambda: d.get((0, 0, 1, 1, 3, 438, 906, 196, 320, False, 0, 0.0, 0.0, dev.index, _lib.fused_epoch)), sync=False)
res["torch.cuda.current_device()"] = per_call(torch.cuda.current_device, sync=False)
res["torch.cuda.current_stream(dev).cuda_stream"] = per_call(lambda: torch.cuda.current_stream(dev).cuda_stream, sync=False)
res["x.data_ptr() x 2"] = per_call(lambda: (x.data_ptr(), x.data_ptr()), sync=False)
th = tables.get_table(_lib.FILTER_LINEAR, _lib.TABLE_PIL, 438, 196, False, 0.0, dev)
tw = tables.get_table(_lib.FILTER_LINEAR, _lib.TABLE_PIL, 906, 320, False, 0.0, dev)
ah, aw = th.axis(), tw.axis()
pah, paw = ctypes.byref(ah), ctypes.byref(aw)
out = torch.empty((1, 3, 196, 320), dtype=torch.uint8, device=dev, memory_format=torch.channels_last)
s = torch.cuda.current_stream(dev).cuda_stream
xp, op = x.data_ptr(), out.data_ptr()
res["ctypes call aa_resample_fwd_ex (argument marshalling + C dispatch + hipLaunchKernel), kernel queued"] = per_call(
    lambda: L.aa_resample_fwd_ex(xp, op, None, 0, _lib.U8, _lib.NHWC, 1, 3, 438, 906, pah, paw, 0, s))
res["ctypes call of a trivial C function (aa_abi_version)"] = per_call(L.aa_abi_version, sync=False)
res["ctypes call aa_workspace_bytes (10 scalars + 2 pointers, no launch)"] = per_call(
    lambda: L.aa_workspace_bytes(_lib.U8, _lib.NHWC, 1, 3, 438, 906, 196, 320, pah, paw), sync=False)
print(json.dumps({k: round(v, 2) for k, v in res.items()}, indent=1))
