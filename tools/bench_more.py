import sys, time, json, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from interpolate_antialiasing_amd import _lib, extension_interpolate as aa
def timeit(fn, steps=10, warm=3):
    for _ in range(warm): y = fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): y = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, y
def report(name, ms, alg_bytes, mpix):
    print(f"{name:58s} {ms:9.4f} ms  {alg_bytes/ms/1e6:8.1f} GB/s alg  {mpix/ms/1e3:10.1f} Mpix/s  [{_lib.last_variant()}]", flush=True)
torch.manual_seed(0)
# fp32 NCHW config A, batch 256
x = torch.rand(256, 3, 438, 906, device='cuda') * 255
ms, y = timeit(lambda: aa.linear_forward(x, [196, 320]))
report("fp32 NCHW [256,3,438,906]->[196,320] bilinear", ms, 256 * 5514576, 256 * 438 * 906 / 1e6)
xc = x.contiguous(memory_format=torch.channels_last)
ms, y = timeit(lambda: aa.linear_forward(xc, [196, 320]))
report("fp32 NHWC [256,3,438,906]->[196,320] bilinear", ms, 256 * 5514576, 256 * 438 * 906 / 1e6)
del x, xc
# config 2: fp32 [64,3,1024,1024] -> [224,224] bicubic
x = torch.rand(64, 3, 1024, 1024, device='cuda') * 255
ms, y = timeit(lambda: aa.cubic_forward(x, [224, 224]))
report("fp32 NCHW [64,3,1024,1024]->[224,224] bicubic (config 2)", ms, 843841536, 64 * 1024 * 1024 / 1e6)
del x
# config 4 shape on one GPU: u8 NHWC [1024,3,906,438] -> [320,196]
x = torch.randint(0, 256, (1024, 906, 438, 3), dtype=torch.uint8, device='cuda').permute(0, 3, 1, 2)
ms, y = timeit(lambda: aa.linear_forward(x, [320, 196]))
report("u8 NHWC [1024,3,906,438]->[320,196] bilinear (config 3 shard)", ms, 1024 * 1378644, 1024 * 438 * 906 / 1e6)
# u8 NCHW contiguous
xn = x.contiguous()
ms, y = timeit(lambda: aa.linear_forward(xn, [320, 196]))
report("u8 NCHW [1024,3,906,438]->[320,196] bilinear", ms, 1024 * 1378644, 1024 * 438 * 906 / 1e6)
# u8 bicubic NHWC
ms, y = timeit(lambda: aa.cubic_forward(x, [320, 196]))
report("u8 NHWC [1024,3,906,438]->[320,196] bicubic", ms, 1024 * 1378644, 1024 * 438 * 906 / 1e6)
# B=1 latency
x1 = x[:1].contiguous(memory_format=torch.channels_last)
ms, y = timeit(lambda: aa.linear_forward(x1, [320, 196]), steps=200, warm=20)
report("u8 NHWC B=1 latency (incl. python shim)", ms, 1378644, 438 * 906 / 1e6)
del x, xn
# config 5: backward, batch 256
g = torch.randn(256, 3, 196, 320, device='cuda')
ms, y = timeit(lambda: aa.linear_backward(g, [196, 320], [256, 3, 438, 906]))
report("bwd gather fp32 [256,3,196,320]->[256,3,438,906]", ms, 256 * 5514576, 256 * 438 * 906 / 1e6)
ms, y = timeit(lambda: aa.linear_backward(g, [196, 320], [256, 3, 438, 906], atomic=True))
report("bwd atomic fp32 (same)", ms, 256 * 5514576, 256 * 438 * 906 / 1e6)
