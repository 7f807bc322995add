import sys, os, subprocess
code = r'''
import sys, os, torch
sys.path.insert(0, os.getcwd())
from interpolate_antialiasing_amd import _lib, extension_interpolate as aa
def timed(fn, reps=30):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
g = torch.randn(256, 3, 196, 320, device="cuda")
ms = timed(lambda: aa.linear_backward(g, [196, 320], [256, 3, 438, 906]))
x = torch.rand(64, 3, 1024, 1024, device="cuda") * 255
ms2 = timed(lambda: aa.cubic_forward(x, [224, 224]))
x = torch.rand(256, 3, 438, 906, device="cuda") * 255
ms3 = timed(lambda: aa.linear_forward(x, [196, 320]))
print(os.environ.get("AA_FUSED_YBANDS"), os.environ.get("AA_F32_SPB"), f"bwd {ms:.4f}  c2 {ms2:.4f}  c0f32 {ms3:.4f}", flush=True)
'''
for yb in (None, "1", "2", "3", "4", "6", "10", "16"):
    for spb in (None, "1", "4"):
        env = dict(os.environ)
        if yb: env["AA_FUSED_YBANDS"] = yb
        if spb: env["AA_F32_SPB"] = spb
        if yb is None and spb is not None and spb != "4": continue
        subprocess.run([sys.executable, "-c", code], env=env)
