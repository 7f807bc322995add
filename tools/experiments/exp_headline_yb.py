import sys, os, subprocess
code = r'''
import sys, os, torch
sys.path.insert(0, os.getcwd())
from interpolate_antialiasing_amd import _lib, extension_interpolate as aa
def timed(fn, reps=100):
    for _ in range(60): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
x = torch.randint(0, 256, (1024, 438, 906, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
ms = timed(lambda: aa.linear_forward(x, [196, 320]))
x2 = torch.randint(0, 256, (1024, 906, 438, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
ms2 = timed(lambda: aa.linear_forward(x2, [320, 196]))
print(os.environ.get("AA_FUSED_YBANDS"), os.environ.get("AA_V3_SPB"), f"headline {ms:.4f}  shard3 {ms2:.4f}", flush=True)
'''
for yb in (None, "1", "2", "3", "4", "5", "6", "8", "12"):
    for spb in (None, "1"):
        env = dict(os.environ)
        if yb: env["AA_FUSED_YBANDS"] = yb
        if spb: env["AA_V3_SPB"] = spb
        subprocess.run([sys.executable, "-c", code], env=env)
