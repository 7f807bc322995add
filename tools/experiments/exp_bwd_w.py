import sys, os, subprocess
code = r'''
import sys, os, torch
sys.path.insert(0, os.getcwd())
from interpolate_antialiasing_amd import _lib, extension_interpolate as aa
def timed(fn, reps=60):
    for _ in range(40): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
g = torch.randn(256, 3, 196, 320, device="cuda")
out = []
for W in (896, 906):
    ms = timed(lambda: aa.linear_backward(g, [196, 320], [256, 3, 438, W]))
    out.append(f"{W}: {ms:.4f} ({256*3*438*W*4/ms/1e6:.0f} GB/s)")
print(os.path.basename(os.environ.get("AA_INTERP_LIB","stock")), "  ".join(out), flush=True)
'''
for lib in ("upbase", "upg2", "upg4", "upg16", "upbase"):
    env = dict(os.environ, AA_INTERP_LIB=os.path.join(os.getcwd(), "interpolate_antialiasing_amd", "csrc", f"libaa_interp_{lib}.so"))
    subprocess.run([sys.executable, "-c", code], env=env)
