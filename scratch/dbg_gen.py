import sys, torch
sys.path.insert(0, '.')
from interpolate_antialiasing_amd import _lib, extension_interpolate as aa
torch.manual_seed(4)
x = torch.randint(0, 256, (7, 438, 906, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
_lib.set_fused(0); ref = aa.linear_forward(x, [196, 320])
for mode in (1, 2, 3):
    _lib.set_fused(mode); y = aa.linear_forward(x, [196, 320])
    d = (y.int() - ref.int()).abs()
    idx = d.nonzero()
    print(mode, _lib.last_variant(), "nbad", len(idx), idx[:6].tolist())
