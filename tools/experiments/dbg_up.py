import sys, os
sys.path.insert(0, os.getcwd())
import torch
from interpolate_antialiasing_amd import _lib, extension_interpolate as aa, tables
x = torch.randint(0, 256, (8, 438, 906, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
for size in ([1200, 1200], [1200, 120], [600, 1000], [500, 906]):
    y = aa.linear_forward(x, size)
    print(size, _lib.last_variant())
x = torch.randint(0, 256, (2, 23, 50, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
y = aa.linear_forward(x, [61, 128]); print(_lib.last_variant())
