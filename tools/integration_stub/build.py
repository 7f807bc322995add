#!/usr/bin/env python3
"""Builds tools/integration_stub/extension_interpolate_amd.cpp — the pybind11 module INTEGRATION.md shows — against the installed
PyTorch headers and libaa_interp.so, in-tree (build/ next to this file), exactly the way the reference's test.py:322 builds its own
extension (torch.utils.cpp_extension.load).  Plain g++: the module holds no device code.  Returns the imported module."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
NAME = "aa_interp_amd_stub"


def build(verbose: bool = False):
    from torch.utils.cpp_extension import load

    csrc = os.path.join(ROOT, "interpolate_antialiasing_amd", "csrc")
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    return load(name=NAME, sources=[os.path.join(HERE, "extension_interpolate_amd.cpp")], build_directory=bdir, verbose=verbose,
                extra_include_paths=[os.path.join(ROOT, "include"), "/opt/rocm/include"],
                extra_cflags=["-O2", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"],
                extra_ldflags=[f"-L{csrc}", "-laa_interp", f"-Wl,-rpath,{csrc}", "-L/opt/rocm/lib", "-lamdhip64"], with_cuda=False)


if __name__ == "__main__":
    m = build(verbose=True)
    print("built", m.__file__, [n for n in dir(m) if not n.startswith("_")])
