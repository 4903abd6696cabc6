#!/usr/bin/env python3
"""Generates tests/golden/reference_rms_norm.npz by EXECUTING the reference's rms_norm_kernel.

oracle/ref_build/build.sh extracts the kernel template from exllamav2_ext/cuda/rms_norm.cu (where it lies under
/root/reference; into the git-ignored oracle/_ref/) and rms_norm_driver.cpp runs it with rms_norm_cuda's launch shape:
one block of 1024 logical threads per row, fp32 square sums reduced by warp shuffles then across warps.
Recorded: y for seeded x, w at the model widths 256 / 2048 / 4096 / 5120 / 8192.

Run from the repo root:  python tests/golden/make_golden_rmsnorm.py
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "oracle", "_ref", "librmsnorm_ref.so")
CASES = [(1, 256), (3, 2048), (2, 4096), (2, 5120), (1, 8192)]          # (rows, dim)
EPS = 1e-5


def load():
    if os.path.isdir("/root/reference/exllamav2/exllamav2_ext/cuda"):
        subprocess.check_call([os.path.join(ROOT, "oracle", "ref_build", "build.sh")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(LIB)
    lib.ref_rms_norm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_float, ctypes.c_int, ctypes.c_int]
    lib.ref_rms_norm.restype = ctypes.c_int
    return lib


def inputs(i: int):
    rows, dim = CASES[i]
    rng = np.random.default_rng(800 + i)
    x = (rng.standard_normal((rows, dim)) * 3).astype(np.float16)
    if i == 2: x[0, 7] = 65504.0                                   # the clamp path
    w = (1 + 0.1 * rng.standard_normal(dim)).astype(np.float16)
    return x, w


def reference_rms_norm(lib, i: int) -> np.ndarray:
    x, w = inputs(i)
    y = np.zeros_like(x)
    assert lib.ref_rms_norm(x.ctypes.data, w.ctypes.data, y.ctypes.data, EPS, x.shape[0], x.shape[1]) == 0
    return y


def main():
    lib = load()
    fx = {f"y_{i}": reference_rms_norm(lib, i) for i in range(len(CASES))}
    path = os.path.join(ROOT, "tests", "golden", "reference_rms_norm.npz")
    np.savez_compressed(path, **fx)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    sys.exit(main())
