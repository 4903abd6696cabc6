#!/usr/bin/env python3
"""Generates tests/golden/reference_act_mul.npz by EXECUTING the reference's act_mul_kernel (q_mlp_activation.cuh, a
header included as it lies under /root/reference; oracle/ref_build/act_mul_driver.cpp): x <- silu(x) * y in fp16 steps.
hexp / hrcp are evaluated as correctly rounded fp16 functions, so the fixture is a yardstick for the tolerance of the
activation tests (the oracle restates the reference's TORCH form, mlp.py:486-494: fp32 silu rounded once), not a pin.

Run from the repo root:  python tests/golden/make_golden_actmul.py
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "oracle", "_ref", "libactmul_ref.so")


def load():
    if os.path.isdir("/root/reference/exllamav2/exllamav2_ext/cuda"):
        subprocess.check_call([os.path.join(ROOT, "oracle", "ref_build", "build.sh")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(LIB)
    lib.ref_act_mul.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.ref_act_mul.restype = ctypes.c_int
    return lib


def inputs():
    rng = np.random.default_rng(2)
    return (rng.standard_normal((6, 1024)) * 2.5).astype(np.float16), rng.standard_normal((6, 1024)).astype(np.float16)


def reference_act_mul(lib) -> np.ndarray:
    g, u = inputs()
    x = g.copy()
    assert lib.ref_act_mul(x.ctypes.data, u.ctypes.data, g.shape[0], g.shape[1], 0) == 0
    return x


def main():
    path = os.path.join(ROOT, "tests", "golden", "reference_act_mul.npz")
    np.savez_compressed(path, silu_mul=reference_act_mul(load()))
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    sys.exit(main())
