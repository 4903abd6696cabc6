#!/usr/bin/env python3
"""Generates tests/golden/reference_moe_routing.npz by EXECUTING the reference's MoE routing kernels.

exllamav2_ext/cuda/q_mlp_softmax.cuh (softmax{4,8,16}_topk_norm_kernel) is a header: oracle/ref_build/ includes it as it lies
under /root/reference and runs it with the launch shape of q_mlp.cu:365-383.  Recorded for seeded fp16 router logits: the
in-place routing weights (softmax, keep top-k, renormalise).

Run from the repo root:  python tests/golden/make_golden_moe.py
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "oracle", "_ref", "libmoe_ref.so")
CASES = [(8, 2, 70), (4, 2, 33), (16, 4, 40), (8, 1, 5)]            # (experts, top-k, rows)


def load():
    if os.path.isdir("/root/reference/exllamav2/exllamav2_ext/cuda"):
        subprocess.check_call([os.path.join(ROOT, "oracle", "ref_build", "build.sh")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(LIB)
    lib.ref_moe_softmax_topk.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.ref_moe_softmax_topk.restype = ctypes.c_int
    return lib


def logits(i: int) -> np.ndarray:
    e, k, rows = CASES[i]
    return (np.random.default_rng(600 + i).standard_normal((rows, e)) * 2.5).astype(np.float16)


def reference_route(lib, i: int) -> np.ndarray:
    e, k, rows = CASES[i]
    x = logits(i).copy()
    assert lib.ref_moe_softmax_topk(x.ctypes.data, rows, e, k) == 0
    return x


def main():
    lib = load()
    fx = {f"w_{i}": reference_route(lib, i) for i in range(len(CASES))}
    path = os.path.join(ROOT, "tests", "golden", "reference_moe_routing.npz")
    np.savez_compressed(path, **fx)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    sys.exit(main())
