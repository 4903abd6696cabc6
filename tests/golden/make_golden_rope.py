#!/usr/bin/env python3
"""Generates tests/golden/reference_rope.npz by EXECUTING the reference's RoPE device functions.

oracle/ref_build/build.sh extracts rope_cuda_arr_neox / rope_cuda_arr_gptj from exllamav2_ext/cuda/rope.cu (where it lies
under /root/reference; into the git-ignored oracle/_ref/) and compiles them for the host; rope_driver.cpp runs them with
rope_cuda's launch shape (block 32 x 4).  Recorded for seeded inputs: the rotated tensors, NeoX and GPT-J style, scalar
past_len and per-sequence past_lens.  sin / cos tables are the oracle's (their construction is pinned separately against
the reference's prepare_sincos, tests/golden/make_golden.py).

Run from the repo root:  python tests/golden/make_golden_rope.py
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "oracle", "_ref", "librope_ref.so")

# (batch, q_len, heads, head_dim, past_len, past_lens or None, neox)
CASES = [(1, 1, 4, 128, 17, None, True), (2, 3, 2, 64, 0, [5, 40], True), (1, 5, 8, 128, 100, None, False),
         (3, 1, 2, 64, 2, [0, 9, 31], False), (1, 2, 1, 256, 250, None, True)]


def load():
    if os.path.isdir("/root/reference/exllamav2/exllamav2_ext/cuda"):
        subprocess.check_call([os.path.join(ROOT, "oracle", "ref_build", "build.sh")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(LIB)
    lib.ref_rope.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int]
    lib.ref_rope.restype = ctypes.c_int
    return lib


def inputs(i: int):
    from oracle import modules as OM
    b, s, nh, hd, past, lens, neox = CASES[i]
    x = (np.random.default_rng(300 + i).standard_normal((b, s, nh, hd)) * 2).astype(np.float16)
    sin, cos = OM.rope_tables(512, hd, 10000.0, neox)
    return x, np.ascontiguousarray(sin), np.ascontiguousarray(cos)


def reference_rope(lib, i: int) -> np.ndarray:
    b, s, nh, hd, past, lens, neox = CASES[i]
    x, sin, cos = inputs(i)
    x = np.ascontiguousarray(x).copy()
    pl = None if lens is None else np.array(lens, dtype=np.int32)
    assert lib.ref_rope(x.ctypes.data, sin.ctypes.data, cos.ctypes.data, b, s * nh, hd, nh, past,
                        None if pl is None else pl.ctypes.data, int(neox)) == 0
    return x


def main():
    lib = load()
    fx = {f"out_{i}": reference_rope(lib, i) for i in range(len(CASES))}
    path = os.path.join(ROOT, "tests", "golden", "reference_rope.npz")
    np.savez_compressed(path, **fx)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    sys.exit(main())
