#!/usr/bin/env python3
"""Generates tests/golden/reference_cache_q.npz by EXECUTING the reference's quantized-KV-cache codec.

oracle/ref_build/build.sh compiles exllamav2_ext/cuda/cache_q.cuh from where it lies under /root/reference into
oracle/_ref/libcacheq_ref.so: its device functions fp16_to_q<wbits> / q_to_fp16<wbits> run with 256 logical threads per
512-element block (warp shuffles over 32 lanes, __syncthreads) on a host fiber scheduler (oracle/ref_build/simt_host.*).
Recorded for seeded blocks: codes + scales for 4 and 8 bits, and the dequantized values of those codes.

Run from the repo root:  python tests/golden/make_golden_cacheq.py
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "oracle", "_ref", "libcacheq_ref.so")


def load():
    if not os.path.exists(LIB) or os.path.isdir("/root/reference/exllamav2/exllamav2_ext/cuda"):
        subprocess.check_call([os.path.join(ROOT, "oracle", "ref_build", "build.sh")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(LIB)
    for f in (lib.ref_cache_fp16_to_q, lib.ref_cache_q_to_fp16):
        f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        f.restype = ctypes.c_int
    return lib


def pack_blocks(lib, wbits: int, x: np.ndarray):
    """x fp16 [blocks, 512] -> codes uint8 [blocks, 256 | 512], scales fp16 [blocks, 16]"""
    nb = x.shape[0]
    codes = np.zeros((nb, 256 if wbits == 4 else 512), dtype=np.uint8)
    scales = np.zeros((nb, 16), dtype=np.float16)
    for b in range(nb):
        xb = np.ascontiguousarray(x[b]); c = codes[b]; s = scales[b]
        assert lib.ref_cache_fp16_to_q(wbits, xb.ctypes.data, c.ctypes.data, s.ctypes.data) == 0
    return codes, scales


def unpack_blocks(lib, wbits: int, codes: np.ndarray, scales: np.ndarray) -> np.ndarray:
    nb = codes.shape[0]
    out = np.zeros((nb, 512), dtype=np.float16)
    for b in range(nb):
        c = np.ascontiguousarray(codes[b]); s = np.ascontiguousarray(scales[b]); o = out[b]
        assert lib.ref_cache_q_to_fp16(wbits, c.ctypes.data, s.ctypes.data, o.ctypes.data) == 0
    return out


def blocks(seed: int, n: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((n, 512)) * rng.uniform(0.05, 6.0, size=(n, 1))).astype(np.float16)
    x[0] = 0                                                    # all-zero block: absmax 0 -> 0/0 -> code 0 (NaN -> 0)
    x[1, :32] = 0                                               # one zero 32-element group inside a live block
    x[2, 5] = 60000.0                                           # near the fp16 maximum: the butterfly overflows to inf
    x[3] = np.float16(6e-8)                                     # subnormals
    x[4, ::2] = 1.0; x[4, 1::2] = -1.0                          # exact ties in the rounding
    return x


def main():
    lib = load()
    x = blocks(20260921, 40)
    fx = {"x": x}
    for wbits in (4, 8):
        codes, scales = pack_blocks(lib, wbits, x)
        fx[f"codes_{wbits}"] = codes
        fx[f"scales_{wbits}"] = scales
        fx[f"unpacked_{wbits}"] = unpack_blocks(lib, wbits, codes, scales)
    path = os.path.join(ROOT, "tests", "golden", "reference_cache_q.npz")
    np.savez_compressed(path, **fx)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    sys.exit(main())
