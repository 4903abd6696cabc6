#!/usr/bin/env python3
"""Generates tests/golden/reference_qdq_decode.npz by EXECUTING the reference's own decode code.

oracle/ref_build/build.sh compiles exllamav2_ext/cuda/quant/qdq_{2,3,4,5,6,8}.cuh + qdq_util.cuh, from where they lie
under /root/reference, with a host shim for the CUDA fp16 vocabulary (oracle/_ref/libqdq_ref.so).  For seeded random
packed words this script records what the reference's load-time shuffle followed by its kernel-side dequant returns
(fp16 bit patterns of q - 2^(bits-1) per row), its dq_scale() for every scale code, and its GPTQ (q - zero) decode.
The fixture travels with the repo; tests/test_oracle_ref.py checks oracle/exl2.py against it everywhere and against
the freshly built library wherever /root/reference exists.

Run from the repo root:  python tests/golden/make_golden_qdq.py
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "oracle", "_ref", "libqdq_ref.so")
BITS = (2, 3, 4, 5, 6, 8)


def load():
    subprocess.check_call([os.path.join(ROOT, "oracle", "ref_build", "build.sh")])
    lib = ctypes.CDLL(LIB)
    lib.ref_decode_column32.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.ref_decode_column32.restype = ctypes.c_int
    lib.ref_dq_scale.argtypes = [ctypes.c_int, ctypes.c_uint16]
    lib.ref_dq_scale.restype = ctypes.c_uint16
    lib.ref_decode_gptq8.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    lib.ref_decode_gptq8.restype = None
    lib.ref_decode_gptq8_scaled.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint16, ctypes.c_void_p]
    lib.ref_decode_gptq8_scaled.restype = None
    return lib


def decode_columns(lib, bits: int, words: np.ndarray) -> np.ndarray:
    """words uint32 [bits, n] (one 32-row chunk of n columns) -> fp16 bit patterns uint16 [32, n]."""
    n = words.shape[1]
    out = np.zeros((32, n), dtype=np.uint16)
    col = np.zeros(bits, dtype=np.uint32)
    o = np.zeros(32, dtype=np.uint16)
    for c in range(n):
        col[:] = words[:, c]
        assert lib.ref_decode_column32(bits, col.ctypes.data, o.ctypes.data) == 0
        out[:, c] = o
    return out


def edge_words(bits: int) -> np.ndarray:
    cols = [np.zeros(bits, np.uint32), np.full(bits, 0xFFFFFFFF, np.uint32)]
    for i in range(bits):
        for b in (0, 15, 16, 31):
            w = np.zeros(bits, np.uint32); w[i] = np.uint32(1) << np.uint32(b); cols.append(w)
    return np.stack(cols, axis=1)


def main():
    lib = load()
    rng = np.random.default_rng(20260921)
    fx = {}
    for bits in BITS:
        words = np.concatenate([edge_words(bits), rng.integers(0, 2 ** 32, size=(bits, 192), dtype=np.uint64).astype(np.uint32)], axis=1)
        fx[f"words_{bits}"] = words
        fx[f"decoded_{bits}"] = decode_columns(lib, bits, words)
    maxes = np.concatenate([np.array([0.0, 1.0, 6.1e-5, 255.0], dtype=np.float16),
                            (rng.uniform(1e-4, 0.05, size=60)).astype(np.float16)])
    fx["scale_max"] = maxes
    fx["dq_scale"] = np.array([[lib.ref_dq_scale(qs, int(m)) for m in maxes.view(np.uint16)] for qs in range(16)], dtype=np.uint16)
    gw = rng.integers(0, 2 ** 32, size=64, dtype=np.uint64).astype(np.uint32)
    gz = rng.integers(1, 17, size=64).astype(np.uint32)
    gs = rng.uniform(0.005, 0.02, size=64).astype(np.float16)
    g_out = np.zeros((64, 8), dtype=np.uint16)
    g_out_s = np.zeros((64, 8), dtype=np.uint16)
    o = np.zeros(8, dtype=np.uint16)
    for i in range(64):
        lib.ref_decode_gptq8(int(gw[i]), int(gz[i]), o.ctypes.data); g_out[i] = o
        lib.ref_decode_gptq8_scaled(int(gw[i]), int(gz[i]), int(gs.view(np.uint16)[i]), o.ctypes.data); g_out_s[i] = o
    fx.update(gptq_words=gw, gptq_zeros=gz, gptq_scales=gs, gptq_decoded=g_out, gptq_decoded_scaled=g_out_s)
    path = os.path.join(ROOT, "tests", "golden", "reference_qdq_decode.npz")
    np.savez_compressed(path, **fx)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    sys.exit(main())
