#!/usr/bin/env python3
"""Generates tests/golden/reference_cache_addressing.npz by EXECUTING the reference's cache kernels.

The four addressing kernels of exllamav2_ext/cuda/cache.cu (fp16_to_q_kv[_paged]_kernel, q_to_fp16_kv[_paged]_kernel;
text extracted at build time into the git-ignored oracle/_ref/, see oracle/ref_build/build.sh) run on the host with the
grids of array_*_cuda, 256 logical threads per block.  Scenarios: paged append across a page boundary, token ranges that
must be widened to whole 512-element blocks (dim 128, 256) and ones that need not (dim 1024), 4 / 6 / 8-bit caches,
contiguous ranges; plus the FP8 codec kernels (cache.cu:20-141) and cache_rotate_kernel (:499-546).  Recorded: for the pack direction the full (mostly zero) code and scale tensors -- i.e. exactly WHICH
blocks the reference writes and what it writes; for the unpack direction SHA-256 of the fp16 outputs.

Run from the repo root:  python tests/golden/make_golden_cache_paged.py
"""
import ctypes
import hashlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "oracle", "_ref", "libcacheq_ref.so")
PS = 256

# paged: (wbits, pages, kv_heads, head_dim, q_len, cache_seqlens, block_table)
PAGED = [
    (4, 6, 1, 128, 3, [254, 600], [[3, 1, 0], [2, 4, 5]]),
    (6, 6, 2, 128, 5, [0, 509], [[5, 0, 1], [4, 2, 3]]),
    (8, 4, 8, 128, 2, [255, 17], [[1, 3], [0, 2]]),
]
# contiguous: (wbits, batch, seq tokens, kv_heads, head_dim, offset, width)
CONTIG = [(4, 2, 16, 2, 128, 4, 8), (4, 2, 16, 1, 128, 5, 2), (6, 1, 8, 2, 128, 2, 4), (8, 2, 8, 4, 128, 3, 1)]


def load():
    if os.path.isdir("/root/reference/exllamav2/exllamav2_ext/cuda"):
        subprocess.check_call([os.path.join(ROOT, "oracle", "ref_build", "build.sh")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(LIB)
    lib.ref_cache_paged.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2 + [ctypes.c_int] * 2
    lib.ref_cache_paged.restype = ctypes.c_int
    lib.ref_cache_contiguous.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5
    lib.ref_cache_contiguous.restype = ctypes.c_int
    lib.ref_cache_fp8.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4
    lib.ref_cache_fp8.restype = ctypes.c_int
    lib.ref_cache_rotate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int]
    lib.ref_cache_rotate.restype = ctypes.c_int
    return lib


def code_bytes(wbits: int, hd: int):
    """bytes per head of the K and of the V code tensors"""
    kb = hd // 2 if wbits == 4 else hd
    vb = hd if wbits == 8 else hd // 2
    return kb, vb


def kv_inputs(tag: str, shape):
    rng = np.random.default_rng(abs(hash(tag)) % (2 ** 31) if False else sum(map(ord, tag)) + 7000)
    return rng.standard_normal(shape).astype(np.float16), (rng.standard_normal(shape) * 0.5).astype(np.float16)


def q_inputs(tag: str, wbits: int, lead, kvh: int, hd: int):
    """random codes + positive fp16 scales for the unpack direction"""
    rng = np.random.default_rng(sum(map(ord, tag)) + 9000)
    kb, vb = code_bytes(wbits, hd)
    kq = rng.integers(0, 256, size=lead + (kvh, kb), dtype=np.uint8)
    vq = rng.integers(0, 256, size=lead + (kvh, vb), dtype=np.uint8)
    ks = rng.uniform(0.01, 0.2, size=lead + (kvh, hd // 32)).astype(np.float16)
    vs = rng.uniform(0.01, 0.2, size=lead + (kvh, hd // 32)).astype(np.float16)
    return kq, ks, vq, vs


def paged_pack(lib, i: int):
    wbits, pages, kvh, hd, q_len, seqlens, table = PAGED[i]
    k, v = kv_inputs(f"paged{i}", (pages, PS, kvh, hd))
    kb, vb = code_bytes(wbits, hd)
    kq = np.zeros((pages, PS, kvh, kb), np.uint8); vq = np.zeros((pages, PS, kvh, vb), np.uint8)
    ks = np.zeros((pages, PS, kvh, hd // 32), np.float16); vs = np.zeros_like(ks)
    sl = np.array(seqlens, np.int32); bt = np.array(table, np.int32)
    assert lib.ref_cache_paged(0, wbits, k.ctypes.data, kq.ctypes.data, ks.ctypes.data, v.ctypes.data, vq.ctypes.data,
                               vs.ctypes.data, len(seqlens), kvh * hd, bt.shape[1], sl.ctypes.data, bt.ctypes.data, PS, q_len) == 0
    return kq, ks, vq, vs


def paged_unpack(lib, i: int):
    wbits, pages, kvh, hd, q_len, seqlens, table = PAGED[i]
    kq, ks, vq, vs = q_inputs(f"paged{i}", wbits, (pages, PS), kvh, hd)
    ko = np.zeros((pages, PS, kvh, hd), np.float16); vo = np.zeros_like(ko)
    sl = (np.array(seqlens) + q_len).astype(np.int32); bt = np.array(table, np.int32)
    assert lib.ref_cache_paged(1, wbits, kq.ctypes.data, ko.ctypes.data, ks.ctypes.data, vq.ctypes.data, vo.ctypes.data,
                               vs.ctypes.data, len(seqlens), kvh * hd, bt.shape[1], sl.ctypes.data, bt.ctypes.data, PS, 0) == 0
    return ko, vo


def contig_pack(lib, i: int):
    wbits, b, T, kvh, hd, offset, width = CONTIG[i]
    k, v = kv_inputs(f"contig{i}", (b, T, kvh, hd))
    kb, vb = code_bytes(wbits, hd)
    kq = np.zeros((b, T, kvh, kb), np.uint8); vq = np.zeros((b, T, kvh, vb), np.uint8)
    ks = np.zeros((b, T, kvh, hd // 32), np.float16); vs = np.zeros_like(ks)
    assert lib.ref_cache_contiguous(0, wbits, k.ctypes.data, kq.ctypes.data, ks.ctypes.data, v.ctypes.data, vq.ctypes.data,
                                    vs.ctypes.data, b, kvh * hd, T, offset, width) == 0
    return kq, ks, vq, vs


def contig_unpack(lib, i: int):
    wbits, b, T, kvh, hd, offset, width = CONTIG[i]
    kq, ks, vq, vs = q_inputs(f"contig{i}", wbits, (b, T), kvh, hd)
    ko = np.zeros((b, T, kvh, hd), np.float16); vo = np.zeros_like(ko)
    assert lib.ref_cache_contiguous(1, wbits, kq.ctypes.data, ko.ctypes.data, ks.ctypes.data, vq.ctypes.data, vo.ctypes.data,
                                    vs.ctypes.data, b, kvh * hd, T, offset, width) == 0
    return ko, vo


# FP8 codec: (offset tokens, width tokens) on a [2, 16, 3, 4] cache (token size 12: ranges get rounded to 8 elements);
# page rotation: (pages, page elements (uint16; the reference kernel needs page bytes % 2048 == 0), order)
FP8_CASES = [(0, 7), (3, 5), (0, 16), (5, 0)]
ROTATE_CASES = [(6, 4096, [4, 1, 3]), (9, 5120, [8, 2, 7, 0, 5, 3, 1]), (3, 1024, [2]), (12, 8192, [11, 0, 10, 1, 9, 2, 8, 3, 7, 4])]


def fp8_input():
    x = (np.random.default_rng(11).standard_normal((2, 16, 3, 4)) * 4).astype(np.float16)
    x.reshape(-1)[:6] = np.array([np.inf, -np.inf, np.nan, -0.0, 6e-8, 65504.0], dtype=np.float16)
    return x


def reference_fp8(lib, i: int):
    off, wd = FP8_CASES[i]
    x = fp8_input()
    b, s, kvh, hd = x.shape
    out = np.full((b, s * kvh * hd), 0xA5, np.uint8)
    lib.ref_cache_fp8(0, x.ctypes.data, out.ctypes.data, s * kvh * hd, b, off * kvh * hd, wd * kvh * hd)
    back = np.full((b, s * kvh * hd), 7.0, np.float16)
    lib.ref_cache_fp8(1, out.ctypes.data, back.ctypes.data, s * kvh * hd, b, off * kvh * hd, wd * kvh * hd)
    return out, back


def rotate_input(i: int):
    n_pages, pe, order = ROTATE_CASES[i]
    return np.random.default_rng(50 + i).integers(0, 65536, size=(n_pages, pe), dtype=np.uint16)


def reference_rotate(lib, i: int):
    n_pages, pe, order = ROTATE_CASES[i]
    c = rotate_input(i).copy()
    temp = np.zeros(pe, np.uint16)
    o = np.array(order, np.uint32)
    lib.ref_cache_rotate(c.ctypes.data, o.ctypes.data, temp.ctypes.data, pe * 2, len(order))
    return c


def digest(a: np.ndarray) -> np.ndarray:
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def main():
    lib = load()
    fx = {}
    for kind, n, pack, unpack in (("paged", len(PAGED), paged_pack, paged_unpack), ("contig", len(CONTIG), contig_pack, contig_unpack)):
        for i in range(n):
            kq, ks, vq, vs = pack(lib, i)
            fx.update({f"{kind}{i}_kq": kq, f"{kind}{i}_ks": ks, f"{kind}{i}_vq": vq, f"{kind}{i}_vs": vs})
            ko, vo = unpack(lib, i)
            fx[f"{kind}{i}_ko_sha"] = digest(ko); fx[f"{kind}{i}_vo_sha"] = digest(vo)
            fx[f"{kind}{i}_ko_written"] = np.array([int((ko.view(np.uint16) != 0).sum()), int((vo.view(np.uint16) != 0).sum())])
    for i in range(len(FP8_CASES)):
        out, back = reference_fp8(lib, i)
        fx[f"fp8_{i}_codes"] = out; fx[f"fp8_{i}_back"] = back
    for i in range(len(ROTATE_CASES)):
        fx[f"rotate_{i}_sha"] = digest(reference_rotate(lib, i))
    path = os.path.join(ROOT, "tests", "golden", "reference_cache_addressing.npz")
    np.savez_compressed(path, **fx)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    sys.exit(main())
