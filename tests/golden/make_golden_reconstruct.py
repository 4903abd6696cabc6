#!/usr/bin/env python3
"""Generates tests/golden/reference_reconstruct.npz by EXECUTING the reference's q_matrix kernels.

oracle/ref_build/build.sh extracts the text of shuffle_kernel and reconstruct_kernel from
exllamav2_ext/cuda/q_matrix.cu (where it lies under /root/reference; into the git-ignored oracle/_ref/) and compiles it,
with matrix_view.cuh and quant/qdq_*.cuh straight from the reference tree, for the host: blocks run one after the other,
their threads as fibers (oracle/ref_build/simt_host.*).  For seeded EXL2 tensor sets of every bit width and mix, with and
without act-order, this script records the reference's reconstruct() output (EXL2, and GPTQ 4-bit incl. act-order g_idx): SHA-256 of the fp16 matrix for every case,
the full matrix for two small ones.  tests/test_oracle_ref.py checks oracle.exl2.exl2_reconstruct against it.

Run from the repo root:  python tests/golden/make_golden_reconstruct.py
"""
import ctypes
import hashlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "oracle", "_ref", "libqmatrix_ref.so")

# (K, N, [(bits, group rows, rows of that width)]) -- the shapes of tests/test_qmatrix.py
CASES = {
    "b4_g128": (256, 32, [(4, 128, 256)]), "b4_tail": (160, 32, [(4, 64, 160)]), "b2": (192, 32, [(2, 64, 192)]),
    "b3": (192, 32, [(3, 32, 192)]), "b5": (192, 32, [(5, 64, 192)]), "b6": (192, 32, [(6, 96, 192)]),
    "b8": (192, 32, [(8, 32, 192)]),
    "mixed_all": (800, 64, [(8, 32, 64), (6, 64, 128), (5, 32, 96), (4, 128, 256), (3, 64, 128), (2, 32, 128)]),
    "mixed_5_4": (1024, 64, [(5, 128, 128), (4, 128, 896)]),
}
FULL = ("b4_tail", "mixed_all")
# GPTQ 4-bit: (K, N, group size, act-order)
GPTQ_CASES = [(256, 32, 128, False), (256, 32, 64, True), (384, 48, 128, True), (160, 16, 32, False), (1024, 128, 128, True)]


def load():
    if os.path.isdir("/root/reference/exllamav2/exllamav2_ext/cuda"):
        subprocess.check_call([os.path.join(ROOT, "oracle", "ref_build", "build.sh")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(LIB)
    lib.ref_exl2_reconstruct.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    lib.ref_exl2_reconstruct.restype = ctypes.c_int
    lib.ref_gptq_reconstruct.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    lib.ref_gptq_reconstruct.restype = ctypes.c_int
    return lib


def reference_gptq_reconstruct(lib, t: dict) -> np.ndarray:
    """GPTQ tensors -> fp16 [K, N] by the reference's make_sequential_kernel (act-order), shuffle_kernel and
    reconstruct_gptq_kernel.  The row order handed to them is the stable counting sort of g_idx that make_sequential's
    host loop computes (q_matrix.cu:606-642), restated in oracle.exl2.gptq_sequential_perm."""
    from oracle import exl2 as OX
    qw = np.ascontiguousarray(t["qweight"]).view(np.uint32).copy()
    k, n = qw.shape[0] * 8, qw.shape[1]
    groups = t["qzeros"].shape[0]
    g_idx = np.asarray(t["g_idx"])
    sequential = np.array_equal(g_idx, np.arange(k) // (k // groups))
    perm = None if sequential else OX.gptq_sequential_perm(g_idx, groups)[0].astype(np.uint16)
    qz = np.ascontiguousarray(t["qzeros"]).view(np.uint32).copy()
    sc = np.ascontiguousarray(t["scales"]).view(np.uint16).copy()
    out = np.zeros((k, n), dtype=np.uint16)
    assert lib.ref_gptq_reconstruct(qw.ctypes.data, None if perm is None else perm.ctypes.data, qz.ctypes.data,
                                    sc.ctypes.data, k, n, groups, out.ctypes.data) == 0
    return out.view(np.float16)


def gptq_tensors(case) -> dict:
    from oracle import exl2 as OX
    k, n, gs, act = case
    return OX.synth_gptq(k, n, gs, seed=43, act_order=act)


def reference_reconstruct(lib, t: dict) -> np.ndarray:
    """on-disk EXL2 tensors -> fp16 [K, N] by the reference's shuffle_kernel + reconstruct_kernel.  The host-side
    preparation is the loader's: q_scale_max / 256 (ext.py:336), q_perm = argsort(q_invperm) (module.py:120), the group map
    (ext.py:301-316; checked against the reference's own make_group_map_py in tests/test_oracle.py)."""
    from oracle import exl2 as OX
    qw = np.ascontiguousarray(t["q_weight"]).view(np.uint32).copy()          # shuffled in place by the reference
    r, n = qw.shape
    groups = t["q_scale"].shape[0]
    gmap = OX.make_group_map(t["q_groups"], r).astype(np.uint16)
    k = gmap.shape[0] // 2
    smax = OX.exl2_prescale_scale_max(t["q_scale_max"]).view(np.uint16).copy()
    inv = t.get("q_invperm")
    perm = None if inv is None else np.argsort(np.asarray(inv).astype(np.int64), kind="stable").astype(np.uint16)
    qs = np.ascontiguousarray(t["q_scale"]).view(np.uint32).copy()
    qg = np.ascontiguousarray(t["q_groups"]).view(np.uint16).copy()
    out = np.zeros((k, n), dtype=np.uint16)
    rc = lib.ref_exl2_reconstruct(qw.ctypes.data, None if perm is None else perm.ctypes.data, qs.ctypes.data,
                                  smax.ctypes.data, qg.ctypes.data, gmap.ctypes.data, k, n, groups, out.ctypes.data)
    assert rc == 0
    return out.view(np.float16)


def case_tensors(name: str, act_order: bool) -> dict:
    from oracle import exl2 as OX
    k, n, spec = CASES[name]
    return OX.synth_exl2(k, n, spec, seed=41, act_order=act_order)


def main():
    lib = load()
    fx = {}
    for name in CASES:
        for act in (False, True):
            out = reference_reconstruct(lib, case_tensors(name, act))
            tag = f"{name}_{'act' if act else 'seq'}"
            fx["sha256_" + tag] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(out).tobytes()).digest(), dtype=np.uint8)
            if name in FULL and act:
                fx["full_" + tag] = out
    for case in GPTQ_CASES:
        out = reference_gptq_reconstruct(lib, gptq_tensors(case))
        tag = "gptq_%d_%d_%d_%s" % (case[0], case[1], case[2], "act" if case[3] else "seq")
        fx["sha256_" + tag] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(out).tobytes()).digest(), dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", "reference_reconstruct.npz")
    np.savez_compressed(path, **fx)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    sys.exit(main())
