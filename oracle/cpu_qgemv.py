"""ctypes front of oracle/cpu_qgemv.c: the multi-threaded C restatement of the EXL2 decode GEMV on the on-disk tensors
(dequantize on the fly, fp32) -- variant A of the CPU baseline.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/__init__.py): used by bench.py's cpu_baseline leg and checked against
oracle/exl2.py in tests/test_oracle.py; never imported by the product.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import exl2 as OX

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpu_qgemv.c")
LIB = os.path.join(HERE, "libcpu_qgemv.so")


def build(force: bool = False) -> str:
    """gcc -O3 -mavx2 -mfma (no -march=native: the library is built in one container and run on another host)."""
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O3", "-mavx2", "-mfma", "-std=gnu11", "-fPIC", "-shared", "-pthread", SRC, "-o", LIB])
    return LIB


class _QMat(C.Structure):
    _fields_ = [("qw", C.c_void_p), ("scales", C.c_void_p), ("perm", C.c_void_p), ("gtab", C.c_void_p),
                ("G", C.c_int), ("R", C.c_int), ("N", C.c_int), ("K", C.c_int)]


class Matrix:
    """The tensors of one EXL2 linear in the form the C code reads; holds the arrays alive."""

    def __init__(self, t: dict, copy: bool = False):
        qw = np.ascontiguousarray(np.asarray(t["q_weight"]).view(np.uint32))
        self.qw = qw.copy() if copy else qw
        r, n = self.qw.shape
        groups = OX.group_table(np.asarray(t["q_groups"]), r)
        self.gtab = np.ascontiguousarray(np.array([[b, q0, rows] for b, q0, rows in groups], dtype=np.int32))
        self.k = int(sum(g[2] for g in groups))
        # 4-bit group scales decoded like the kernels do (qdq_util.cuh:24-30), kept as fp32 of the fp16 values
        self.scales = np.ascontiguousarray(OX.exl2_scales(np.asarray(t["q_scale"]),
                                                          OX.exl2_prescale_scale_max(np.asarray(t["q_scale_max"]), 1.0)).astype(np.float32))
        inv = t.get("q_invperm")
        if inv is None:
            perm = np.arange(self.k, dtype=np.int32)
        else:
            perm = np.argsort(np.asarray(inv).astype(np.int64), kind="stable").astype(np.int32)      # module.py:120
        self.perm = np.ascontiguousarray(perm)
        self.n = n
        self.c = _QMat(self.qw.ctypes.data, self.scales.ctypes.data, self.perm.ctypes.data, self.gtab.ctypes.data,
                       len(groups), r, n, self.k)

    def clone(self) -> "Matrix":
        """Same matrix in its own memory (the unsampled baseline streams distinct bytes for every layer)."""
        m = Matrix.__new__(Matrix)
        m.qw, m.scales, m.perm, m.gtab = self.qw.copy(), self.scales.copy(), self.perm, self.gtab
        m.k, m.n = self.k, self.n
        m.c = _QMat(m.qw.ctypes.data, m.scales.ctypes.data, m.perm.ctypes.data, m.gtab.ctypes.data,
                    self.c.G, self.c.R, self.c.N, self.c.K)
        return m

    def nbytes(self) -> int:
        return self.qw.nbytes + self.scales.nbytes + self.perm.nbytes


class Pool:
    def __init__(self, threads: int, n_cap: int):
        self.lib = C.CDLL(build())
        self.lib.cpu_qgemv_init.argtypes = [C.c_int, C.c_int]
        self.lib.cpu_qgemv_init.restype = C.c_int
        self.lib.cpu_qgemv.argtypes = [C.POINTER(_QMat), C.c_void_p, C.c_void_p]
        self.lib.cpu_qgemv.restype = C.c_int
        self.lib.cpu_qgemv_shutdown.restype = None
        got = self.lib.cpu_qgemv_init(int(threads), int(n_cap))
        if got <= 0:
            raise RuntimeError(f"cpu_qgemv_init failed ({got}): host without AVX2 + FMA, or out of memory")
        self.threads = got

    def gemv(self, m: Matrix, x: np.ndarray, out: np.ndarray | None = None) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.shape == (m.k,)
        y = np.empty(m.n, dtype=np.float32) if out is None else out
        rc = self.lib.cpu_qgemv(C.byref(m.c), x.ctypes.data, y.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"cpu_qgemv failed ({rc})")
        return y

    def close(self):
        self.lib.cpu_qgemv_shutdown()
