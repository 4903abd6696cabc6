#!/usr/bin/env python3
"""Generates tests/golden/reference_sampling.npz by EXECUTING the reference's CPU sampler (cpp/sampling.cpp compiled as
it lies under /root/reference; oracle/ref_build/sampling_driver.cpp drives its stages in sample_basic's order,
ext_sampling.cpp:137-296): seeded logits (fp16-valued, so ties occur like on real fp16 logits), a grid of temperature /
top-k / top-p / min-p settings, logit filters, batches of 3 rows (the batch random recurrence).  Two outputs per case:
the portable softmax (`softmax_cpu_nonavx2`: the bit-level pin of oracle/sampling.py) and `softmax_cpu` as dispatched on
this host (AVX2 here), for the record.

Run from the repo root:  python tests/golden/make_golden_sampling.py
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "oracle", "_ref", "libsampling_ref.so")

# (temperature, top_k, top_p, min_p)
SETTINGS = [(0.8, 50, 0.8, 0.0), (1.0, 40, 0.0, 0.0), (0.7, 100, 0.9, 0.05), (1.3, 2, 0.0, 0.0), (1.0, 1, 0.0, 0.0),
            (0.005, 50, 0.8, 0.0), (0.6, 500, 0.95, 0.02), (1.0, 8, 0.5, 0.3), (2.0, 64, 0.0, 0.1), (0.9, 20, 0.3, 0.0)]
VOCAB = 1000


def load():
    if os.path.isdir("/root/reference/exllamav2/exllamav2_ext/cpp"):
        subprocess.check_call([os.path.join(ROOT, "oracle", "ref_build", "build.sh")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(LIB)
    lib.ref_sample_basic.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_int,
                                     ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p]
    lib.ref_sample_basic.restype = ctypes.c_int
    lib.ref_apply_rep_penalty.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_int]
    lib.ref_apply_rep_penalty.restype = ctypes.c_int
    return lib


# (penalty, sustain, decay, frequency, presence)
REP_SETTINGS = [(1.025, -1, 0, 0.0, 0.0), (1.2, 16, 8, 0.1, 0.05), (1.1, 4, 40, 0.0, 0.3), (1.0, -1, 0, 0.25, 0.0)]


def rep_cases():
    rng = np.random.default_rng(17)
    seq = rng.integers(0, 260, size=(2, 48)).astype(np.int64)        # ids >= 256 fall outside the vocabulary: skipped
    lg = rng.standard_normal((2, 256)).astype(np.float32) * 3
    return seq, lg


def run_reference_rep_penalty(lib, seq, lg, st):
    out = lg.copy()
    s = np.ascontiguousarray(seq.astype(np.uint64))
    assert lib.ref_apply_rep_penalty(s.ctypes.data, seq.shape[0], seq.shape[1], st[0], st[1], st[2], st[3], st[4],
                                     out.ctypes.data, lg.shape[1]) == 0
    return out


def cases():
    """[(logits fp32 [3, VOCAB] with fp16 values, filter or None, (T, k, p, mp), random)]"""
    rng = np.random.default_rng(11)
    out = []
    for ci, st in enumerate(SETTINGS):
        for rep in range(4):
            scale = [1.5, 3.0, 6.0, 0.25][rep]
            lg = (rng.standard_normal((3, VOCAB)) * scale).astype(np.float16)
            if rep == 1:
                lg = np.round(lg.astype(np.float32) * 4) / 4          # coarse grid: many exact ties at the top-k boundary
            flt = None
            if rep == 2:
                flt = rng.random((3, VOCAB)) < 0.7
                flt[:, 0] = True
            out.append((np.ascontiguousarray(lg, dtype=np.float32), flt, st, float(rng.random())))
    return out


def run_reference(lib, logits, flt, st, random, dispatch):
    bsz, vocab = logits.shape
    tok = np.zeros(bsz, dtype=np.int32)
    pr = np.zeros(bsz, dtype=np.float32)
    nc = np.zeros(bsz, dtype=np.int32)
    f8 = None if flt is None else np.ascontiguousarray(flt, dtype=np.uint8)
    rc = lib.ref_sample_basic(logits.ctypes.data, bsz, vocab, None if f8 is None else f8.ctypes.data, st[0], st[1], st[2], st[3],
                              random, dispatch, tok.ctypes.data, pr.ctypes.data, nc.ctypes.data)
    assert rc == 0
    return tok, pr, nc


def main():
    lib = load()
    toks, prs, ncs, toks_d = [], [], [], []
    for lg, flt, st, rnd in cases():
        t, p, n = run_reference(lib, lg, flt, st, rnd, 0)
        td, _, _ = run_reference(lib, lg, flt, st, rnd, 1)
        toks.append(t); prs.append(p); ncs.append(n); toks_d.append(td)
    path = os.path.join(ROOT, "tests", "golden", "reference_sampling.npz")
    seq, lg = rep_cases()
    rep = np.stack([run_reference_rep_penalty(lib, seq, lg, st) for st in REP_SETTINGS])
    np.savez_compressed(path, tokens=np.stack(toks), probs=np.stack(prs), num_candidates=np.stack(ncs),
                        tokens_dispatch=np.stack(toks_d), rep_penalty_logits=rep)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    sys.exit(main())
