"""Shared helpers for the parity tests."""
import numpy as np
import torch

from oracle import exl2 as OX


def exl2_to_torch(be, t: dict) -> dict:
    """on-disk numpy tensors -> the dict the reference's loader hands to ext.make_q_matrix (module.py:116-121)."""
    w = {}
    for k, v in t.items():
        w[k] = be.t(v)
    if "q_invperm" in w:
        w["q_perm"] = torch.argsort(w["q_invperm"].cpu().long()).to(torch.int32).to(be.device)   # module.py:120
    return w


def gptq_to_torch(be, t: dict) -> dict:
    return {k: be.t(v) for k, v in t.items()}


def make_exl2(be, k, n, spec, seed=0, act_order=True, bias=False):
    t = OX.synth_exl2(k, n, spec, seed=seed, act_order=act_order, bias=bias)
    ref = OX.exl2_reconstruct(t)
    w = exl2_to_torch(be, t)
    h = be.ext.make_q_matrix_from_dict(w, None)
    return t, ref, w, h


def half_tol(ref64: np.ndarray, k: int) -> np.ndarray:
    """fp16 tolerance of a K-term fp32-accumulated dot product rounded to fp16: half an fp16 ulp of the result
    (2^-11 relative) plus fp32 summation noise (<= K * 2^-24 of the absolute row sum), with a small absolute floor."""
    return np.abs(ref64) * 2.0 ** -10 + 1e-3
