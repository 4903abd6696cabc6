"""Oracle: the reference's CPU sampler for the settings the device sampler covers (numpy fp32 scalars + heapq).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference files restated here (under /root/reference/exllamav2/exllamav2_ext/):
  * order of the stages, greedy short-cut, 0.9998 scaling of the random point, batch random recurrence
    .................. ext_sampling.cpp:137-296 (sample_basic)
  * softmax .......... cpp/sampling.cpp:113-176 (softmax_cpu_nonavx2: first maximum, expf((l - max) / T), sequential fp32 sum,
                       multiply by 1 / sum, filtered entries = 0)
  * top-k ............ cpp/sampling.cpp:443-520 (k = 1: swap the arg-max to the front; 2 <= k <= 500: min-heap of (p, index)
                       pairs seeded with the first k entries, a later entry replaces the minimum only if STRICTLY larger,
                       heap emptied from the back -> descending (p, index) order)
  * normalize ........ cpp/sampling.cpp:265-281 (sequential fp32 sum, multiply by 1 / sum)
  * top-p ............ cpp/sampling.cpp:524-566 (heap walk with the `sum > top_p` pops, 1e-6 floor)
  * min-p ............ cpp/sampling.cpp:620-640 + keep_threshold :569-592 (in-place partition, swaps from the back)
  * multinomial ...... cpp/sampling.cpp:872-915 (sequential accumulation, roll-back over zero entries)

Pinned by execution: oracle/_ref/libsampling_ref.so is the reference's own cpp/sampling.cpp compiled for the host
(oracle/ref_build/sampling_driver.cpp); tests/golden/reference_sampling.npz holds its outputs
(tests/golden/make_golden_sampling.py), tests/test_sampling.py checks this file against both, token for token and
probability bit for probability bit.  expf is libm's (ctypes), the function the compiled reference calls.
"""
from __future__ import annotations

import ctypes
import ctypes.util
import heapq

import numpy as np

F32 = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.expf.restype = ctypes.c_float
_libm.expf.argtypes = [ctypes.c_float]

TOP_K_HEAP_THRESHOLD = 500                      # cpp/sampling.cpp:12


def _expf(x: np.float32) -> np.float32:
    return F32(_libm.expf(float(x)))


def softmax(logits: np.ndarray, temperature: float, flt: np.ndarray | None):
    """cpp/sampling.cpp:113-176.  Returns (probs fp32 [vocab], index of the first maximum among the unfiltered)."""
    v = logits.shape[0]
    l32 = logits.astype(F32)
    itemp = F32(1.0) / F32(temperature)
    maxl, maxi = F32(-1e38), 0
    for i in range(v):
        if flt is not None and not flt[i]:
            continue
        if l32[i] > maxl:
            maxl, maxi = l32[i], i
    out = np.zeros(v, dtype=F32)
    esum = F32(0.0)
    for i in range(v):
        if flt is not None and not flt[i]:
            continue
        e = _expf(F32(l32[i] - maxl) * itemp)
        out[i] = e
        esum = F32(esum + e)
    isum = F32(1.0) / esum
    for i in range(v):
        out[i] = F32(out[i] * isum) if (flt is None or flt[i]) else F32(0.0)
    return out, maxi


def normalize(P: np.ndarray, n: int) -> None:
    """cpp/sampling.cpp:265-281, in place on the first n entries."""
    s = F32(0.0)
    for i in range(n):
        s = F32(s + P[i])
    isum = F32(1.0) / s
    for i in range(n):
        P[i] = F32(P[i] * isum)


def top_k(P: np.ndarray, I: np.ndarray, n: int, k: int, maxlogit: int) -> int:
    """cpp/sampling.cpp:443-520, k <= 500, in place: the k kept entries land in positions 0..k-1 in the reference's order,
    everything from position k on keeps what it held (later stages can reach it: see min_p)."""
    if k == 1:
        P[0], P[maxlogit] = P[maxlogit], P[0]
        I[0], I[maxlogit] = I[maxlogit], I[0]
        return 1
    assert k <= TOP_K_HEAP_THRESHOLD, "the quicksort regime (k > 500) is not restated"
    heap = [(P[i], int(I[i])) for i in range(k)]
    heapq.heapify(heap)
    t = heap[0][0]
    for i in range(k, n):
        p = P[i]
        if p > t:
            heapq.heapreplace(heap, (p, int(I[i])))
            t = heap[0][0]
    for j in range(k - 1, -1, -1):                       # ascending (p, index) pops fill from the back
        P[j], I[j] = heapq.heappop(heap)
    return k


def top_p(P: np.ndarray, I: np.ndarray, n: int, tp: float):
    """cpp/sampling.cpp:524-566, in place.  Returns (count, margin): margin = how far the closest `sum > top_p` decision
    was from flipping (the tests skip rows whose outcome hinges on the last bits of a sum)."""
    tp = F32(tp)
    heap: list = []
    s = F32(0.0)
    margin = 1.0
    for i in range(n):
        p = P[i]
        if p < F32(1e-6):
            continue
        margin = min(margin, abs(float(s) - float(tp)))
        if s > tp and p < heap[0][0]:
            continue
        heapq.heappush(heap, (p, int(I[i])))
        s = F32(s + p)
        margin = min(margin, abs(float(s) - float(tp)))
        while s > tp and len(heap) > 1:
            s = F32(s - heap[0][0])
            heapq.heappop(heap)
            margin = min(margin, abs(float(s) - float(tp)))
    k = len(heap)
    for j in range(k - 1, -1, -1):
        P[j], I[j] = heapq.heappop(heap)
    return k, margin


def keep_threshold(P: np.ndarray, I: np.ndarray, n: int, threshold) -> int:
    """cpp/sampling.cpp:569-592, in place, INCLUDING its behaviour when every entry passes: the inner loop runs i to j + 1,
    the last entry is then swapped with position n (whatever an earlier stage left there) and n + 1 is returned -- the
    reference samples from that extra entry too, so parity means reproducing it."""
    i, j = 0, n - 1
    while j >= i:
        while P[i] >= threshold and j >= i:
            i += 1
        if P[j] >= threshold:
            P[i], P[j] = P[j], P[i]
            I[i], I[j] = I[j], I[i]
            i += 1
        j -= 1
    return i


def min_p(P: np.ndarray, I: np.ndarray, n: int, mp: float):
    """cpp/sampling.cpp:620-640."""
    top = P[0]
    for i in range(1, n):
        if P[i] > top:
            top = P[i]
    thr = F32(top * F32(mp))
    margin = min(abs(float(P[i]) - float(thr)) for i in range(n + 1))
    return keep_threshold(P, I, n, thr), margin


def multinomial(P: np.ndarray, I: np.ndarray, n: int, random: np.float32):
    """cpp/sampling.cpp:872-915.  Returns (token, its probability, margin of the stopping comparison)."""
    k = 0
    accum = P[0]
    margin = 1.0
    while True:
        margin = min(margin, abs(float(accum) - float(random)))
        if accum >= random:
            break
        if k == n - 1:
            while k > 0 and P[k] == F32(0.0):
                k -= 1
            break
        k += 1
        accum = F32(accum + P[k])
    return int(I[k]), P[k], margin


def next_random(random: np.float32) -> np.float32:
    """ext_sampling.cpp:286-296: r += 1.337 + random (in double, stored as float); r *= r; r = fmod(r, 1)."""
    r = F32(random)
    for _ in range(10):
        r = F32(float(r) + (1.337 + float(random)))
        r = F32(r * r)
        r = F32(np.fmod(r, F32(1.0)))
    return r


def sample_basic(logits: np.ndarray, temperature: float, k: int, tp: float, mp: float, random: float,
                 logit_filter: np.ndarray | None = None):
    """ext_sampling.cpp:137-296 for temperature / top-k / top-p / min-p.  logits [bsz, vocab] (any float dtype, taken to
    fp32 like the reference's `.float()` logits).  Returns (tokens int32 [bsz], probs fp32 [bsz], margins [bsz],
    candidates per row)."""
    bsz, vocab = logits.shape
    temperature = F32(temperature)
    if temperature < F32(0.01):
        temperature, k = F32(1.0), 1
    random = F32(random)
    toks, prs, margins, ncs = [], [], [], []
    for b in range(bsz):
        flt = None if logit_filter is None else logit_filter[b].astype(bool)
        P, maxi = softmax(logits[b], temperature, flt)
        I = np.arange(vocab, dtype=np.int64)
        n = vocab
        margin = 1.0
        assert 0 < k < vocab, "only the top-k regimes 1 <= k < vocab are restated (what the device sampler covers)"
        if k > 1:
            # the k-th / (k+1)-th probabilities: a near-tie between DIFFERENT logits may order differently elsewhere
            srt = np.sort(P)[::-1]
            if srt[k] > 0 and srt[k - 1] != srt[k]:
                margin = min(margin, float(srt[k - 1] - srt[k]) / float(srt[k - 1]) * 1e2)
        n = top_k(P, I, n, k, maxi)
        normalize(P, n)
        if n > 1 and 0.0 < tp < 1.0:
            n, m = top_p(P, I, n, tp)
            margin = min(margin, m)
            normalize(P, n)
        if n > 1 and 0.0 < mp < 1.0:
            n, m = min_p(P, I, n, mp)
            margin = min(margin, m)
            normalize(P, n)
        radj = F32(float(random) * 0.9998)
        tok, pr, m = multinomial(P, I, n, radj)
        margin = min(margin, m)
        toks.append(tok); prs.append(pr); margins.append(margin); ncs.append(n)
        if bsz > 1:
            random = next_random(random)
    return (np.asarray(toks, dtype=np.int32), np.asarray(prs, dtype=F32), np.asarray(margins, dtype=np.float64),
            np.asarray(ncs, dtype=np.int32))
