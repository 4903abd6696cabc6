"""Oracle: EXL2 / GPTQ weight formats, reconstruct and q_gemm semantics (numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Weight / scale / GPTQ decode and exl2_reconstruct: pinned by
execution of the reference's qdq_*.cuh and q_matrix.cu kernels (tests/test_oracle_ref.py); the multiply: parity unpinned
by execution, pinned by relation to reconstruct.

Reference files restated here (all under /root/reference/exllamav2/):
  * on-disk packing ............ exllamav2_ext/cuda/pack_tensor.cu:10-35 (pack_rows_4), :118-248 (pack_columns)
                                 conversion/adaptivegptq.py:608-677 (tensor set, q_groups layout)
  * plain (un-shuffled) decode . exllamav2_ext/cuda/quant/qdq_{2,3,4,5,6,8}.cuh `#else` branches
  * scale decode ............... exllamav2_ext/cuda/quant/qdq_util.cuh:24-30 (dq_scale), ext.py:336 (/256)
  * group map .................. ext.py:301-316 (make_group_map_py)
  * reconstruct ................ exllamav2_ext/cuda/q_matrix.cu:328-497 (EXL2), :204-323 (GPTQ)
  * GPTQ act-order sort ........ exllamav2_ext/cuda/q_matrix.cu:597-643 (make_sequential)
  * gemm semantics ............. linear.py:370-379 (matmul(x, reconstruct) + bias)
"""
from __future__ import annotations

import numpy as np

BITS_ALLOWED = (2, 3, 4, 5, 6, 8)


# ----------------------------------------------------------------------------------------------------------------------
# bit packing along K  (pack_tensor.cu:118-248): for a run of 32 rows at `bits` bits the codes of one column form one
# LSB-first bitstream of `bits` consecutive int32 words of that column.
# ----------------------------------------------------------------------------------------------------------------------

def pack_columns(codes: np.ndarray, bits: int) -> np.ndarray:
    """codes: uint [rows, N] with rows % 32 == 0 and values < 2**bits  ->  int32 [rows*bits/32, N]."""
    assert bits in BITS_ALLOWED
    rows, n = codes.shape
    assert rows % 32 == 0
    chunks = rows // 32
    c = codes.astype(np.uint64).reshape(chunks, 32, n)
    out = np.zeros((chunks, bits, n), dtype=np.uint64)
    for i in range(32):
        pos = i * bits
        w0, sh = divmod(pos, 32)
        v = c[:, i, :] << np.uint64(sh)
        out[:, w0, :] |= v & np.uint64(0xFFFFFFFF)
        if sh + bits > 32:
            out[:, w0 + 1, :] |= v >> np.uint64(32)
    return out.reshape(chunks * bits, n).astype(np.uint32).view(np.int32)


def unpack_columns(words: np.ndarray, bits: int) -> np.ndarray:
    """Inverse of pack_columns (the plain decoders of qdq_*.cuh): int32 [R, N] -> uint16 [R*32/bits, N]."""
    assert bits in BITS_ALLOWED
    r, n = words.shape
    assert r % bits == 0
    chunks = r // bits
    w = words.view(np.uint32).astype(np.uint64).reshape(chunks, bits, n)
    out = np.zeros((chunks, 32, n), dtype=np.uint16)
    mask = np.uint64((1 << bits) - 1)
    for i in range(32):
        pos = i * bits
        w0, sh = divmod(pos, 32)
        v = w[:, w0, :] >> np.uint64(sh)
        if sh + bits > 32:
            v = v | (w[:, w0 + 1, :] << np.uint64(32 - sh))
        out[:, i, :] = (v & mask).astype(np.uint16)
    return out.reshape(chunks * 32, n)


def pack_rows_4(scale_codes: np.ndarray) -> np.ndarray:
    """pack_tensor.cu:10-35: 8 four-bit values per int32 along the row, storing (value - 1)."""
    g, n = scale_codes.shape
    assert n % 8 == 0
    x = (scale_codes.astype(np.int64) - 1).astype(np.uint64) & np.uint64(0xF)
    x = x.reshape(g, n // 8, 8)
    out = np.zeros((g, n // 8), dtype=np.uint64)
    for i in range(8):
        out |= x[:, :, i] << np.uint64(4 * i)
    return out.astype(np.uint32).view(np.int32)


def unpack_rows_4(packed: np.ndarray) -> np.ndarray:
    """-> raw nibbles uint8 [G, N] (the kernel adds 1: qdq_util.cuh:24-30)."""
    g, n8 = packed.shape
    w = packed.view(np.uint32)
    out = np.zeros((g, n8, 8), dtype=np.uint8)
    for i in range(8):
        out[:, :, i] = ((w >> np.uint32(4 * i)) & np.uint32(0xF)).astype(np.uint8)
    return out.reshape(g, n8 * 8)


# ----------------------------------------------------------------------------------------------------------------------
# group bookkeeping
# ----------------------------------------------------------------------------------------------------------------------

def group_table(q_groups: np.ndarray, num_qrows: int):
    """Per group: (bits, first_packed_row, rows).  q_groups = int16 pairs (bits, first packed row),
    adaptivegptq.py:645-647; rows of the last group from the tensor height (q_matrix.cu:130-159)."""
    gr = [int(x) & 0xFFFF for x in q_groups.tolist()]
    g = len(gr) // 2
    out = []
    for i in range(g):
        bits = gr[2 * i]
        q0 = gr[2 * i + 1]
        q1 = gr[2 * i + 3] if i < g - 1 else num_qrows
        qrows = q1 - q0
        out.append((bits, q0, qrows * 32 // bits))
    return out


def make_group_map(q_groups: np.ndarray, num_qrows: int) -> np.ndarray:
    """ext.py:301-316 / ext_qmatrix.cpp:341-361: int16 [2K] pairs (group index, rows remaining in group)."""
    gm = []
    for i, (_bits, _q0, rows) in enumerate(group_table(q_groups, num_qrows)):
        for j in range(rows):
            gm += [i, rows - j]
    return np.asarray(gm, dtype=np.int16)


# ----------------------------------------------------------------------------------------------------------------------
# fp16 helpers (numpy computes half ops in float32 and rounds once; for + * / that is correctly rounded fp16)
# ----------------------------------------------------------------------------------------------------------------------

def h(x):
    return np.asarray(x, dtype=np.float16)


def hmul(a, b):
    return (h(a).astype(np.float32) * h(b).astype(np.float32)).astype(np.float16)


# ----------------------------------------------------------------------------------------------------------------------
# EXL2 reconstruct
# ----------------------------------------------------------------------------------------------------------------------

def exl2_prescale_scale_max(q_scale_max_disk: np.ndarray, prescale: float = 1.0) -> np.ndarray:
    """ext.py:336  `w["q_scale_max"] *= prescale / 256` (in-place fp16 multiply by a python float)."""
    return (q_scale_max_disk.astype(np.float16).astype(np.float32) * np.float32(prescale / 256)).astype(np.float16)


def exl2_scales(q_scale: np.ndarray, q_scale_max_pre: np.ndarray) -> np.ndarray:
    """fp16 [G, N]:  half((s+1)^2) * max'   (qdq_util.cuh:24-30)."""
    s = unpack_rows_4(q_scale).astype(np.int32) + 1
    return hmul((s * s).astype(np.float16), q_scale_max_pre.astype(np.float16)[:, None])


def exl2_reconstruct(t: dict, prescale: float = 1.0) -> np.ndarray:
    """Full dequant to fp16 [K, N] in ORIGINAL row order (q_matrix.cu:328-497).

    t holds the on-disk tensors: q_weight int32 [R,N], q_scale int32 [G,N/8], q_scale_max fp16 [G] (as stored,
    before the /256), q_groups int16 [2G], q_invperm int32 [K] (optional)."""
    qw = t["q_weight"]
    r, n = qw.shape
    groups = group_table(t["q_groups"], r)
    scales = exl2_scales(t["q_scale"], exl2_prescale_scale_max(t["q_scale_max"], prescale))
    k = sum(g[2] for g in groups)
    w = np.zeros((k, n), dtype=np.float16)
    row = 0
    for gi, (bits, q0, rows) in enumerate(groups):
        qrows = rows * bits // 32
        codes = unpack_columns(qw[q0:q0 + qrows], bits).astype(np.int32)
        ints = (codes - (1 << (bits - 1))).astype(np.float16)            # exact
        w[row:row + rows] = hmul(ints, scales[gi][None, :])
        row += rows
    if "q_invperm" in t and t["q_invperm"] is not None:
        perm = np.argsort(np.asarray(t["q_invperm"]).astype(np.int64), kind="stable")   # module.py:120
        out = np.zeros_like(w)
        out[perm] = w                                                     # packed row i -> original row perm[i]
        return out
    return w


# ----------------------------------------------------------------------------------------------------------------------
# GPTQ reconstruct
# ----------------------------------------------------------------------------------------------------------------------

def gptq_sequential_perm(g_idx: np.ndarray, groups: int):
    """q_matrix.cu:606-642: stable counting sort of rows by group -> (x_map = q_perm, x_map_inv = q_invperm)."""
    g_idx = np.asarray(g_idx).astype(np.int64)
    k = g_idx.shape[0]
    hist = np.bincount(g_idx, minlength=groups)
    start = np.concatenate([[0], np.cumsum(hist)[:-1]])
    nxt = start.copy()
    x_map_inv = np.zeros(k, dtype=np.int64)
    for row in range(k):
        grp = g_idx[row]
        x_map_inv[row] = nxt[grp]
        nxt[grp] += 1
    x_map = np.zeros(k, dtype=np.int64)
    x_map[x_map_inv] = np.arange(k)
    return x_map, x_map_inv


def gptq_reconstruct(t: dict) -> np.ndarray:
    """GPTQ 4-bit: w[k,n] = half(q - (z+1)) * scale[g(k), n]  (q_matrix.cu:204-323, :261-318).

    t: qweight int32 [K/8, N], qzeros int32 [G, N/8], scales fp16 [G, N], g_idx int32 [K] or None."""
    qweight = t["qweight"]
    k8, n = qweight.shape
    k = k8 * 8
    g = t["qzeros"].shape[0]
    codes = unpack_columns(qweight, 4).astype(np.int32)                  # [K, N], row = original row
    zeros = unpack_rows_4(t["qzeros"]).astype(np.int32) + 1              # [G, N]
    scales = np.asarray(t["scales"]).astype(np.float16)
    g_idx = t.get("g_idx")
    if g_idx is None or not np.any(np.asarray(g_idx) != 0):
        gs = 1
        while gs * g < k:
            gs *= 2                                                       # q_matrix.cu:101-105
        grp = np.arange(k) // gs
    else:
        grp = np.asarray(g_idx).astype(np.int64)
    ints = (codes - zeros[grp]).astype(np.float16)
    return hmul(ints, scales[grp])


# ----------------------------------------------------------------------------------------------------------------------
# q_gemm semantics  (linear.py:370-379): matmul(x, reconstruct) (+ bias); BLAS accumulates in fp32.  The oracle
# accumulates in float64 and rounds once to fp16: tests state the fp16 tolerance explicitly.
# ----------------------------------------------------------------------------------------------------------------------

def gemm_ref(a: np.ndarray, w: np.ndarray, bias: np.ndarray | None = None, c_in: np.ndarray | None = None,
             exact: bool = False):
    acc = a.astype(np.float64) @ w.astype(np.float64)
    if bias is not None:
        acc = acc + bias.astype(np.float64)[None, :]
    if c_in is not None:
        acc = acc + c_in.astype(np.float64)
    return acc if exact else acc.astype(np.float16)


# ----------------------------------------------------------------------------------------------------------------------
# synthetic tensors in the on-disk format
# ----------------------------------------------------------------------------------------------------------------------

def synth_exl2(k: int, n: int, spec, seed: int = 0, act_order: bool = True, sigma: float = 0.02,
               bias: bool = False) -> dict:
    """spec: list of (bits, group_size, rows) with sum(rows) == k, rows % 32 == 0 (last group of a run may be short but
    still a multiple of 32).  Bits are expected in descending order like the quantizer emits (qparams.py)."""
    rng = np.random.default_rng(seed)
    assert n % 32 == 0 and sum(s[2] for s in spec) == k
    qw, groups, smax = [], [], []
    out_row = 0
    for bits, gs, rows in spec:
        assert rows % 32 == 0 and gs % 32 == 0 and bits in BITS_ALLOWED
        done = 0
        while done < rows:
            r = min(gs, rows - done)
            codes = rng.integers(0, 1 << bits, size=(r, n), dtype=np.int64).astype(np.uint16)
            qw.append(pack_columns(codes, bits))
            groups += [bits, out_row]
            out_row += r * bits // 32
            # dequantized weights ~ sigma: (q - 2^(b-1)) has std ~ 2^b/sqrt(12); scale_g[n] = (s+1)^2 * max/256
            smax.append(rng.uniform(0.5, 1.5) * sigma * np.sqrt(12.0) / (1 << bits) / 93.5 * 256.0)
            done += r
    g = len(smax)
    t = {
        "q_weight": np.concatenate(qw, axis=0),
        "q_groups": np.asarray(groups, dtype=np.int16),
        "q_scale_max": np.asarray(smax, dtype=np.float16),
        "q_scale": pack_rows_4(rng.integers(1, 17, size=(g, n))),
    }
    assert out_row < 65536
    if act_order:
        t["q_invperm"] = rng.permutation(k).astype(np.int32)
    else:
        t["q_invperm"] = np.arange(k, dtype=np.int32)
    if bias:
        t["bias"] = (rng.standard_normal(n) * 0.1).astype(np.float16)
    return t


def synth_gptq(k: int, n: int, group_size: int = 128, seed: int = 0, act_order: bool = False) -> dict:
    rng = np.random.default_rng(seed)
    assert k % 8 == 0 and n % 8 == 0 and k % group_size == 0
    g = k // group_size
    codes = rng.integers(0, 16, size=(k, n)).astype(np.uint16)
    t = {
        "qweight": pack_columns(codes, 4),
        "qzeros": pack_rows_4(rng.integers(1, 17, size=(g, n))),         # stored nibble = zero-1 ; kernel adds 1
        "scales": rng.uniform(0.005, 0.02, size=(g, n)).astype(np.float16),
    }
    if act_order:
        t["g_idx"] = rng.permutation(np.arange(k) // group_size).astype(np.int32)
    else:
        t["g_idx"] = (np.arange(k) // group_size).astype(np.int32)
    return t
