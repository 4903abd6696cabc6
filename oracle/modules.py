"""Oracle: RMSNorm, RoPE, SiLU*mul, attention (contiguous + paged), Q4 KV-cache codec, MoE routing (numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference files restated here (under /root/reference/exllamav2/):
  * RMSNorm ........ exllamav2_ext/cuda/rms_norm.cu:33-175 (clamp, fp32 sum, rsqrtf, (x*w)*r in fp32, RN to fp16);
                     torch twin rmsnorm.py:187-211
  * RoPE ........... tables device.py:118-169 ; rotation exllamav2_ext/cuda/rope.cu:10-68 (NeoX), :70-123 (GPT-J)
  * act * mul ...... mlp.py:486-494 (F.silu(gate) * up, clamp) ; kernel twin q_mlp_activation.cuh:1-21,54-112
  * attention ...... attn.py:869-937 (_attn_torch: SDPA lower-right causal / matmul + softmax) ; paged contract
                     attn.py:602-613 (flash_attn_with_kvcache: append at cache_seqlens through block_table, causal
                     bottom-right aligned) -- third-party arithmetic, parity unpinned
  * Q4 KV cache .... exllamav2_ext/cuda/cache_q.cuh:4-185 ; addressing cache.cu:143-223, 324-401
  * MoE routing .... moe_mlp.py:276-314 ; q_mlp_softmax.cuh:156-189
"""
from __future__ import annotations

import numpy as np

F16 = np.float16
F32 = np.float32
F64 = np.float64


def _h(x):
    return np.asarray(x, dtype=F16)


def hadd(a, b):
    return (_h(a).astype(F32) + _h(b).astype(F32)).astype(F16)


def hmul(a, b):
    return (_h(a).astype(F32) * _h(b).astype(F32)).astype(F16)


def hfma(a, b, c):
    """fp16 fused multiply-add, single rounding (product exact in float64)."""
    return (_h(a).astype(F64) * _h(b).astype(F64) + _h(c).astype(F64)).astype(F16)


def hdiv(a, b):
    with np.errstate(divide="ignore", invalid="ignore"):
        return (_h(a).astype(F32) / _h(b).astype(F32)).astype(F16)


# ----------------------------------------------------------------------------------------------------------------------
# RMSNorm  (rms_norm.cu:68-76,118,137-142)
# ----------------------------------------------------------------------------------------------------------------------

def rms_norm(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    """x fp16 [rows, dim], w fp16 [dim] -> fp16 [rows, dim]."""
    xf = np.clip(x.astype(F32), -65504.0, 65504.0)
    ss = (xf.astype(F64) ** 2).sum(axis=-1, keepdims=True)
    rmf = (1.0 / np.sqrt(ss / x.shape[-1] + eps)).astype(F32)
    return ((xf * w.astype(F32)[None, :]) * rmf).astype(F16)


# ----------------------------------------------------------------------------------------------------------------------
# RoPE
# ----------------------------------------------------------------------------------------------------------------------

def rope_tables(max_seq_len: int, head_dim: int, theta: float = 10000.0, neox: bool = True):
    """device.py:118-169: sin/cos fp16 [max_seq_len, head_dim] (halves duplicated NeoX / interleaved GPT-J)."""
    inv_freq = (1.0 / (F32(theta) ** (np.arange(0, head_dim, 2, dtype=F32) / F32(head_dim)))).astype(F32)
    t = np.arange(max_seq_len, dtype=F32)
    freqs = np.einsum("i,j->ij", t, inv_freq).astype(F32)
    emb = np.concatenate([freqs, freqs], axis=-1) if neox else np.repeat(freqs, 2, axis=-1)
    return np.sin(emb).astype(F16), np.cos(emb).astype(F16)


def rope_(x: np.ndarray, sin: np.ndarray, cos: np.ndarray, positions: np.ndarray, neox: bool = True) -> np.ndarray:
    """x fp16 [b, s, heads, hd]; positions int [b] = position of token 0 of each sequence (past_len + past_lens[b]).
    NeoX (rope.cu:49-67): l' = fma(l, cos, r * (-sin)), r' = fma(r, cos, l * sin), all fp16.
    GPT-J (rope.cu:108-122): r = x01 * cos01 ; r = fma(x10, (-sin0, sin1), r)."""
    b, s, nh, hd = x.shape
    out = x.copy()
    half = hd // 2
    for bi in range(b):
        for si in range(s):
            p = max(int(positions[bi]) + si, 0)
            if neox:
                c = cos[p, :half][None, :]
                sn = sin[p, :half][None, :]
                l = x[bi, si, :, :half]
                r = x[bi, si, :, half:]
                ls = hmul(r, (-sn).astype(F16))
                rs = hmul(l, sn)
                out[bi, si, :, :half] = hfma(l, c, ls)
                out[bi, si, :, half:] = hfma(r, c, rs)
            else:
                c = cos[p][None, :]
                sn = sin[p].copy()
                sn[0::2] = -sn[0::2]
                x01 = x[bi, si]
                x10 = x01.reshape(nh, half, 2)[:, :, ::-1].reshape(nh, hd)
                out[bi, si] = hfma(x10, sn[None, :], hmul(x01, c))
    return out


# ----------------------------------------------------------------------------------------------------------------------
# SiLU(gate) * up   (mlp.py:486-494: F.silu on fp16 = fp32 math rounded to fp16, then fp16 multiply, clamp)
# ----------------------------------------------------------------------------------------------------------------------

def silu_mul(gate: np.ndarray, up: np.ndarray) -> np.ndarray:
    g = gate.astype(F32)
    y = (g / (1.0 + np.exp(-g.astype(F64)))).astype(F16)
    y = hmul(y, up)
    return np.clip(y, F16(-65504.0), F16(65504.0))


# ----------------------------------------------------------------------------------------------------------------------
# attention (attn.py:905-933 semantics, fp32 softmax like SDPA / flash-attn)
# ----------------------------------------------------------------------------------------------------------------------

def attention(q: np.ndarray, k: np.ndarray, v: np.ndarray, scale: float | None = None, window_left: int = -1,
              softcap: float = 0.0) -> np.ndarray:
    """q [b, sq, H, hd], k/v [b, sk, KVH, hd] fp16, causal bottom-right aligned, GQA by head repetition.
    softcap (attn.py:919-920 -> ext_c.softcap_, applied to the SCALED scores before the mask; flash-attn's `softcap` keyword,
    attn.py:596-600): s = softcap * tanh(s / softcap).  window_left (flash-attn's window_size[0], attn.py:590-594): the query at
    absolute position p sees keys [p - window_left, p] -- for a decode step that is the reference's own torch route up to its
    off-by-one (attn.py:924-926 keeps the last `sliding_window` keys = window_left + 1 of them with window_left = W - 1; the
    flash-attn call it prefers passes (W, W))."""
    b, sq, nh, hd = q.shape
    sk, kvh = k.shape[1], k.shape[2]
    g = nh // kvh
    scale = (hd ** -0.5) if scale is None else scale
    out = np.zeros((b, sq, nh, hd), dtype=F16)
    for bi in range(b):
        for hi in range(nh):
            kh = hi // g
            s = (q[bi, :, hi, :].astype(F64) @ k[bi, :, kh, :].astype(F64).T) * scale      # [sq, sk]
            if softcap:
                s = softcap * np.tanh(s / softcap)
            qi = np.arange(sq)[:, None]
            ki = np.arange(sk)[None, :]
            s = np.where(ki <= qi + (sk - sq), s, -np.inf)
            if window_left >= 0:
                s = np.where(ki >= qi + (sk - sq) - window_left, s, -np.inf)
            s = s - s.max(axis=-1, keepdims=True)
            p = np.exp(s)
            p = p / p.sum(axis=-1, keepdims=True)
            out[bi, :, hi, :] = (p @ v[bi, :, kh, :].astype(F64)).astype(F16)
    return out


def paged_attention(q, k_new, v_new, k_cache, v_cache, cache_seqlens, block_table, scale=None, page_size=256):
    """flash_attn_with_kvcache contract.  k_cache/v_cache [pages, page_size, KVH, hd] are UPDATED IN
    PLACE when k_new/v_new [b, s, KVH, hd] are given (append at cache_seqlens); returns fp16 [b, s, H, hd]."""
    b, s, nh, hd = q.shape
    out = np.zeros_like(q)
    for bi in range(b):
        past = int(cache_seqlens[bi])
        if k_new is not None:
            for si in range(s):
                pos = past + si
                pg = int(block_table[bi, pos // page_size])
                k_cache[pg, pos % page_size] = k_new[bi, si]
                v_cache[pg, pos % page_size] = v_new[bi, si]
            total = past + s
        else:
            total = past                      # caller already wrote k/v and passes cache_seqlens + q_len
        pos = np.arange(total)
        pg = block_table[bi, pos // page_size].astype(np.int64)
        kk = k_cache[pg, pos % page_size][None]
        vv = v_cache[pg, pos % page_size][None]
        out[bi:bi + 1] = attention(q[bi:bi + 1], kk, vv, scale)
    return out


# ----------------------------------------------------------------------------------------------------------------------
# Q4 KV cache  (cache_q.cuh): block = 512 consecutive fp16 elements; thread t holds elements (2t, 2t+1)
# ----------------------------------------------------------------------------------------------------------------------

def _wht32_pairs(w: np.ndarray) -> np.ndarray:
    """w fp16 [..., 32 lanes, 2]: butterfly over the lane axis exactly as cache_q.cuh:26-33 (fp16 adds)."""
    w = w.copy()
    lanes = np.arange(32)
    for i in (1, 2, 4, 8, 16):
        p = w[..., lanes ^ i, :]
        neg = ((lanes & i) != 0)[:, None]
        w = hadd(np.where(neg, -w, w).astype(F16), p)
    return w


def q4_pack(x: np.ndarray):
    """x fp16 flat [n], n % 512 == 0 -> (uint8 [n/2], fp16 scales [n/32])   (cache_q.cuh:4-77, wbits = 4)."""
    n = x.shape[0]
    assert n % 512 == 0
    w = _wht32_pairs(x.astype(F16).reshape(n // 64, 32, 2))                       # [warps, 32 lanes, 2]
    a = np.abs(w).max(axis=-1)                                                    # per lane
    absmax = a.reshape(-1, 2, 16).max(axis=-1)                                    # per 16-lane group = 32 elements
    am = np.repeat(absmax, 16, axis=-1).reshape(-1, 32, 1).astype(F16)
    wn = hfma(hdiv(w, am), F16(8.0), F16(8.0))
    with np.errstate(invalid="ignore"):
        q = np.where(np.isnan(wn), 0, np.rint(wn.astype(F32)))                    # __half2int_rn(NaN) == 0
    q = np.clip(q, 0, 15).astype(np.uint8)
    packed = (q[..., 0] | (q[..., 1] << 4)).reshape(-1)
    scales = hmul(absmax.reshape(-1), F16(1.0 / 8.0))
    return packed, scales


def q4_unpack(packed: np.ndarray, scales: np.ndarray) -> np.ndarray:
    """(cache_q.cuh:109-185): (code - 8) * scale, same butterfly, * 1/32."""
    n = packed.shape[0] * 2
    q0 = (packed & 0xF).astype(np.int32) - 8
    q1 = (packed >> 4).astype(np.int32) - 8
    w = np.stack([q0, q1], axis=-1).astype(F16).reshape(n // 64, 32, 2)
    sc = np.repeat(scales.astype(F16).reshape(-1, 2), 16, axis=-1).reshape(-1, 32, 1)
    w = hmul(w, sc)
    w = _wht32_pairs(w)
    return hmul(w, F16(1.0 / 32.0)).reshape(-1)


def q8_pack(x: np.ndarray):
    """x fp16 flat [n], n % 512 == 0 -> (uint8 [n], fp16 scales [n/32])   (cache_q.cuh:78-107, wbits = 8: the K side of
    the Q6 cache and both sides of the Q8 cache, cache.cu:259-276)."""
    n = x.shape[0]
    assert n % 512 == 0
    w = _wht32_pairs(x.astype(F16).reshape(n // 64, 32, 2))
    a = np.abs(w).max(axis=-1)
    absmax = a.reshape(-1, 2, 16).max(axis=-1)
    am = np.repeat(absmax, 16, axis=-1).reshape(-1, 32, 1).astype(F16)
    wn = hfma(hdiv(w, am), F16(128.0), F16(128.0))
    with np.errstate(invalid="ignore"):
        q = np.where(np.isnan(wn), 0, np.rint(wn.astype(F32)))
    q = np.clip(q, 0, 255).astype(np.uint8)
    return q.reshape(-1), hmul(absmax.reshape(-1), F16(1.0 / 128.0))


def q8_unpack(codes: np.ndarray, scales: np.ndarray) -> np.ndarray:
    """(cache_q.cuh:147-185, wbits = 8): (code - 128) * scale, butterfly, * 1/32."""
    n = codes.shape[0]
    w = (codes.astype(np.int32) - 128).astype(F16).reshape(n // 64, 32, 2)
    sc = np.repeat(scales.astype(F16).reshape(-1, 2), 16, axis=-1).reshape(-1, 32, 1)
    w = _wht32_pairs(hmul(w, sc))
    return hmul(w, F16(1.0 / 32.0)).reshape(-1)


# ----------------------------------------------------------------------------------------------------------------------
# MoE routing (moe_mlp.py:276-314 ; q_mlp_softmax.cuh:156-189): softmax fp32, keep top-k, renormalise
# ----------------------------------------------------------------------------------------------------------------------

def moe_route(logits: np.ndarray, top_k: int):
    """logits fp16 [rows, E] -> (weights fp16 [rows, E] with zeros outside the top-k, selected bool mask)."""
    x = logits.astype(F32)
    x = x - x.max(axis=-1, keepdims=True)
    p = np.exp(x)
    p = p / p.sum(axis=-1, keepdims=True)
    order = np.argsort(-p, axis=-1, kind="stable")
    mask = np.zeros_like(p, dtype=bool)
    np.put_along_axis(mask, order[:, :top_k], True, axis=-1)
    pk = np.where(mask, p, 0.0)
    pk = pk / pk.sum(axis=-1, keepdims=True)
    return pk.astype(F16), mask


# ---- cache utilities of the dynamic generator's data path -------------------------------------------------------------

def fp16_to_fp8(x: np.ndarray) -> np.ndarray:
    """FP8 cache store (cuda/cache.cu:20-37 `compress`): the upper byte of each fp16, i.e. E5M2 by truncation."""
    return (np.ascontiguousarray(x, dtype=np.float16).view(np.uint16) >> 8).astype(np.uint8)


def fp8_to_fp16(c: np.ndarray) -> np.ndarray:
    """cuda/cache.cu:39-44 `decompress`: the byte becomes the upper byte of an fp16, low byte zero."""
    return (c.astype(np.uint16) << 8).view(np.float16)


def fp8_range(token_size: int, offset: int, width: int):
    """Element range [lo, hi) of a cache row that array_fp16_to_fp8_cuda touches (cache.cu:86-100): the token range in
    elements, start rounded down and length rounded up to 8."""
    lo = offset * token_size // 8 * 8
    hi = lo + ((offset + width) * token_size - lo + 7) // 8 * 8
    return lo, hi


def cache_rotate(pages: np.ndarray, order) -> np.ndarray:
    """cuda/cache.cu:534-546: temp <- page[order[0]]; for a, b in pairwise(order): page[a] <- page[b];
    page[order[-1]] <- temp.  Returns a new array."""
    out = pages.copy()
    order = [int(o) for o in order]
    temp = pages[order[0]].copy()
    for a, b in zip(order[:-1], order[1:]):
        out[a] = pages[b]
    out[order[-1]] = temp
    return out


def count_match(a: np.ndarray, b: np.ndarray, max_a: int) -> int:
    """ext_cache.cpp:285-302 / generator/dynamic.py:1504-1509: matching elements from the left between two (1, n) rows."""
    m = min(max_a, b.shape[-1])
    i = 0
    while i < m and a[0, i] == b[0, i]:
        i += 1
    return i
