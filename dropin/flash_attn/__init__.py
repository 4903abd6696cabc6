"""`flash_attn` shim: lets the reference's flash-attn modes (attn.py:35-59 detects `import flash_attn` and `__version__ >=
2.5.7`; attn.py:602-613 calls `flash_attn_with_kvcache` in paged mode, attn.py:960-977 `flash_attn_func` otherwise) run on
MI355X without flash-attn, on top of libexl2_hip.so (`exl2_rope_kv_append`, `exl2_paged_attn`, `exl2_flash_prefill`:
csrc/attn.hip, csrc/attn_prefill.hip).  Only the call shapes the reference uses are accepted; anything else raises instead of
silently computing something different.  window_size (left part) and softcap -- the two keyword arguments the reference adds for
Mistral / Gemma-type checkpoints (attn.py:590-600) -- are served by the general kernels (`exl2_paged_attn_ex`, `exl2_flash_prefill_ex`:
a window that clips, a score cap); a window that cannot clip and no cap keep the one-launch decode route."""
import torch

from exllamav2_amd.ext import ext_c as _e

__version__ = "2.5.7"


def flash_attn_with_kvcache(q, k_cache, v_cache, k=None, v=None, rotary_cos=None, rotary_sin=None, cache_seqlens=None,
                            cache_batch_idx=None, cache_leftpad=None, block_table=None, softmax_scale=None, causal=False,
                            window_size=(-1, -1), softcap=0.0, rotary_interleaved=True, alibi_slopes=None, num_splits=0,
                            return_softmax_lse=False):
    """q [b, s, H, hd]; k_cache / v_cache [pages, 256, KVH, hd]; k / v [b, s, KVH, hd] appended at cache_seqlens through
    block_table; causal bottom-right; returns [b, s, H, hd] (flash-attn's contract as the reference relies on it)."""
    if rotary_cos is not None or rotary_sin is not None or cache_batch_idx is not None or cache_leftpad is not None \
            or alibi_slopes is not None or return_softmax_lse:
        raise NotImplementedError("flash_attn shim: only the arguments exllamav2 passes are supported")
    # sliding window (attn.py:590-594 passes (W, W) for Mistral-family checkpoints; causal attention makes the right part moot): a
    # window that cannot clip -- every key a query may see lies within W positions of it, decided from HOST-side sizes (the table's
    # capacity), never by reading cache_seqlens back -- is plain causal attention and keeps the fast route; one that may clip goes to
    # the general kernels with window_left = W
    window_left = -1
    if tuple(window_size) != (-1, -1) and window_size[0] >= 0:
        capacity = (block_table.shape[1] * k_cache.shape[1]) if block_table is not None else k_cache.shape[1]
        if capacity - 1 > window_size[0]:
            if not causal:
                raise NotImplementedError("flash_attn shim: a sliding window without causal attention is not a call the reference makes")
            window_left = int(window_size[0])
    if isinstance(cache_seqlens, int):
        cache_seqlens = torch.full((q.shape[0],), cache_seqlens, dtype=torch.int32, device=q.device)
    # decode-sized steps of the dynamic generator: append + attention + merge in ONE launch through the compiled binding (which also
    # leaves the act-ordered copy q_attn_forward_2's chained launch reads); everything else takes the general route
    fast = _fast()
    if window_left >= 0 or softcap:
        return _e.flash_attn_with_kvcache(q, k_cache, v_cache, k, v, cache_seqlens, block_table, causal, softmax_scale, None,
                                          window_left, float(softcap or 0.0))
    if fast is not None and k is not None and v is not None and block_table is not None and causal and q.shape[1] <= 8:
        out = fast.flash_attn_kvcache_decode(q, k_cache, v_cache, k, v, cache_seqlens, block_table,
                                             float(q.shape[-1] ** -0.5 if softmax_scale is None else softmax_scale))
        if out is not None:
            return out
    return _e.flash_attn_with_kvcache(q, k_cache, v_cache, k, v, cache_seqlens, block_table, causal, softmax_scale)


_scratch = {}


def _fast():
    """the compiled half of the drop-in, when `exllamav2_ext` (dropin/exllamav2_ext.py) has loaded it"""
    import sys
    m = sys.modules.get("exllamav2_ext")
    return getattr(m, "_fast", None) if m is not None else None


def _split_scratch(q):
    """per-device work space of the split-KV decode kernel (csrc/attn.hip), grown on demand"""
    b, s, nh, hd = q.shape
    need = _e.paged_attn_scratch_bytes(b * s * nh, hd, 16) // 4 + 1
    t = _scratch.get(q.device)
    if t is None or t.numel() < need:
        t = torch.zeros((need,), dtype=torch.float32, device=q.device)
        _scratch[q.device] = t
    return t


def _whole_rows(t):
    """k / v as the reference hands them over: `batch_keys[:b, :past + q_len]` (attn.py:1166-1167), a view whose batch stride is
    the cache's max_seq_len -> the [b, T, KVH, hd] tensor the kernels index (same storage)."""
    if t.is_contiguous():
        return t
    b, n, kvh, hd = t.shape
    if t.stride(3) != 1 or t.stride(2) != hd or t.stride(1) != kvh * hd or t.stride(0) % (kvh * hd):
        return t.contiguous()
    return torch.as_strided(t, (b, t.stride(0) // (kvh * hd), kvh, hd), (t.stride(0), kvh * hd, hd, 1))


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                    alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """The reference's NON-paged attention call (attn.py:960-977 `_attn_flash`, chosen by attn.py:1141-1142 whenever flash-attn
    is importable): q [b, s, H, hd] against k / v [b, past + s, KVH, hd] = the live rows of the cache (RoPE already applied by
    q_attn_forward_1), causal bottom-right.  One launch of csrc/attn.hip (decode-shaped) or csrc/attn_prefill.hip (many query
    rows) instead of the ~8 torch launches per layer of the reference's own `_attn_torch` fallback."""
    if dropout_p or alibi_slopes is not None or return_attn_probs:
        raise NotImplementedError("flash_attn shim: only the arguments exllamav2 passes are supported")
    b, s, nh, hd = q.shape
    n = k.shape[1]
    window_left = -1
    if tuple(window_size) != (-1, -1) and window_size[0] >= 0 and n - 1 > window_size[0]:     # (see flash_attn_with_kvcache)
        if not causal:
            raise NotImplementedError("flash_attn shim: a sliding window without causal attention is not a call the reference makes")
        window_left = int(window_size[0])
    if not causal and s > 1:
        raise NotImplementedError("flash_attn shim: the kernels are causal (bottom-right aligned), like every call of the reference")
    if not q.is_contiguous():
        q = q.contiguous()
    kf, vf = _whole_rows(k), _whole_rows(v)
    # a decode step of one sequence (attn.py:1088-1091 "direct": its K / V rows are in the cache already): ONE launch of
    # csrc/attn.hip through the compiled binding, which also leaves the act-ordered copy q_attn_forward_2's chained launch reads
    # (dropin/_exl2_fast.cpp).  Anything else -- or a shape that kernel does not cover -- takes the general route below.
    fast = _fast()
    if fast is not None and b == 1 and s <= 16 and window_left < 0 and not softcap:
        out = fast.flash_attn_decode(q, kf, vf, float(hd ** -0.5 if softmax_scale is None else softmax_scale))
        if out is not None:
            return out
    out = torch.empty_like(q)
    if not (s > 16 and _e.flash_prefill(q, kf, vf, out, None, None, n - s, s, softmax_scale, True, window_left, float(softcap or 0.0))):
        _e.paged_attn(q, kf, vf, out, None, None, n - s, s, softmax_scale, True, 0, _split_scratch(q), window_left, float(softcap or 0.0))
    return out


def flash_attn_varlen_func(*a, **k):
    raise NotImplementedError("flash_attn shim: block-diagonal (varlen) attention is not built; the reference takes this path only "
                              "for attn_params.block_diag_layers (vision towers, SURVEY.md 2: out of scope)")
