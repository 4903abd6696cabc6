"""`flash_attn` shim: lets the reference's paged-attention mode (attn.py:35-59 detects `import flash_attn` and
`__version__ >= 2.5.7`; attn.py:602-613 calls `flash_attn_with_kvcache`) run on MI355X without flash-attn, on top of
libexl2_hip.so (`exl2_rope_kv_append` + `exl2_paged_attn`, csrc/attn.hip).  Only the call shape the reference uses is
accepted; anything else raises instead of silently computing something different."""
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
    if tuple(window_size) != (-1, -1) or softcap:
        raise NotImplementedError("flash_attn shim: sliding window / softcap are not built (SURVEY.md 8a row a15)")
    if isinstance(cache_seqlens, int):
        cache_seqlens = torch.full((q.shape[0],), cache_seqlens, dtype=torch.int32, device=q.device)
    return _e.flash_attn_with_kvcache(q, k_cache, v_cache, k, v, cache_seqlens, block_table, causal, softmax_scale)


def flash_attn_func(*a, **k):
    raise NotImplementedError("flash_attn shim: the reference only needs flash_attn_with_kvcache for paged mode; "
                              "non-paged attention goes through its own _attn_torch / SDPA path")


flash_attn_varlen_func = flash_attn_func
