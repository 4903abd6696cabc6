"""`flash_attn_2_cuda` stand-in: the reference imports it next to `flash_attn` when it detects paged-attention support
(attn.py:56) and calls `fwd_kvcache` from its single-process tensor-parallel bindings (ext_qattn.cpp:291, 416-438, 653-675).
The drop-in's own TP entry points (exllamav2_amd/ext_tp.py) go straight to libexl2_hip.so; this function serves the same
positional call for anybody else."""
from flash_attn import flash_attn_with_kvcache as _fa


def fwd_kvcache(q, k_cache, v_cache, k, v, cache_seqlens, rotary_cos, rotary_sin, cache_batch_idx, cache_leftpad,
                block_table, alibi_slopes, out, softmax_scale, causal, window_left, window_right, softcap,
                rotary_interleaved, num_splits):
    """Argument order of flash-attn 2.5.7+'s C++ entry as the reference calls it.  Returns [out, softmax_lse(None)]."""
    # (a window that cannot clip keeps the one-launch route, one that may clip and softcap take the general kernels:
    # dropin/flash_attn/__init__.py)
    window = (-1, -1) if (window_left < 0 and window_right in (-1, 0)) else (window_left, window_right)
    o = _fa(q, k_cache, v_cache, k=k, v=v, rotary_cos=rotary_cos, rotary_sin=rotary_sin, cache_seqlens=cache_seqlens,
            cache_batch_idx=cache_batch_idx, cache_leftpad=cache_leftpad, block_table=block_table,
            softmax_scale=softmax_scale, causal=causal, window_size=window, softcap=softcap or 0.0, alibi_slopes=alibi_slopes)
    if out is not None:
        out.copy_(o)
        o = out
    return [o, None]
