"""KV caches (reference cache.py): FP16 and Q4, contiguous [batch, max_seq_len, kv_heads, head_dim] per layer, or paged
views [pages, 256, kv_heads, head_dim] of the same storage (attn.py:504-505)."""
from __future__ import annotations

import torch

from .ext import none_tensor

PAGE_SIZE = 256                                           # generator/dynamic.py:33


class ExLlamaV2Cache:
    """FP16 cache (cache.py:230-300)."""

    def __init__(self, model, batch_size: int = 1, max_seq_len: int | None = None):
        cfg = model.config
        self.model = model
        self.batch_size = batch_size
        self.max_seq_len = max_seq_len or cfg.max_seq_len
        self.current_seq_len = 0
        shape = (batch_size, self.max_seq_len, cfg.num_key_value_heads, cfg.head_dim)
        n_layers = len(model.layers) if model.layers else cfg.num_hidden_layers          # layers held by this device
        self.key_states = [torch.zeros(shape, dtype=torch.float16, device=model.device) for _ in range(n_layers)]
        self.value_states = [torch.zeros(shape, dtype=torch.float16, device=model.device) for _ in range(n_layers)]

    def get_kv_state(self, layer_idx: int, batch_size: int, offset: int, width: int, page_size: int = 0,
                     cache_seqlens=None, block_table=None):
        return self.key_states[layer_idx], self.value_states[layer_idx]

    def store_kv_state(self, layer_idx: int, batch_size: int, offset: int, width: int, page_size: int = 0,
                       cache_seqlens=None, block_table=None):
        pass

    def paged_view(self, layer_idx: int):
        cfg = self.model.config
        k = self.key_states[layer_idx].view(-1, PAGE_SIZE, cfg.num_key_value_heads, cfg.head_dim)
        v = self.value_states[layer_idx].view(-1, PAGE_SIZE, cfg.num_key_value_heads, cfg.head_dim)
        return k, v

    def footprint(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.key_states + self.value_states)


class ExLlamaV2Cache_Q4(ExLlamaV2Cache):
    """Q4 cache (cache.py:409-606): uint8 codes + fp16 scales per 32 elements; one shared fp16 temp pair per device.
    get_kv_state dequantizes the live range into the temp, store_kv_state re-quantizes the touched blocks."""

    wbits = 4

    def __init__(self, model, batch_size: int = 1, max_seq_len: int | None = None):
        cfg = model.config
        self.model = model
        self.ext = model.ext
        self.batch_size = batch_size
        self.max_seq_len = max_seq_len or cfg.max_seq_len
        self.current_seq_len = 0
        kvh, hd, dev = cfg.num_key_value_heads, cfg.head_dim, model.device
        qshape = (batch_size, self.max_seq_len, kvh, hd // 2)
        sshape = (batch_size, self.max_seq_len, kvh, hd // 32)
        L = len(model.layers) if model.layers else cfg.num_hidden_layers
        self.key_states = [torch.zeros(qshape, dtype=torch.uint8, device=dev) for _ in range(L)]
        self.value_states = [torch.zeros(qshape, dtype=torch.uint8, device=dev) for _ in range(L)]
        self.key_scales = [torch.zeros(sshape, dtype=torch.float16, device=dev) for _ in range(L)]
        self.value_scales = [torch.zeros(sshape, dtype=torch.float16, device=dev) for _ in range(L)]
        tshape = (batch_size, self.max_seq_len, kvh, hd)
        self.temp_k = torch.zeros(tshape, dtype=torch.float16, device=dev)       # cache.py:464-469
        self.temp_v = torch.zeros(tshape, dtype=torch.float16, device=dev)

    def get_kv_state(self, layer_idx, batch_size, offset, width, page_size=0, cache_seqlens=None, block_table=None):
        """cache.py:472-514"""
        if width > 0 or page_size:
            self.ext.q_to_fp16_kv(self.key_states[layer_idx], self.temp_k, self.key_scales[layer_idx],
                                  self.value_states[layer_idx], self.temp_v, self.value_scales[layer_idx],
                                  batch_size, offset, width, page_size,
                                  cache_seqlens if cache_seqlens is not None else none_tensor,
                                  block_table if block_table is not None else none_tensor, self.wbits)
        return self.temp_k, self.temp_v

    def store_kv_state(self, layer_idx, batch_size, offset, width, page_size=0, cache_seqlens=None, block_table=None):
        """cache.py:517-556"""
        if width > 0:
            self.ext.fp16_to_q_kv(self.temp_k, self.key_states[layer_idx], self.key_scales[layer_idx],
                                  self.temp_v, self.value_states[layer_idx], self.value_scales[layer_idx],
                                  batch_size, offset, width, page_size,
                                  cache_seqlens if cache_seqlens is not None else none_tensor,
                                  block_table if block_table is not None else none_tensor, self.wbits)

    def paged_view(self, layer_idx):
        cfg = self.model.config
        return (self.temp_k.view(-1, PAGE_SIZE, cfg.num_key_value_heads, cfg.head_dim),
                self.temp_v.view(-1, PAGE_SIZE, cfg.num_key_value_heads, cfg.head_dim))

    def q4_views(self, layer_idx, paged: bool):
        """(k_codes, k_scales, v_codes, v_scales) of a layer for the fused Q4 attention kernel (attn_q4.hip): contiguous
        [b, T, KVH, .] or, paged, [pages, 256, KVH, .]."""
        ts = (self.key_states[layer_idx], self.key_scales[layer_idx], self.value_states[layer_idx], self.value_scales[layer_idx])
        if not paged:
            return ts
        kvh = self.model.config.num_key_value_heads
        return tuple(t.view(-1, PAGE_SIZE, kvh, t.shape[-1]) for t in ts)

    def footprint(self) -> int:
        ts = self.key_states + self.value_states + self.key_scales + self.value_scales
        return sum(t.numel() * t.element_size() for t in ts)
