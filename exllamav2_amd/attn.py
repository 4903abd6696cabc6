"""ExLlamaV2Attention (reference attn.py:241-330 load/make_q_attn, :1017-1196 forward, :466-638 forward_paged)."""
from __future__ import annotations

import os

import torch

from .ext import none_tensor
from .linear import ExLlamaV2Linear


class ExLlamaV2Attention:
    def __init__(self, model, key: str, layer_idx: int):
        cfg = model.config
        self.model, self.ext, self.key, self.layer_idx = model, model.ext, key, layer_idx
        h = cfg.hidden_size
        self.q_proj = ExLlamaV2Linear(self.ext, key + ".self_attn.q_proj", h, cfg.num_attention_heads * cfg.head_dim, model)
        self.k_proj = ExLlamaV2Linear(self.ext, key + ".self_attn.k_proj", h, cfg.num_key_value_heads * cfg.head_dim, model)
        self.v_proj = ExLlamaV2Linear(self.ext, key + ".self_attn.v_proj", h, cfg.num_key_value_heads * cfg.head_dim, model)
        self.o_proj = ExLlamaV2Linear(self.ext, key + ".self_attn.o_proj", cfg.num_attention_heads * cfg.head_dim, h, model)
        self.pre_layernorm = None
        self.q_handle = None
        self.fused_decode = os.environ.get("EXL2_ATTN_FUSED", "1") != "0"       # A/B switch for measurements
        self.q4_fused = os.environ.get("EXL2_Q4_FUSED", "1") != "0"
        # Q4 decode steps in two launches (RoPE + pack of the new rows; attention with the split merge inside) instead of four
        # (EXL2_Q4_LAUNCHES=4: the round-4 sequence rope_kv_append -> fp16_to_q_kv -> paged_attn_q4 -> its combine launch) ...
        self.q4_two_launches = os.environ.get("EXL2_Q4_LAUNCHES", "1") != "4"
        # ... or ONE (RoPE + pack + attention + merge: csrc/attn_q4.hip, the FUSED form; head_dim 128) -- EXL2_Q4_LAUNCHES=2 keeps two
        self.q4_one_launch = os.environ.get("EXL2_Q4_LAUNCHES", "1") == "1"

    def load(self, ck: dict):
        cfg, m = self.model.config, self.model
        for lin in (self.q_proj, self.k_proj, self.v_proj, self.o_proj):
            lin.load(ck[lin.key])
        self.pre_layernorm = ck[self.key + ".input_layernorm"]
        self.q_handle = self.ext.make_q_attn(                                   # attn.py:300-330
            self.pre_layernorm, none_tensor, True, False, cfg.norm_eps,
            self.q_proj.q_handle, self.k_proj.q_handle, self.v_proj.q_handle, self.o_proj.q_handle,
            m.temp_state, none_tensor, m.max_rows, cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads,
            cfg.head_dim, cfg.max_seq_len, True, cfg.rope_style, cfg.head_dim, none_tensor, none_tensor, none_tensor,
            none_tensor, False, False)
        return self

    def unload(self):
        if self.q_handle is not None:
            self.ext.free_q_attn(self.q_handle)                                  # attn handle before its linears
            self.q_handle = None
        for lin in (self.q_proj, self.k_proj, self.v_proj, self.o_proj):
            lin.unload()

    def forward(self, hidden_states: torch.Tensor, cache, past_len: int = 0, cache_seqlens=None, block_table=None):
        """In place on hidden_states [b, q_len, hidden] (residual add in o_proj's epilogue).

        Contiguous mode (attn.py:1017-1196): scalar `past_len`, cache viewed [b, max_seq_len, kvh, hd].
        Paged mode (attn.py:466-638): `cache_seqlens` int32 [b] + `block_table` int32 [b, pages] on the device,
        cache viewed [pages, 256, kvh, hd]; positions are read on the device (graph-capturable)."""
        b, q_len, _ = hidden_states.shape
        q, k, v = self._project_qkv(hidden_states, b, q_len)
        attn_out = self._attend(q, k, v, cache, past_len, cache_seqlens, block_table)
        return self._project_out(hidden_states, attn_out, b, q_len)

    def _project_qkv(self, hidden_states, b: int, q_len: int):
        """Front half (q_attn_forward_1, q_attn.cu:153-317): RMSNorm + q/k/v projections into the scratch rows; RoPE is
        applied later, inside the attention launch."""
        cfg, m, ext = self.model.config, self.model, self.ext
        rows = b * q_len
        q = m.temp_q[:rows].view(b, q_len, cfg.num_attention_heads, cfg.head_dim)
        k = m.temp_k[:rows].view(b, q_len, cfg.num_key_value_heads, cfg.head_dim)
        v = m.temp_v[:rows].view(b, q_len, cfg.num_key_value_heads, cfg.head_dim)
        ext.q_attn_forward_1(self.q_handle, hidden_states, b, q_len, 0, none_tensor, q, k, v, m.sin, m.cos,
                             apply_rope=False)
        return q, k, v

    def _attend(self, q, k, v, cache, past_len, cache_seqlens, block_table):
        """RoPE(q, new k) + KV append + attention over the (paged or contiguous, FP16 or Q4) cache -> attn_out rows."""
        cfg, m, ext = self.model.config, self.model, self.ext
        b, q_len = q.shape[0], q.shape[1]
        rows = b * q_len
        paged = block_table is not None
        if cache is None:
            raise RuntimeError("ExLlamaV2Attention.forward: a cache is required")
        attn_out = m.temp_attn[:rows].view(b, q_len, cfg.num_attention_heads, cfg.head_dim)
        # Q4 cache, decode-sized step: attend straight from the codes (attn_q4.hip) -- the reference unpacks the whole live
        # range to an fp16 temp for every layer of every step (cache.py:472-514)
        # (the step's own K/V are attended in fp16 and quantised afterwards, the reference's order)
        q4_direct = (getattr(cache, "wbits", 0) == 4 and hasattr(cache, "q4_views") and self.q4_fused and q_len <= 8
                     and q_len * (cfg.num_attention_heads // cfg.num_key_value_heads) <= 64
                     and cfg.head_dim in (64, 128, 256)
                     and (cfg.num_key_value_heads * cfg.head_dim) % 512 == 0)     # codec blocks (512 elements) must not span tokens
        if paged:
            if not q4_direct:
                cache.get_kv_state(self.layer_idx, b, 0, 0, 256, cache_seqlens, block_table)
            kc, vc = cache.paged_view(self.layer_idx)
            sl, bt, past = cache_seqlens, block_table, 0
        else:
            if q4_direct:
                kc, vc = cache.get_kv_state(self.layer_idx, b, 0, 0)           # width 0: the temp pair, nothing unpacked
            else:
                kc, vc = cache.get_kv_state(self.layer_idx, b, 0, past_len)
            sl, bt, past = None, None, past_len
        if q4_direct:
            # new K/V -> (rotated) into the fp16 staging rows -> quantised into the cache -> attention over codes
            kq, ks, vq, vs = cache.q4_views(self.layer_idx, paged)
            two = getattr(self, "q4_two_launches", False)
            if getattr(self, "q4_one_launch", False) and ext.attn_q4_decode_fused(
                    q, k, v, kq, ks, vq, vs, attn_out, m.sin, m.cos, sl, bt, past, cfg.rope_style, m.attn_scratch, m.attn_counters):
                return attn_out
            if not (two and ext.rope_quant_append_q4(q, k, v, kq, ks, vq, vs, m.sin, m.cos, past if not paged else 0,
                                                     sl if paged else none_tensor, bt if paged else none_tensor, cfg.rope_style)):
                ext.rope_kv_append(q, k, v, kc, vc, m.sin, m.cos, past if not paged else 0,
                                   sl if paged else none_tensor, bt if paged else none_tensor, cfg.rope_style)
                if paged:
                    cache.store_kv_state(self.layer_idx, b, 0, q_len, 256, cache_seqlens, block_table)
                else:
                    cache.store_kv_state(self.layer_idx, b, past_len, q_len)
            ok = ext.paged_attn_q4(q, kq, ks, vq, vs, attn_out, sl, bt, len_const=past, len_offset=q_len,
                                   scratch=m.attn_scratch, k_new=k, v_new=v, counters=m.attn_counters if two else None)
            if not ok:
                raise RuntimeError("ExLlamaV2Attention: fused Q4 attention rejected a shape it was selected for")
        else:
            # decode-shaped steps: one launch does RoPE + append + attention + split merge; otherwise three launches
            fused = self.fused_decode and ext.attn_decode_fused(
                q, k, v, kc, vc, attn_out, m.sin, m.cos, sl, bt, past, cfg.rope_style, m.attn_scratch, m.attn_counters)
            if not fused:
                ext.rope_kv_append(q, k, v, kc, vc, m.sin, m.cos, past if not paged else 0,
                                   sl if paged else none_tensor, bt if paged else none_tensor, cfg.rope_style)
                # many query rows: MFMA flash attention (csrc/attn_prefill.hip); decode-shaped: the split-KV kernel
                if not (q_len > 16 and ext.flash_prefill(q, kc, vc, attn_out, sl, bt, len_const=past, len_offset=q_len)):
                    ext.paged_attn(q, kc, vc, attn_out, sl, bt, len_const=past, len_offset=q_len, scratch=m.attn_scratch)
            if paged:
                cache.store_kv_state(self.layer_idx, b, 0, q_len, 256, cache_seqlens, block_table)
            else:
                cache.store_kv_state(self.layer_idx, b, past_len, q_len)
        return attn_out

    def q4_chain_capable(self, cache) -> bool:
        """the chained step can attend straight from this Q4 cache's codes (the conditions of _attend's q4_direct at q_len 1)"""
        cfg = self.model.config
        return (getattr(cache, "wbits", 0) == 4 and hasattr(cache, "q4_views") and self.q4_fused
                and cfg.num_attention_heads // cfg.num_key_value_heads <= 64 and cfg.head_dim in (64, 128, 256)
                and (cfg.num_key_value_heads * cfg.head_dim) % 512 == 0)

    def attend_chain(self, q, k, v, cache, cache_seqlens, block_table, out_invperm):
        """Decode attention of the chained step (model.GreedyGraphDecoder), output written in o_proj's packed order.  Paged
        FP16 cache: the one-launch kernel.  Paged Q4 cache: RoPE + Q4 pack of the new rows (csrc/cache_q.hip: rope_quant_q4_kernel),
        attention straight from the codes with the split merge inside (csrc/attn_q4.hip) -- the two launches of _attend's q4_direct route.  Raises when
        the shape needs another path (the decoder then un-chains)."""
        cfg, m, ext = self.model.config, self.model, self.ext
        b, q_len = q.shape[0], q.shape[1]
        if getattr(cache, "wbits", 0):
            if not self.q4_chain_capable(cache):
                raise RuntimeError("attend_chain: shape not covered by the Q4 attention kernel")
            attn_out = m.temp_attn[:b * q_len].view(b, q_len, cfg.num_attention_heads, cfg.head_dim)
            kq, ks, vq, vs = cache.q4_views(self.layer_idx, True)
            two = getattr(self, "q4_two_launches", False)
            if getattr(self, "q4_one_launch", False) and ext.attn_q4_decode_fused(
                    q, k, v, kq, ks, vq, vs, attn_out, m.sin, m.cos, cache_seqlens, block_table, 0, cfg.rope_style, m.attn_scratch,
                    m.attn_counters, out_invperm=out_invperm):
                return attn_out
            if not (two and ext.rope_quant_append_q4(q, k, v, kq, ks, vq, vs, m.sin, m.cos, 0, cache_seqlens, block_table, cfg.rope_style)):
                kc, vc = cache.paged_view(self.layer_idx)
                ext.rope_kv_append(q, k, v, kc, vc, m.sin, m.cos, 0, cache_seqlens, block_table, cfg.rope_style)
                cache.store_kv_state(self.layer_idx, b, 0, q_len, 256, cache_seqlens, block_table)
            if not ext.paged_attn_q4(q, kq, ks, vq, vs, attn_out, cache_seqlens, block_table, len_const=0, len_offset=q_len,
                                     scratch=m.attn_scratch, k_new=k, v_new=v, out_invperm=out_invperm,
                                     counters=m.attn_counters if two else None):
                raise RuntimeError("attend_chain: shape not covered by the Q4 attention kernel")
            return attn_out
        kc, vc = cache.paged_view(self.layer_idx)
        attn_out = m.temp_attn[:b * q_len].view(b, q_len, cfg.num_attention_heads, cfg.head_dim)
        if not ext.attn_decode_fused(q, k, v, kc, vc, attn_out, m.sin, m.cos, cache_seqlens, block_table, 0, cfg.rope_style,
                                     m.attn_scratch, m.attn_counters, out_invperm=out_invperm):
            raise RuntimeError("attend_chain: shape not covered by the one-launch attention kernel")
        return attn_out

    def _project_out(self, hidden_states, attn_out, b: int, q_len: int):
        """Back half (q_attn_forward_2, q_attn.cu:319-345): x += attn_out . Wo"""
        self.ext.q_attn_forward_2(self.q_handle, hidden_states, attn_out, b, q_len)
        return hidden_states

