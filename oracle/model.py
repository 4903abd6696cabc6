"""Oracle: full forward of a Llama-family EXL2 model (numpy) = the reference's torch path restated.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows ExLlamaV2.forward_chunk (model.py:936-1054) with every module in its "torch" form: Linear.forward(force_recons)
= matmul(x, reconstruct) (linear.py:370-379), RMSNorm (rmsnorm.py:187-211 / rms_norm.cu numerics), attention
(attn.py:1395-1536 forward_torch: q/k/v proj -> RoPE -> cache append -> causal softmax attention -> o_proj + residual),
MLP (mlp.py:453-505), greedy argmax (test_inference.py:607).  Activations are rounded to fp16 wherever the reference
stores fp16 tensors.
"""
from __future__ import annotations

import numpy as np

from . import exl2 as OX
from . import modules as OM

F16 = np.float16


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def _lin_tensors(w: dict) -> dict:
    """torch dict as handed to make_q_matrix -> on-disk numpy dict.  NOTE: must be taken BEFORE make_q_matrix mutates
    q_weight / q_scale_max in place."""
    return {k: _np(v).copy() for k, v in w.items() if k in ("q_weight", "q_scale", "q_scale_max", "q_groups", "q_invperm", "bias")}


class OracleModel:
    def __init__(self, cfg, ck: dict, rounding: str = "reference"):
        """rounding = "reference": fp16 roundings where the reference's kernel composition has them (q_attn.cu:153-345, q_mlp.cu:153-236:
        the normalised row, gate and up before the activation).  rounding = "chain": where our chained decode kernels have them instead
        (csrc/qgemv_lean.hip: the row times the norm weight rounded to fp16, the norm's scale applied to the fp32 dot product, act(gate) *
        up from the unrounded sums) -- the same mathematics, used by tests that want to tell a rounding-point difference from a defect:
        over random small models the two settings differ from each other by up to ~2 x the model tolerance in single logits
        (tests/test_chain.py::test_chain_decode_random_models).  Dense MLP, FP16 cache only."""
        assert rounding in ("reference", "chain")
        self.rounding = rounding
        self.cfg = cfg
        self.w = {}
        for k, v in ck.items():
            if isinstance(v, dict) and "qweight" in v:
                t = {kk: _np(vv).copy() for kk, vv in v.items() if kk in ("qweight", "qzeros", "scales", "g_idx")}
                self.w[k] = OX.gptq_reconstruct(t).astype(np.float32)
                self.w[k + ".bias"] = _np(v["bias"]).astype(np.float32) if "bias" in v else None
            elif isinstance(v, dict):
                self.w[k] = OX.exl2_reconstruct(_lin_tensors(v)).astype(np.float32)
                self.w[k + ".bias"] = _np(v["bias"]).astype(np.float32) if "bias" in v else None
            else:
                self.w[k] = _np(v).copy()
        self.sin, self.cos = OM.rope_tables(cfg.max_seq_len, cfg.head_dim, cfg.rotary_embedding_base, cfg.rope_style == 2)
        self.k_cache = None
        self.v_cache = None

    def linear(self, x, key):
        y = x.astype(np.float64) @ self.w[key].astype(np.float64)
        if self.w.get(key + ".bias") is not None:
            y = y + self.w[key + ".bias"]
        return y

    def reset(self, batch: int):
        c = self.cfg
        shape = (c.num_hidden_layers, batch, c.max_seq_len, c.num_key_value_heads, c.head_dim)
        self.k_cache = np.zeros(shape, dtype=F16)
        self.v_cache = np.zeros(shape, dtype=F16)
        self.seq_len = 0
        # Q4 cache state (ExLlamaV2Cache_Q4, cache.py:409-470): codes + one fp16 scale per 32 elements per layer, and ONE fp16
        # temp pair shared by all layers (cache.py:464-469) -- created on first use by forward(q4_cache=True)
        self.q4 = None

    def _q4_state(self, batch: int):
        c = self.cfg
        if self.q4 is None:
            kv_dim = c.num_key_value_heads * c.head_dim
            q_block = 1                                                    # cache.py:452-457
            while (kv_dim * q_block) % 512:
                q_block += 1
            T = (c.max_seq_len + q_block - 1) // q_block * q_block
            L = c.num_hidden_layers
            self.q4 = dict(
                q_block=q_block, kv_dim=kv_dim,
                k_codes=np.zeros((L, batch, T * kv_dim // 2), dtype=np.uint8), v_codes=np.zeros((L, batch, T * kv_dim // 2), dtype=np.uint8),
                k_scales=np.zeros((L, batch, T * kv_dim // 32), dtype=F16), v_scales=np.zeros((L, batch, T * kv_dim // 32), dtype=F16),
                temp_k=np.zeros((batch, T * kv_dim), dtype=F16), temp_v=np.zeros((batch, T * kv_dim), dtype=F16))
        return self.q4

    def _q4_get(self, layer: int, past: int):
        """ExLlamaV2Cache_Q.get_kv_state(layer, b, 0, past) (cache.py:472-514): the live range, widened to whole q_blocks of
        tokens, is dequantized into the shared temp pair (q_to_fp16_kv_kernel, cache.cu:372-400); width 0 unpacks nothing."""
        st = self.q4
        if past == 0:
            return
        qb, d = st["q_block"], st["kv_dim"]
        hi = (past + qb - 1) // qb * qb
        for side in ("k", "v"):
            for r in range(st["temp_" + side].shape[0]):
                st["temp_" + side][r, :hi * d] = OM.q4_unpack(st[side + "_codes"][layer, r, :hi * d // 2], st[side + "_scales"][layer, r, :hi * d // 32])

    def _q4_store(self, layer: int, past: int, s: int):
        """store_kv_state(layer, b, past, s) (cache.py:517-556): the touched token range, widened to whole q_blocks, is
        re-quantized from the temp pair (fp16_to_q_kv_kernel, cache.cu:196-220) -- earlier tokens of a shared block are
        re-quantized from their DEQUANTIZED values, like the reference does."""
        st = self.q4
        qb, d = st["q_block"], st["kv_dim"]
        lo = past // qb * qb
        hi = (past + s + qb - 1) // qb * qb
        for side in ("k", "v"):
            for r in range(st["temp_" + side].shape[0]):
                codes, scales = OM.q4_pack(st["temp_" + side][r, lo * d:hi * d])
                st[side + "_codes"][layer, r, lo * d // 2:hi * d // 2] = codes
                st[side + "_scales"][layer, r, lo * d // 32:hi * d // 32] = scales

    def q4_adopt(self, layer: int, k_codes, k_scales, v_codes, v_scales, n_tokens: int) -> float:
        """Checker's side of "follow the device": replaces the oracle's Q4 state of `layer`, tokens [0, n_tokens), by what the
        path under test left in ITS cache (uint8 codes [b, ...], fp16 scales [b, ...], any trailing shape) and returns the
        fraction of 4-bit codes that differed.  A 4-bit quantizer is discontinuous: a K/V element one fp16 ulp off flips a
        code at a rounding boundary, and from then on the two caches hold different values -- a multi-step comparison must
        either follow the device's codes (this) or accept a tolerance that grows with every step."""
        st = self.q4
        d = st["kv_dim"]
        b = st["temp_k"].shape[0]
        diff, total = 0, 0
        for side, codes, scales in (("k", k_codes, k_scales), ("v", v_codes, v_scales)):
            gc = np.asarray(codes).reshape(b, -1)[:, :n_tokens * d // 2]
            gs = np.asarray(scales).reshape(b, -1)[:, :n_tokens * d // 32]
            wc = st[side + "_codes"][layer][:, :n_tokens * d // 2]
            diff += int(((gc & 0xF) != (wc & 0xF)).sum() + ((gc >> 4) != (wc >> 4)).sum())
            total += 2 * gc.size
            st[side + "_codes"][layer][:, :n_tokens * d // 2] = gc
            st[side + "_scales"][layer][:, :n_tokens * d // 32] = gs.astype(F16)
        return diff / max(total, 1)

    def moe_mlp(self, x, n, p: str):
        """ExLlamaV2MoEMLP.forward_torch (moe_mlp.py:255-323): router logits = fp16 linear on the normed rows; softmax in fp32,
        top-k, renormalise, round to fp16 (oracle.modules.moe_route); per selected expert silu(w1 n) * (w3 n) -> w2, times the
        routing weight, summed over the experts in index order (index_add_ into a zero fp16 tensor: one fp16 rounding per
        expert contribution), plus the residual."""
        c = self.cfg
        gate = self.w[p + ".block_sparse_moe.gate"]
        raw = (n.astype(np.float32) @ gate.astype(np.float32).T).astype(F16)
        wts, mask = OM.moe_route(raw, c.num_experts_per_token)
        # how close the selection was (checker's bookkeeping, not part of the restated arithmetic): probability gap between the
        # last selected and the first rejected expert per row.  top-k is discontinuous: a path whose router logit is one fp16
        # ulp off may legitimately select another expert where this gap is ~ 1e-3, and a comparison must skip such rows.
        pr = raw.astype(np.float32)
        pr = np.exp(pr - pr.max(-1, keepdims=True)); pr /= pr.sum(-1, keepdims=True)
        srt = -np.sort(-pr, axis=-1)
        k = c.num_experts_per_token
        self.router_margin = np.minimum(self.router_margin, srt[:, k - 1] - srt[:, k]) if k < srt.shape[1] else self.router_margin
        out = np.zeros(x.shape, dtype=F16)
        for e in range(c.num_experts):
            sel = np.nonzero(mask[:, e])[0]
            if len(sel) == 0:
                continue
            q = f"{p}.block_sparse_moe.experts.{e}"
            g = self.linear(n[sel], q + ".w1").astype(F16)
            u = self.linear(n[sel], q + ".w3").astype(F16)
            d = self.linear(OM.silu_mul(g, u), q + ".w2").astype(F16)
            d = (d.astype(np.float32) * wts[sel, e].astype(np.float32)[:, None]).astype(F16)
            out[sel] = (out[sel].astype(np.float32) + d.astype(np.float32)).astype(F16)
        return (out.astype(np.float32) + x.astype(np.float32)).astype(F16)

    def _norm_linear(self, x, norm_key, key):
        """rounding = "chain": (fp16(x * w_norm) . W) * rsqrt(mean(x^2) + eps), float64"""
        xf = np.clip(x.astype(np.float64), -65504.0, 65504.0)
        xp = (x.astype(np.float32) * self.w[norm_key].astype(np.float32)[None, :]).astype(F16)
        rs = 1.0 / np.sqrt((xf ** 2).sum(-1, keepdims=True) / x.shape[-1] + self.cfg.norm_eps)
        return self.linear(xp, key) * rs

    def _forward_chain_rounding(self, ids: np.ndarray) -> np.ndarray:
        c = self.cfg
        b, s = ids.shape
        past = self.seq_len
        x = self.w["model.embed_tokens"][ids.reshape(-1)].reshape(b * s, c.hidden_size).astype(F16)
        for i in range(c.num_hidden_layers):
            p = f"model.layers.{i}"
            if (p + ".block_sparse_moe.gate") in self.w:
                raise NotImplementedError("OracleModel(rounding='chain'): dense MLP only")
            nk = p + ".input_layernorm"
            q = self._norm_linear(x, nk, p + ".self_attn.q_proj").astype(F16).reshape(b, s, c.num_attention_heads, c.head_dim)
            k = self._norm_linear(x, nk, p + ".self_attn.k_proj").astype(F16).reshape(b, s, c.num_key_value_heads, c.head_dim)
            v = self._norm_linear(x, nk, p + ".self_attn.v_proj").astype(F16).reshape(b, s, c.num_key_value_heads, c.head_dim)
            pos = np.full((b,), past)
            q = OM.rope_(q, self.sin, self.cos, pos, c.rope_style == 2)
            k = OM.rope_(k, self.sin, self.cos, pos, c.rope_style == 2)
            self.k_cache[i, :, past:past + s] = k
            self.v_cache[i, :, past:past + s] = v
            a = OM.attention(q, self.k_cache[i, :, :past + s], self.v_cache[i, :, :past + s])
            a = a.reshape(b * s, c.num_attention_heads * c.head_dim)
            x = (x.astype(np.float64) + self.linear(a, p + ".self_attn.o_proj")).astype(F16)
            nk = p + ".post_attention_layernorm"
            g = self._norm_linear(x, nk, p + ".mlp.gate_proj")
            u = self._norm_linear(x, nk, p + ".mlp.up_proj")
            y = np.clip((g / (1.0 + np.exp(-g))) * u, -65504.0, 65504.0).astype(F16)
            x = (x.astype(np.float64) + self.linear(y, p + ".mlp.down_proj")).astype(F16)
        self.seq_len = past + s
        self.router_margin = np.full((b,), np.inf)
        return self._norm_linear(x, "model.norm", "lm_head")[:, :c.vocab_size].reshape(b, s, c.vocab_size)

    def forward(self, ids: np.ndarray, q4_cache: bool = False) -> np.ndarray:
        """ids int [b, q_len] -> logits float64 [b, q_len, vocab] (before the final fp16 rounding)."""
        if self.rounding == "chain":
            if q4_cache:
                raise NotImplementedError("OracleModel(rounding='chain'): FP16 cache only")
            return self._forward_chain_rounding(ids)
        c = self.cfg
        b, s = ids.shape
        past = self.seq_len
        x = self.w["model.embed_tokens"][ids.reshape(-1)].reshape(b * s, c.hidden_size).astype(F16)
        self.router_margin = np.full((b * s,), np.inf)       # sparse-MoE layers: min over layers, per row (see moe_mlp)
        for i in range(c.num_hidden_layers):
            p = f"model.layers.{i}"
            n = OM.rms_norm(x, self.w[p + ".input_layernorm"], c.norm_eps)
            q = self.linear(n, p + ".self_attn.q_proj").astype(F16).reshape(b, s, c.num_attention_heads, c.head_dim)
            k = self.linear(n, p + ".self_attn.k_proj").astype(F16).reshape(b, s, c.num_key_value_heads, c.head_dim)
            v = self.linear(n, p + ".self_attn.v_proj").astype(F16).reshape(b, s, c.num_key_value_heads, c.head_dim)
            pos = np.full((b,), past)
            q = OM.rope_(q, self.sin, self.cos, pos, c.rope_style == 2)
            k = OM.rope_(k, self.sin, self.cos, pos, c.rope_style == 2)
            if q4_cache:
                # attn.py:1108-1196 over an ExLlamaV2Cache_Q4: earlier tokens are read back DEQUANTIZED, the step's own K/V
                # sit in the temp pair in fp16 while it attends, and are quantized into the codes afterwards
                st = self._q4_state(b)
                d = st["kv_dim"]
                self._q4_get(i, past)
                st["temp_k"][:, past * d:(past + s) * d] = k.reshape(b, s * d)
                st["temp_v"][:, past * d:(past + s) * d] = v.reshape(b, s * d)
                kk = st["temp_k"][:, :(past + s) * d].reshape(b, past + s, c.num_key_value_heads, c.head_dim)
                vv = st["temp_v"][:, :(past + s) * d].reshape(b, past + s, c.num_key_value_heads, c.head_dim)
                a = OM.attention(q, kk, vv)
                self._q4_store(i, past, s)
            else:
                self.k_cache[i, :, past:past + s] = k
                self.v_cache[i, :, past:past + s] = v
                a = OM.attention(q, self.k_cache[i, :, :past + s], self.v_cache[i, :, :past + s])
            a = a.reshape(b * s, c.num_attention_heads * c.head_dim)
            x = (x.astype(np.float64) + self.linear(a, p + ".self_attn.o_proj")).astype(F16)
            n = OM.rms_norm(x, self.w[p + ".post_attention_layernorm"], c.norm_eps)
            if (p + ".block_sparse_moe.gate") in self.w:
                x = self.moe_mlp(x, n, p)
                continue
            g = self.linear(n, p + ".mlp.gate_proj").astype(F16)
            u = self.linear(n, p + ".mlp.up_proj").astype(F16)
            y = OM.silu_mul(g, u)
            x = (x.astype(np.float64) + self.linear(y, p + ".mlp.down_proj")).astype(F16)
        self.seq_len = past + s
        self.router_margin = self.router_margin.reshape(b, s).min(axis=1)      # per sequence of this call
        n = OM.rms_norm(x, self.w["model.norm"], c.norm_eps)
        logits = self.linear(n, "lm_head")[:, :c.vocab_size]
        return logits.reshape(b, s, c.vocab_size)
