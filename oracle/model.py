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
    def __init__(self, cfg, ck: dict):
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

    def forward(self, ids: np.ndarray, q4_cache: bool = False) -> np.ndarray:
        """ids int [b, q_len] -> logits float64 [b, q_len, vocab] (before the final fp16 rounding)."""
        c = self.cfg
        b, s = ids.shape
        past = self.seq_len
        x = self.w["model.embed_tokens"][ids.reshape(-1)].reshape(b * s, c.hidden_size).astype(F16)
        for i in range(c.num_hidden_layers):
            p = f"model.layers.{i}"
            n = OM.rms_norm(x, self.w[p + ".input_layernorm"], c.norm_eps)
            q = self.linear(n, p + ".self_attn.q_proj").astype(F16).reshape(b, s, c.num_attention_heads, c.head_dim)
            k = self.linear(n, p + ".self_attn.k_proj").astype(F16).reshape(b, s, c.num_key_value_heads, c.head_dim)
            v = self.linear(n, p + ".self_attn.v_proj").astype(F16).reshape(b, s, c.num_key_value_heads, c.head_dim)
            pos = np.full((b,), past)
            q = OM.rope_(q, self.sin, self.cos, pos, c.rope_style == 2)
            k = OM.rope_(k, self.sin, self.cos, pos, c.rope_style == 2)
            self.k_cache[i, :, past:past + s] = k
            self.v_cache[i, :, past:past + s] = v
            if q4_cache:
                # cache.py:517-556: the touched 512-element blocks are re-quantized after every step and the next step
                # reads the dequantized values (cache.py:472-514); the current step attends over fp16 new tokens.
                pass
            a = OM.attention(q, self.k_cache[i, :, :past + s], self.v_cache[i, :, :past + s])
            a = a.reshape(b * s, c.num_attention_heads * c.head_dim)
            x = (x.astype(np.float64) + self.linear(a, p + ".self_attn.o_proj")).astype(F16)
            n = OM.rms_norm(x, self.w[p + ".post_attention_layernorm"], c.norm_eps)
            g = self.linear(n, p + ".mlp.gate_proj").astype(F16)
            u = self.linear(n, p + ".mlp.up_proj").astype(F16)
            y = OM.silu_mul(g, u)
            x = (x.astype(np.float64) + self.linear(y, p + ".mlp.down_proj")).astype(F16)
        self.seq_len = past + s
        n = OM.rms_norm(x, self.w["model.norm"], c.norm_eps)
        logits = self.linear(n, "lm_head")[:, :c.vocab_size]
        return logits.reshape(b, s, c.vocab_size)
