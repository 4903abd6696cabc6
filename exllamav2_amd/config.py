"""Model hyper-parameters the hot path needs (subset of the reference's ExLlamaV2Config, config.py:58-196)."""
from __future__ import annotations

from dataclasses import dataclass, field


@dataclass
class ExLlamaV2Config:
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 32
    head_dim: int = 128
    vocab_size: int = 32000
    max_seq_len: int = 4096
    norm_eps: float = 1e-5
    rotary_embedding_base: float = 10000.0
    rope_style: int = 2                     # RopeStyle.NEOX (architecture.py:109-112)
    max_batch_size: int = 1
    max_input_len: int = 2048               # chunk size of forward (config.py:196, model.py:833)
    arch: str = "llama"
    num_experts: int = 0                    # > 0: Mixtral-style sparse MLP (moe_mlp.py), experts per token below
    num_experts_per_token: int = 2

    @staticmethod
    def llama2_7b(**kw):
        return ExLlamaV2Config(**kw)

    @staticmethod
    def tinyllama_1b(**kw):
        return ExLlamaV2Config(hidden_size=2048, intermediate_size=5632, num_hidden_layers=22, num_attention_heads=32,
                               num_key_value_heads=4, head_dim=64, vocab_size=32000, max_seq_len=2048, **kw)

    @staticmethod
    def llama2_70b(**kw):
        return ExLlamaV2Config(hidden_size=8192, intermediate_size=28672, num_hidden_layers=80, num_attention_heads=64,
                               num_key_value_heads=8, head_dim=128, **kw)

    @staticmethod
    def mixtral_8x7b(**kw):
        return ExLlamaV2Config(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                               num_key_value_heads=8, head_dim=128, vocab_size=32000, num_experts=8,
                               num_experts_per_token=2, arch="mixtral", **kw)

    @staticmethod
    def tiny_test(**kw):
        d = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                 num_key_value_heads=2, head_dim=64, vocab_size=320, max_seq_len=512, max_input_len=64)
        d.update(kw)
        return ExLlamaV2Config(**d)
