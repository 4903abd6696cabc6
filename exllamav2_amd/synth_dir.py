"""A synthetic EXL2 checkpoint as a MODEL DIRECTORY (config.json + model.safetensors) in the reference's on-disk format,
so that the reference's own loader (config.py:214-, stloader.py, module.py:116-121) can read what `synth.py` draws.

Tensor names and dtypes follow SURVEY.md A.1: `<key>.q_weight` int32, `.q_invperm` int32, `.q_scale` int32,
`.q_scale_max` fp16, `.q_groups` int16; norms / embedding `<key>.weight` fp16.  (`q_perm` is derived at load time,
module.py:120; `q_group_map` by make_group_map, ext.py:340-345.)
"""
from __future__ import annotations

import json
import os

import torch


def write_model_dir(path: str, cfg, ck: dict) -> str:
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    tensors = {}
    for key, v in ck.items():
        if isinstance(v, dict):
            for name in ("q_weight", "q_invperm", "q_scale", "q_scale_max", "q_groups", "bias"):
                if name in v:
                    tensors[f"{key}.{name}"] = v[name].detach().cpu().contiguous()
        else:
            tensors[f"{key}.weight"] = v.detach().cpu().contiguous()
    save_file(tensors, os.path.join(path, "model.safetensors"))
    config = {
        "architectures": ["LlamaForCausalLM"], "model_type": "llama",
        "hidden_size": cfg.hidden_size, "intermediate_size": cfg.intermediate_size,
        "num_hidden_layers": cfg.num_hidden_layers, "num_attention_heads": cfg.num_attention_heads,
        "num_key_value_heads": cfg.num_key_value_heads, "head_dim": cfg.head_dim, "vocab_size": cfg.vocab_size,
        "max_position_embeddings": cfg.max_seq_len, "rms_norm_eps": cfg.norm_eps, "rope_theta": cfg.rotary_embedding_base,
        "hidden_act": "silu", "bos_token_id": 1, "eos_token_id": 2, "pad_token_id": 0, "tie_word_embeddings": False,
        "torch_dtype": "float16",
    }
    if getattr(cfg, "num_experts", 0):
        # a sparse-MoE checkpoint in the reference's Mixtral form (architecture.py:291-305: block_sparse_moe.gate / experts.*.w1|w2|w3,
        # config.py:322-323: num_local_experts / num_experts_per_tok)
        config.update({"architectures": ["MixtralForCausalLM"], "model_type": "mixtral",
                       "num_local_experts": cfg.num_experts, "num_experts_per_tok": cfg.num_experts_per_token})
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(config, f)
    return path


def write_tokenizer(path: str, vocab_size: int) -> None:
    """A tiny HF tokenizer.json (byte-level BPE without merges: ids 0 / 1 / 2 = <pad> / <s> / </s>, 3..258 = the 256 byte
    symbols, the rest filler tokens) so that the reference's ExLlamaV2Tokenizer (tokenizer/tokenizer.py:122-) and with it
    ExLlamaV2DynamicGenerator can be constructed on a synthetic model directory.  Needs the `tokenizers` package (host
    side, load time only)."""
    from tokenizers import Tokenizer, models, pre_tokenizers, decoders
    assert vocab_size >= 259
    vocab = {"<pad>": 0, "<s>": 1, "</s>": 2}
    for ch in pre_tokenizers.ByteLevel.alphabet():
        pass
    alphabet = sorted(pre_tokenizers.ByteLevel.alphabet())
    for i, ch in enumerate(alphabet):
        vocab[ch] = 3 + i
    for i in range(3 + len(alphabet), vocab_size):
        vocab[f"<filler{i}>"] = i
    tok = Tokenizer(models.BPE(vocab=vocab, merges=[], unk_token=None))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    tok.save(os.path.join(path, "tokenizer.json"))
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump({"bos_token": "<s>", "eos_token": "</s>", "pad_token": "<pad>"}, f)
