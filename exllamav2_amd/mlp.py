"""ExLlamaV2MLP (reference mlp.py:150-223 load/make_q_mlp, :318-361 forward)."""
from __future__ import annotations

import torch

from .ext import none_tensor
from .linear import ExLlamaV2Linear


class ExLlamaV2MLP:
    def __init__(self, model, key: str, layer_idx: int):
        cfg = model.config
        self.model, self.ext, self.key, self.layer_idx = model, model.ext, key, layer_idx
        self.gate_proj = ExLlamaV2Linear(self.ext, key + ".mlp.gate_proj", cfg.hidden_size, cfg.intermediate_size, model)
        self.up_proj = ExLlamaV2Linear(self.ext, key + ".mlp.up_proj", cfg.hidden_size, cfg.intermediate_size, model)
        self.down_proj = ExLlamaV2Linear(self.ext, key + ".mlp.down_proj", cfg.intermediate_size, cfg.hidden_size, model)
        self.q_handle = None

    def load(self, ck: dict):
        cfg, m = self.model.config, self.model
        for lin in (self.gate_proj, self.up_proj, self.down_proj):
            lin.load(ck[lin.key])
        self.post_attention_layernorm = ck[self.key + ".post_attention_layernorm"]
        self.q_handle = self.ext.make_q_mlp(                                     # mlp.py:204-223
            self.post_attention_layernorm, none_tensor, True, cfg.norm_eps,
            self.gate_proj.q_handle, self.up_proj.q_handle, self.down_proj.q_handle,
            m.temp_state, m.temp_a, m.temp_b, none_tensor, m.max_rows, False, True, none_tensor, none_tensor, False, False)
        return self

    def unload(self):
        if self.q_handle is not None:
            self.ext.free_q_mlp(self.q_handle)
            self.q_handle = None
        for lin in (self.gate_proj, self.up_proj, self.down_proj):
            lin.unload()

    def forward(self, hidden_states: torch.Tensor):
        """mlp.py:353: ext_c.q_mlp_forward_ in place, for every row count (<= 16 rows: the fused decode launches; above:
        row pre-pass + dequantize-into-MFMA GEMMs, csrc/modules.hip:exl2_q_mlp_forward)."""
        self.ext.q_mlp_forward_(self.q_handle, hidden_states)
        return hidden_states
