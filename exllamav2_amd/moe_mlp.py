"""ExLlamaV2MoEMLP (reference moe_mlp.py:25-364): Mixtral-style sparse MLP over per-expert q_matrix handles.

The reference runs its fused extension path only for <= 4 rows and otherwise loops over experts in torch with a host
sync per layer (moe_mlp.py:238, 255-323).  Here the fused path (ext_c.q_moe_mlp_forward_) takes every decode-sized batch:
routing weights stay on the device, each expert's kernels see all rows and skip the ones not routed to them."""
from __future__ import annotations

import torch

from .ext import none_tensor
from .linear import ExLlamaV2Linear


class ExLlamaV2MoEMLP:
    def __init__(self, model, key: str, layer_idx: int, num_experts: int, num_experts_per_token: int):
        cfg = model.config
        self.model, self.ext, self.key, self.layer_idx = model, model.ext, key, layer_idx
        self.num_experts, self.num_experts_per_token = num_experts, num_experts_per_token
        h, inter = cfg.hidden_size, cfg.intermediate_size
        mk = lambda e, name, k, n: ExLlamaV2Linear(self.ext, f"{key}.block_sparse_moe.experts.{e}.{name}", k, n, model)
        self.w1 = [mk(e, "w1", h, inter) for e in range(num_experts)]       # gate projection
        self.w2 = [mk(e, "w2", inter, h) for e in range(num_experts)]       # down projection
        self.w3 = [mk(e, "w3", h, inter) for e in range(num_experts)]       # up projection
        self.gate = None
        self.q_handle = None

    def load(self, ck: dict):
        cfg, m = self.model.config, self.model
        for lin in self.w1 + self.w2 + self.w3:
            lin.load(ck[lin.key])
        self.post_attention_layernorm = ck[self.key + ".post_attention_layernorm"]
        self.gate = ck[self.key + ".block_sparse_moe.gate"]                  # fp16 [E, hidden] (nn.Linear weight)
        self.temp_logits = torch.empty((m.max_rows, self.num_experts), dtype=torch.float16, device=m.device)
        self.q_handle = self.ext.make_q_moe_mlp(                             # moe_mlp.py:114-133
            self.post_attention_layernorm, none_tensor, True, cfg.norm_eps, self.gate, self.num_experts,
            self.num_experts_per_token, [l.q_handle for l in self.w1], [l.q_handle for l in self.w2],
            [l.q_handle for l in self.w3], m.temp_state, none_tensor, m.temp_a, m.temp_b, self.temp_logits, none_tensor,
            m.max_rows, False)
        return self

    def unload(self):
        if self.q_handle is not None:
            self.ext.free_q_moe_mlp(self.q_handle)
            self.q_handle = None
        for lin in self.w1 + self.w2 + self.w3:
            lin.unload()

    def forward(self, hidden_states: torch.Tensor):
        """moe_mlp.py:238-253: ext_c.q_moe_mlp_forward_ in place (x += sum_e w_e * down_e(act(gate_e n) * up_e n))."""
        h = hidden_states.shape[-1]
        self.ext.q_moe_mlp_forward_(self.q_handle, hidden_states.view(-1, h))
        return hidden_states
