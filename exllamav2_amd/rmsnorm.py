"""ExLlamaV2RMSNorm (reference rmsnorm.py:115-211)."""
from __future__ import annotations

import torch


class ExLlamaV2RMSNorm:
    def __init__(self, ext, key: str, weight: torch.Tensor, eps: float):
        self.ext = ext
        self.key = key
        self.weight = weight
        self.variance_epsilon = eps

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        """rmsnorm.py:128-150 -> ext_c.rms_norm"""
        out = torch.empty_like(hidden_states)
        d = hidden_states.shape[-1]
        self.ext.rms_norm(hidden_states.view(-1, d), self.weight, out.view(-1, d), self.variance_epsilon)
        return out

    def forward_torch(self, hidden_states: torch.Tensor) -> torch.Tensor:
        """rmsnorm.py:187-211 (fp32 torch reference path)"""
        variance = hidden_states.to(torch.float32).pow(2).mean(-1, keepdim=True)
        hidden_states = hidden_states * torch.rsqrt(variance + self.variance_epsilon)
        hidden_states = hidden_states.to(self.weight.dtype)
        return hidden_states * self.weight
