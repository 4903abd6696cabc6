"""ExLlamaV2Linear (reference linear.py:118-395): a quantized linear layer over a q_matrix handle."""
from __future__ import annotations

import torch

from .ext import none_tensor


class ExLlamaV2Linear:
    # Every row count goes to the HIP library: <= 16 rows the decode kernels, above that the dequantize-into-MFMA kernels
    # (qgemm_prefill.hip / qgemm_mfma.hip).  The reference's own M > 32 method -- reconstruct the fp16 matrix, library GEMM
    # (cuda/q_gemm.cu:243-263) -- exists here only where the reference exposes it as an argument: force_recons.

    def __init__(self, ext, key: str, in_features: int, out_features: int, model=None):
        self.model = model
        self.ext = ext
        self.key = key
        self.in_features = in_features
        self.out_features = out_features
        self.padding = -out_features % 32                     # linear.py:82-88
        self.q_handle = None
        self.q_tensors = None

    def load(self, w: dict, temp_dq=None, prescale: float = 1.0):
        """linear.py:118-171: keep the tensors alive (the handle stores raw pointers) and build the handle."""
        self.q_tensors = w
        self.q_handle = self.ext.make_q_matrix_from_dict(w, temp_dq if temp_dq is not None else none_tensor,
                                                         key=self.key, prescale=prescale)
        return self

    def unload(self):
        if self.q_handle is not None:
            self.ext.free_q_matrix(self.q_handle)             # linear.py:236-239
            self.q_handle = None
        self.q_tensors = None

    def forward(self, hidden_states: torch.Tensor, force_recons: bool = False, force_cuda: bool = False) -> torch.Tensor:
        """linear.py:361-379: gemm_half_q_half; reconstruct + torch.matmul ONLY when the caller asks for it (force_recons,
        the reference's argument of the same name)."""
        n = self.out_features + self.padding
        if force_recons:
            if self.model is not None:
                w = self.model.dq_scratch(self.in_features, n, hidden_states.device)
            else:
                w = torch.empty((self.in_features, n), dtype=torch.float16, device=hidden_states.device)
            self.ext.reconstruct(self.q_handle, w)
            out = torch.matmul(hidden_states, w)
            b = self.q_tensors.get("bias") if self.q_tensors else None
            if b is not None: out = out + b
        else:
            out = torch.empty(hidden_states.shape[:-1] + (n,), dtype=torch.float16, device=hidden_states.device)
            self.ext.gemm_half_q_half(hidden_states.contiguous().view(-1, self.in_features), self.q_handle,
                                      out.view(-1, n), force_cuda)
        return out[..., :self.out_features] if self.padding else out

    def weight_bytes(self) -> int:
        return self.ext.q_matrix_info(self.q_handle)["bytes"]
