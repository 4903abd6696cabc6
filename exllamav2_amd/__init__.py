"""exllamav2_amd -- MI355X-native (gfx950) implementation of ExLlamaV2's quantized forward path.

Scope: the `ext_c.*` operator surface of the reference (q_matrix handles, gemm_half_q_half,
reconstruct, RMSNorm, RoPE, fused attention / MLP module forwards, paged attention, Q4 KV-cache codec) as hand-written
HIP kernels behind a C-ABI shared library (include/exl2_hip.h), plus the thin Python host mirror of the reference's
module interface needed to drive it.
"""
from .ext import ext_c, none_tensor, ExtC  # noqa: F401
from .config import ExLlamaV2Config  # noqa: F401
from .model import ExLlamaV2, GreedyGraphDecoder  # noqa: F401
from .cache import ExLlamaV2Cache, ExLlamaV2Cache_Q4  # noqa: F401

__version__ = "0.1.0"
