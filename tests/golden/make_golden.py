#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING the reference's own pure-Python/torch code on the CPU (this container only:
/root/reference is not on the GPU box).  The reference's CUDA kernels cannot run here, so what can be pinned by
execution is the host-side / torch-path semantics the oracle restates:

  * ext.make_group_map_py ............ exllamav2/ext.py:301-316
  * ExLlamaV2RMSNorm.forward_torch ... exllamav2/rmsnorm.py:187-211
  * ExLlamaV2Attention._attn_torch ... exllamav2/attn.py:905-933 (matmul + causal mask attn_params.py:109-133 + fp16
                                       softmax, GQA repeat_kv).  NOTE: with SDPA enabled, v0.3.2's branch at attn.py:884-885
                                       takes `get_block_diag_mask` for causal=True, which is None without cu_seqlens, i.e.
                                       NO mask; the fixture therefore pins the matmul branch (cfg.no_sdpa = True), whose
                                       causal bottom-right semantics equal flash-attn's.
  * sin / cos tables ................. exllamav2/device.py:118-169 (prepare_sincos) with rope.get_rope_params
  * MLP activation ................... exllamav2/mlp.py:486-494 (F.silu(gate) * up, clamp)

The compiled extension is replaced by an inert stub module (nothing on these paths calls into it).
Run:  python tests/golden/make_golden.py     (writes next to this file)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    stub = types.ModuleType("exllamav2_ext")
    stub.__file__ = "exllamav2_ext_stub"

    def _getattr(name):
        if name.startswith("__"):
            raise AttributeError(name)
        return lambda *a, **k: None
    stub.__getattr__ = _getattr
    sys.modules["exllamav2_ext"] = stub
    sys.path.insert(0, "/root/reference")
    import exllamav2  # noqa: F401
    return exllamav2


def main():
    import_reference()
    from types import SimpleNamespace as NS
    from exllamav2 import ext as ref_ext
    from exllamav2.rmsnorm import ExLlamaV2RMSNorm
    from exllamav2.attn import ExLlamaV2Attention
    from exllamav2.attn_params import Params
    from exllamav2.device import ExLlamaV2DeviceContext
    from exllamav2.architecture import RopeStyle
    import torch.nn.functional as F

    torch.manual_seed(0)
    out = {}

    # ---- group map --------------------------------------------------------------------------------------------------
    q_groups = torch.tensor([8, 0, 6, 8, 6, 14, 4, 20, 4, 36, 3, 52, 2, 58], dtype=torch.short)
    num_qrows = 62
    out["gm_q_groups"] = q_groups.numpy()
    out["gm_num_qrows"] = np.array(num_qrows)
    out["gm_map"] = ref_ext.make_group_map_py(q_groups, num_qrows).numpy()

    # ---- RMSNorm ----------------------------------------------------------------------------------------------------
    x = (torch.randn(5, 256) * 3).half()
    w = (1 + 0.1 * torch.randn(256)).half()
    fake = NS(weight=w, variance_epsilon=1e-5)
    out["rms_x"], out["rms_w"] = x.numpy(), w.numpy()
    out["rms_y"] = ExLlamaV2RMSNorm.forward_torch(fake, x.clone()).numpy()

    # ---- attention (SDPA lower-right causal, GQA) ---------------------------------------------------------------------
    b, sq, sk, nh, kvh, hd = 2, 3, 11, 4, 2, 64
    q = torch.randn(b, sq, nh, hd).half()
    k = torch.randn(b, sk, kvh, hd).half()
    v = torch.randn(b, sk, kvh, hd).half()
    attn_self = NS(num_key_value_groups=nh // kvh, sliding_window=0, layer_idx=0, scaling=hd ** -0.5,
                   repeat_kv=lambda t, n: ExLlamaV2Attention.repeat_kv(None, t, n))
    params = Params(b, sq, sk - sq, None, None)
    cfg = NS(no_sdpa=True, attn_logit_softcapping=None)
    y = ExLlamaV2Attention._attn_torch(attn_self, b, sq, q.clone(), k.clone(), v.clone(), params, cfg, causal=True)
    out["att_q"], out["att_k"], out["att_v"] = q.numpy(), k.numpy(), v.numpy()
    out["att_y"] = y.float().numpy().reshape(b, sq, nh, hd)

    # ---- sin / cos tables ---------------------------------------------------------------------------------------------
    for style, tag in ((RopeStyle.NEOX, "neox"), (RopeStyle.GPTJ, "gptj")):
        mcfg = NS(rotary_embedding_base=10000.0, scale_pos_emb=1.0, rotary_embedding_base_alt=None, scale_pos_emb_alt=None,
                  max_seq_len=48, pos_id_index=0, head_dim=64, hidden_size=256, num_attention_heads=4,
                  alt_rope_method=None, scale_alpha_value=1.0, partial_rotary_factor=1.0, l3_rope_factor=None,
                  yarn_rope_factor=None, original_max_seq_len=0, scale_long_factor=None, scale_short_factor=None,
                  arch=NS(lm=NS(rope_style=style, rope_freq_half=False)))
        ctx = object.__new__(ExLlamaV2DeviceContext)
        ctx.model = NS(config=mcfg)
        ctx.device_idx = -1
        ctx.archparams = NS(rope_style=style)
        ctx.sin = ctx.cos = None
        ExLlamaV2DeviceContext.prepare_sincos(ctx)
        out[f"sin_{tag}"] = ctx.sin[0][0, 0].numpy()
        out[f"cos_{tag}"] = ctx.cos[0][0, 0].numpy()

    # ---- act * mul ----------------------------------------------------------------------------------------------------
    g = (torch.randn(4, 128) * 2).half()
    u = torch.randn(4, 128).half()
    yy = F.silu(g)
    yy *= u
    yy.clamp_(min=-65504.0, max=65504.0)
    out["act_g"], out["act_u"], out["act_y"] = g.numpy(), u.numpy(), yy.numpy()

    np.savez_compressed(os.path.join(HERE, "reference_torch_paths.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_torch_paths.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
