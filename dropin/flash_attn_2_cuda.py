"""`flash_attn_2_cuda` stand-in: the reference imports it next to `flash_attn` when it detects paged-attention support
(attn.py:56) and only calls into it from its single-process tensor-parallel bindings (ext_qattn.cpp:291, 416-438), which
this drop-in serves through dropin/exllamav2_ext.py's own TP entry points -- so nothing here is ever called."""


def fwd_kvcache(*a, **k):
    raise NotImplementedError("flash_attn_2_cuda.fwd_kvcache: not used by the drop-in (paged attention goes through "
                              "flash_attn.flash_attn_with_kvcache -> libexl2_hip.so)")
