"""ctypes binding of the C-ABI library (include/exl2_hip.h).

The product library is `exllamav2_amd/libexl2_hip.so` (gfx950 code objects, built by `__graft_entry__.build()`).
There is NO CPU fallback: `hip_lib()` raises if the library is missing.  `Lib(path)` can bind any library exporting the
same ABI; the test-suite uses that to drive the CPU *emulation build of the same sources* (tests/emu) for host-logic
tests -- the package itself never does.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# EXL2_HIP_LIB: alternative build of the SAME library (kernel-tuning A/B runs, trace build); never a different backend
HIP_LIB_PATH = os.environ.get("EXL2_HIP_LIB") or os.path.join(_HERE, "libexl2_hip.so")

vp = C.c_void_p
ci = C.c_int
cf = C.c_float
cll = C.c_longlong

# name -> (restype, argtypes).  Every symbol declared in include/exl2_hip.h appears here (tests check both ways).
PROTOTYPES = {
    "exl2_last_error": (C.c_char_p, []),
    "exl2_abi_version": (ci, []),
    # q_matrix
    "exl2_make_q_matrix": (ci, [C.POINTER(vp), ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp]),
    "exl2_free_q_matrix": (ci, [vp]),
    "exl2_q_matrix_info": (ci, [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), C.POINTER(cll)]),
    "exl2_reconstruct": (ci, [vp, vp, vp]),
    "exl2_gemm_half_q_half": (ci, [vp, vp, vp, ci, ci, vp, ci, ci, vp]),
    "exl2_make_group_map": (ci, [vp, ci, ci, vp, ci]),
    # norm / rope / activation
    "exl2_rms_norm": (ci, [vp, vp, vp, cf, ci, ci, ci, ci, ci, vp]),
    "exl2_rope_qk": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, ci, ci, vp]),
    "exl2_act_mul": (ci, [vp, vp, ci, ci, ci, vp, ci, vp]),
    # decode-loop utilities
    "exl2_embed_rows": (ci, [vp, vp, vp, ci, ci, ci, vp]),
    "exl2_argmax_rows": (ci, [vp, vp, ci, ci, ci, vp, vp, ci, ci, vp]),
    "exl2_add_i32": (ci, [vp, ci, ci, vp]),
    "exl2_sample_rows": (ci, [vp, ci, ci, ci, ci, vp, cf, ci, cf, cf, cf, vp, vp, vp, vp]),
    "exl2_sample_rows_step": (ci, [vp, ci, ci, ci, ci, vp, cf, ci, cf, cf, vp, ci, vp, vp, vp, vp, vp, vp, ci, ci, vp]),
    # quantized KV cache
    "exl2_fp16_to_q_kv": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, ci, ci, vp]),
    "exl2_q_to_fp16_kv": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, ci, ci, vp]),
    "exl2_fp16_to_fp8": (ci, [vp, vp, ci, cll, ci, ci, ci, vp]),
    "exl2_fp8_to_fp16": (ci, [vp, vp, ci, cll, ci, ci, ci, vp]),
    "exl2_cache_rotate": (ci, [vp, vp, cll, ci, vp]),
    "exl2_count_match": (ci, [vp, vp, ci, ci, vp]),
    # peer copies (single-process tensor parallel)
    "exl2_release_scratch": (cll, [vp, ci]),
    "exl2_memcpy_2d_async": (ci, [vp, cll, vp, cll, cll, cll, vp]),
    "exl2_enable_peer_access": (ci, [C.POINTER(ci), ci]),
    # load path
    "exl2_stloader_read": (ci, [C.c_char_p, C.c_ulonglong, C.c_ulonglong, vp, ci, vp]),
    "exl2_tensor_remap": (ci, [vp, ci, ci, vp]),
    "exl2_tensor_remap_4bit": (ci, [vp, ci, ci, vp]),
    # attention
    "exl2_paged_attn_scratch_bytes": (cll, [ci, ci, ci]),
    "exl2_flash_prefill": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, cf, ci, vp]),
    "exl2_paged_attn": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, cf, ci, ci, vp, cll, vp]),
    "exl2_paged_attn_ex": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, cf, ci, ci, vp, cll, ci, cf, vp]),
    "exl2_flash_prefill_ex": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, cf, ci, ci, cf, vp]),
    "exl2_rope_kv_append": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, ci, ci, ci, ci, vp]),
    "exl2_attn_decode_fused": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, cf, ci, ci, ci,
                                    vp, cll, vp, ci, vp, vp]),
    "exl2_attn_decode_fused_dual": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, cf, ci, ci, ci,
                                    vp, cll, vp, ci, vp, vp, vp]),
    "exl2_paged_attn_q4": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, cf, ci, ci, vp, cll, vp, vp]),
    "exl2_paged_attn_q4_merged": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, cf, ci, ci, vp, cll, vp, vp, ci, vp]),
    "exl2_attn_q4_decode_fused": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, cf, ci, ci, ci, vp, cll, vp, ci, vp, vp]),
    "exl2_rope_quant_append_q4": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, ci, ci, ci, ci, vp]),
    # fused modules
    "exl2_make_q_attn": (ci, [C.POINTER(vp), vp, vp, ci, ci, cf, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci,
                              ci, vp, vp, vp, vp, ci, ci]),
    "exl2_free_q_attn": (ci, [vp]),
    "exl2_q_attn_forward_1": (ci, [vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, ci, vp]),
    "exl2_q_attn_forward_2": (ci, [vp, vp, vp, ci, ci, vp]),
    "exl2_make_q_mlp": (ci, [C.POINTER(vp), vp, vp, ci, cf, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, ci, ci]),
    "exl2_free_q_mlp": (ci, [vp]),
    "exl2_q_mlp_forward": (ci, [vp, vp, ci, vp]),
    "exl2_make_q_moe_mlp": (ci, [C.POINTER(vp), vp, vp, ci, cf, vp, ci, ci, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                 vp, vp, vp, vp, vp, vp, ci, ci]),
    "exl2_free_q_moe_mlp": (ci, [vp]),
    "exl2_q_moe_mlp_forward": (ci, [vp, vp, ci, vp]),
    "exl2_q_moe_mlp_forward_chain": (ci, [vp, vp, ci, vp, vp, vp, vp, C.POINTER(ci), vp]),
    "exl2_moe_route": (ci, [vp, vp, vp, ci, ci, ci, ci, vp]),
    # chained decode (csrc/qgemv_flat.hip)
    "exl2_q_attn_chain_info": (ci, [vp, C.POINTER(ci), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
    "exl2_q_mlp_chain_info": (ci, [vp, C.POINTER(ci), C.POINTER(vp), C.POINTER(vp)]),
    "exl2_q_matrix_perm_info": (ci, [vp, C.POINTER(vp), C.POINTER(vp)]),
    "exl2_q_attn_forward_1_chain": (ci, [vp, vp, vp, ci, ci, vp, vp, vp, vp]),
    "exl2_q_attn_forward_1_chain_rope": (ci, [vp, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]),
    "exl2_q_attn_forward_2_chain": (ci, [vp, vp, vp, ci, vp, vp, vp, vp, C.POINTER(ci), vp]),
    "exl2_q_mlp_forward_chain": (ci, [vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, C.POINTER(ci), vp]),
    "exl2_q_mlp_forward_chain_part": (ci, [vp, ci, ci, vp, vp, vp, ci, ci, vp, vp, vp, vp, C.POINTER(ci), vp]),
    "exl2_gemm_half_q_half_chain": (ci, [vp, vp, ci, cf, vp, vp, ci, vp]),
    "exl2_embed_rows_chain": (ci, [vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp]),
    "exl2_chain_set_tiled": (ci, [ci]),
    "exl2_publish_rows": (ci, [vp, ci, ci, vp, vp, vp, vp, vp]),
    "exl2_gather_f16": (ci, [vp, vp, vp, ci, vp]),
    "exl2_chain_overlap_begin": (ci, [vp, ci, vp, vp]),
    "exl2_chain_overlap_end": (ci, [C.POINTER(ci)]),
    "exl2_chain_route_counts": (ci, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), ci]),
    "exl2_prefill_route_info": (ci, [C.POINTER(ci)]),
    # graphs
    "exl2_graph_begin_capture": (ci, [vp]),
    "exl2_graph_end_capture": (ci, [vp, C.POINTER(vp)]),
    "exl2_graph_launch": (ci, [vp, vp]),
    "exl2_graph_free": (ci, [vp]),
}


class Exl2Error(RuntimeError):
    pass


class Lib:
    def __init__(self, path: str):
        if not os.path.exists(path):
            raise Exl2Error(
                f"{path} not found: the HIP library is not built. Run `python -c \"import __graft_entry__ as g; "
                f"g.build()\"` at the repo root (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        self.path = path
        self.dll = C.CDLL(path)
        self.missing = []
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(self.dll, name)
            except AttributeError:
                self.missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        # (EXL2_LIB_ALLOW_MISSING=1: bisecting with a library built from an OLDER source state -- tools only; a call of a missing entry
        # point is then an AttributeError)
        if self.missing and os.environ.get("EXL2_LIB_ALLOW_MISSING", "0") == "0":
            raise Exl2Error(f"{path} does not export: {', '.join(self.missing)}")

    def last_error(self) -> str:
        return (self.exl2_last_error() or b"").decode("utf-8", "replace")

    def check(self, rc: int) -> int:
        if rc < 0:
            raise Exl2Error(self.last_error() or f"exl2 error {rc}")
        return rc


_hip = None
_lock = threading.Lock()


def hip_lib() -> Lib:
    global _hip
    if _hip is None:
        with _lock:
            if _hip is None:
                _hip = Lib(HIP_LIB_PATH)
    return _hip
