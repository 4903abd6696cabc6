"""Drop-in module `exllamav2_ext`: put this directory on sys.path and the reference's `exllamav2/ext.py:105-109` picks
it up instead of JIT-building its CUDA sources (`import exllamav2_ext` succeeds -> `ext_c = exllamav2_ext`).

Every name of the reference's pybind module (ext_bindings.cpp:27-138) that lies on the quantized forward path is
forwarded, with the reference's argument order, to the MI355X library through `exllamav2_amd.ext_c` (ctypes over the
C ABI of include/exl2_hip.h).  Names outside the hot path (sampler, safetensors loader, quantizer, LoRA, vision, TP
host-staging) raise NotImplementedError with the SURVEY.md section that scopes them out -- loudly, never silently.
"""
from exllamav2_amd.ext import ext_c as _e

make_q_matrix = _e.make_q_matrix
free_q_matrix = _e.free_q_matrix
reconstruct = _e.reconstruct
gemm_half_q_half = _e.gemm_half_q_half
make_group_map = _e.make_group_map
rms_norm = _e.rms_norm
rms_norm_ = _e.rms_norm_
rope_ = _e.rope_
fp16_to_q_kv = _e.fp16_to_q_kv
q_to_fp16_kv = _e.q_to_fp16_kv
make_q_attn = _e.make_q_attn
free_q_attn = _e.free_q_attn
q_attn_forward_1 = _e.q_attn_forward_1
q_attn_forward_2 = _e.q_attn_forward_2
make_q_mlp = _e.make_q_mlp
free_q_mlp = _e.free_q_mlp
q_mlp_forward_ = _e.q_mlp_forward_


def set_flash_attn_func():          # ext_qattn.cpp:256-259 is a no-op in the reference too
    return None


def q_attn_set_loras(*a, **k):      # LoRA is out of scope; an empty set is the only accepted state
    return 0


def q_mlp_set_loras(*a, **k):
    return 0


def _out_of_scope(name, why):
    def f(*a, **k):
        raise NotImplementedError(f"exllamav2_ext.{name}: {why}")
    f.__name__ = name
    return f


make_q_matrix_split = _e.make_q_matrix_split
for _n in ("gemm_half_q_half_tp", "make_tp_context", "free_tp_context", "tp_broadcast", "tp_gather",
           "tp_cross_device_barrier", "tp_all_reduce", "tp_attn_forward_", "tp_attn_forward_paged_", "tp_mlp_forward_",
           "rms_norm_tp"):
    globals()[_n] = _out_of_scope(_n, "the reference's single-process, host-staged tensor-parallel bindings are not mirrored; tensor parallel here is one process per GPU over RCCL: exllamav2_amd/tensor_p.py (SURVEY.md 8e)")
make_q_moe_mlp = _e.make_q_moe_mlp
free_q_moe_mlp = _e.free_q_moe_mlp
q_moe_mlp_forward_ = _e.q_moe_mlp_forward_
fp16_to_fp8 = _e.fp16_to_fp8
fp8_to_fp16 = _e.fp8_to_fp16
cache_rotate = _e.cache_rotate
count_match = _e.count_match
matrix_fp16_to_q4 = _e.matrix_fp16_to_q4
matrix_q4_to_fp16 = _e.matrix_q4_to_fp16
for _n in ("layer_norm", "layer_norm_", "head_norm", "head_norm_", "softcap_", "gen_mrope_pos_ids", "gemm_half_half_half",
           "had_paley", "had_paley2", "pack_rows_4", "pack_columns", "quantize", "quantize_err", "quantize_range",
           "quantize_range_inplace", "sim_anneal", "apply_rep_penalty", "sample_basic", "logit_filter_exclusive",
           "fast_fill_cpu_ones_bool", "fast_fadd_cpu", "fast_copy_cpu", "partial_strings_match", "dump_profile_results",
           "stloader_read", "stloader_open_file", "stloader_close_file", "tensor_remap", "tensor_remap_4bit"):
    globals()[_n] = _out_of_scope(_n, "outside the quantized forward path (SURVEY.md 2.2: OUT OF SCOPE)")
