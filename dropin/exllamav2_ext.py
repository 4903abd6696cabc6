"""Drop-in module `exllamav2_ext`: put this directory on sys.path and the reference's `exllamav2/ext.py:105-109` picks
it up instead of JIT-building its CUDA sources (`import exllamav2_ext` succeeds -> `ext_c = exllamav2_ext`).

Every name of the reference's pybind module (ext_bindings.cpp:27-138) that lies on the quantized forward path is
forwarded, with the reference's argument order, to the MI355X library through `exllamav2_amd.ext_c` (ctypes over the
C ABI of include/exl2_hip.h).  Names outside the hot path (sampler chain, quantizer, LoRA, vision) raise
NotImplementedError with the SURVEY.md section that scopes them out -- loudly, never silently.
"""
import os as _os

import numpy as _np
import torch as _torch

from exllamav2_amd.ext import ext_c as _e
from exllamav2_amd import ext_tp as _tp

# make_q_matrix (ext_qmatrix.cpp:21-111) re-lays q_weight out IN PLACE -- the reference shuffles it in place too
# (q_matrix.cu:123-196) and keeps its own q_tensors alive, so no second copy of the packed weights exists (a private copy per
# handle doubled the weight VRAM of every drop-in user).
# The one consumer of the ORIGINAL layout is the reference's single-process tensor-parallel loader: after load() it slices
# q_weight by output columns (linear.py:572-576) and hands the slices to make_q_matrix_split.  Its own shuffle is column-local,
# so its slices stay valid; the tile-major layout of libexl2_hip.so is not sliceable.  That loader is recognised by what it
# passes: `model.load_tp` loads every module with `device_context = False` (model.py:449), and load() then hands make_q_matrix
# the placeholder `none_tensor` -- a meta tensor -- as temp_dq (linear.py:144-147); every other load path passes a scratch slice.
# For such a call make_q_matrix re-lays out a PRIVATE copy (held while the handle lives) and leaves the caller's tensor as
# loaded, so the column slices make_q_matrix_split receives afterwards are weights.  EXL2_DROPIN_TP=1 forces the private copy for
# every handle (a host that slices q_weight on a path of its own).
_private_weights = {}


def _tp_opt_in():
    return _os.environ.get("EXL2_DROPIN_TP", "0") != "0"


def _is_placeholder(t):
    return t is None or (isinstance(t, _torch.Tensor) and t.is_meta)


def make_q_matrix(q_weight, q_perm, q_invperm, q_scale, q_scale_max, q_groups, q_group_map, gptq_qzeros, gptq_scales,
                  gptq_g_idx, bias, temp_dq, max_dq_rows):
    if not (_tp_opt_in() or _is_placeholder(temp_dq)):
        return _e.make_q_matrix(q_weight, q_perm, q_invperm, q_scale, q_scale_max, q_groups, q_group_map, gptq_qzeros,
                                gptq_scales, gptq_g_idx, bias, temp_dq, max_dq_rows)
    own = q_weight.clone()
    h = _e.make_q_matrix(own, q_perm, q_invperm, q_scale, q_scale_max, q_groups, q_group_map, gptq_qzeros, gptq_scales,
                         gptq_g_idx, bias, temp_dq, max_dq_rows)
    _private_weights[h] = own
    return h


def free_q_matrix(handle):
    _e.free_q_matrix(handle)
    _private_weights.pop(handle, None)


def make_q_matrix_split(*args):
    # (its inputs are fresh contiguous column slices of the tensors load(device_context = False) kept -- see above: re-laid out in
    # place, they are this handle's own)
    return _e.make_q_matrix_split(*args)


reconstruct = _e.reconstruct
make_group_map = _e.make_group_map
fp16_to_q_kv = _e.fp16_to_q_kv
q_to_fp16_kv = _e.q_to_fp16_kv
make_q_attn = _e.make_q_attn
make_q_mlp = _e.make_q_mlp


# ---- the per-token calls: compiled binding + the module chain behind the boundary (dropin/_exl2_fast.cpp) --------------------
# `_exl2_fast.so` is built next to this file by __graft_entry__.build() (exllamav2_amd/build.py: build_fast).  It binds
# q_attn_forward_1 / q_attn_forward_2 / q_mlp_forward_ (and the decode case of flash_attn_func, dropin/flash_attn) with pybind11
# straight over the C ABI -- what the reference's own extension is (ext_bindings.cpp) -- and chains consecutive modules through the
# library's fast decode kernels.  EXL2_DROPIN_FAST=0 keeps every call on the ctypes route of exllamav2_amd/ext.py (same library,
# module by module); a missing .so does the same with a warning, never silently.
def _load_fast():
    if _os.environ.get("EXL2_DROPIN_FAST", "1") == "0":
        return None
    import importlib.machinery, importlib.util, warnings
    path = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "_exl2_fast.so")
    try:
        if not _os.path.exists(path):
            raise ImportError(path + " is not built (python -c 'import __graft_entry__ as g; g.build()')")
        loader = importlib.machinery.ExtensionFileLoader("_exl2_fast", path)
        mod = importlib.util.module_from_spec(importlib.util.spec_from_loader("_exl2_fast", loader))
        loader.exec_module(mod)
        from exllamav2_amd import _lib
        mod.init(_lib.HIP_LIB_PATH, False)
        return mod
    except Exception as e:                                        # noqa: BLE001 -- reported, and the ctypes route is the same library
        warnings.warn(f"exllamav2_ext drop-in: compiled binding unavailable ({e}); per-token calls take the ctypes route", RuntimeWarning)
        return None


_fast = _load_fast()

if _fast is not None:
    q_attn_forward_1 = _fast.q_attn_forward_1
    q_attn_forward_2 = _fast.q_attn_forward_2
    q_mlp_forward_ = _fast.q_mlp_forward_

    def free_q_attn(handle):
        _fast.forget_module(handle)
        _e.free_q_attn(handle)

    def free_q_mlp(handle):
        _fast.forget_module(handle)
        _e.free_q_mlp(handle)

    # entry points of the ctypes half that write activations through raw pointers: a hand-off published for that address is stale
    def gemm_half_q_half(a, b, c, force_cuda=False):
        _fast.note_write(c)
        return _e.gemm_half_q_half(a, b, c, force_cuda)

    def rms_norm(x, w, y, epsilon):
        _fast.note_write(y)
        return _e.rms_norm(x, w, y, epsilon)

    def rms_norm_(x, w, epsilon):
        _fast.note_write(x)
        return _e.rms_norm_(x, w, epsilon)

    def rope_(x, sin, cos, past_len, num_heads, head_dim, offsets, neox_style):
        _fast.note_write(x)
        return _e.rope_(x, sin, cos, past_len, num_heads, head_dim, offsets, neox_style)

    def q_moe_mlp_forward_(q_moe_mlp, x):
        _fast.note_write(x)
        return _e.q_moe_mlp_forward_(q_moe_mlp, x)
else:
    q_attn_forward_1 = _e.q_attn_forward_1
    q_attn_forward_2 = _e.q_attn_forward_2
    q_mlp_forward_ = _e.q_mlp_forward_
    free_q_attn = _e.free_q_attn
    free_q_mlp = _e.free_q_mlp
    gemm_half_q_half = _e.gemm_half_q_half
    rms_norm = _e.rms_norm
    rms_norm_ = _e.rms_norm_
    rope_ = _e.rope_
    q_moe_mlp_forward_ = _e.q_moe_mlp_forward_


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


# the reference's single-process tensor parallel (ext_tp.cpp, tp_* of ext_qattn.cpp / ext_qmlp.cpp): host staging through
# pinned buffers + per-device kernels, mirrored in exllamav2_amd/ext_tp.py
make_tp_context = _tp.make_tp_context
free_tp_context = _tp.free_tp_context
tp_broadcast = _tp.tp_broadcast
tp_gather = _tp.tp_gather
tp_cross_device_barrier = _tp.tp_cross_device_barrier
tp_all_reduce = _tp.tp_all_reduce
gemm_half_q_half_tp = _tp.gemm_half_q_half_tp
rms_norm_tp = _tp.rms_norm_tp
tp_mlp_forward_ = _tp.tp_mlp_forward_
tp_attn_forward_ = _tp.tp_attn_forward_
tp_attn_forward_paged_ = _tp.tp_attn_forward_paged_
make_q_moe_mlp = _e.make_q_moe_mlp
free_q_moe_mlp = _e.free_q_moe_mlp
fp16_to_fp8 = _e.fp16_to_fp8
fp8_to_fp16 = _e.fp8_to_fp16
cache_rotate = _e.cache_rotate
count_match = _e.count_match
matrix_fp16_to_q4 = _e.matrix_fp16_to_q4
matrix_q4_to_fp16 = _e.matrix_q4_to_fp16
for _n in ("layer_norm", "layer_norm_", "head_norm", "head_norm_", "softcap_", "gen_mrope_pos_ids", "gemm_half_half_half",
           "had_paley", "had_paley2", "pack_rows_4", "pack_columns", "quantize", "quantize_err", "quantize_range",
           "quantize_range_inplace", "sim_anneal", "logit_filter_exclusive",
           "dump_profile_results", "stloader_open_file", "stloader_close_file"):
    globals()[_n] = _out_of_scope(_n, "outside the quantized forward path (SURVEY.md 2.2: OUT OF SCOPE)")


# ---- load path: csrc/stloader.hip through the C ABI -------------------------------------------------
# stloader_read: host targets are read in place by 8 readers; device targets go through a ring of pinned slots with the
# host-to-device copies overlapping the reads.  tensor_remap / tensor_remap_4bit: the CPU column re-orderings of
# linear.py:156-158.
stloader_read = _e.stloader_read
tensor_remap = _e.tensor_remap
tensor_remap_4bit = _e.tensor_remap_4bit


def fast_fill_cpu_ones_bool(tensor) -> None:                          # ext_sampling.cpp: logit-filter helper of the sampler
    tensor.fill_(True)


def fast_fadd_cpu(a, b) -> None:
    a.add_(b)


def fast_copy_cpu(a, b) -> None:
    a.copy_(b)


def partial_strings_match(match, offsets, strings) -> int:
    """cpp/generator.cpp:12-55 (stop-string scan of the generators): `match` and `strings` are UTF-32 buffers, `offsets`
    the byte offsets of the strings.  Returns the position in `match` of the first full occurrence of a string (strings
    in order), -2 when a string matches up to the end of `match` (could still complete), -1 otherwise."""
    q = _np.frombuffer(match, dtype=_np.uint32)
    off = _np.frombuffer(offsets, dtype=_np.uint32)
    st = _np.frombuffer(strings, dtype=_np.uint32)
    q_len = q.size
    for i in range(off.size - 1):
        beg = int(off[i]) // 4
        s = st[beg:int(off[i + 1]) // 4]
        s_len = s.size
        for a0 in range(q_len):
            a, b = a0, 0
            while a < q_len and b < s_len and q[a] == s[b]:
                a += 1; b += 1
                if b == s_len:
                    return a0
                if a == q_len:
                    return -2
    return -1


def apply_rep_penalty(sequence, penalty_max, sustain, decay, alpha_frequency, alpha_presence, logits) -> None:
    """ext_sampling.cpp:32-72 over cpp/sampling.cpp:20-110 (host helper of the sampler, like fast_fadd_cpu): walks the last
    `sustain + decay` tokens of each sequence backwards; the first encounter of a token divides (positive logit) or
    multiplies its logit by the repetition penalty and subtracts the presence penalty, every encounter subtracts the
    frequency penalty; inside the decay range the three penalties move linearly towards 1 / 0 / 0 (fp32 steps).
    sequence int64 [bsz, seq_len], logits fp32 [bsz, 1, vocab] on the host, modified in place."""
    f32 = _np.float32
    seq = sequence.numpy()
    lg = logits.view(logits.shape[0], -1).numpy()
    vocab, seq_len = lg.shape[-1], seq.shape[-1]
    # Vectorised (the reference's default settings call this on every generated token over the whole context: a Python loop
    # per token made the drop-in's generators host-bound).  Same fp32 operations in the same order PER TOKEN ID as the walk:
    # the penalties in force at walk step w (w = 0 is the last token) are sequential fp32 sums, the first encounter of an id
    # applies repetition then presence, and the k-th encounters of all ids are subtracted together, k = 0, 1, ...
    sust0 = seq_len if sustain == -1 else int(sustain)
    beg = max(seq_len - sust0 - int(decay), 0)
    n = seq_len - beg
    if n <= 0:
        return
    rep0, freq0, pres0 = f32(penalty_max), f32(alpha_frequency), f32(alpha_presence)
    if decay:
        d = (f32(1.0) - rep0) / f32(decay), (f32(0.0) - freq0) / f32(decay), (f32(0.0) - pres0) / f32(decay)
    else:
        d = f32(0.0), f32(0.0), f32(0.0)
    n_upd = max(0, n - 1 - sust0)                              # updates applied before the last walk step
    def ramp(v0, dv):
        seq_v = _np.add.accumulate(_np.concatenate([_np.array([v0], dtype=f32), _np.full(n_upd, dv, dtype=f32)]), dtype=f32)
        w = _np.arange(n)
        return seq_v[_np.maximum(w - sust0, 0)]                # value in force at walk step w
    rep_w, freq_w, pres_w = ramp(rep0, d[0]), ramp(freq0, d[1]), ramp(pres0, d[2])
    any_freq = bool(_np.any(freq_w != 0))
    for b in range(seq.shape[0]):
        row = lg[b]
        toks = seq[b, beg:][::-1]                              # walk order
        steps = _np.nonzero((toks >= 0) & (toks < vocab))[0]
        if steps.size == 0:
            continue
        t = toks[steps].astype(_np.int64)
        order = _np.argsort(t, kind="stable")                  # groups of equal ids, walk order inside a group
        ts, ws = t[order], steps[order]
        first = _np.ones(ts.size, dtype=bool); first[1:] = ts[1:] != ts[:-1]
        ids, w1 = ts[first], ws[first]
        x = row[ids]
        x = _np.where(x > 0.0, x / rep_w[w1], x * rep_w[w1]).astype(f32)
        row[ids] = x - pres_w[w1]
        if any_freq:
            start = _np.nonzero(first)[0]
            rank = _np.arange(ts.size) - _np.repeat(start, _np.diff(_np.append(start, ts.size)))
            for k in range(int(rank.max()) + 1):
                sel = rank == k
                row[ts[sel]] = row[ts[sel]] - freq_w[ws[sel]]


def sample_basic(logits, temperature, top_k, top_p, top_a, min_p, tfs, typical, random, output_tokens, output_probs,
                 output_kprobs, output_ktokens, logit_filter, mirostat, mirostat_mu, mirostat_tau, mirostat_eta,
                 post_temperature, xtc_mask, xtc_probability, xtc_threshold, min_temp, max_temp, temp_exponent,
                 smoothing_factor, skew):
    """ext_sampling.cpp:93-301.  Greedy (top_k == 1 or temperature ~ 0: what ExLlamaV2Sampler.Settings.greedy() asks for and
    test_inference.py measures) is an arg-max under the logit filter.  Temperature / top-k (1..500) / top-p / min-p -- the
    reference's default Settings (sampler.py:54-69) -- run on the DEVICE sampler (csrc/sampling.hip: same candidates, same
    order, same fp32 threshold sums, same random recurrence as the CPU sampler).  Everything else of the sampler chain
    (top-a, tfs, typical, mirostat, XTC, skew, smoothing, dynamic temperature, the top-token report with sampling, top_k = 0
    or > 500) is outside the quantized forward path and raises."""
    greedy = top_k == 1 or temperature < 0.01
    has_filter = logit_filter is not None and logit_filter.device.type != "meta"
    if not greedy:
        unsupported = [n for n, on in (("top_a", top_a > 0.0), ("tfs", 0.0 < tfs < 1.0), ("typical", 0.0 < typical < 1.0),
                                       ("mirostat", bool(mirostat)), ("xtc", bool(xtc_probability and xtc_probability > 0.0)),
                                       ("post_temperature / dynamic temperature", post_temperature != 1.0 or max_temp > min_temp),
                                       ("smoothing_factor", smoothing_factor > 0.0), ("skew", skew != 0.0),
                                       ("return_top_tokens", output_ktokens is not None and output_ktokens.device.type != "meta"),
                                       ("top_k outside 1..500", not (1 <= top_k <= 500 and top_k < logits.shape[-1])))
                       if on]
        if unsupported:
            raise NotImplementedError("exllamav2_ext.sample_basic: " + ", ".join(unsupported) + " not built in this drop-in "
                                      "(device sampler: temperature, top_k 1..500, top_p, min_p, logit filter; SURVEY.md 2.2)")
        dev = _torch.device("cuda", _torch.cuda.current_device())
        x = logits.reshape(-1, logits.shape[-1]).to(dev, dtype=_torch.float32).contiguous()
        f = logit_filter.reshape(x.shape).to(dev).contiguous() if has_filter else None
        tok = _torch.empty((x.shape[0],), dtype=_torch.int32, device=dev)
        prob = _torch.empty((x.shape[0],), dtype=_torch.float32, device=dev)
        _e.sample_rows(x, temperature, top_k, top_p, min_p, random, tok, prob, logit_filter=f)
        output_tokens.copy_(tok.to(output_tokens.device, dtype=output_tokens.dtype).view(output_tokens.shape))
        output_probs.copy_(prob.to(output_probs.device).view(output_probs.shape))
        return []
    if mirostat or (xtc_probability and xtc_probability > 0.0):
        raise NotImplementedError("exllamav2_ext.sample_basic: mirostat / XTC are not built in this drop-in (SURVEY.md 2.2)")
    x = logits.reshape(-1, logits.shape[-1]).float()
    if has_filter:
        x = x.masked_fill(~logit_filter.reshape(x.shape).bool(), float("-inf"))
    tok = _torch.argmax(x, dim=-1)
    output_tokens.copy_(tok.view(output_tokens.shape))
    output_probs.fill_(1.0)
    if output_ktokens is not None and output_ktokens.device.type != "meta":
        k = output_ktokens.shape[-1]
        p = _torch.softmax(x, dim=-1)
        kp, kt = _torch.topk(p, k, dim=-1)
        output_ktokens.copy_(kt.view(output_ktokens.shape)); output_kprobs.copy_(kp.view(output_kprobs.shape))
    return []
