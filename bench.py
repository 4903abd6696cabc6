#!/usr/bin/env python3
"""Headline benchmark: greedy decode tokens/s of a synthetic Llama-2-7B EXL2 4.0bpw model (BASELINE.json configs[1]).

Procedure = the reference's `test_inference.py -s` (test_inference.py:584-618): forward(ids[:, -1:], cache); argmax;
append -- one "step" is one generated token; the whole step (embedding, 32 layers, head, greedy sampling, position
bookkeeping) runs on the device as one HIP graph.  Inputs (weights, cache, first token) are resident in HBM when the
timed region starts.  Rank 0 prints ONE JSON line.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--model llama2-7b|llama2-70b|mixtral-8x7b|tinyllama|tiny]
                  [--recipe 4.0bpw|3.5bpw|2.5bpw|gptq-4bit-128g] [--ctx C] [--batch B] [--cache fp16|q4] [--no-cpu-baseline]
                  [--no-prefill] [--no-parity-check] [--no-ctx-window] [--no-graph] [--parallel pipeline|tp]
  python bench.py --cpu-baseline-only [--model ...] [--recipe ...]     (what the main run starts as a child for `cpu_baseline`)

Beside the headline value the line carries `roofline` (q_gemm launches: algorithmic bytes / HIP-event time per launch against
8 TB/s), `parity_check` (device logits vs the oracle on the first two layers + head of the timed checkpoint, before timing),
`extra.ctx1920_tokens_per_s` (SURVEY.md 8d's second window), `prefill` (BASELINE configs[2]: 8 x 2048 tokens on the native
kernels) and `cpu_baseline` (the oracle "port" on the host cores: variant B unsampled fp32 GEMV pass, variant A the C port
on the packed tensors; run in a child process).

N > 1 (launched by torch.distributed.run, one rank per GPU): layer-split ("gpu_split", model.py:176-263) as a pipeline:
rank r owns layers [r L/N, (r+1) L/N); N independent sequences are in flight, one per stage, hidden states hop
rank -> rank over RCCL point-to-point; value = all sequences' tokens / time (weak scaling: per-GPU work per token fixed).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is what a float4 copy achieves


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=128)
    p.add_argument("--warmup", type=int, default=16)
    p.add_argument("--model", default="llama2-7b")
    p.add_argument("--recipe", default="4.0bpw")
    p.add_argument("--ctx", type=int, default=0, help="tokens already in the KV cache when timing starts")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-baseline-only", action="store_true",
                   help="print only the cpu_baseline object (how the main run obtains it: in a child process, so that a host "
                        "out-of-memory kill or a hang there cannot take the GPU line with it)")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-parity-check", action="store_true", help="skip the pre-timing oracle check of the device logits")
    p.add_argument("--no-ctx-window", action="store_true", help="skip the extra ctx-1920 decode window")
    p.add_argument("--cache", default="fp16", choices=["fp16", "q4"], help="KV cache type (q4: ExLlamaV2Cache_Q4)")
    p.add_argument("--no-prefill", action="store_true", help="skip the extra prefill measurement (BASELINE configs[2])")
    p.add_argument("--no-dropin", action="store_true",
                   help="skip the extra figure of the UNMODIFIED reference host on the drop-in (tools/dropin_decode_bench.py, child process)")
    p.add_argument("--batch", type=int, default=1, help="sequences decoded together (BASELINE configs[4]: 16)")
    p.add_argument("--parallel", default="pipeline", choices=["pipeline", "tp"],
                   help="N > 1: layer-split pipeline (default, weak scaling) or tensor parallel (column shards + all-gather, strong scaling)")
    p.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps (each after --warmup untimed steps from the same "
                                                           "cache position); the MEDIAN window is the reported value")
    p.add_argument("--headline-only", action="store_true", help="= --no-cpu-baseline --no-prefill --no-dropin --no-ctx-window (A/B runs)")
    a = p.parse_args()
    if a.headline_only:
        a.no_cpu_baseline = a.no_prefill = a.no_dropin = a.no_ctx_window = True
    return a


def make_cfg(name: str, max_seq_len: int):
    from exllamav2_amd.config import ExLlamaV2Config
    if name == "llama2-7b":
        return ExLlamaV2Config.llama2_7b(max_seq_len=max_seq_len, max_input_len=256)
    if name == "llama2-70b":
        return ExLlamaV2Config.llama2_70b(max_seq_len=max_seq_len, max_input_len=256)
    if name == "mixtral-8x7b":
        return ExLlamaV2Config.mixtral_8x7b(max_seq_len=max_seq_len, max_input_len=256, max_batch_size=16)
    if name == "tinyllama":
        return ExLlamaV2Config.tinyllama_1b(max_input_len=256)
    if name == "tiny":
        return ExLlamaV2Config.tiny_test(max_seq_len=max_seq_len)
    raise SystemExit(f"unknown model {name}")


def time_gemv_calls(model, dec, reps: int = 5):
    """Average duration of the q_gemm launches of one decode step.  The 129 launches (fused q|k|v, o, fused gate|up, down
    per layer + head) are captured alone into a HIP graph on the decoder's stream and replayed between two HIP events on
    that stream -- no host launch gaps in the figure, only the kernels and their boundaries (what rocprofv3's
    per-kernel average must agree with).  All 32 layers: weights are HBM-cold (3.4 GB >> 256 MB Infinity Cache), the
    rotation tests/test_gemv.py:84-128 relies on.  Returns (ms per step, launches, algorithmic bytes)."""
    import torch
    from exllamav2_amd.ext import none_tensor
    ext, cfg = model.ext, model.config
    b = dec.b
    x = dec.x
    q = model.temp_q[:b].view(b, 1, cfg.num_attention_heads, cfg.head_dim)
    k = model.temp_k[:b].view(b, 1, cfg.num_key_value_heads, cfg.head_dim)
    v = model.temp_v[:b].view(b, 1, cfg.num_key_value_heads, cfg.head_dim)
    ao = model.temp_attn[:b].view(b, 1, -1)

    ch = getattr(dec, "chain", None)

    def one_step_chain():
        # the q_gemm launches of the chained decode step (csrc/qgemv_flat.hip), with the decoder's own buffers
        launches, nbytes, npart = 0, 0, 1
        plan = ch["plan"]
        x2 = x.view(b, cfg.hidden_size)
        overlap = "flags" in ch                                  # EXL2_CHAIN_OVERLAP=1: the same two-stream hand-off as the step
        if overlap:
            ext.chain_overlap_begin(ch["flags"], dec.stream.cuda_stream, ch["stream_b"].cuda_stream)
        try:
            launches, nbytes = chain_launches(plan, x2, npart)
        finally:
            if overlap:
                ext.chain_overlap_end()
        return launches, nbytes

    def chain_launches(plan, x2, npart):
        launches, nbytes = 0, 0
        for i, (attn, mlp) in enumerate(model.layers):
            in_a, o_inv, in_m, nw_a, nw_m = plan[i]
            ext.q_attn_forward_1_chain(attn.q_handle, ch["xp_a"], ch["ss_a"], npart, b, q, k, v)
            npart = ext.q_attn_forward_2_chain(attn.q_handle, x2, ao, b, in_m, nw_m, ch["xp_b"], ch["ss_b"])
            nxt, nxt_w = (plan[i + 1][0], plan[i + 1][3]) if i + 1 < len(plan) else (ch["head_inv"], ch["norm_head"])
            npart = ext.q_mlp_forward_chain(mlp.q_handle, x2, ch["xp_b"], ch["ss_b"], npart, b, nxt, nxt_w, ch["xp_a"], ch["ss_a"])
            launches += 4
            nbytes += sum(l.weight_bytes() for l in (attn.q_proj, attn.k_proj, attn.v_proj, attn.o_proj,
                                                     mlp.gate_proj, mlp.up_proj, mlp.down_proj))
        ext.gemm_half_q_half_chain(ch["xp_a"], ch["ss_a"], npart, cfg.norm_eps, model.lm_head.q_handle, dec.logits, b)
        return launches + 1, nbytes + model.lm_head.weight_bytes()

    def one_step_modules():
        launches, nbytes = 0, 0
        for attn, mlp in model.layers:
            ext.q_attn_forward_1(attn.q_handle, x, b, 1, 0, none_tensor, q, k, v, model.sin, model.cos, apply_rope=False)
            ext.q_attn_forward_2(attn.q_handle, x, ao, b, 1)
            ext.q_mlp_forward_(mlp.q_handle, x)
            launches += 4
            nbytes += sum(l.weight_bytes() for l in (attn.q_proj, attn.k_proj, attn.v_proj, attn.o_proj,
                                                     mlp.gate_proj, mlp.up_proj, mlp.down_proj))
        ext.gemm_half_q_half(dec.xn.view(b, -1), model.lm_head.q_handle, dec.logits)
        return launches + 1, nbytes + model.lm_head.weight_bytes()

    one_step = one_step_chain if ch is not None else one_step_modules
    stream = dec.stream
    two = ch is not None and "flags" in ch                      # overlapped chain: two graphs, one per stream, side by side
    with torch.cuda.stream(stream):
        if two:
            sb = ch["stream_b"]
            dec._ordered(one_step)                              # warm-up, eager (stream B ordered behind / ahead of stream A)
            stream.wait_event(ch["ev_b"]); stream.synchronize(); sb.synchronize()
            launches, nbytes = 0, 0
            ext.graph_begin_capture(stream.cuda_stream); ext.graph_begin_capture(sb.cuda_stream)
            try:
                launches, nbytes = one_step()
            finally:
                graph = ext.graph_end_capture(stream.cuda_stream)
                graph_b = ext.graph_end_capture(sb.cuda_stream)
            replay = lambda: dec._ordered(lambda: (ext.graph_launch(graph, stream.cuda_stream), ext.graph_launch(graph_b, sb.cuda_stream)))
        else:
            launches, nbytes = one_step()                       # warm-up, eager
            stream.synchronize()
            ext.graph_begin_capture(stream.cuda_stream)
            try:
                one_step()
            finally:
                graph = ext.graph_end_capture(stream.cuda_stream)
            graph_b = None
            replay = lambda: ext.graph_launch(graph, stream.cuda_stream)
        replay()
        if two: stream.wait_event(ch["ev_b"])
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            replay()
        if two: stream.wait_event(ch["ev_b"])
        e1.record(stream)
        stream.synchronize()
        ext.graph_free(graph)
        if graph_b is not None: ext.graph_free(graph_b)
    return e0.elapsed_time(e1) / reps, launches, nbytes


def oracle_for_parity(cfg, ck, layers: int = 2):
    """CHECKER, not product: the numpy oracle (oracle/model.py) over the first `layers` layers + final norm + head of the
    very checkpoint the bench decodes with.  Must be built BEFORE model.load() re-lays q_weight out in place."""
    import copy
    from oracle.model import OracleModel
    ocfg = copy.copy(cfg)
    ocfg.num_hidden_layers = layers
    keep = {k: v for k, v in ck.items()
            if not k.startswith("model.layers.") or int(k.split(".")[2]) < layers}
    return OracleModel(ocfg, keep)


def parity_check(model, oracle, device, n_decode: int = 6, cache_type: str = "fp16", batch: int = 1):
    """Before anything is timed, on the first layers + head of the very checkpoint the bench times: a 4-token prompt through
    `model.forward` (the prefill route), then `n_decode` greedy steps through **GreedyGraphDecoder -- the captured chain the
    timed region replays** (same kernels, same graph mechanism; `route` in the result says which decode route it took).
    Every step's device logits (`dec.logits`) must match the oracle within the model-level fp16 tolerance of
    tests/test_model.py (0.03 + |x| 2^-8); the device's own greedy token must be the oracle's wherever the oracle's
    top-1 / top-2 margin exceeds 4x that tolerance, and at least 3 steps must be that confident (the synthetic head is
    structured for it: exllamav2_amd/synth.py).

    cache_type "q4" (configs[3]): the check runs THROUGH an ExLlamaV2Cache_Q4 -- the cache type the timed region uses -- against
    OracleModel.forward(q4_cache=True) (the reference's cache.py:472-556 semantics); after every step the oracle adopts the
    device's codes (a 4-bit quantizer is discontinuous: oracle/model.py:q4_adopt) and the fraction of codes that differed from
    the oracle's own is bounded.

    `batch` = the number of sequences the timed region decodes together (configs[4]: 16 -- a sparse-MoE layer then takes its
    grouped-expert route, a dense one the kernels' many-row forms): the check decodes that many DIFFERENT sequences together."""
    import numpy as np
    import torch
    from exllamav2_amd import ExLlamaV2Cache, GreedyGraphDecoder
    layers = oracle.cfg.num_hidden_layers
    full = model.layers
    model.layers = full[:layers]
    dec = None
    q4 = cache_type == "q4"
    worst_flip = 0.0
    try:
        if q4:
            from exllamav2_amd.cache import ExLlamaV2Cache_Q4
            cache = ExLlamaV2Cache_Q4(model, batch_size=batch, max_seq_len=256)
            cache.key_scales, cache.value_scales = cache.key_scales[:layers], cache.value_scales[:layers]
        else:
            cache = ExLlamaV2Cache(model, batch_size=batch, max_seq_len=256)
        cache.key_states, cache.value_states = cache.key_states[:layers], cache.value_states[:layers]

        def follow_codes(n_tokens, what):
            nonlocal worst_flip
            if not q4:
                return
            torch.cuda.synchronize()
            for layer in range(layers):
                f = oracle.q4_adopt(layer, cache.key_states[layer].cpu().numpy(), cache.key_scales[layer].cpu().numpy(),
                                    cache.value_states[layer].cpu().numpy(), cache.value_scales[layer].cpu().numpy(), n_tokens)
                worst_flip = max(worst_flip, f)
                if f > 0.01:
                    raise SystemExit(f"[bench] parity check FAILED at {what}: {f:.4f} of the Q4 cache codes of layer {layer} differ from the oracle's")
        ids = (np.array([[1, 15043, 3186, 29892]]) + 977 * np.arange(batch)[:, None]) % model.config.vocab_size    # a row per sequence
        oracle.reset(batch)
        worst, checked, tok_checked = 0.0, 0, 0

        near_tie = 0

        def compare(got, want, what):
            """rows whose expert selection in the oracle was a near tie (sparse-MoE models; oracle/model.py: router_margin) are
            not compared -- top-k is discontinuous, a router logit one fp16 ulp off selects another expert there -- and counted"""
            nonlocal worst, checked, near_tie
            ok = oracle.router_margin > 2e-3
            near_tie += int((~ok).sum())
            err = np.abs(got - want)[ok]
            # fp16 noise in the hidden state reaches EVERY logit of a row in proportion to the row's scale, not to the single logit's
            # value: a row whose logits reach 46 (sequence 2 of the --batch >= 3 prompts on the 7B synthetic weights) shows +-0.25 on
            # logits of any size -- the chained and the module-by-module route deviate from the oracle in OPPOSITE directions there
            # (0.23 / 0.28, i.e. 0.5-0.6 % of the row's scale; 0.51 apart: tools/debug/batch_parity_debug.py,
            # profiles/history/r05q_batch_parity_debug.txt).  Hence the third term: 2^-7 of what the row's largest |logit| exceeds 8 by -- rows
            # inside the range the bar of tests/test_model.py was set on (|logit| <= 8: every other row of every configuration) keep it.
            # MEASURED yardstick for that term (round 6; tests/golden/reference_model_yardstick.json: "outlier_rows_7b", made by
            # tests/golden/make_golden_model_yardstick.py --outlier): the reference's OWN decode kernels (gemm_half_q_half_kernel +
            # rms_norm_kernel + act_mul_kernel + rope, executed on the host over the same 7B-wide 2-layer model and prompts) sit 0.39
            # from this oracle on their outlier row (|logit| 38.8: 11.7 x the two-term tolerance; this bar there: 0.42) and 0.022-0.033
            # on ordinary rows (0.5-0.96 x) -- the device measures 0.23-0.28 and 0.003-0.004.  The term is the reference's own distance.
            row_scale = np.maximum(np.abs(want).max(axis=-1, keepdims=True) - 8.0, 0.0)
            tol = (0.03 + np.abs(want) * 2.0 ** -8 + row_scale * 2.0 ** -7)[ok]
            if err.size:
                worst = max(worst, float((err / tol).max()))
            checked += err.size
            if not np.all(err <= tol):
                raise SystemExit(f"[bench] parity check FAILED at {what}: max |logit - oracle| = {err.max():.4f}")
            return ok

        want = oracle.forward(ids, q4_cache=q4)[:, -1]
        got = model.forward(torch.from_numpy(ids), cache).float().cpu().numpy()[:, -1].astype(np.float64)
        compare(got, want, "the prompt")
        follow_codes(ids.shape[1], "the prompt")
        tok = want.argmax(-1).astype(np.int64)                   # [batch]
        dec = GreedyGraphDecoder(model, cache, batch_size=batch).capture()
        route = "chain (qgemv chain kernels, one HIP graph per step)" if dec.chain is not None else "module by module"
        dec.reset(torch.from_numpy(tok), ids.shape[1])
        for step in range(n_decode):
            dec.run(1)
            torch.cuda.synchronize()
            want = oracle.forward(tok[:, None], q4_cache=q4)[:, -1]
            got = dec.logits.float().cpu().numpy()[:, :model.config.vocab_size].astype(np.float64)
            ok = compare(got, want, f"decode step {step}")
            follow_codes(ids.shape[1] + step + 1, f"decode step {step}")
            dev_tok = dec.tokens(ids.shape[1] + step, 1).cpu().numpy()[:, 0].astype(np.int64)
            top2 = np.sort(want, axis=-1)[:, -2:]
            conf = ((top2[:, 1] - top2[:, 0]) > 0.12) & ok
            if conf[0]:
                tok_checked += 1
            bad = np.nonzero(conf & (dev_tok != want.argmax(-1)))[0]
            if len(bad):
                r = int(bad[0])
                raise SystemExit(f"[bench] parity check FAILED at decode step {step}, sequence {r}: greedy token {int(dev_tok[r])} != "
                                 f"oracle {int(want[r].argmax())}")
            tok = dev_tok                                    # follow the device: each step is checked alone
        if tok_checked < 3:
            raise SystemExit(f"[bench] parity check FAILED: only {tok_checked} of {n_decode} steps had a confident oracle margin")
        if near_tie * 10 > batch * (1 + n_decode):
            raise SystemExit(f"[bench] parity check FAILED: {near_tie} of {batch * (1 + n_decode)} rows skipped as router near-ties")
        del cache
    finally:
        if dec is not None:
            dec.free()
        model.layers = full
    res = {"layers": layers, "sequences": batch, "steps": 1 + n_decode, "decode_route": route, "cache": cache_type, "logits_checked": checked,
           "worst_err_over_tol": round(worst, 3), "confident_tokens_equal": tok_checked,
           "tolerance": "0.03 + |x| * 2^-8 (fp16, tests/test_model.py) + max(0, max_row|x| - 8) * 2^-7 (rows of outlier scale: bench.py parity_check)"}
    if getattr(model.config, "num_experts", 0):
        res["rows_skipped_router_near_tie"] = near_tie
    if q4:
        res["q4_codes_differing_from_oracle_max_frac"] = round(worst_flip, 5)
        res["oracle"] = "OracleModel.forward(q4_cache=True): cache.py:472-556 semantics, following the device's codes step by step"
    return res


def prefill_parity_check(model, oracle, ids, tail: int = 128):
    """The prefill leg's kernels against the oracle: the SAME [batch, seq] call shape the leg times (so the same row counts reach
    the same dequantize-into-MFMA / flash-prefill variants) through the first layers + head; last-position logits of EVERY
    sequence (their rows sit in different M-tiles of the same launches, and each sequence attends over its own seq K/V rows).
    To keep the float64 oracle affordable the check's sequences share their first seq - `tail` tokens: the oracle runs the
    common prefix once and every sequence's own tail on top of it (causal attention: exactly what the device computed for that
    sequence); the device gets no such help -- it runs all batch x seq rows."""
    import numpy as np
    import torch
    from exllamav2_amd import ExLlamaV2Cache
    layers = oracle.cfg.num_hidden_layers
    full = model.layers
    model.layers = full[:layers]
    try:
        b, s = ids.shape
        tail = min(tail, s - 1)
        ids = ids.clone()
        ids[1:, :s - tail] = ids[0, :s - tail]
        cache = ExLlamaV2Cache(model, batch_size=b, max_seq_len=s)
        cache.key_states, cache.value_states = cache.key_states[:layers], cache.value_states[:layers]
        got = model.forward(ids, cache, last_id_only=True).float().cpu().numpy()[:, -1].astype(np.float64)       # [b, vocab]
        want = np.zeros_like(got)
        ids_np = ids.cpu().numpy()
        oracle.reset(1)
        oracle.forward(ids_np[:1, :s - tail])                # the shared prefix: fills the oracle's K/V rows [0, s - tail)
        for i in range(b):
            oracle.seq_len = s - tail                         # (rows >= s - tail are overwritten by this sequence's own tail)
            want[i] = oracle.forward(ids_np[i:i + 1, s - tail:])[0, -1]
        err = np.abs(got - want)
        tol = 0.03 + np.abs(want) * 2.0 ** -8
        if not np.all(err <= tol):
            bad = sorted(set(np.nonzero(err > tol)[0].tolist()))
            raise RuntimeError(f"prefill parity check FAILED: max |logit - oracle| = {err.max():.4f} in sequences {bad} (not timed)")
        del cache
    finally:
        model.layers = full
    top2 = np.sort(want, axis=-1)[:, -2:]
    conf = (top2[:, 1] - top2[:, 0]) > 0.12
    return {"layers": layers, "rows": int(ids.numel()), "sequences_checked": int(b), "logits_checked": int(want.size),
            "worst_err_over_tol": round(float((err / tol).max()), 3),
            "tokens_equal_where_confident": f"{int((got.argmax(-1) == want.argmax(-1))[conf].sum())} of {int(conf.sum())}",
            "tolerance": "0.03 + |x| * 2^-8"}


def pmc_traffic_gb(launches_per_step, kernel_prefix: str):
    """HBM GB per decode step fetched by the q_gemm launches, from the COMMITTED PMC pass of this same command
    (profiles/*_pmc_summary.json: separate `rocprofv3 --pmc FETCH_SIZE` run, KB per launch; x2 = the gfx950 correction for
    wide coalesced reads, MI355X_MICROARCH.md section HBM).  Counters cannot be read from inside this process, so the
    figure is labelled with the file it comes from.  (None, None) when no profile of this kernel is committed."""
    try:
        # the round's own pass of THIS command only (rNN_pmc_summary[_graph|_nograph].json), newest round first, the pass over the
        # replayed graph before the one over eager launches
        import re
        files = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d\d_pmc_summary(_graph|_nograph)?\.json", f)),
                       key=lambda f: (f[:3], {"_graph": 2, "": 1, "_nograph": 0}[re.fullmatch(r"r\d\d_pmc_summary(_graph|_nograph)?\.json", f).group(1) or ""]))
        for f in reversed(files):
            d = json.load(open(os.path.join(ROOT, "profiles", f)))
            # every instantiation of the kernel (geometries of one template), weighted by its launches
            hits = [(v["avg"], v.get("launches", 1)) for k, v in d.items() if k.startswith("FETCH_SIZE:") and kernel_prefix in k]
            if hits:
                kb = sum(a * n for a, n in hits) / sum(n for _, n in hits)
                return round(kb * 1024 * 2 * launches_per_step / 1e9, 3), "committed profile profiles/" + f
    except Exception:
        pass
    return None, None


def profile_frac(alg_bytes_per_launch: float):
    """roofline.frac_profile: the same fraction from the COMMITTED rocprofv3 --kernel-trace --stats summary of this command
    (profiles/rNN_kernel_stats.csv, newest round): launch-weighted average duration of the qgemv_* kernels there.  The run's own
    `frac` (HIP events, this box) and this figure (rocprofv3, the builder's box) side by side make the two boxes visible."""
    import csv
    try:
        import re
        files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d\d_kernel_stats\.csv", f))   # (the headline loop's pass only)
        if not files:
            return None, None
        rows = [r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", files[-1]))) if "qgemv_" in r["Name"]]
        calls = sum(int(r["Calls"]) for r in rows)
        avg_s = sum(float(r["TotalDurationNs"]) for r in rows) / calls * 1e-9
        return round(alg_bytes_per_launch / avg_s / 1e9 / HBM_PEAK_GBS, 4), f"profiles/{files[-1]} ({avg_s * 1e6:.2f} us per launch)"
    except Exception:
        return None, None


def cpu_baseline(cfg, recipe: str, seed: int = 0):
    """The oracle ("port": the reference has no CPU path) timed on the host cores: matmul(x,
    reconstruct) with pre-dequantized fp32 weights (compute-fair variant B), bounded sample = ONE transformer layer's 7
    linears x 8 tokens, extrapolated to layers + head."""
    import numpy as np
    import torch
    from exllamav2_amd.synth import synth_linear, synth_linear_gptq, RECIPES, GPTQ_RECIPES
    from oracle import exl2 as OX
    gptq = recipe in GPTQ_RECIPES
    rec = RECIPES["4.0bpw"] if gptq else RECIPES[recipe]
    gen = torch.Generator(device="cpu"); gen.manual_seed(seed)
    h, inter = cfg.hidden_size, cfg.intermediate_size
    qd, kvd = cfg.num_attention_heads * cfg.head_dim, cfg.num_key_value_heads * cfg.head_dim
    shapes = [("q_proj", h, qd), ("k_proj", h, kvd), ("v_proj", h, kvd), ("o_proj", qd, h),
              ("gate_proj", h, inter), ("up_proj", h, inter), ("down_proj", inter, h)]
    ws, ts = [], []
    for name, k, n in shapes:
        if gptq:
            w = synth_linear_gptq(k, n, GPTQ_RECIPES[recipe], "cpu", gen)
            ws.append(torch.from_numpy(OX.gptq_reconstruct({kk: vv.numpy() for kk, vv in w.items()}).astype(np.float32)))
            continue
        w = synth_linear(k, n, rec[name], "cpu", gen)
        t = {kk: vv.numpy() for kk, vv in w.items() if kk != "q_perm"}
        ts.append(t)
        ws.append(torch.from_numpy(OX.exl2_reconstruct(t).astype(np.float32)))
    # weights as [N, K] rows so that a token is 7 row-major GEMVs (torch.mv): every thread streams whole rows.  The thread
    # count is part of the port: try the machine's hardware threads, half of them (one per core with SMT) and two smaller
    # pools, keep the fastest (an M = 1 product is bound by DRAM bandwidth and fork/join cost, not by core count).
    wts = [w.t().contiguous() for w in ws]
    del ws
    tokens = 8
    ncpu = os.cpu_count() or 1
    layer_elems = sum(k * n for _, k, n in shapes)
    full_gb = 4 * (cfg.num_hidden_layers * layer_elems + h * cfg.vocab_size) / 1e9
    head_scale = (h * cfg.vocab_size) / layer_elems
    silu = torch.nn.functional.silu

    # mode "blas": torch.mv with the BLAS library's own threading.  mode "rows": every matrix cut into one row block per
    # worker, each worker a single-threaded torch.mv on its block (a token = two fork/joins per layer: q|k|v|gate|up, then
    # o|down) -- on many-core hosts the library's M = 1 threading is the bottleneck, not DRAM.  Both are timed on a one-layer
    # probe over a few pool sizes; the faster one runs the full pass.
    def layer_blas(lw, x):
        q = torch.mv(lw[0], x); torch.mv(lw[1], x); torch.mv(lw[2], x); o = torch.mv(lw[3], q)
        g = torch.mv(lw[4], x); u = torch.mv(lw[5], x); d = torch.mv(lw[6], silu(g) * u)
        return x + 1e-3 * (o + d)

    class RowPool:
        def __init__(self, T):
            from concurrent.futures import ThreadPoolExecutor
            self.T, self.pool = T, ThreadPoolExecutor(T)
        def cut(self, w, own=False):
            """row blocks of one [N, K] matrix, one per worker; own=True: each worker makes its own copy (first touch on its node)"""
            n = w.shape[0]
            bounds = [(i * n // self.T, (i + 1) * n // self.T) for i in range(self.T)]
            if not own:
                return [w[a:b] for a, b in bounds]
            return list(self.pool.map(lambda ab: w[ab[0]:ab[1]].clone(), bounds))
        def stage(self, blocks_list, vec_list, outs):
            """outs[m][rows of worker i] = blocks_list[m][i] @ vec_list[m] for every matrix m, one task per worker"""
            def work(i):
                for blocks, vec, out in zip(blocks_list, vec_list, outs):
                    n = out.shape[0]
                    torch.mv(blocks[i], vec, out=out[i * n // self.T:(i + 1) * n // self.T])
            list(self.pool.map(work, range(self.T)))
        def layer(self, lb, x, bufs):
            self.stage([lb[0], lb[1], lb[2], lb[4], lb[5]], [x] * 5, [bufs[0], bufs[1], bufs[2], bufs[4], bufs[5]])
            act = silu(bufs[4]) * bufs[5]
            self.stage([lb[3], lb[6]], [bufs[0], act], [bufs[3], bufs[6]])
            return x + 1e-3 * (bufs[3] + bufs[6])
        def close(self):
            self.pool.shutdown()

    bufs = [torch.empty(w.shape[0]) for w in wts]
    best = None                                             # (seconds per layer-token, mode, threads)
    for threads in sorted({ncpu, max(1, ncpu // 2), min(ncpu, 64), min(ncpu, 32)}, reverse=True):
        torch.set_num_threads(threads)
        x = torch.randn(h)
        for rep in range(tokens + 1):                      # first pass = warm-up (thread pool start, page faults)
            if rep == 1: t0 = time.perf_counter()
            x = layer_blas(wts, x)
        dt = (time.perf_counter() - t0) / tokens
        if best is None or dt < best[0]: best = (dt, "blas", threads)
    torch.set_num_threads(1)
    for threads in sorted({min(ncpu, 128), min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}, reverse=True):
        rp = RowPool(threads)
        lb = [rp.cut(w) for w in wts]
        x = torch.randn(h)
        for rep in range(tokens + 1):
            if rep == 1: t0 = time.perf_counter()
            x = rp.layer(lb, x, bufs)
        dt = (time.perf_counter() - t0) / tokens
        rp.close()
        if dt < best[0] or (os.environ.get("EXL2_CPU_BASELINE_MODE") == "rows" and best[1] == "blas"): best = (dt, "rows", threads)
    dt, mode, threads = best
    how = ("torch.mv with the BLAS library's threading" if mode == "blas" else
           "one row block per worker thread, single-threaded torch.mv on each block")
    per_token = dt * (cfg.num_hidden_layers + head_scale)
    out = {"value": round(1.0 / per_token, 4), "unit": "tokens/s", "cores": threads, "kind": "port",
           "sample": f"variant B (BASELINE.md 3): 1 of {cfg.num_hidden_layers} layers (7 linears, pre-dequantized fp32 [N, K] rows, "
                     f"{how}, best of 8 threading set-ups) x {tokens} tokens, extrapolated to {cfg.num_hidden_layers} layers + head; "
                     f"a full pass streams {full_gb:.1f} GB of fp32 per token"}
    # the UNSAMPLED pass (SURVEY.md 8d variant 2: "Llama-2-7B ~ 26 GB in fp32 fits"): every layer gets its own copy of the
    # layer's fp32 matrices (distinct memory, so a token streams the full weight set from DRAM instead of re-reading one
    # layer), plus the head; a warm-up token and 3 timed ones with the set-up found above.  Only when the host has the
    # memory to spare; otherwise the extrapolated figure above stands and says so.
    try:
        import psutil
        avail = psutil.virtual_memory().available
        for f in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):    # a container's own limit
            try:
                lim = open(f).read().strip()
                if lim.isdigit():
                    avail = min(avail, int(lim))
            except OSError:
                pass
    except Exception:
        avail = 0
    if avail > 2.5 * full_gb * 1e9 and full_gb < 200:
        try:
            head = torch.randn(cfg.vocab_size, h) * 0.02
            x = torch.randn(h)
            n_timed = 3
            if mode == "blas":
                torch.set_num_threads(threads)
                layers = [wts] + [[w.clone() for w in wts] for _ in range(cfg.num_hidden_layers - 1)]
                for rep in range(n_timed + 1):
                    if rep == 1: t0 = time.perf_counter()
                    for lw in layers:
                        x = layer_blas(lw, x)
                    x = x + 1e-6 * torch.mv(head, x)[:h]
            else:
                torch.set_num_threads(1)
                rp = RowPool(threads)
                layers = [[rp.cut(w, own=True) for w in wts] for _ in range(cfg.num_hidden_layers)]
                hb, hout = rp.cut(head, own=True), torch.empty(cfg.vocab_size)
                for rep in range(n_timed + 1):
                    if rep == 1: t0 = time.perf_counter()
                    for lb in layers:
                        x = rp.layer(lb, x, bufs)
                    rp.stage([hb], [x], [hout])
                    x = x + 1e-6 * hout[:h]
                rp.close()
            dt_full = (time.perf_counter() - t0) / n_timed
            out["sampled_extrapolated"] = {"value": out["value"], "sample": out["sample"]}
            out["value"] = round(1.0 / dt_full, 4)
            out["sample"] = (f"variant B (BASELINE.md 3), UNSAMPLED: {n_timed} tokens (after 1 warm-up) through all {cfg.num_hidden_layers} layers "
                             f"+ head, {full_gb:.1f} GB of pre-dequantized fp32 weights in distinct memory per token (each layer holds "
                             f"its own copy of one synthesized layer's 7 matrices), [N, K] rows, {how}, {threads} threads "
                             f"(fastest of 8 threading set-ups on a 1-layer probe); {full_gb / dt_full:.0f} GB/s of host DRAM")
            del layers, head
        except Exception as e:                             # never lose the baseline to an allocation failure
            out["full_pass_error"] = str(e)[:160]
    # variant A (dequantize on the fly: what a CPU port of the q_gemm path itself does -- it reads the PACKED weights, the same
    # 3.46 GB per token the GPU kernels stream): oracle/cpu_qgemv.c, the multi-threaded C restatement of the EXL2 decode GEMV
    # (AVX2 + FMA, fp32 accumulation, a persistent thread pool), UNSAMPLED like variant B: every layer its own copy of the
    # packed tensors + the head, a warm-up token and 3 timed ones, pool size = fastest of four on a one-layer probe.
    if not gptq:
        try:
            out["variant_a"] = cpu_variant_a(cfg, rec, ts, gen, ncpu)
        except Exception as e:
            out["variant_a"] = {"value": None, "error": str(e)[:200]}
    return out


def cpu_variant_a(cfg, rec, ts, gen, ncpu: int):
    import numpy as np
    from exllamav2_amd.synth import synth_linear
    from oracle.cpu_qgemv import Matrix, Pool
    h, L = cfg.hidden_size, cfg.num_hidden_layers
    mats = [Matrix(t) for t in ts]                                  # q, k, v, o, gate, up, down of the synthesized layer
    hw = synth_linear(h, cfg.vocab_size, rec["lm_head"], "cpu", gen)
    head = Matrix({kk: vv.numpy() for kk, vv in hw.items() if kk != "q_perm"})
    n_cap = max([m.n for m in mats] + [head.n])

    def silu(v):
        return v / (1.0 + np.exp(-v))

    def layer(pool, lm, x):
        q = pool.gemv(lm[0], x); pool.gemv(lm[1], x); pool.gemv(lm[2], x); o = pool.gemv(lm[3], q)
        g = pool.gemv(lm[4], x); u = pool.gemv(lm[5], x); d = pool.gemv(lm[6], (silu(g) * u).astype(np.float32))
        return (x + np.float32(1e-3) * (o + d)).astype(np.float32)

    best = None
    for threads in sorted({min(ncpu, 128), min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}, reverse=True):
        pool = Pool(threads, n_cap)
        x = np.random.default_rng(0).standard_normal(h).astype(np.float32)
        for rep in range(5):
            if rep == 1: t0 = time.perf_counter()
            x = layer(pool, mats, x)
        dt = (time.perf_counter() - t0) / 4
        pool.close()
        if best is None or dt < best[0]: best = (dt, threads)
    threads = best[1]
    layers = [mats] + [[m.clone() for m in mats] for _ in range(L - 1)]
    packed_gb = (sum(m.nbytes() for lm in layers for m in lm) + head.nbytes()) / 1e9
    pool = Pool(threads, n_cap)
    x = np.random.default_rng(1).standard_normal(h).astype(np.float32)
    n_timed = 3
    for rep in range(n_timed + 1):
        if rep == 1: t0 = time.perf_counter()
        for lm in layers:
            x = layer(pool, lm, x)
        x = (x + np.float32(1e-6) * pool.gemv(head, x)[:h]).astype(np.float32)
    dt = (time.perf_counter() - t0) / n_timed
    pool.close()
    return {"value": round(1.0 / dt, 4), "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"variant A (BASELINE.md 3), UNSAMPLED: {n_timed} tokens (after 1 warm-up) through all {L} layers + head, "
                      f"dequantize-on-the-fly EXL2 GEMV in C (oracle/cpu_qgemv.c: AVX2 + FMA, fp32, {threads} pool threads = fastest "
                      f"of 4 sizes on a 1-layer probe) over {packed_gb:.2f} GB of packed weights + decoded scales in distinct memory "
                      f"per token; {packed_gb / dt:.0f} GB/s of host DRAM"}


def cpu_baseline_in_child(args, timeout_s: int = 420):
    """Runs `bench.py --cpu-baseline-only` as a child process and returns its JSON object: the baseline allocates tens of GB
    of host memory and runs for tens of seconds -- a kill or a hang there must not cost the GPU line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--model", args.model, "--recipe", args.recipe]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"value": None, "error": f"child rc={r.returncode}: {(r.stderr or '').strip()[-160:]}"}
    except subprocess.TimeoutExpired:
        return {"value": None, "error": f"child exceeded {timeout_s} s"}
    except Exception as e:
        return {"value": None, "error": str(e)[:200]}


def dropin_rate_in_child(timeout_s: int = 300):
    """SURVEY.md 8(d)(i) defines the headline metric by the reference's own loop (test_inference.py:584-618: model.forward(ids[:,
    -1:], cache) + host argmax per token).  That loop, run by the UNMODIFIED reference host code on dropin/exllamav2_ext.py ->
    libexl2_hip.so over a synthetic 7B model directory, in a child process (tools/dropin_decode_bench.py; it writes ~3.7 GB to
    /tmp first).  Informational beside the GreedyGraphDecoder figure: never the headline value."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "dropin_decode_bench.py"), "--tokens", "128", "--attn", "flash"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            d = json.loads(lines[-1])
            if "value" in d:
                return {"tokens_per_s": d["value"], "ms_per_token": d["ms_per_token"], "tokens": d["tokens"], "attention": d["attention"],
                        "loop": "unmodified reference host (ExLlamaV2 / ExLlamaV2Cache / test_inference.py -s loop) on dropin/exllamav2_ext.py",
                        "binding": d.get("binding"),
                        "route": "compiled binding + module chain behind the operator boundary (dropin/_exl2_fast.cpp): the lean decode "
                                 "kernels, 6 launches per layer (q|k|v, RoPE, attention, o, gate|up, down), eager launches from the "
                                 "reference's Python loop; no whole-step graph (DESIGN.md section 3.5)"}
            return d
        return {"error": f"child rc={r.returncode}: {(r.stderr or '').strip()[-200:]}"}
    except subprocess.TimeoutExpired:
        return {"error": f"child exceeded {timeout_s} s"}
    except Exception as e:
        return {"error": str(e)[:200]}


def prefill_rate(model_name: str, recipe: str, device: str, batch: int = 8, seq: int = 2048, parity: bool = True):
    """BASELINE configs[2] beside the headline: test_inference.py -ps procedure (:533-579), forward(ids[8, 2048],
    preprocess_only=True) on a fresh synthetic model, all layers, the product route (row pre-pass + dequantize-into-MFMA
    GEMMs + flash-prefill attention: no library GEMM)."""
    import torch
    from exllamav2_amd import ExLlamaV2, ExLlamaV2Cache
    from exllamav2_amd.config import ExLlamaV2Config
    from exllamav2_amd.synth import synth_checkpoint
    out = {"workload": f"{model_name} EXL2 {recipe}, {batch} x {seq} tokens, forward(preprocess_only=True)", "unit": "tokens/s"}
    cfg = ExLlamaV2Config.llama2_7b(max_seq_len=seq, max_input_len=2048, max_batch_size=batch)
    ck = synth_checkpoint(cfg, device, recipe=recipe, seed=1)
    oracle = oracle_for_parity(cfg, ck) if parity else None               # checker; before load() re-lays q_weight out
    model = ExLlamaV2(cfg, device=device).load(ck)
    ids = torch.randint(0, cfg.vocab_size - 1, (batch, seq), generator=torch.Generator().manual_seed(0)).to(device)
    if parity:
        t0 = time.perf_counter()
        out["parity_check"] = prefill_parity_check(model, oracle, ids)
        out["parity_check"]["seconds"] = round(time.perf_counter() - t0, 1)
        del oracle
    cache = ExLlamaV2Cache(model, batch_size=batch, max_seq_len=seq)
    model.forward(ids, cache, preprocess_only=True); torch.cuda.synchronize()
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        cache.current_seq_len = 0
        model.forward(ids, cache, preprocess_only=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    flops = 0.0
    for attn, mlp in model.layers:
        for lin in (attn.q_proj, attn.k_proj, attn.v_proj, attn.o_proj, mlp.gate_proj, mlp.up_proj, mlp.down_proj):
            flops += 2.0 * batch * seq * lin.in_features * lin.out_features
    out["native"] = {"value": round(batch * seq / best, 1), "ms": round(best * 1e3, 2), "layers": cfg.num_hidden_layers,
                     "linear_TFLOPs": round(flops / best / 1e12, 1)}
    # roofline of the dominant prefill kernel (qgemm_mfma_kernel, csrc/qgemm_mfma.hip): the largest linear of a layer alone, the
    # same 16384-row pass the forward above makes, timed with HIP events on the launch stream; dense fp16 MFMA peak 2.5 PFLOP/s
    # (MI355X_MICROARCH.md).  MFMA-busy counters of the CURRENT kernel source: profiles/r06_pmc_prefill_summary.json (58.8 %; the round-3 source: 61 %).
    try:
        lin = model.layers[0][1].gate_proj
        rows = batch * seq
        x = torch.randn((rows, lin.in_features), dtype=torch.float32, device=device).half()      # (sigma 1, like tools/prefill_bench.py)
        y = torch.empty((rows, lin.out_features), dtype=torch.float16, device=device)
        model.ext.gemm_half_q_half(x, lin.q_handle, y); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            model.ext.gemm_half_q_half(x, lin.q_handle, y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * rows * lin.in_features * lin.out_features
        out["roofline"] = {"bound": "mfma", "kernel": "qgemm_mfma_kernel (gate_proj %d x %d at %d rows: row pre-pass + GEMM launches of one call)" % (lin.in_features, lin.out_features, rows),
                           "achieved": round(fl / (ms * 1e-3) / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                           "frac": round(fl / (ms * 1e-3) / 2.5e15, 4), "traffic": None, "flops_per_call": fl, "avg_call_us": round(ms * 1e3, 1),
                           "pmc_source": "profiles/r06_pmc_prefill_summary.json (round-6 pass over the current source: SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES) = 58.8 % for the 256-row instantiation at 16384 rows, 56.0 % for the 128-row one; waves parked 35 %, issue-stalled 47 %, LDS bank conflicts 0)"}
        del x, y
    except Exception as e:  # informational
        out["roofline"] = {"error": str(e)[:200]}
    model.unload(); del model, cache
    torch.cuda.empty_cache()
    return out


def spawn_ranks(n: int) -> None:
    """Re-executes this command under `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` on 127.0.0.1 (one rank per
    GPU over RCCL), after checking that n GPUs are there; the child's exit code is ours.  Fails loudly -- exit code 2 and one line
    on stderr -- when fewer than n devices are visible."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n:
        print(f"[bench] --gpus {n}: only {have} GPU(s) visible on this node -- not running (a line with n_gpus < {n} would be "
              f"mistaken for the {n}-GPU figure)", file=sys.stderr)
        raise SystemExit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC only on this driver (RCCL peer buffers)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env, cwd=ROOT))


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(make_cfg(args.model, 2048), args.recipe)))
        return
    import torch
    import torch.distributed as dist

    if args.gpus < 1:
        raise SystemExit("[bench] --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` (the driver's N = 1 form with a larger N): start the N ranks ourselves, one per GPU, exactly as
        # the driver's own launcher line does -- never fall through to a one-GPU run that would print an n_gpus = 1 line
        return spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a figure "
                         f"for a GPU count that was not asked for")
    visible = torch.cuda.device_count()
    if visible < int(os.environ.get("LOCAL_WORLD_SIZE", world)) or local_rank >= visible:
        raise SystemExit(f"[bench] rank {rank}: {visible} GPU(s) visible, local rank {local_rank} of "
                         f"{os.environ.get('LOCAL_WORLD_SIZE', world)} needs its own -- one process per GPU, no sharing")
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"[bench] communicator has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    n_gpus = world
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(device)

    from exllamav2_amd import ExLlamaV2, ExLlamaV2Cache, GreedyGraphDecoder
    from exllamav2_amd.synth import synth_checkpoint

    max_seq = max(2048, ((args.ctx + args.steps + args.warmup + 1 + 255) // 256) * 256)
    cfg = make_cfg(args.model, max_seq)

    if n_gpus > 1 and args.parallel == "tp":
        from exllamav2_amd.tensor_p import run_tp_bench
        result = run_tp_bench(cfg, args, rank, world, device)
    elif n_gpus > 1:
        from exllamav2_amd.pipeline import run_layer_split_bench
        result = run_layer_split_bench(cfg, args, rank, world, device)
        if os.environ.get("EXL2_BENCH_TP_LINE", "1") != "0" and not getattr(cfg, "num_experts", 0):
            # the second labelled figure of an N > 1 run: ONE sequence decoded by all ranks together (tensor parallel, strong
            # scaling) -- what a "bs=1 on N GPUs" reader expects; nested in the same JSON line (the contract is one line)
            try:
                from exllamav2_amd.tensor_p import run_tp_bench
                torch.cuda.empty_cache()
                tp = run_tp_bench(cfg, args, rank, world, device)
                wb = tp.get("weight_bytes_per_rank")
                result["strong_scaling_tp"] = {
                    "metric": f"decode tokens/s, {args.model} {fmt_name(args.recipe)}, bs={args.batch} greedy, ONE sequence over {world} ranks",
                    "value": round(tp["value"], 2), "unit": "tokens/s", "ms_per_step": round(tp["ms_per_step"], 4), "scaling": "strong",
                    "parallelism": tp["parallelism"], "ranks": world, "weight_bytes_per_rank": wb,
                    "per_gpu_weight_roofline_frac": ([round(b / (tp["ms_per_step"] * 1e-3) / 8.0e12, 4) for b in wb] if isinstance(wb, list) else None)}
            except Exception as e:                                   # informational; never lose the headline line
                result["strong_scaling_tp"] = {"error": str(e)[:300]}
    else:
        t_load = time.perf_counter()
        ck = synth_checkpoint(cfg, device, recipe=args.recipe, seed=0)
        do_parity = not args.no_parity_check
        moe = bool(getattr(cfg, "num_experts", 0))           # (a sparse-MoE layer = 3 x num_experts matrices to reconstruct on the host: one layer)
        oracle = oracle_for_parity(cfg, ck, layers=1 if moe else 2) if do_parity else None    # checker; before load() re-lays q_weight out
        t_load = time.perf_counter()
        model = ExLlamaV2(cfg, device=device).load(ck)
        torch.cuda.synchronize()
        t_load = time.perf_counter() - t_load
        parity = parity_check(model, oracle, device, cache_type=args.cache, batch=args.batch) if do_parity else None
        del oracle
        if args.cache == "q4":
            from exllamav2_amd.cache import ExLlamaV2Cache_Q4
            cache = ExLlamaV2Cache_Q4(model, batch_size=args.batch, max_seq_len=max_seq)
        else:
            cache = ExLlamaV2Cache(model, batch_size=args.batch, max_seq_len=max_seq)
        dec = GreedyGraphDecoder(model, cache, batch_size=args.batch)
        if not args.no_graph:
            dec.capture()
        # clock ramp: a GPU that has been idle (fresh box) runs its first few hundred milliseconds ~15 % slow; this untimed
        # stretch is part of set-up, the W warm-up steps below are still run and not timed
        dec.reset(torch.tensor([1] * args.batch), 0)
        t_ramp = time.perf_counter()
        # (EXL2_BENCH_RAMP_S: the counter pass of tools/gpu_run.sh sets 0 -- every graph replay is serialised kernel by kernel there)
        while time.perf_counter() - t_ramp < float(os.environ.get("EXL2_BENCH_RAMP_S", "4.0")):
            dec.reset(torch.tensor([1] * args.batch), 0)
            dec.run(64, use_graph=not args.no_graph)
            torch.cuda.synchronize()
        # `--windows` identical timed windows (default 5), each = W untimed warm-up steps then EXACTLY K timed steps between two
        # device synchronisations, from the same cache position; the reported value is the MEDIAN window (20 steps are 28 ms of
        # work: one window is inside box noise), every window's rate is printed beside it
        window_dt = []
        for _ in range(max(1, args.windows)):
            dec.reset(torch.tensor([1] * args.batch), args.ctx)     # KV of the first `ctx` positions = resident (zeros)
            dec.run(args.warmup, use_graph=not args.no_graph)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dec.run(args.steps, use_graph=not args.no_graph)
            torch.cuda.synchronize()
            window_dt.append(time.perf_counter() - t0)
        dt = sorted(window_dt)[len(window_dt) // 2]
        windows = {"n": len(window_dt), "tokens_per_s": [round(args.batch * args.steps / d, 1) for d in window_dt],
                   "reported": "median", "spread_pct": round(100.0 * (max(window_dt) - min(window_dt)) / dt, 2)}
        toks = dec.tokens(args.ctx + args.warmup, args.steps)
        assert int(dec.cache_seqlens[0]) == args.ctx + args.warmup + args.steps
        assert int(toks.min()) >= 0 and int(toks.max()) < cfg.vocab_size

        if getattr(cfg, "num_experts", 0) or args.batch != 1:
            # MoE / batched runs: headline rate only (the q_gemm roofline figure is defined on configs[1])
            result = {"value": args.batch * args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "load_s": t_load, "windows": windows}
            if parity is not None: result["parity_check"] = parity
            dec.free()
            return finish(args, cfg, result, rank, world, n_gpus, device, dist)
        model.ext.chain_route_counts(reset=True)
        gemv_ms, launches, gemv_bytes = time_gemv_calls(model, dec)
        n_lean, n_flat = model.ext.chain_route_counts(reset=True)
        kv_bytes = 2 * cfg.num_hidden_layers * cfg.num_key_value_heads * cfg.head_dim * 2 * (args.ctx + args.warmup + args.steps // 2)
        achieved = gemv_bytes / (gemv_ms * 1e-3) / 1e9
        # the committed PMC pass is of the headline configuration only
        chained = getattr(dec, "chain", None) is not None
        kname = ("qgemv_lean_kernel<false" if n_lean >= n_flat else "qgemv_flat_kernel<false") if chained else "qgemv_stream_kernel<false, 4"
        traffic_gb, traffic_src = pmc_traffic_gb(launches, kname) if (args.model == "llama2-7b" and args.recipe == "4.0bpw") else (None, None)
        fp_frac, fp_src = profile_frac(gemv_bytes / launches) if (args.model == "llama2-7b" and args.recipe == "4.0bpw") else (None, None)
        extra = {}
        if chained:
            extra["chain_route_launches"] = {"qgemv_lean_kernel": int(n_lean), "qgemv_flat_kernel": int(n_flat)}
        if dec.chain is not None and "flags" in dec.chain:
            # overlapped chain (EXL2_CHAIN_OVERLAP=1): waits that gave up would make the timing meaningless -- must be 0
            words = [0] + [32 * (1 + c) for c in range(8)] + [320 + 32 * c for c in range(8)] + [576 + 32 * c for c in range(8)]
            extra["chain_overlap"] = {"hand_off_words_left_set": int((dec.chain["flags"][:, words] != 0).sum()),
                                      "waits_given_up": int(dec.chain["flags"][:, 2].sum()),
                                      "given_up_by_launch": {int(i): int(v) for i, v in enumerate(dec.chain["flags"][:, 2].tolist()) if v}}
        if args.model == "llama2-7b" and args.ctx == 0 and not args.no_ctx_window:
            # SURVEY.md 8d's second window: the same decode with 1920 tokens already in the cache (steps 1921..1984)
            dec.reset(torch.tensor([1] * args.batch), 1920)
            dec.run(8, use_graph=not args.no_graph); torch.cuda.synchronize()
            t1 = time.perf_counter()
            dec.run(64, use_graph=not args.no_graph); torch.cuda.synchronize()
            extra["ctx1920_tokens_per_s"] = round(64 / (time.perf_counter() - t1), 2)
        result = {
            "value": args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "load_s": t_load, "windows": windows,
            "weight_bytes_per_rank": [int(model.weight_bytes())],
            "roofline": {
                "bound": "hbm", "kernel": (kname.split("<")[0] if chained else "qgemv_stream_kernel<false, MB>") + " (all q_gemm launches of a decode step: fused q|k|v, o, fused gate|up, down per layer + head)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None if traffic_gb is None else round(traffic_gb * 1e9 / launches),
                "traffic_unit": "HBM bytes per q_gemm launch (PMC FETCH_SIZE x2)", "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(gemv_bytes / launches),
                "traffic_per_step_GB": traffic_gb, "bytes_per_step": gemv_bytes, "launches_per_step": launches,
                "avg_launch_us": round(gemv_ms * 1e3 / launches, 2),
                "frac_profile": fp_frac, "frac_profile_source": fp_src,
                "step_frac_of_weight_roofline": round((gemv_bytes + kv_bytes) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
            },
        }
        if extra: result["extra"] = extra
        dec.free()
        if parity is not None:
            result["parity_check"] = parity

    return finish(args, cfg, result, rank, world, n_gpus, device, dist)


def _metric_name(args, n_gpus: int, result: dict) -> str:
    """says what ran: one GPU = the BASELINE metric; N > 1 layer split = aggregate over the sequences in flight (weak scaling),
    N > 1 tensor parallel = one sequence (strong scaling)"""
    model = "Llama-2-7B EXL2 4.0bpw" if (args.model == "llama2-7b" and args.recipe == "4.0bpw") else f"{args.model} {fmt_name(args.recipe)}"
    if n_gpus == 1:
        return f"decode tokens/s, {model}, bs={args.batch} greedy"
    if result.get("scaling") == "strong":
        return f"decode tokens/s, {model}, bs={args.batch} greedy, ONE sequence over {n_gpus} GPUs (tensor parallel, strong scaling)"
    return (f"decode tokens/s AGGREGATE, {model}, greedy, {result.get('sequences_in_flight', n_gpus)} bs=1 sequences in flight over a "
            f"{n_gpus}-GPU layer split (weak scaling: per-GPU weight bytes per token constant)")


def fmt_name(recipe: str) -> str:
    return ("GPTQ " + recipe[5:]) if recipe.startswith("gptq-") else ("EXL2 " + recipe)


def finish(args, cfg, result, rank, world, n_gpus, device, dist):
    if rank == 0:
        out = {
            "metric": _metric_name(args, n_gpus, result),
            "value": round(result["value"], 2), "unit": "tokens/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(result["ms_per_step"], 4), "higher_is_better": True,
            "scaling": result.get("scaling", "weak"),
            # the reference's own published figure for THIS model/metric (README.md:71, RTX 4090)
            "vs_baseline": round(result["value"] / 211.0, 3) if (args.model == "llama2-7b" and n_gpus == 1 and args.batch == 1) else None,
            "baseline_ref": "211 tokens/s, Llama2 7B EXL2 4.0bpw, RTX 4090 (reference README.md:71)",
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{args.model} {fmt_name(args.recipe)} (synthetic weights, act-order), greedy decode, "
                                   f"bs={args.batch}, ctx {args.ctx}+{args.warmup}..+{args.steps}, {args.cache.upper()} KV cache, "
                                   f"whole step in one HIP graph",
                       "parallelism": "single GPU" if n_gpus == 1 else
                                      result.get("parallelism", f"layer-split pipeline x{n_gpus}, {n_gpus} sequences in flight")},
        }
        for k in ("roofline", "windows", "load_s", "weight_bytes_per_rank", "per_gpu_weight_roofline_frac", "sequences_in_flight", "ranks",
                  "strong_scaling_tp", "parity_check", "extra"):
            if k in result: out[k] = result[k]
        if n_gpus > 1 and args.model == "llama2-7b" and args.recipe == "4.0bpw" and args.batch == 1 and result.get("scaling", "weak") == "weak":
            # what DESIGN.md section 7 expects of this line, stated BEFORE it was ever measured (one-GPU boxes only): the first scaling run
            # tests a prediction.  Aggregate = N x (one GPU's rate) x efficiency; one GPU: profiles/r09_bench.json (721.8 tok/s)
            eff = {2: 0.93, 4: 0.84, 8: 0.72}.get(n_gpus)
            if eff is not None:
                out["predicted"] = {"tokens_per_s": round(n_gpus * 721.8 * eff, 1), "efficiency": eff, "one_gpu_tokens_per_s": 721.8,
                                    "basis": "DESIGN.md section 7 (arithmetic: per-tick hand-off of one hidden row over xGMI against the "
                                             "stage's compute; never measured on N > 1 GPUs before this run)",
                                    "measured_over_predicted": round(result["value"] / (n_gpus * 721.8 * eff), 3)}
        if not args.no_prefill and n_gpus == 1 and args.model == "llama2-7b" and args.batch == 1:
            try:
                out["prefill"] = prefill_rate(args.model, args.recipe, device, parity=not args.no_parity_check)
            except Exception as e:  # informational; never lose the headline number
                out["prefill"] = {"error": str(e)[:200]}
        if (not args.no_dropin and n_gpus == 1 and args.model == "llama2-7b" and args.recipe == "4.0bpw" and args.batch == 1
                and args.cache == "fp16"):
            d = dropin_rate_in_child()
            if "tokens_per_s" in d:
                d["fraction_of_headline"] = round(d["tokens_per_s"] / result["value"], 3)
            out.setdefault("extra", {})["reference_host_on_dropin"] = d
        if not args.no_cpu_baseline and n_gpus == 1:
            out["cpu_baseline"] = cpu_baseline_in_child(args)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()



if __name__ == "__main__":
    main()
