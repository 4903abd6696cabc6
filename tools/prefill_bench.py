#!/usr/bin/env python3
"""Prefill-shaped q_gemm (dequantize-into-MFMA, qgemm_prefill.hip): TFLOP/s per Llama-2-7B linear at M = 8 x 2048 rows
(BASELINE config 3) and smaller chunks; one JSON line per case.  --model adds the whole-model prefill rate."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exllamav2_amd.ext import ext_c as ext, none_tensor
from exllamav2_amd.synth import synth_linear

MFMA_PEAK_F16 = 2500.0          # dense TFLOP/s, MI355X_MICROARCH.md


def _time_once(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def _interleaved(fns: dict, rounds: int, ramp_s: float = 1.0):
    """{name: [ms, ...]}: every variant once per round, `rounds` rounds, after a clock ramp on the first one -- a variant is never
    timed as a block of its own (round 4's file had the SAME kernel at 973 and 1031 TFLOP/s: run order and clock state, not code).
    The figure to quote is the MEDIAN; min is the clock's best case."""
    first = next(iter(fns.values()))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ramp_s:
        first()
    torch.cuda.synchronize()
    for fn in fns.values():
        fn()
    torch.cuda.synchronize()
    out = {k: [] for k in fns}
    for _ in range(rounds):
        for k, fn in fns.items():
            out[k].append(_time_once(fn))
    return out


def _stat(ms_list, flops):
    ms = sorted(ms_list)
    med, best = ms[len(ms) // 2], ms[0]
    return {"ms": round(med, 4), "ms_min": round(best, 4), "TFLOPs": round(flops / med / 1e9, 1), "TFLOPs_best": round(flops / best / 1e9, 1),
            "frac_mfma_peak": round(flops / med / 1e9 / MFMA_PEAK_F16, 4), "reps": len(ms)}


# kernel selection of csrc/qgemm_prefill.hip's host driver (read per call): the shipped choice and each tile height of the
# 256-column LDS-decode kernel (qgemm_mfma.hip)
VARIANTS = {"auto": {}, "tile256_mt8": {"EXL2_PREFILL_MT": "8"}, "tile256_mt4": {"EXL2_PREFILL_MT": "4"},
            "decode_in_gemm": {"EXL2_PREFILL_WPRE_MIN_ROWS": "0"},       # round 3: >= 2048 rows decode the weights once per call (auto)
            # round 6, 17-128 rows (qgemm_skinny.hip): forced K splits
            "nosplit": {"EXL2_SKINNY_SPLITK": "1"}, "split2": {"EXL2_SKINNY_SPLITK": "2"},
            "split4": {"EXL2_SKINNY_SPLITK": "4"}, "split8": {"EXL2_SKINNY_SPLITK": "8"}, "split16": {"EXL2_SKINNY_SPLITK": "16"}}


def bench_linear(k, n, m, recipe, reps=20, variants=("auto", "tile256_mt8", "tile256_mt4")):
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    w = synth_linear(k, n, recipe, "cuda", gen)
    h = ext.make_q_matrix_from_dict(w, none_tensor)
    a = torch.randn((m, k), device="cuda", dtype=torch.float16)
    c = torch.empty((m, n), device="cuda", dtype=torch.float16)
    wd = torch.empty((k, n), device="cuda", dtype=torch.float16)
    out = {"k": k, "n": n, "m": m, "recipe": str(recipe), "timing": f"interleaved rounds, {reps} per variant, median (and min)"}
    flops = 2.0 * m * k * n
    keys = ("EXL2_PREFILL_MT", "EXL2_PREFILL_MFMA_MIN_ROWS", "EXL2_PREFILL_WPRE_MIN_ROWS", "EXL2_SKINNY_SPLITK")

    def variant(name):
        def run():
            for key in keys: os.environ.pop(key, None)
            os.environ.update(VARIANTS[name])                # (the host driver reads its switches per call)
            ext.gemm_half_q_half(a, h, c)
        return run

    def lib():
        # the reference's method for M > 32: reconstruct + fp16 library GEMM (q_gemm.cu:243-263), here torch.matmul = hipBLASLt
        ext.reconstruct(h, wd); torch.matmul(a, wd, out=c)

    fns = {name: variant(name) for name in variants}
    fns["reconstruct_plus_hipblaslt"] = lib
    times = _interleaved(fns, reps)
    for key in keys: os.environ.pop(key, None)
    ref = None
    for name in variants:
        out[name] = _stat(times[name], flops)
        fns[name](); torch.cuda.synchronize()
        if ref is None: ref = c.clone()
        else: out[name]["max_abs_diff_vs_auto"] = float((c.float() - ref.float()).abs().max())
    for key in keys: os.environ.pop(key, None)
    out["reconstruct_plus_hipblaslt"] = _stat(times["reconstruct_plus_hipblaslt"], flops)
    lib(); torch.cuda.synchronize()
    out["auto_speedup_vs_reconstruct_gemm"] = round(out["reconstruct_plus_hipblaslt"]["ms"] / out[variants[0]]["ms"], 3)
    out["max_abs_diff_lib_vs_auto"] = float((c.float() - ref.float()).abs().max())
    ext.free_q_matrix(h)
    return out


def bench_model(batch=8, seq=2048, layers=None, reps=2):
    """test_inference.py -ps procedure (:533-579) at BASELINE config 3: forward(ids[batch, seq], preprocess_only=True)"""
    from exllamav2_amd.config import ExLlamaV2Config
    from exllamav2_amd.synth import synth_checkpoint
    from exllamav2_amd.model import ExLlamaV2
    from exllamav2_amd.cache import ExLlamaV2Cache
    cfg = ExLlamaV2Config.llama2_7b(max_seq_len=seq, max_input_len=2048, max_batch_size=batch)
    if layers: cfg.num_hidden_layers = layers
    ck = synth_checkpoint(cfg, "cuda", recipe="4.0bpw")
    model = ExLlamaV2(cfg, device="cuda").load(ck)
    cache = ExLlamaV2Cache(model, batch_size=batch, max_seq_len=seq)
    ids = torch.randint(0, cfg.vocab_size - 1, (batch, seq), generator=torch.Generator().manual_seed(0)).to("cuda")
    model.forward(ids, cache, preprocess_only=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        cache.current_seq_len = 0
        model.forward(ids, cache, preprocess_only=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return {"metric": "prefill tokens/s", "batch": batch, "seq": seq, "layers": cfg.num_hidden_layers,
            "value": round(batch * seq / dt, 1), "ms": round(dt * 1e3, 2)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--model", action="store_true")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--variants", default="auto,tile256_mt8,tile256_mt4", help="kernel selections to time (first = the baseline)")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--small", action="store_true", help="the 17-128-row kernel: the three 7B linears at 32 / 64 / 128 rows, K split off / auto / forced")
    args = ap.parse_args()
    if args.model:
        print(json.dumps(bench_model(layers=args.layers or None)), flush=True)
        sys.exit(0)
    r4 = ([4], [1.0], 128)
    cases = [(4096, 4096, 16384, ([5, 4], [0.1, 0.9], 128)), (4096, 11008, 16384, r4), (11008, 4096, 16384, ([8, 4], [0.05, 0.95], [32, 128])),
             (4096, 11008, 2048, r4), (4096, 11008, 256, r4), (4096, 11008, 64, r4)]
    if args.quick: cases = cases[1:2] + cases[3:4]
    if args.small:
        cases = [(k, n, m, rec) for m in (32, 64, 128) for (k, n, _, rec) in cases[:3]]
        if args.variants == ap.get_default("variants"): args.variants = "auto,nosplit,split2,split4,split8,split16"
    for k, n, m, rec in cases:
        print(json.dumps(bench_linear(k, n, m, rec, reps=args.reps, variants=tuple(args.variants.split(",")))), flush=True)
