#!/usr/bin/env python3
"""Decode tokens/s of the UNMODIFIED reference host (ExLlamaV2 / ExLlamaV2Cache / the `test_inference.py -s` loop,
/root/reference/test_inference.py:584-618: model.forward(ids[:, -1:], cache) + host argmax + torch.cat per token) running on
the drop-in (dropin/exllamav2_ext.py -> libexl2_hip.so), on a synthetic Llama-2-7B EXL2 4.0bpw MODEL DIRECTORY written to
local disk first -- SURVEY.md 8(d)(i)'s definition of the headline metric, beside bench.py's GreedyGraphDecoder figure.

  python tools/dropin_decode_bench.py [--tokens 128] [--layers 32] [--dir /tmp/synth7b]

Prints one JSON line.  The reference package comes from /root/reference, or from the build-time mirror of its *.py files
(oracle/_ref/reference_py) on a machine without it.  This is a measurement of the drop-in, not a parity test (parity of this
very route: tests/test_dropin_reference.py)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--tokens", type=int, default=128)
    p.add_argument("--layers", type=int, default=32)
    p.add_argument("--dir", default="/tmp/synth7b")
    p.add_argument("--recipe", default="4.0bpw")
    p.add_argument("--attn", default="flash", choices=["flash", "torch"],
                   help="flash: the reference's _attn_flash -> dropin/flash_attn.flash_attn_func (what it does wherever flash-attn is "
                        "importable); torch: its own _attn_torch fallback (matmul branch)")
    p.add_argument("--profile", action="store_true", help="cProfile the timed loop and print the top host-side entries to stderr")
    p.add_argument("--generator", type=int, default=0, metavar="JOBS",
                   help="instead of the -s loop: ExLlamaV2DynamicGenerator (paged) with JOBS concurrent greedy jobs of --tokens new tokens each; "
                        "aggregate tokens/s of the generation phase")
    args = p.parse_args()
    ref = next((d for d in ("/root/reference", os.path.join(ROOT, "oracle", "_ref", "reference_py"))
                if os.path.isfile(os.path.join(d, "exllamav2", "model.py"))), None)
    if ref is None:
        print(json.dumps({"error": "no reference package (neither /root/reference nor oracle/_ref/reference_py)"}))
        return
    sys.path[:0] = [os.path.join(ROOT, "dropin"), ROOT, ref]
    import torch
    from exllamav2_amd.config import ExLlamaV2Config as OurCfg
    from exllamav2_amd.synth import synth_checkpoint
    from exllamav2_amd.synth_dir import write_model_dir
    t0 = time.perf_counter()
    cfg = OurCfg.llama2_7b(max_seq_len=2048)
    cfg.num_hidden_layers = args.layers
    if not os.path.exists(os.path.join(args.dir, "model.safetensors")):
        ck = synth_checkpoint(cfg, "cpu", recipe=args.recipe, seed=0)
        write_model_dir(args.dir, cfg, ck)
        del ck
    if args.generator and not os.path.exists(os.path.join(args.dir, "tokenizer.json")):
        from exllamav2_amd.synth_dir import write_tokenizer
        write_tokenizer(args.dir, cfg.vocab_size)
    t_write = time.perf_counter() - t0

    from exllamav2 import ExLlamaV2, ExLlamaV2Config, ExLlamaV2Cache
    from exllamav2.ext import ext_c
    assert ext_c.__name__ == "exllamav2_ext" and "dropin" in ext_c.__file__, ext_c.__file__
    t0 = time.perf_counter()
    config = ExLlamaV2Config(args.dir)
    config.max_seq_len = 2048
    if args.attn == "torch":
        config.no_flash_attn = True        # the reference's own _attn_torch fallback (attn.py:869-937) ...
        config.no_sdpa = True              # ... its matmul branch: the SDPA branch of v0.3.2 is not causal when cu_seqlens is unset
    model = ExLlamaV2(config)
    model.load()
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0
    if args.generator:
        return generator_mode(args, model, config, ext_c, t_write, t_load)
    cache = ExLlamaV2Cache(model, max_seq_len=2048)
    ids = torch.tensor([[1, 15043, 3186, 29892]])
    # test_inference.py:590-618 (-s): prompt through forward(preprocess_only), then the timed per-token loop
    model.forward(ids[:, :-1], cache, preprocess_only=True)
    for _ in range(8):                     # untimed warm-up tokens (clock ramp, lazy set-up)
        logits = model.forward(ids[:, -1:], cache)
        sample = torch.argmax(logits[0, -1]).cpu().unsqueeze(0).unsqueeze(0)
        ids = torch.cat((ids, sample), dim=-1)
    torch.cuda.synchronize()
    fast = getattr(ext_c, "_fast", None)
    if fast is not None: fast.stats(True)
    prof = None
    if args.profile:
        import cProfile
        prof = cProfile.Profile(); prof.enable()
    t0 = time.perf_counter()
    for _ in range(args.tokens):
        logits = model.forward(ids[:, -1:], cache)
        sample = torch.argmax(logits[0, -1]).cpu().unsqueeze(0).unsqueeze(0)
        ids = torch.cat((ids, sample), dim=-1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(28)
    print(json.dumps({"metric": "decode tokens/s, unmodified reference host on the drop-in (test_inference.py -s loop)",
                      "value": round(args.tokens / dt, 2), "unit": "tokens/s", "ms_per_token": round(dt / args.tokens * 1e3, 3),
                      "tokens": args.tokens, "layers": args.layers, "recipe": args.recipe, "attention": "reference _attn_flash -> dropin/flash_attn.flash_attn_func (csrc/attn.hip)" if args.attn == "flash" else "reference _attn_torch (matmul branch)",
                      "binding": ("compiled (dropin/_exl2_fast.so); per timed token: " + json.dumps({k: (round(v / args.tokens, 2) if isinstance(v, int) and not isinstance(v, bool) and k != "max_rows" and k != "known_successors" else v)
                                                                                                          for k, v in fast.stats().items()})) if fast is not None else "ctypes (exllamav2_amd/ext.py)",
                      "write_dir_s": round(t_write, 1), "load_s": round(t_load, 1), "last_tokens": ids[0, -4:].tolist()}))


def generator_mode(args, model, config, ext_c, t_write, t_load):
    """The reference's serving loop: ExLlamaV2DynamicGenerator.iterate() (generator/dynamic.py) in paged mode -- forward_paged
    (attn.py:466-638) -> q_attn_forward_1 / flash_attn_with_kvcache / q_attn_forward_2 / q_mlp_forward_ with batch = live jobs."""
    import torch
    from exllamav2 import ExLlamaV2Cache, ExLlamaV2Tokenizer
    from exllamav2.generator import ExLlamaV2DynamicGenerator, ExLlamaV2DynamicJob, ExLlamaV2Sampler
    jobs_n, new = args.generator, args.tokens
    cache = ExLlamaV2Cache(model, max_seq_len=max(4096, jobs_n * 512))
    tokenizer = ExLlamaV2Tokenizer(config)
    gen = ExLlamaV2DynamicGenerator(model=model, cache=cache, tokenizer=tokenizer, max_batch_size=jobs_n, max_chunk_size=2048, paged=True)
    assert gen.paged
    def enqueue(n_new):
        for i in range(jobs_n):
            prompt = [1 + i, 15043 + i, 3186, 29892 + 3 * i][: 3 + (i & 1)]
            gen.enqueue(ExLlamaV2DynamicJob(input_ids=torch.tensor([prompt]), max_new_tokens=n_new, stop_conditions=[],
                                            gen_settings=ExLlamaV2Sampler.Settings.greedy(), identifier=i))
    def drain():
        n = 0
        while gen.num_remaining_jobs():
            for r in gen.iterate():
                if r["stage"] == "streaming" and "token_ids" in r:
                    n += r["token_ids"].shape[-1]
        return n
    enqueue(8); drain()                                          # warm-up (lazy set-up, clock ramp, the binding learns the module order)
    torch.cuda.synchronize()
    fast = getattr(ext_c, "_fast", None)
    if fast is not None: fast.stats(True)
    enqueue(new)
    t0 = time.perf_counter()
    n = drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "aggregate generated tokens/s, unmodified reference ExLlamaV2DynamicGenerator (paged) on the drop-in, prompt processing included",
                      "value": round(n / dt, 2), "unit": "tokens/s", "jobs": jobs_n, "new_tokens_per_job": new, "tokens": n, "seconds": round(dt, 3),
                      "layers": args.layers, "recipe": args.recipe,
                      "binding": ("compiled (dropin/_exl2_fast.so): " + json.dumps(fast.stats())) if fast is not None else "ctypes (exllamav2_amd/ext.py)",
                      "write_dir_s": round(t_write, 1), "load_s": round(t_load, 1)}))


if __name__ == "__main__":
    main()
