#!/usr/bin/env python3
"""Runs the UNMODIFIED reference host code (ExLlamaV2Config / ExLlamaV2 / ExLlamaV2Cache, the greedy loop of
test_inference.py:604-609) on top of the drop-in (dropin/exllamav2_ext.py -> libexl2_hip.so) and prints logits / tokens
as JSON for the caller to compare with the oracle.

  PYTHONPATH=dropin:<repo>:<dir holding the reference's exllamav2 package> python tools/run_reference_dropin.py <model_dir> <out.npz>

The reference package comes from /root/reference where that exists, or from the build-time mirror of its *.py files under
the git-ignored oracle/_ref/reference_py (oracle/ref_build/build.sh) on a machine without the reference tree."""
import os
import sys
import numpy as np
import torch


def main(model_dir: str, out: str, steps: int = 4, flash: bool = False):
    import exllamav2
    from exllamav2 import ExLlamaV2, ExLlamaV2Config, ExLlamaV2Cache
    from exllamav2.ext import ext_c
    assert ext_c.__name__ == "exllamav2_ext" and "dropin" in ext_c.__file__, ext_c.__file__
    config = ExLlamaV2Config(model_dir)
    config.max_seq_len = 256
    config.max_input_len = 32
    if not flash:
        config.no_flash_attn = True             # contiguous cache + the reference's own _attn_torch for this run ...
        config.no_sdpa = True                   # ... in its matmul form: the SDPA branch of v0.3.2 passes get_block_diag_mask()
                                                # == None as the mask when cu_seqlens is unset (attn.py:890-891), i.e. it is
                                                # NOT causal for q_len > 1 -- a reference bug off the flash-attn path
    # flash: what the reference does wherever flash-attn is importable (attn.py:1141-1142): _attn_flash -> flash_attn_func,
    # served by dropin/flash_attn (one launch of csrc/attn.hip per layer)
    model = ExLlamaV2(config)
    model.load()
    cache = ExLlamaV2Cache(model, max_seq_len=256)
    ids = torch.tensor([[3, 17, 42, 7]])
    # test_inference.py:604-609: logits = model.forward(ids[:, -1:], cache); sample = argmax; ids = cat
    logits = model.forward(ids, cache, last_id_only=False)
    all_logits = [logits.float().cpu().numpy()]
    toks = []
    for _ in range(steps):
        sample = torch.argmax(logits[0, -1]).cpu().unsqueeze(0).unsqueeze(0)
        toks.append(int(sample))
        ids = torch.cat((ids, sample), dim=-1)
        logits = model.forward(ids[:, -1:], cache)
        all_logits.append(logits.float().cpu().numpy())
    res = dict(prefill=all_logits[0], steps=np.concatenate(all_logits[1:], axis=1), tokens=np.array(toks))
    print("reference-on-dropin ok (contiguous cache, greedy loop%s):" % (", flash_attn_func shim" if flash else ""), toks)
    if flash:
        from exllamav2 import attn as ref_attn
        assert ref_attn.has_flash_attn and not config.no_flash_attn
        np.savez(out, **res)
        return

    # ---- ExLlamaV2DynamicGenerator, paged mode: attn.py:466-638 forward_paged -> flash_attn_with_kvcache (dropin/flash_attn ->
    # exl2_rope_kv_append + exl2_paged_attn), page table / defragmenter / prefix matching of dynamic.py untouched
    import os
    if os.path.exists(os.path.join(model_dir, "tokenizer.json")):
        from exllamav2 import ExLlamaV2Tokenizer
        from exllamav2.generator import ExLlamaV2DynamicGenerator, ExLlamaV2DynamicJob, ExLlamaV2Sampler
        config2 = ExLlamaV2Config(model_dir)
        config2.max_seq_len = 1024
        config2.max_input_len = 256
        model2 = ExLlamaV2(config2)
        model2.load()
        cache2 = ExLlamaV2Cache(model2, max_seq_len=1024)
        tokenizer = ExLlamaV2Tokenizer(config2)
        gen = ExLlamaV2DynamicGenerator(model=model2, cache=cache2, tokenizer=tokenizer, max_batch_size=4, max_chunk_size=256, paged=True)
        assert gen.paged
        prompts = [[3, 17, 42, 7], [5, 9, 77, 31, 100, 250]]
        jobs = []
        for pr in prompts:
            job = ExLlamaV2DynamicJob(input_ids=torch.tensor([pr]), max_new_tokens=steps, gen_settings=ExLlamaV2Sampler.Settings.greedy(),
                                      return_logits=True, stop_conditions=[], identifier=len(jobs))
            gen.enqueue(job); jobs.append(job)
        toks2 = {i: [] for i in range(len(prompts))}
        logits2 = {i: [] for i in range(len(prompts))}
        while gen.num_remaining_jobs():
            for r in gen.iterate():
                if r["stage"] == "streaming" and "token_ids" in r:
                    toks2[r["identifier"]] += r["token_ids"][0].tolist()
                    if "logits" in r: logits2[r["identifier"]].append(r["logits"].float().cpu().numpy())
        for i in range(len(prompts)):
            res[f"dyn_tokens_{i}"] = np.array(toks2[i])
            res[f"dyn_logits_{i}"] = np.concatenate(logits2[i], axis=1)
            res[f"dyn_prompt_{i}"] = np.array(prompts[i])
        print("reference-on-dropin ok (dynamic generator, paged):", toks2)
    np.savez(out, **res)


def main_tp(model_dir: str, out: str, steps: int = 4):
    """The reference's single-process tensor parallel: model.load_tp (model.py:354-473: every linear split by output columns,
    linear.py:540-619) + ExLlamaV2Cache_TP + the same greedy loop.  Attention goes through tp_attn_forward_ (attn.py:1198-
    1259), the MLP through tp_mlp_forward_ (mlp.py:365-399), norm / head through rms_norm_tp / gemm_half_q_half_tp, the
    exchanges through tp_broadcast / tp_gather and the pinned host buffers -- all served by exllamav2_amd/ext_tp.py.
    Split over every visible device (one on the GPU box of this build)."""
    # (load_tp slices q_weight after load(): the drop-in recognises that loader by the placeholder temp_dq it passes and keeps the
    # loaded layout -- no switch needed; dropin/exllamav2_ext.py)
    from exllamav2 import ExLlamaV2, ExLlamaV2Config, ExLlamaV2Cache, ExLlamaV2Cache_TP
    from exllamav2.ext import ext_c
    assert ext_c.__name__ == "exllamav2_ext" and "dropin" in ext_c.__file__, ext_c.__file__
    config = ExLlamaV2Config(model_dir)
    config.max_seq_len = 256
    config.max_input_len = 32
    model = ExLlamaV2(config)
    model.load_tp(gpu_split=[8.0] * torch.cuda.device_count(), progress=False)
    assert model.tp_context is not None
    cache = ExLlamaV2Cache_TP(model, base=ExLlamaV2Cache, max_seq_len=256)
    ids = torch.tensor([[3, 17, 42, 7]])
    logits = model.forward(ids, cache, last_id_only=False)
    all_logits = [logits.float().cpu().numpy()]
    toks = []
    for _ in range(steps):
        sample = torch.argmax(logits[0, -1]).cpu().unsqueeze(0).unsqueeze(0)
        toks.append(int(sample))
        ids = torch.cat((ids, sample), dim=-1)
        logits = model.forward(ids[:, -1:], cache)
        all_logits.append(logits.float().cpu().numpy())
    np.savez(out, prefill=all_logits[0], steps=np.concatenate(all_logits[1:], axis=1), tokens=np.array(toks),
             devices=np.array(model.tp_context.all_devs))
    print("reference-on-dropin ok (load_tp, ExLlamaV2Cache_TP, greedy loop):", toks, "devices", model.tp_context.all_devs)


def main_split(model_dir: str, out: str, steps: int = 4):
    """The reference's OWN layer split: one process, model.load(gpu_split=[...]) (model.py:176-263 set_device_map: modules are
    dealt to devices by a byte budget; :1012-1016 the hidden state hops with safe_move_tensor at every device change), the same
    greedy loop.  The budget of device 0 is bisected on set_device_map so that about half of the modules of this (small) model
    land on each of the first two devices."""
    from exllamav2 import ExLlamaV2, ExLlamaV2Config, ExLlamaV2Cache
    from exllamav2.ext import ext_c
    assert ext_c.__name__ == "exllamav2_ext" and "dropin" in ext_c.__file__, ext_c.__file__
    assert torch.cuda.device_count() >= 2, "needs two visible devices"
    config = ExLlamaV2Config(model_dir)
    config.max_seq_len = 256
    config.max_input_len = 32
    config.no_flash_attn = True
    config.no_sdpa = True
    model = ExLlamaV2(config)
    lo, hi = 0.0, 8.0
    for _ in range(48):
        mid = 0.5 * (lo + hi)
        try:
            model.set_device_map([mid, 8.0])
            idxs = [m.device_idx for m in model.modules if m.device_idx is not None and m.device_idx >= 0]
            on0 = sum(1 for i in idxs if i == 0)
        except AssertionError:
            on0 = 0
        if on0 * 2 < len(idxs): lo = mid
        else: hi = mid
    model.load(gpu_split=[hi, 8.0])
    devs = [m.device_idx for m in model.modules]
    used = sorted(set(d for d in devs if d is not None and d >= 0))
    assert used == [0, 1], devs
    cache = ExLlamaV2Cache(model, max_seq_len=256)
    ids = torch.tensor([[3, 17, 42, 7]])
    logits = model.forward(ids, cache, last_id_only=False)
    all_logits = [logits.float().cpu().numpy()]
    toks = []
    for _ in range(steps):
        sample = torch.argmax(logits[0, -1]).cpu().unsqueeze(0).unsqueeze(0)
        toks.append(int(sample))
        ids = torch.cat((ids, sample), dim=-1)
        logits = model.forward(ids[:, -1:], cache)
        all_logits.append(logits.float().cpu().numpy())
    np.savez(out, prefill=all_logits[0], steps=np.concatenate(all_logits[1:], axis=1), tokens=np.array(toks),
             module_devices=np.array([-2 if d is None else d for d in devs]))
    print("reference-on-dropin ok (load(gpu_split), layer split over devices", used, "):", toks)


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "tp":
        main_tp(sys.argv[1], sys.argv[2])
    elif len(sys.argv) > 3 and sys.argv[3] == "split":
        main_split(sys.argv[1], sys.argv[2])
    elif len(sys.argv) > 3 and sys.argv[3] == "flash":
        main(sys.argv[1], sys.argv[2], flash=True)
    else:
        main(sys.argv[1], sys.argv[2])
