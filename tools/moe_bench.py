#!/usr/bin/env python3
"""Fused MoE expert path at Mixtral-8x7B shape (BASELINE config 5: hidden 4096, inter 14336, 8 experts, top-2, EXL2 ~3.5 bpw):
time per q_moe_mlp_forward_ call and the weight bytes actually streamed (only routed experts), one JSON line per batch."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exllamav2_amd.ext import ext_c as ext, none_tensor
from exllamav2_amd.synth import synth_linear

def main():
    hidden, inter, E, topk, layers = 4096, 14336, 8, 2, 3
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    rec_up = ([4, 3], [0.5, 0.5], 128); rec_dn = ([4, 3], [0.6, 0.4], 128)
    mods, keep = [], []
    max_rows = 128
    ts = torch.empty((max_rows, hidden), device="cuda", dtype=torch.float16)
    ta = torch.empty((max_rows, inter), device="cuda", dtype=torch.float16)
    tb = torch.empty((max_rows, inter), device="cuda", dtype=torch.float16)
    bytes_expert = 0
    for l in range(layers):
        hs = {"w1": [], "w2": [], "w3": []}
        ip = None                                         # one act-order permutation for every expert's w1 / w3 (quantize.py:190-192)
        for e in range(E):
            for name, (k, n, rec) in (("w1", (hidden, inter, rec_up)), ("w3", (hidden, inter, rec_up)), ("w2", (inter, hidden, rec_dn))):
                w = synth_linear(k, n, rec, "cuda", gen, invperm=(ip if name != "w2" else None)); keep.append(w)
                if name == "w1" and ip is None: ip = w["q_invperm"]
                h = ext.make_q_matrix_from_dict(w, none_tensor); hs[name].append(h)
                if l == 0 and e == 0: bytes_expert += ext.q_matrix_info(h)["bytes"]
        norm = torch.ones((hidden,), device="cuda", dtype=torch.float16)
        gate = (torch.randn((E, hidden), device="cuda", generator=gen) * 0.05).half()
        tl = torch.empty((max_rows, E), device="cuda", dtype=torch.float16)
        keep += [norm, gate, tl]
        mods.append(ext.make_q_moe_mlp(norm, none_tensor, True, 1e-5, gate, E, topk, hs["w1"], hs["w2"], hs["w3"], ts, none_tensor,
                                       ta, tb, tl, none_tensor, max_rows, False))
    for rows in (1, 4, 16):
        x = torch.randn((rows, hidden), device="cuda", dtype=torch.float16, generator=gen)
        for m in mods: ext.q_moe_mlp_forward_(m, x.clone())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        xs = [x.clone() for _ in range(reps * layers)]
        e0.record()
        i = 0
        for _ in range(reps):
            for m in mods:
                ext.q_moe_mlp_forward_(m, xs[i]); i += 1
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * layers)
        # experts touched: expected distinct experts among rows * topk draws (upper bound E)
        print(json.dumps({"workload": "Mixtral-8x7B-shaped MoE MLP layer, EXL2 3.5bpw synthetic", "rows": rows, "us_per_layer": round(us, 1),
                          "expert_bytes": bytes_expert, "all_experts_bytes": bytes_expert * E,
                          "GBs_if_all_experts_streamed": round(bytes_expert * E / us / 1e3, 1),
                          "GBs_if_topk_only": round(bytes_expert * min(E, rows * topk) / us / 1e3, 1)}), flush=True)

if __name__ == "__main__":
    main()
