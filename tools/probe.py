#!/usr/bin/env python3
"""Bisect where a GEMV launch spends its time (EXL2_GEMV_PROBE switches) and how it scales with residency knobs."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exllamav2_amd.ext import ext_c, none_tensor
from exllamav2_amd.synth import synth_linear


def run(k, n, recipe, nm, iters=20, warm=False, act_order=True):
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    mats, hs = [], []
    for _ in range(nm):
        w = synth_linear(k, n, recipe, "cuda", gen, act_order=act_order)
        hs.append(ext_c.make_q_matrix_from_dict(w, none_tensor)); mats.append(w)
    a = torch.randn((1, k), device="cuda", dtype=torch.float16)
    c = torch.empty((1, n), device="cuda", dtype=torch.float16)
    seq = [hs[0]] * nm if warm else hs
    for h in seq: ext_c.gemm_half_q_half(a, h, c)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ext_c.graph_begin_capture(st.cuda_stream)
        for h in seq: ext_c.gemm_half_q_half(a, h, c)
        g = ext_c.graph_end_capture(st.cuda_stream)
        ext_c.graph_launch(g, st.cuda_stream); st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters): ext_c.graph_launch(g, st.cuda_stream)
        e1.record(st); st.synchronize()
    ext_c.graph_free(g)
    us = e0.elapsed_time(e1) / (iters * nm) * 1e3
    b = ext_c.q_matrix_info(hs[0])["bytes"]
    for h in hs: ext_c.free_q_matrix(h)
    return round(us, 2), round(b / us / 1e3, 1)


r4 = ([4], [1.0], 128)
KEYS = ("EXL2_GEMV_PROBE", "EXL2_GEMV_WGS", "EXL2_GEMV_WAVES", "EXL2_GEMV_GENERIC")
for shape, nm in (((4096, 4096), 48), ((4096, 11008), 20), ((11008, 4096), 20), ((4096, 32000), 8)):
    for env in ({}, {"EXL2_GEMV_PROBE": "16"}, {"EXL2_GEMV_PROBE": "18"}, {"EXL2_GEMV_WAVES": "8"}, {"EXL2_GEMV_WAVES": "8", "EXL2_GEMV_WGS": "512"}):
        for k_ in KEYS: os.environ.pop(k_, None)
        os.environ.update(env)
        cold = run(*shape, r4, nm)
        warm = run(*shape, r4, nm, warm=True)
        print(json.dumps({"shape": shape, "env": env, "cold_us_GBs": cold, "warm_us_GBs": warm}), flush=True)
