#!/usr/bin/env python3
"""Times the device sampler (csrc/sampling.hip, exl2_sample_rows) on an MI355X: one JSON line per (vocabulary, rows, setting),
microseconds per launch from HIP events on the launching stream (100 launches after 10 warm-up ones), next to the greedy
arg-max kernel of the decode graph on the same logits.  Usage: python tools/sampler_bench.py > profiles/rNN_sampler_bench.jsonl"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exllamav2_amd.ext import ext_c  # noqa: E402


def timed(fn, reps=100, warm=10):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    dev = "cuda:0"
    g = torch.Generator(device="cpu"); g.manual_seed(0)
    for vocab in (32000, 128256):
        for rows in (1, 16):
            lg = (torch.randn((rows, vocab), generator=g) * 2.5).to(torch.float16).to(dev)
            tok = torch.zeros(rows, dtype=torch.int32, device=dev)
            pr = torch.zeros(rows, dtype=torch.float32, device=dev)
            ws = torch.empty((rows, vocab), dtype=torch.float32, device=dev)
            us_arg = timed(lambda: ext_c.argmax_rows(lg, tok, vocab))
            for name, st in (("default T0.8 k50 p0.8", (0.8, 50, 0.8, 0.0)), ("k500 p0.95 minp0.02", (1.0, 500, 0.95, 0.02)),
                             ("k2", (1.0, 2, 0.0, 0.0))):
                us = timed(lambda: ext_c.sample_rows(lg, st[0], st[1], st[2], st[3], 0.37, tok, pr, workspace=ws))
                print(json.dumps({"vocab": vocab, "rows": rows, "setting": name, "sample_rows_us": round(us, 2),
                                  "argmax_rows_us": round(us_arg, 2), "logits_bytes": rows * vocab * 2}), flush=True)


if __name__ == "__main__":
    main()
