#!/bin/bash
# the device sampler inside the decode-step graph: tokens/s next to greedy (round-3 script without its test run)
mkdir -p gpurun_out; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r04_sampled_decode.txt
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from exllamav2_amd import ExLlamaV2, ExLlamaV2Cache, GreedyGraphDecoder
from exllamav2_amd.config import ExLlamaV2Config
from exllamav2_amd.synth import synth_checkpoint
cfg = ExLlamaV2Config.llama2_7b(max_seq_len=2048, max_input_len=32)
model = ExLlamaV2(cfg, device="cuda:0").load(synth_checkpoint(cfg, "cuda:0", recipe="4.0bpw", seed=0))
cache = ExLlamaV2Cache(model, batch_size=1)
dec = GreedyGraphDecoder(model, cache, batch_size=1).capture()
def timed(fn, n=128):
    dec.reset(torch.tensor([1]), 0); fn(16); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)
print("greedy (arg-max in the graph)            %.1f tok/s" % timed(lambda n: dec.run(n)))
print("sampler launched per token (round 2)     %.1f tok/s" % timed(lambda n: dec.run_sampled(n, 0.8, 50, 0.8, 0.0, seed=1)))
dec.capture_sampled(0.8, 50, 0.8, 0.0)
print("sampler inside the step graph (round 3)  %.1f tok/s" % timed(lambda n: dec.run_sampled(n, 0.8, 50, 0.8, 0.0, seed=1)))
PY
