#!/bin/bash
# Round 3 script, re-run in round 4: the device sampler inside the decode-step graph (capture_sampled): parity tests + tokens/s next to greedy and to the
# per-token sampler launch of round 2
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_sampling.py tests/test_model.py -m gpu -q --timeout 200 > $R/r04s_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $R/r03s_pytest.log
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
timeout -k 10 200 python tools/sampler_bench.py 2>/dev/null > $R/r04_sampler_bench.jsonl; grep '"rows": 1,' $R/r04_sampler_bench.jsonl | cut -c1-130
echo "== smoke + full gpu suite on the final source"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $R/r04_smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -3 | tee $R/r04_pytest_gpu_tail.log
