"""whole-model prefill (8 x 2048 tokens, the first LAYERS layers of the synthetic 7B) for rocprofv3 --kernel-trace --stats: where a layer's time goes
usage: rocprofv3 --kernel-trace --stats ... -- python tools/prefill_model_profile.py [layers] [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exllamav2_amd import ExLlamaV2, ExLlamaV2Cache, ExLlamaV2Config
from exllamav2_amd.synth import synth_checkpoint
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = ExLlamaV2Config.llama2_7b(max_seq_len=2048, max_input_len=2048, max_batch_size=8)
cfg.num_hidden_layers = layers
ck = synth_checkpoint(cfg, "cuda:0", recipe="4.0bpw", seed=0)
model = ExLlamaV2(cfg, device="cuda:0").load(ck)
cache = ExLlamaV2Cache(model, batch_size=8, max_seq_len=2048)
ids = torch.randint(0, cfg.vocab_size - 1, (8, 2048)).to('cuda:0')
for i in range(reps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cache.current_seq_len = 0
    model.forward(ids, cache, preprocess_only=True)
    torch.cuda.synchronize()
    print(f"pass {i}: {(time.perf_counter() - t0) * 1e3 / layers:.3f} ms per layer")
