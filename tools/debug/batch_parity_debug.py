#!/usr/bin/env python3
"""Where does bench.py's parity check at --batch N deviate?  Per-sequence errors of (a) the prompt through model.forward, (b) decode
steps through GreedyGraphDecoder, (c) the same decode steps through model.forward on a second cache -- against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from exllamav2_amd import ExLlamaV2, ExLlamaV2Cache, GreedyGraphDecoder
from exllamav2_amd.config import ExLlamaV2Config
from exllamav2_amd.synth import synth_checkpoint

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = "cuda:0"
cfg = ExLlamaV2Config.llama2_7b(max_seq_len=2048)
cfg.max_batch_size = batch
cfg.num_hidden_layers = layers
ck = synth_checkpoint(cfg, dev, recipe="4.0bpw", seed=0)
oracle = bench.oracle_for_parity(cfg, ck, layers=layers)
model = ExLlamaV2(cfg, device=dev).load(ck)
ids = (np.array([[1, 15043, 3186, 29892]]) + 977 * np.arange(batch)[:, None]) % cfg.vocab_size
cache = ExLlamaV2Cache(model, batch_size=batch, max_seq_len=256)
cache2 = ExLlamaV2Cache(model, batch_size=batch, max_seq_len=256)
oracle.reset(batch)
want = oracle.forward(ids)
got = model.forward(torch.from_numpy(ids), cache, last_id_only=False).float().cpu().numpy().astype(np.float64)
model.forward(torch.from_numpy(ids), cache2)
print("prompt: max err per (sequence, position):\n", np.abs(got[..., :cfg.vocab_size] - want).max(-1).round(3))
tok = want[:, -1].argmax(-1).astype(np.int64)
dec = GreedyGraphDecoder(model, cache, batch_size=batch)
if "--graph" in sys.argv: dec.capture()
print("decoder route:", "chain" if dec.chain is not None else "modules")
dec.reset(torch.from_numpy(tok), ids.shape[1])
for step in range(3):
    dec.run(1, use_graph="--graph" in sys.argv)
    torch.cuda.synchronize()
    w = oracle.forward(tok[:, None])[:, -1]
    g = dec.logits.float().cpu().numpy()[:, :cfg.vocab_size].astype(np.float64)
    g2 = model.forward(torch.from_numpy(tok[:, None]), cache2).float().cpu().numpy()[:, -1, :cfg.vocab_size].astype(np.float64)
    print(f"step {step}: decoder vs oracle per sequence", np.abs(g - w).max(-1).round(3), "| model.forward vs oracle", np.abs(g2 - w).max(-1).round(3),
          "| decoder vs model.forward", np.abs(g - g2).max(-1).round(3), "| |logit| max", np.abs(w).max().round(2))
    tok = w.argmax(-1).astype(np.int64)
    # teacher-force the decoder's next token to the oracle's
    dec.reset(torch.from_numpy(tok), ids.shape[1] + step + 1)
