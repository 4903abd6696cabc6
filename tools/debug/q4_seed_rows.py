"""per row and step of one seed of test_q4_cache_random_models on the module-by-module route: device vs the Q4 oracle, and how far the
oracle's own admissible variants sit from each other at that row (usage: python tools/debug/q4_seed_rows.py SEED [emu] [chain])"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.conftest import Backend
from tests.test_model import tiny_cfg
from exllamav2_amd.synth import synth_checkpoint
from exllamav2_amd.model import ExLlamaV2, GreedyGraphDecoder
from exllamav2_amd.cache import ExLlamaV2Cache_Q4
from oracle.model import OracleModel
seed = int(sys.argv[1]); kind = sys.argv[2] if len(sys.argv) > 2 else "hip"
os.environ["EXL2_CHAIN"] = "1" if (len(sys.argv) > 3 and sys.argv[3] == "chain") else "0"
be = Backend(kind)
rng = np.random.default_rng(29000 + seed)
kvh = int(rng.choice([4, 8])); g = int(rng.choice([1, 2, 4, 8]))
cfg = tiny_cfg(num_attention_heads=kvh * g, num_key_value_heads=kvh, head_dim=128, hidden_size=128 * int(rng.integers(1, 7)),
               intermediate_size=128 * int(rng.integers(1, 7)), num_hidden_layers=int(rng.integers(1, 3)))
recipe, batch = str(rng.choice(["4.0bpw", "3.5bpw", "2.5bpw"])), int(rng.integers(1, 5))
ck = synth_checkpoint(cfg, be.device, recipe=recipe, seed=700 + seed)
O = {"ref_q4": OracleModel(cfg, ck), "ref_fp16": OracleModel(cfg, ck), "chain_fp16": OracleModel(cfg, ck, rounding="chain")}
model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
cache = ExLlamaV2Cache_Q4(model, batch_size=batch)
dec = GreedyGraphDecoder(model, cache, batch_size=batch)
first = np.random.default_rng(700 + seed).integers(0, cfg.vocab_size, size=(batch,))
dec.reset(torch.from_numpy(first), 0)
for o in O.values(): o.reset(batch)
tok = first.copy()
tol = lambda w: 0.03 + np.abs(w) * 2.0 ** -8
for i in range(4):
    dec.run(1, use_graph=False)
    got = be.n(dec.logits)[:, :cfg.vocab_size].astype(np.float64)
    W = {k: o.forward(tok[:, None], q4_cache=k.endswith("q4"))[:, -1] for k, o in O.items()}
    for r in range(batch):
        f = lambda a, b: float((np.abs(a[r] - b[r]) / tol(b)[r]).max())
        print(f"step {i} row {r}: device vs ref_q4 {f(got, W['ref_q4']):.2f} | oracles: "
              f"chain_fp16 vs ref_fp16 {f(W['chain_fp16'], W['ref_fp16']):.2f}  ref_q4 vs ref_fp16 {f(W['ref_q4'], W['ref_fp16']):.2f}")
    tok = be.n(dec.tokens(i, 1))[:, 0].copy()
