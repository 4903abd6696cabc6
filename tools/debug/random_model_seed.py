"""worst error / tolerance of ONE seed of tests/test_chain.py::test_chain_decode_random_models per decode step, against both oracles
(usage: python tools/debug/random_model_seed.py SEED [emu]) -- what a failing seed of the sweep looks like from close up"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.conftest import Backend
from tests.test_model import tiny_cfg
from exllamav2_amd.synth import synth_checkpoint
from exllamav2_amd.model import ExLlamaV2, GreedyGraphDecoder
from exllamav2_amd.cache import ExLlamaV2Cache
from oracle.model import OracleModel

seed = int(sys.argv[1]); kind = sys.argv[2] if len(sys.argv) > 2 else "hip"
be = Backend(kind)
rng = np.random.default_rng(17000 + seed)
hd = int(rng.choice([64, 128])); kvh = int(rng.choice([1, 2, 4])); g = int(rng.choice([1, 2, 4, 8]))
hidden = 128 * int(rng.integers(1, 9)); inter = 128 * int(rng.integers(1, 13))
recipe = str(rng.choice(["4.0bpw", "3.5bpw", "2.5bpw", "4.0bpw_plain", "gptq-4bit-128g"]))
batch = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 11, 16]))
cfg = tiny_cfg(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=int(rng.integers(1, 3)), num_attention_heads=kvh * g,
               num_key_value_heads=kvh, head_dim=hd, max_batch_size=16)
act_order = not recipe.startswith("gptq") or bool(rng.integers(0, 2))
print("spec", dict(hd=hd, kvh=kvh, g=g, hidden=hidden, inter=inter, recipe=recipe, batch=batch, layers=cfg.num_hidden_layers))
for route in ("chain", "modules"):
    os.environ["EXL2_CHAIN"] = "1" if route == "chain" else "0"
    ck = synth_checkpoint(cfg, be.device, recipe=recipe, seed=600 + seed, act_order=act_order)
    oracles = {r: OracleModel(cfg, ck, rounding=r) for r in ("chain", "reference")}
    model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
    cache = ExLlamaV2Cache(model, batch_size=batch)
    dec = GreedyGraphDecoder(model, cache, batch_size=batch)
    r2 = np.random.default_rng(600 + seed)
    first = r2.integers(0, cfg.vocab_size, size=(batch,))
    dec.reset(torch.from_numpy(first), 0)
    for o in oracles.values(): o.reset(batch)
    tok = first.copy()
    for i in range(2):
        dec.run(1, use_graph=False)
        got = be.n(dec.logits)[:, :cfg.vocab_size].astype(np.float64)
        for name, o in oracles.items():
            want = o.forward(tok[:, None])[:, -1]
            ratio = np.abs(got - want) / (0.03 + np.abs(want) * 2.0 ** -8)
            r, c = np.unravel_index(ratio.argmax(), ratio.shape)
            print(f"{route} step {i} vs {name}-rounding oracle: worst {ratio.max():.3f} x tol at row {r} col {c} (|logit| {abs(want[r, c]):.2f}), rows above 1x: {sorted(set(np.nonzero(ratio > 1)[0].tolist()))}")
        tok = be.n(dec.tokens(i, 1))[:, 0].copy()
    dec.free(); model.unload()
