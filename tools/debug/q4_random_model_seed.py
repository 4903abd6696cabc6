"""one seed of tests/test_chain.py::test_q4_cache_random_models, route by route and launch form by launch form
(usage: python tools/debug/q4_random_model_seed.py SEED [emu])"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.conftest import Backend
from tests.test_model import tiny_cfg
import tests.test_chain as T

class MP:
    def __init__(self, chain): self.chain = chain
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k, raising=False): os.environ.pop(k, None)

seed = int(sys.argv[1]); kind = sys.argv[2] if len(sys.argv) > 2 else "hip"
be = Backend(kind)
rng = np.random.default_rng(29000 + seed)
kvh = int(rng.choice([4, 8])); g = int(rng.choice([1, 2, 4, 8]))
cfg = tiny_cfg(num_attention_heads=kvh * g, num_key_value_heads=kvh, head_dim=128, hidden_size=128 * int(rng.integers(1, 7)),
               intermediate_size=128 * int(rng.integers(1, 7)), num_hidden_layers=int(rng.integers(1, 3)))
recipe, batch = str(rng.choice(["4.0bpw", "3.5bpw", "2.5bpw"])), int(rng.integers(1, 5))
print("spec", kvh, g, cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, recipe, batch)
for launches in ("1", "2"):
    for chain in ("both (chained first)",):
        os.environ["EXL2_Q4_LAUNCHES"] = launches
        try:
            T._q4_chain_case(be, MP(chain), cfg, recipe, batch, steps=4, ck_seed=700 + seed, slack=T.reference_yardstick(), rows_ok=np.ones((batch,), dtype=bool))
            print(f"launches={launches} chain={chain}: ok")
        except AssertionError as e:
            print(f"launches={launches} chain={chain}: FAIL {str(e)[:160]}")
