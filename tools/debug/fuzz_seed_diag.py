"""Diagnostic for tests/test_dropin_fast.py::test_hosts_that_do_random_things_between_module_calls: per seed, the worst
|chained - plain| / (0.03 + |plain| 2^-8) per token, on the backend named by argv[1] (emu | hip), seeds argv[2:]."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.conftest import Backend
from tests.test_dropin_fast import Host, cfg_small, load_fast

be = Backend(sys.argv[1])
fast = load_fast(be)
for seed in [int(s) for s in sys.argv[2:]]:
    rng = np.random.default_rng(23000 + seed)
    cfg = cfg_small(num_hidden_layers=int(rng.integers(1, 4)), hidden_size=int(rng.choice([128, 256, 384])),
                    intermediate_size=int(rng.choice([256, 384, 640])))
    recipe = str(rng.choice(["4.0bpw", "3.5bpw", "2.5bpw", "gptq-4bit-128g"]))
    host = Host(be, fast, cfg, seed=100 + seed, recipe=recipe)
    tokens = rng.integers(0, cfg.vocab_size, size=6).tolist()
    acts = ["none", "none", "scale", "touch", "clone", "new", "raw"]
    script = {}

    def between(li, where, x):
        key = (host.past, li, where)
        if key not in script:
            script[key] = str(rng.choice(acts))
        a = script[key]
        if a == "scale": x[..., ::2].mul_(0.5)
        elif a == "touch": x.add_(0)
        elif a == "clone": x = x.clone()
        elif a == "new": x = x * 1.0
        elif a == "raw":
            x.view(torch.int16).bitwise_xor_(0); fast.note_write(x)
        return x
    host.between = between
    fast.set_chain(True); fast.set_verify(bool(seed & 1)); fast.stats(True)
    chained = host.run(tokens); st = fast.stats(True); fast.set_verify(False)
    fast.set_chain(False)
    plain = host.run(tokens)
    # third route: no chain, no hooks changed -- and the "all none" script for scale of the model's own sensitivity
    fast.set_chain(True)
    ratios = []
    for a, b in zip(chained, plain):
        err = np.abs(a.astype(np.float64) - b)
        ratios.append(float((err / (0.03 + np.abs(b) * 2.0 ** -8)).max()))
    print(seed, recipe, cfg.num_hidden_layers, cfg.hidden_size, cfg.intermediate_size, "ratios", [round(r, 2) for r in ratios],
          "max|logit|", [round(float(np.abs(b).max()), 1) for b in plain], "mismatch", st.get("verify_mismatch"), flush=True)
    host.close()
