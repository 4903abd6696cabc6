"""Where do the chained and the plain route part?  Residual stream at every hook of the fuzz test, both routes, one seed."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.conftest import Backend
from tests.test_dropin_fast import Host, cfg_small, load_fast

be = Backend(sys.argv[1])
fast = load_fast(be)
seed = int(sys.argv[2])
rng = np.random.default_rng(23000 + seed)
cfg = cfg_small(num_hidden_layers=int(rng.integers(1, 4)), hidden_size=int(rng.choice([128, 256, 384])),
                intermediate_size=int(rng.choice([256, 384, 640])))
recipe = str(rng.choice(["4.0bpw", "3.5bpw", "2.5bpw", "gptq-4bit-128g"]))
host = Host(be, fast, cfg, seed=100 + seed, recipe=recipe)
tokens = rng.integers(0, cfg.vocab_size, size=6).tolist()
acts = ["none", "none", "scale", "touch", "clone", "new", "raw"]
script = {}
trace = None

def between(li, where, x):
    key = (host.past, li, where)
    if key not in script:
        script[key] = str(rng.choice(acts))
    a = script[key]
    trace.append((key, a, be.n(x).astype(np.float64).copy()))
    if a == "scale": x[..., ::2].mul_(0.5)
    elif a == "touch": x.add_(0)
    elif a == "clone": x = x.clone()
    elif a == "new": x = x * 1.0
    elif a == "raw":
        x.view(torch.int16).bitwise_xor_(0); fast.note_write(x)
    return x
host.between = between
fast.set_chain(True); fast.set_verify(bool(seed & 1)); fast.stats(True)
trace = []; chained = host.run(tokens); tc = trace; st = fast.stats(True); fast.set_verify(False)
fast.set_chain(False)
trace = []; plain = host.run(tokens); tp = trace
fast.set_chain(True)
print(st)
for (k, a, xc), (k2, a2, xp) in zip(tc, tp):
    assert k == k2 and a == a2
    d = np.abs(xc - xp)
    print(k, a, "max|x|", round(float(np.abs(xp).max()), 3), "rms", round(float(np.sqrt((xp ** 2).mean())), 4), "max diff", round(float(d.max()), 5),
          "rel", round(float(d.max() / (np.abs(xp).max() + 1e-9)), 5))
for i, (a, b) in enumerate(zip(chained, plain)):
    err = np.abs(a.astype(np.float64) - b)
    print("token", i, "ratio", round(float((err / (0.03 + np.abs(b) * 2.0 ** -8)).max()), 3))
