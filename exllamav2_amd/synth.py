"""Synthetic EXL2 checkpoints of a given architecture (no network, no real checkpoints on the box: BASELINE.md).

Tensors are drawn directly in the on-disk format of SURVEY.md A.1 (uniform random codes == uniform random packed
words, so no packing pass is needed) with the reference quantizer's own bit mixes (conversion/qparams.py:129-260).
Returned dicts are exactly what the reference loader hands to `ext.make_q_matrix` (module.py:116-121), on `device`.
"""
from __future__ import annotations

import math

import torch

# (bits list, proportions, group size) per linear role for a ~4.0 bpw model: qparams.py:193-198 ("[5,4] 0.1/0.9" attn),
# :250-260 (MLP), lm_head 6 bit g128 (EXL2 "h6").
RECIPES = {
    "4.0bpw": {
        "q_proj": ([5, 4], [0.1, 0.9], 128), "k_proj": ([5, 4], [0.1, 0.9], 128), "v_proj": ([5, 4], [0.1, 0.9], 64),
        "o_proj": ([5, 4], [0.1, 0.9], 128), "gate_proj": ([4], [1.0], 128), "up_proj": ([4], [1.0], 32),
        "down_proj": ([8, 4], [0.05, 0.95], [32, 128]), "lm_head": ([6], [1.0], 128),
    },
    "4.0bpw_plain": {k: ([4], [1.0], 128) for k in
                     ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")} | {"lm_head": ([6], [1.0], 128)},
    "3.5bpw": {
        "q_proj": ([4, 3], [0.5, 0.5], 128), "k_proj": ([4, 3], [0.5, 0.5], 128), "v_proj": ([4, 3], [0.6, 0.4], 64),
        "o_proj": ([4, 3], [0.5, 0.5], 128), "gate_proj": ([4, 3], [0.5, 0.5], 128), "up_proj": ([4, 3], [0.5, 0.5], 64),
        "down_proj": ([4, 3], [0.6, 0.4], 128), "lm_head": ([6], [1.0], 128),
    },
    "2.5bpw": {
        "q_proj": ([3, 2], [0.1, 0.9], 64), "k_proj": ([3, 2], [0.1, 0.9], 64), "v_proj": ([4, 3], [0.1, 0.9], 128),
        "o_proj": ([3, 2], [0.1, 0.9], 64), "gate_proj": ([3, 2], [0.1, 0.9], 64), "up_proj": ([3, 2], [0.3, 0.7], 64),
        "down_proj": ([5, 3], [0.05, 0.95], 32), "lm_head": ([6], [1.0], 128),
    },
}


def group_plan(rows: int, bits: list, props: list, group_size) -> list:
    """(bits, rows) per group, following QParams.total_bits / AdaptiveGPTQ.quantize (qparams.py:63-77)."""
    gsz = group_size if isinstance(group_size, list) else [group_size] * len(bits)
    plan = []
    remaining = rows
    for b, p, g in zip(bits, props, gsz):
        n = math.ceil(min(rows * p, remaining) / g)
        for _ in range(n):
            r = min(g, remaining)
            if r <= 0: break
            plan.append((b, r))
            remaining -= r
    assert remaining == 0, (rows, bits, props, group_size)
    return plan


def synth_linear(k: int, n: int, recipe, device, gen: torch.Generator, sigma: float = 0.02, act_order: bool = True,
                 invperm: torch.Tensor | None = None) -> dict:
    """`invperm`: use this act-order permutation instead of drawing one (projections that share their input share it)."""
    bits, props, gs = recipe
    plan = group_plan(k, bits, props, gs)
    q_groups, smax = [], []
    qrow = 0
    for b, r in plan:
        assert r % 32 == 0
        q_groups += [b, qrow]
        qrow += r * b // 32
        smax.append(sigma * math.sqrt(12.0) / (1 << b) / 93.5 * 256.0)
    g = len(plan)
    assert qrow < 65536
    dev = torch.device(device)
    w = {
        "q_weight": torch.randint(-2 ** 31, 2 ** 31 - 1, (qrow, n), dtype=torch.int32, device=dev, generator=gen),
        "q_scale": torch.randint(-2 ** 31, 2 ** 31 - 1, (g, n // 8), dtype=torch.int32, device=dev, generator=gen),
        "q_scale_max": (torch.tensor(smax, dtype=torch.float32, device=dev)
                        * (0.5 + torch.rand(g, device=dev, generator=gen))).half(),
        "q_groups": torch.tensor(q_groups, dtype=torch.int16, device=dev),
    }
    if invperm is not None:
        w["q_invperm"] = invperm.clone()
    elif act_order:
        w["q_invperm"] = torch.randperm(k, device=dev, generator=gen).to(torch.int32)
    else:
        w["q_invperm"] = torch.arange(k, device=dev, dtype=torch.int32)
    w["q_perm"] = torch.argsort(w["q_invperm"]).to(torch.int32)              # module.py:120
    return w


GPTQ_RECIPES = {"gptq-4bit-128g": 128, "gptq-4bit-32g": 32}     # BASELINE configs[0]: TinyLlama GPTQ 4-bit g128


def synth_linear_gptq(k: int, n: int, group_size: int, device, gen: torch.Generator, sigma: float = 0.02,
                      act_order: bool = False, g_idx: torch.Tensor | None = None) -> dict:
    """GPTQ tensor set (module.py:125-130): qweight int32 [K/8, N] (8 nibbles along K per word), qzeros
    int32 [G, N/8] (stored nibble = zero - 1), scales fp16 [G, N], g_idx int32 [K] (shuffled = act-order)."""
    assert k % group_size == 0 and k % 8 == 0 and n % 8 == 0
    g = k // group_size
    dev = torch.device(device)
    # uniform 4-bit codes around zero 8 with std sqrt(21.25): scale so that dequantized weights have std ~ sigma
    sc = sigma / math.sqrt(21.25)
    w = {
        "qweight": torch.randint(-2 ** 31, 2 ** 31 - 1, (k // 8, n), dtype=torch.int32, device=dev, generator=gen),
        "qzeros": torch.full((g, n // 8), 0x77777777, dtype=torch.int32, device=dev),                 # zero = 8 everywhere
        "scales": (sc * (0.5 + torch.rand(g, n, device=dev, generator=gen))).half(),
    }
    gi = torch.arange(k, device=dev, dtype=torch.int32) // group_size
    if g_idx is not None:                                   # projections of one input share their act-order (same Hessian)
        gi = g_idx.clone()
    elif act_order:
        gi = gi[torch.randperm(k, device=dev, generator=gen)]
    w["g_idx"] = gi.contiguous()
    return w


def pack_codes(codes: torch.Tensor, bits: int) -> torch.Tensor:
    """codes int64 [K, N] in [0, 2^bits) -> int32 [K * bits / 32, N] in the EXL2 bitstream (pack_tensor.cu:118-248):
    per column the 32 codes of a 32-row chunk are concatenated LSB-first into `bits` little-endian words."""
    k, n = codes.shape
    assert k % 32 == 0
    c = codes.reshape(k // 32, 32, n)
    words = torch.zeros((k // 32, bits + 1, n), dtype=torch.int64, device=codes.device)
    for i in range(32):
        w0, off = (i * bits) // 32, (i * bits) % 32
        words[:, w0] |= (c[:, i] << off) & 0xFFFFFFFF
        if off + bits > 32:
            words[:, w0 + 1] |= c[:, i] >> (32 - off)
    words = words[:, :bits].reshape(k // 32 * bits, n)
    return torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)


def structure_head(w: dict, emb: torch.Tensor, vpad: int) -> None:
    """Overwrite the codes / scales of a synthetic head (synth_linear / synth_linear_gptq output, [K = hidden, N = vpad]) with the
    quantized, shifted transpose of the embedding table `emb` [vocab, hidden] (see synth_checkpoint: structured_head)."""
    vocab, h = emb.shape
    dev = emb.device
    beta = 1.5 / math.sqrt(h)
    src = torch.roll(emb.float(), shifts=1, dims=0)                    # row v = E[v - 1]
    wt = torch.zeros((h, vpad), dtype=torch.float32, device=dev)
    wt[:, :vocab] = src.t() * beta                                     # W[k, v], original feature order
    if "q_weight" in w:                                                # EXL2: uniform bit width per recipe (lm_head: 6 bit)
        qg = w["q_groups"].cpu().tolist()
        bits = qg[0]
        assert all(b == bits for b in qg[0::2]), "structured head: one bit width expected"
        half = 1 << (bits - 1)
        sc = 3.5 * beta / (half - 1)
        sc16 = float(torch.tensor(sc).half())
        codes = torch.clamp(torch.round(wt / sc16) + half, 0, 2 * half - 1).to(torch.int64)
        codes = codes[w["q_perm"].long()]                              # packed row i holds original feature q_perm[i]
        w["q_weight"] = pack_codes(codes, bits)
        w["q_scale"] = torch.zeros_like(w["q_scale"])                  # every nibble 0: scale = 1^2 * q_scale_max / 256 (ext.py:336)
        w["q_scale_max"] = torch.full_like(w["q_scale_max"], sc16 * 256.0)
    else:                                                              # GPTQ 4 bit, zero 8
        sc = 3.0 * beta / 7
        sc16 = float(torch.tensor(sc).half())
        codes = torch.clamp(torch.round(wt / sc16) + 8, 0, 15).to(torch.int64)
        c = codes.reshape(h // 8, 8, vpad)
        words = torch.zeros((h // 8, vpad), dtype=torch.int64, device=dev)
        for i in range(8):
            words |= c[:, i] << (4 * i)
        w["qweight"] = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)
        w["scales"] = torch.full_like(w["scales"], sc16)


def synth_checkpoint(cfg, device, recipe: str = "4.0bpw", seed: int = 0, act_order: bool = True,
                     layers=None, with_embed: bool = True, with_head: bool = True, shared_perm: bool = True,
                     down_act_order: bool = False, structured_head: bool = True) -> dict:
    """{'model.layers.0.self_attn.q_proj': {...}, ..., 'model.norm': tensor, 'model.embed_tokens': tensor}.
    Every layer draws from its own generator (seed, layer index), so a rank of a layer-split run can build exactly its
    slice of the same checkpoint (`layers` = iterable of layer indices).
    shared_perm (EXL2): q/k/v share one act-order permutation, gate/up (and every expert's w1/w3) another -- as in every
    checkpoint the reference's quantizer writes: the permutation is argsort(diag(H)) (adaptivegptq.py:236-251) and k/v
    reuse q's Hessian, gate reuses up's, the experts reuse w1.0's (conversion/quantize.py:138-139,165,190-192).
    shared_perm=False draws one permutation per linear (format-legal, never produced by the quantizer).
    structured_head (greedy-parity caveat, option ii: "give the synthetic model structure ... so the model has
    a confident next token"): the head is the QUANTIZED transpose of the embedding table shifted by one token,
    W[k, v] = beta * E[(v - 1) mod vocab, k] with beta = 1.5 / sqrt(hidden), so the logit of token t + 1 collects the
    part of the residual stream that still is E[t] (~ 1.5 sqrt(hidden) x that fraction: 10 .. 80) while every other logit
    stays ~ N(0, 1.5^2) like a random head's.  The oracle's top-1 / top-2 margin is then tens of times the stated logit
    tolerance on every step instead of a fraction of it, so the greedy-token comparisons are never vacuous; weights are
    still in the on-disk format (6-bit codes / 4-bit GPTQ nibbles, one scale per group), bytes and shapes are unchanged.
    structured_head=False draws the head at random like every other linear."""
    if recipe in GPTQ_RECIPES:
        gs = GPTQ_RECIPES[recipe]
        rec = {k: gs for k in RECIPES["4.0bpw"]}
        # (`ip`: the tensor set whose act-order this one shares -- GPTQ with desc_act sorts by diag(H), and q/k/v, gate/up see
        # the same inputs, hence the same H and the same g_idx: AutoGPTQ / the reference's loader module.py:125-130 alike)
        make_ = lambda k, n, r, dev, gen, sigma, act, ip=None: synth_linear_gptq(k, n, r, dev, gen, sigma, act, g_idx=ip)
    else:
        rec = RECIPES[recipe]
        make_ = lambda k, n, r, dev, gen, sigma, act, ip=None: synth_linear(k, n, r, dev, gen, sigma, act, invperm=ip)
    make = make_
    h, inter = cfg.hidden_size, cfg.intermediate_size
    qd = cfg.num_attention_heads * cfg.head_dim
    kvd = cfg.num_key_value_heads * cfg.head_dim
    ck = {}
    s_attn = 1.0 / math.sqrt(h)

    def gen_for(tag: int):
        g = torch.Generator(device=device)
        g.manual_seed(seed * 100003 + tag)
        return g

    for i in (range(cfg.num_hidden_layers) if layers is None else layers):
        gen = gen_for(i + 1)
        p = f"model.layers.{i}"
        ck[f"{p}.self_attn.q_proj"] = make(h, qd, rec["q_proj"], device, gen, s_attn, act_order)
        shared_of = lambda w: w.get("q_invperm") if "q_invperm" in w else w.get("g_idx")
        ip = shared_of(ck[f"{p}.self_attn.q_proj"]) if (shared_perm and act_order) else None
        ck[f"{p}.self_attn.k_proj"] = make(h, kvd, rec["k_proj"], device, gen, s_attn, act_order, ip)
        ck[f"{p}.self_attn.v_proj"] = make(h, kvd, rec["v_proj"], device, gen, s_attn, act_order, ip)
        ck[f"{p}.self_attn.o_proj"] = make(qd, h, rec["o_proj"], device, gen, 0.5 / math.sqrt(qd), act_order)
        if getattr(cfg, "num_experts", 0):
            # Mixtral-style sparse MLP (moe_mlp.py:25-133): experts w1 (gate), w3 (up), w2 (down) + fp16 router
            ip = None
            for e in range(cfg.num_experts):
                q = f"{p}.block_sparse_moe.experts.{e}"
                ck[f"{q}.w1"] = make(h, inter, rec["gate_proj"], device, gen, s_attn, act_order, ip)
                if shared_perm and act_order and ip is None: ip = shared_of(ck[f"{q}.w1"])
                ck[f"{q}.w3"] = make(h, inter, rec["up_proj"], device, gen, s_attn, act_order, ip)
                ck[f"{q}.w2"] = make(inter, h, rec["down_proj"], device, gen, 0.5 / math.sqrt(inter), False)
            ck[f"{p}.block_sparse_moe.gate"] = (torch.randn(cfg.num_experts, h, device=device, generator=gen) * s_attn).half()
        else:
            ck[f"{p}.mlp.up_proj"] = make(h, inter, rec["up_proj"], device, gen, s_attn, act_order)
            ip = shared_of(ck[f"{p}.mlp.up_proj"]) if (shared_perm and act_order) else None
            ck[f"{p}.mlp.gate_proj"] = make(h, inter, rec["gate_proj"], device, gen, s_attn, act_order, ip)
            # the reference folds down_proj's act-order into gate/up at load -> identity perm here
            # (down_act_order=True keeps down_proj's own permutation, as on disk: the reference's loader does the folding)
            ck[f"{p}.mlp.down_proj"] = make(inter, h, rec["down_proj"], device, gen, 0.5 / math.sqrt(inter), down_act_order and act_order)
        ck[f"{p}.input_layernorm"] = (1 + 0.1 * torch.randn(h, device=device, generator=gen)).half()
        ck[f"{p}.post_attention_layernorm"] = (1 + 0.1 * torch.randn(h, device=device, generator=gen)).half()
    if with_embed:
        ck["model.embed_tokens"] = torch.randn(cfg.vocab_size, h, device=device, generator=gen_for(0)).half()
    if with_head:
        gen = gen_for(99991)
        ck["model.norm"] = (1 + 0.1 * torch.randn(h, device=device, generator=gen)).half()
        vpad = (cfg.vocab_size + 31) // 32 * 32                               # linear.py:82-88 pads out_features to x32
        ck["lm_head"] = make(h, vpad, rec["lm_head"], device, gen, s_attn, act_order)
        if structured_head:
            emb = torch.randn(cfg.vocab_size, h, device=device, generator=gen_for(0)).half()      # == model.embed_tokens
            structure_head(ck["lm_head"], emb, vpad)
    return ck
