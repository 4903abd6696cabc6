"""Chained decode (csrc/qgemv_flat.hip + the chain entry points of modules.hip) against the oracle and against the
module-by-module route.

What the chain replaces is a COMPOSITION of reference kernels (q_attn.cu:153-345, q_mlp.cu:153-236: rms_norm -> q_gemm ->
rope / act_mul -> q_gemm -> residual): its results must equal the oracle's within the same fp16 tolerance as the
unchained route (tests/test_model.py), step by step, for every bit-width mix, for GPTQ, for several rows.
"""
import os
import re

import numpy as np
import pytest
import torch

from exllamav2_amd.cache import ExLlamaV2Cache, ExLlamaV2Cache_Q4
from exllamav2_amd.config import ExLlamaV2Config
from exllamav2_amd.model import ExLlamaV2, GreedyGraphDecoder
from exllamav2_amd.synth import synth_checkpoint
from oracle.model import OracleModel
from oracle import exl2 as OX
from oracle import modules as OM
from tests.test_model import tiny_cfg, check_logits, confident
from tests.util import exl2_to_torch, half_tol


def _decode_and_check(be, cfg, recipe, batch, steps=3, seed=0, act_order=True, expect_chain=True, logit_slack=1.0, rounding="reference",
                      skip_rounding_sensitive_rows=False, **ck_kw):
    """skip_rounding_sensitive_rows: rows of a step at which the oracle's two admissible roundings (fp16 where the reference's kernels
    round / where ours round) are themselves more than the tolerance apart are left out of that step's LOGIT check -- such a row
    amplifies one-ulp differences (random weights, attention over one or two keys) and says nothing about either path; found on the
    MI355X at seed 35 of test_chain_decode_random_models: one row of 11, the two oracles 3.4 x tol apart, the chained route at
    1.2 / 2.2 x and the module-by-module route at 2.2 / 1.1 x from them (tools/debug/random_model_seed.py)."""
    ck = synth_checkpoint(cfg, be.device, recipe=recipe, seed=seed, act_order=act_order, **ck_kw)
    oracle = OracleModel(cfg, ck, rounding=rounding)
    other = OracleModel(cfg, ck, rounding="reference" if rounding == "chain" else "chain") if skip_rounding_sensitive_rows else None
    model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
    cache = ExLlamaV2Cache(model, batch_size=batch)
    dec = GreedyGraphDecoder(model, cache, batch_size=batch)
    assert (dec.chain is not None) == expect_chain
    if not be.is_emu:
        dec.capture()
    rng = np.random.default_rng(seed)
    first = rng.integers(0, cfg.vocab_size, size=(batch,))
    dec.reset(torch.from_numpy(first), 0)
    oracle.reset(batch)
    if other is not None: other.reset(batch)
    tok = first.copy()
    n_conf = 0
    n_rows = n_kept = 0
    for i in range(steps):
        dec.run(1, use_graph=not be.is_emu)
        want = oracle.forward(tok[:, None])[:, -1]
        got = be.n(dec.logits)[:, :cfg.vocab_size]
        if other is not None:
            w2 = other.forward(tok[:, None])[:, -1]
            keep = np.all(np.abs(want - w2) <= 0.03 + np.abs(want) * 2.0 ** -8, axis=-1)
            n_rows += batch; n_kept += int(keep.sum())
            check_logits(got[keep][:, None], want[keep][:, None])
        elif logit_slack == 1.0:
            check_logits(got[:, None], want[:, None])
        else:
            err = np.abs(got.astype(np.float64) - want)
            assert np.all(err <= logit_slack * (0.03 + np.abs(want) * 2.0 ** -8)), float((err / (0.03 + np.abs(want) * 2.0 ** -8)).max())
        g = be.n(dec.tokens(i, 1))[:, 0]
        assert np.array_equal(g, got.argmax(-1))                      # the device samples its own logits greedily
        conf = confident(want)
        assert np.array_equal(g[conf], want.argmax(-1)[conf])
        n_conf += int(conf.sum())
        tok = g.copy()                                                  # follow the device: each step is checked alone
    assert n_conf >= 1, "no step had a confident oracle margin: the token check would be vacuous"
    assert other is None or 2 * n_kept >= n_rows, (n_kept, n_rows)      # (the exclusion must stay the exception)
    assert (dec.chain is not None) == expect_chain
    dec.free()
    model.unload()


def reference_yardstick() -> float:
    """worst distance (in units of the model tolerance) of the reference's own kernel composition from the float64 oracle over the
    128 random models of test_chain_decode_random_models -- measured by execution: tests/golden/reference_model_yardstick.json"""
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_model_yardstick.json")) as f:
        y = json.load(f)["random_models"]
    assert y["n"] >= 128 and 1.0 <= y["worst"] < 8.0, y["worst"]
    return float(y["worst"])


@pytest.mark.parametrize("recipe", ["4.0bpw", "3.5bpw", "2.5bpw"])
@pytest.mark.parametrize("batch", [1, 3])
def test_chain_decode_equals_oracle(be, recipe, batch):
    cfg = tiny_cfg(num_attention_heads=4, num_key_value_heads=2, intermediate_size=384)
    _decode_and_check(be, cfg, recipe, batch, seed=11)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("EXL2_TEST_SEEDS", "4")))))        # (more seeds: a longer hunt, by hand)
def test_chain_decode_random_models(be, seed, monkeypatch):
    """Seeded random small models through the chained decoder against the oracle, step by step: hidden / intermediate sizes that are
    not powers of two, 1-8 query heads per kv head, head_dim 64 / 128, every recipe (EXL2 2.5-4.0 bpw mixes, GPTQ), 1-16 sequences
    (<= 4 rows: the one-row forms; more: ROWS / XMEM forms and row groups), FP16 cache.

    The bar is MEASURED, not asserted (round-5 review, item 5): tests/golden/reference_model_yardstick.json holds, for the same 128
    model specs / weights / first tokens, how far the reference's OWN decode kernels -- gemm_half_q_half_kernel, rms_norm_kernel,
    act_mul_kernel, rope, executed on the host and composed as q_attn.cu / q_mlp.cu compose them (oracle/ref_kernel_model.py;
    generator: tests/golden/make_golden_model_yardstick.py) -- sit from this same float64 oracle in units of the model tolerance
    0.03 + |x| 2^-8: median 0.32, 10 of 128 above 1.0, worst 4.09 (single rows where attention over one or two keys amplifies
    one-ulp differences in q and k).  The device must be no farther from the reference-rounding oracle than the reference's own
    kernels were at their worst; over the same 128 models it measures median 0.14, 4 above 1.0, worst 2.03 chained (0.12 / 2 /
    1.62 module by module).  OracleModel(rounding="chain") -- fp16 roundings where OUR kernels have them -- stays as a diagnostic at
    1 x the tolerance: it tells a rounding-point difference from a defect, it is not the bar."""
    bar = reference_yardstick()
    rng = np.random.default_rng(17000 + seed)
    hd = int(rng.choice([64, 128]))
    kvh = int(rng.choice([1, 2, 4])); g = int(rng.choice([1, 2, 4, 8]))
    hidden = 128 * int(rng.integers(1, 9))
    inter = 128 * int(rng.integers(1, 13))
    recipe = str(rng.choice(["4.0bpw", "3.5bpw", "2.5bpw", "4.0bpw_plain", "gptq-4bit-128g"]))
    batch = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 11, 16]))
    cfg = tiny_cfg(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=int(rng.integers(1, 3)), num_attention_heads=kvh * g,
                   num_key_value_heads=kvh, head_dim=hd, max_batch_size=16)
    act_order = not recipe.startswith("gptq") or bool(rng.integers(0, 2))
    _decode_and_check(be, cfg, recipe, batch, steps=2, seed=600 + seed, act_order=act_order, expect_chain=True, rounding="chain",
                      skip_rounding_sensitive_rows=True)
    _decode_and_check(be, cfg, recipe, batch, steps=2, seed=600 + seed, act_order=act_order, expect_chain=True, logit_slack=bar)
    # ... and the module-by-module route (the kernels behind the plain operator calls) on the same model
    monkeypatch.setenv("EXL2_CHAIN", "0")
    _decode_and_check(be, cfg, recipe, batch, steps=2, seed=600 + seed, act_order=act_order, expect_chain=False, logit_slack=bar)


@pytest.mark.parametrize("recipe", ["4.0bpw", "3.5bpw"])
def test_chain_decode_wide_k(be, recipe):
    """hidden 4096 (32 items per tile): the one-row gate|up launch takes 4 waves per tile with 8 items each in registers, two
    LDS-DMA instructions per activation slice (qgemv_lean.hip, S = 4 pair geometry); q|k|v 8 waves x 4 items"""
    cfg = tiny_cfg(hidden_size=4096, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1,
                   head_dim=128)
    _decode_and_check(be, cfg, recipe, 1, steps=2, seed=5)


@pytest.mark.parametrize("xp_tiled", ["1", "0"])
@pytest.mark.parametrize("batch", [16, 11, 8, 5, 4])
def test_chain_decode_many_rows(be, batch, xp_tiled, monkeypatch):
    """4 sequences: one launch of the lean kernel's wave-private form with four finalising waves; 5..16: its ROWS form (round 4: the
    workgroup stages the whole rows once, every finalising wave takes several rows; the default) or -- EXL2_XP_TILED=1 -- the
    hand-off buffers in the matrix cores' layout and q|k|v, gate|up, down, head in the XMEM form (o_proj: ROWS; measured slower on
    the 7B shapes, kept as an option).  Every launch of the step on the lean kernel, none left to the round-2 kernel"""
    monkeypatch.setenv("EXL2_XP_TILED", xp_tiled)
    cfg = tiny_cfg(max_batch_size=16)
    be.ext.chain_route_counts(reset=True)
    _decode_and_check(be, cfg, "4.0bpw", batch, steps=2, seed=12)
    lean, flat = be.ext.chain_route_counts(reset=True)
    assert lean > 0 and flat == 0, (lean, flat)


@pytest.mark.parametrize("recipe,batch", [("4.0bpw", 16), ("3.5bpw", 13), ("2.5bpw", 9), ("4.0bpw_plain", 5), ("gptq-4bit-128g", 16)])
def test_chain_decode_many_rows_operands_from_memory(be, recipe, batch, monkeypatch):
    """XMEM form (round 4): no staged activations -- every wave requests the A operands of its items from memory, three items ahead
    of the one it decodes; what the host picks when M x K does not fit the LDS (7B down_proj at 16 rows), forced here for every
    launch of the step (EXL2_LEAN_XMEM=2).  Same results as the ROWS form's: the oracle's."""
    monkeypatch.setenv("EXL2_LEAN_XMEM", "2")
    cfg = tiny_cfg(max_batch_size=16, intermediate_size=384, num_attention_heads=4, num_key_value_heads=2)
    be.ext.chain_route_counts(reset=True)
    _decode_and_check(be, cfg, recipe, batch, steps=2, seed=23)
    lean, flat = be.ext.chain_route_counts(reset=True)
    assert lean > 0 and flat == 0, (lean, flat)


@pytest.mark.parametrize("recipe,batch,inter", [("4.0bpw", 16, 384), ("3.5bpw", 13, 416), ("2.5bpw", 9, 384), ("4.0bpw_plain", 5, 640),
                                                ("gptq-4bit-128g", 16, 384)])
def test_chain_decode_many_rows_tiled_hand_off(be, recipe, batch, inter, monkeypatch):
    """gate | up -> down at 5 .. 16 rows in the layout the matrix cores read ([K / 8][16 rows][8 halfs]: FlatIn.c_tiled / a_tiled):
    the pair launch's epilogue writes it, down_proj's XMEM form reads one coalesced kilobyte per A operand.  Default where the rows'
    staged copy would not fit the LDS (7B: 16 x 11008); forced here for every step (EXL2_MLP_TILED=2).  K = 416: a partial last item."""
    monkeypatch.setenv("EXL2_MLP_TILED", "2")
    cfg = tiny_cfg(max_batch_size=16, intermediate_size=inter, num_attention_heads=4, num_key_value_heads=2)
    be.ext.chain_route_counts(reset=True)
    _decode_and_check(be, cfg, recipe, batch, steps=2, seed=29)
    lean, flat = be.ext.chain_route_counts(reset=True)
    assert lean > 0 and flat == 0, (lean, flat)


def test_chain_decode_rows_that_do_not_fit_take_one_launch(be, monkeypatch):
    """default policy: rows x (K + 8) x 2 bytes beyond the LDS -> ONE launch in the XMEM form, no row groups (EXL2_LEAN_XMEM=0 brings
    the groups back: test_chain_decode_many_rows_mixed_groupings)"""
    cfg = tiny_cfg(max_batch_size=16, intermediate_size=640, num_attention_heads=4, num_key_value_heads=2)
    calls = {"n": 0}
    orig = be.ext.q_mlp_forward_chain_part
    def spy(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    monkeypatch.setattr(be.ext, "q_mlp_forward_chain_part", spy)
    monkeypatch.setenv("EXL2_CHAIN_ROWS_LDS", str(16 * (cfg.hidden_size + 8) * 2))
    _decode_and_check(be, cfg, "3.5bpw", 16, steps=2, seed=17)
    assert calls["n"] == 0


@pytest.mark.parametrize("batch,grid", [(16, 1), (11, 2), (6, 3)])
def test_chain_decode_many_rows_workgroups_walk_their_units(be, batch, grid, monkeypatch):
    """ROWS form: the grid is sized to the CUs and a workgroup takes units u, u + grid, ... with one staged copy of the rows
    (a 7B gate|up launch: 688 tile pairs on 256 workgroups).  Small shapes never have more units than CUs, so the walk is forced."""
    monkeypatch.setenv("EXL2_LEAN_ROWS_GRID", str(grid))
    monkeypatch.setenv("EXL2_XP_TILED", "0")                     # (the ROWS form for every launch of the step)
    cfg = tiny_cfg(max_batch_size=16, intermediate_size=384, num_attention_heads=4, num_key_value_heads=2)
    be.ext.chain_route_counts(reset=True)
    _decode_and_check(be, cfg, "3.5bpw", batch, steps=2, seed=19)
    lean, flat = be.ext.chain_route_counts(reset=True)
    assert lean > 0 and flat == 0, (lean, flat)


@pytest.mark.parametrize("recipe,batch", [("3.5bpw", 16), ("2.5bpw", 7)])
def test_chain_decode_many_rows_mixed_groupings(be, recipe, batch, monkeypatch):
    """down_proj's K is larger than hidden: its rows x (K + 8) x 2 bytes stop fitting in LDS at fewer rows than those of q|k|v / o /
    gate|up, so a step runs gate|up once over all rows and down over row groups (exl2_q_mlp_forward_chain_part).  Forced here on
    a small shape by shrinking the LDS the host grants the staged rows (the grouping logic is the host's: model.step_chain)."""
    cfg = tiny_cfg(max_batch_size=16, intermediate_size=640, num_attention_heads=4, num_key_value_heads=2)
    import exllamav2_amd.model as M
    calls = {"n": 0}
    orig = be.ext.q_mlp_forward_chain_part
    def spy(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    monkeypatch.setattr(be.ext, "q_mlp_forward_chain_part", spy)
    monkeypatch.setenv("EXL2_CHAIN_ROWS_LDS", str(16 * (cfg.hidden_size + 8) * 2))      # 16 rows of K = hidden fit, of K = 640 do not
    monkeypatch.setenv("EXL2_LEAN_XMEM", "0")                                            # (default: one launch in the XMEM form)
    _decode_and_check(be, cfg, recipe, batch, steps=2, seed=17)
    assert calls["n"] > 0


def test_llama2_70b_widths_stay_on_the_lean_kernel(be, monkeypatch):
    """configs[3]'s shapes (hidden 8192, intermediate 28672, 2.5 bpw: 2 / 3-bit items, 1792-row slices on 16 waves, shares of two
    register passes) through one layer + head of the chained step over a Q4 cache: every launch must be planned by the lean kernel
    -- a host-side limit that declines one of them silently un-chains the WHOLE decoder (it did, for one GPU call of round 4).
    Emulator: the plans only (EXL2_LEAN_PLAN_ONLY: a 0.9 G-weight layer takes minutes to emulate); GPU: the real launches."""
    from exllamav2_amd.config import ExLlamaV2Config
    if be.is_emu:
        monkeypatch.setenv("EXL2_LEAN_PLAN_ONLY", "1")
    cfg = ExLlamaV2Config.llama2_70b(max_seq_len=256, max_input_len=32)
    cfg.num_hidden_layers = 1
    cfg.vocab_size = 512
    ck = synth_checkpoint(cfg, be.device, recipe="2.5bpw", seed=0)
    model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
    cache = ExLlamaV2Cache_Q4(model, batch_size=1)
    dec = GreedyGraphDecoder(model, cache, batch_size=1)
    assert dec.chain is not None
    be.ext.chain_route_counts(reset=True)
    dec.reset(torch.tensor([3]), 0)
    dec.run(1, use_graph=False)
    lean, flat = be.ext.chain_route_counts(reset=True)
    assert dec.chain is not None and lean == 5 and flat == 0, (dec.chain is not None, lean, flat)     # q|k|v, o, gate|up, down, head
    if not be.is_emu:
        tok = be.n(dec.tokens(0, 1))
        assert 0 <= int(tok[0, 0]) < cfg.vocab_size
    dec.free(); model.unload()


@pytest.mark.parametrize("hidden,inter,heads,kv,recipe,batch", [(4096, 14336, 32, 8, "4.0bpw", 8), (4096, 14336, 32, 8, "4.0bpw", 16),
                                                                (8192, 28672, 64, 8, "2.5bpw", 8)])
def test_many_rows_at_wide_intermediate_stay_chained(be, monkeypatch, hidden, inter, heads, kv, recipe, batch):
    """Llama-3-8B / Mistral / Mixtral-dense widths (intermediate 14336) and 70B widths (28672) at 6..16 sequences: down_proj's K is
    beyond the lean kernel's one-pass XMEM form, so the optimistic "all rows in one launch" grouping is declined for that launch
    -- the decoder must regroup (down_proj in row groups of 4) and stay on the chained route, never silently un-chain the whole
    step (round-4 advisor finding).  Emulator: host plans only; GPU: the real launches, tokens in range."""
    from exllamav2_amd.config import ExLlamaV2Config
    if be.is_emu:
        monkeypatch.setenv("EXL2_LEAN_PLAN_ONLY", "1")
    cfg = ExLlamaV2Config(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=1, num_attention_heads=heads,
                          num_key_value_heads=kv, head_dim=128, vocab_size=512, max_seq_len=256, max_input_len=16, max_batch_size=16)
    ck = synth_checkpoint(cfg, be.device, recipe=recipe, seed=0)
    model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
    cache = ExLlamaV2Cache(model, batch_size=batch)
    dec = GreedyGraphDecoder(model, cache, batch_size=batch)
    assert dec.chain is not None
    be.ext.chain_route_counts(reset=True)
    dec.reset(torch.arange(3, 3 + batch), 0)
    dec.run(1, use_graph=False)
    lean, flat = be.ext.chain_route_counts(reset=True)
    assert dec.chain is not None, "the decoder left the chained route"
    g = dec.chain.get("group_override", {}).get(inter)
    assert g in (1, 2, 4) and set(dec.chain["group_override"]) == {inter}   # down_proj regrouped, the other launches kept all rows
    assert flat == 0 and lean >= 5, (lean, flat)
    be.ext.chain_route_counts(reset=True)
    dec.run(1, use_graph=False)                                           # a settled step: q|k|v, o, gate|up, down x groups, head
    lean, flat = be.ext.chain_route_counts(reset=True)
    assert (lean, flat) == (4 + -(-batch // g), 0), (lean, flat)
    if not be.is_emu:
        tok = be.n(dec.tokens(0, 2))
        assert tok.min() >= 0 and tok.max() < cfg.vocab_size
    dec.free(); model.unload()


@pytest.mark.parametrize("recipe,act_order,shared", [("gptq-4bit-128g", False, True), ("gptq-4bit-32g", True, True),
                                                     ("gptq-4bit-32g", True, False)])
def test_chain_decode_gptq(be, recipe, act_order, shared):
    cfg = tiny_cfg(num_attention_heads=4, num_key_value_heads=2, intermediate_size=384)
    # GPTQ act-order (desc_act) permutations are derived per matrix from g_idx: q/k/v (gate/up) of a real checkpoint share it
    # (same inputs, same Hessian) -> chained; three different shuffles (format-legal) -> the module-by-module route
    _decode_and_check(be, cfg, recipe, 2, seed=13, act_order=act_order, expect_chain=shared or not act_order, shared_perm=shared)


def test_distinct_permutations_fall_back(be):
    """q/k/v with three different act-order permutations (format-legal, never written by the quantizer): the modules report
    not chain-capable and the decoder takes the module-by-module route with the same results."""
    cfg = tiny_cfg()
    _decode_and_check(be, cfg, "4.0bpw", 1, seed=14, expect_chain=False, shared_perm=False)


def test_chain_tokens_equal_unchained_route(be, monkeypatch):
    cfg = tiny_cfg(num_hidden_layers=3)
    outs = []
    for chain in ("1", "0"):
        monkeypatch.setenv("EXL2_CHAIN", chain)
        ck = synth_checkpoint(cfg, be.device, recipe="4.0bpw", seed=15)
        model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
        cache = ExLlamaV2Cache(model, batch_size=2)
        dec = GreedyGraphDecoder(model, cache, batch_size=2)
        assert (dec.chain is not None) == (chain == "1")
        dec.reset(torch.tensor([3, 50]), 0)
        dec.run(5, use_graph=False)
        outs.append((be.n(dec.tokens(0, 5)).copy(), be.n(dec.logits).astype(np.float64).copy()))
        dec.free(); model.unload()
    # same arithmetic per element up to fp32 summation order: logits within a few fp16 ulps, tokens equal where it matters
    assert np.abs(outs[0][1] - outs[1][1]).max() < 0.02
    assert (outs[0][0] == outs[1][0]).mean() >= 0.8


@pytest.mark.parametrize("recipe,batch", [("4.0bpw", 1), ("3.5bpw", 3), ("gptq-4bit-128g", 2)])
def test_overlapped_chain_equals_serial_chain(be, monkeypatch, recipe, batch):
    """EXL2_CHAIN_OVERLAP=1 (csrc/chain_sync.h): the same launches of the lean kernel on two alternating streams, dependencies
    through words in memory, a gate kernel ahead of every launch.  Same arithmetic, same order of every sum (the pipelined
    form of the serial chain changes WHEN an item is decoded, not the order of the sums): logits and tokens are bit-identical to
    the one-stream chain -- on the GPU through two captured graphs (one per stream) replayed side by side for several steps;
    the hand-off words (counters, shards, entry counts, "go") are back to zero after every step and no wait gave up."""
    cfg = tiny_cfg(num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2, intermediate_size=384)
    outs = []
    for overlap in ("0", "1"):
        monkeypatch.setenv("EXL2_CHAIN_OVERLAP", overlap)
        ck = synth_checkpoint(cfg, be.device, recipe=recipe, seed=21, act_order=not recipe.startswith("gptq"))
        model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
        cache = ExLlamaV2Cache(model, batch_size=batch)
        dec = GreedyGraphDecoder(model, cache, batch_size=batch)
        assert dec.chain is not None and ("flags" in dec.chain) == (overlap == "1")
        if not be.is_emu:
            dec.capture()
        dec.reset(torch.arange(batch) + 5, 0)
        logits = []
        for _ in range(6):
            dec.run(1, use_graph=not be.is_emu)
            logits.append(be.n(dec.logits).copy())
        outs.append((be.n(dec.tokens(0, 6)).copy(), np.stack(logits)))
        if overlap == "1":
            # the hand-off words are back to zero after every step (their consumers zero them), no wait gave up
            with dec._on_stream():
                dec.step_eager()
            if not be.is_emu:
                torch.cuda.synchronize()
            flags = be.n(dec.chain["flags"]).copy()
            # word 2 of a launch's block counts waits that GAVE UP (hw.h: bounded spins -- a wait whose producer's graph had not
            # started ~15 ms later): timing of the two graph launches, not arithmetic -- reported, the results are compared below
            gave_up = int(flags.reshape(-1, 1024)[:, 2].sum())
            flags.reshape(-1, 1024)[:, 2] = 0
            if gave_up:
                import warnings
                warnings.warn(f"overlapped chain: {gave_up} waits gave up (the two graphs did not run side by side)")
            assert np.all(flags == 0), np.nonzero(flags)
            assert sum(be.ext.chain_route_counts()) > 0
        dec.free(); model.unload()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])


def _q4_chain_case(be, monkeypatch, cfg, recipe, batch, steps, ck_seed=16, slack=1.0, rows_ok=None):
    """both routes (chained, module by module) over an ExLlamaV2Cache_Q4 against OracleModel.forward(q4_cache=True), step by step:
    logits, tokens where the oracle is confident, the codes each route wrote"""
    for chain in ("1", "0"):
        monkeypatch.setenv("EXL2_CHAIN", chain)
        ck = synth_checkpoint(cfg, be.device, recipe=recipe, seed=ck_seed)
        oracle = OracleModel(cfg, ck)
        # (both before load(): it re-lays the checkpoint's tensors out in place)
        cond = (OracleModel(cfg, ck), OracleModel(cfg, ck, rounding="chain")) if rows_ok is not None else None
        model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
        cache = ExLlamaV2Cache_Q4(model, batch_size=batch)
        dec = GreedyGraphDecoder(model, cache, batch_size=batch)
        assert (dec.chain is not None) == (chain == "1")
        be.ext.chain_route_counts(reset=True)
        if not be.is_emu:
            dec.capture()
        first = np.random.default_rng(ck_seed).integers(0, cfg.vocab_size, size=(batch,)) if ck_seed != 16 else np.array([3, 50][:batch])
        dec.reset(torch.from_numpy(first), 0)
        oracle.reset(batch)
        if cond is not None:                                            # (see test_q4_cache_random_models: rows nothing can be said about)
            cond[0].reset(batch); cond[1].reset(batch)
        tok = first.copy()
        n_conf = 0
        for i in range(steps):
            dec.run(1, use_graph=not be.is_emu)
            want = oracle.forward(tok[:, None], q4_cache=True)[:, -1]
            got = be.n(dec.logits)[:, :cfg.vocab_size]
            if cond is not None:
                w_a, w_b = cond[0].forward(tok[:, None])[:, -1], cond[1].forward(tok[:, None])[:, -1]
                rows_ok = rows_ok & (np.abs(w_a - w_b) <= 0.5 * (0.03 + np.abs(w_a) * 2.0 ** -8)).all(axis=-1)      # (once lost, a sequence stays out)
                # ... and rows at which the 4-bit cache ITSELF moves the oracle's logits by more than 8 tolerances (FP16-cache oracle
                # against Q4-cache oracle): one admissible code flip -- the codec's allowance below -- is then worth a tolerance or more.
                # Seeds 31 / 83 on the MI355X: 70 x / 752 x at the row that failed (its neighbours 1-4 x), both routes alike
                # (profiles/r10c_gpu_sweeps_128_seeds.txt, tools/debug/q4_seed_rows.py)
                rows_ok = rows_ok & (np.abs(want - w_a) <= 8.0 * (0.03 + np.abs(w_a) * 2.0 ** -8)).all(axis=-1)
            if slack == 1.0:
                check_logits(got[:, None], want[:, None])              # the model tolerance of the FP16-cache tests, not a multiple
            else:
                err = np.abs(got.astype(np.float64) - want)
                ok = err <= slack * (0.03 + np.abs(want) * 2.0 ** -8)
                if rows_ok is not None:
                    ok[~rows_ok] = True
                assert np.all(ok), (chain, i, float((err / (0.03 + np.abs(want) * 2.0 ** -8)).max()))
            g = be.n(dec.tokens(i, 1))[:, 0]
            conf = confident(want)
            if rows_ok is not None:
                conf = conf & rows_ok
            assert np.array_equal(g[conf], want.argmax(-1)[conf])
            n_conf += int(conf.sum())
            tok = g.copy()                                              # follow the device's tokens ...
            # ... and its cache codes: the oracle quantized ITS K/V of this step; the route's K/V are the same up to fp16
            # ulps, so all but a few codes at rounding boundaries must agree (everything written so far is compared, i.e.
            # also that nothing outside the step's blocks was touched) -- then the oracle continues from the device's codes
            for layer in range(cfg.num_hidden_layers):
                flipped = oracle.q4_adopt(layer, be.n(cache.key_states[layer]), be.n(cache.key_scales[layer]),
                                          be.n(cache.value_states[layer]), be.n(cache.value_scales[layer]), i + 1)
                assert flipped < slack * (0.01 / (i + 1) + 0.002), (chain, i, layer, flipped)
        assert n_conf >= 1 or slack != 1.0
        n_chain = sum(be.ext.chain_route_counts())
        assert (n_chain > 0) == (chain == "1")                      # chained launches ran / did not run
        dec.free(); model.unload()


@pytest.mark.parametrize("recipe,batch,hd,launches", [("4.0bpw", 1, 64, "1"), ("2.5bpw", 2, 64, "1"), ("3.5bpw", 2, 64, "4"), ("4.0bpw", 1, 128, "1"),
                                                      ("2.5bpw", 2, 128, "1"), ("4.0bpw", 1, 128, "2"), ("2.5bpw", 2, 128, "4")])
def test_q4_cache_decodes_on_the_chain(be, monkeypatch, recipe, batch, hd, launches):
    """Q4 KV cache (configs[3]) on the chained route: q|k|v from the chain, RoPE + quantised append, attention straight from
    the codes with the output in o_proj's packed order (attn_q4.hip out_invperm), o / gate|up / down chained -- and the same
    steps on the module-by-module route.  Checker: OracleModel.forward(q4_cache=True), the reference's ExLlamaV2Cache_Q4
    semantics (cache.py:472-556: earlier tokens read back dequantized, the step's own K/V in fp16, touched blocks quantized
    after attention) -- BOTH routes at the model tolerance step by step, tokens where the oracle is confident, and the codes
    each route wrote against the oracle's codes."""
    # head_dim 128: the decode step's attention side is ONE launch (csrc/attn_q4.hip FUSED form; EXL2_Q4_LAUNCHES=2: RoPE + pack, then
    # attention; =4: the round-4 sequence) -- head_dim 64: the one-launch form since round 6 (even kv-head counts), else the four-launch sequence
    cfg = tiny_cfg(num_attention_heads=512 // hd, num_key_value_heads=512 // hd, head_dim=hd, hidden_size=512, intermediate_size=512,
                   num_hidden_layers=2)
    steps = 6
    monkeypatch.setenv("EXL2_Q4_LAUNCHES", launches)
    taken = {"one": 0, "two": 0}
    one_launch, two_launch = be.ext.attn_q4_decode_fused, be.ext.rope_quant_append_q4

    def spy_one(*a, **k):
        ok = one_launch(*a, **k)
        taken["one"] += int(ok)
        return ok

    def spy_two(*a, **k):
        ok = two_launch(*a, **k)
        taken["two"] += int(ok)
        return ok
    monkeypatch.setattr(be.ext, "attn_q4_decode_fused", spy_one)
    monkeypatch.setattr(be.ext, "rope_quant_append_q4", spy_two)
    _q4_chain_case(be, monkeypatch, cfg, recipe, batch, steps)
    # (round 6: head_dim 64 takes the one-launch form too -- a wave packs the rows of two adjacent kv heads; the two-launch form stays head_dim 128)
    assert (taken["one"] > 0) == (launches == "1") and (taken["two"] > 0) == (hd == 128 and launches == "2"), taken


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("EXL2_TEST_SEEDS", "4")))))        # (more seeds: a longer hunt, by hand)
def test_q4_cache_random_models(be, monkeypatch, seed):
    """Seeded random small models over the Q4 cache, head_dim 128 (the one-launch decode step; on odd seeds the two-launch form):
    4 / 8 kv heads, 1-8 query heads per kv head, 1-4 sequences, every EXL2 recipe, both routes.  The bar is the measured one of
    test_chain_decode_random_models (how far the reference's own kernels sit from this oracle at their worst over 128 random models,
    FP16 cache: 4.09 x the model tolerance -- a Q4 cache only adds a quantizer on top); the code-flip allowance scales with it."""
    rng = np.random.default_rng(29000 + seed)
    kvh = int(rng.choice([4, 8])); g = int(rng.choice([1, 2, 4, 8]))   # (kv width a multiple of the codec's 512-element block: the direct route)
    cfg = tiny_cfg(num_attention_heads=kvh * g, num_key_value_heads=kvh, head_dim=128, hidden_size=128 * int(rng.integers(1, 7)),
                   intermediate_size=128 * int(rng.integers(1, 7)), num_hidden_layers=int(rng.integers(1, 3)))
    monkeypatch.setenv("EXL2_Q4_LAUNCHES", "2" if seed & 1 else "1")
    recipe, batch = str(rng.choice(["4.0bpw", "3.5bpw", "2.5bpw"])), int(rng.integers(1, 5))
    # Rows a comparison means nothing for: random weights now and then give a row whose residual stream nearly cancels, and the next
    # RMSNorm multiplies every rounding difference -- e.g. seed 20 of this sweep, hidden 640, two layers, token 77: the two admissible
    # roundings of the ORACLE sit 11.6 x the tolerance apart in its logits (its neighbours in the batch: 0.05), the chained route
    # 8.8 x from the reference rounding and the module-by-module route 1.7 x, on the FP16 cache alike.  Such rows are
    # found by exactly that -- the oracle in both roundings (FP16 cache), step by step on the device's tokens -- and left out of the
    # logit / token checks from then on (they still run, their codes are still compared).
    _q4_chain_case(be, monkeypatch, cfg, recipe, batch, steps=4, ck_seed=700 + seed, slack=reference_yardstick(), rows_ok=np.ones((batch,), dtype=bool))


def test_row_groups_taken_by_different_kernels(be, monkeypatch):
    """7 sequences as row groups 4 + 3 where the 4-row group is declined by the round-3 kernel (here: forced; in the field: its
    LDS budget at K ~ 20 k) and goes to the round-2 kernel while the 3-row group stays: the two kernels publish different
    numbers of partial sums of squares per row, so every group must carry ITS OWN count to its consumer."""
    monkeypatch.setenv("EXL2_CHAIN_ROWGROUPS", "4")
    monkeypatch.setenv("EXL2_LEAN_DECLINE_M", "4")
    cfg = tiny_cfg(num_attention_heads=4, num_key_value_heads=2, intermediate_size=384)
    be.ext.chain_route_counts(reset=True)
    _decode_and_check(be, cfg, "4.0bpw", 7, steps=2, seed=12)
    lean, flat = be.ext.chain_route_counts(reset=True)
    assert lean > 0 and flat > 0, (lean, flat)


# ---- op level: the chain entry points one by one ------------------------------------------------------------------------

def _mk(be, k, n, spec, seed, invperm=None):
    t = OX.synth_exl2(k, n, spec, seed=seed, act_order=True, sigma=0.05)
    if invperm is not None:
        t["q_invperm"] = invperm.copy()
    ref = OX.exl2_reconstruct(t)
    w = exl2_to_torch(be, t)
    return t, ref, w, be.ext.make_q_matrix_from_dict(w, None)


CHAIN_SPECS = {   # full runs and partial super-chunks of every bit width
    "all_widths": (800, [(8, 32, 32), (6, 32, 96), (5, 64, 128), (4, 128, 256), (3, 64, 160), (2, 64, 128)]),
    "8_5_3": (416, [(8, 32, 128), (5, 64, 160), (3, 64, 128)]),
    "6_4_2": (544, [(6, 32, 128), (4, 128, 256), (2, 64, 160)]),
    "tails_only": (96, [(5, 32, 32), (4, 32, 64)]),
    "two_big_runs": (8448, [(4, 128, 4224), (3, 64, 4096), (2, 64, 128)]),    # a second run of >= 32 super-chunks: register ring
    # K = 11008 (7B down_proj): 86 items per tile -> the 16-wave geometry (5-6 items per wave); 7 rows do not fit the LDS next to
    # the scale rows: the XMEM form
    "k11008_8_4": (11008, [(8, 32, 544), (4, 128, 10464)]),
}


@pytest.mark.parametrize("spec_name", list(CHAIN_SPECS))
@pytest.mark.parametrize("rows", [1, 2, 7, 16])
def test_gemm_chain_norm_pre(be, rows, spec_name, monkeypatch):
    """exl2_gemm_half_q_half_chain: c = rmsnorm(x) . W from (xp = x * permuted norm weight as its producer leaves it, ss partials);
    every bit width in K"""
    if rows == 16 and spec_name != "k11008_8_4":
        pytest.skip("16 rows: the shape whose rows do not fit the LDS only")
    k, spec = CHAIN_SPECS[spec_name]
    n = 96
    t, ref, w, h = _mk(be, k, n, spec, 21)
    rng = np.random.default_rng(rows)
    x = (rng.standard_normal((rows, k)) * 2).astype(np.float16)
    nw = (1 + 0.1 * rng.standard_normal(k)).astype(np.float16)
    perm = np.argsort(t["q_invperm"]).astype(np.int64)                 # packed row -> input feature
    # what a producer publishes: x times the consumer's norm weight (fp32 product, one rounding), in the consumer's packed order
    xp = (x.astype(np.float32) * nw.astype(np.float32)).astype(np.float16)[:, perm]
    npart = 5                                                          # any split of the sum of squares into partials
    sq = x.astype(np.float32) ** 2
    ss = np.stack([sq[:, i::npart].sum(-1) for i in range(npart)], axis=-1).astype(np.float32)
    c = torch.zeros((rows, n), dtype=torch.float16, device=be.device)
    if spec_name == "k11008_8_4" and rows > 4:
        # 7 x (K + 8) x 2 bytes + scale rows + partial sums > LDS: the XMEM form (A operands from memory) takes it; without that
        # form the entry point says so -- loud, nothing launched: the decoder un-chains on this
        monkeypatch.setenv("EXL2_LEAN_XMEM", "0")
        with pytest.raises(RuntimeError, match="not covered"):
            be.ext.gemm_half_q_half_chain(be.t(xp), be.t(ss), npart, 1e-5, h, c, rows)
        monkeypatch.delenv("EXL2_LEAN_XMEM")
    be.ext.gemm_half_q_half_chain(be.t(xp), be.t(ss), npart, 1e-5, h, c, rows)
    want = OX.gemm_ref(OM.rms_norm(x, nw, 1e-5), ref, exact=True)
    # the normalised activations may sit one fp16 ulp from the oracle's (fp32 partial sums vs float64): K independent
    # 2^-11 relative perturbations add ~ sqrt(K) * 2^-11 * |x| |w| on top of the output rounding
    slack = 2 * max(1.0, (k / 1024.0) ** 0.5)
    assert np.all(np.abs(be.n(c).astype(np.float64) - want) <= slack * half_tol(want, k))
    be.ext.free_q_matrix(h)


# shares of SEVERAL register loads (qgemv_lean.hip: ring_passes): 16 waves x 9 three-bit items (two loads of 8), 16 x 13 two-bit items
# (two loads of 10), 16 x 14 four-bit items (three loads of 6) + a partial item behind the last full one
DEEP_SPECS = {
    "k18432_3b": (18432, [(3, 32, 18432)]),
    "k26624_2b": (26624, [(2, 64, 26624)]),
    "k28768_4b_3b": (28768, [(4, 128, 28672), (3, 32, 96)]),
}


@pytest.mark.parametrize("spec_name", list(DEEP_SPECS))
@pytest.mark.parametrize("rows", [1, 3])
def test_gemm_chain_shares_of_several_register_loads(be, rows, spec_name, capfd, monkeypatch):
    """A wave's share beyond its registers (70B down_proj: K = 28672 on 16 waves) is decoded as a RING: item q of the next register
    load is requested as soon as item q of the current one is decoded.  Numerics against the oracle for 2, 3 and 4-bit shares of
    two and three loads, one and three rows; the plan is checked to be the multi-load one (a wave with more items than LeanDepth)."""
    k, spec = DEEP_SPECS[spec_name]
    if rows * k * 2 > 140 * 1024:
        pytest.skip("rows x K beyond one workgroup's LDS: the decoder serves such launches as smaller row groups")
    n = 32
    t, ref, w, h = _mk(be, k, n, spec, 5)
    rng = np.random.default_rng(rows + 40)
    x = (rng.standard_normal((rows, k)) * 2).astype(np.float16)
    nw = (1 + 0.1 * rng.standard_normal(k)).astype(np.float16)
    perm = np.argsort(t["q_invperm"]).astype(np.int64)
    xp = (x.astype(np.float32) * nw.astype(np.float32)).astype(np.float16)[:, perm]
    ss = (x.astype(np.float32) ** 2).sum(-1, keepdims=True).astype(np.float32)
    c = torch.zeros((rows, n), dtype=torch.float16, device=be.device)
    be.ext.chain_route_counts(reset=True)                              # (process-wide: not the launches of whatever ran before)
    monkeypatch.setenv("EXL2_LEAN_TRACE", "1")
    be.ext.gemm_half_q_half_chain(be.t(xp), be.t(ss), 1, 1e-5, h, c, rows)
    monkeypatch.delenv("EXL2_LEAN_TRACE")
    err = capfd.readouterr().err
    assert " S=16 " in err, err[:400]
    depth = {2: 10, 3: 8, 4: 6}
    shares = [(int(m.group(1)), int(m.group(2))) for m in re.finditer(r"; (\d+) x (\d+)b", err)]
    assert any(cnt > depth[bits] for cnt, bits in shares), shares
    want = OX.gemm_ref(OM.rms_norm(x, nw, 1e-5), ref, exact=True)
    slack = 2 * max(1.0, (k / 1024.0) ** 0.5)
    assert np.all(np.abs(be.n(c).astype(np.float64) - want) <= slack * half_tol(want, k))
    lean, flat = be.ext.chain_route_counts(reset=True)
    assert flat == 0 and lean > 0
    be.ext.free_q_matrix(h)


ONE_ROW_SPECS = {
    "k18432_3b_g32": (18432, [(3, 32, 18432)]),                          # 18 items = 72 scale rows per wave: the third copy instruction
    "k28672_5b_3b_g32": (28672, [(5, 32, 1408), (3, 32, 27264)]),        # configs[3]'s down_proj (2.5 bpw recipe): 5-bit shares of several loads too
    "k14336_4b_3b": (14336, [(4, 128, 8576), (3, 128, 5760)]),           # Mixtral's down_proj
}


@pytest.mark.parametrize("spec_name", list(ONE_ROW_SPECS))
def test_gemm_chain_one_row_two_tiles_per_workgroup(be, spec_name, capfd, monkeypatch):
    """Round 6, the LOADS form of the chained decode kernel: ONE row of a long K whose tiles outnumber the CUs (70B down_proj: 512 tiles
    of K = 28672) -- the plain form is a 16-wave workgroup per tile, alone on its CU (two rounds of 256); here a workgroup takes TWO
    tiles over one staged copy of the row, 8 waves per tile, every wave's share a ring of up to four register loads, up to 128 scale
    rows per wave, and walks its units.  Emulator and GPU: EXL2_LEAN_CUS=2 makes 6 tiles 'more than the CUs' (3 units on 2
    workgroups: one of them walks).  Plan from the host's trace, numerics against the oracle; EXL2_LEAN_ROWS1=0 = the plain form."""
    k, spec = ONE_ROW_SPECS[spec_name]
    n = 96
    t, ref, w, h = _mk(be, k, n, spec, 9)
    rng = np.random.default_rng(77)
    x = (rng.standard_normal((1, k)) * 2).astype(np.float16)
    nw = (1 + 0.1 * rng.standard_normal(k)).astype(np.float16)
    perm = np.argsort(t["q_invperm"]).astype(np.int64)
    xp = (x.astype(np.float32) * nw.astype(np.float32)).astype(np.float16)[:, perm]
    ss = (x.astype(np.float32) ** 2).sum(-1, keepdims=True).astype(np.float32)
    c = torch.zeros((1, n), dtype=torch.float16, device=be.device)
    monkeypatch.setenv("EXL2_LEAN_CUS", "2")
    monkeypatch.setenv("EXL2_LEAN_TRACE", "1")
    be.ext.chain_route_counts(reset=True)
    be.ext.gemm_half_q_half_chain(be.t(xp), be.t(ss), 1, 1e-5, h, c, 1)
    monkeypatch.delenv("EXL2_LEAN_TRACE")
    err = capfd.readouterr().err
    head = [l for l in err.split("\n") if l.startswith("[lean] M=1 ")]
    assert head and " S=8 slots=2 wgs=3 " in head[0] and "form=rows" in head[0], head[:2]
    depth = {2: 10, 3: 8, 4: 6, 5: 5}
    shares = [(int(m.group(1)), int(m.group(2))) for m in re.finditer(r"; (\d+) x (\d+)b", err)]
    assert any(cnt > 2 * depth[bits] for cnt, bits in shares), shares          # (three register loads at least)
    want = OX.gemm_ref(OM.rms_norm(x, nw, 1e-5), ref, exact=True)
    slack = 2 * max(1.0, (k / 1024.0) ** 0.5)
    assert np.all(np.abs(be.n(c).astype(np.float64) - want) <= slack * half_tol(want, k))
    monkeypatch.setenv("EXL2_LEAN_ROWS1", "0")                                # (the plain form: same sums within the same bar)
    c2 = torch.zeros((1, n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half_chain(be.t(xp), be.t(ss), 1, 1e-5, h, c2, 1)
    assert np.all(np.abs(be.n(c2).astype(np.float64) - want) <= slack * half_tol(want, k))
    lean, flat = be.ext.chain_route_counts(reset=True)
    assert flat == 0 and lean == 2
    be.ext.free_q_matrix(h)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("EXL2_TEST_SEEDS", "12")))))       # (more seeds: a longer hunt, by hand)
def test_gemm_chain_random_bit_mixes_and_depths(be, seed):
    """Seeded random matrices through exl2_gemm_half_q_half_chain: K from 2048 to ~20 k, one to three bit-width sections in the
    quantizer's order (descending), group sizes 32 / 64 / 128, one to four rows -- so that the host plan meets shares of one, two and
    three register loads on 8 and on 16 waves, partial items at section ends, uniform and per-chunk scales.  Checker: the oracle's
    reconstruct-then-matmul on rmsnorm(x).  What the plan does not cover must raise (never a silent wrong launch)."""
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([2048, 4096, 8192, 11008, 14336, 18432])) + 32 * int(rng.integers(0, 3))
    widths = sorted(rng.choice([8, 6, 5, 4, 3, 2], size=int(rng.integers(1, 4)), replace=False).tolist(), reverse=True)
    cuts = sorted(rng.choice(np.arange(1, k // 32), size=len(widths) - 1, replace=False).tolist()) if len(widths) > 1 else []
    edges = [0] + [32 * c for c in cuts] + [k]
    spec = [(int(b), int(rng.choice([32, 64, 128])), edges[i + 1] - edges[i]) for i, b in enumerate(widths)]
    rows = int(rng.integers(1, 5))
    n = 32
    t, ref, w, h = _mk(be, k, n, spec, 300 + seed)
    x = (rng.standard_normal((rows, k)) * 2).astype(np.float16)
    nw = (1 + 0.1 * rng.standard_normal(k)).astype(np.float16)
    perm = np.argsort(t["q_invperm"]).astype(np.int64)
    xp = (x.astype(np.float32) * nw.astype(np.float32)).astype(np.float16)[:, perm]
    ss = (x.astype(np.float32) ** 2).sum(-1, keepdims=True).astype(np.float32)
    c = torch.zeros((rows, n), dtype=torch.float16, device=be.device)
    try:
        be.ext.gemm_half_q_half_chain(be.t(xp), be.t(ss), 1, 1e-5, h, c, rows)
    except RuntimeError as e:
        assert "not covered" in str(e), (spec, rows, str(e))
        be.ext.free_q_matrix(h)
        pytest.skip(f"plan declined (loudly): K={k} {spec} rows={rows}")
    want = OX.gemm_ref(OM.rms_norm(x, nw, 1e-5), ref, exact=True)
    slack = 2 * max(1.0, (k / 1024.0) ** 0.5)
    bad = np.abs(be.n(c).astype(np.float64) - want) > slack * half_tol(want, k)
    assert not bad.any(), (k, spec, rows, int(bad.sum()))
    be.ext.free_q_matrix(h)


def test_qkv_launch_takes_two_register_loads_on_eight_waves(be, capfd, monkeypatch):
    """hidden 8192 at 2.5 bpw (configs[3]'s q|k|v): v_proj's 3 / 4-bit items do not fit ONE register load of 8 waves (capacity 60 of
    64 items), and the launch's three matrices share one geometry.  Round 4 took 16 waves for all three -- a 16-wave workgroup is alone
    on its CU, 640 tiles ran as three rounds; now 8 waves with a second (ring) load for the waves that need it, three workgroups per
    CU.  Plan checked from the host's trace, numerics through the chained decode against the oracle."""
    cfg = tiny_cfg(hidden_size=8192, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1,
                   head_dim=128)
    monkeypatch.setenv("EXL2_LEAN_TRACE", "1")
    _decode_and_check(be, cfg, "2.5bpw", 1, steps=2, seed=5)
    monkeypatch.delenv("EXL2_LEAN_TRACE")
    err = capfd.readouterr().err
    blocks = err.split("[lean] M=")
    qkv = [b for b in blocks if b.startswith("1 K=8192 mats=3 ")]
    assert qkv and all(" S=8 " in b.split("\n")[0] for b in qkv), [b.split("\n")[0] for b in blocks][:8]
    depth = {2: 10, 3: 8, 4: 6}
    shares = [(int(m.group(1)), int(m.group(2))) for m in re.finditer(r"; (\d+) x (\d+)b", qkv[0])]
    assert any(cnt > depth.get(bits, 99) for cnt, bits in shares), shares


def test_gate_up_pair_takes_several_register_loads_on_four_waves(be, capfd, monkeypatch):
    """hidden 8192 at 2.5 bpw (configs[3]'s gate|up): the 4 waves of a pair's tile hold 16-19 items of 2 / 3 bits each, more than ONE
    register load (10 / 8 items).  Rounds 4-5 took the 16-wave workgroup (alone on its CU: seven rounds of 256 for the 70B's 1792
    tile pairs); round 6: the 4-waves-per-tile pair with up to three (ring) loads per wave, its own instantiation, two workgroups per
    CU (70B + Q4 94 -> 101 tok/s same box, profiles/r08b_ab_70b.txt).  Plan checked from the host's trace, numerics through the
    chained decode against the oracle; EXL2_LEAN_PAIR4_PASSES=1 (the former plan) must give the same logits within the same bar."""
    cfg = tiny_cfg(hidden_size=8192, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1,
                   head_dim=128)
    monkeypatch.setenv("EXL2_LEAN_TRACE", "1")
    _decode_and_check(be, cfg, "2.5bpw", 1, steps=2, seed=5)
    monkeypatch.delenv("EXL2_LEAN_TRACE")
    err = capfd.readouterr().err
    blocks = err.split("[lean] M=")
    gu = [b for b in blocks if b.startswith("1 K=8192 mats=2 ")]
    assert gu and all(" S=4 " in b.split("\n")[0] for b in gu), [b.split("\n")[0] for b in blocks][:8]
    depth = {2: 10, 3: 8, 4: 6}
    shares = [(int(m.group(1)), int(m.group(2))) for m in re.finditer(r"; (\d+) x (\d+)b", gu[0])]
    assert any(cnt > depth.get(bits, 99) for cnt, bits in shares), shares


@pytest.mark.parametrize("spec_id", ["5_4", "3_2_pairs"])
@pytest.mark.parametrize("rows", [1, 4])
def test_lean_kernel_identity_rows_equal_reconstruct(be, rows, spec_id):
    """The reference's own parity relation (tests/test_gemv.py:136-165: gemm(I) == reconstruct()) on the CHAINED decode kernel,
    bit for bit: one-hot activation rows through exl2_gemm_half_q_half_chain with sum(x^2) = K and eps = 0 (the epilogue's
    1 / rms factor is exactly 1) must return the rows of reconstruct(): 5-bit and 4-bit sections, one scale per item (g128: the scale
    on the fp32 partial sum) and a scale per chunk (g32 / g64: on the weights).  (Written for round 5's raw 4-bit feed experiment,
    profiles/history/r05_raw4_experiment.txt, which it passed; kept because the chained kernel had no such test.)"""
    k, n = 1024, 96
    spec = [(5, 128, 128), (4, 128, 512), (4, 32, 256), (4, 64, 128)]
    if spec_id == "3_2_pairs":
        # round 6: 3 / 2-bit items whose groups are chunk PAIRS (group size 64) take the scale on the pair's fp32 sum (LEAN_DUO);
        # a g32 section of each width beside them (a scale per chunk, on the weights)
        spec = [(4, 128, 128), (3, 64, 384), (3, 32, 128), (2, 64, 256), (2, 32, 128)]
    t, ref, w, h = _mk(be, k, n, spec, 77)
    be.ext.chain_route_counts(reset=True)                              # (a process-wide counter: other tests' launches are not this test's)
    perm = np.argsort(t["q_invperm"]).astype(np.int64)                 # packed row i holds input feature perm[i]
    eye = np.eye(k, dtype=np.float16)
    ss = np.full((rows, 1), float(k), dtype=np.float32)
    picks = range(0, k, rows) if not be.is_emu else list(range(0, 160, rows)) + list(range(160, k, 7 * rows) if spec_id == "3_2_pairs" else range(k - 416, k, 7 * rows))
    for r0 in picks:
        idx = [(r0 + i) % k for i in range(rows)]
        c = torch.zeros((rows, n), dtype=torch.float16, device=be.device)
        be.ext.gemm_half_q_half_chain(be.t(eye[idx]), be.t(ss), 1, 0.0, h, c, rows)
        got = be.n(c)
        want = ref[perm[idx]]
        assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), (r0, np.abs(got.astype(np.float32) - want.astype(np.float32)).max())
    lean, flat = be.ext.chain_route_counts(reset=True)
    assert flat == 0 and lean > 0
    be.ext.free_q_matrix(h)


@pytest.mark.parametrize("case", ["outliers", "tiny", "saturating"])
def test_gemm_chain_norm_pre_extreme_activations(be, case):
    """The chained hand-off stores fp16(clamp(x * w)) and scales the finished sums by 1 / rms(x), where the reference's rms_norm
    rounds x * (1 / rms) * w once (rms_norm.cu:33-175).  What that costs at the edges of fp16's range, against rms_norm + q_gemm:
      * outliers: a few channels at +-3e4 (the largest residual-stream activations seen in 7B-class models are ~1e3) with norm
        weights <= 1.5: products stay below 65504 -- the ordinary tolerance holds;
      * tiny: a residual stream of ~1e-4 with norm weights ~0.05, products ~5e-6 = fp16 SUBNORMALS (quantised to 6e-8): the
        reference normalises first and keeps 11 bits; here each product keeps 6-7 bits, the K-term sum averages that out -- the
        ordinary tolerance holds (measured: 2 % of it; the outputs are ~0.05 in magnitude);
      * saturating: an outlier whose product exceeds 65504 is CLAMPED (documented deviation: the producers saturate, they never
        emit inf) -- the result equals the reference's for the clamped activation, which the test states explicitly."""
    k, spec = CHAIN_SPECS["6_4_2"]
    n = 96
    t, ref, w, h = _mk(be, k, n, spec, 33)
    rng = np.random.default_rng(5)
    rows = 3
    x = rng.standard_normal((rows, k)).astype(np.float32)
    nw = (1 + 0.2 * rng.standard_normal(k)).astype(np.float32).clip(0.2, 1.5)
    slack = 2.0
    if case == "outliers":
        x[:, rng.choice(k, 6, replace=False)] = 3e4 * rng.choice([-1.0, 1.0], (rows, 6))
    elif case == "tiny":
        x *= 1e-4
        nw = (0.05 * (1 + 0.2 * rng.standard_normal(k))).astype(np.float32).clip(0.02, 0.1)
    else:
        x[:, 7] = 4e4
        nw[7] = 2.5                                                     # 1e5 > 65504
    x = x.astype(np.float16); nw = nw.astype(np.float16)
    perm = np.argsort(t["q_invperm"]).astype(np.int64)
    prod = (x.astype(np.float32) * nw.astype(np.float32)).clip(-65504.0, 65504.0)
    xp = prod.astype(np.float16)[:, perm]
    sq = x.astype(np.float32) ** 2
    npart = 3
    ss = np.stack([sq[:, i::npart].sum(-1) for i in range(npart)], axis=-1).astype(np.float32)
    c = torch.zeros((rows, n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half_chain(be.t(xp), be.t(ss), npart, 1e-5, h, c, rows)
    x_eff = x.astype(np.float64)
    if case == "saturating":
        x_eff[:, 7] = 65504.0 / float(nw[7])                            # what the clamp makes of that activation
    rms = 1.0 / np.sqrt((x.astype(np.float64) ** 2).mean(-1, keepdims=True) + 1e-5)
    xn = x_eff * rms * nw.astype(np.float64)                            # (fp64: the reference rounds this to fp16 once)
    want = xn @ ref.astype(np.float64)
    err = np.abs(be.n(c).astype(np.float64) - want)
    tol = slack * (half_tol(want, k) + 2.0 ** -10 * (np.abs(xn) @ np.abs(ref.astype(np.float64))) / np.sqrt(k))
    assert np.all(err <= tol), (case, float((err / tol).max()))
    be.ext.free_q_matrix(h)


def test_embed_rows_chain(be):
    rng = np.random.default_rng(5)
    vocab, hidden = 50, 256
    table = rng.standard_normal((vocab, hidden)).astype(np.float16)
    ids = np.array([3, 49, 0], dtype=np.int32)
    invperm = rng.permutation(hidden).astype(np.uint16)
    x = torch.zeros((3, hidden), dtype=torch.float16, device=be.device)
    xp = torch.zeros_like(x)
    ss = torch.zeros((3, 256), dtype=torch.float32, device=be.device)
    inv_t = be.t(invperm.view(np.int16))
    be.ext.embed_rows_chain(be.t(table), be.t(ids), x, inv_t.data_ptr(), None, xp, ss)
    assert np.array_equal(be.n(x), table[ids])
    want_xp = np.zeros_like(table[ids]); want_xp[:, invperm] = table[ids]
    assert np.array_equal(be.n(xp), want_xp)
    # with the first consumer's norm weight (given in that consumer's packed order): xp = fp16(x * w)
    nw = (1 + 0.1 * rng.standard_normal(hidden)).astype(np.float16)
    nw_t = be.t(nw)
    be.ext.embed_rows_chain(be.t(table), be.t(ids), x, inv_t.data_ptr(), nw_t, xp, ss)
    assert np.array_equal(be.n(xp), (want_xp.astype(np.float32) * nw.astype(np.float32)).astype(np.float16))
    want_ss = (table[ids].astype(np.float32) ** 2).sum(-1)
    assert np.allclose(be.n(ss).reshape(-1)[:3], want_ss, rtol=1e-5)      # [rows, npart = 1]
