"""End-to-end: fused module forwards, contiguous + paged decode, greedy tokens -- tiny synthetic Llama vs the oracle.

Greedy parity follows SURVEY.md 8d: random weights give near-flat logits, so token ids are compared only where the
oracle's top-1/top-2 margin exceeds the stated fp16 tolerance; logits are compared everywhere within that tolerance.
"""
import copy

import numpy as np
import pytest
import torch

from exllamav2_amd.cache import ExLlamaV2Cache, ExLlamaV2Cache_Q4
from exllamav2_amd.config import ExLlamaV2Config
from exllamav2_amd.model import ExLlamaV2, GreedyGraphDecoder
from exllamav2_amd.synth import synth_checkpoint
from oracle.model import OracleModel

LOGIT_TOL = 0.03          # absolute, fp16 logits of magnitude ~1-4 after 2 layers (each linear rounds to fp16)


def tiny_cfg(**kw):
    d = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
             num_key_value_heads=1, head_dim=64, vocab_size=96, max_seq_len=256, max_input_len=32)
    d.update(kw)
    return ExLlamaV2Config(**d)


def build(be, cfg, recipe="4.0bpw", seed=0):
    ck = synth_checkpoint(cfg, be.device, recipe=recipe, seed=seed)
    oracle = OracleModel(cfg, ck)                       # reconstructs from the on-disk tensors BEFORE the re-layout
    model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
    return model, oracle


def check_logits(got, want):
    got = got.astype(np.float64)
    assert np.all(np.abs(got - want) <= LOGIT_TOL + np.abs(want) * 2.0 ** -8), np.abs(got - want).max()


def confident(want_logits):
    top2 = np.sort(want_logits, axis=-1)[..., -2:]
    return (top2[..., 1] - top2[..., 0]) > 4 * LOGIT_TOL


def test_prefill_then_decode_contiguous(be):
    cfg = tiny_cfg()
    model, oracle = build(be, cfg)
    cache = ExLlamaV2Cache(model, batch_size=2)
    oracle.reset(2)
    rng = np.random.default_rng(0)
    ids = rng.integers(0, cfg.vocab_size, size=(2, 5))
    logits = model.forward(torch.from_numpy(ids), cache, last_id_only=False)
    want = oracle.forward(ids)
    check_logits(be.n(logits), want)
    nxt = np.argmax(want[:, -1], axis=-1)[:, None]
    n_conf = 0
    for _ in range(2):
        logits = model.forward(torch.from_numpy(nxt), cache)
        want = oracle.forward(nxt)
        check_logits(be.n(logits), want)
        got_tok = be.n(logits)[:, -1].argmax(-1)
        conf = confident(want[:, -1])
        assert np.array_equal(got_tok[conf], np.argmax(want[:, -1], -1)[conf])
        n_conf += int(conf.sum())
        nxt = np.argmax(want[:, -1], axis=-1)[:, None]
    assert n_conf >= 1, "vacuous token check"
    model.unload()


def test_chunked_prefill_equals_oracle(be):
    cfg = tiny_cfg(max_input_len=4)                      # forces 3 chunks (model.py:873-927)
    model, oracle = build(be, cfg, seed=1)
    cache = ExLlamaV2Cache(model, batch_size=1)
    oracle.reset(1)
    ids = np.random.default_rng(1).integers(0, cfg.vocab_size, size=(1, 10))
    logits = model.forward(torch.from_numpy(ids), cache)                 # last_id_only
    want = oracle.forward(ids)[:, -1:]
    check_logits(be.n(logits), want)
    assert cache.current_seq_len == 10
    model.unload()


@pytest.mark.parametrize("kernel", ["skinny", "skinny_unfused", "tile256"])
def test_prefill_sized_forward_equals_oracle(be, kernel, monkeypatch):
    """rows > 16 through the module handles (q_attn_forward_1 / _2, q_mlp_forward_): row pre-pass + dequantize-into-MFMA GEMM,
    with the 17-128-row kernel (qgemm_skinny.hip; q | k | v and gate | up in ONE launch each, or one launch per matrix) and with the
    256-column LDS-decode kernel (qgemm_mfma.hip) forced for every row count; attention is the MFMA flash-prefill kernel
    (attn_prefill.hip); then one decode step on the cache they filled.  No torch GEMM / SDPA anywhere on this route."""
    if kernel == "tile256": monkeypatch.setenv("EXL2_PREFILL_MFMA_MIN_ROWS", "17")
    if kernel == "skinny_unfused": monkeypatch.setenv("EXL2_SKINNY_UNFUSED", "1")
    cfg = tiny_cfg(max_input_len=128, max_seq_len=256, num_hidden_layers=1)
    model, oracle = build(be, cfg, seed=3)
    cache = ExLlamaV2Cache(model, batch_size=1)
    oracle.reset(1)
    ids = np.random.default_rng(3).integers(0, cfg.vocab_size, size=(1, 70))
    logits = model.forward(torch.from_numpy(ids), cache)
    want = oracle.forward(ids)[:, -1:]
    check_logits(be.n(logits), want)
    # a second long chunk on top of the first (flash-prefill attention with past > 0: csrc/attn_prefill.hip)
    ids2 = np.random.default_rng(4).integers(0, cfg.vocab_size, size=(1, 40))
    logits = model.forward(torch.from_numpy(ids2), cache, last_id_only=False)
    check_logits(be.n(logits), oracle.forward(ids2))
    nxt = np.array([[5]])
    logits = model.forward(torch.from_numpy(nxt), cache)
    check_logits(be.n(logits), oracle.forward(nxt)[:, -1:])
    model.unload()


@pytest.mark.parametrize("recipe", ["4.0bpw", "2.5bpw", "gptq-4bit-128g"])
@pytest.mark.parametrize("batch", [17, 40])
def test_batch_decode_of_17_to_128_sequences_equals_oracle(be, recipe, batch):
    """More than 16 sequences decoding together (round 6): every module GEMM has 17-128 rows and runs on qgemm_skinny_kernel
    (q | k | v and gate | up fused per launch, K split over workgroups, deterministic ticket reduction); logits against the oracle
    on every step, and a second run of the same steps is bit-identical."""
    cfg = tiny_cfg(num_hidden_layers=2, hidden_size=256, intermediate_size=384, max_input_len=64)
    model, oracle = build(be, cfg, recipe=recipe, seed=8)
    rng = np.random.default_rng(8)
    ids0 = rng.integers(0, cfg.vocab_size, size=(batch, 1))
    runs = []
    for rep in range(2):
        cache = ExLlamaV2Cache(model, batch_size=batch)
        oracle.reset(batch)
        ids, outs = ids0, []
        for _ in range(3):
            logits = be.n(model.forward(torch.from_numpy(ids), cache))
            want = oracle.forward(ids)
            check_logits(logits, want)
            outs.append(logits.copy())
            ids = np.argmax(want[:, -1], axis=-1)[:, None]
        runs.append(outs)
    for a, b in zip(*runs):
        assert np.array_equal(a, b)
    model.unload()


def test_batch_decode_phased_route_equals_oracle(be, monkeypatch):
    """A batch of 16 sequences decoding: every fused module GEMV (RMSNorm -> q/k/v, o, RMSNorm -> gate/up, act*up -> down
    with the scatter epilogue) through the row pre-pass + phased kernel."""
    monkeypatch.setenv("EXL2_GEMV_PHASED", "1")
    monkeypatch.setenv("EXL2_GEMV_PHASE_ITEMS", "1")
    cfg = tiny_cfg(num_hidden_layers=2)
    model, oracle = build(be, cfg, seed=5)
    cache = ExLlamaV2Cache(model, batch_size=16)
    oracle.reset(16)
    rng = np.random.default_rng(5)
    ids = rng.integers(0, cfg.vocab_size, size=(16, 1))
    for _ in range(3):
        logits = model.forward(torch.from_numpy(ids), cache)
        want = oracle.forward(ids)
        check_logits(be.n(logits), want)
        ids = np.argmax(want[:, -1], axis=-1)[:, None]
    model.unload()


def test_device_side_greedy_decode_paged(be):
    """The benchmark's decode loop (GreedyGraphDecoder, eager on the emulator / graph on the GPU)."""
    cfg = tiny_cfg()
    model, oracle = build(be, cfg, seed=2)
    cache = ExLlamaV2Cache(model, batch_size=1)
    dec = GreedyGraphDecoder(model, cache, batch_size=1)
    if not be.is_emu:
        dec.capture()
    n = 4
    dec.reset(torch.tensor([7]), 0)
    dec.run(n, use_graph=not be.is_emu)
    got = be.n(dec.tokens(0, n))[0]
    oracle.reset(1)
    tok = 7
    n_conf = 0
    for i in range(n):
        want = oracle.forward(np.array([[tok]]))[0, -1]
        if confident(want[None])[0]:
            assert got[i] == int(np.argmax(want)), (i, got, np.argmax(want))
            n_conf += 1
        tok = int(got[i])                                 # follow the device's own choice: each step is checked alone
    assert n_conf >= 1, "vacuous token check"
    assert be.n(dec.cache_seqlens)[0] == n
    # the kernels index cache pages and sin/cos rows unchecked: running past the cache is refused on the host
    dec.reset(torch.tensor([7]), cache.max_seq_len - 2)
    dec.run(2, use_graph=False)
    with pytest.raises(RuntimeError, match="exceed the cache"):
        dec.run(1, use_graph=False)
    dec.free()
    model.unload()


def test_q4_cache_direct_attention_equals_unpack_route(be):
    """Decode steps on a Q4 cache: attention straight from the codes (attn_q4.hip) gives the logits of the reference's
    unpack-everything route (cache.py:472-514) up to the fp16 rounding that route applies to the unpacked values."""
    cfg = tiny_cfg(hidden_size=256, num_attention_heads=4, num_key_value_heads=4, head_dim=128,   # 512 KV elements / token
                   num_hidden_layers=1, intermediate_size=128)
    ck = synth_checkpoint(cfg, be.device, seed=5)
    ck2 = copy.deepcopy(ck)
    ma = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
    mb = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck2)
    for attn, _ in mb.layers:
        attn.q4_fused = False                                                   # reference route
    ca, cb = ExLlamaV2Cache_Q4(ma, batch_size=1), ExLlamaV2Cache_Q4(mb, batch_size=1)
    ids = torch.from_numpy(np.random.default_rng(5).integers(0, cfg.vocab_size, size=(1, 12)))
    ma.forward(ids, ca); mb.forward(ids, cb)
    for layer in range(cfg.num_hidden_layers):                                  # same codes in both caches so far?
        assert torch.equal(ca.key_states[layer][:, :12], cb.key_states[layer][:, :12])
    for tok in (5, 9):
        nxt = torch.tensor([[tok]])
        a = be.n(ma.forward(nxt, ca)).astype(np.float64)
        b = be.n(mb.forward(nxt, cb)).astype(np.float64)
        assert np.abs(a - b).max() < 3e-2, float(np.abs(a - b).max())     # the unpack route rounds every K/V element to fp16
    ma.unload(); mb.unload()


@pytest.mark.parametrize("kvh,hd,fused", [(4, 128, True), (4, 128, False), (1, 64, False), (2, 128, False)])
def test_q4_cache_model_equals_q4_oracle(be, kvh, hd, fused):
    """Prefill + decode on an ExLlamaV2Cache_Q4 through model.forward (contiguous cache, attn.py:1017-1196) against
    OracleModel.forward(q4_cache=True) = the reference's cache semantics (cache.py:472-556), including the widening of the
    touched range to whole q_blocks of tokens when a token is smaller than a 512-element codec block (kv_dim 64 -> 8 tokens,
    256 -> 2).  Both the attention-from-codes route and the unpack route; the oracle follows the device's codes from step to
    step (oracle/model.py:q4_adopt), every step at the model tolerance."""
    cfg = tiny_cfg(hidden_size=256, num_attention_heads=max(kvh, 2) if hd == 128 else 4, num_key_value_heads=kvh, head_dim=hd,
                   num_hidden_layers=2, intermediate_size=128)
    ck = synth_checkpoint(cfg, be.device, seed=7)
    oracle = OracleModel(cfg, ck)
    model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
    for attn, _ in model.layers:
        attn.q4_fused = fused
    cache = ExLlamaV2Cache_Q4(model, batch_size=2)
    oracle.reset(2)
    rng = np.random.default_rng(7)
    ids = rng.integers(0, cfg.vocab_size, size=(2, 11))
    past = 0

    def step(chunk):
        nonlocal past
        got = be.n(model.forward(torch.from_numpy(chunk), cache, last_id_only=False))
        want = oracle.forward(chunk, q4_cache=True)
        check_logits(got[..., :cfg.vocab_size], want)
        past += chunk.shape[1]
        for layer in range(cfg.num_hidden_layers):
            flipped = oracle.q4_adopt(layer, be.n(cache.key_states[layer]), be.n(cache.key_scales[layer]),
                                      be.n(cache.value_states[layer]), be.n(cache.value_scales[layer]), past)
            assert flipped < 0.01, (past, layer, flipped)

    step(ids)                                                 # prefill: 11 tokens (an odd count: partial q_blocks)
    for t in (5, 9, 3):
        step(np.array([[t], [t + 1]]))
    model.unload()


def test_q4_cache_decode_close_to_fp16(be):
    """Q4 KV path end to end (cache.py:409-606): logits stay close to the FP16-cache logits (doc/qcache_eval.md)."""
    cfg = tiny_cfg()
    ck = synth_checkpoint(cfg, be.device, seed=3)
    ck2 = copy.deepcopy(ck)
    m16 = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
    m4 = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck2)
    c16 = ExLlamaV2Cache(m16, batch_size=1)
    c4 = ExLlamaV2Cache_Q4(m4, batch_size=1)
    ids = torch.from_numpy(np.random.default_rng(3).integers(0, cfg.vocab_size, size=(1, 8)))
    a = be.n(m16.forward(ids, c16)).astype(np.float64)
    b = be.n(m4.forward(ids, c4)).astype(np.float64)
    assert np.abs(a - b).max() < 1e-2                   # prefill attends over fp16 new tokens in both
    nxt = torch.tensor([[5]])
    a = be.n(m16.forward(nxt, c16)).astype(np.float64)
    b = be.n(m4.forward(nxt, c4)).astype(np.float64)
    d = np.abs(a - b)
    assert d.max() < 0.5 and d.mean() < 0.15            # 4-bit keys/values: small but non-zero drift (head: logits of sigma ~1.5)
    assert np.any(d > 0)
    m16.unload(); m4.unload()


@pytest.mark.parametrize("recipe,act_order", [("gptq-4bit-128g", False), ("gptq-4bit-32g", True)])
def test_gptq_model_equals_oracle(be, recipe, act_order):
    """BASELINE configs[0] shape in miniature: a GQA Llama whose linears are GPTQ 4-bit (sequential and act-order
    g_idx) through the same module path -- fused attention front half, fused MLP, head -- prefill then greedy decode."""
    cfg = tiny_cfg(num_attention_heads=4, num_key_value_heads=2, intermediate_size=384)
    ck = synth_checkpoint(cfg, be.device, recipe=recipe, seed=3, act_order=act_order)
    oracle = OracleModel(cfg, ck)
    model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
    assert be.ext.q_matrix_info(model.layers[0][0].q_proj.q_handle)["is_gptq"]
    cache = ExLlamaV2Cache(model, batch_size=1)
    oracle.reset(1)
    ids = np.array([[5, 9, 77, 31]])
    logits = model.forward(torch.from_numpy(ids), cache, last_id_only=False)
    want = oracle.forward(ids)
    check_logits(be.n(logits), want)
    nxt = np.argmax(want[:, -1], axis=-1)[:, None]
    n_conf = 0
    for _ in range(2):
        logits = model.forward(torch.from_numpy(nxt), cache)
        want = oracle.forward(nxt)
        check_logits(be.n(logits), want)
        conf = confident(want[:, -1])
        assert np.array_equal(be.n(logits)[:, -1].argmax(-1)[conf], np.argmax(want[:, -1], -1)[conf])
        n_conf += int(conf.sum())
        nxt = np.argmax(want[:, -1], axis=-1)[:, None]
    assert n_conf >= 1, "vacuous token check"
    model.unload()


@pytest.mark.gpu
def test_llama2_7b_width_two_layers_equals_oracle():
    """BASELINE configs[1]'s widths (hidden 4096, intermediate 11008, 32 heads, vocab 32000) on two layers: prefill logits,
    then graph-decoded greedy tokens through the exact decoder bench.py times, against the oracle.  Product path only
    (libexl2_hip.so on cuda:0); too large for the CPU emulation backend."""
    from tests.conftest import Backend
    be = Backend("hip")
    cfg = ExLlamaV2Config.llama2_7b(max_seq_len=256, max_input_len=32)
    cfg.num_hidden_layers = 2
    ck = synth_checkpoint(cfg, be.device, recipe="4.0bpw", seed=0)
    oracle = OracleModel(cfg, ck)
    model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
    cache = ExLlamaV2Cache(model, batch_size=1)
    ids = np.array([[1, 15043, 3186, 29892, 445]])
    oracle.reset(1)
    want = oracle.forward(ids)
    logits = model.forward(torch.from_numpy(ids), cache, last_id_only=False)
    check_logits(be.n(logits), want)
    # decode 6 tokens with the device-side greedy loop; teacher-force the oracle with the device's tokens and compare
    # logits every step, token ids wherever the oracle's margin is confident
    dec = GreedyGraphDecoder(model, cache, batch_size=1).capture()
    first = int(np.argmax(want[0, -1]))
    dec.reset(torch.tensor([first]), ids.shape[1])
    dec.run(6)
    torch.cuda.synchronize()
    toks = dec.tokens(ids.shape[1], 6).cpu().numpy()[0]
    tok = first
    n_conf = 0
    for i in range(6):
        w = oracle.forward(np.array([[tok]]))[0, -1]
        if confident(w[None])[0]:
            assert toks[i] == int(np.argmax(w)), (i, toks[i], int(np.argmax(w)))
            n_conf += 1
        tok = int(toks[i])
    assert n_conf >= 4, f"only {n_conf} of 6 decode steps had a confident oracle margin (synth logit_gain)"
    # the last step's logits are still in the decoder's buffer
    check_logits(be.n(dec.logits)[:, :cfg.vocab_size].reshape(1, 1, -1), w[None, None])
    dec.free()
    model.unload()


def test_staging_scratch_goes_with_the_last_model_of_a_device(be, monkeypatch):
    """model.unload() releases the library's per-stream staging buffers (exl2_release_scratch) only when no other loaded model of
    the device is left: a second model's captured graphs may hold their addresses."""
    cfg = tiny_cfg(num_hidden_layers=1)
    calls = []
    real = be.ext.release_scratch
    monkeypatch.setattr(type(be.ext), "release_scratch", lambda self, device=None: calls.append(str(device)) or real(device))
    base = ExLlamaV2._live_on_device.get(str(torch.device(be.device)), 0)
    ma = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(synth_checkpoint(cfg, be.device, seed=1))
    mb = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(synth_checkpoint(cfg, be.device, seed=2))
    assert ExLlamaV2._live_on_device[str(torch.device(be.device))] == base + 2
    ma.unload()
    assert len(calls) == 0                                           # another model of the device is still loaded
    n_before = len(calls)
    mb.unload()
    assert ExLlamaV2._live_on_device[str(torch.device(be.device))] == base
    if base == 0:
        assert len(calls) == n_before + 1                           # the last one out released the scratch
    ma.unload()                                                      # unloading twice is harmless
    assert ExLlamaV2._live_on_device[str(torch.device(be.device))] == base
