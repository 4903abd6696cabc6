"""Device sampler (csrc/sampling.hip) against the reference's CPU sampler.

Chain of evidence: the reference's own cpp/sampling.cpp, compiled for the host and driven in sample_basic's order
(oracle/ref_build/sampling_driver.cpp), produced tests/golden/reference_sampling.npz; oracle/sampling.py restates it and must
reproduce the fixture bit for bit (tokens, probabilities, candidate counts -- including keep_threshold's n + 1 quirk); the
kernel must then sample the SAME TOKEN as the oracle for the same random number, on every row whose decisions do not hinge
on the last bits of a sum (the oracle reports how close the closest threshold comparison was; the kernel's softmax sum is a
tree sum, the reference's a sequential one)."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden_sampling as gold          # noqa: E402
from oracle import sampling as osamp         # noqa: E402

MARGIN = 2e-5                                 # decisions closer than this to a threshold are not compared (fp32 sums near 1)


def test_oracle_reproduces_the_executed_reference_fixture():
    z = np.load(os.path.join(ROOT, "tests", "golden", "reference_sampling.npz"))
    quirk = 0
    for i, (lg, flt, st, rnd) in enumerate(gold.cases()):
        tok, pr, _, nc = osamp.sample_basic(lg, st[0], st[1], st[2], st[3], rnd, flt)
        assert np.array_equal(tok, z["tokens"][i]), (i, st)
        assert np.array_equal(pr.view(np.uint32), z["probs"][i].view(np.uint32)), (i, st)
        assert np.array_equal(nc, z["num_candidates"][i]), (i, st)
        quirk += int((nc > st[1]).sum())
    # the fixture exercises keep_threshold's "every entry passes -> n + 1 entries" path (cpp/sampling.cpp:569-592)
    assert quirk >= 1


def test_oracle_reproduces_the_live_reference():
    if not os.path.exists(gold.LIB):
        pytest.skip("oracle/_ref/libsampling_ref.so not built (needs /root/reference at build time)")
    lib = gold.load()
    rng = np.random.default_rng(5)
    for trial in range(12):
        v = int(rng.integers(40, 700))
        lg = np.ascontiguousarray((rng.standard_normal((2, v)) * rng.uniform(0.5, 5)).astype(np.float16), dtype=np.float32)
        st = (float(rng.uniform(0.3, 1.6)), int(rng.integers(1, min(v - 1, 300))), float(rng.choice([0.0, 0.4, 0.9])),
              float(rng.choice([0.0, 0.02, 0.2])))
        rnd = float(rng.random())
        tok, pr, nc = gold.run_reference(lib, lg, None, st, rnd, 0)
        t2, p2, _, n2 = osamp.sample_basic(lg, st[0], st[1], st[2], st[3], rnd, None)
        assert np.array_equal(tok, t2) and np.array_equal(pr.view(np.uint32), p2.view(np.uint32)) and np.array_equal(nc, n2), (trial, st)


def _run_kernel(be, lg, flt, st, rnd, dtype=np.float32):
    torch = be.torch
    rows = lg.shape[0]
    out_t = torch.zeros(rows, dtype=torch.int32, device=be.device)
    out_p = torch.zeros(rows, dtype=torch.float32, device=be.device)
    f = None if flt is None else be.t(flt.astype(np.uint8))
    be.ext.sample_rows(be.t(lg.astype(dtype)), st[0], st[1], st[2], st[3], rnd, out_t, out_p, logit_filter=f)
    return be.n(out_t), be.n(out_p)


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_sample_rows_equals_oracle_on_the_fixture_cases(be, dtype):
    compared = total = 0
    for i, (lg, flt, st, rnd) in enumerate(gold.cases()):
        tok, pr, margin, _ = osamp.sample_basic(lg, st[0], st[1], st[2], st[3], rnd, flt)
        got_t, got_p = _run_kernel(be, lg, flt, st, rnd, dtype)
        for r in range(lg.shape[0]):
            total += 1
            if margin[r] < MARGIN:
                continue
            compared += 1
            assert got_t[r] == tok[r], (i, r, st, got_t, tok, margin)
            assert abs(float(got_p[r]) - float(pr[r])) <= 2e-5 * max(1.0, float(pr[r])) + 1e-6, (i, r, st, got_p, pr)
    assert compared >= 0.8 * total, (compared, total)


def test_sample_rows_random_sweep_ties_and_the_extra_candidate(be):
    """One row, coarse logits (exact ties across the top-k boundary), a min-p low enough that every candidate passes (the
    reference then samples from k + 1 entries), swept over the whole range of the random point."""
    rng = np.random.default_rng(21)
    lg = (np.round(rng.standard_normal((1, 300)) * 8) / 4).astype(np.float32)
    compared = 0
    seen = set()
    for st in [(1.5, 12, 0.0, 0.001), (1.0, 30, 0.0, 0.0), (2.5, 7, 0.97, 0.001), (1.0, 64, 0.6, 0.0)]:
        for rnd in np.linspace(0.0, 0.99999, 41):
            tok, pr, margin, nc = osamp.sample_basic(lg, st[0], st[1], st[2], st[3], float(rnd), None)
            got_t, got_p = _run_kernel(be, lg, None, st, float(rnd))
            if margin[0] < MARGIN:
                continue
            compared += 1
            seen.add((st[1], int(nc[0])))
            assert got_t[0] == tok[0], (st, rnd, got_t, tok, nc, margin)
    assert compared > 120
    assert (12, 13) in seen                   # keep_threshold handed back k + 1 entries and the kernel followed


def test_sample_rows_full_vocabulary_fp16_logits(be):
    rng = np.random.default_rng(8)
    lg = (rng.standard_normal((2, 32000)) * 2.5).astype(np.float16)
    for st, rnd in [((0.8, 50, 0.8, 0.0), 0.37), ((1.0, 500, 0.95, 0.01), 0.81)]:
        tok, pr, margin, _ = osamp.sample_basic(lg, st[0], st[1], st[2], st[3], rnd, None)
        got_t, got_p = _run_kernel(be, lg, None, st, rnd, np.float16)
        for r in range(2):
            if margin[r] >= MARGIN:
                assert got_t[r] == tok[r], (st, r, got_t, tok)


def test_sample_rows_greedy_is_the_first_maximum(be):
    lg = np.zeros((2, 900), dtype=np.float32)
    lg[0, 17] = lg[0, 400] = 5.0
    lg[1, 3] = 2.0
    got_t, got_p = _run_kernel(be, lg, None, (0.0, 50, 0.8, 0.0), 0.5)       # temperature < 0.01 -> greedy (ext_sampling.cpp:143-147)
    assert got_t.tolist() == [17, 3]
    flt = np.ones((2, 900), dtype=bool); flt[0, 17] = False
    got_t, _ = _run_kernel(be, lg, flt, (1.0, 1, 0.0, 0.0), 0.5)
    assert got_t.tolist() == [400, 3]


def test_sample_rows_refuses_what_is_not_built(be):
    torch = be.torch
    lg = be.t(np.zeros((1, 64), dtype=np.float32))
    t, p = torch.zeros(1, dtype=torch.int32, device=be.device), torch.zeros(1, dtype=torch.float32, device=be.device)
    for k in (0, 501, 64):
        with pytest.raises(RuntimeError):
            be.ext.sample_rows(lg, 1.0, k, 0.0, 0.0, 0.5, t, p)
    with pytest.raises(RuntimeError):
        be.ext.sample_rows(lg, 1.0, 10, 0.0, 0.0, 1.5, t, p)


def test_row_in_registers_equals_memory_walks(be, monkeypatch):
    """sample_rows with the row held in registers (default where it applies: quad-aligned rows up to 32768 entries, no filter)
    against the same kernel walking memory for every stage (EXL2_SAMPLE_REG=0): tokens, probabilities and the whole workspace of
    probabilities identical bit for bit -- fp16 and fp32 logits, 1 and 3 rows, a 32 000-entry vocabulary, tie-heavy logits.  Round 4:
    the register route finds theta by a quick select (lower bound from the per-thread maxima, the entries above it compared among
    themselves) in front of the radix passes (EXL2_SAMPLE_QUICK=0): all three routes agree bit for bit."""
    import torch
    rng = np.random.default_rng(11)
    for vocab, rows, dt, coarse in ((96, 3, np.float16, False), (4096, 1, np.float32, False), (32000, 3, np.float16, False), (2048, 2, np.float16, True)):
        lg = (rng.standard_normal((rows, vocab)) * 3).astype(np.float32)
        if coarse: lg = np.round(lg * 2) / 2                          # exact ties at the top-k boundary
        lg = lg.astype(dt)
        out = []
        for reg, quick in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
            # (registers / memory walks x quick select / radix passes: the default is the first, round 3's kernel the last)
            monkeypatch.setenv("EXL2_SAMPLE_REG", reg)
            monkeypatch.setenv("EXL2_SAMPLE_QUICK", quick)
            t = torch.zeros((rows,), dtype=torch.int32, device=be.device); p = torch.zeros((rows,), dtype=torch.float32, device=be.device)
            ws = be.ext.sample_rows(be.t(lg), 0.8, 50, 0.8, 0.02, 0.37, t, p)
            out.append((be.n(t).copy(), be.n(p).view(np.uint32).copy(), be.n(ws).view(np.uint32).copy()))
        for other in out[1:]:
            assert np.array_equal(out[0][0], other[0]) and np.array_equal(out[0][1], other[1]) and np.array_equal(out[0][2], other[2]), (vocab, rows)


def test_dropin_apply_rep_penalty_reproduces_the_executed_reference():
    """dropin apply_rep_penalty (host helper of the sampler, ext_sampling.cpp:32-72) against the reference's
    apply_rep_penalty_cpu run on the host: bit for bit, every setting of the fixture (decay ramps, ids outside the vocabulary)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    try:
        import exllamav2_ext as dropin
    except (OSError, RuntimeError) as e:                      # the drop-in binds libexl2_hip.so at import
        pytest.skip(f"drop-in not importable here: {e}")
    z = np.load(os.path.join(ROOT, "tests", "golden", "reference_sampling.npz"))
    seq, lg = gold.rep_cases()
    for i, st in enumerate(gold.REP_SETTINGS):
        got = torch.from_numpy(lg.copy()).view(2, 1, -1)
        dropin.apply_rep_penalty(torch.from_numpy(seq), st[0], st[1], st[2], st[3], st[4], got)
        assert np.array_equal(got.view(2, -1).numpy().view(np.uint32), z["rep_penalty_logits"][i].view(np.uint32)), st


@pytest.mark.gpu
def test_dropin_sample_basic_with_the_reference_default_settings():
    """ext_c.sample_basic as ExLlamaV2Sampler.sample calls it (sampler.py:540-568: fp32 logits [bsz, 1, vocab] on the host,
    int64 / fp32 outputs on the host, `none_tensor` on the meta device for the unused ones) with the reference's default
    Settings (temperature 0.8, top_k 50, top_p 0.8; sampler.py:54-69) after its default repetition penalty."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    import exllamav2_ext as dropin
    rng = np.random.default_rng(3)
    none = torch.empty((1, 1), device="meta")
    ok = 0
    for trial in range(6):
        lg = (rng.standard_normal((2, 4096)) * 3).astype(np.float16).astype(np.float32)
        seq = rng.integers(0, 4096, size=(2, 20)).astype(np.int64)
        logits = torch.from_numpy(lg.copy()).view(2, 1, -1)
        dropin.apply_rep_penalty(torch.from_numpy(seq), 1.025, -1, 0, 0.0, 0.0, logits)
        rnd = float(rng.random())
        out_t = torch.empty((2, 1), dtype=torch.long)
        out_p = torch.empty((2, 1), dtype=torch.float)
        dropin.sample_basic(logits, 0.8, 50, 0.8, 0.0, 0.0, 0.0, 0.0, rnd, out_t, out_p, none, none, none, False, [], 1.5, 0.1,
                            1.0, none, 0.0, 0.1, 0.0, 0.0, 1.0, 0.0, 0.0)
        tok, pr, margin, _ = osamp.sample_basic(logits.view(2, -1).numpy(), 0.8, 50, 0.8, 0.0, rnd, None)
        for r in range(2):
            if margin[r] >= MARGIN:
                assert int(out_t[r, 0]) == int(tok[r]), (trial, r, out_t, tok)
                ok += 1
    assert ok >= 8
    with pytest.raises(NotImplementedError):
        dropin.sample_basic(logits, 0.8, 50, 0.8, 0.0, 0.0, 0.5, 0.0, 0.5, out_t, out_p, none, none, none, False, [], 1.5, 0.1,
                            1.0, none, 0.0, 0.1, 0.0, 0.0, 1.0, 0.0, 0.0)       # tfs


def test_sample_rows_row_lengths_off_the_quad_path(be):
    """Vocabulary sizes that are not multiples of 4 (and a row stride wider than the vocabulary) take the element-wise row
    walk of the kernel instead of the quad one: same tokens."""
    rng = np.random.default_rng(31)
    compared = 0
    for v, ld in ((997, 997), (1001, 1008), (130, 131)):
        lg = np.zeros((2, ld), dtype=np.float32)
        lg[:, :v] = (rng.standard_normal((2, v)) * 3).astype(np.float16)
        lg[:, v:] = 50.0                                             # padding columns must never be looked at
        for st, rnd in (((0.8, 50, 0.8, 0.0), 0.41), ((1.2, 9, 0.0, 0.05), 0.77)):
            tok, pr, margin, _ = osamp.sample_basic(lg[:, :v], st[0], st[1], st[2], st[3], rnd, None)
            torch = be.torch
            out_t = torch.zeros(2, dtype=torch.int32, device=be.device)
            out_p = torch.zeros(2, dtype=torch.float32, device=be.device)
            be.ext.sample_rows(be.t(lg), st[0], st[1], st[2], st[3], rnd, out_t, out_p, vocab=v)
            got = be.n(out_t)
            for r in range(2):
                if margin[r] >= MARGIN:
                    assert got[r] == tok[r], (v, ld, st, r, got, tok)
                    compared += 1
    assert compared >= 8


def test_decoder_run_sampled(be):
    """GreedyGraphDecoder.run_sampled: the device sampler inside the decode loop (token fed back on the device, token log,
    no host read-back).  (1) with top_k = 1 it must reproduce run()'s greedy tokens exactly; (2) with the reference's default
    settings every token must be the one the oracle's sampler draws from THAT step's device logits with that step's random
    point, for both rows of the batch (second row: the reference's recurrence of the point)."""
    import torch
    from exllamav2_amd.cache import ExLlamaV2Cache
    from exllamav2_amd.config import ExLlamaV2Config
    from exllamav2_amd.model import ExLlamaV2, GreedyGraphDecoder
    from exllamav2_amd.synth import synth_checkpoint
    cfg = ExLlamaV2Config(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                          num_key_value_heads=1, head_dim=64, vocab_size=96, max_seq_len=256, max_input_len=32)
    model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(synth_checkpoint(cfg, be.device, seed=3))
    graph = be.device != "cpu"

    def decoder():
        cache = ExLlamaV2Cache(model, batch_size=2, max_seq_len=256)
        dec = GreedyGraphDecoder(model, cache, batch_size=2)
        if graph:
            dec.capture()
        dec.reset(torch.tensor([5, 17]), 0)
        return dec

    dec = decoder(); dec.run(5, use_graph=graph)
    greedy = be.n(dec.tokens(0, 5)).copy(); dec.free()
    dec = decoder(); dec.run_sampled(5, 1.0, 1, use_graph=graph)
    assert np.array_equal(be.n(dec.tokens(0, 5)), greedy)
    dec.free()

    dec = decoder()
    rnds = [0.13, 0.58, 0.91, 0.34, 0.77, 0.05]
    compared = 0
    for i, rnd in enumerate(rnds):
        dec.run_sampled(1, 0.8, 50, 0.8, 0.0, randoms=[rnd], use_graph=graph)
        lg = be.n(dec.logits)[:, :cfg.vocab_size].astype(np.float32)
        tok, _, margin, _ = osamp.sample_basic(lg, 0.8, 50, 0.8, 0.0, rnd, None)
        got = be.n(dec.tokens(i, 1))[:, 0]
        assert np.array_equal(be.n(dec.ids), got)                   # the logged token is the one fed to the next step
        for r in range(2):
            if margin[r] >= MARGIN:
                assert got[r] == tok[r], (i, r, got, tok)
                compared += 1
    assert compared >= 8
    dec.free()

    # (3) the sampler INSIDE the step (capture_sampled: exl2_sample_rows_step reads the point from a device buffer, logs the token,
    # advances the position; one graph launch per token on the GPU): same tokens as (2)'s route step by step, and several
    # tokens per call consume consecutive points
    dec = decoder()
    dec.capture_sampled(0.8, 50, 0.8, 0.0, n_randoms=8)
    compared = 0
    for i, rnd in enumerate(rnds):
        dec.run_sampled(1, 0.8, 50, 0.8, 0.0, randoms=[rnd], use_graph=graph)
        lg = be.n(dec.logits)[:, :cfg.vocab_size].astype(np.float32)
        tok, _, margin, _ = osamp.sample_basic(lg, 0.8, 50, 0.8, 0.0, rnd, None)
        got = be.n(dec.tokens(i, 1))[:, 0]
        assert np.array_equal(be.n(dec.ids), got)
        for r in range(2):
            if margin[r] >= MARGIN:
                assert got[r] == tok[r], (i, r, got, tok)
                compared += 1
    assert compared >= 8
    one_by_one = be.n(dec.tokens(0, len(rnds))).copy()
    dec.free()
    dec = decoder()
    dec.capture_sampled(0.8, 50, 0.8, 0.0, n_randoms=8)
    dec.run_sampled(len(rnds), 0.8, 50, 0.8, 0.0, randoms=rnds, use_graph=graph)
    assert np.array_equal(be.n(dec.tokens(0, len(rnds))), one_by_one)
    dec.free()
    model.unload()
