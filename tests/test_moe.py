"""MoE expert path: router (softmax -> top-k -> renormalise) and the fused q_moe_mlp_forward_
against a composition of oracle pieces (oracle.modules.moe_route / rms_norm / silu_mul + reconstructed experts).
Reference semantics: QMoEMLP::forward_ (cuda/q_mlp.cu:318-402), routing arithmetic cuda/q_mlp_softmax.cuh."""
import os

import numpy as np
import pytest
import torch

from oracle import exl2 as OX
from oracle import modules as OM
from tests.util import make_exl2

F16 = np.float16


@pytest.mark.parametrize("experts,topk", [(8, 2), (4, 1), (16, 4)])
def test_moe_route(be, experts, topk):
    rng = np.random.default_rng(30)
    rows, hidden = 5, 128
    x = rng.standard_normal((rows, hidden)).astype(F16)
    gate = (rng.standard_normal((experts, hidden)) * 0.2).astype(F16)
    logits = torch.zeros((rows, experts), dtype=torch.float16, device=be.device)
    be.ext.moe_route(be.t(x), be.t(gate), logits, topk)
    raw = (x.astype(np.float32) @ gate.astype(np.float32).T).astype(F16)
    want, mask = OM.moe_route(raw, topk)
    got = be.n(logits)
    assert np.array_equal(got != 0, mask)                                    # same experts selected
    assert np.all(np.abs(got.astype(np.float32) - want.astype(np.float32)) <= 2e-3)
    assert np.allclose(got.astype(np.float32).sum(-1), 1.0, atol=2e-3)


def _moe_case(be, rows, shared_perm, E=8, topk=2, hidden=128, inter=256, seed=31, spec_up=None, spec_dn=None, gate_scale=0.3):
    """one sparse-MoE block (q_moe_mlp_forward_) against the oracle's composition: routing mask, weighted expert sum into the residual"""
    from tests.util import exl2_to_torch
    rng = np.random.default_rng(seed)
    spec_up = spec_up or [(5, 32, 32), (4, 32, hidden - 32)]
    spec_dn = spec_dn or [(5, 32, 64), (4, 64, inter - 64)]
    keep, handles, refs = [], {"w1": [], "w2": [], "w3": []}, {"w1": [], "w2": [], "w3": []}
    shared = rng.permutation(hidden).astype(np.int32)
    for e in range(E):
        for name, (k, n, spec) in (("w1", (hidden, inter, spec_up)), ("w3", (hidden, inter, spec_up)), ("w2", (inter, hidden, spec_dn))):
            t = OX.synth_exl2(k, n, spec, seed=100 + 3 * e + len(name) + ord(name[1]), act_order=True)
            if shared_perm and name != "w2":
                t["q_invperm"] = shared.copy()
            ref = OX.exl2_reconstruct(t)
            w = exl2_to_torch(be, t)
            h = be.ext.make_q_matrix_from_dict(w, None)
            keep.append(w); handles[name].append(h); refs[name].append(ref)
    norm_w = (1.0 + 0.1 * rng.standard_normal(hidden)).astype(F16)
    gate = (rng.standard_normal((E, hidden)) * gate_scale).astype(F16)
    x = rng.standard_normal((rows, hidden)).astype(F16)

    max_rows = 128                                        # room for the grouped route's per-expert scratch (E * rows rows)
    dev = be.device
    ts = torch.zeros((max_rows, hidden), dtype=torch.float16, device=dev)
    ta = torch.zeros((max_rows, inter), dtype=torch.float16, device=dev)
    tb = torch.zeros((max_rows, inter), dtype=torch.float16, device=dev)
    tl = torch.zeros((max_rows, E), dtype=torch.float16, device=dev)
    from exllamav2_amd.ext import none_tensor
    nw, gt = be.t(norm_w), be.t(gate)
    moe = be.ext.make_q_moe_mlp(nw, none_tensor, True, 1e-5, gt, E, topk, handles["w1"], handles["w2"], handles["w3"],
                                ts, none_tensor, ta, tb, tl, none_tensor, max_rows, False)
    xt = be.t(x)
    be.ext.q_moe_mlp_forward_(moe, xt)
    got = be.n(xt).astype(np.float64)

    # oracle composition
    xn = OM.rms_norm(x, norm_w, 1e-5)
    raw = (xn.astype(np.float32) @ gate.astype(np.float32).T).astype(F16)
    wts, mask = OM.moe_route(raw, topk)
    chosen = be.n(tl)[:rows] != 0
    # rows whose k-th and (k + 1)-th router logits are within a few fp16 steps of each other may legitimately go to either expert (the
    # device sums the router's dot product in another order than the oracle): they are not compared (bench.py's parity check does the same)
    srt = np.sort(raw.astype(np.float32), axis=-1)[:, ::-1]
    clear = np.ones((rows,), dtype=bool) if topk >= E else (srt[:, topk - 1] - srt[:, topk]) > 4 * np.maximum(np.abs(srt[:, topk - 1]), 1.0) * 2.0 ** -10
    # (a peaked router's second / third weights underflow to 0 in fp16: "nonzero" then marks fewer experts than were chosen)
    tiny = wts.astype(np.float32) < 1e-4
    assert not np.any((chosen & ~mask)[clear]) and np.array_equal((chosen | tiny)[clear], (mask | tiny)[clear])
    want = x.astype(np.float64).copy()
    acc16 = x.copy()
    for e in range(E):
        sel = np.nonzero(mask[:, e])[0]
        if len(sel) == 0: continue
        g = OX.gemm_ref(xn[sel], refs["w1"][e], exact=True).astype(F16)
        u = OX.gemm_ref(xn[sel], refs["w3"][e], exact=True).astype(F16)
        a = OM.silu_mul(g, u)
        d = OX.gemm_ref(a, refs["w2"][e], exact=True)
        want[sel] += d * wts[sel, e].astype(np.float64)[:, None]
    tol = np.abs(want) * 2.0 ** -8 + 6e-3                   # E sequential fp16 accumulations into the residual
    assert np.all((np.abs(got - want) <= tol)[clear]), float(np.abs(got - want)[clear].max())
    assert np.all(np.isfinite(got))
    be.ext.free_q_moe_mlp(moe)
    for hs in handles.values():
        for h in hs: be.ext.free_q_matrix(h)

@pytest.mark.parametrize("route", ["default", "batched"])
@pytest.mark.parametrize("shared_perm", [True, False])
@pytest.mark.parametrize("rows", [1, 3, 16, 21])
def test_moe_mlp_forward(be, rows, shared_perm, route, monkeypatch):
    """Every expert's kernels see all rows; rows not routed to it are skipped, launches with no routed row exit.
    rows = 16 is BASELINE config 5's decode batch (the reference falls back to a torch loop above 4 rows).
    shared_perm: every expert's w1 / w3 carry ONE act-order permutation, as the quantizer writes them
    (conversion/quantize.py:190-192) -> the grouped route (all experts in one launch per projection stage) for rows <= 16;
    per-matrix permutations (format-legal, never produced) -> the per-expert launch loop."""
    if route == "batched":
        # round 5: what a Mixtral layer takes at 5..16 rows (the grouped launch declines its 16 x 14336 down_proj rows): experts on the
        # streaming / phased kernels, rows gathered once, 2 (gate | up) / 4 (down) experts per launch, outputs summed by the combine
        if not shared_perm or rows > 16:
            pytest.skip("the batched route needs the experts' shared permutation and <= 16 rows")
        monkeypatch.setenv("EXL2_MOE_NO_GROUP", "1")
        monkeypatch.setenv("EXL2_DEBUG_ROUTE", "1")
    _moe_case(be, rows, shared_perm)



def test_moe_model_forward(be):
    """Mixtral-style model through the ordinary ExLlamaV2 loop (config.num_experts > 0): prefill + decode run, logits are
    finite, two identical sequences in a batch produce identical rows, and the sparse MLP of layer 0 applied by hand to
    the same input equals what the model computed (the model adds nothing around the module)."""
    from exllamav2_amd.config import ExLlamaV2Config
    from exllamav2_amd.synth import synth_checkpoint
    from exllamav2_amd.model import ExLlamaV2
    from exllamav2_amd.cache import ExLlamaV2Cache
    cfg = ExLlamaV2Config(hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
                          num_key_value_heads=1, head_dim=64, vocab_size=96, max_seq_len=256, max_input_len=32,
                          max_batch_size=2, num_experts=4, num_experts_per_token=2, arch="mixtral")
    ck = synth_checkpoint(cfg, be.device, recipe="3.5bpw", seed=7)
    model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
    cache = ExLlamaV2Cache(model, batch_size=2)
    ids = torch.from_numpy(np.random.default_rng(7).integers(0, cfg.vocab_size, size=(1, 6))).repeat(2, 1)
    logits = be.n(model.forward(ids, cache))
    assert np.all(np.isfinite(logits)) and np.array_equal(logits[0], logits[1])
    nxt = torch.tensor([[5], [5]])
    step = be.n(model.forward(nxt, cache))
    assert np.all(np.isfinite(step)) and np.array_equal(step[0], step[1])
    attn, moe = model.layers[0]
    x = torch.from_numpy(np.random.default_rng(8).standard_normal((2, 1, cfg.hidden_size)).astype(np.float16)).to(be.device)
    y1 = x.clone(); moe.forward(y1)
    y2 = x.clone(); be.ext.q_moe_mlp_forward_(moe.q_handle, y2.view(2, -1))
    assert torch.equal(y1, y2)
    # bytes one token streams: attention + head + router + num_experts_per_token of the experts
    dense = model.lm_head.weight_bytes() + sum(l.weight_bytes() for l in (attn.q_proj, attn.k_proj, attn.v_proj, attn.o_proj))
    experts = sum(l.weight_bytes() for l in moe.w1 + moe.w2 + moe.w3)
    assert model.weight_bytes() == dense + experts // 4 * 2 + moe.gate.numel() * 2
    model.unload()



@pytest.mark.parametrize("batch", [1, 3, 16])
def test_moe_model_equals_oracle(be, batch):
    """A Mixtral-style model (8 experts, top 2, 3.5 bpw mixed widths) against OracleModel's restatement of the reference's torch
    route (moe_mlp.py:255-323): prompt logits through model.forward, then decode steps of `batch` different sequences through the
    graph decoder (batch 16 = BASELINE configs[4]'s row count: the grouped-expert launches).  Sequences whose expert selection
    was a near tie in the oracle are skipped (top-k is discontinuous) -- and must be few."""
    from exllamav2_amd.config import ExLlamaV2Config
    from exllamav2_amd.synth import synth_checkpoint
    from exllamav2_amd.model import ExLlamaV2, GreedyGraphDecoder
    from exllamav2_amd.cache import ExLlamaV2Cache
    from oracle.model import OracleModel
    cfg = ExLlamaV2Config(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                          num_key_value_heads=1, head_dim=64, vocab_size=96, max_seq_len=256, max_input_len=32,
                          max_batch_size=16, num_experts=8, num_experts_per_token=2, arch="mixtral")
    ck = synth_checkpoint(cfg, be.device, recipe="3.5bpw", seed=11)
    oracle = OracleModel(cfg, ck)
    model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
    cache = ExLlamaV2Cache(model, batch_size=batch, max_seq_len=256)
    ids = (np.array([[5, 17, 42, 7]]) + 13 * np.arange(batch)[:, None]) % cfg.vocab_size
    oracle.reset(batch)
    skipped, compared = 0, 0

    def compare(got, want):
        nonlocal skipped, compared
        ok = oracle.router_margin > 2e-3
        skipped += int((~ok).sum()); compared += int(ok.sum())
        err, tol = np.abs(got - want)[ok], (0.03 + np.abs(want) * 2.0 ** -8)[ok]
        assert np.all(err <= tol), float((err / tol).max())

    want = oracle.forward(ids)[:, -1]
    compare(be.n(model.forward(torch.from_numpy(ids), cache)).astype(np.float64)[:, -1], want)
    tok = want.argmax(-1).astype(np.int64)
    dec = GreedyGraphDecoder(model, cache, batch_size=batch)
    assert dec.chain is not None                # (round 6: the attention half of a MoE layer is chained, the block publishes its result)
    if not be.is_emu:
        dec.capture()
    dec.reset(torch.from_numpy(tok), ids.shape[1])
    for step in range(4):
        dec.run(1, use_graph=not be.is_emu)
        assert dec.chain is not None
        want = oracle.forward(tok[:, None])[:, -1]
        compare(be.n(dec.logits).astype(np.float64)[:, :cfg.vocab_size], want)
        tok = be.n(dec.tokens(ids.shape[1] + step, 1))[:, 0].astype(np.int64)
    assert compared >= 4 * batch and skipped <= max(1, compared // 5), (compared, skipped)
    dec.free()
    model.unload()


@pytest.mark.gpu
def test_moe_layer_mixtral_widths():
    """BASELINE configs[4] at its real widths: ONE Mixtral-8x7B sparse-MoE block (hidden 4096, intermediate 14336, 8 experts,
    top 2, the 3.5 bpw recipe's mixed 4/3-bit matrices), rows 1 / 4 / 16 (single launches per expert, the <= 4-row fused form,
    the grouped-expert route) against OracleModel.moe_mlp (moe_mlp.py:255-323 semantics).  ~1 minute of host time for the 24
    reconstructs."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import dataclasses
    from exllamav2_amd.config import ExLlamaV2Config
    from exllamav2_amd.synth import synth_checkpoint
    from exllamav2_amd.model import ExLlamaV2
    from oracle.model import OracleModel
    from oracle import modules as OMod
    cfg = dataclasses.replace(ExLlamaV2Config.mixtral_8x7b(max_seq_len=256, max_input_len=16, max_batch_size=16),
                              num_hidden_layers=1, vocab_size=512)
    ck = synth_checkpoint(cfg, "cuda:0", recipe="3.5bpw", seed=3)
    oracle = OracleModel(cfg, ck)
    model = ExLlamaV2(cfg, device="cuda:0").load(ck)
    moe = model.layers[0][1]
    rng = np.random.default_rng(5)
    worst = 0.0
    for rows in (1, 4, 16):
        x = rng.standard_normal((rows, cfg.hidden_size)).astype(F16)
        n = OMod.rms_norm(x, oracle.w["model.layers.0.post_attention_layernorm"], cfg.norm_eps)
        oracle.router_margin = np.full((rows,), np.inf)
        want = oracle.moe_mlp(x, n, "model.layers.0").astype(np.float64)
        ok = oracle.router_margin > 2e-3
        xt = torch.from_numpy(x).to("cuda:0").view(rows, 1, -1).contiguous()
        moe.forward(xt)
        torch.cuda.synchronize()
        got = xt.view(rows, -1).float().cpu().numpy().astype(np.float64)
        assert np.all(np.isfinite(got))
        err, tol = np.abs(got - want)[ok], (np.abs(want) * 2.0 ** -8 + 8e-3)[ok]
        assert ok.sum() >= max(1, rows - 2), "too many router near-ties"
        assert np.all(err <= tol), (rows, float(err.max()), float((err / tol).max()))
        # the update itself (not only the residual passed through): the MoE output has a visible magnitude
        assert np.abs(want - x.astype(np.float64)).mean() > 0.02
        worst = max(worst, float((err / tol).max()))
    print(f"mixtral-width MoE block: worst error / tolerance = {worst:.3f}")
    model.unload()


@pytest.mark.parametrize("rows", [1, 4, 16])
def test_moe_fused_front_is_bit_identical(be, rows, monkeypatch):
    """Round 5: norm + router logits + top-k + row gather in ONE launch (csrc/moe.hip: moe_front_kernel) keeps the arithmetic and
    the summation order of the four kernels it replaces -- same routing weights, same block output, bit for bit
    (EXL2_MOE_UNFUSED_FRONT=1 = the separate kernels)."""
    from exllamav2_amd.config import ExLlamaV2Config
    from exllamav2_amd.synth import synth_checkpoint
    from exllamav2_amd.model import ExLlamaV2
    cfg = ExLlamaV2Config(hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=4,
                          num_key_value_heads=2, head_dim=64, vocab_size=96, max_seq_len=256, max_input_len=32,
                          max_batch_size=16, num_experts=8, num_experts_per_token=2, arch="mixtral")
    ck = synth_checkpoint(cfg, be.device, recipe="3.5bpw", seed=21)
    model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
    moe = model.layers[0][1]
    x = torch.from_numpy(np.random.default_rng(rows).standard_normal((rows, 1, cfg.hidden_size)).astype(np.float16)).to(be.device)
    outs = []
    monkeypatch.setenv("EXL2_MOE_NO_LEAN", "1")            # (one row: the same expert kernels behind both fronts -- the grouped launches)
    for unfused in ("0", "1"):
        if unfused == "1": monkeypatch.setenv("EXL2_MOE_UNFUSED_FRONT", "1")
        y = x.clone()
        moe.forward(y)
        outs.append((be.n(y).copy(), be.n(moe.temp_logits[:rows]).copy()))
    assert np.array_equal(outs[0][1].view(np.uint16), outs[1][1].view(np.uint16))      # routing weights
    assert np.array_equal(outs[0][0].view(np.uint16), outs[1][0].view(np.uint16))      # block output
    model.unload()


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_moe_one_row_runs_the_selected_experts_on_the_lean_kernel(be, seed, capfd, monkeypatch):
    """Round 6: at ONE row the two selected experts run on the chained decode kernel (csrc/qgemv_lean.hip: qgemv_lean_moe_kernel) --
    every expert's argument block is planned at load time, the front kernel copies the selected experts' blocks to the slots the
    gate|up and the down launch read, the down launch multiplies by the routing weight (q_mlp.cu:373-384).  Checked: the route is
    taken; the routing weights are bit-identical to the grouped route's; the block output is within the fp16 bar of the float64
    oracle AND of the grouped route (other kernels, other summation order); a second call with another input (other experts
    selected) through the SAME handle is right as well -- the copied blocks are per step, not per handle."""
    from exllamav2_amd.config import ExLlamaV2Config
    from exllamav2_amd.synth import synth_checkpoint
    from exllamav2_amd.model import ExLlamaV2
    from oracle.model import OracleModel
    rng = np.random.default_rng(900 + seed)
    hidden, inter = int(rng.choice([256, 512])), int(rng.choice([512, 768, 1024]))
    cfg = ExLlamaV2Config(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=1, num_attention_heads=4,
                          num_key_value_heads=2, head_dim=64, vocab_size=96, max_seq_len=256, max_input_len=32,
                          max_batch_size=16, num_experts=8, num_experts_per_token=2, arch="mixtral")
    ck = synth_checkpoint(cfg, be.device, recipe=str(rng.choice(["3.5bpw", "4.0bpw", "2.5bpw"])), seed=40 + seed)
    oracle = OracleModel(cfg, ck)
    model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
    moe = model.layers[0][1]
    picked = set()
    for call in range(3):
        x = torch.from_numpy((rng.standard_normal((1, 1, hidden)) * 0.7).astype(np.float16)).to(be.device)
        monkeypatch.setenv("EXL2_DEBUG_ROUTE", "1")
        y = x.clone(); moe.forward(y)
        monkeypatch.delenv("EXL2_DEBUG_ROUTE")
        assert "route: lean (down pair sums) rows=1" in capfd.readouterr().err      # (top-2: one pair launch adds both experts' weighted outputs to x)
        w_lean = be.n(moe.temp_logits[:1]).copy()
        monkeypatch.setenv("EXL2_MOE_NO_SUM", "1"); monkeypatch.setenv("EXL2_DEBUG_ROUTE", "1")
        y2 = x.clone(); moe.forward(y2)                                               # (the two down launches + the combine launch)
        monkeypatch.delenv("EXL2_MOE_NO_SUM"); monkeypatch.delenv("EXL2_DEBUG_ROUTE")
        assert "route: lean rows=1" in capfd.readouterr().err
        assert np.array_equal(w_lean.view(np.uint16), be.n(moe.temp_logits[:1]).view(np.uint16))
        monkeypatch.setenv("EXL2_MOE_NO_LEAN", "1")
        z = x.clone(); moe.forward(z)
        monkeypatch.delenv("EXL2_MOE_NO_LEAN")
        assert np.array_equal(w_lean.view(np.uint16), be.n(moe.temp_logits[:1]).view(np.uint16))
        picked.add(tuple(np.nonzero(w_lean[0])[0].tolist()))
        xh = be.n(x).reshape(1, hidden)
        pfx = "model.layers.0"
        oracle.router_margin = np.full((1,), np.inf)
        want = oracle.moe_mlp(xh, OM.rms_norm(xh, oracle.w[pfx + ".post_attention_layernorm"], cfg.norm_eps), pfx).astype(np.float64)
        if oracle.router_margin[0] <= 2e-3: want = None       # (a near tie in the oracle's router: another expert may be selected)
        other = be.n(z).astype(np.float64).reshape(1, hidden)
        for got in (be.n(y).astype(np.float64).reshape(1, hidden), be.n(y2).astype(np.float64).reshape(1, hidden)):
            tol = 0.01 + np.abs(other) * 2.0 ** -7
            assert np.all(np.abs(got - other) <= tol), float((np.abs(got - other) / tol).max())
            if want is not None:
                tol = 0.01 + np.abs(want) * 2.0 ** -7
                assert np.all(np.abs(got - want) <= tol), float((np.abs(got - want) / tol).max())
    assert len(picked) >= 2, picked                # (the calls selected different experts: the blocks were re-copied)
    model.unload()


@pytest.mark.parametrize("rows", [2, 3, 4])
@pytest.mark.parametrize("seed", [0, 1])
def test_moe_two_to_four_rows_run_on_the_lean_kernel(be, rows, seed, capfd, monkeypatch):
    """Round 6: the reference's fused MoE form covers <= 4 rows (q_mlp.cu:316-436, moe_mlp.py:238); here 2-4 rows run the experts on the
    chained decode kernel too -- every expert's argument blocks planned per row count at load time, ONE gate|up launch and ONE down
    launch over ALL experts (blockIdx.y = expert; workgroups of an expert no row is routed to leave at entry, q_gemm_kernel.cuh:189-200),
    the down launch multiplies row r by ITS routing weight, then the combine launch.  Against the float64 oracle (rows whose router
    decision is a near tie left out) and against the grouped route; routing weights bit-identical."""
    from exllamav2_amd.config import ExLlamaV2Config
    from exllamav2_amd.synth import synth_checkpoint
    from exllamav2_amd.model import ExLlamaV2
    from oracle.model import OracleModel
    rng = np.random.default_rng(1200 + 10 * rows + seed)
    hidden, inter = int(rng.choice([256, 512])), int(rng.choice([512, 768]))
    cfg = ExLlamaV2Config(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=1, num_attention_heads=4,
                          num_key_value_heads=2, head_dim=64, vocab_size=96, max_seq_len=256, max_input_len=32,
                          max_batch_size=16, num_experts=8, num_experts_per_token=2, arch="mixtral")
    ck = synth_checkpoint(cfg, be.device, recipe=str(rng.choice(["3.5bpw", "4.0bpw", "2.5bpw"])), seed=60 + seed)
    oracle = OracleModel(cfg, ck)
    # (off by default: measured slower than the grouped launches at Mixtral's widths, profiles/r09fg_moe_rows_2_4.txt; planned when the module is made)
    monkeypatch.setenv("EXL2_MOE_LEAN_ROWS", "1")
    model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(ck)
    moe = model.layers[0][1]
    x = torch.from_numpy((rng.standard_normal((rows, 1, hidden)) * 0.7).astype(np.float16)).to(be.device)
    monkeypatch.setenv("EXL2_DEBUG_ROUTE", "1")
    y = x.clone(); moe.forward(y)
    monkeypatch.delenv("EXL2_DEBUG_ROUTE")
    assert f"route: lean rows={rows}" in capfd.readouterr().err
    w_lean = be.n(moe.temp_logits[:rows]).copy()
    monkeypatch.setenv("EXL2_MOE_NO_LEAN", "1")
    z = x.clone(); moe.forward(z)
    monkeypatch.delenv("EXL2_MOE_NO_LEAN")
    assert np.array_equal(w_lean.view(np.uint16), be.n(moe.temp_logits[:rows]).view(np.uint16))
    assert len({tuple(np.nonzero(w_lean[r])[0].tolist()) for r in range(rows)}) >= 1
    xh = be.n(x).reshape(rows, hidden)
    pfx = "model.layers.0"
    oracle.router_margin = np.full((rows,), np.inf)
    want = oracle.moe_mlp(xh, OM.rms_norm(xh, oracle.w[pfx + ".post_attention_layernorm"], cfg.norm_eps), pfx).astype(np.float64)
    ok = oracle.router_margin > 2e-3
    got, other = be.n(y).astype(np.float64).reshape(rows, hidden), be.n(z).astype(np.float64).reshape(rows, hidden)
    tol = 0.01 + np.abs(other) * 2.0 ** -7
    assert np.all(np.abs(got - other) <= tol), float((np.abs(got - other) / tol).max())
    tol = 0.01 + np.abs(want) * 2.0 ** -7
    assert ok.any() and np.all(np.abs(got - want)[ok] <= tol[ok]), float((np.abs(got - want)[ok] / tol[ok]).max())
    model.unload()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("EXL2_TEST_SEEDS", "6")))))        # (more seeds: a longer hunt, by hand)
def test_moe_mlp_forward_random_shapes(be, seed, monkeypatch):
    """Seeded random sparse-MoE blocks: 4 / 8 experts, top-1..3, 1-24 rows, hidden / intermediate sizes off the powers of two, 4 / 3 / 2-bit
    sections, routers from near-uniform to peaked (experts without a row, every row to one expert), the grouped route / the batched
    expert launches / the per-expert loop."""
    rng = np.random.default_rng(31000 + seed)
    E = int(rng.choice([4, 8])); topk = int(rng.integers(1, 4))
    hidden = 64 * int(rng.integers(2, 7)); inter = 64 * int(rng.integers(2, 9))
    rows = int(rng.choice([1, 2, 4, 5, 8, 13, 16, 17, 24]))
    b1, b2 = sorted(rng.choice([5, 4, 3, 2], size=2, replace=False).tolist(), reverse=True)
    cut_u = 32 * int(rng.integers(1, hidden // 32)); cut_d = 32 * int(rng.integers(1, inter // 32))
    spec_up = [(b1, 32, cut_u), (b2, int(rng.choice([32, 64])), hidden - cut_u)]
    spec_dn = [(b1, 32, cut_d), (b2, int(rng.choice([32, 64])), inter - cut_d)]
    route = str(rng.choice(["default", "batched", "loop"]))
    shared = route != "loop"
    if route == "batched" and rows <= 16:
        monkeypatch.setenv("EXL2_MOE_NO_GROUP", "1")
    _moe_case(be, rows, shared, E=E, topk=topk, hidden=hidden, inter=inter, seed=32000 + seed, spec_up=spec_up, spec_dn=spec_dn,
              gate_scale=float(rng.choice([0.02, 0.3, 3.0])))
