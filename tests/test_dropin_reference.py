"""The UNMODIFIED reference host code running on top of the drop-in.

BASELINE.json north_star: "ExLlamaV2 / ExLlamaV2DynamicGenerator load unmodified EXL2/GPTQ checkpoints as a drop-in".
tools/run_reference_dropin.py imports the reference's own `exllamav2` package (config.py, stloader.py, model.py, linear.py,
attn.py, mlp.py, cache.py, ... untouched) with dropin/ first on sys.path, so that `exllamav2/ext.py:105-109` binds
`ext_c` to dropin/exllamav2_ext.py -> libexl2_hip.so; it loads a synthetic EXL2 MODEL DIRECTORY (config.json +
model.safetensors in the on-disk format, down_proj with its own act-order permutation so that the loader's
tensor_remap / tensor_remap_4bit folding of linear.py:156-158 runs) and plays test_inference.py:604-609's greedy loop.
Logits are compared with the numpy oracle.

Where the package comes from: /root/reference where that exists; on the GPU box the build-time mirror of its *.py files
under the git-ignored oracle/_ref/reference_py (oracle/ref_build/build.sh).  Neither present -> skipped.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.conftest import ROOT


def _reference_pkg():
    for d in ("/root/reference", os.path.join(ROOT, "oracle", "_ref", "reference_py")):
        if os.path.isfile(os.path.join(d, "exllamav2", "model.py")):
            return d
    return None


# ---- the load-path functions of the drop-in against the reference's loops (ext_stloader.cpp:160-219), CPU ----------------

def _dropin():
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    try:
        import exllamav2_ext
    finally:
        sys.path.pop(0)
    return exllamav2_ext


def test_tensor_remap_matches_reference_loops():
    try:
        E = _dropin()
    except Exception as e:                                      # the drop-in binds libexl2_hip.so at import
        pytest.skip(f"drop-in not importable here: {e}")
    rng = np.random.default_rng(0)
    rows, cols = 5, 64
    t = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(rows, cols), dtype=np.int64).astype(np.int32)
    idx = rng.permutation(cols).astype(np.int32)
    want = t[:, idx]                                            # ext_stloader.cpp:176-182: *a++ = temp[idx[c]]
    tt = torch.from_numpy(t.copy())
    E.tensor_remap(tt, torch.from_numpy(idx))
    assert np.array_equal(tt.numpy(), want)
    # 4-bit: nibble c of a row <- nibble idx[c] (ext_stloader.cpp:203-216)
    q = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(rows, cols // 8), dtype=np.int64).astype(np.int32)
    u = q.view(np.uint32)
    nib = np.stack([(u >> (4 * b)) & 0xF for b in range(8)], axis=-1).reshape(rows, cols)
    want_n = nib[:, idx].reshape(rows, cols // 8, 8)
    want_q = np.zeros((rows, cols // 8), dtype=np.uint32)
    for b in range(8):
        want_q |= want_n[:, :, b].astype(np.uint32) << np.uint32(4 * b)
    qt = torch.from_numpy(q.copy())
    E.tensor_remap_4bit(qt, torch.from_numpy(idx))
    assert np.array_equal(qt.numpy().view(np.uint32), want_q)


def test_stloader_read_cpu(tmp_path):
    try:
        E = _dropin()
    except Exception as e:
        pytest.skip(f"drop-in not importable here: {e}")
    data = np.arange(1000, dtype=np.uint8)
    f = tmp_path / "blob.bin"
    data.tofile(f)
    t = torch.zeros((3, 5), dtype=torch.float16)
    E.stloader_read(str(f), 100, 30, t)
    assert np.array_equal(t.numpy().view(np.uint8).reshape(-1), data[100:130])
    with pytest.raises(RuntimeError):
        E.stloader_read(str(f), 990, 30, t)                     # short read
    with pytest.raises(RuntimeError):
        E.stloader_read(str(f), 0, 28, t)                       # size mismatch


def test_partial_strings_match():
    try:
        E = _dropin()
    except Exception as e:
        pytest.skip(f"drop-in not importable here: {e}")
    def call(text, strings):
        enc = [s.encode("utf-32-le") for s in strings]
        offs = np.cumsum([0] + [len(e) for e in enc]).astype(np.uint32)
        return E.partial_strings_match(text.encode("utf-32-le"), offs.tobytes(), b"".join(enc))
    assert call("hello world", ["wor"]) == 6
    assert call("hello wo", ["world"]) == -2                    # could still complete
    assert call("hello", ["xyz"]) == -1


# ---- the whole thing, on the GPU ------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_unmodified_reference_runs_on_the_dropin(tmp_path):
    ref = _reference_pkg()
    if ref is None:
        pytest.skip("no copy of the reference's host package on this machine")
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from exllamav2_amd.config import ExLlamaV2Config
    from exllamav2_amd.synth import synth_checkpoint
    from exllamav2_amd.synth_dir import write_model_dir
    from oracle.model import OracleModel
    cfg = ExLlamaV2Config(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                          num_key_value_heads=2, head_dim=64, vocab_size=320, max_seq_len=256, max_input_len=32)
    ck = synth_checkpoint(cfg, "cpu", seed=0, down_act_order=True)
    oracle = OracleModel(cfg, ck)
    model_dir = write_model_dir(str(tmp_path / "model"), cfg, ck)
    try:
        from exllamav2_amd.synth_dir import write_tokenizer
        write_tokenizer(model_dir, cfg.vocab_size)                 # enables the dynamic-generator leg of the runner
    except ImportError:
        pass
    out = str(tmp_path / "out.npz")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "dropin"), ROOT, ref]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_reference_dropin.py"), model_dir, out],
                       env=env, capture_output=True, text=True, cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    got = np.load(out)
    ids = np.array([[3, 17, 42, 7]])
    oracle.reset(1)
    want = oracle.forward(ids)
    tol = lambda w: 0.03 + np.abs(w) * 2.0 ** -8
    assert np.all(np.abs(got["prefill"][..., :cfg.vocab_size].astype(np.float64) - want) <= tol(want))
    for i, tok in enumerate(got["tokens"]):
        w = oracle.forward(np.array([[int(tok)]]))
        g = got["steps"][:, i:i + 1, :cfg.vocab_size].astype(np.float64)
        assert np.all(np.abs(g - w) <= tol(w)), i
    # the dynamic generator (paged attention through dropin/flash_attn): per job, logits of every generated position
    if "dyn_tokens_0" in got:
        n_conf = 0
        for j in range(2):
            prompt, toks, lg = got[f"dyn_prompt_{j}"], got[f"dyn_tokens_{j}"], got[f"dyn_logits_{j}"]
            assert len(toks) == 4 and lg.shape[1] == len(toks)
            oracle.reset(1)
            w = oracle.forward(prompt[None, :])[:, -1:]
            for i, tok in enumerate(toks):
                g = lg[:, i:i + 1, :cfg.vocab_size].astype(np.float64)
                assert np.all(np.abs(g - w) <= tol(w)), (j, i)
                top2 = np.sort(w[0, 0])[-2:]
                if top2[1] - top2[0] > 0.12:
                    assert int(tok) == int(np.argmax(w[0, 0])), (j, i)
                    n_conf += 1
                w = oracle.forward(np.array([[int(tok)]]))
        assert n_conf >= 1, "vacuous token check"



@pytest.mark.gpu
def test_unmodified_reference_runs_a_sparse_moe_model_on_the_dropin(tmp_path):
    """SURVEY.md 8a rows a13 x a17: the UNMODIFIED reference's Mixtral path -- architecture.py:291-305, moe_mlp.py:238-253
    (`ext_c.q_moe_mlp_forward_` for <= 4 rows: the 4-token prompt and every decode step) -- on the drop-in: a synthetic 8-expert top-2
    model directory, prompt logits and greedy decode steps against the float64 oracle (steps at which the oracle's router decision is
    a near tie are left out: top-k is discontinuous).  At one row the block runs the round-6 route behind the operator boundary (the two
    selected experts on the lean kernel, x updated in place through whatever pointer the host passes)."""
    ref = _reference_pkg()
    if ref is None:
        pytest.skip("no copy of the reference's host package on this machine")
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from exllamav2_amd.config import ExLlamaV2Config
    from exllamav2_amd.synth import synth_checkpoint
    from exllamav2_amd.synth_dir import write_model_dir
    from oracle.model import OracleModel
    cfg = ExLlamaV2Config(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                          num_key_value_heads=2, head_dim=64, vocab_size=320, max_seq_len=256, max_input_len=32,
                          num_experts=8, num_experts_per_token=2, arch="mixtral")
    ck = synth_checkpoint(cfg, "cpu", recipe="3.5bpw", seed=4)       # (a seed whose router margins stay >= 0.008 over the prompt and four steps)
    oracle = OracleModel(cfg, ck)
    model_dir = write_model_dir(str(tmp_path / "model"), cfg, ck)
    out = str(tmp_path / "out.npz")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "dropin"), ROOT, ref]), EXL2_DEBUG_ROUTE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_reference_dropin.py"), model_dir, out],
                       env=env, capture_output=True, text=True, cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "q_moe_mlp route: lean (down pair sums) rows=1" in r.stderr, r.stderr[-1500:]     # (the decode steps took the one-row route)
    got = np.load(out)
    ids = np.array([[3, 17, 42, 7]])
    tol = lambda w: 0.03 + np.abs(w) * 2.0 ** -8
    oracle.reset(1)
    want = oracle.forward(ids)
    assert oracle.router_margin[0] > 2e-3                           # (min over the prompt's rows and layers: OracleModel.forward)
    g = got["prefill"][..., :cfg.vocab_size].astype(np.float64)
    assert np.all(np.abs(g - want) <= tol(want))
    compared = 0
    for i, tok in enumerate(got["tokens"]):
        w = oracle.forward(np.array([[int(tok)]]))
        if oracle.router_margin[0] <= 2e-3:
            break                                                   # (from a near tie on the device and the oracle may hold different caches)
        g = got["steps"][:, i:i + 1, :cfg.vocab_size].astype(np.float64)
        assert np.all(np.abs(g - w) <= tol(w)), i
        compared += 1
    assert compared >= 1


@pytest.mark.gpu
def test_unmodified_reference_tensor_parallel_on_the_dropin(tmp_path):
    """`model.load_tp()` + `ExLlamaV2Cache_TP` + the greedy loop of the UNMODIFIED reference on the drop-in's mirror of its
    single-process TP bindings (exllamav2_amd/ext_tp.py): split over every visible device (one on the build's GPU box: the
    slicing, the make_q_matrix_split handles, the pinned-buffer staging and the fused tp_* forwards all run; the two-way
    split of the same code is covered on the emulator, tests/test_ext_tp.py)."""
    ref = _reference_pkg()
    if ref is None:
        pytest.skip("no copy of the reference's host package on this machine")
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from exllamav2_amd.config import ExLlamaV2Config
    from exllamav2_amd.synth import synth_checkpoint
    from exllamav2_amd.synth_dir import write_model_dir
    from oracle.model import OracleModel
    cfg = ExLlamaV2Config(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                          num_key_value_heads=2, head_dim=64, vocab_size=320, max_seq_len=256, max_input_len=32)
    ck = synth_checkpoint(cfg, "cpu", seed=0, down_act_order=True)
    oracle = OracleModel(cfg, ck)
    model_dir = write_model_dir(str(tmp_path / "model"), cfg, ck)
    out = str(tmp_path / "out_tp.npz")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "dropin"), ROOT, ref]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_reference_dropin.py"), model_dir, out, "tp"],
                       env=env, capture_output=True, text=True, cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    got = np.load(out)
    ids = np.array([[3, 17, 42, 7]])
    oracle.reset(1)
    want = oracle.forward(ids)
    tol = lambda w: 0.03 + np.abs(w) * 2.0 ** -8
    assert np.all(np.abs(got["prefill"][..., :cfg.vocab_size].astype(np.float64) - want) <= tol(want))
    for i, tok in enumerate(got["tokens"]):
        w = oracle.forward(np.array([[int(tok)]]))
        g = got["steps"][:, i:i + 1, :cfg.vocab_size].astype(np.float64)
        assert np.all(np.abs(g - w) <= tol(w)), i


@pytest.mark.gpu
def test_unmodified_reference_layer_split_on_the_dropin(tmp_path):
    """The reference's own multi-GPU mode for every architecture: ONE process, `model.load(gpu_split=[...])`, modules dealt to
    devices by byte budget (model.py:176-263), the hidden state hopping device to device (model.py:1012-1016) -- on the
    drop-in, with two devices.  Gated on two visible GPUs (the build's box has one; the driver's 8-GPU node runs it)."""
    ref = _reference_pkg()
    if ref is None:
        pytest.skip("no copy of the reference's host package on this machine")
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    from exllamav2_amd.config import ExLlamaV2Config
    from exllamav2_amd.synth import synth_checkpoint
    from exllamav2_amd.synth_dir import write_model_dir
    from oracle.model import OracleModel
    cfg = ExLlamaV2Config(hidden_size=256, intermediate_size=512, num_hidden_layers=4, num_attention_heads=4,
                          num_key_value_heads=2, head_dim=64, vocab_size=320, max_seq_len=256, max_input_len=32)
    ck = synth_checkpoint(cfg, "cpu", seed=0, down_act_order=True)
    oracle = OracleModel(cfg, ck)
    model_dir = write_model_dir(str(tmp_path / "model"), cfg, ck)
    out = str(tmp_path / "out_split.npz")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "dropin"), ROOT, ref]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_reference_dropin.py"), model_dir, out, "split"],
                       env=env, capture_output=True, text=True, cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    got = np.load(out)
    assert set(int(d) for d in got["module_devices"] if d >= 0) == {0, 1}
    ids = np.array([[3, 17, 42, 7]])
    oracle.reset(1)
    want = oracle.forward(ids)
    tol = lambda w: 0.03 + np.abs(w) * 2.0 ** -8
    assert np.all(np.abs(got["prefill"][..., :cfg.vocab_size].astype(np.float64) - want) <= tol(want))
    for i, tok in enumerate(got["tokens"]):
        w = oracle.forward(np.array([[int(tok)]]))
        g = got["steps"][:, i:i + 1, :cfg.vocab_size].astype(np.float64)
        assert np.all(np.abs(g - w) <= tol(w)), i


@pytest.mark.gpu
def test_unmodified_reference_flash_attn_func_on_the_dropin(tmp_path):
    """The reference's non-paged decode as it runs wherever flash-attn is importable (attn.py:1141-1142 -> _attn_flash ->
    flash_attn_func, attn.py:960-977): dropin/flash_attn serves the call with one launch of csrc/attn.hip on the live rows of the
    reference's own ExLlamaV2Cache (K/V written there by q_attn_forward_1, the reference's `direct` mode).  Prompt of 4 rows,
    then the greedy loop; logits against the oracle."""
    ref = _reference_pkg()
    if ref is None:
        pytest.skip("no copy of the reference's host package on this machine")
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from exllamav2_amd.config import ExLlamaV2Config
    from exllamav2_amd.synth import synth_checkpoint
    from exllamav2_amd.synth_dir import write_model_dir
    from oracle.model import OracleModel
    cfg = ExLlamaV2Config(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                          num_key_value_heads=2, head_dim=64, vocab_size=320, max_seq_len=256, max_input_len=32)
    ck = synth_checkpoint(cfg, "cpu", seed=0, down_act_order=True)
    oracle = OracleModel(cfg, ck)
    model_dir = write_model_dir(str(tmp_path / "model"), cfg, ck)
    out = str(tmp_path / "out_flash.npz")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "dropin"), ROOT, ref]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_reference_dropin.py"), model_dir, out, "flash"],
                       env=env, capture_output=True, text=True, cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    got = np.load(out)
    ids = np.array([[3, 17, 42, 7]])
    oracle.reset(1)
    want = oracle.forward(ids)
    tol = lambda w: 0.03 + np.abs(w) * 2.0 ** -8
    assert np.all(np.abs(got["prefill"][..., :cfg.vocab_size].astype(np.float64) - want) <= tol(want))
    for i, tok in enumerate(got["tokens"]):
        w = oracle.forward(np.array([[int(tok)]]))
        g = got["steps"][:, i:i + 1, :cfg.vocab_size].astype(np.float64)
        assert np.all(np.abs(g - w) <= tol(w)), i
