"""q_attn_forward_1 / q_attn_forward_2 at the op level, in the form the reference calls them.

Reference call site: attn.py:1089-1141 (scalar `past_len` or per-sequence `past_lens`, default RoPE on, and -- for batch
1 -- K/V written STRAIGHT into the cache rows `batch_keys[:1, past_len:past_len + q_len]`); bindings
ext_qattn.cpp:115-191; device code q_attn.cu:247-345 (RMSNorm -> q/k/v projections -> rope_ on q and k in place).
Checker: the numpy oracle (RMSNorm, matmul(x, reconstruct()), RoPE -- each pinned by the executed reference kernels).
"""
import numpy as np
import pytest
import torch

from oracle import exl2 as OX
from oracle import modules as OM
from tests.util import exl2_to_torch, half_tol
from exllamav2_amd.ext import none_tensor

F16 = np.float16
HID, NH, NKV, HD, MAXSEQ = 256, 4, 2, 64, 64


def _lin(be, k, n, seed):
    t = OX.synth_exl2(k, n, [(5, 32, 32), (4, 32, k - 32)], seed=seed, act_order=True, sigma=0.06)
    ref = OX.exl2_reconstruct(t)
    w = exl2_to_torch(be, t)                  # caller-owned tensors: the handle only borrows them (q_matrix.cuh:11-83)
    return ref, be.ext.make_q_matrix_from_dict(w, None), w


@pytest.fixture
def attn(be):
    refs, hs, owned = {}, {}, []
    for i, (name, k, n) in enumerate((("q", HID, NH * HD), ("k", HID, NKV * HD), ("v", HID, NKV * HD), ("o", NH * HD, HID))):
        refs[name], hs[name], w = _lin(be, k, n, 40 + i)
        owned.append(w)
    rng = np.random.default_rng(7)
    nw = (1 + 0.1 * rng.standard_normal(HID)).astype(F16)
    temp_state = torch.zeros((64, HID), dtype=torch.float16, device=be.device)
    nwt = be.t(nw)
    h = be.ext.make_q_attn(nwt, none_tensor, True, False, 1e-5, hs["q"], hs["k"], hs["v"], hs["o"], temp_state, none_tensor,
                           64, HID, NH, NKV, HD, MAXSEQ, True, 2, HD, none_tensor, none_tensor, none_tensor, none_tensor,
                           False, False)
    sin, cos = OM.rope_tables(MAXSEQ, HD, neox=True)
    yield dict(h=h, refs=refs, nw=nw, sin=sin, cos=cos, keep=(nwt, temp_state, owned))
    be.ext.free_q_attn(h)
    for v in hs.values():
        be.ext.free_q_matrix(v)


def _want(attn, x, positions):
    b, s, _ = x.shape
    n = OM.rms_norm(x.reshape(b * s, HID), attn["nw"], 1e-5)
    out = {}
    for name, heads in (("q", NH), ("k", NKV), ("v", NKV)):
        out[name] = OX.gemm_ref(n, attn["refs"][name], exact=True).reshape(b, s, heads, HD)
    q = OM.rope_(out["q"].astype(F16), attn["sin"], attn["cos"], positions, neox=True)
    k = OM.rope_(out["k"].astype(F16), attn["sin"], attn["cos"], positions, neox=True)
    return q, k, out["v"].astype(F16), out


def _close(got, want16, exact64):
    """|got - float64 result| within the q_gemm tolerance; RoPE of a value 1 ulp off moves the result by < 2 ulps."""
    err = np.abs(got.astype(np.float64) - want16.astype(np.float64))
    return np.all(err <= 3 * half_tol(exact64, HID))


@pytest.mark.parametrize("q_len", [1, 5])
def test_forward_1_scalar_past_len_kv_alias_cache_rows(be, attn, q_len):
    """batch 1, scalar past_len, apply_rope default (True), K/V aliasing the cache rows (attn.py:1089-1092)."""
    rng = np.random.default_rng(q_len)
    past = 9
    x = rng.standard_normal((1, q_len, HID)).astype(F16)
    kc = torch.full((1, MAXSEQ, NKV * HD), 7.0, dtype=torch.float16, device=be.device)
    vc = torch.full((1, MAXSEQ, NKV * HD), -7.0, dtype=torch.float16, device=be.device)
    q = torch.zeros((1, q_len, NH * HD), dtype=torch.float16, device=be.device)
    k_states, v_states = kc[:1, past:past + q_len, :], vc[:1, past:past + q_len, :]
    be.ext.q_attn_forward_1(attn["h"], be.t(x), 1, q_len, past, none_tensor, q, k_states, v_states,
                            be.t(attn["sin"]), be.t(attn["cos"]), [], none_tensor)
    wq, wk, wv, ex = _want(attn, x, np.array([past]))
    assert _close(be.n(q).reshape(1, q_len, NH, HD), wq, ex["q"])
    assert _close(be.n(kc)[:, past:past + q_len].reshape(1, q_len, NKV, HD), wk, ex["k"])
    assert _close(be.n(vc)[:, past:past + q_len].reshape(1, q_len, NKV, HD), wv, ex["v"])
    # nothing outside the aliased rows was touched
    assert np.all(be.n(kc)[:, :past] == 7.0) and np.all(be.n(kc)[:, past + q_len:] == 7.0)
    assert np.all(be.n(vc)[:, :past] == -7.0) and np.all(be.n(vc)[:, past + q_len:] == -7.0)
    # RoPE really happened, with the positions past .. past + q_len - 1 (not 0)
    unrot = ex["k"].astype(F16)
    assert not np.array_equal(be.n(kc)[:, past:past + q_len].reshape(1, q_len, NKV, HD), unrot)


def test_forward_1_per_sequence_past_lens(be, attn):
    """batch 3, position offsets as an int32 tensor (attn.py:1104-1110: past_len + past_lens[b])."""
    rng = np.random.default_rng(11)
    b, q_len, past = 3, 2, 4
    offs = np.array([0, 17, 30], dtype=np.int32)
    x = rng.standard_normal((b, q_len, HID)).astype(F16)
    q = torch.zeros((b, q_len, NH * HD), dtype=torch.float16, device=be.device)
    k = torch.zeros((b, q_len, NKV * HD), dtype=torch.float16, device=be.device)
    v = torch.zeros_like(k)
    be.ext.q_attn_forward_1(attn["h"], be.t(x), b, q_len, past, be.t(offs), q, k, v, be.t(attn["sin"]), be.t(attn["cos"]),
                            [], none_tensor)
    wq, wk, wv, ex = _want(attn, x, offs + past)
    assert _close(be.n(q).reshape(b, q_len, NH, HD), wq, ex["q"])
    assert _close(be.n(k).reshape(b, q_len, NKV, HD), wk, ex["k"])
    assert _close(be.n(v).reshape(b, q_len, NKV, HD), wv, ex["v"])


def test_forward_1_many_rows(be, attn):
    """rows > 16: the unfused branch (norm launch + projections) with the same contract."""
    rng = np.random.default_rng(13)
    b, q_len, past = 2, 20, 3
    x = rng.standard_normal((b, q_len, HID)).astype(F16)
    q = torch.zeros((b, q_len, NH * HD), dtype=torch.float16, device=be.device)
    k = torch.zeros((b, q_len, NKV * HD), dtype=torch.float16, device=be.device)
    v = torch.zeros_like(k)
    be.ext.q_attn_forward_1(attn["h"], be.t(x), b, q_len, past, none_tensor, q, k, v, be.t(attn["sin"]), be.t(attn["cos"]))
    wq, wk, wv, ex = _want(attn, x, np.array([past, past]))
    assert _close(be.n(q).reshape(b, q_len, NH, HD), wq, ex["q"])
    assert _close(be.n(k).reshape(b, q_len, NKV, HD), wk, ex["k"])
    assert _close(be.n(v).reshape(b, q_len, NKV, HD), wv, ex["v"])


@pytest.mark.parametrize("rows", [1, 6, 24])
def test_forward_2_residual(be, attn, rows):
    """q_attn_forward_2 (ext_qattn.cpp:161-191): x += attn_out . Wo, in place."""
    rng = np.random.default_rng(17 + rows)
    x = rng.standard_normal((1, rows, HID)).astype(F16)
    a = rng.standard_normal((1, rows, NH * HD)).astype(F16)
    xt = be.t(x)
    be.ext.q_attn_forward_2(attn["h"], xt, be.t(a), 1, rows, [], none_tensor)
    want = OX.gemm_ref(a.reshape(rows, -1), attn["refs"]["o"], c_in=x.reshape(rows, -1), exact=True)
    assert np.all(np.abs(be.n(xt).reshape(rows, -1).astype(np.float64) - want) <= half_tol(want, NH * HD))


def test_rows_beyond_max_rows_are_refused(be, attn):
    """temp_state is sized for max_rows (64 here): a larger chunk must be an error, not a write past the scratch."""
    rows = 65
    x = torch.zeros((1, rows, HID), dtype=torch.float16, device=be.device)
    q = torch.zeros((1, rows, NH, HD), dtype=torch.float16, device=be.device)
    k = torch.zeros((1, rows, NKV, HD), dtype=torch.float16, device=be.device)
    v = torch.zeros_like(k)
    with pytest.raises(RuntimeError, match="exceed max_rows"):
        be.ext.q_attn_forward_1(attn["h"], x, 1, rows, 0, none_tensor, q, k, v, be.t(attn["sin"]), be.t(attn["cos"]))
    with pytest.raises(RuntimeError, match="exceed max_rows"):
        be.ext.q_attn_forward_2(attn["h"], x, q.view(1, rows, NH * HD), 1, rows)
