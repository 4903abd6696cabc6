"""Host-side mirror of the reference's single-process tensor-parallel bindings (exllamav2_amd/ext_tp.py <-> ext_tp.cpp,
tp_attn_forward_ / tp_mlp_forward_ of ext_qattn.cpp:261-700 / ext_qmlp.cpp:326-473).

The test plays the reference's own host sequence: tensors sliced by output columns exactly as `ExLlamaV2Linear.tp_split`
does (linear.py:567-611: `q_weight[:, a:b]`, `q_scale[:, a/8:b/8]`, everything else shared) and handed to
`make_q_matrix_split`; scratch lists as `tp_split` of attention / MLP allocates them (attn.py:1591-1599, mlp.py:577-584);
hidden states living in `pinned_temp[0]` between modules (tensor_p.py:252-256).  Checker: the numpy oracle.
emu: a two-way split with both "devices" on the host; hip: a one-way split on the GPU with real streams / pinned memory (the
GPU box has one device -- the multi-device ordering is events on streams, exercised by construction only).
"""
import numpy as np
import pytest
import torch

from oracle import exl2 as OX
from oracle import modules as OM
from tests.util import exl2_to_torch, half_tol
from exllamav2_amd import ext_tp as TP

F16 = np.float16
HID, NH, NKV, HD, INTER, MAXSEQ, MAXROWS = 256, 4, 2, 64, 512, 64, 16
G = NH // NKV


def _splits(be):
    if be.is_emu:
        kv = [(0, 0, 1), (1, 1, 2)]
        idc = [(0, 0, 256), (1, 256, 512)]
        rs = [(0, 0, 128), (1, 128, 256)]
    else:
        kv, idc, rs = [(0, 0, NKV)], [(0, 0, INTER)], [(0, 0, HID)]
    q = [(d, a * G, b * G) for d, a, b in kv]
    vc = [(d, a, b) for d, a, b in rs]
    return kv, idc, vc, rs, q


@pytest.fixture
def tp(be):
    TP._Lib.bound = be.ext
    kv, idc, vc, rs, q = _splits(be)
    pinned = [torch.empty((MAXROWS * INTER,), dtype=torch.float16, pin_memory=not be.is_emu) for _ in range(2)]
    if be.is_emu:
        streams = [1, 1]                                   # any non-zero handle: the host-only mode ignores them
        keep = None
    else:
        keep = torch.cuda.Stream()
        streams = [keep.cuda_stream]
    h = TP.make_tp_context(kv, idc, vc, rs, q, pinned, streams)
    yield dict(h=h, kv=kv, id=idc, rs=rs, q=q, pinned=pinned, keep=keep)
    TP.free_tp_context(h)
    TP._Lib.bound = None


def _sliced_linear(be, k, n, split, dim, seed):
    """the full matrix (oracle) + one handle per device over column slices, as linear.py:567-611 builds them"""
    t = OX.synth_exl2(k, n, [(5, 32, 32), (4, 32, k - 32)], seed=seed, act_order=True, sigma=0.06)
    ref = OX.exl2_reconstruct(t)
    handles, keep = [], []
    for _, a, b in split:
        a, b = a * dim, b * dim
        s = dict(t)
        s["q_weight"] = np.ascontiguousarray(t["q_weight"][:, a:b])
        s["q_scale"] = np.ascontiguousarray(t["q_scale"][:, a // 8:b // 8])
        w = exl2_to_torch(be, s)
        handles.append(be.ext.make_q_matrix_from_dict(w, None, split=True))
        keep.append(w)
    return ref, handles, keep


def _dev_tensors(be, split, cols_of):
    return [torch.zeros((MAXROWS, cols_of(a, b)), dtype=torch.float16, device=be.device) for _, a, b in split]


def _sync(be):
    if not be.is_emu:
        torch.cuda.synchronize()


def test_broadcast_gather_all_reduce(be, tp):
    rng = np.random.default_rng(3)
    rows = 5
    x = rng.standard_normal((rows, HID)).astype(F16)
    pin0 = tp["pinned"][0][:rows * HID].view(rows, HID)
    pin0.copy_(torch.from_numpy(x))
    targets = [torch.zeros((rows, HID), dtype=torch.float16, device=be.device) for _ in tp["rs"]]
    TP.tp_broadcast(tp["h"], 0, pin0, TP.BROADCAST_RS, targets, 1, -1)        # host source that IS the staging buffer
    _sync(be)
    for t in targets:
        assert np.array_equal(be.n(t), x)
    # device source -> staging buffer -> the other devices
    src = be.t((x * 2).astype(F16))
    _sync(be)                                   # (be.t uploads on torch's default stream; the TP entry points work on the context's streams)
    TP.tp_broadcast(tp["h"], 1, src, TP.BROADCAST_RS, targets, 1, -1)
    _sync(be)
    for t in targets:
        assert np.array_equal(be.n(t), (x * 2).astype(F16))
    # gather of column slices, then back out to every device of another split
    parts = [be.t(np.ascontiguousarray(x[:, a:b])) for _, a, b in tp["rs"]]
    outs = [torch.zeros((rows, HID), dtype=torch.float16, device=be.device) for _ in tp["rs"]]
    _sync(be)
    TP.tp_gather(tp["h"], 0, parts, TP.BROADCAST_RS, outs, TP.BROADCAST_RS, 1, -1)
    _sync(be)
    for o in outs:
        assert np.array_equal(be.n(o), x)
    # gather to the host only (ctx.gather of the reference, tensor_p.py:298-322: the result is read from pinned_temp[buffer]);
    # more than one row: the slices' rows are not contiguous in the gathered matrix
    pin1 = tp["pinned"][1][:rows * HID].view(rows, HID)
    pin1.zero_()
    parts3 = [be.t(np.ascontiguousarray((x * 3).astype(F16)[:, a:b])) for _, a, b in tp["rs"]]
    _sync(be)
    TP.tp_gather(tp["h"], 1, parts3, TP.BROADCAST_RS, [], -1, 1, -1)
    _sync(be)
    assert np.array_equal(pin1.numpy(), (x * 3).astype(F16))
    # a host source that is NOT the staging buffer (copied into it first, behind the uploads still reading it)
    y = rng.standard_normal((rows, HID)).astype(F16)
    TP.tp_broadcast(tp["h"], 0, torch.from_numpy(y), TP.BROADCAST_RS, targets, 1, -1)
    _sync(be)
    for t in targets:
        assert np.array_equal(be.n(t), y)
    # all-reduce through the host buffer: every residual ends up as sum_i (residual_0 + tensors_i) chained
    ts = [be.t(rng.standard_normal((rows, HID)).astype(F16)) for _ in tp["rs"]]
    rs = [be.t(np.zeros((rows, HID), dtype=F16)) for _ in tp["rs"]]
    want = np.zeros((rows, HID), dtype=F16)
    for t in ts:
        want = (want.astype(np.float32) + be.n(t).astype(np.float32)).astype(F16)
    _sync(be)
    TP.tp_all_reduce(tp["h"], 0, ts, rs)
    _sync(be)
    for r in rs:
        assert np.array_equal(be.n(r), want)


@pytest.mark.parametrize("rows", [1, 7])
def test_tp_mlp_forward(be, tp, rows):
    rng = np.random.default_rng(rows)
    g_ref, g_h, k1 = _sliced_linear(be, HID, INTER, tp["id"], 1, 51)
    u_ref, u_h, k2 = _sliced_linear(be, HID, INTER, tp["id"], 1, 52)
    d_ref, d_h, k3 = _sliced_linear(be, INTER, HID, tp["rs"], 1, 53)
    nw = (1 + 0.1 * rng.standard_normal(HID)).astype(F16)
    x = rng.standard_normal((rows, HID)).astype(F16)
    norm = [be.t(nw) for _ in tp["id"]]
    bc0 = _dev_tensors(be, tp["id"], lambda a, b: HID)
    bc1 = _dev_tensors(be, tp["id"], lambda a, b: HID)
    bc2 = _dev_tensors(be, tp["id"], lambda a, b: INTER)
    tg = _dev_tensors(be, tp["id"], lambda a, b: b - a)
    tu = _dev_tensors(be, tp["id"], lambda a, b: b - a)
    td = _dev_tensors(be, tp["rs"], lambda a, b: b - a)
    hs = tp["pinned"][0][:rows * HID].view(rows, HID)                       # the reference's ctx.get_pinned(0, ...)
    hs.copy_(torch.from_numpy(x))
    TP.tp_mlp_forward_(tp["h"], hs, bc0, bc1, bc2, tg, tu, td, norm, 1e-5, g_h, u_h, d_h, False)
    _sync(be)
    got = hs.numpy().astype(np.float64)
    n = OM.rms_norm(x, nw, 1e-5)
    a = OM.silu_mul(OX.gemm_ref(n, g_ref, exact=True).astype(F16), OX.gemm_ref(n, u_ref, exact=True).astype(F16))
    d = OX.gemm_ref(a, d_ref, exact=True)
    want = x.astype(np.float64) + d
    # one fp16 rounding of the down projection, one of the residual add (ext_qmlp.cpp:434-451 adds in fp16)
    tol = 3 * half_tol(d, INTER) + np.abs(want) * 2.0 ** -10
    assert np.all(np.abs(got - want) <= tol), float(np.abs(got - want).max())
    for h in g_h + u_h + d_h:
        be.ext.free_q_matrix(h)


def test_tp_attn_forward_prefill_then_decode(be, tp):
    """contiguous per-device caches [batch, max_seq, kv_heads_i, head_dim] (ExLlamaV2Cache_TP, cache.py:680-692); a 5-token
    prefill, then one decode step on top of it -- K/V appended at past_len_tp, RoPE at past + j, causal attention"""
    rng = np.random.default_rng(9)
    q_ref, q_h, k1 = _sliced_linear(be, HID, NH * HD, tp["q"], HD, 61)
    k_ref, k_h, k2 = _sliced_linear(be, HID, NKV * HD, tp["kv"], HD, 62)
    v_ref, v_h, k3 = _sliced_linear(be, HID, NKV * HD, tp["kv"], HD, 63)
    o_ref, o_h, k4 = _sliced_linear(be, NH * HD, HID, tp["q"], HD, 64)
    nw = (1 + 0.1 * rng.standard_normal(HID)).astype(F16)
    norm = [be.t(nw) for _ in tp["q"]]
    sin, cos = OM.rope_tables(MAXSEQ, HD, neox=True)
    ndev = max(d for d, _, _ in tp["q"]) + 1
    sin_l = [be.t(sin) for _ in range(ndev)]
    cos_l = [be.t(cos) for _ in range(ndev)]
    width = tp["q"][-1][2] * HD
    bc0 = _dev_tensors(be, tp["q"], lambda a, b: width)
    bc1 = _dev_tensors(be, tp["q"], lambda a, b: width)
    bc2 = _dev_tensors(be, tp["q"], lambda a, b: width)
    tq = _dev_tensors(be, tp["q"], lambda a, b: (b - a) * HD)
    tk = _dev_tensors(be, tp["kv"], lambda a, b: (b - a) * HD)
    tv = _dev_tensors(be, tp["kv"], lambda a, b: (b - a) * HD)
    to = _dev_tensors(be, tp["q"], lambda a, b: (b - a) * HD)
    bsz = 1
    kc = [torch.zeros((bsz, MAXSEQ, b - a, HD), dtype=torch.float16, device=be.device) for _, a, b in tp["kv"]]
    vc = [torch.zeros((bsz, MAXSEQ, b - a, HD), dtype=torch.float16, device=be.device) for _, a, b in tp["kv"]]
    assert HID == width
    k_all = np.zeros((bsz, 0, NKV, HD), dtype=F16)
    v_all = np.zeros((bsz, 0, NKV, HD), dtype=F16)
    past = 0
    for q_len in (5, 1):
        x = rng.standard_normal((bsz, q_len, HID)).astype(F16)
        rows = bsz * q_len
        hs = tp["pinned"][0][:rows * HID].view(rows, HID)
        hs.copy_(torch.from_numpy(x.reshape(rows, HID)))
        past_tp = [be.t(np.full((bsz,), past, dtype=np.int32)) for _ in tp["kv"]]
        TP.tp_attn_forward_(tp["h"], hs, bc0, bc1, bc2, tq, tk, tv, to, kc, vc, norm, 1e-5, q_h, k_h, v_h, o_h, HD,
                            TP.ROPE_STYLE_NEOX, bsz, q_len, sin_l, cos_l, past_tp, HD ** -0.5)
        _sync(be)
        got = hs.numpy().astype(np.float64).reshape(bsz, q_len, HID)
        n = OM.rms_norm(x.reshape(rows, HID), nw, 1e-5)
        pos = np.full((bsz,), past)
        qq = OM.rope_(OX.gemm_ref(n, q_ref, exact=True).astype(F16).reshape(bsz, q_len, NH, HD), sin, cos, pos, neox=True)
        kk = OM.rope_(OX.gemm_ref(n, k_ref, exact=True).astype(F16).reshape(bsz, q_len, NKV, HD), sin, cos, pos, neox=True)
        vv = OX.gemm_ref(n, v_ref, exact=True).astype(F16).reshape(bsz, q_len, NKV, HD)
        k_all = np.concatenate([k_all, kk], axis=1)
        v_all = np.concatenate([v_all, vv], axis=1)
        ao = OM.attention(qq, k_all, v_all).reshape(rows, NH * HD)
        o = OX.gemm_ref(ao, o_ref, exact=True)
        want = x.reshape(rows, HID).astype(np.float64) + o
        # q/k/v roundings move the attention output by a few fp16 ulps before o_proj; then one rounding of o, one of the add
        tol = 6 * half_tol(o, NH * HD) + np.abs(want) * 2.0 ** -10 + 4e-3
        err = np.abs(got.reshape(rows, HID) - want)
        assert np.all(err <= tol), float(err.max())
        # the caches hold the rotated keys / raw values of this step at positions past .. past + q_len - 1
        for i, (_, a, b) in enumerate(tp["kv"]):
            gk = be.n(kc[i])[:, past:past + q_len].astype(np.float64)
            assert np.all(np.abs(gk - kk[:, :, a:b].astype(np.float64)) <= 3 * half_tol(kk[:, :, a:b].astype(np.float64), HID))
        past += q_len
    for h in q_h + k_h + v_h + o_h:
        be.ext.free_q_matrix(h)


def test_copy_2d_async_strided(be):
    """csrc/peer.hip exl2_memcpy_2d_async: rows of a narrow matrix into a column range of a wider one and back (device <->
    device, device <-> pinned host)"""
    rng = np.random.default_rng(5)
    a = rng.standard_normal((7, 24)).astype(F16)
    wide = torch.zeros((7, 64), dtype=torch.float16, device=be.device)
    stream = None if be.is_emu else torch.cuda.current_stream().cuda_stream
    be.ext.copy_2d_async(wide[:, 16:40], be.t(a), stream)
    host = torch.zeros((7, 64), dtype=torch.float16, pin_memory=not be.is_emu)
    be.ext.copy_2d_async(host[:, 8:32], wide[:, 16:40], stream)
    back = torch.zeros((7, 24), dtype=torch.float16, device=be.device)
    _sync(be)
    be.ext.copy_2d_async(back, host[:, 8:32], stream)
    _sync(be)
    w = be.n(wide)
    assert np.array_equal(w[:, 16:40], a) and not w[:, :16].any() and not w[:, 40:].any()
    assert np.array_equal(host.numpy()[:, 8:32], a) and np.array_equal(be.n(back), a)
    with pytest.raises(RuntimeError):
        be.ext.copy_2d_async(back, host[:, 8:30], stream)


@pytest.mark.gpu
def test_two_device_broadcast_gather_over_peer_copies():
    """The two-device form of tp_broadcast / tp_gather on real hardware: slices on cuda:0 and cuda:1, device targets written
    by peer copies on the targets' streams, host target by strided copies into the pinned buffer.  Needs two visible GPUs
    (the build's GPU box has one; the emulator covers the two-way split's indexing)."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from exllamav2_amd.ext import ext_c
    TP._Lib.bound = None
    devs = [torch.device("cuda", 0), torch.device("cuda", 1)]
    rs = [(0, 0, 128), (1, 128, 256)]
    kv, q = [(0, 0, 1), (1, 1, 2)], [(0, 0, 2), (1, 2, 4)]
    idc = [(0, 0, 256), (1, 256, 512)]
    pinned = [torch.empty((MAXROWS * INTER,), dtype=torch.float16, pin_memory=True) for _ in range(2)]
    streams = [torch.cuda.Stream(device=d) for d in devs]
    h = TP.make_tp_context(kv, idc, rs, rs, q, pinned, [s.cuda_stream for s in streams])
    try:
        rng = np.random.default_rng(9)
        rows = 5
        x = rng.standard_normal((rows, HID)).astype(F16)
        src = torch.from_numpy(x).to(devs[1])
        for s in streams:
            s.synchronize()
        targets = [torch.zeros((rows, HID), dtype=torch.float16, device=d) for d in devs]
        TP.tp_broadcast(h, 0, src, TP.BROADCAST_RS, targets, 1, -1)                  # device source on cuda:1
        torch.cuda.synchronize(0); torch.cuda.synchronize(1)
        for t in targets:
            assert np.array_equal(t.cpu().numpy(), x)
        assert np.array_equal(pinned[0][:rows * HID].view(rows, HID).numpy(), x)
        parts = [torch.from_numpy(np.ascontiguousarray(x[:, a:b])).to(devs[d]) for d, a, b in rs]
        outs = [torch.zeros((rows, HID), dtype=torch.float16, device=d) for d in devs]
        TP.tp_gather(h, 1, parts, TP.BROADCAST_RS, outs, TP.BROADCAST_RS, 1, -1)     # all-gather over peer copies
        torch.cuda.synchronize(0); torch.cuda.synchronize(1)
        for o in outs:
            assert np.array_equal(o.cpu().numpy(), x)
        pinned[1].zero_()
        TP.tp_gather(h, 1, parts, TP.BROADCAST_RS, [], -1, 1, -1)                    # gather to the host
        torch.cuda.synchronize(0); torch.cuda.synchronize(1)
        assert np.array_equal(pinned[1][:rows * HID].view(rows, HID).numpy(), x)
    finally:
        TP.free_tp_context(h)
