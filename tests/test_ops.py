"""RMSNorm, RoPE, act*mul, Q-cache codec, attention, RoPE+append, decode utilities -- against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import modules as OM

F16 = np.float16


def _ulp_close(got, want, ulps=1):
    """fp16 outputs equal up to `ulps` units in the last place (fp32 reduction order may move the pre-rounding value)."""
    g = got.astype(np.float32)
    w = want.astype(np.float32)
    tol = np.maximum(np.abs(w) * 2.0 ** -10, 2.0 ** -24) * ulps
    return np.all(np.abs(g - w) <= tol)


@pytest.mark.parametrize("rows,dim", [(1, 256), (3, 4096), (5, 5120)])
def test_rms_norm(be, rows, dim):
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((rows, dim)) * 3).astype(F16)
    w = (1 + 0.1 * rng.standard_normal(dim)).astype(F16)
    y = torch.zeros((rows, dim), dtype=torch.float16, device=be.device)
    be.ext.rms_norm(be.t(x), be.t(w), y, 1e-5)
    assert _ulp_close(be.n(y), OM.rms_norm(x, w, 1e-5))
    xt = be.t(x)
    be.ext.rms_norm_(xt, be.t(w), 1e-5)
    assert np.array_equal(be.n(xt), be.n(y))


@pytest.mark.parametrize("neox", [True, False])
@pytest.mark.parametrize("hd,heads", [(128, 4), (64, 2)])
def test_rope(be, neox, hd, heads):
    rng = np.random.default_rng(1)
    b, s = 2, 3
    x = rng.standard_normal((b, s, heads, hd)).astype(F16)
    sin, cos = OM.rope_tables(64, hd, neox=neox)
    offsets = np.array([5, 17], dtype=np.int32)
    want = OM.rope_(x, sin, cos, offsets + 4, neox=neox)
    xt = be.t(x)
    be.ext.rope_(xt, be.t(sin), be.t(cos), 4, heads, hd, be.t(offsets), neox)
    assert np.array_equal(be.n(xt).view(np.uint16), want.view(np.uint16))
    # scalar past_len only, meta sentinel for offsets (attn.py:1113-1115)
    from exllamav2_amd.ext import none_tensor
    xt = be.t(x)
    be.ext.rope_(xt, be.t(sin), be.t(cos), 9, heads, hd, none_tensor, neox)
    want = OM.rope_(x, sin, cos, np.array([9, 9]), neox=neox)
    assert np.array_equal(be.n(xt).view(np.uint16), want.view(np.uint16))


def test_act_mul(be):
    rng = np.random.default_rng(2)
    g = (rng.standard_normal((3, 512)) * 2).astype(F16)
    u = rng.standard_normal((3, 512)).astype(F16)
    gt = be.t(g)
    be.ext.act_mul_(gt, be.t(u))
    assert _ulp_close(be.n(gt), OM.silu_mul(g, u), ulps=2)


# ---- Q cache -----------------------------------------------------------------------------------------------------------

def _q4_ref_tensor(x):
    """x fp16 [b, T, KVH, hd] flattened per batch row -> packed uint8 [b, T, KVH, hd/2], scales [b, T, KVH, hd/32]."""
    b = x.shape[0]
    pk, sc = [], []
    for i in range(b):
        p, s = OM.q4_pack(x[i].reshape(-1))
        pk.append(p); sc.append(s)
    return np.stack(pk).reshape(*x.shape[:3], x.shape[3] // 2), np.stack(sc).reshape(*x.shape[:3], x.shape[3] // 32)


def test_q4_cache_contiguous_roundtrip(be):
    rng = np.random.default_rng(3)
    b, T, kvh, hd = 2, 16, 2, 128
    k = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    v = (rng.standard_normal((b, T, kvh, hd)) * 0.5).astype(F16)
    kq = torch.zeros((b, T, kvh, hd // 2), dtype=torch.uint8, device=be.device)
    vq = torch.zeros_like(kq)
    ks = torch.zeros((b, T, kvh, hd // 32), dtype=torch.float16, device=be.device)
    vs = torch.zeros_like(ks)
    from exllamav2_amd.ext import none_tensor
    # pack tokens [4, 12) of every row (dim = 256 -> 2 tokens per 512-block, aligned)
    be.ext.fp16_to_q_kv(be.t(k), kq, ks, be.t(v), vq, vs, b, 4, 8, 0, none_tensor, none_tensor, 4)
    kp, ksr = _q4_ref_tensor(k)
    vp, vsr = _q4_ref_tensor(v)
    got_k = be.n(kq); got_ks = be.n(ks)
    assert np.array_equal(got_k[:, 4:12], kp[:, 4:12])          # byte work: bit-exact (correctly rounded fp16 quotient)
    assert np.array_equal(got_ks[:, 4:12].view(np.uint16), ksr[:, 4:12].view(np.uint16))
    assert np.all(got_k[:, :4] == 0) and np.all(got_k[:, 12:] == 0)
    assert np.array_equal(be.n(vq)[:, 4:12], vp[:, 4:12])
    # unpack what the device packed and compare with the oracle's unpack of the same bytes (bit exact)
    ko = torch.zeros((b, T, kvh, hd), dtype=torch.float16, device=be.device)
    vo = torch.zeros_like(ko)
    be.ext.q_to_fp16_kv(kq, ko, ks, vq, vo, vs, b, 4, 8, 0, none_tensor, none_tensor, 4)
    for i in range(b):
        want = OM.q4_unpack(got_k[i, 4:12].reshape(-1), got_ks[i, 4:12].reshape(-1)).reshape(8, kvh, hd)
        assert np.array_equal(be.n(ko)[i, 4:12].view(np.uint16), want.view(np.uint16))
    # 4-bit round trip error: |x - dq(q(x))| bounded by the quantization step (size-independent property)
    err = np.abs(be.n(ko)[:, 4:12].astype(np.float32) - k[:, 4:12].astype(np.float32))
    assert err.max() < 0.6 and err.mean() < 0.12


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("EXL2_TEST_SEEDS", "6")))))        # (more seeds: a longer hunt, by hand)
def test_kv_codec_random_ranges(be, seed):
    """Seeded random contiguous pack / unpack calls of the Q4 / Q8 cache codec: kv widths 128-1024 (so that a 512-element block holds
    4 tokens, 2, 1 or half of one), random [offset, offset + width) token ranges (moved / widened to whole blocks exactly like the
    reference, ext_cache.cpp:148-155), 1-3 sequences, values from tiny to outliers.  Codes, scales and the unpacked fp16 are compared BIT for
    bit with the oracle (cache_q.cuh numerics); what lies outside the widened range must stay untouched."""
    from exllamav2_amd.ext import none_tensor
    rng = np.random.default_rng(37000 + seed)
    kvh = int(rng.choice([1, 2, 4, 8])); hd = 128
    dim = kvh * hd
    b = int(rng.integers(1, 4)); T = 8 * int(rng.integers(2, 9))
    wbits = int(rng.choice([4, 8]))
    scale = float(rng.choice([1e-3, 1.0, 30.0]))
    k = (rng.standard_normal((b, T, kvh, hd)) * scale).astype(F16)
    v = (rng.standard_normal((b, T, kvh, hd)) * scale).astype(F16)
    k[rng.integers(0, b), rng.integers(0, T), rng.integers(0, kvh), rng.integers(0, hd)] = F16(2000.0)      # an outlier
    offset = int(rng.integers(0, T - 1)); width = int(rng.integers(1, T - offset + 1))
    per = hd // 2 if wbits == 4 else hd
    kq = torch.full((b, T, kvh, per), 9, dtype=torch.uint8, device=be.device); vq = torch.full_like(kq, 9)
    ks = torch.full((b, T, kvh, hd // 32), 0.25, dtype=torch.float16, device=be.device); vs = torch.full_like(ks, 0.25)
    be.ext.fp16_to_q_kv(be.t(k), kq, ks, be.t(v), vq, vs, b, offset, width, 0, none_tensor, none_tensor, wbits)
    # the range the reference actually packs, in tokens (ext_cache.cpp:149-153): the offset moves DOWN to a block boundary, the width
    # grows to a whole number of blocks -- from the ORIGINAL width, so for kv widths below 512 the range may end before
    # offset + width (e.g. dim 256, tokens [33, 35): packed [32, 34)).  Reproduced, not corrected.
    lo, w = offset, width
    if dim % 512:
        while (lo * dim) % 512: lo -= 1
        while (w * dim) % 512: w += 1
    hi = min(lo + w, T)
    pack, unpack = (OM.q4_pack, OM.q4_unpack) if wbits == 4 else (OM.q8_pack, OM.q8_unpack)
    for src, codes, scales in ((k, kq, ks), (v, vq, vs)):
        c, sc = be.n(codes), be.n(scales)
        for i in range(b):
            cw, sw = pack(src[i, lo:hi].reshape(-1))
            assert np.array_equal(c[i, lo:hi].reshape(-1), cw.reshape(-1)), (kvh, wbits, offset, width, lo, hi)
            assert np.array_equal(sc[i, lo:hi].reshape(-1).view(np.uint16), sw.reshape(-1).view(np.uint16))
        assert np.all(c[:, :lo] == 9) and np.all(c[:, hi:] == 9) and np.all(sc[:, :lo] == F16(0.25)) and np.all(sc[:, hi:] == F16(0.25))
    ko = torch.zeros((b, T, kvh, hd), dtype=torch.float16, device=be.device); vo = torch.zeros_like(ko)
    be.ext.q_to_fp16_kv(kq, ko, ks, vq, vo, vs, b, offset, width, 0, none_tensor, none_tensor, wbits)
    for codes, scales, out in ((kq, ks, ko), (vq, vs, vo)):
        c, sc, o = be.n(codes), be.n(scales), be.n(out)
        for i in range(b):
            want = unpack(c[i, lo:hi].reshape(-1), sc[i, lo:hi].reshape(-1)).reshape(hi - lo, kvh, hd)
            assert np.array_equal(o[i, lo:hi].view(np.uint16), want.view(np.uint16))


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("EXL2_TEST_SEEDS", "6")))))        # (more seeds: a longer hunt, by hand)
def test_kv_codec_random_paged_appends(be, seed):
    """Seeded random PAGED appends of the Q4 / Q8 codec (cache.cu:143-195): 1-3 sequences at random lengths (page starts, page ends,
    appends that cross a page boundary), 1-9 new tokens, kv widths 128-1024, shuffled block tables.  Per page the reference widens the
    touched token range to whole 512-element blocks on BOTH sides and clips it to the page: exactly those tokens must hold the
    oracle's codes and scales (bit for bit), everything else must be untouched."""
    rng = np.random.default_rng(41000 + seed)
    kvh = int(rng.choice([1, 2, 4, 8])); hd, ps = 128, 256
    dim = kvh * hd
    b = int(rng.integers(1, 4)); q_len = int(rng.integers(1, 10))
    pages_per_seq = int(rng.integers(2, 4))
    pages = b * pages_per_seq + 1
    wbits = int(rng.choice([4, 8]))
    table = rng.permutation(pages)[:b * pages_per_seq].astype(np.int32).reshape(b, pages_per_seq)
    cap = pages_per_seq * ps
    seqlens = np.array([min(int(rng.choice([0, 1, 3, 250, 255, 256, 257, 300, cap - q_len])), cap - q_len) for _ in range(b)], dtype=np.int32)
    k = rng.standard_normal((pages, ps, kvh, hd)).astype(F16); v = (rng.standard_normal((pages, ps, kvh, hd)) * 0.3).astype(F16)
    per = hd // 2 if wbits == 4 else hd
    kq = torch.full((pages, ps, kvh, per), 9, dtype=torch.uint8, device=be.device); vq = torch.full_like(kq, 9)
    ks = torch.full((pages, ps, kvh, hd // 32), 0.25, dtype=torch.float16, device=be.device); vs = torch.full_like(ks, 0.25)
    be.ext.fp16_to_q_kv(be.t(k), kq, ks, be.t(v), vq, vs, b, 0, q_len, ps, be.t(seqlens), be.t(table), wbits)
    written = np.zeros((pages, ps), dtype=bool)
    for i in range(b):
        for x in range(pages_per_seq):
            a, e = int(seqlens[i]) - ps * x, int(seqlens[i]) - ps * x + q_len
            if e <= 0 or a >= ps:
                continue                                           # (pages the append does not reach)
            if dim % 512:
                while (a * dim) % 512: a -= 1
                while (e * dim) % 512: e += 1
            a, e = max(a, 0), min(e, ps)
            written[table[i, x], a:e] = True
    pack = OM.q4_pack if wbits == 4 else OM.q8_pack
    for src, codes, scales in ((k, kq, ks), (v, vq, vs)):
        c, sc = be.n(codes), be.n(scales)
        cw, sw = pack(src.reshape(-1))
        cw, sw = cw.reshape(pages, ps, -1), sw.reshape(pages, ps, -1)
        what = (kvh, wbits, q_len, seqlens.tolist())
        assert np.array_equal(c.reshape(pages, ps, -1)[written], cw[written]), what
        assert np.array_equal(sc.reshape(pages, ps, -1)[written].view(np.uint16), sw[written].view(np.uint16)), what
        assert np.all(c.reshape(pages, ps, -1)[~written] == 9) and np.all(sc.reshape(pages, ps, -1)[~written] == F16(0.25)), what


def test_q4_cache_paged(be):
    rng = np.random.default_rng(4)
    pages, ps, kvh, hd = 6, 256, 1, 128
    b, q_len = 2, 3
    seqlens = np.array([254, 600], dtype=np.int32)           # first sequence crosses a page boundary
    table = np.array([[3, 1, 0], [2, 4, 5]], dtype=np.int32)
    k = rng.standard_normal((pages, ps, kvh, hd)).astype(F16)
    v = rng.standard_normal((pages, ps, kvh, hd)).astype(F16)
    kq = torch.zeros((pages, ps, kvh, hd // 2), dtype=torch.uint8, device=be.device)
    vq = torch.zeros_like(kq)
    ks = torch.zeros((pages, ps, kvh, hd // 32), dtype=torch.float16, device=be.device)
    vs = torch.zeros_like(ks)
    be.ext.fp16_to_q_kv(be.t(k), kq, ks, be.t(v), vq, vs, b, 0, q_len, ps, be.t(seqlens), be.t(table), 4)
    got = be.n(kq)
    touched = np.zeros((pages, ps), dtype=bool)
    for i in range(b):
        for pos in range(seqlens[i], seqlens[i] + q_len):
            touched[table[i, pos // ps], pos % ps] = True
    # dim = 128 -> 4 tokens per 512-element block: the kernel widens to whole blocks (cache.cu:169-176)
    widened = touched.reshape(pages, ps // 4, 4).any(axis=-1).repeat(4, axis=-1)
    nz = (got.reshape(pages, ps, -1) != 0).any(axis=-1)
    assert np.all(nz[touched]) and not np.any(nz[~widened])
    kp, _ = OM.q4_pack(k.reshape(-1))
    kp = kp.reshape(pages, ps, -1)
    assert np.array_equal(got.reshape(pages, ps, -1)[widened], kp[widened])
    # paged unpack: everything valid in each page
    ko = torch.zeros((pages, ps, kvh, hd), dtype=torch.float16, device=be.device)
    vo = torch.zeros_like(ko)
    kq_full = be.t(kp.reshape(pages, ps, kvh, hd // 2))
    _, ksc = OM.q4_pack(k.reshape(-1))
    ks_full = be.t(ksc.reshape(pages, ps, kvh, hd // 32))
    new_len = (seqlens + q_len).astype(np.int32)
    be.ext.q_to_fp16_kv(kq_full, ko, ks_full, kq_full, vo, ks_full, b, 0, 0, ps, be.t(new_len), be.t(table), 4)
    want_all = OM.q4_unpack(kp.reshape(-1), ksc).reshape(pages, ps, kvh, hd)
    got_o = be.n(ko)
    for i in range(b):
        for pos in range(0, new_len[i]):
            pg = table[i, pos // ps]
            assert np.array_equal(got_o[pg, pos % ps].view(np.uint16), want_all[pg, pos % ps].view(np.uint16))


# ---- attention ---------------------------------------------------------------------------------------------------------

def _attn_tol(want):
    return np.abs(want.astype(np.float32)) * 2.0 ** -9 + 2e-3


@pytest.mark.parametrize("hd,nh,kvh,s", [(128, 4, 4, 1), (128, 8, 2, 1), (64, 4, 1, 2), (128, 2, 1, 5)])
def test_attention_contiguous(be, hd, nh, kvh, s):
    rng = np.random.default_rng(5)
    b, T, past = 2, 96, 70
    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    k = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    v = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    total = past + s
    want = OM.attention(q, k[:, :total], v[:, :total])
    for nsplit in (1, 3):
        out = torch.zeros((b, s, nh, hd), dtype=torch.float16, device=be.device)
        scratch = torch.zeros((be.ext.paged_attn_scratch_bytes(b * s * nh, hd, nsplit) // 4 + 1,), dtype=torch.float32,
                              device=be.device)
        be.ext.paged_attn(be.t(q), be.t(k), be.t(v), out, None, None, len_const=past, len_offset=s, nsplit=nsplit,
                          scratch=scratch)
        got = be.n(out)
        assert np.all(np.abs(got.astype(np.float32) - want.astype(np.float32)) <= _attn_tol(want)), nsplit


@pytest.mark.parametrize("hd,nh,kvh,s,past,window,cap", [(128, 4, 4, 1, 70, 16, 0.0), (128, 8, 2, 1, 70, 5, 3.0), (64, 4, 1, 3, 90, 33, 0.0),
                                                         (128, 2, 1, 5, 60, -1, 4.0), (128, 4, 2, 70, 100, 40, 0.0), (64, 4, 2, 130, 7, 64, 3.0),
                                                         (128, 2, 2, 200, 0, 17, 0.0), (256, 2, 1, 40, 30, -1, 2.0)])
def test_attention_sliding_window_and_softcap(be, hd, nh, kvh, s, past, window, cap):
    """flash-attn's window_size[0] and softcap (the keyword arguments the reference adds for Mistral / Gemma-type checkpoints,
    attn.py:590-600) on the general kernels: decode-shaped (exl2_paged_attn_ex, one and several splits, splits that lie wholly in
    front of the window) and prefill-shaped (exl2_flash_prefill_ex: key tiles in front of a workgroup's window are not even fetched,
    rows whose first visible key sits mid-tile).  Checker: the fp64 oracle with the same two arguments (oracle/modules.py:attention)."""
    rng = np.random.default_rng(hd + s + past)
    b, T = 2, past + s + 9
    q = (2.0 * rng.standard_normal((b, s, nh, hd))).astype(F16)           # (scores large enough for the cap to bite)
    k = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    v = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    total = past + s
    want = OM.attention(q, k[:, :total], v[:, :total], window_left=window, softcap=cap)
    plain = OM.attention(q, k[:, :total], v[:, :total])
    assert np.abs(want.astype(np.float32) - plain.astype(np.float32)).max() > 0.05          # (the arguments do something in every case)
    if s <= 16:
        for nsplit in (1, 3, 7):
            out = torch.full((b, s, nh, hd), 77.0, dtype=torch.float16, device=be.device)
            scratch = torch.zeros((be.ext.paged_attn_scratch_bytes(b * s * nh, hd, nsplit) // 4 + 1,), dtype=torch.float32, device=be.device)
            be.ext.paged_attn(be.t(q), be.t(k), be.t(v), out, None, None, len_const=past, len_offset=s, nsplit=nsplit, scratch=scratch,
                              window_left=window, softcap=cap)
            err = np.abs(be.n(out).astype(np.float32) - want.astype(np.float32))
            assert np.all(err <= _attn_tol(want)), (nsplit, float(err.max()))
    else:
        out = torch.full((b, s, nh, hd), 77.0, dtype=torch.float16, device=be.device)
        assert be.ext.flash_prefill(be.t(q), be.t(k), be.t(v), out, None, None, len_const=past, len_offset=s, window_left=window, softcap=cap)
        err = np.abs(be.n(out).astype(np.float32) - want.astype(np.float32))
        assert np.all(err <= _attn_tol(want)), float(err.max())


def test_flash_attn_shim_window_and_softcap(be):
    """The flash-attn stand-in (dropin/flash_attn) with the reference's keyword arguments: a window that clips and a score cap reach the
    general kernels (round 5 raised); a window that cannot clip keeps the plain route.  Paged call with append + the non-paged call."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("flash_attn_shim_ws", os.path.join(root, "dropin", "flash_attn", "__init__.py"))
    FA = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(FA)
    FA._e = be.ext                                                    # (the emulation build's binding here, libexl2_hip.so's under -m gpu)
    rng = np.random.default_rng(77)
    hd, nh, kvh, ps = 128, 4, 2, 256
    # paged decode step, 300 cached tokens, window 64, cap 3
    b, s = 2, 1
    seqlens = np.array([300, 270], dtype=np.int32)
    table = np.array([[1, 3], [2, 0]], dtype=np.int32)
    kc = rng.standard_normal((4, ps, kvh, hd)).astype(F16); vc = rng.standard_normal((4, ps, kvh, hd)).astype(F16)
    q = (2.0 * rng.standard_normal((b, s, nh, hd))).astype(F16)
    kn = rng.standard_normal((b, s, kvh, hd)).astype(F16); vn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    kct, vct = be.t(kc), be.t(vc)
    got = FA.flash_attn_with_kvcache(be.t(q), kct, vct, k=be.t(kn), v=be.t(vn), cache_seqlens=be.t(seqlens), block_table=be.t(table),
                                     causal=True, window_size=(64, 64), softcap=3.0)
    for i in range(b):
        n = int(seqlens[i])
        kk = np.concatenate([kc[table[i]].reshape(-1, kvh, hd)[:n], kn[i]], 0)[None]
        vv = np.concatenate([vc[table[i]].reshape(-1, kvh, hd)[:n], vn[i]], 0)[None]
        want = OM.attention(q[i:i + 1], kk, vv, window_left=64, softcap=3.0)
        err = np.abs(be.n(got)[i:i + 1].astype(np.float32) - want.astype(np.float32))
        assert np.all(err <= _attn_tol(want)), (i, float(err.max()))
    # non-paged prompt of 90 rows over 40 cached, window 32
    s, past = 90, 40
    q = rng.standard_normal((1, s, nh, hd)).astype(F16)
    k = rng.standard_normal((1, past + s, kvh, hd)).astype(F16); v = rng.standard_normal((1, past + s, kvh, hd)).astype(F16)
    got = FA.flash_attn_func(be.t(q), be.t(k), be.t(v), causal=True, window_size=(32, 32))
    want = OM.attention(q, k, v, window_left=32)
    assert np.all(np.abs(be.n(got).astype(np.float32) - want.astype(np.float32)) <= _attn_tol(want))
    # a window that cannot clip is plain causal attention
    got = FA.flash_attn_func(be.t(q), be.t(k), be.t(v), causal=True, window_size=(4096, 4096))
    want = OM.attention(q, k, v)
    assert np.all(np.abs(be.n(got).astype(np.float32) - want.astype(np.float32)) <= _attn_tol(want))


@pytest.mark.parametrize("hd,nh,kvh,s,past", [(128, 4, 2, 70, 0), (128, 2, 2, 64, 37), (64, 4, 1, 33, 5), (256, 2, 1, 20, 0),
                                              (128, 4, 4, 130, 3)])
def test_flash_prefill_contiguous(be, hd, nh, kvh, s, past):
    """csrc/attn_prefill.hip against attn.py:905-933 semantics (fp64 oracle): many query rows, lower-right causal mask, GQA,
    query tiles of 64 with a ragged last tile, key tiles of 32 with a ragged last tile, rows whose first tiles are fully
    masked, head sizes 64 / 128 / 256.  P is rounded to fp16 before the second product (as flash-attn does): same tolerance as
    the decode kernel."""
    rng = np.random.default_rng(hd + s)
    b, T = 2, 200
    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    k = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    v = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    total = past + s
    want = OM.attention(q, k[:, :total], v[:, :total])
    out = torch.full((b, s, nh, hd), 77.0, dtype=torch.float16, device=be.device)
    assert be.ext.flash_prefill(be.t(q), be.t(k), be.t(v), out, None, None, len_const=past, len_offset=s)
    got = be.n(out)
    assert np.all(np.abs(got.astype(np.float32) - want.astype(np.float32)) <= _attn_tol(want)), \
        float(np.abs(got.astype(np.float32) - want.astype(np.float32)).max())
    # not causal: every query sees every key
    out2 = torch.zeros_like(out)
    assert be.ext.flash_prefill(be.t(q), be.t(k), be.t(v), out2, None, None, len_const=past, len_offset=s, causal=False)
    qq = q.astype(np.float64); kk = k[:, :total].astype(np.float64); vv = v[:, :total].astype(np.float64)
    g = nh // kvh
    want2 = np.zeros((b, s, nh, hd))
    for hh in range(nh):
        sc = np.einsum("bsd,btd->bst", qq[:, :, hh], kk[:, :, hh // g]) * hd ** -0.5
        p = np.exp(sc - sc.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
        want2[:, :, hh] = np.einsum("bst,btd->bsd", p, vv[:, :, hh // g])
    assert np.all(np.abs(be.n(out2).astype(np.float64) - want2) <= _attn_tol(want2.astype(np.float32)))


def test_flash_prefill_paged(be):
    """the same through a block table with per-sequence lengths on the device (the dynamic generator's prefill chunks,
    dynamic.py:1139-1294): keys already appended, query rows are the last s positions of each sequence"""
    rng = np.random.default_rng(8)
    pages, ps, kvh, hd, nh = 6, 256, 2, 128, 4
    b, s = 2, 40
    seqlens = np.array([250, 300], dtype=np.int32)                  # cached tokens BEFORE this chunk
    table = np.array([[2, 0, 4], [1, 3, 5]], dtype=np.int32)
    kc = rng.standard_normal((pages, ps, kvh, hd)).astype(F16)
    vc = rng.standard_normal((pages, ps, kvh, hd)).astype(F16)
    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    out = torch.zeros((b, s, nh, hd), dtype=torch.float16, device=be.device)
    assert be.ext.flash_prefill(be.t(q), be.t(kc), be.t(vc), out, be.t(seqlens), be.t(table), len_const=0, len_offset=s)
    want = OM.paged_attention(q, None, None, kc, vc, seqlens + s, table)
    got = be.n(out)
    assert np.all(np.abs(got.astype(np.float32) - want.astype(np.float32)) <= _attn_tol(want))


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("EXL2_TEST_SEEDS", "6")))))        # (more seeds: a longer hunt, by hand)
def test_flash_prefill_random_shapes(be, seed):
    """Seeded random shapes for the MFMA flash-prefill kernel (round 5: two 16-row query blocks per wave, one page look-up per key
    tile made two tiles ahead): 17-300 query rows, 0-600 cached tokens, GQA 1-8, head_dim 64 / 128, paged (lengths on the device,
    ragged last query / key tiles, page boundaries inside a key tile's range) or contiguous.  Checker: the fp64 oracle."""
    rng = np.random.default_rng(11000 + seed)
    ps = 256
    hd = int(rng.choice([64, 128])); b = int(rng.integers(1, 4)); s = int(rng.choice([17, 40, 64, 65, 128, 129, 200, 300]))
    kvh = int(rng.choice([1, 2])); g = int(rng.choice([1, 2, 4, 8])); nh = kvh * g
    paged = bool(rng.integers(0, 2))
    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    out = torch.full((b, s, nh, hd), 77.0, dtype=torch.float16, device=be.device)
    if paged:
        seqlens = np.array([int(rng.choice([0, 1, 31, 200, 250, 256, 480, 600])) for _ in range(b)], dtype=np.int32)
        pages_per_seq = int((seqlens.max() + s + ps - 1) // ps) + int(rng.integers(0, 2))
        pages = b * pages_per_seq + 1
        table = rng.permutation(pages)[:b * pages_per_seq].astype(np.int32).reshape(b, pages_per_seq)
        kc = rng.standard_normal((pages, ps, kvh, hd)).astype(F16); vc = rng.standard_normal((pages, ps, kvh, hd)).astype(F16)
        what = (hd, b, s, nh, kvh, "paged", seqlens.tolist(), pages_per_seq)
        if not be.ext.flash_prefill(be.t(q), be.t(kc), be.t(vc), out, be.t(seqlens), be.t(table), len_const=0, len_offset=s):
            pytest.skip(f"declined: {what}")
        want = OM.paged_attention(q, None, None, kc, vc, seqlens + s, table)
    else:
        past = int(rng.choice([0, 1, 33, 200, 511]))
        T = past + s + int(rng.integers(0, 40))
        k = rng.standard_normal((b, T, kvh, hd)).astype(F16); v = rng.standard_normal((b, T, kvh, hd)).astype(F16)
        what = (hd, b, s, nh, kvh, "contiguous", past, T)
        if not be.ext.flash_prefill(be.t(q), be.t(k), be.t(v), out, None, None, len_const=past, len_offset=s):
            pytest.skip(f"declined: {what}")
        want = OM.attention(q, k[:, :past + s], v[:, :past + s])
    err = np.abs(be.n(out).astype(np.float32) - want.astype(np.float32))
    assert np.all(err <= _attn_tol(want)), (what, float(err.max()))


def test_attention_paged_with_append(be):
    """flash_attn_with_kvcache contract (attn.py:602-613): append new k/v at cache_seqlens through the block table,
    attend bottom-right causal."""
    rng = np.random.default_rng(6)
    pages, ps, kvh, hd, nh = 5, 256, 2, 128, 4
    b, s = 2, 2
    seqlens = np.array([255, 300], dtype=np.int32)
    table = np.array([[2, 0, 4], [1, 3, 4]], dtype=np.int32)
    kc = rng.standard_normal((pages, ps, kvh, hd)).astype(F16)
    vc = rng.standard_normal((pages, ps, kvh, hd)).astype(F16)
    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    kn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    vn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    kc_ref, vc_ref = kc.copy(), vc.copy()
    want = OM.paged_attention(q, kn, vn, kc_ref, vc_ref, seqlens, table)
    kct, vct = be.t(kc), be.t(vc)
    out = be.ext.flash_attn_with_kvcache(be.t(q), kct, vct, be.t(kn), be.t(vn), be.t(seqlens), be.t(table))
    assert np.array_equal(be.n(kct).view(np.uint16), kc_ref.view(np.uint16))        # appended in place
    assert np.array_equal(be.n(vct).view(np.uint16), vc_ref.view(np.uint16))
    got = be.n(out)
    assert np.all(np.abs(got.astype(np.float32) - want.astype(np.float32)) <= _attn_tol(want))


@pytest.mark.parametrize("hd,nh,kvh,s,rope", [(128, 4, 4, 1, True), (128, 8, 2, 1, True), (64, 4, 2, 2, True),
                                               (128, 4, 2, 3, False), (256, 2, 1, 1, True)])
def test_attention_fused_decode_step(be, hd, nh, kvh, s, rope):
    """One launch == rope_(q, k) + flash_attn_with_kvcache(k=new_k, v=new_v) (attn.py:602-613): bit-exact cache append
    (rope.cu numerics), attention within tolerance, for sequence lengths on both sides of the device-side split choice,
    ticket counters left zeroed; repeated so a stale ticket or partial would show."""
    rng = np.random.default_rng(16)
    pages, ps = 7, 256
    b = 2
    table = np.array([[2, 0, 4], [1, 3, 5]], dtype=np.int32)
    sin, cos = OM.rope_tables(1024, hd, neox=True)
    scratch = torch.zeros((be.ext.paged_attn_scratch_bytes(b * s * nh, hd, 16) // 4 + 1,), dtype=torch.float32,
                          device=be.device)
    counters = torch.zeros((64,), dtype=torch.int32, device=be.device)
    for seqlens in ([0, 5], [255, 300], [700, 127], [513, 640]):
        seqlens = np.array(seqlens, dtype=np.int32)
        kc = rng.standard_normal((pages, ps, kvh, hd)).astype(F16)
        vc = rng.standard_normal((pages, ps, kvh, hd)).astype(F16)
        q = rng.standard_normal((b, s, nh, hd)).astype(F16)
        kn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
        vn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
        q_r = OM.rope_(q, sin, cos, seqlens, neox=True) if rope else q
        k_r = OM.rope_(kn, sin, cos, seqlens, neox=True) if rope else kn
        kc_ref, vc_ref = kc.copy(), vc.copy()
        want = OM.paged_attention(q_r, k_r, vn, kc_ref, vc_ref, seqlens, table)
        kct, vct, qt, knt = be.t(kc), be.t(vc), be.t(q), be.t(kn)
        out = torch.zeros((b, s, nh, hd), dtype=torch.float16, device=be.device)
        ok = be.ext.attn_decode_fused(qt, knt, be.t(vn), kct, vct, out, be.t(sin), be.t(cos), be.t(seqlens), be.t(table),
                                      0, 2 if rope else 0, scratch, counters)
        assert ok
        assert np.array_equal(be.n(qt).view(np.uint16), q.view(np.uint16))          # inputs untouched
        assert np.array_equal(be.n(kct).view(np.uint16), kc_ref.view(np.uint16))
        assert np.array_equal(be.n(vct).view(np.uint16), vc_ref.view(np.uint16))
        got = be.n(out)
        assert np.all(np.abs(got.astype(np.float32) - want.astype(np.float32)) <= _attn_tol(want)), seqlens
        assert int(be.n(counters).sum()) == 0


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("EXL2_TEST_SEEDS", "8")))))        # (more seeds: a longer hunt, by hand)
def test_attention_fused_decode_step_random_shapes(be, seed, monkeypatch):
    """Seeded random shapes for the one-launch FP16 decode step (attn_decode_fused): batch 1-4, 1-4 new tokens, 1-8 query heads per kv
    head, head_dim 64 / 128, lengths from empty to several pages (page boundaries, the last slot of the table), split policy left to
    the library or forced (EXL2_ATT_KPS / EXL2_ATT_NSPLIT_MAX are read per launch), NeoX RoPE or none.  Checker: the oracle's paged
    attention with the append; the caches must hold exactly the rotated new rows afterwards."""
    rng = np.random.default_rng(9000 + seed)
    ps = 256
    hd = int(rng.choice([64, 128])); b = int(rng.integers(1, 5)); s = int(rng.integers(1, 5))
    kvh = int(rng.choice([1, 2, 4])); g = int(rng.choice([1, 2, 4, 8])); nh = kvh * g
    rope = bool(rng.integers(0, 2))
    pages_per_seq = int(rng.integers(1, 7))
    T = ps * pages_per_seq
    pages = b * pages_per_seq + 1
    table = rng.permutation(pages)[:b * pages_per_seq].astype(np.int32).reshape(b, pages_per_seq)
    seqlens = np.array([min(int(rng.choice([0, 1, 17, 255, 256, 257, 511, 700, 1290, T - s])), T - s) for _ in range(b)], dtype=np.int32)
    if rng.integers(0, 2):
        monkeypatch.setenv("EXL2_ATT_KPS", str(int(rng.choice([16, 64, 256]))))
    if rng.integers(0, 2):
        monkeypatch.setenv("EXL2_ATT_NSPLIT_MAX", str(int(rng.choice([1, 2, 5, 16]))))
    sin, cos = OM.rope_tables(T + 8, hd, neox=True)
    kc = (rng.standard_normal((pages, ps, kvh, hd)) * 0.5).astype(F16); vc = rng.standard_normal((pages, ps, kvh, hd)).astype(F16)
    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    kn = (rng.standard_normal((b, s, kvh, hd)) * 0.5).astype(F16); vn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    q_r = OM.rope_(q, sin, cos, seqlens, neox=True) if rope else q
    k_r = OM.rope_(kn, sin, cos, seqlens, neox=True) if rope else kn
    kc_ref, vc_ref = kc.copy(), vc.copy()
    want = OM.paged_attention(q_r, k_r, vn, kc_ref, vc_ref, seqlens, table)
    scratch = torch.zeros((be.ext.paged_attn_scratch_bytes(b * s * nh, hd, 64) // 4 + 1,), dtype=torch.float32, device=be.device)
    counters = torch.zeros((b * s * nh + 64,), dtype=torch.int32, device=be.device)
    kct, vct = be.t(kc), be.t(vc)
    out = torch.zeros((b, s, nh, hd), dtype=torch.float16, device=be.device)
    what = (hd, b, s, nh, kvh, rope, seqlens.tolist(), pages_per_seq)
    ok = be.ext.attn_decode_fused(be.t(q), be.t(kn), be.t(vn), kct, vct, out, be.t(sin), be.t(cos), be.t(seqlens), be.t(table),
                                  0, 2 if rope else 0, scratch, counters)
    if not ok:
        pytest.skip(f"shape declined by the one-launch kernel (the caller takes the three-launch variant): {what}")
    assert np.array_equal(be.n(kct).view(np.uint16), kc_ref.view(np.uint16)), what
    assert np.array_equal(be.n(vct).view(np.uint16), vc_ref.view(np.uint16)), what
    err = np.abs(be.n(out).astype(np.float32) - want.astype(np.float32))
    assert np.all(err <= _attn_tol(want)), (what, float(err.max()))
    assert int(be.n(counters).astype(np.int64).sum()) == 0


@pytest.mark.parametrize("nh,kvh,ctx", [(8, 1, 5000), (4, 4, 3000), (8, 2, 1300)])
def test_attention_fused_long_context_many_splits(be, nh, kvh, ctx):
    """Contexts long enough for the second slope of the split policy (few KV heads: 16 splits of 64 keys, then a split per 256 keys) and
    for the batched merge of more than 8 partials (csrc/attn_merge.h): one sequence, one new token, no rotation, against the oracle."""
    rng = np.random.default_rng(ctx)
    hd, ps = 128, 256
    pages = (ctx + ps) // ps + 1
    table = rng.permutation(pages).astype(np.int32).reshape(1, pages)
    kc = (rng.standard_normal((pages, ps, kvh, hd)) * 0.5).astype(F16)
    vc = rng.standard_normal((pages, ps, kvh, hd)).astype(F16)
    q = rng.standard_normal((1, 1, nh, hd)).astype(F16)
    kn = (rng.standard_normal((1, 1, kvh, hd)) * 0.5).astype(F16)
    vn = rng.standard_normal((1, 1, kvh, hd)).astype(F16)
    seqlens = np.array([ctx], dtype=np.int32)
    kc_ref, vc_ref = kc.copy(), vc.copy()
    want = OM.paged_attention(q, kn, vn, kc_ref, vc_ref, seqlens, table)
    scratch = torch.zeros((be.ext.paged_attn_scratch_bytes(nh, hd, 64) // 4 + 1,), dtype=torch.float32, device=be.device)
    counters = torch.zeros((64,), dtype=torch.int32, device=be.device)
    kct, vct = be.t(kc), be.t(vc)
    for _ in range(2):                                           # twice: the tickets of the first launch must have been left zeroed
        out = torch.zeros((1, 1, nh, hd), dtype=torch.float16, device=be.device)
        assert be.ext.attn_decode_fused(be.t(q), be.t(kn), be.t(vn), kct, vct, out, None, None, be.t(seqlens), be.t(table), 0, 0,
                                        scratch, counters)
        got = be.n(out)
        assert np.all(np.abs(got.astype(np.float32) - want.astype(np.float32)) <= _attn_tol(want))
        assert int(be.n(counters).sum()) == 0
    assert np.array_equal(be.n(kct).view(np.uint16), kc_ref.view(np.uint16))


def test_attention_fused_contiguous_and_fallback(be):
    rng = np.random.default_rng(17)
    b, T, past, hd, nh, kvh, s = 2, 512, 300, 128, 4, 2, 2
    sin, cos = OM.rope_tables(1024, hd, neox=True)
    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    k = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    v = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    kn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    vn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    pos = np.full((b,), past, dtype=np.int32)
    q_r, k_r = OM.rope_(q, sin, cos, pos, neox=True), OM.rope_(kn, sin, cos, pos, neox=True)
    k_full, v_full = k.copy(), v.copy()
    k_full[:, past:past + s], v_full[:, past:past + s] = k_r, vn
    want = OM.attention(q_r, k_full[:, :past + s], v_full[:, :past + s])
    kt, vt = be.t(k), be.t(v)
    out = torch.zeros((b, s, nh, hd), dtype=torch.float16, device=be.device)
    scratch = torch.zeros((be.ext.paged_attn_scratch_bytes(b * s * nh, hd, 16) // 4 + 1,), dtype=torch.float32,
                          device=be.device)
    counters = torch.zeros((64,), dtype=torch.int32, device=be.device)
    assert be.ext.attn_decode_fused(be.t(q), be.t(kn), be.t(vn), kt, vt, out, be.t(sin), be.t(cos), None, None, past, 2,
                                    scratch, counters)
    assert np.array_equal(be.n(kt).view(np.uint16), k_full.view(np.uint16))
    got = be.n(out)
    assert np.all(np.abs(got.astype(np.float32) - want.astype(np.float32)) <= _attn_tol(want))
    # shapes outside the fused kernel report "not handled" (the caller then uses the three-launch path)
    assert not be.ext.attn_decode_fused(be.t(q), be.t(kn), be.t(vn), kt, vt, out, be.t(sin), be.t(cos), None, None, past,
                                        1, scratch, counters)                      # GPT-J style rope
    big_q = torch.zeros((1, 40, nh, hd), dtype=torch.float16, device=be.device)
    big_k = torch.zeros((1, 40, kvh, hd), dtype=torch.float16, device=be.device)
    assert not be.ext.attn_decode_fused(big_q, big_k, big_k, kt, vt, big_q.clone(), be.t(sin), be.t(cos), None, None, 0, 2,
                                        scratch, counters)                         # prefill-sized


@pytest.mark.parametrize("hd,nh,kvh,s,paged", [(128, 4, 4, 1, False), (128, 8, 2, 1, True), (64, 8, 8, 2, False), (256, 2, 2, 1, True)])
def test_attention_from_q4_cache(be, hd, nh, kvh, s, paged):
    """attn_q4.hip: attention straight from Q4 codes + scales == attention over the unpacked cache (oracle q4_unpack of
    the same codes), i.e. q_to_fp16_kv + attention of the reference (cache.py:472-514) without the unpack."""
    rng = np.random.default_rng(40)
    b, T = 2, 512
    total = np.array([300, 511], dtype=np.int32)
    kf = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    vf = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    kq, ks = OM.q4_pack(kf.reshape(-1))
    vq, vs = OM.q4_pack(vf.reshape(-1))
    k_un = OM.q4_unpack(kq, ks).reshape(b, T, kvh, hd)
    v_un = OM.q4_unpack(vq, vs).reshape(b, T, kvh, hd)
    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    want = np.stack([OM.attention(q[i:i + 1], k_un[i:i + 1, :total[i]], v_un[i:i + 1, :total[i]])[0] for i in range(b)])
    kqt = be.t(kq.reshape(b, T, kvh, hd // 2)); kst = be.t(ks.reshape(b, T, kvh, hd // 32))
    vqt = be.t(vq.reshape(b, T, kvh, hd // 2)); vst = be.t(vs.reshape(b, T, kvh, hd // 32))
    scratch = torch.zeros((be.ext.paged_attn_scratch_bytes(b * s * nh, hd, 16) // 4 + 1,), dtype=torch.float32, device=be.device)
    seq = be.t(total - s)
    for nsplit in (1, 3):
        out = torch.zeros((b, s, nh, hd), dtype=torch.float16, device=be.device)
        if paged:
            ps = 256
            table = np.array([[0, 1], [2, 3]], dtype=np.int32)                      # contiguous pages of each sequence
            view = lambda x: x.view(-1, ps, kvh, x.shape[-1])
            ok = be.ext.paged_attn_q4(be.t(q), view(kqt), view(kst), view(vqt), view(vst), out, seq, be.t(table),
                                      len_const=0, len_offset=s, nsplit=nsplit, scratch=scratch)
        else:
            ok = be.ext.paged_attn_q4(be.t(q), kqt, kst, vqt, vst, out, seq, None, len_const=0, len_offset=s, nsplit=nsplit,
                                      scratch=scratch)
        assert ok
        got = be.n(out)
        assert np.all(np.abs(got.astype(np.float32) - want.astype(np.float32)) <= _attn_tol(want) + 2e-3), (nsplit, float(np.abs(got.astype(np.float32) - want.astype(np.float32)).max()))


@pytest.mark.parametrize("hd,nh,kvh,s", [(128, 4, 4, 1), (128, 8, 2, 3), (64, 4, 4, 2), (256, 2, 1, 1)])
def test_attention_from_q4_cache_with_fp16_new_tokens(be, hd, nh, kvh, s):
    """The step's own K/V are attended in fp16 (the reference quantises them after attention, cache.py:517-556): keys
    below total - s come from the Q4 codes, the last s from k_new / v_new."""
    rng = np.random.default_rng(41)
    b, T = 2, 256
    total = np.array([77, 256], dtype=np.int32)
    kf = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    vf = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    kq, ks = OM.q4_pack(kf.reshape(-1)); vq, vs = OM.q4_pack(vf.reshape(-1))
    k_un = OM.q4_unpack(kq, ks).reshape(b, T, kvh, hd); v_un = OM.q4_unpack(vq, vs).reshape(b, T, kvh, hd)
    kn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    vn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    want = []
    for i in range(b):
        kk = np.concatenate([k_un[i, :total[i] - s], kn[i]])[None]
        vv = np.concatenate([v_un[i, :total[i] - s], vn[i]])[None]
        want.append(OM.attention(q[i:i + 1], kk, vv)[0])
    want = np.stack(want)
    scratch = torch.zeros((be.ext.paged_attn_scratch_bytes(b * s * nh, hd, 16) // 4 + 1,), dtype=torch.float32, device=be.device)
    for nsplit in (1, 2):
        out = torch.zeros((b, s, nh, hd), dtype=torch.float16, device=be.device)
        assert be.ext.paged_attn_q4(be.t(q), be.t(kq.reshape(b, T, kvh, hd // 2)), be.t(ks.reshape(b, T, kvh, hd // 32)),
                                    be.t(vq.reshape(b, T, kvh, hd // 2)), be.t(vs.reshape(b, T, kvh, hd // 32)), out,
                                    be.t(total - s), None, len_const=0, len_offset=s, nsplit=nsplit, scratch=scratch,
                                    k_new=be.t(kn), v_new=be.t(vn))
        got = be.n(out)
        err = np.abs(got.astype(np.float32) - want.astype(np.float32))
        assert np.all(err <= _attn_tol(want) + 2e-3), (nsplit, float(err.max()))


@pytest.mark.parametrize("hd,nh,kvh,s", [(128, 8, 2, 1), (128, 16, 2, 1), (64, 4, 4, 3)])
def test_attention_from_q4_cache_merges_its_splits_in_the_launch(be, hd, nh, kvh, s, monkeypatch):
    """exl2_paged_attn_q4_merged: with tickets the last split of a (sequence, kv head, row block) to finish merges the partials
    inside the launch -- bit-identical to the two-launch form (same merge function), tickets left at zero, also when a sequence is
    short enough for ONE split while the grid was sized for many (HIP graph: the grid is fixed, the lengths are not)."""
    monkeypatch.setenv("EXL2_Q4_RB", "4")                              # (several rows per workgroup and ticket: the long-capacity form)
    rng = np.random.default_rng(47)
    b, T = 3, 2048
    total = np.array([1500, 40, 700], dtype=np.int32)
    kf = rng.standard_normal((b, T, kvh, hd)).astype(F16); vf = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    kq, ks = OM.q4_pack(kf.reshape(-1)); vq, vs = OM.q4_pack(vf.reshape(-1))
    kn = rng.standard_normal((b, s, kvh, hd)).astype(F16); vn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    scratch = torch.zeros((be.ext.paged_attn_scratch_bytes(b * s * nh, hd, 16) // 4 + 1,), dtype=torch.float32, device=be.device)
    counters = torch.zeros((256,), dtype=torch.int32, device=be.device)
    args = (be.t(q), be.t(kq.reshape(b, T, kvh, hd // 2)), be.t(ks.reshape(b, T, kvh, hd // 32)),
            be.t(vq.reshape(b, T, kvh, hd // 2)), be.t(vs.reshape(b, T, kvh, hd // 32)))
    for nsplit in (1, 4, 7):
        outs = []
        for cnt in (None, counters):
            out = torch.zeros((b, s, nh, hd), dtype=torch.float16, device=be.device)
            assert be.ext.paged_attn_q4(*args, out, be.t(total - s), None, len_const=0, len_offset=s, nsplit=nsplit, scratch=scratch,
                                        k_new=be.t(kn), v_new=be.t(vn), counters=cnt)
            outs.append(be.n(out))
        assert np.array_equal(outs[0].view(np.uint16), outs[1].view(np.uint16)), nsplit
        assert int(be.n(counters).astype(np.int64).sum()) == 0
    # too few tickets: the two-launch form runs (same result as without any)
    res = []
    for cnt in (None, counters[:1]):
        out = torch.zeros((b, s, nh, hd), dtype=torch.float16, device=be.device)
        assert be.ext.paged_attn_q4(*args, out, be.t(total - s), None, len_const=0, len_offset=s, nsplit=4, scratch=scratch,
                                    k_new=be.t(kn), v_new=be.t(vn), counters=cnt)
        res.append(be.n(out))
    assert np.array_equal(res[0].view(np.uint16), res[1].view(np.uint16))


@pytest.mark.parametrize("hd,nh,kvh,s,paged", [(128, 4, 4, 1, True), (64, 8, 8, 1, False), (128, 8, 2, 2, True)])
def test_attention_from_q4_cache_in_consumer_order(be, hd, nh, kvh, s, paged):
    """exl2_paged_attn_q4(out_invperm=...): the chained decode step's form -- feature n of a token row lands at
    out[row, out_invperm[n]] (o_proj's packed K order, q_matrix.cu:606-642 act-order), the step's own K/V attended in fp16.
    Checker: the oracle's attention over q4_unpack of the same codes + the fp16 new rows, permuted on the host."""
    rng = np.random.default_rng(43)
    b, T = 2, 512
    total = np.array([130, 400], dtype=np.int32)
    kf = rng.standard_normal((b, T, kvh, hd)).astype(F16); vf = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    kq, ks = OM.q4_pack(kf.reshape(-1)); vq, vs = OM.q4_pack(vf.reshape(-1))
    k_un = OM.q4_unpack(kq, ks).reshape(b, T, kvh, hd); v_un = OM.q4_unpack(vq, vs).reshape(b, T, kvh, hd)
    kn = rng.standard_normal((b, s, kvh, hd)).astype(F16); vn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    want = []
    for i in range(b):
        kk = np.concatenate([k_un[i, :total[i] - s], kn[i]])[None]; vv = np.concatenate([v_un[i, :total[i] - s], vn[i]])[None]
        want.append(OM.attention(q[i:i + 1], kk, vv)[0])
    want = np.stack(want).reshape(b * s, nh * hd)
    invperm = rng.permutation(nh * hd).astype(np.uint16)
    want_p = np.zeros_like(want); want_p[:, invperm] = want                       # out[row, invperm[n]] = natural[row, n]
    inv_t = be.t(invperm.view(np.int16))                                          # (same 16-bit patterns; the kernel reads u16)
    scratch = torch.zeros((be.ext.paged_attn_scratch_bytes(b * s * nh, hd, 16) // 4 + 1,), dtype=torch.float32, device=be.device)
    kqt = be.t(kq.reshape(b, T, kvh, hd // 2)); kst = be.t(ks.reshape(b, T, kvh, hd // 32))
    vqt = be.t(vq.reshape(b, T, kvh, hd // 2)); vst = be.t(vs.reshape(b, T, kvh, hd // 32))
    table = None
    if paged:
        view = lambda x: x.view(-1, 256, kvh, x.shape[-1])
        kqt, kst, vqt, vst = view(kqt), view(kst), view(vqt), view(vst)
        table = be.t(np.array([[0, 1], [2, 3]], dtype=np.int32))
    for nsplit in (1, 3):
        out = torch.zeros((b, s, nh, hd), dtype=torch.float16, device=be.device)
        assert be.ext.paged_attn_q4(be.t(q), kqt, kst, vqt, vst, out, be.t(total - s), table, len_const=0, len_offset=s,
                                    nsplit=nsplit, scratch=scratch, k_new=be.t(kn), v_new=be.t(vn), out_invperm=inv_t.data_ptr())
        got = be.n(out).reshape(b * s, nh * hd)
        err = np.abs(got.astype(np.float32) - want_p.astype(np.float32))
        assert np.all(err <= _attn_tol(want_p) + 2e-3), (nsplit, float(err.max()))


@pytest.mark.gpu
def test_attention_fused_handoff_stress():
    """The split hand-off (ticket + agent-scope fences) under real concurrency: many back-to-back launches on the same
    scratch must keep matching the three-launch path."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from exllamav2_amd.ext import ext_c as ext
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    pages, ps, kvh, hd, nh, b, s = 40, 256, 32, 128, 32, 1, 1
    table = torch.arange(pages, dtype=torch.int32, device=dev).view(1, pages)
    sin, cos = (torch.from_numpy(x).to(dev) for x in OM.rope_tables(pages * ps, hd, neox=True))
    kc = torch.randn((pages, ps, kvh, hd), generator=g).half().to(dev)
    vc = torch.randn((pages, ps, kvh, hd), generator=g).half().to(dev)
    scratch = torch.zeros((ext.paged_attn_scratch_bytes(b * s * nh, hd, 16) // 4 + 1,), dtype=torch.float32, device=dev)
    scratch2 = torch.zeros_like(scratch)
    counters = torch.zeros((64,), dtype=torch.int32, device=dev)
    worst = 0.0
    for it in range(200):
        n = int(torch.randint(0, pages * ps - 2, (1,), generator=g))
        sl = torch.tensor([n], dtype=torch.int32, device=dev)
        q = torch.randn((b, s, nh, hd), generator=g).half().to(dev)
        kn = torch.randn((b, s, kvh, hd), generator=g).half().to(dev)
        vn = torch.randn((b, s, kvh, hd), generator=g).half().to(dev)
        out = torch.zeros_like(q)
        assert ext.attn_decode_fused(q, kn, vn, kc, vc, out, sin, cos, sl, table, 0, 2, scratch, counters)
        q2, k2, out2 = q.clone(), kn.clone(), torch.zeros_like(q)
        ext.rope_kv_append(q2, k2, vn, kc, vc, sin, cos, 0, sl, table, 2)
        ext.paged_attn(q2, kc, vc, out2, sl, table, len_const=0, len_offset=s, scratch=scratch2)
        worst = max(worst, float((out.float() - out2.float()).abs().max()))
    assert worst <= 2e-3, worst
    assert int(counters.sum()) == 0


@pytest.mark.parametrize("s,hd", [(2, 128), (40, 128), (40, 64)])       # (>= 64 rows, NeoX: the one-workgroup-per-token kernel of round 6, 16-byte accesses)
@pytest.mark.parametrize("neox", [True, False])
def test_rope_kv_append(be, neox, s, hd):
    rng = np.random.default_rng(7)
    pages, ps, kvh, nh = 3, 256, 2, 4
    b = 2
    past = np.array([10, 255 if s == 2 else 230], dtype=np.int32)      # (the second sequence crosses a page boundary)
    table = np.array([[2, 0], [1, 0]], dtype=np.int32)
    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    kn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    vn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    sin, cos = OM.rope_tables(512, hd, neox=neox)
    kc = torch.zeros((pages, ps, kvh, hd), dtype=torch.float16, device=be.device)
    vc = torch.zeros_like(kc)
    qt, kt = be.t(q), be.t(kn)
    be.ext.rope_kv_append(qt, kt, be.t(vn), kc, vc, be.t(sin), be.t(cos), 0, be.t(past), be.t(table), 2 if neox else 1)
    q_want = OM.rope_(q, sin, cos, past, neox=neox)
    k_want = OM.rope_(kn, sin, cos, past, neox=neox)
    assert np.array_equal(be.n(qt).view(np.uint16), q_want.view(np.uint16))
    assert np.array_equal(be.n(kt).view(np.uint16), k_want.view(np.uint16))
    kcn, vcn = be.n(kc), be.n(vc)
    for i in range(b):
        for j in range(s):
            pos = past[i] + j
            pg = table[i, pos // ps]
            assert np.array_equal(kcn[pg, pos % ps].view(np.uint16), k_want[i, j].view(np.uint16))
            assert np.array_equal(vcn[pg, pos % ps].view(np.uint16), vn[i, j].view(np.uint16))


@pytest.mark.parametrize("neox", [True, False])
@pytest.mark.parametrize("paged", [True, False])
def test_rope_quant_append_q4(be, neox, paged):
    """exl2_rope_quant_append_q4: RoPE on q / new k in place + Q4 pack of the rotated k and of v at device-side positions in ONE launch.
    Checkers: the oracle's rope_ (bit-exact, like rope_kv_append) and the oracle's q4_pack of the rotated rows (codes and scales
    bit-exact: cache_q.cuh numerics); rows the step did not touch stay untouched.  Then the same through the two entry points it
    replaces (rope_kv_append into fp16 staging pages + fp16_to_q_kv): identical bytes."""
    rng = np.random.default_rng(9)
    pages, ps, kvh, hd, nh = 4, 256, 4, 128, 8
    b, s = 2, 3
    past = np.array([10, 254], dtype=np.int32)                # the second sequence crosses a page boundary
    table = np.array([[2, 0], [1, 3]], dtype=np.int32)
    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    kn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    vn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    sin, cos = OM.rope_tables(1024, hd, neox=neox)
    if paged:
        shape, T = (pages, ps), ps
        slot = lambda i, pos: (table[i, pos // ps], pos % ps)
    else:
        shape, T = (b, 2 * ps), 2 * ps
        slot = lambda i, pos: (i, pos)
    mk = lambda last, dt: torch.full(shape + (kvh, last), 7 if dt == torch.uint8 else 0.5, dtype=dt, device=be.device)
    kq, vq = mk(hd // 2, torch.uint8), mk(hd // 2, torch.uint8)
    ks, vs = mk(hd // 32, torch.float16), mk(hd // 32, torch.float16)
    qt, kt = be.t(q), be.t(kn)
    style = 2 if neox else 1
    assert be.ext.rope_quant_append_q4(qt, kt, be.t(vn), kq, ks, vq, vs, be.t(sin), be.t(cos), 0, be.t(past),
                                       be.t(table) if paged else None, style)
    q_want = OM.rope_(q, sin, cos, past, neox=neox)
    k_want = OM.rope_(kn, sin, cos, past, neox=neox)
    assert np.array_equal(be.n(qt).view(np.uint16), q_want.view(np.uint16))
    assert np.array_equal(be.n(kt).view(np.uint16), k_want.view(np.uint16))
    got = [be.n(x) for x in (kq, ks, vq, vs)]
    touched = np.zeros(shape, dtype=bool)
    for i in range(b):
        for j in range(s):
            a0, a1 = slot(i, past[i] + j)
            touched[a0, a1] = True
            for rows, codes, scales in ((k_want[i, j], got[0], got[1]), (vn[i, j], got[2], got[3])):
                c_want, s_want = OM.q4_pack(rows.reshape(-1))
                assert np.array_equal(codes[a0, a1].reshape(-1), c_want)
                assert np.array_equal(scales[a0, a1].reshape(-1).view(np.uint16), s_want.view(np.uint16))
    assert np.all(got[0][~touched] == 7) and np.all(got[2][~touched] == 7) and np.all(got[1][~touched] == F16(0.5))
    # the two launches it replaces leave the same bytes
    kc = torch.zeros(shape + (kvh, hd), dtype=torch.float16, device=be.device)
    vc = torch.zeros_like(kc)
    kq2, vq2, ks2, vs2 = mk(hd // 2, torch.uint8), mk(hd // 2, torch.uint8), mk(hd // 32, torch.float16), mk(hd // 32, torch.float16)
    q2, k2 = be.t(q), be.t(kn)
    if paged:
        be.ext.rope_kv_append(q2, k2, be.t(vn), kc, vc, be.t(sin), be.t(cos), 0, be.t(past), be.t(table), style)
        be.ext.fp16_to_q_kv(kc, kq2, ks2, vc, vq2, vs2, b, 0, s, ps, be.t(past), be.t(table), 4)
        assert np.array_equal(be.n(q2).view(np.uint16), be.n(qt).view(np.uint16))
        for new, old in zip(got, (kq2, ks2, vq2, vs2)):
            o = be.n(old)
            assert np.array_equal(new[touched].view(np.uint8), o[touched].view(np.uint8))
    # shapes it does not cover are declined, not approximated
    q64 = torch.zeros((1, 1, 2, 64), dtype=torch.float16, device=be.device)
    k64 = torch.zeros((1, 1, 2, 64), dtype=torch.float16, device=be.device)
    c64 = torch.zeros((1, 256, 2, 32), dtype=torch.uint8, device=be.device)
    s64 = torch.zeros((1, 256, 2, 2), dtype=torch.float16, device=be.device)
    sin64, cos64 = OM.rope_tables(16, 64, neox=True)
    assert not be.ext.rope_quant_append_q4(q64, k64, k64, c64, s64, c64, s64, be.t(sin64), be.t(cos64), 3, None, None, 2)


@pytest.mark.parametrize("rows_per_wg", ["", "4", "2"])
@pytest.mark.parametrize("nh,kvh,s,paged,neox", [(8, 2, 1, True, True), (8, 4, 3, True, False), (16, 2, 2, False, True), (4, 4, 1, False, True)])
def test_attention_q4_decode_step_in_one_launch(be, nh, kvh, s, paged, neox, rows_per_wg, monkeypatch):
    """exl2_attn_q4_decode_fused: RoPE + Q4 pack of the step's rows + attention over codes and the step's own fp16 rows + split merge in
    ONE launch, against the two launches it replaces (rope_quant_append_q4, itself checked against the oracle above, + paged_attn_q4 with
    tickets): attention output, codes and scales bit-identical; q and k_new are left unrotated; tickets left at zero.  Sequences of 1
    and of many splits in one call."""
    # (query rows per workgroup: by default ONE while the sequence's capacity is <= 4096 keys -- every table here --, else up to four;
    # EXL2_Q4_RB forces the other instantiations)
    if rows_per_wg:
        monkeypatch.setenv("EXL2_Q4_RB", rows_per_wg)
    rng = np.random.default_rng(53 + nh + s)
    hd, ps, b = 128, 256, 3
    past = np.array([700, 3, 255], dtype=np.int32)
    pages_per_seq = 4
    T = ps * pages_per_seq
    table = rng.permutation(b * pages_per_seq).astype(np.int32).reshape(b, pages_per_seq)
    shape = (b * pages_per_seq, ps) if paged else (b, T)
    kf = rng.standard_normal(shape + (kvh, hd)).astype(F16); vf = rng.standard_normal(shape + (kvh, hd)).astype(F16)
    kq0, ks0 = OM.q4_pack(kf.reshape(-1)); vq0, vs0 = OM.q4_pack(vf.reshape(-1))
    mkc = lambda x: be.t(x.reshape(shape + (kvh, hd // 2)).copy())
    mks = lambda x: be.t(x.reshape(shape + (kvh, hd // 32)).copy())
    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    kn = rng.standard_normal((b, s, kvh, hd)).astype(F16); vn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    sin, cos = OM.rope_tables(2048, hd, neox=neox)
    style = 2 if neox else 1
    scratch = torch.zeros((be.ext.paged_attn_scratch_bytes(b * s * nh, hd, 16) // 4 + 1,), dtype=torch.float32, device=be.device)
    counters = torch.zeros((256,), dtype=torch.int32, device=be.device)
    sl, bt = be.t(past), (be.t(table) if paged else None)
    for nsplit in (1, 4):
        # two launches
        c2 = [mkc(kq0), mks(ks0), mkc(vq0), mks(vs0)]
        q2, k2 = be.t(q), be.t(kn)
        assert be.ext.rope_quant_append_q4(q2, k2, be.t(vn), *c2, be.t(sin), be.t(cos), 0, sl, bt, style)
        out2 = torch.zeros((b, s, nh, hd), dtype=torch.float16, device=be.device)
        assert be.ext.paged_attn_q4(q2, *c2, out2, sl, bt, len_const=0, len_offset=s, nsplit=nsplit, scratch=scratch,
                                    k_new=k2, v_new=be.t(vn), counters=counters)
        # one launch
        c1 = [mkc(kq0), mks(ks0), mkc(vq0), mks(vs0)]
        q1, k1 = be.t(q), be.t(kn)
        out1 = torch.zeros((b, s, nh, hd), dtype=torch.float16, device=be.device)
        assert be.ext.attn_q4_decode_fused(q1, k1, be.t(vn), *c1, out1, be.t(sin), be.t(cos), sl, bt, 0, style, scratch, counters,
                                           nsplit=nsplit)
        assert np.array_equal(be.n(q1).view(np.uint16), q.view(np.uint16)) and np.array_equal(be.n(k1).view(np.uint16), kn.view(np.uint16))
        assert np.array_equal(be.n(out1).view(np.uint16), be.n(out2).view(np.uint16)), nsplit
        for x1, x2 in zip(c1, c2):
            assert np.array_equal(be.n(x1).view(np.uint8), be.n(x2).view(np.uint8))
        assert int(be.n(counters).astype(np.int64).sum()) == 0
    # the oracle, end to end: attention over the unpacked old keys + the rotated new rows
    k_un = OM.q4_unpack(kq0, ks0).reshape(shape + (kvh, hd)); v_un = OM.q4_unpack(vq0, vs0).reshape(shape + (kvh, hd))
    q_rot = OM.rope_(q, sin, cos, past, neox=neox); k_rot = OM.rope_(kn, sin, cos, past, neox=neox)
    got = be.n(out1)
    for i in range(b):
        if paged:
            ko = np.concatenate([k_un[table[i, pg]] for pg in range(pages_per_seq)])[:past[i]]
            vo = np.concatenate([v_un[table[i, pg]] for pg in range(pages_per_seq)])[:past[i]]
        else:
            ko, vo = k_un[i, :past[i]], v_un[i, :past[i]]
        want = OM.attention(q_rot[i:i + 1], np.concatenate([ko, k_rot[i]])[None], np.concatenate([vo, vn[i]])[None])[0]
        err = np.abs(got[i].astype(np.float32) - want.astype(np.float32))
        assert np.all(err <= _attn_tol(want) + 2e-3), (i, float(err.max()))
    # shapes it does not cover are declined: head_dim 64 with an ODD number of kv heads (a wave packs the rows of two adjacent heads;
    # even counts run the one-launch form since round 6 -- tests/test_chain.py::test_q4_cache_decodes_on_the_chain checks it against the oracle)
    one = lambda x: x[..., :1, :].contiguous()
    assert not be.ext.attn_q4_decode_fused(be.t(q[:, :, :2, :64].copy()), be.t(kn[:, :, :1, :64].copy()), be.t(vn[:, :, :1, :64].copy()),
                                           *[one(x[..., :x.shape[-1] // 2]) for x in c1], torch.zeros((b, s, 2, 64), dtype=torch.float16, device=be.device),
                                           be.t(sin[:, :64].copy()), be.t(cos[:, :64].copy()), sl, bt, 0, style, scratch, counters)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("EXL2_TEST_SEEDS", "8")))))        # (more seeds: a longer hunt, by hand)
def test_attention_q4_decode_step_random_shapes(be, seed, monkeypatch):
    """Seeded random shapes for the one-launch Q4 decode step (exl2_attn_q4_decode_fused): batch 1-4, 1-8 new tokens, 1-8 query heads per
    kv head, paged or contiguous, lengths from empty to several splits, either RoPE style, every row blocking (1 / 2 / 4 query rows per
    workgroup), split counts 1-8.  Checkers: the two launches it replaces (bit-identical output, codes, scales) and the oracle's
    attention over the unpacked old keys + the rotated new rows."""
    rng = np.random.default_rng(7000 + seed)
    hd, ps = 128, 256
    b = int(rng.integers(1, 5)); s = int(rng.integers(1, 9))
    kvh = int(rng.choice([1, 2, 4])); g = int(rng.choice([1, 2, 4, 8])); nh = kvh * g
    if s * g > 64:
        g = max(1, 64 // s // 1); g = 1 << (g.bit_length() - 1); nh = kvh * g
    paged = bool(rng.integers(0, 2)); neox = bool(rng.integers(0, 2))
    pages_per_seq = int(rng.integers(2, 6))
    T = ps * pages_per_seq
    past = np.array([int(rng.choice([0, 1, 30, 255, 256, 300, T - s])) for _ in range(b)], dtype=np.int32)
    past = np.minimum(past, T - s).astype(np.int32)
    monkeypatch.setenv("EXL2_Q4_RB", str(int(rng.choice([1, 2, 4]))))
    nsplit = int(rng.choice([1, 2, 3, 8]))
    table = rng.permutation(b * pages_per_seq).astype(np.int32).reshape(b, pages_per_seq)
    shape = (b * pages_per_seq, ps) if paged else (b, T)
    kf = rng.standard_normal(shape + (kvh, hd)).astype(F16); vf = rng.standard_normal(shape + (kvh, hd)).astype(F16)
    kq0, ks0 = OM.q4_pack(kf.reshape(-1)); vq0, vs0 = OM.q4_pack(vf.reshape(-1))
    mkc = lambda x: be.t(x.reshape(shape + (kvh, hd // 2)).copy())
    mks = lambda x: be.t(x.reshape(shape + (kvh, hd // 32)).copy())
    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    kn = rng.standard_normal((b, s, kvh, hd)).astype(F16); vn = rng.standard_normal((b, s, kvh, hd)).astype(F16)
    sin, cos = OM.rope_tables(T + 8, hd, neox=neox)
    style = 2 if neox else 1
    scratch = torch.zeros((be.ext.paged_attn_scratch_bytes(b * s * nh, hd, 16) // 4 + 1,), dtype=torch.float32, device=be.device)
    counters = torch.zeros((b * s * nh + 8,), dtype=torch.int32, device=be.device)
    sl, bt = be.t(past), (be.t(table) if paged else None)
    c2 = [mkc(kq0), mks(ks0), mkc(vq0), mks(vs0)]
    q2, k2 = be.t(q), be.t(kn)
    assert be.ext.rope_quant_append_q4(q2, k2, be.t(vn), *c2, be.t(sin), be.t(cos), 0, sl, bt, style)
    out2 = torch.zeros((b, s, nh, hd), dtype=torch.float16, device=be.device)
    assert be.ext.paged_attn_q4(q2, *c2, out2, sl, bt, len_const=0, len_offset=s, nsplit=nsplit, scratch=scratch,
                                k_new=k2, v_new=be.t(vn), counters=counters)
    c1 = [mkc(kq0), mks(ks0), mkc(vq0), mks(vs0)]
    out1 = torch.zeros((b, s, nh, hd), dtype=torch.float16, device=be.device)
    assert be.ext.attn_q4_decode_fused(be.t(q), be.t(kn), be.t(vn), *c1, out1, be.t(sin), be.t(cos), sl, bt, 0, style, scratch, counters,
                                       nsplit=nsplit)
    what = (b, s, nh, kvh, paged, neox, past.tolist(), nsplit)
    assert np.array_equal(be.n(out1).view(np.uint16), be.n(out2).view(np.uint16)), what
    for x1, x2 in zip(c1, c2):
        assert np.array_equal(be.n(x1).view(np.uint8), be.n(x2).view(np.uint8)), what
    assert int(be.n(counters).astype(np.int64).sum()) == 0
    k_un = OM.q4_unpack(kq0, ks0).reshape(shape + (kvh, hd)); v_un = OM.q4_unpack(vq0, vs0).reshape(shape + (kvh, hd))
    q_rot = OM.rope_(q, sin, cos, past, neox=neox); k_rot = OM.rope_(kn, sin, cos, past, neox=neox)
    got = be.n(out1)
    for i in range(b):
        if paged:
            ko = np.concatenate([k_un[table[i, pg]] for pg in range(pages_per_seq)])[:past[i]]
            vo = np.concatenate([v_un[table[i, pg]] for pg in range(pages_per_seq)])[:past[i]]
        else:
            ko, vo = k_un[i, :past[i]], v_un[i, :past[i]]
        want = OM.attention(q_rot[i:i + 1], np.concatenate([ko, k_rot[i]])[None], np.concatenate([vo, vn[i]])[None])[0]
        err = np.abs(got[i].astype(np.float32) - want.astype(np.float32))
        assert np.all(err <= _attn_tol(want) + 2e-3), (what, i, float(err.max()))


def test_decode_utilities(be):
    rng = np.random.default_rng(8)
    table = rng.standard_normal((50, 64)).astype(F16)
    ids = np.array([3, 49, 0], dtype=np.int32)
    out = torch.zeros((3, 64), dtype=torch.float16, device=be.device)
    be.ext.embed_rows(be.t(table), be.t(ids), out)
    assert np.array_equal(be.n(out), table[ids])
    logits = rng.standard_normal((3, 1000)).astype(F16)
    logits[1, 7] = logits[1, 900] = F16(9.0)            # tie -> first index, like torch.argmax
    got = torch.zeros((3,), dtype=torch.int32, device=be.device)
    be.ext.argmax_rows(be.t(logits), got)
    assert np.array_equal(be.n(got), np.argmax(logits.astype(np.float32), axis=-1))
    lens = be.t(np.array([5, 9], dtype=np.int32))
    be.ext.add_i32_(lens, 3)
    assert np.array_equal(be.n(lens), [8, 12])


# ---- FP8 cache codec, page rotation, prefix match (cache_util.hip) -------------------------------------------------------

@pytest.mark.parametrize("offset,width,bsz", [(0, 7, 2), (3, 5, 1), (5, 0, 2), (0, 16, 2)])
def test_fp8_cache_codec(be, offset, width, bsz):
    """fp16_to_fp8 / fp8_to_fp16 over a token range: bit-exact bytes inside the (8-element rounded) range, nothing
    outside it is touched; special values (inf, nan, subnormals, -0) travel as bytes."""
    rng = np.random.default_rng(40 + offset)
    b, s, kvh, hd = 2, 16, 3, 4                         # token size 12: ranges that are not multiples of 8 get rounded
    x = (rng.standard_normal((b, s, kvh, hd)) * 4).astype(F16)
    x.reshape(-1)[:6] = np.array([np.inf, -np.inf, np.nan, -0.0, 6e-8, 65504.0], dtype=F16)
    lo, hi = OM.fp8_range(kvh * hd, offset, width)
    out = torch.full((b, s, kvh, hd), 0xA5, dtype=torch.uint8, device=be.device)
    be.ext.fp16_to_fp8(be.t(x), out, bsz, offset, width)
    got = be.n(out).reshape(b, -1)
    want = np.full((b, s * kvh * hd), 0xA5, dtype=np.uint8)
    want[:bsz, lo:hi] = OM.fp16_to_fp8(x).reshape(b, -1)[:bsz, lo:hi]
    assert np.array_equal(got, want)
    back = torch.full((b, s, kvh, hd), 7.0, dtype=torch.float16, device=be.device)
    be.ext.fp8_to_fp16(out, back, bsz, offset, width)
    wantb = np.full((b, s * kvh * hd), 7.0, dtype=F16)
    wantb[:bsz, lo:hi] = OM.fp8_to_fp16(want[:bsz, lo:hi])
    assert np.array_equal(be.n(back).reshape(b, -1).view(np.uint16), wantb.view(np.uint16))
    # truncation error bound of E5M2: |x - fp8(x)| < 2^-2 |x| for normal values
    xs = x.reshape(b, -1)[:bsz, lo:hi].astype(np.float32)
    ys = wantb[:bsz, lo:hi].astype(np.float32)
    ok = np.isfinite(xs) & (np.abs(xs) > 1e-4)
    with np.errstate(invalid="ignore"):
        err = np.abs(xs - ys)
    assert np.all(err[ok] <= np.abs(xs)[ok] * 0.25)


def test_fp8_cache_codec_errors(be):
    x = torch.zeros((1, 4, 2, 8), dtype=torch.float16, device=be.device)
    o = torch.zeros((1, 4, 2, 8), dtype=torch.uint8, device=be.device)
    with pytest.raises(RuntimeError):
        be.ext.fp16_to_fp8(o, x, 1, 0, 1)                # dtypes swapped
    with pytest.raises(RuntimeError):
        be.ext.fp16_to_fp8(x, o, 1, 3, 2)                # range past the row
    with pytest.raises(RuntimeError):
        be.ext.fp8_to_fp16(o, x[:, :2], 1, 0, 1)         # shape mismatch


@pytest.mark.parametrize("n_pages,page_elems,order", [
    (6, 256 * 2 * 8, [4, 1, 3]), (5, 40, [0, 4]), (9, 5000, [8, 2, 7, 0, 5, 3, 1]), (3, 64, [2]),
    (12, 8 * 1024, [11, 0, 10, 1, 9, 2, 8, 3, 7, 4])])
def test_cache_rotate(be, n_pages, page_elems, order):
    """Cyclic page move of the defragmenter: bit-exact against the sequential restatement, untouched pages unchanged,
    n rotations of an n-cycle are the identity."""
    rng = np.random.default_rng(len(order))
    pages = rng.integers(0, 65536, size=(n_pages, page_elems), dtype=np.uint16)
    cache = be.t(pages.view(np.float16))
    temp = torch.zeros((page_elems,), dtype=torch.float16, device=be.device)
    o = be.t(np.array(order, dtype=np.int32))
    be.ext.cache_rotate(cache, o, temp)
    want = OM.cache_rotate(pages, order)
    assert np.array_equal(be.n(cache).view(np.uint16), want)
    for _ in range(len(order) - 1):
        be.ext.cache_rotate(cache, o, temp)
    assert np.array_equal(be.n(cache).view(np.uint16), pages)
    with pytest.raises(RuntimeError):
        be.ext.cache_rotate(cache, o, temp[:-8])         # "temp tensor incorrect size"


def test_count_match(be):
    a = torch.tensor([[5, 6, 7, 8, 9, 10]], dtype=torch.int64)
    for b_list, max_a, want in (([5, 6, 7, 1, 9], 6, 3), ([5, 6, 7, 8, 9, 10, 11], 6, 6), ([5, 6, 7, 8], 6, 4),
                                ([5, 6, 7, 8, 9, 10], 2, 2), ([1], 6, 0), ([5, 6], 0, 0)):
        b = torch.tensor([b_list], dtype=torch.int64)
        assert be.ext.count_match(a, b, max_a) == want == OM.count_match(a.numpy(), b.numpy(), max_a)
    with pytest.raises(RuntimeError):
        be.ext.count_match(a.int(), a, 3)


def test_matrix_q4_roundtrip(be):
    """matrix_fp16_to_q4 / matrix_q4_to_fp16 (ext_qmatrix.cpp:293-335): the Q4 cache codec over a flat matrix."""
    rng = np.random.default_rng(77)
    w = (rng.standard_normal((48, 64)) * 0.05).astype(F16)          # 3072 elements = 6 codec blocks
    codes = torch.zeros((48, 32), dtype=torch.uint8, device=be.device)
    scales = torch.zeros((48 * 64 // 32,), dtype=torch.float16, device=be.device)
    be.ext.matrix_fp16_to_q4(be.t(w), codes, scales)
    want_codes, want_scales = OM.q4_pack(w.reshape(-1))
    assert np.array_equal(be.n(scales).view(np.uint16), want_scales.view(np.uint16))
    assert np.array_equal(be.n(codes).reshape(-1), want_codes)
    back = torch.zeros((48, 64), dtype=torch.float16, device=be.device)
    be.ext.matrix_q4_to_fp16(codes, scales, back)
    want = OM.q4_unpack(be.n(codes).reshape(-1), be.n(scales))
    assert np.array_equal(be.n(back).reshape(-1).view(np.uint16), want.view(np.uint16))
    assert np.abs(be.n(back).astype(np.float32) - w.astype(np.float32)).max() < 0.03
    with pytest.raises(RuntimeError):
        be.ext.matrix_fp16_to_q4(be.t(w)[:, :60].contiguous(), codes, scales)      # size mismatch


@pytest.mark.parametrize("wbits", [8, 6])
def test_q8_q6_cache_contiguous_roundtrip(be, wbits):
    """Q8 cache (8-bit keys and values) and Q6 cache (8-bit keys, 4-bit values: cache.cu:259-276) against the oracle's
    codec, which is pinned to the reference's cache_q.cuh by execution (tests/test_oracle_ref.py)."""
    from exllamav2_amd.ext import none_tensor
    rng = np.random.default_rng(60 + wbits)
    b, T, kvh, hd = 2, 8, 2, 128
    k = rng.standard_normal((b, T, kvh, hd)).astype(F16)
    v = (rng.standard_normal((b, T, kvh, hd)) * 0.5).astype(F16)
    vbits = 8 if wbits == 8 else 4
    kq = torch.zeros((b, T, kvh, hd), dtype=torch.uint8, device=be.device)
    vq = torch.zeros((b, T, kvh, hd // (8 // vbits)), dtype=torch.uint8, device=be.device)
    ks = torch.zeros((b, T, kvh, hd // 32), dtype=torch.float16, device=be.device)
    vs = torch.zeros_like(ks)
    be.ext.fp16_to_q_kv(be.t(k), kq, ks, be.t(v), vq, vs, b, 2, 4, 0, none_tensor, none_tensor, wbits)
    vpack, vunpack = (OM.q8_pack, OM.q8_unpack) if vbits == 8 else (OM.q4_pack, OM.q4_unpack)
    for i in range(b):
        kc, ksc = OM.q8_pack(k[i, 2:6].reshape(-1))
        vc, vsc = vpack(v[i, 2:6].reshape(-1))
        got_k, got_v = be.n(kq)[i, 2:6].reshape(-1), be.n(vq)[i, 2:6].reshape(-1)
        assert np.array_equal(got_k, kc) and np.array_equal(got_v, vc)
        assert np.array_equal(be.n(ks)[i, 2:6].reshape(-1).view(np.uint16), ksc.view(np.uint16))
        assert np.array_equal(be.n(vs)[i, 2:6].reshape(-1).view(np.uint16), vsc.view(np.uint16))
    assert np.all(be.n(kq)[:, :2] == 0) and np.all(be.n(kq)[:, 6:] == 0)          # outside the token range: untouched
    ko = torch.zeros((b, T, kvh, hd), dtype=torch.float16, device=be.device)
    vo = torch.zeros_like(ko)
    be.ext.q_to_fp16_kv(kq, ko, ks, vq, vo, vs, b, 2, 4, 0, none_tensor, none_tensor, wbits)
    for i in range(b):
        want_k = OM.q8_unpack(be.n(kq)[i, 2:6].reshape(-1), be.n(ks)[i, 2:6].reshape(-1))
        want_v = vunpack(be.n(vq)[i, 2:6].reshape(-1), be.n(vs)[i, 2:6].reshape(-1))
        assert np.array_equal(be.n(ko)[i, 2:6].reshape(-1).view(np.uint16), want_k.view(np.uint16))
        assert np.array_equal(be.n(vo)[i, 2:6].reshape(-1).view(np.uint16), want_v.view(np.uint16))
    errk = np.abs(be.n(ko)[:, 2:6].astype(np.float32) - k[:, 2:6].astype(np.float32))
    assert errk.max() < 0.05                                                      # 8-bit keys: step = absmax / 128


# ---- cache addressing against the EXECUTED reference kernels (tests/golden/make_golden_cache_paged.py) -------------------

def _cache_golden():
    import os, sys
    from tests.conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_cache_paged as G
    return G, np.load(os.path.join(ROOT, "tests", "golden", "reference_cache_addressing.npz"))


def _check_pack(be, got, fx, tag):
    """scales (hence the set of written 512-element blocks) and codes bit for bit; nothing written where the reference
    writes nothing"""
    kq, ks, vq, vs = (be.n(t) for t in got)
    for name, g in (("kq", kq), ("ks", ks), ("vq", vq), ("vs", vs)):
        want = fx[f"{tag}_{name}"]
        if name in ("ks", "vs"):
            assert np.array_equal(g.view(np.uint16), want.view(np.uint16)), (tag, name)
        else:
            written = np.repeat((fx[f"{tag}_{name[0]}s"].view(np.uint16) != 0), g.shape[-1] // fx[f"{tag}_{name[0]}s"].shape[-1], axis=-1)
            assert np.all(g[~written] == 0), (tag, name)
            assert np.array_equal(g[written], want[written]), (tag, name)


def _sha(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_paged_cache_addressing_equals_reference_kernels(be, i):
    G, fx = _cache_golden()
    wbits, pages, kvh, hd, q_len, seqlens, table = G.PAGED[i]
    k, v = G.kv_inputs(f"paged{i}", (pages, G.PS, kvh, hd))
    kb, vb = G.code_bytes(wbits, hd)
    kq = torch.zeros((pages, G.PS, kvh, kb), dtype=torch.uint8, device=be.device)
    vq = torch.zeros((pages, G.PS, kvh, vb), dtype=torch.uint8, device=be.device)
    ks = torch.zeros((pages, G.PS, kvh, hd // 32), dtype=torch.float16, device=be.device)
    vs = torch.zeros_like(ks)
    sl, bt = np.array(seqlens, np.int32), np.array(table, np.int32)
    be.ext.fp16_to_q_kv(be.t(k), kq, ks, be.t(v), vq, vs, len(seqlens), 0, q_len, G.PS, be.t(sl), be.t(bt), wbits)
    _check_pack(be, (kq, ks, vq, vs), fx, f"paged{i}")
    qk, qks, qv, qvs = G.q_inputs(f"paged{i}", wbits, (pages, G.PS), kvh, hd)
    ko = torch.zeros((pages, G.PS, kvh, hd), dtype=torch.float16, device=be.device)
    vo = torch.zeros_like(ko)
    be.ext.q_to_fp16_kv(be.t(qk), ko, be.t(qks), be.t(qv), vo, be.t(qvs), len(seqlens), 0, 0, G.PS,
                        be.t((sl + q_len).astype(np.int32)), be.t(bt), wbits)
    assert np.array_equal(_sha(be.n(ko)), fx[f"paged{i}_ko_sha"]) and np.array_equal(_sha(be.n(vo)), fx[f"paged{i}_vo_sha"])


@pytest.mark.parametrize("i", [0, 1, 2, 3])
def test_contiguous_cache_addressing_equals_reference_kernels(be, i):
    from exllamav2_amd.ext import none_tensor
    G, fx = _cache_golden()
    wbits, b, T, kvh, hd, offset, width = G.CONTIG[i]
    k, v = G.kv_inputs(f"contig{i}", (b, T, kvh, hd))
    kb, vb = G.code_bytes(wbits, hd)
    kq = torch.zeros((b, T, kvh, kb), dtype=torch.uint8, device=be.device)
    vq = torch.zeros((b, T, kvh, vb), dtype=torch.uint8, device=be.device)
    ks = torch.zeros((b, T, kvh, hd // 32), dtype=torch.float16, device=be.device)
    vs = torch.zeros_like(ks)
    be.ext.fp16_to_q_kv(be.t(k), kq, ks, be.t(v), vq, vs, b, offset, width, 0, none_tensor, none_tensor, wbits)
    _check_pack(be, (kq, ks, vq, vs), fx, f"contig{i}")
    qk, qks, qv, qvs = G.q_inputs(f"contig{i}", wbits, (b, T), kvh, hd)
    ko = torch.zeros((b, T, kvh, hd), dtype=torch.float16, device=be.device)
    vo = torch.zeros_like(ko)
    be.ext.q_to_fp16_kv(be.t(qk), ko, be.t(qks), be.t(qv), vo, be.t(qvs), b, offset, width, 0, none_tensor, none_tensor, wbits)
    assert np.array_equal(_sha(be.n(ko)), fx[f"contig{i}_ko_sha"]) and np.array_equal(_sha(be.n(vo)), fx[f"contig{i}_vo_sha"])
