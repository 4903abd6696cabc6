"""q_matrix: re-layout + reconstruct (bit-exact) and gemm_half_q_half (fp16 tolerance) against the oracle.

Parity method = the reference's own (tests/test_gemv.py:136-165): kernel vs reconstruct+matmul on identity and randn
inputs, for every bit width, mixed sections, partial super-chunks, act-order on/off, GPTQ with and without g_idx.
"""
import os

import numpy as np
import pytest
import torch

from oracle import exl2 as OX
from tests.util import make_exl2, gptq_to_torch, half_tol

SPECS = {
    "b4_g128": (256, 32, [(4, 128, 256)]),
    "b4_tail": (160, 32, [(4, 32, 160)]),
    "b2": (192, 32, [(2, 64, 192)]),
    "b3": (192, 32, [(3, 64, 192)]),
    "b5": (192, 32, [(5, 32, 192)]),
    "b6": (192, 32, [(6, 32, 192)]),
    "b8": (192, 32, [(8, 32, 192)]),
    "mixed_all": (800, 64, [(8, 32, 32), (6, 32, 96), (5, 64, 128), (4, 128, 256), (3, 64, 160), (2, 64, 128)]),
    "mixed_5_4": (1024, 64, [(5, 128, 128), (4, 128, 896)]),
    "long_runs": (2304, 32, [(4, 128, 2304)]),            # > QDESC_MAX_SUPER super-chunks: several descriptors
}


@pytest.mark.parametrize("name", list(SPECS))
@pytest.mark.parametrize("act_order", [True, False])
def test_reconstruct_bit_exact(be, name, act_order):
    k, n, spec = SPECS[name]
    t, ref, w, h = make_exl2(be, k, n, spec, seed=1, act_order=act_order)
    out = torch.zeros((k, n), dtype=torch.float16, device=be.device)
    be.ext.reconstruct(h, out)
    got = be.n(out)
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16))
    info = be.ext.q_matrix_info(h)
    assert (info["height"], info["width"], info["is_gptq"]) == (k, n, False)
    be.ext.free_q_matrix(h)


@pytest.mark.parametrize("name", ["b4_g128", "b4_tail", "b3", "b6", "mixed_all", "long_runs"])
@pytest.mark.parametrize("m", [1, 2, 5, 16])
def test_gemm_vs_oracle(be, name, m):
    k, n, spec = SPECS[name]
    t, ref, w, h = make_exl2(be, k, n, spec, seed=2, act_order=True)
    rng = np.random.default_rng(3)
    a = rng.standard_normal((m, k)).astype(np.float16)
    c = torch.zeros((m, n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(be.t(a), h, c)
    want = OX.gemm_ref(a, ref, exact=True)
    got = be.n(c).astype(np.float64)
    assert np.all(np.abs(got - want) <= half_tol(want, k)), np.abs(got - want).max()
    be.ext.free_q_matrix(h)


def test_gemm_identity_equals_reconstruct(be):
    """tests/test_gemv.py:155-159: gemm(I) must reproduce reconstruct() exactly (each output is one weight)."""
    k, n, spec = SPECS["mixed_all"]
    t, ref, w, h = make_exl2(be, k, n, spec, seed=4, act_order=True)
    eye = np.eye(k, dtype=np.float16)
    got = np.zeros((k, n), dtype=np.float16)
    for r0 in range(0, k, 16):
        a = be.t(eye[r0:r0 + 16])
        c = torch.zeros((a.shape[0], n), dtype=torch.float16, device=be.device)
        be.ext.gemm_half_q_half(a, h, c)
        got[r0:r0 + 16] = be.n(c)
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16))
    be.ext.free_q_matrix(h)


def test_gemm_more_than_16_rows(be):
    k, n, spec = SPECS["mixed_5_4"]
    t, ref, w, h = make_exl2(be, k, n, spec, seed=5)
    rng = np.random.default_rng(6)
    m = 37
    a = rng.standard_normal((m, k)).astype(np.float16)
    c = torch.zeros((m, n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(be.t(a), h, c)
    want = OX.gemm_ref(a, ref, exact=True)
    assert np.all(np.abs(be.n(c).astype(np.float64) - want) <= half_tol(want, k))
    be.ext.free_q_matrix(h)


@pytest.mark.parametrize("name", ["b4_g128", "b4_tail", "b2", "b3", "b5", "b6", "b8", "mixed_all"])
def test_prefill_identity_equals_reconstruct(be, name):
    """M > 16 takes the dequantize-into-MFMA kernel (qgemm_prefill.hip): the B fragments must be reconstruct()'s weights
    bit for bit, so gemm(I) == reconstruct() exactly (tests/test_gemv.py:155-159 relation), one call, M = K rows."""
    k, n, spec = SPECS[name]
    t, ref, w, h = make_exl2(be, k, n, spec, seed=14, act_order=True)
    c = torch.zeros((k, n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(be.t(np.eye(k, dtype=np.float16)), h, c)
    assert np.array_equal(be.n(c).view(np.uint16), ref.view(np.uint16))
    be.ext.free_q_matrix(h)


@pytest.mark.parametrize("m", [17, 300])
def test_prefill_vs_oracle(be, m):
    k, n, spec = SPECS["mixed_5_4"]
    t, ref, w, h = make_exl2(be, k, n, spec, seed=15, bias=True)
    rng = np.random.default_rng(16)
    a = rng.standard_normal((m, k)).astype(np.float16)
    c = torch.zeros((m, n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(be.t(a), h, c)
    want = OX.gemm_ref(a, ref, bias=t["bias"], exact=True)
    assert np.all(np.abs(be.n(c).astype(np.float64) - want) <= half_tol(want, k))
    be.ext.free_q_matrix(h)


@pytest.mark.parametrize("name", ["mixed_5_4", "mixed_all", "b4_g128"])
@pytest.mark.parametrize("m", [17, 64, 128])
def test_skinny_kernel_split_k(be, name, m, monkeypatch):
    """Round 6: 17-128 rows run on qgemm_skinny_kernel (weight-stream bound: 128 columns x all rows x a slice of K per workgroup).
    EXL2_SKINNY_SPLITK forces the number of K splits here (the host picks it from the grid otherwise): partial tiles leave as
    agent-scope stores, the last split to arrive adds them in split order.  Same bar against the oracle for every split count; two
    runs are bit-identical (fixed order); identity rows return reconstruct() bit for bit; the ticket reset (a second call on the
    same scratch) works."""
    k, n, spec = SPECS[name]
    t, ref, w, h = make_exl2(be, k, n, spec, seed=21, bias=True)
    rng = np.random.default_rng(22)
    a = rng.standard_normal((m, k)).astype(np.float16)
    want = OX.gemm_ref(a, ref, bias=t["bias"], exact=True)
    outs = {}
    for ks in ("1", "2", "3", "5"):
        monkeypatch.setenv("EXL2_SKINNY_SPLITK", ks)
        for rep in range(2):
            c = torch.full((m, n), 7.0, dtype=torch.float16, device=be.device)
            be.ext.gemm_half_q_half(be.t(a), h, c)
            got = be.n(c).copy()
            assert np.all(np.abs(got.astype(np.float64) - want) <= half_tol(want, k)), (ks, rep)
            if rep: assert np.array_equal(got, outs[ks])
            outs[ks] = got
    # a matrix with more bit-width sections than the kernel arguments hold walks its descriptors in device memory instead: forced here
    monkeypatch.setenv("EXL2_SKINNY_FORCE_DESC", "1")
    c = torch.full((m, n), 7.0, dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(be.t(a), h, c)
    assert np.array_equal(be.n(c), outs["5"])
    monkeypatch.delenv("EXL2_SKINNY_FORCE_DESC")
    # identity rows: every output has ONE non-zero term, whatever the split
    eye = np.zeros((m, k), dtype=np.float16)
    rows = rng.choice(k, size=m, replace=False)
    eye[np.arange(m), rows] = 1.0
    monkeypatch.setenv("EXL2_SKINNY_SPLITK", "3")
    c = torch.zeros((m, n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(be.t(eye), h, c)
    bias = t["bias"].astype(np.float32) if t.get("bias") is not None else 0.0
    assert np.array_equal(be.n(c), (ref[rows].astype(np.float32) + bias).astype(np.float16))
    be.ext.free_q_matrix(h)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("EXL2_TEST_SEEDS", "6")))))        # (more seeds: a longer hunt, by hand)
def test_gemm_random_rows_shapes_and_bit_mixes(be, seed, monkeypatch):
    """Seeded random (rows, K, N, bit-width sections, group sizes, act-order, bias) through exl2_gemm_half_q_half: every row-count
    route -- the decode kernels (<= 16 rows), the 17-128-row prefill kernel, the 256-column MFMA kernel with the weights decoded
    inside the GEMM or once per call (forced at random: its natural threshold is beyond what the emulator runs in seconds) --
    against the oracle's reconstruct-then-matmul; one-hot rows must return rows of reconstruct() bit for bit."""
    rng = np.random.default_rng(13000 + seed)
    k = 32 * int(rng.integers(4, 49))
    n = 32 * int(rng.choice([1, 2, 3, 8, 9]))
    widths = sorted(rng.choice([8, 6, 5, 4, 3, 2], size=int(rng.integers(1, 4)), replace=False).tolist(), reverse=True)
    cuts = sorted(rng.choice(np.arange(1, k // 32), size=len(widths) - 1, replace=False).tolist()) if len(widths) > 1 else []
    edges = [0] + [32 * c for c in cuts] + [k]
    spec = [(int(b), int(rng.choice([32, 64, 128])), edges[i + 1] - edges[i]) for i, b in enumerate(widths)]
    m = int(rng.choice([1, 3, 4, 5, 11, 16, 17, 40, 128, 129, 200, 300, 520]))
    if m >= 129 and rng.integers(0, 2):
        monkeypatch.setenv("EXL2_PREFILL_MT", str(int(rng.choice([4, 8]))))
        monkeypatch.setenv("EXL2_PREFILL_WPRE_MIN_ROWS", str(int(rng.integers(0, 2))))
    bias = bool(rng.integers(0, 2))
    t, ref, w, h = make_exl2(be, k, n, spec, seed=400 + seed, act_order=bool(rng.integers(0, 2)), bias=bias)
    a = rng.standard_normal((m, k)).astype(np.float16)
    hot = rng.choice(m, size=min(m, 8), replace=False)
    hot_k = rng.integers(0, k, size=len(hot))
    if not bias:
        a[hot] = 0
        a[hot, hot_k] = 1.0
    c = torch.zeros((m, n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(be.t(a), h, c)
    got = be.n(c)
    what = (m, k, n, spec, bias)
    if not bias:
        assert np.array_equal(got[hot].view(np.uint16), ref[hot_k].view(np.uint16)), what
    want = OX.gemm_ref(a, ref, bias=t["bias"] if bias else None, exact=True)
    assert np.all(np.abs(got.astype(np.float64) - want) <= half_tol(want, k)), what
    be.ext.free_q_matrix(h)


@pytest.mark.parametrize("mt", ["8", "4"])
@pytest.mark.parametrize("wpre", ["0", "1"])
@pytest.mark.parametrize("m", [300, 1024])
def test_prefill_tile256_variant_forced(be, m, wpre, mt, monkeypatch):
    """The 256 x 256 instantiation of the dequantize-into-MFMA kernel (qgemm_mfma_kernel<., 8, .>), forced at shapes both backends
    can run, with the weights decoded inside the GEMM (wpre 0) and decoded once per call into fragment images by wfrag_kernel
    (wpre 1: the route of >= 2048 rows): one-hot rows return rows of reconstruct() bit for bit, random rows land within the fp16
    bar of the oracle (the reference's relation, tests/test_gemv.py:155-165)."""
    monkeypatch.setenv("EXL2_PREFILL_MT", mt)                    # (256-row and 128-row tile of the same kernel)
    monkeypatch.setenv("EXL2_PREFILL_WPRE_MIN_ROWS", wpre)
    k, n, spec = SPECS["mixed_5_4"]
    t, ref, w, h = make_exl2(be, k, n, spec, seed=21, bias=False)
    rng = np.random.default_rng(22)
    a = rng.standard_normal((m, k)).astype(np.float16)
    hot = rng.choice(m, size=64, replace=False)
    hot_k = rng.integers(0, k, size=64)
    a[hot] = 0
    a[hot, hot_k] = 1.0
    c = torch.zeros((m, n), dtype=torch.float16, device=be.device)
    calls = be.ext.prefill_route_info()[3]
    be.ext.gemm_half_q_half(be.t(a), h, c)
    got = be.n(c)
    assert be.ext.prefill_route_info() == (m, 32 * int(mt), wpre == "1", calls + 1)      # the variant that was forced is the one that ran
    assert np.array_equal(got[hot].view(np.uint16), ref[hot_k].view(np.uint16))
    want = OX.gemm_ref(a, ref, exact=True)
    assert np.all(np.abs(got.astype(np.float64) - want) <= half_tol(want, k))
    be.ext.free_q_matrix(h)


def test_prefill_predecode_scratch_cap_falls_back(be, monkeypatch):
    """The weights-decoded-once route needs K * N * 2 bytes of scratch; above the cap (or when the allocation fails) the call must
    run with the decode inside the GEMM instead of failing -- same result bit for bit (same decoders, same rounding)."""
    k, n, spec = SPECS["mixed_5_4"]
    t, ref, w, h = make_exl2(be, k, n, spec, seed=23, bias=False)
    a = np.random.default_rng(24).standard_normal((300, k)).astype(np.float16)
    outs = []
    for cap in (None, "1"):
        monkeypatch.setenv("EXL2_PREFILL_WPRE_MIN_ROWS", "1")
        if cap: monkeypatch.setenv("EXL2_PREFILL_WPRE_MAX_BYTES", cap)
        c = torch.zeros((300, n), dtype=torch.float16, device=be.device)
        be.ext.gemm_half_q_half(be.t(a), h, c)
        outs.append(be.n(c).copy())
        assert be.ext.prefill_route_info()[2] == (cap is None)
    assert np.array_equal(outs[0].view(np.uint16), outs[1].view(np.uint16))
    be.ext.free_q_matrix(h)


@pytest.mark.gpu
@pytest.mark.parametrize("role", ["q_proj", "gate_proj", "down_proj"])
@pytest.mark.parametrize("force", [None, "8", "in-kernel decode"])
def test_prefill_llama2_7b_shapes_4096_rows(role, force, monkeypatch):
    """BASELINE configs[2]'s kernel at the three Llama-2-7B linear shapes, M = 4096 rows (the natural selection there is the
    256 x 256 tile -- the variant the 8 x 2048 prefill runs -- and it is also forced): 256 sampled one-hot rows == reconstruct()
    bit for bit, 64 random rows within the fp16 bar of the float64 product with the oracle's reconstruct()."""
    from tests.conftest import Backend
    from exllamav2_amd.synth import RECIPES, synth_linear
    be = Backend("hip")
    if force == "in-kernel decode": monkeypatch.setenv("EXL2_PREFILL_WPRE_MIN_ROWS", "0")      # (default at 4096 rows: wfrag_kernel)
    elif force: monkeypatch.setenv("EXL2_PREFILL_MT", force)
    monkeypatch.setenv("EXL2_PREFILL_TRACE", "1")
    k, n = {"q_proj": (4096, 4096), "gate_proj": (4096, 11008), "down_proj": (11008, 4096)}[role]
    gen = torch.Generator(); gen.manual_seed(321)
    w = synth_linear(k, n, RECIPES["4.0bpw"][role], "cpu", gen, sigma=0.02, act_order=True)
    ref = OX.exl2_reconstruct({kk: vv.numpy().copy() for kk, vv in w.items() if kk != "q_perm"})
    wd = {kk: vv.to(be.device) for kk, vv in w.items()}      # kept alive: the handle re-lays q_weight out in place and reads it
    h = be.ext.make_q_matrix_from_dict(wd, None)
    m = 4096
    rng = np.random.default_rng(7)
    a = rng.standard_normal((m, k)).astype(np.float16)
    hot = rng.choice(m, size=256, replace=False)
    hot_k = rng.integers(0, k, size=256)
    a[hot] = 0
    a[hot, hot_k] = 1.0
    c = torch.zeros((m, n), dtype=torch.float16, device=be.device)
    calls = be.ext.prefill_route_info()[3]
    be.ext.gemm_half_q_half(be.t(a), h, c)
    got = be.n(c)
    # the variant: 256-row tile at 4096 rows (natural selection and forced alike), weights decoded once per call unless forced
    # into the GEMM
    assert be.ext.prefill_route_info() == (m, 256, force != "in-kernel decode", calls + 1), be.ext.prefill_route_info()
    assert np.array_equal(got[hot].view(np.uint16), ref[hot_k].view(np.uint16))
    rows = np.setdiff1d(np.arange(m), hot)[:64]
    want = a[rows].astype(np.float64) @ ref.astype(np.float64)
    w_rms = float(np.sqrt(np.mean(ref.astype(np.float32) ** 2)))
    tol = np.abs(want) * 2.0 ** -10 + 1.5e-3 + 6.0 * 2.0 ** -11 * np.sqrt(k) * w_rms      # (test_full_size_linear's bar)
    err = np.abs(got[rows].astype(np.float64) - want)
    assert np.all(err <= tol), float(err.max())
    be.ext.free_q_matrix(h)


def test_prefill_gptq_identity(be):
    k, n, gs = 384, 48, 128
    t = OX.synth_gptq(k, n, gs, seed=17, act_order=True)
    ref = OX.gptq_reconstruct(t)
    w = gptq_to_torch(be, t)                      # the handle keeps raw pointers: the tensors must outlive it
    h = be.ext.make_q_matrix_from_dict(w, None)
    c = torch.zeros((k, n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(be.t(np.eye(k, dtype=np.float16)), h, c)
    assert np.array_equal(be.n(c).view(np.uint16), ref.view(np.uint16))
    be.ext.free_q_matrix(h)


@pytest.mark.parametrize("m", [3, 16])
def test_gemm_many_rows_staging_route(be, m, monkeypatch):
    """When M x K does not fit twice in LDS the streaming kernel gathers the rows from global memory instead of staging
    them through LDS; forced here on a small shape."""
    monkeypatch.setenv("EXL2_GEMV_NO_LDS_STAGE", "1")
    k, n, spec = SPECS["mixed_5_4"]
    t, ref, w, h = make_exl2(be, k, n, spec, seed=21, bias=True)
    a = np.random.default_rng(22).standard_normal((m, k)).astype(np.float16)
    c = torch.zeros((m, n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(be.t(a), h, c)
    want = OX.gemm_ref(a, ref, bias=t["bias"], exact=True)
    assert np.all(np.abs(be.n(c).astype(np.float64) - want) <= half_tol(want, k))
    be.ext.free_q_matrix(h)


@pytest.mark.parametrize("name,m,items", [("mixed_all", 5, 1), ("mixed_all", 16, 2), ("mixed_5_4", 16, 2), ("long_runs", 9, 5),
                                          ("b4_tail", 16, 1), ("long_runs", 16, 64)])
def test_gemm_phased_route(be, name, m, items, monkeypatch):
    """Batch-of-sequences decode (M = 5..16 rows at large K): row pre-pass into the packed K order + the phased kernel
    (qgemv_stream.hip), forced here on small shapes with slabs of `items` super-chunks (several phases per section,
    partial super-chunks as phases of their own)."""
    monkeypatch.setenv("EXL2_GEMV_PHASED", "1")
    monkeypatch.setenv("EXL2_GEMV_PHASE_ITEMS", str(items))
    k, n, spec = SPECS[name]
    t, ref, w, h = make_exl2(be, k, n, spec, seed=41, bias=True)
    a = np.random.default_rng(42).standard_normal((m, k)).astype(np.float16)
    c = torch.zeros((m, n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(be.t(a), h, c)
    want = OX.gemm_ref(a, ref, bias=t["bias"], exact=True)
    assert np.all(np.abs(be.n(c).astype(np.float64) - want) <= half_tol(want, k))
    # identity rows through the phased route reproduce reconstruct() bit for bit
    if k <= 1024:
        eye = np.eye(k, dtype=np.float16)
        t2, ref2, w2, h2 = make_exl2(be, k, n, spec, seed=41, bias=False)
        for r0 in range(0, k, 16):
            c2 = torch.zeros((16, n), dtype=torch.float16, device=be.device)
            be.ext.gemm_half_q_half(be.t(eye[r0:r0 + 16]), h2, c2)
            assert np.array_equal(be.n(c2).view(np.uint16), ref2[r0:r0 + 16].view(np.uint16))
        be.ext.free_q_matrix(h2)
    be.ext.free_q_matrix(h)


def test_gemm_phased_route_gptq(be, monkeypatch):
    monkeypatch.setenv("EXL2_GEMV_PHASED", "1")
    monkeypatch.setenv("EXL2_GEMV_PHASE_ITEMS", "1")
    k, n, gs = 384, 48, 128
    t = OX.synth_gptq(k, n, gs, seed=43, act_order=True)
    ref = OX.gptq_reconstruct(t)
    w = gptq_to_torch(be, t)
    h = be.ext.make_q_matrix_from_dict(w, None)
    a = np.random.default_rng(44).standard_normal((16, k)).astype(np.float16)
    c = torch.zeros((16, n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(be.t(a), h, c)
    want = OX.gemm_ref(a, ref, exact=True)
    assert np.all(np.abs(be.n(c).astype(np.float64) - want) <= half_tol(want, k))
    be.ext.free_q_matrix(h)


def test_make_q_matrix_split_column_shards(be):
    """Tensor-parallel sharding along the output features (tensor_p.py -> make_q_matrix_split, ext_qmatrix.cpp:113-187):
    handles over column slices reproduce the matching columns of the full product, and reconstruct() of a shard is the
    slice of the full reconstruct() bit for bit."""
    k, n, spec = SPECS["mixed_5_4"]
    t = OX.synth_exl2(k, n, spec, seed=31, act_order=True, bias=True)
    ref = OX.exl2_reconstruct(t)
    a = np.random.default_rng(32).standard_normal((3, k)).astype(np.float16)
    want = OX.gemm_ref(a, ref, bias=t["bias"], exact=True)
    from tests.util import exl2_to_torch
    keep = []
    for (c0, c1) in ((0, 32), (32, 64)):
        ts = dict(t)
        ts["q_weight"] = np.ascontiguousarray(t["q_weight"][:, c0:c1])
        ts["q_scale"] = np.ascontiguousarray(t["q_scale"][:, c0 // 8:c1 // 8])
        ts["bias"] = np.ascontiguousarray(t["bias"][c0:c1])
        w = exl2_to_torch(be, ts); keep.append(w)
        h = be.ext.make_q_matrix_from_dict(w, None, split=True)
        out = torch.zeros((k, c1 - c0), dtype=torch.float16, device=be.device)
        be.ext.reconstruct(h, out)
        assert np.array_equal(be.n(out).view(np.uint16), ref[:, c0:c1].view(np.uint16))
        c = torch.zeros((3, c1 - c0), dtype=torch.float16, device=be.device)
        be.ext.gemm_half_q_half(be.t(a), h, c)
        assert np.all(np.abs(be.n(c).astype(np.float64) - want[:, c0:c1]) <= half_tol(want[:, c0:c1], k))
        be.ext.free_q_matrix(h)


def test_gemm_bias(be):
    k, n, spec = SPECS["b4_g128"]
    t, ref, w, h = make_exl2(be, k, n, spec, seed=7, bias=True)
    rng = np.random.default_rng(8)
    a = rng.standard_normal((3, k)).astype(np.float16)
    c = torch.zeros((3, n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(be.t(a), h, c)
    want = OX.gemm_ref(a, ref, bias=t["bias"], exact=True)
    assert np.all(np.abs(be.n(c).astype(np.float64) - want) <= half_tol(want, k))
    be.ext.free_q_matrix(h)


@pytest.mark.parametrize("k,n,gs,act", [(256, 32, 128, False), (256, 32, 64, True), (384, 48, 128, True), (160, 16, 32, False)])
def test_gptq(be, k, n, gs, act):
    t = OX.synth_gptq(k, n, gs, seed=9, act_order=act)
    ref = OX.gptq_reconstruct(t)
    w = gptq_to_torch(be, t)
    h = be.ext.make_q_matrix_from_dict(w, None)
    out = torch.zeros((k, n), dtype=torch.float16, device=be.device)
    be.ext.reconstruct(h, out)
    assert np.array_equal(be.n(out).view(np.uint16), ref.view(np.uint16))
    if act:
        xm, xi = OX.gptq_sequential_perm(t["g_idx"], k // gs)
        assert np.array_equal(be.n(w["q_perm"]).astype(np.int64) & 0xFFFF, xm)
        assert np.array_equal(be.n(w["q_invperm"]).astype(np.int64) & 0xFFFF, xi)
    rng = np.random.default_rng(10)
    a = rng.standard_normal((4, k)).astype(np.float16)
    c = torch.zeros((4, n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(be.t(a), h, c)
    want = OX.gemm_ref(a, ref, exact=True)
    assert np.all(np.abs(be.n(c).astype(np.float64) - want) <= half_tol(want, k))
    assert be.ext.q_matrix_info(h)["is_gptq"]
    be.ext.free_q_matrix(h)


def test_make_group_map_matches_oracle(be):
    t = OX.synth_exl2(800, 64, SPECS["mixed_all"][2], seed=11)
    got = be.ext.make_group_map(torch.from_numpy(t["q_groups"]), t["q_weight"].shape[0]).numpy()
    assert np.array_equal(got, OX.make_group_map(t["q_groups"], t["q_weight"].shape[0]))


def test_errors_are_loud(be):
    """Argument errors surface as RuntimeError (TORCH_CHECK convention, cpp/util.h:33-38)."""
    t = OX.synth_exl2(256, 32, [(4, 128, 256)], seed=12)
    from tests.util import exl2_to_torch
    w = exl2_to_torch(be, t)
    h = be.ext.make_q_matrix_from_dict(w, None)
    with pytest.raises(RuntimeError):
        be.ext.gemm_half_q_half(torch.zeros((1, 128), dtype=torch.float16, device=be.device), h,
                                torch.zeros((1, 32), dtype=torch.float16, device=be.device))
    with pytest.raises(RuntimeError):
        be.ext.reconstruct(h, torch.zeros((256, 32), dtype=torch.float32, device=be.device))
    be.ext.free_q_matrix(h)
    bad = exl2_to_torch(be, OX.synth_exl2(256, 32, [(4, 128, 256)], seed=13))
    bad["q_groups"][0] = 7      # unsupported bit width
    with pytest.raises(RuntimeError):
        be.ext.make_q_matrix_from_dict(bad, None)


# ---- BASELINE.json full sizes: the Llama-2-7B linears with the reference's 4.0 bpw bit mixes ---------------------------

FULL_SIZE = {   # role -> (K, N, emulation too?)   (the emulation runs the 4096 x 4096 and 11008 x 4096 shapes in seconds)
    "q_proj": (4096, 4096, True), "down_proj": (11008, 4096, True), "gate_proj": (4096, 11008, False),
    "lm_head": (4096, 32000, False),
}


@pytest.mark.parametrize("role", list(FULL_SIZE))
def test_full_size_linear(be, role):
    """Size-independent properties at the real shapes: (i) one-hot rows through q_gemm return rows of the matrix, bit
    for bit the oracle's reconstruct() (a product by 1.0 and a single fp16 rounding: the gemm(I) == reconstruct()
    relation on a handful of rows); (ii) the device's reconstruct() of those rows' columns agrees; (iii) a random
    activation row lands within the fp16 tolerance of the float64 product, where the tolerance carries the K-dependent
    term of the reference's fp16-rounded weights (2^-11 relative per weight, random walk over K)."""
    from exllamav2_amd.synth import RECIPES, synth_linear
    k, n, on_emu = FULL_SIZE[role]
    if be.is_emu and not on_emu:
        pytest.skip("full-size shape runs on the hip backend only")
    gen = torch.Generator(); gen.manual_seed(123)
    w = synth_linear(k, n, RECIPES["4.0bpw"][role], "cpu", gen, sigma=0.02, act_order=True)
    t = {kk: vv.numpy().copy() for kk, vv in w.items() if kk != "q_perm"}
    ref = OX.exl2_reconstruct(t)                                         # fp16 [K, N], original row order
    wd = {kk: vv.to(be.device) for kk, vv in w.items()}
    h = be.ext.make_q_matrix_from_dict(wd, None)
    rows = [0, 1, k // 2 + 3, k - 1]
    a = np.zeros((len(rows), k), dtype=np.float16)
    for i, r in enumerate(rows): a[i, r] = 1.0
    c = torch.zeros((len(rows), n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(be.t(a), h, c)
    assert np.array_equal(be.n(c).view(np.uint16), ref[rows].view(np.uint16))
    if not be.is_emu:
        out = torch.zeros((k, n), dtype=torch.float16, device=be.device)
        be.ext.reconstruct(h, out)
        assert np.array_equal(be.n(out).view(np.uint16), ref.view(np.uint16))
    x = np.random.default_rng(5).standard_normal((1, k)).astype(np.float16)
    y = torch.zeros((1, n), dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(be.t(x), h, y)
    want = x.astype(np.float64) @ ref.astype(np.float64)
    w_rms = float(np.sqrt(np.mean(ref.astype(np.float32) ** 2)))
    tol = np.abs(want) * 2.0 ** -10 + 1.5e-3 + 6.0 * 2.0 ** -11 * np.sqrt(k) * w_rms
    err = np.abs(be.n(y).astype(np.float64) - want)
    assert np.all(err <= tol), float(err.max())
    # the yardstick at this size: the reference's own GEMV kernel, executed on the host for these very tensors
    # (tests/golden/make_golden_qgemm.py), is 2e-2 .. 5e-2 away from matmul(x, reconstruct()); ours must not be farther
    import os
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_q_gemm.npz"))
    if f"c_full_{role}_64" in fx.files:
        e_ref = min(np.abs(fx[f"c_full_{role}_{bk}"].astype(np.float64) - want).max() for bk in (32, 64))
        assert err.max() <= e_ref, (role, float(err.max()), float(e_ref))
    be.ext.free_q_matrix(h)


def test_empty_and_ragged_inputs(be):
    """Edge cases of the boundary: zero rows is a no-op (the reference's grid of height 0 launches nothing), every row
    count 1..17 through the skinny / phased / staged routes agrees with the oracle row by row (a ragged batch is just a
    different M), and shapes that do not match the handle are refused loudly."""
    k, n, spec = SPECS["mixed_5_4"]
    t, ref, w, h = make_exl2(be, k, n, spec, seed=21)
    c0 = torch.full((0, n), 7.0, dtype=torch.float16, device=be.device)
    be.ext.gemm_half_q_half(torch.zeros((0, k), dtype=torch.float16, device=be.device), h, c0)      # no-op, no error
    rng = np.random.default_rng(22)
    a = rng.standard_normal((17, k)).astype(np.float16)
    want = OX.gemm_ref(a, ref, exact=True)
    for m in (1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17):
        c = torch.zeros((m, n), dtype=torch.float16, device=be.device)
        be.ext.gemm_half_q_half(be.t(a[:m]), h, c)
        assert np.all(np.abs(be.n(c).astype(np.float64) - want[:m]) <= half_tol(want[:m], k)), m
    with pytest.raises(RuntimeError):
        be.ext.gemm_half_q_half(be.t(a[:2, :k - 32]), h, torch.zeros((2, n), dtype=torch.float16, device=be.device))
    with pytest.raises(RuntimeError):
        be.ext.gemm_half_q_half(be.t(a[:2]), h, torch.zeros((3, n), dtype=torch.float16, device=be.device))
    be.ext.free_q_matrix(h)
