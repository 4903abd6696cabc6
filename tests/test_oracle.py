"""The oracle itself: pinned against fixtures produced by RUNNING the reference's torch/python paths
(tests/golden/make_golden.py -> reference_torch_paths.npz) and by the relations the reference's own tests print
(tests/test_gemv.py:155-165).  No GPU, no HIP library."""
import os

import numpy as np
import pytest

from oracle import exl2 as OX
from oracle import modules as OM

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_torch_paths.npz"))


def test_group_map_matches_reference_python():
    got = OX.make_group_map(G["gm_q_groups"], int(G["gm_num_qrows"]))
    assert np.array_equal(got, G["gm_map"])


def test_rms_norm_matches_reference_forward_torch():
    # rmsnorm.py:187-211 rounds x * rsqrt(var) to fp16 BEFORE multiplying by w; the kernel (rms_norm.cu) rounds once.
    got = OM.rms_norm(G["rms_x"], G["rms_w"], 1e-5).astype(np.float32)
    want = G["rms_y"].astype(np.float32)
    assert np.all(np.abs(got - want) <= np.abs(want) * 2.0 ** -9 + 1e-3)


def test_attention_matches_reference_attn_torch():
    got = OM.attention(G["att_q"], G["att_k"], G["att_v"]).astype(np.float32)
    want = G["att_y"].astype(np.float32)
    assert np.all(np.abs(got - want) <= 2e-3 + np.abs(want) * 2.0 ** -9)


@pytest.mark.parametrize("tag,neox", [("neox", True), ("gptj", False)])
def test_rope_tables_match_reference_device_context(tag, neox):
    sin, cos = OM.rope_tables(48, 64, 10000.0, neox)
    # fp32 sin/cos of fp32 angles rounded to fp16: numpy and torch may differ in the last fp32 ulp before rounding
    assert np.mean(sin.view(np.uint16) == G[f"sin_{tag}"].view(np.uint16)) > 0.995
    assert np.max(np.abs(sin.astype(np.float32) - G[f"sin_{tag}"].astype(np.float32))) <= 2.0 ** -10
    assert np.max(np.abs(cos.astype(np.float32) - G[f"cos_{tag}"].astype(np.float32))) <= 2.0 ** -10


def test_silu_mul_matches_reference_mlp_torch():
    got = OM.silu_mul(G["act_g"], G["act_u"])
    same = np.mean(got.view(np.uint16) == G["act_y"].view(np.uint16))
    assert same > 0.995 and np.max(np.abs(got.astype(np.float32) - G["act_y"].astype(np.float32))) < 4e-3


@pytest.mark.parametrize("bits", [2, 3, 4, 5, 6, 8])
def test_pack_unpack_round_trip(bits):
    rng = np.random.default_rng(bits)
    codes = rng.integers(0, 1 << bits, size=(96, 40)).astype(np.uint16)
    packed = OX.pack_columns(codes, bits)
    assert packed.shape == (96 * bits // 32, 40)
    assert np.array_equal(OX.unpack_columns(packed, bits), codes)
    # LSB-first bitstream: code 0 of every column sits in the low bits of word 0 (pack_tensor.cu:118-248)
    assert np.array_equal(packed.view(np.uint32)[0] & ((1 << bits) - 1), codes[0])


def test_scale_nibbles_store_value_minus_one():
    s = np.arange(1, 17).reshape(2, 8)
    p = OX.pack_rows_4(s)
    assert np.array_equal(OX.unpack_rows_4(p) + 1, s)


def test_gemm_identity_is_reconstruct():
    t = OX.synth_exl2(192, 32, [(6, 32, 32), (4, 64, 128), (3, 32, 32)], seed=1)
    w = OX.exl2_reconstruct(t)
    assert np.array_equal(OX.gemm_ref(np.eye(192, dtype=np.float16), w).view(np.uint16), w.view(np.uint16))
    # act-order: packed row i holds original feature argsort(invperm)[i]
    t2 = dict(t); t2["q_invperm"] = np.arange(192, dtype=np.int32)
    w2 = OX.exl2_reconstruct(t2)
    perm = np.argsort(t["q_invperm"], kind="stable")
    assert np.array_equal(w[perm], w2)


def test_gptq_act_order_is_a_permutation_of_plain():
    t = OX.synth_gptq(256, 16, 64, seed=2, act_order=True)
    w = OX.gptq_reconstruct(t)
    xm, xi = OX.gptq_sequential_perm(t["g_idx"], 4)
    assert np.array_equal(np.sort(xm), np.arange(256)) and np.array_equal(xm[xi], np.arange(256))
    assert np.all(np.diff(t["g_idx"][xm]) >= 0)          # sorted by group, stable


def test_q4_cache_is_near_inverse():
    rng = np.random.default_rng(3)
    x = rng.standard_normal(2048).astype(np.float16)
    p, s = OM.q4_pack(x)
    y = OM.q4_unpack(p, s)
    assert p.shape == (1024,) and s.shape == (64,)
    err = np.abs(y.astype(np.float32) - x.astype(np.float32))
    assert err.max() < 0.6 and err.mean() < 0.12
    # all-zero block: absmax = 0 -> NaN codes -> 0 (cache_q.cuh:54-60 with __half2int_rn(NaN) == 0) -> decodes to 0
    p0, s0 = OM.q4_pack(np.zeros(512, dtype=np.float16))
    assert np.all(OM.q4_unpack(p0, s0) == 0)


def test_wht_involution():
    rng = np.random.default_rng(4)
    w = rng.standard_normal((2, 32, 2)).astype(np.float16)
    back = OM._wht32_pairs(OM._wht32_pairs(w)).astype(np.float32) / 32.0
    assert np.max(np.abs(back - w.astype(np.float32))) < 2e-2
