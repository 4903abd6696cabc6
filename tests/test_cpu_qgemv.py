"""oracle/cpu_qgemv.c -- the multi-threaded C restatement of the EXL2 decode GEMV that bench.py times as the CPU baseline
(variant A: dequantize on the fly from the packed on-disk tensors) -- against the numpy oracle.  Test infrastructure checking
test infrastructure: the C port must compute y = x[q_perm] . W with W = (code - 2^(b-1)) * scale exactly as oracle/exl2.py
decodes it (float64 restatement here, 1e-5), and sit within the fp16 bar of x @ reconstruct() (which rounds each weight to fp16)."""
import numpy as np
import pytest
import torch

from exllamav2_amd.synth import synth_linear
from oracle import exl2 as OX

CASES = [(([2], [1.0], 32), 64, 16), (([3], [1.0], 64), 128, 24), (([4], [1.0], 128), 256, 64), (([5], [1.0], 32), 96, 8),
         (([6], [1.0], 128), 256, 96), (([8], [1.0], 32), 64, 40), (([5, 4], [0.1, 0.9], 128), 512, 256),
         (([3, 2], [0.3, 0.7], 64), 256, 64), (([8, 4], [0.05, 0.95], [32, 128]), 1024, 128), (([4, 3], [0.5, 0.5], 128), 4096, 1024)]


@pytest.fixture(scope="module")
def qgemv():
    try:
        from oracle import cpu_qgemv
        cpu_qgemv.build()
    except Exception as e:                                     # no gcc on this host
        pytest.skip(f"oracle/cpu_qgemv.c not buildable here: {e}")
    return cpu_qgemv


def _exact(t):
    """float64: sum_k x[perm[k]] * (code - zero) * scale, scales as the kernels decode them"""
    qw = t["q_weight"]
    groups = OX.group_table(t["q_groups"], qw.shape[0])
    scales = OX.exl2_scales(t["q_scale"], OX.exl2_prescale_scale_max(t["q_scale_max"], 1.0)).astype(np.float64)
    rows = []
    for gi, (bits, q0, r) in enumerate(groups):
        codes = OX.unpack_columns(qw[q0:q0 + r * bits // 32], bits).astype(np.float64) - (1 << (bits - 1))
        rows.append(codes * scales[gi][None, :])
    return np.concatenate(rows, axis=0)                      # packed row order


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_cpu_qgemv_equals_the_oracle_decode(qgemv, threads):
    gen = torch.Generator(device="cpu"); gen.manual_seed(7)
    try:
        pool = qgemv.Pool(threads, 4096)
    except RuntimeError as e:
        pytest.skip(str(e))
    try:
        for recipe, k, n in CASES:
            for act_order in (True, False):
                w = synth_linear(k, n, recipe, "cpu", gen, act_order=act_order)
                t = {kk: vv.numpy() for kk, vv in w.items()}
                m = qgemv.Matrix(t)
                x = np.random.default_rng(k + n).standard_normal(k).astype(np.float32)
                y = pool.gemv(m, x)
                perm = np.argsort(t["q_invperm"].astype(np.int64), kind="stable")
                want = x.astype(np.float64)[perm] @ _exact(t)
                assert np.abs(y - want).max() <= 1e-5 * max(1.0, np.abs(want).max()) + 1e-5, (recipe, k, n, act_order)
                recons = x.astype(np.float64) @ OX.exl2_reconstruct({a: b for a, b in t.items() if a != "q_perm"}).astype(np.float64)
                assert np.abs(y - recons).max() <= np.abs(recons).max() * 2.0 ** -10 + 1.5e-3, (recipe, k, n)
                assert np.array_equal(pool.gemv(m.clone(), x), y)          # a clone in its own memory is the same matrix
    finally:
        pool.close()


def test_bench_cpu_baseline_child_prints_one_json_object_with_both_variants():
    """`bench.py --cpu-baseline-only` (what the main bench run starts as a child process so that an out-of-memory kill or a hang
    in the CPU leg cannot cost the GPU line): one JSON object, variant B at the top level, variant A (the C port) beside it."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-baseline-only", "--model", "tiny"],
                       capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-400:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["kind"] == "port" and d["unit"] == "tokens/s" and d["value"] > 0 and d["cores"] >= 1 and "UNSAMPLED" in d["sample"]
    a = d["variant_a"]
    assert a.get("error") is None and a["value"] > 0 and "oracle/cpu_qgemv.c" in a["sample"]
