"""The C ABI: every symbol include/exl2_hip.h declares is exported by the built libraries and bound by the host layer."""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT, build_emu_if_needed


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "exl2_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(exl2_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from exllamav2_amd import _lib
    assert declared_symbols() == sorted(_lib.PROTOTYPES)


def test_emu_library_exports_every_symbol():
    dll = ctypes.CDLL(build_emu_if_needed())
    for s in declared_symbols():
        assert hasattr(dll, s), s


def test_hip_library_exports_every_symbol():
    """Built by __graft_entry__.build() (hipcc cross-compiles here); no compute call is made without a GPU."""
    from exllamav2_amd import _lib, build
    path = build.build()
    lib = _lib.Lib(path)
    assert lib.exl2_abi_version() == 1
    dll = ctypes.CDLL(path)
    for s in declared_symbols():
        assert hasattr(dll, s), s


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: the product operator surface rejects non-device tensors loudly."""
    import torch
    from exllamav2_amd.ext import ExtC
    ext = ExtC()
    with pytest.raises(RuntimeError, match="no CPU path"):
        ext.rms_norm(torch.zeros((1, 64), dtype=torch.float16), torch.ones((64,), dtype=torch.float16),
                     torch.zeros((1, 64), dtype=torch.float16), 1e-5)


def test_missing_library_is_loud(tmp_path):
    from exllamav2_amd import _lib
    with pytest.raises(_lib.Exl2Error, match="no CPU fallback"):
        _lib.Lib(str(tmp_path / "libexl2_hip.so"))


def test_flash_attn_shim_signature():
    """dropin/flash_attn exposes what the reference probes and calls (attn.py:35-59, 602-613)."""
    import importlib.util, inspect, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropin", "flash_attn", "__init__.py")
    spec = importlib.util.spec_from_file_location("flash_attn_shim", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert tuple(int(x) for x in mod.__version__.split(".")) >= (2, 5, 7)
    params = inspect.signature(mod.flash_attn_with_kvcache).parameters
    for name in ("q", "k_cache", "v_cache", "k", "v", "cache_seqlens", "block_table", "causal", "softmax_scale"):
        assert name in params



def test_unmodified_reference_imports_on_top_of_the_dropin():
    """`import exllamav2` (the reference, untouched, from /root/reference) with dropin/ on sys.path: ext.py:105-109 must
    pick up our `exllamav2_ext` instead of JIT-building its CUDA sources, and every name its pybind module defines
    (ext_bindings.cpp m.def list) must exist on ours -- callable for the hot path, raising NotImplementedError by name for
    what SURVEY.md scopes out.  Runs in a subprocess (the reference pins torch threads and patches globals at import)."""
    import re
    import subprocess
    import sys
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "exllamav2")):
        pytest.skip("reference tree not present on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(ref, "exllamav2", "exllamav2_ext", "ext_bindings.cpp")).read()
    names = sorted(set(re.findall(r'^\s*m\.def\("([A-Za-z0-9_]+)"', src, flags=re.M)))       # commented-out defs excluded
    assert len(names) > 60
    code = (
        "import exllamav2, exllamav2.ext as e, json, sys\n"
        "m = e.ext_c\n"
        "assert m.__name__ == 'exllamav2_ext' and m.__file__.endswith('dropin/exllamav2_ext.py'), m.__file__\n"
        f"names = {names!r}\n"
        "missing = [n for n in names if not callable(getattr(m, n, None))]\n"
        "print(json.dumps(missing))\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "dropin"), root, ref]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd="/tmp", timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    assert json.loads(out.stdout.strip().splitlines()[-1]) == []


def test_flash_attn_shim_window_and_softcap_reach_the_kernels():
    """attn.py:590-600: Mistral / Gemma-family checkpoints pass window_size = (W, W) / softcap.  The shim decides from host-side sizes
    only whether a window can clip; round 6: a clipping window and a cap go to the general kernels (tests/test_ops.py::
    test_flash_attn_shim_window_and_softcap checks the numbers) -- which refuse CPU tensors: there is no CPU path."""
    import importlib.util, sys, types
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("flash_attn_shim_w", os.path.join(root, "dropin", "flash_attn", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    q = torch.zeros((1, 1, 2, 64), dtype=torch.float16)
    k = torch.zeros((1, 9, 1, 64), dtype=torch.float16)
    with pytest.raises(RuntimeError, match="no CPU path|HIP device"):
        mod.flash_attn_func(q, k, k, causal=True, window_size=(4, 4))
    with pytest.raises(RuntimeError, match="no CPU path|HIP device"):
        mod.flash_attn_func(q, k, k, causal=True, softcap=30.0)
    with pytest.raises(NotImplementedError, match="without causal"):
        mod.flash_attn_func(q, k, k, causal=False, window_size=(4, 4))
    # a window of 4096 over 9 keys cannot clip: the call goes on to the kernels (which refuse CPU tensors -- there is no CPU path)
    with pytest.raises(RuntimeError, match="no CPU path|HIP device"):
        mod.flash_attn_func(q, k, k, causal=True, window_size=(4096, 4096))


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("EXL2_TEST_SEEDS", "6")))))        # (more seeds: a longer hunt, by hand)
def test_flash_attn_shim_random_calls(be, seed):
    """The flash-attn stand-in as the reference calls it (attn.py:960-977 / 602-613), seeded random shapes: `flash_attn_func` with k / v
    handed over as VIEWS of a longer cache (batch stride = max_seq_len: the shim recovers the whole rows), 1-40 query rows (decode-shaped
    and prefill-shaped kernels), GQA 1-8, head_dim 64 / 128, a window that cannot clip; `flash_attn_with_kvcache` with the append through
    a shuffled block table.  Checker: the oracle's attention; the appended rows must sit in the cache afterwards."""
    import importlib.util
    import numpy as np
    import torch
    from oracle import modules as OM
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(f"flash_attn_shim_r{seed}", os.path.join(root, "dropin", "flash_attn", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod._e = be.ext                                                  # (the emulation build's binding here, libexl2_hip.so's under -m gpu)
    rng = np.random.default_rng(43000 + seed)
    F16 = np.float16
    hd = int(rng.choice([64, 128])); kvh = int(rng.choice([1, 2, 4])); g = int(rng.choice([1, 2, 4, 8])); nh = kvh * g
    b = int(rng.integers(1, 4)); s = int(rng.choice([1, 2, 5, 16, 17, 40])); past = int(rng.choice([0, 3, 100, 255, 300]))
    if s * g > 64 and s <= 16:
        s = 1
    max_seq = past + s + int(rng.integers(0, 50))

    def tol(want):
        return 4e-3 + np.abs(want) * 2.0 ** -7

    q = rng.standard_normal((b, s, nh, hd)).astype(F16)
    kc = (rng.standard_normal((b, max_seq, kvh, hd)) * 0.5).astype(F16); vc = rng.standard_normal((b, max_seq, kvh, hd)).astype(F16)
    kt, vt = be.t(kc), be.t(vc)
    window = (-1, -1) if rng.integers(0, 2) else (8192, 8192)
    out = mod.flash_attn_func(be.t(q), kt[:, :past + s], vt[:, :past + s], causal=True, window_size=window)
    want = OM.attention(q, kc[:, :past + s], vc[:, :past + s])
    err = np.abs(be.n(out).astype(np.float32) - want.astype(np.float32))
    assert np.all(err <= tol(want)), ("flash_attn_func", b, s, nh, kvh, hd, past, float(err.max()))
    # paged, with the append
    ps = 256
    s2 = int(rng.integers(1, 9))
    if s2 * g > 64:
        s2 = 1
    pages_per_seq = int(rng.integers(1, 4))
    pages = b * pages_per_seq + 1
    table = rng.permutation(pages)[:b * pages_per_seq].astype(np.int32).reshape(b, pages_per_seq)
    cap = ps * pages_per_seq
    seqlens = np.array([min(int(rng.choice([0, 5, 250, 255, 256, 400, cap - s2])), cap - s2) for _ in range(b)], dtype=np.int32)
    kp = (rng.standard_normal((pages, ps, kvh, hd)) * 0.5).astype(F16); vp = rng.standard_normal((pages, ps, kvh, hd)).astype(F16)
    q2 = rng.standard_normal((b, s2, nh, hd)).astype(F16)
    kn = (rng.standard_normal((b, s2, kvh, hd)) * 0.5).astype(F16); vn = rng.standard_normal((b, s2, kvh, hd)).astype(F16)
    kp_ref, vp_ref = kp.copy(), vp.copy()
    want2 = OM.paged_attention(q2, kn, vn, kp_ref, vp_ref, seqlens, table)
    kpt, vpt = be.t(kp), be.t(vp)
    out2 = mod.flash_attn_with_kvcache(be.t(q2), kpt, vpt, k=be.t(kn), v=be.t(vn), cache_seqlens=be.t(seqlens), block_table=be.t(table),
                                       causal=True, window_size=(-1, -1) if rng.integers(0, 2) else (4096, 4096))
    assert np.array_equal(be.n(kpt).view(np.uint16), kp_ref.view(np.uint16)) and np.array_equal(be.n(vpt).view(np.uint16), vp_ref.view(np.uint16))
    err2 = np.abs(be.n(out2).astype(np.float32) - want2.astype(np.float32))
    assert np.all(err2 <= tol(want2)), ("flash_attn_with_kvcache", b, s2, nh, kvh, hd, seqlens.tolist(), float(err2.max()))
