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


def test_flash_attn_shim_window_that_cannot_clip_is_causal_attention_and_one_that_could_is_refused():
    """attn.py:590-594: Mistral-family checkpoints pass window_size = (W, W).  The shim decides from host-side sizes only."""
    import importlib.util, sys, types
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("flash_attn_shim_w", os.path.join(root, "dropin", "flash_attn", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    q = torch.zeros((1, 1, 2, 64), dtype=torch.float16)
    k = torch.zeros((1, 9, 1, 64), dtype=torch.float16)
    with pytest.raises(NotImplementedError, match="would clip"):
        mod.flash_attn_func(q, k, k, causal=True, window_size=(4, 4))
    with pytest.raises(NotImplementedError, match="softcap"):
        mod.flash_attn_func(q, k, k, causal=True, softcap=30.0)
    # a window of 4096 over 9 keys cannot clip: the call goes on to the kernels (which refuse CPU tensors -- there is no CPU path)
    with pytest.raises(RuntimeError, match="no CPU path|HIP device"):
        mod.flash_attn_func(q, k, k, causal=True, window_size=(4096, 4096))
