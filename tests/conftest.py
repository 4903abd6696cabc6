"""Test configuration.

Two backends drive the same parity tests through the same C ABI and the same host mirror (`exllamav2_amd.ext.ExtC`):

* ``hip`` -- the product: libexl2_hip.so (gfx950) on cuda:0.  Marked ``gpu``.
* ``emu`` -- tests/emu: the SAME kernel + host sources compiled for the CPU with an emulation of the wave64 primitives
  (TEST INFRASTRUCTURE; exercises layout / indexing / host logic on tiny shapes without a GPU).  Never a product path.

The checker is always the numpy oracle under oracle/.
"""
import os
import subprocess
import sys

import pytest

# the library's per-process switches (csrc/modules.hip: ENV_SET) are re-read on every call in the test processes: tests flip them in place
os.environ.setdefault("EXL2_ENV_DYNAMIC", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMU_LIB = os.path.join(ROOT, "tests", "emu", "libexl2_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _emu_sources():
    src = [os.path.join(ROOT, "tests", "emu", f) for f in ("hw_emu.h", "emu_runtime.cpp", "build_emu.sh")]
    d = os.path.join(ROOT, "exllamav2_amd", "csrc")
    src += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".h"))]
    return src


def build_emu_if_needed():
    newest = max(os.path.getmtime(f) for f in _emu_sources())
    if not os.path.exists(EMU_LIB) or os.path.getmtime(EMU_LIB) < newest:
        subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], cwd=ROOT,
                              stdout=subprocess.DEVNULL)
    return EMU_LIB


class Backend:
    def __init__(self, name):
        import torch
        from exllamav2_amd import _lib
        from exllamav2_amd.ext import ExtC
        self.name = name
        self.torch = torch
        if name == "emu":
            self.ext = ExtC(_lib.Lib(build_emu_if_needed()), allow_cpu=True)
            self.device = "cpu"
        else:
            if not torch.cuda.is_available():
                pytest.skip("no GPU visible")
            self.ext = ExtC()          # binds libexl2_hip.so; raises if it is not built
            self.ext.lib
            self.device = "cuda:0"

    def t(self, arr):
        """numpy -> torch tensor on the backend's device (copy)."""
        import numpy as np
        return self.torch.from_numpy(np.ascontiguousarray(arr).copy()).to(self.device)

    def n(self, tensor):
        if self.device != "cpu":
            self.torch.cuda.synchronize()
        return tensor.detach().cpu().numpy()

    @property
    def is_emu(self):
        return self.name == "emu"


_backends = {}


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def be(request):
    name = request.param
    if name not in _backends:
        _backends[name] = Backend(name)
    return _backends[name]
