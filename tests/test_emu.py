"""The CPU emulation harness itself (tests/emu: TEST INFRASTRUCTURE).  Its one-workgroup-at-a-time mode is exercised by every
`emu` parity test; this file covers the co-resident mode (`emu_launch_coop`): all workgroups of a small grid alive at once, so
that kernels whose workgroups wait for each other INSIDE one launch -- persistent multi-phase kernels with grid-wide hand-offs,
DESIGN.md section 8 -- can be checked without a GPU."""
import ctypes

import pytest

from tests.conftest import build_emu_if_needed


@pytest.mark.parametrize("n_wg,n_phases,block", [(1, 2, 64), (2, 3, 128), (4, 6, 256), (8, 4, 1024), (5, 3, 320)])
def test_coresident_grid_with_grid_wide_handoffs(n_wg, n_phases, block):
    """A persistent multi-phase kernel: every workgroup reduces a value over its waves (workgroup barrier + dynamic LDS), writes
    it, arrives at a counter, the last arrival publishes the phase number, one thread per workgroup polls it (emu_spin_yield),
    then every workgroup checks what the OTHERS wrote in that phase.  0 wrong reads, and no deadlock / livelock abort."""
    lib = ctypes.CDLL(build_emu_if_needed())
    lib.emu_selftest_coop.argtypes = [ctypes.c_int] * 3
    lib.emu_selftest_coop.restype = ctypes.c_int
    assert lib.emu_selftest_coop(n_wg, n_phases, block) == 0
