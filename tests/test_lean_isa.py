"""The request / wait structure of the pipelined decode kernel (csrc/qgemv_lean.hip: head) as the COMPILER emitted it.

The kernel's point is that the requests of a wave's last NB items stay in flight while its first items are decoded: they are
issued behind a "fence load" (hw.h) and the compiler must answer the fence's use with a COUNTED wait, `s_waitcnt vmcnt(n)`,
n = the vector-memory instructions it placed behind the fence load.  Two ways this silently degrades to "wait for everything"
were met while building it -- a pending global_load_lds (the compiler then answers every wait with vmcnt(0)) and a loop that
issues vector-memory operations ahead of the region (the compiler then drains its guess of what is pending before the next
register reuse, i.e. in FRONT of the requests) -- and neither changes a result, so no parity test can see them.  This test reads
the gfx950 assembly: hipcc cross-compiles without a GPU.
"""
import os
import re
import shutil
import subprocess

import pytest

from tests.conftest import ROOT


_CACHE = {}


def _asm(tmp_path):
    if "s" not in _CACHE:
        _CACHE["s"] = _compile(tmp_path)
    return _CACHE["s"]


def _compile(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "exllamav2_amd", "csrc")
    src = os.path.join(csrc, "qgemv_lean.hip")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-pass-failed", "-I", csrc, "--cuda-device-only", "-S"]
    # the assembly of the 60-odd instantiations takes minutes to make: kept under tests/.isa_cache (git-ignored), keyed by the CONTENT
    # of the kernel source, every header beside it, the flags and the compiler -- a changed byte anywhere recompiles
    import hashlib
    h = hashlib.sha1(" ".join(flags).encode())
    h.update(subprocess.run([hipcc, "--version"], capture_output=True).stdout)
    for f in [src] + sorted(os.path.join(csrc, n) for n in os.listdir(csrc) if n.endswith(".h")):
        h.update(open(f, "rb").read())
    cache_dir = os.path.join(ROOT, "tests", ".isa_cache")
    cached = os.path.join(cache_dir, f"lean_{h.hexdigest()[:20]}.s")
    if os.path.isfile(cached):
        return open(cached).read()
    out = str(tmp_path / "lean.s")
    subprocess.check_call([hipcc] + flags + [src, "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read()
    try:
        os.makedirs(cache_dir, exist_ok=True)
        for n in os.listdir(cache_dir):
            if n.startswith("lean_") and n.endswith(".s"):
                os.remove(os.path.join(cache_dir, n))
        tmp = cached + f".{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            f.write(text)
        os.replace(tmp, cached)
    except OSError:
        pass
    return text


def test_pipelined_regions_keep_their_last_requests_in_flight(tmp_path):
    s = _asm(tmp_path)
    kernels = list(re.finditer(r"^(_Z17qgemv_lean_kernel\w+):", s, re.M))
    assert len(kernels) >= 10                                       # 5 geometries x {EXL2, GPTQ}
    checked = 0
    for m in kernels:
        body = s[m.end():s.index(".Lfunc_end", m.end())].splitlines()
        for i, line in enumerate(body):
            mk = re.search(r"; lean mark (0x[0-9a-f]+|\d+)", line)
            if not mk:
                continue
            ident = int(mk.group(1), 0)
            if ident % 10 != 2 or ident >= 99000:                  # (...2 = the marker in front of the fence load)
                continue
            nb = (ident % 1000) // 10
            # the fence load, then the requests behind it, then the first wait on the vector-memory counter
            j = i + 1
            while not body[j].strip().startswith("global_load_dword "):
                assert j < i + 12, (m.group(1), ident, "fence load not found behind its marker")
                j += 1
            loads, k = 0, j + 1
            while True:
                t = body[k].strip()
                if t.startswith("s_waitcnt") and "vmcnt" in t:
                    break
                if t.startswith(("global_load", "buffer_load")):
                    assert " lds" not in t, (m.group(1), ident, "a staging copy behind the fence load")
                    loads += 1
                assert k < j + 200, (m.group(1), ident, "no wait behind the fence load")
                k += 1
            wait = int(re.search(r"vmcnt\((\d+)\)", body[k]).group(1))
            xmem = m.group(1).endswith("Lb0ELb0ELb0ELb1ELb0EEv8LeanArgs")      # <..., ROWS, WALK, DEP, XMEM = true, LOADS>
            if nb == 0 and xmem:
                # XMEM form: the A operands of the first three items (4 x 16 bytes per lane each) are requested behind the fence load
                # and stay in flight across its wait
                # (where the compiler placed them directly behind the fence load; on some paths it first waits for a register)
                if loads:
                    assert loads == 12 and wait == 12, (m.group(1), ident, loads, wait)
            elif nb == 0:
                assert loads == 0 and wait == 0, (m.group(1), ident, loads, wait)
            else:
                # every one of the NB items' loads (1-2 instructions per item) sits between the fence load and the wait, and the wait
                # leaves exactly those in flight
                assert nb <= loads <= 2 * nb, (m.group(1), ident, nb, loads)
                assert wait == loads, (m.group(1), ident, "the wait behind the fence load drains requests that were meant to stay in flight", loads, wait)
            checked += 1
    assert checked >= 40, checked


def test_no_global_load_lds_in_the_lean_kernel(tmp_path):
    """with one pending the compiler answers every later wait with vmcnt(0) (hw.h: dma_buf_to_lds16)"""
    s = _asm(tmp_path)
    for m in re.finditer(r"^(_Z17qgemv_lean_kernel\w+):", s, re.M):
        body = s[m.end():s.index(".Lfunc_end", m.end())]
        assert "global_load_lds" not in body, m.group(1)
        assert "buffer_load_dwordx4" in body and " lds" in body


def test_ring_loads_keep_a_whole_register_load_in_flight(tmp_path):
    """Shares of several register loads (qgemv_lean.hip: ring_passes; 2 / 3 / 4-bit items).  The first two loads are straight-line
    code: item q of the second load is requested right behind the decode of item q of the first, and decoded behind a wait that
    leaves the younger requests in flight.  Read off the instruction stream per bit width (the ring's requests are the only
    non-temporal loads of D consecutive items that are followed by counted waits):
      * 8 waves, one tile (two loads at most, no third request): D requests, then the waits D - 1, D - 2, ... 0;
      * 16 waves (up to four loads): D requests, then D times (wait D - 1, request) -- never a vmcnt(0) between them.
    A vmcnt(0) in front of the second load's first item is the round-4 behaviour: one exposed round trip per load."""
    s = _asm(tmp_path)

    def events(name):
        m = re.search(r"^" + name + r":", s, re.M)
        body = s[m.end():s.index(".Lfunc_end", m.end())].splitlines()
        ev = []
        for line in body:
            t = line.strip()
            ld = re.match(r"global_load_dword(x\d)? ", t)
            if ld and " nt" in t:
                ev.append("L" + (ld.group(1) or "x1")[1:])
            elif t.startswith("s_waitcnt") and "vmcnt" in t:
                ev.append("W" + re.search(r"vmcnt\((\d+)\)", t).group(1))
        return " ".join(ev)

    e8 = events("_Z17qgemv_lean_kernelILb0ELi8ELi1ELb0ELi6ELb0ELb0ELb0ELb0ELb0EEv8LeanArgs")
    e16 = events("_Z17qgemv_lean_kernelILb0ELi16ELi1ELb0ELi6ELb0ELb0ELb0ELb0ELb0EEv8LeanArgs")
    for width, depth in (("4", 6), ("2", 10), ("3", 8)):
        req = " ".join(["L" + width] * depth)
        down = " ".join("W%d" % i for i in range(depth - 1, -1, -1))
        assert req + " " + down in e8, (width, e8)
        ring = " ".join(["W%d L%s" % (depth - 1, width)] * (depth - 1))
        assert req + " " + ring in e16, (width, e16)
    # round 6: the several-loads pair (70B gate|up), the LOADS form (70B down_proj) and the MoE pair that sums two experts' down
    # projections carry the same ring (depth of the 4-wave geometry: 8 items of 3 bits, 10 of 2)
    for name, widths in (("_Z17qgemv_lean_kernelILb0ELi4ELi2ELb1ELi6ELb0ELb1ELb0ELb0ELb0EEv8LeanArgs", (("2", 10), ("3", 8))),
                         ("_Z17qgemv_lean_kernelILb0ELi8ELi2ELb0ELi4ELb1ELb1ELb0ELb0ELb1EEv8LeanArgs", (("3", 8),)),
                         ("_Z21qgemv_lean_moe_kernelILi8ELi2ELb1ELi6ELb1EEvPK8LeanArgs7LeanDyn", (("4", 6), ("3", 8)))):
        ev = events(name)
        for width, depth in widths:
            ring = " ".join(["W%d L%s" % (depth - 1, width)] * (depth - 1))
            assert " ".join(["L" + width] * depth) + " " + ring in ev, (name, width, ev)
    e82 = events("_Z17qgemv_lean_kernelILb0ELi8ELi2ELb1ELi6ELb0ELb0ELb0ELb0ELb0EEv8LeanArgs")           # the pair geometry: no such shares
    assert " ".join(["W9 L2"] * 3) not in e82 and " ".join(["L2"] * 10) + " W9 W8 W7" not in e82
