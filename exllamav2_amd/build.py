"""Builds the C-ABI HIP library for gfx950 in-tree (exllamav2_amd/libexl2_hip.so).  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libexl2_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h"))
    return os.path.getmtime(OUT) < max(os.path.getmtime(f) for f in deps)


def _obj_dir() -> str:
    d = os.path.join(HERE, "build")                      # git-ignored; objects are a cache, only the .so is loaded
    os.makedirs(d, exist_ok=True)
    return d


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per .hip file (compiled in parallel, re-compiled only when the file or a header is newer), one link."""
    if not force and not needs_build():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-I", CSRC]
    headers = glob.glob(os.path.join(CSRC, "*.h"))
    hdr_time = max(os.path.getmtime(f) for f in headers) if headers else 0.0
    objs, jobs = [], []
    for src in sources():
        obj = os.path.join(_obj_dir(), os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append([hipcc] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 4))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))


# ---- the compiled half of the drop-in (dropin/_exl2_fast.cpp): pybind11 over the C ABI ---------------------------------------
FAST_SRC = os.path.join(os.path.dirname(HERE), "dropin", "_exl2_fast.cpp")
FAST_OUT = os.path.join(os.path.dirname(HERE), "dropin", "_exl2_fast.so")


def build_fast(force: bool = False, verbose: bool = False) -> str:
    """g++ on one host-only source (no device code: it dlopens libexl2_hip.so); in-tree like the library, so it travels to the GPU box."""
    if not force and os.path.exists(FAST_OUT) and os.path.getmtime(FAST_OUT) >= os.path.getmtime(FAST_SRC):
        return FAST_OUT
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", FAST_SRC, "-o", FAST_OUT, "-DTORCH_EXTENSION_NAME=_exl2_fast",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-I/opt/rocm/include", "-I" + sysconfig.get_paths()["include"]] + ["-I" + i for i in ce.include_paths()] + \
          ["-L" + libdir, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_python", "-ldl", "-Wl,-rpath," + libdir]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return FAST_OUT
