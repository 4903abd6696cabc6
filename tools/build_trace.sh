#!/bin/bash
# EXL2_TRACE build of the library (in-kernel timestamps for tools/trace_gemv.py); not used by the product path
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DEXL2_TRACE \
  -I exllamav2_amd/csrc -o exllamav2_amd/libexl2_hip_trace.so exllamav2_amd/csrc/*.hip
