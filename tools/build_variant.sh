#!/bin/bash
# Builds exllamav2_amd/libexl2_hip_<name>.so = the product library with one source file (SRC=..., default qgemv_lean) recompiled under extra -D flags
# (tuning aid for same-box A/B runs: EXL2_LIB_VARIANT=<name> makes exllamav2_amd/_lib.py load it; never the default).
# usage: tools/build_variant.sh <name> [-DFLAG ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
SRC=${SRC:-qgemv_lean}            # which csrc/<SRC>.hip is recompiled under the extra flags
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
python -c "from exllamav2_amd import build; build.build()" > /dev/null
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I exllamav2_amd/csrc "$@" -c exllamav2_amd/csrc/$SRC.hip -o exllamav2_amd/build/${SRC}_variant_$name.o
objs=$(ls exllamav2_amd/build/*.o | grep -v "_variant_" | grep -v "/$SRC.o" )
$HIPCC --offload-arch=gfx950 -fPIC -shared -o exllamav2_amd/libexl2_hip_$name.so $objs exllamav2_amd/build/${SRC}_variant_$name.o
echo "built exllamav2_amd/libexl2_hip_$name.so"
