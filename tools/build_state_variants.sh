#!/bin/bash
# Same-box A/B of SOURCE STATES of the lean decode kernel (profiles/history/r04_states_ab*.txt): builds exllamav2_amd/libexl2_hip_<name>.so =
# the current library with qgemv_lean.o compiled from another commit's csrc/qgemv_lean.hip (against the current headers), so that
# the libraries differ in that one object only.  EXL2_HIP_LIB=<path> makes exllamav2_amd/_lib.py load one of them.
# usage: tools/build_state_variants.sh [name:commit ...]      default: callb:549e139 (pipelined form, before ROWS / overlap support)
#                                                                      r3:d352c7c   (end of round 3)
set -e
cd "$(dirname "$0")/.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
python -c "from exllamav2_amd import build; build.build()" > /dev/null
tmp=$(mktemp -d)
for v in ${@:-callb:549e139 r3:d352c7c}; do
  n=${v%%:*}; c=${v##*:}
  git show $c:exllamav2_amd/csrc/qgemv_lean.hip > $tmp/qgemv_lean_$n.hip
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I exllamav2_amd/csrc -c $tmp/qgemv_lean_$n.hip -o $tmp/lean_$n.o
  objs=$(ls exllamav2_amd/build/*.o | grep -v "_variant_" | grep -v "/qgemv_lean.o")
  $HIPCC --offload-arch=gfx950 -fPIC -shared -o exllamav2_amd/libexl2_hip_$n.so $objs $tmp/lean_$n.o
  echo "built exllamav2_amd/libexl2_hip_$n.so (qgemv_lean.hip of $c)"
done
rm -rf $tmp
