#!/bin/bash
# Builds the stand-alone probes for gfx950 (hipcc cross-compiles without a GPU).  The binaries are git-ignored but travel to
# the GPU box with the gpurun snapshot: run this in the build container before a GPU call that uses them
# (tools/gpu_run.sh STAGES=probes runs them).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for p in chain_probe fork_probe mall_probe persist_probe tr_probe kernarg_probe ifetch_probe; do
    if [ ! -x $p ] || [ $p.hip -nt $p ]; then $HIPCC --offload-arch=gfx950 -O3 $p.hip -o $p && echo "built $p"; fi
done
# probes that use the library's decoders
for p in lean_probe engine_probe; do
    if [ ! -x $p ] || [ $p.hip -nt $p ]; then $HIPCC --offload-arch=gfx950 -O3 -I ../../exllamav2_amd/csrc $p.hip -o $p && echo "built $p"; fi
done
# kernel-argument preload on / off: the same source with and without the backend option
if [ ! -x preload_on_probe ] || [ preload_probe.hip -nt preload_on_probe ]; then
    $HIPCC --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 preload_probe.hip -o preload_on_probe && echo "built preload_on_probe"
    $HIPCC --offload-arch=gfx950 -O3 preload_probe.hip -o preload_off_probe && echo "built preload_off_probe"
fi
