#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== probe"; timeout 900 python tools/probe.py > gpurun_out/probe.log 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids gpurun_out/probe.log | cut -c1-220
echo "== rocprof trace columns"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof6 -o run6 -- python $GRAFT_REPO_ROOT/tools/microbench.py --quick > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,collections,glob
f=glob.glob('gpurun_out/prof6/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
seen=set()
for r in rows:
    if 'qgemv' in r['Kernel_Name']:
        key=(r['Kernel_Name'][:40],r['Grid_Size_X'],r['Workgroup_Size_X'],r['LDS_Block_Size'],r['Scratch_Size'],r['VGPR_Count'],r['Accum_VGPR_Count'],r['SGPR_Count'])
        if key not in seen:
            seen.add(key); print(key)
PY
rm -rf gpurun_out/prof6
