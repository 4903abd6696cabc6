#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
echo "== microbench"; timeout 900 python tools/microbench.py > gpurun_out/microbench.log 2>&1; echo "micro rc=$?"; cat gpurun_out/microbench.log | grep -v amdgpu.ids | cut -c1-200
echo "== microbench split sweep"; for s in 2 4 8 16; do EXL2_GEMV_SPLIT=$s timeout 300 python tools/microbench.py --quick 2>&1 | grep -v "amdgpu.ids\|copy" | cut -c1-160 | sed "s/^/S$s /"; done > gpurun_out/microbench_split.log; cat gpurun_out/microbench_split.log
echo "== bench"; timeout 900 python bench.py --steps 64 --warmup 8 > gpurun_out/bench1.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench1.log | cut -c1-1200
echo "== rocprof"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof5 -o run5 -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof5.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof5 -name "*kernel_stats*" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-160
python - <<'PY'
import csv,collections,glob
f=glob.glob('gpurun_out/prof5/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
# per-kernel durations by grid size for the stream kernel (last 2000 launches = decode steps)
d=collections.defaultdict(list)
for r in rows[-4000:]:
    if 'qgemv_stream' in r['Kernel_Name']:
        d[(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), r.get('Workgroup_Size_X', r.get('Workgroup_Size')))].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items()): print('grid',k,'n',len(v),'avg_us',round(sum(v)/len(v)/1e3,2),'min_us',min(v)/1e3)
print(list(rows[0].keys()))
PY
rm -f gpurun_out/prof5/*kernel_trace.csv 2>/dev/null
