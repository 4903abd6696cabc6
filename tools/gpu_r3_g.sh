#!/bin/bash
# Round 3, GPU call G: instruction accounting of the lean kernel -- SQ instruction counters per wave for the library and its
# kill-switch variants (LEAN_KILL: 1 no decode, 2 no norm arithmetic, 4 no epilogue; 7 all three).
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
P="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS"
for lib in ${LIBS:-default kill1 kill2 kill4 kill7}; do
  L=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip_$lib.so; [ "$lib" = default ] && L=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip.so
  (cd /tmp && EXL2_HIP_LIB=$L EXL2_LEAN_OCC=${OCC:-8} timeout -k 10 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/pmc_g_$lib -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-prefill --no-ctx-window --no-parity-check --no-graph > $R/r03g_pmc_$lib.log 2>&1)
  python - $lib <<'PY'
import csv, glob, collections, json, sys
lib = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(f"gpurun_out/pmc_g_{lib}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "qgemv_lean" in k:
            agg[(k[24:52], r["Counter_Name"])][0] += 1; agg[(k[24:52], r["Counter_Name"])][1] += float(r["Counter_Value"])
out = {}
for (k, c), (n, v) in sorted(agg.items()): out.setdefault(k, {})[c] = v / n
for k, d in out.items():
    w = d.get("SQ_WAVES", 1)
    print("%-8s %-30s per wave: VALU %4.0f SALU %4.0f SMEM %4.1f LDS %4.1f  wave-cycles(x4) %5.0f wait %3.0f%%" % (lib, k, d["SQ_INSTS_VALU"] / w, d["SQ_INSTS_SALU"] / w, d["SQ_INSTS_SMEM"] / w, d["SQ_INSTS_LDS"] / w, d["SQ_WAVE_CYCLES"] / w, 100 * d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]))
PY
  rm -rf $R/pmc_g_$lib
done
