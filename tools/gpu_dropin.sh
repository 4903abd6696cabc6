#!/bin/bash
# GPU call for the drop-in route: parity tests of the compiled binding / module chain, then the unmodified reference host's decode loop
# (tools/dropin_decode_bench.py) on the three binding states, a kernel trace and a host profile of the fast one.
#   gpurun --timeout 900 -- 'TAG=r05g tools/gpu_dropin.sh'
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; R=$ROOT/gpurun_out; T=${TAG:-r05g}
cd $ROOT
echo "== pytest -m gpu (drop-in)"
timeout -k 10 900 python -m pytest ${TESTS:-tests/test_dropin_fast.py tests/test_dropin_reference.py tests/test_abi.py tests/test_qattn_op.py} -m gpu -q --timeout 600 -x > $R/${T}_pytest_dropin.log 2>&1; echo "rc=$?"; tail -5 $R/${T}_pytest_dropin.log
for rep in $(seq 1 ${REPS:-2}); do
  echo "== reference host on the drop-in, round $rep"
  echo -n "compiled + chain : "; timeout -k 10 300 python tools/dropin_decode_bench.py --tokens ${TOKENS:-256} 2>$R/${T}_dropin_fast.err | tee $R/${T}_dropin_fast.json | cut -c1-900
  echo -n "compiled, no chain: "; EXL2_MODULE_CHAIN=0 timeout -k 10 300 python tools/dropin_decode_bench.py --tokens ${TOKENS:-256} 2>/dev/null | tee $R/${T}_dropin_nochain.json | cut -c100-260
  echo -n "ctypes           : "; EXL2_DROPIN_FAST=0 timeout -k 10 300 python tools/dropin_decode_bench.py --tokens ${TOKENS:-256} 2>/dev/null | tee $R/${T}_dropin_ctypes.json | cut -c100-260
done
tail -3 $R/${T}_dropin_fast.err
echo "== reference ExLlamaV2DynamicGenerator (paged) on the drop-in: ${GEN_JOBS:-8} greedy jobs x ${GEN_TOKENS:-96} tokens"
echo -n "compiled + chain : "; timeout -k 10 400 python tools/dropin_decode_bench.py --generator ${GEN_JOBS:-8} --tokens ${GEN_TOKENS:-96} 2>$R/${T}_dropin_gen.err | tee $R/${T}_dropin_gen_fast.json | cut -c1-700
echo -n "ctypes           : "; EXL2_DROPIN_FAST=0 timeout -k 10 400 python tools/dropin_decode_bench.py --generator ${GEN_JOBS:-8} --tokens ${GEN_TOKENS:-96} 2>/dev/null | tee $R/${T}_dropin_gen_ctypes.json | cut -c1-330
tail -3 $R/${T}_dropin_gen.err
if [ "${PROFILE:-1}" = 1 ]; then
echo "== host profile (compiled + chain)"
timeout -k 10 300 python tools/dropin_decode_bench.py --tokens 64 --profile > /dev/null 2> $R/${T}_dropin_host_profile.txt; head -30 $R/${T}_dropin_host_profile.txt | cut -c1-160
echo "== rocprofv3 kernel stats (compiled + chain)"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_dropin -o $T -- python $ROOT/tools/dropin_decode_bench.py --tokens 128 > $R/${T}_rocprof_dropin.log 2>&1); echo "rc=$?"
f=$(find $R/prof_dropin -name "${T}_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/${T}_dropin_kernel_stats.csv && head -12 $f | cut -c1-170
rm -rf $R/prof_dropin
fi
echo "== done"
