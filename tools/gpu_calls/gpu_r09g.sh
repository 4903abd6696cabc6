for b in 2 3; do for v in "" "EXL2_MOE_NO_LEAN=1"; do
  echo "== mixtral b$b $v"; env $v python bench.py --model mixtral-8x7b --recipe 3.5bpw --batch $b --steps 32 --warmup 4 --headline-only --no-parity-check 2>gpurun_out/r09g_err.txt | tail -1 | cut -c1-140; tail -2 gpurun_out/r09g_err.txt | cut -c1-300
done; done
