for v in "" "EXL2_ATT_RB=2" "EXL2_ATT_RB=1"; do
  echo "== tinyllama gptq $v"; env $v python bench.py --model tinyllama --recipe gptq-4bit-128g --steps 64 --warmup 8 --headline-only --no-parity-check 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['windows']['tokens_per_s'])"
done
for c in 8000 30000; do for v in "" "EXL2_ATT_RB=2"; do
  echo "== mixtral b1 ctx $c $v"; env $v python bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 32 --warmup 4 --headline-only --no-parity-check --ctx $c 2>&1 | tail -1 | cut -c1-130
done; done
for v in "" "EXL2_ATT_RB=2"; do
  echo "== mixtral b4 $v"; env $v python bench.py --model mixtral-8x7b --recipe 3.5bpw --batch 4 --steps 32 --warmup 4 --headline-only --no-parity-check 2>&1 | tail -1 | cut -c1-130
done
