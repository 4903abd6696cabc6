for v in "" "EXL2_ATT_RB=2" "EXL2_ATT_RB=1"; do
  echo "== mixtral b1 $v"; env $v python bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 32 --warmup 4 --headline-only --no-parity-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['windows']['tokens_per_s'])"
done
for v in "" "EXL2_ATT_RB=2" "EXL2_ATT_RB=1"; do
  echo "== mixtral b1 ctx 1920 $v"; env $v python bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 32 --warmup 4 --headline-only --no-parity-check --ctx 1920 2>&1 | tail -1 | cut -c1-300
done
